import sqlite3, sys
db=sqlite3.connect(sys.argv[1]); cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
cols=[r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol='queue_id' if 'queue_id' in cols else cols[0]
rows=cur.execute(f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# steady-state window: take the last 40% of the trace, print ~1.3 ms
t_end=rows[-1][2]; t0=rows[int(len(rows)*0.7)][1]
win=float(sys.argv[2]) if len(sys.argv)>2 else 1300.0
qs={}
for n,s,e,q in rows:
    if s<t0 or (s-t0)/1e3>win: continue
    qi=qs.setdefault(q,len(qs))
    nm=n.replace('_ZN6lipasr','').split('(')[0][:34]
    print(f"{'    '*0}{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  q{qi} {'          '*qi}{nm}")
