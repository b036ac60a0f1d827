import sqlite3,collections,sys
db=sqlite3.connect(sys.argv[1]); steps=int(sys.argv[2]) if len(sys.argv)>2 else 35
cur=db.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=cur.execute(f"select s.kernel_name, d.end-d.start from {kd} d join {ks} s on d.kernel_id=s.id").fetchall()
agg=collections.defaultdict(list)
for n,t in rows: agg[n.replace('_ZN6lipasr','').replace('void ','').split('(')[0][:50]].append(t)
tot=0
for n,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    per=sum(v)/steps/1e3; tot+=per
    print(f"{n:52s} calls {len(v):5d} avg {sum(v)/len(v)/1e3:7.1f} per-step {per:7.1f}")
print(tot)
