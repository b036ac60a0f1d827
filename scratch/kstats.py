import csv, sys, collections
f=sys.argv[1]; steps=int(sys.argv[2]) if len(sys.argv)>2 else 35
rows=list(csv.DictReader(open(f)))
tot=0
for r in rows[:24]:
    name=r['Name'].replace('lipasr::','').replace('void ','').split('(')[0][:44]
    per=float(r['TotalDurationNs'])/steps/1e3; tot+=per
    print(f"{name:46s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e3:8.1f} us  per-step {per:8.1f} us  {float(r['Percentage']):5.1f}%")
print('sum per-step', tot)
