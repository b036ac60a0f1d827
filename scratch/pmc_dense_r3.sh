#!/bin/bash
# round 3: matrix-pipe counters of the classifier's GEMM kernels (config 2: pre-extracted features, whole chip)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_dense; mkdir -p $O
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i --output-format csv -- python3 $R/bench.py --steps 12 --warmup 4 --skip-cpu-baseline --skip-b512 --skip-other-configs --pool-clips 16384 --pre-extracted > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'lipasr' in r['Kernel_Name'] and 'gemm' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('lipasr::','')
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for f in sorted(glob.glob("$O/p1/**/*kernel_trace.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'lipasr' in r['Kernel_Name'] and 'gemm' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('lipasr::','')
            dur[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
out={}
for k,d in acc.items():
    out[k]={c:sum(v)/len(v) for c,v in d.items()}
    out[k]['calls']=len(dur.get(k,[])); out[k]['avg_us_under_counters']=sum(dur[k])/max(1,len(dur[k]))
json.dump(out,open("$O/summary.json","w"),indent=1)
for k,d in out.items():
    print(k)
    for c,v in sorted(d.items()): print(f"   {c:32s} {v:16.1f}")
PY
