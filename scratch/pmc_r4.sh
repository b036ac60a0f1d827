#!/bin/bash
# round 4: rocprofv3 counter passes on the stand-alone MFCC stage (scratch/one_mfcc.py <mask>).  $1 = mask, $2 = tag, $3 = counter-set selection (default all).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; MASK=${1:-0}; TAG=${2:-bdft}
O=$R/gpurun_out/pmc_$TAG; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/p$i --output-format csv -- python3 $R/scratch/one_mfcc.py $MASK > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('lipasr::','')
        if 'lipasr' in r['Kernel_Name']:
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
out={k:{c:sum(v[-4:])/len(v[-4:]) for c,v in d.items()} for k,d in acc.items()}
json.dump(out,open("$O/summary.json","w"),indent=1)
for k,d in out.items():
    print(k)
    for c,v in sorted(d.items()): print(f"   {c:28s} {v:16.1f}")
PY
rm -rf $O/p*/  # the raw traces stay on the box
