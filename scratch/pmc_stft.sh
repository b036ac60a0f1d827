#!/bin/bash
# rocprofv3 counter passes on the stand-alone MFCC loop; summaries land in gpurun_out/pmc_stft/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_stft
mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set -d $O/p$i --output-format csv -- python3 $R/scratch/time_mfcc.py 1024 0 > $O/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $O/p$i.log; }
done
python3 - <<PY
import csv,glob,collections
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    acc=collections.defaultdict(lambda: [0,0.0])
    for r in csv.DictReader(open(f)):
        if 'stft_mel_kernel' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for k,(n,v) in acc.items(): print(f"{k:28s} launches {n:4d}  per-launch {v/n:16.1f}")
PY
