#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the MFCC kernels when the batch is processed in chunks (scratch/chunk_mfcc.py).  $1 = clips per chunk
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; C=${1:-1024}
O=$R/gpurun_out/pmc_chunk_$C; mkdir -p $O
for set in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set -d $O/$set --output-format csv -- python3 $R/scratch/chunk_mfcc.py $C > $O/$set.log 2>&1 || { echo "pass $set failed"; tail -5 $O/$set.log; exit 1; }
done
grep "per 1024" $O/FETCH_SIZE.log
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(int)
for f in sorted(glob.glob("$O/*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'lipasr' in r['Kernel_Name']:
            k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('lipasr::','').split('<')[0]
            acc[k][r['Counter_Name']]+=float(r['Counter_Value']); 
            if r['Counter_Name']=='FETCH_SIZE': n[k]+=1
# 6 batches of 1024 clips were extracted (1 warm-up + 5); counters are in units of 32 B x 2 on gfx950 per profiles/README (FETCH_SIZE kB-units: see guide)
for k,d in acc.items():
    print(f"{k:32s} dispatches {n[k]:5d}  read {2*d.get('FETCH_SIZE',0)*1024/6/1e6:8.1f} MB  written {d.get('WRITE_SIZE',0)*1024/6/1e6:8.1f} MB   per 1024 clips (FETCH_SIZE KiB doubled, WRITE_SIZE KiB)")
PY
rm -rf $O/FETCH_SIZE $O/WRITE_SIZE
