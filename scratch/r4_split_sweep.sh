#!/bin/bash
# round 4: config 3 step time against the CU split (MFCC share), 300-step runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_split; mkdir -p $O
B="--steps 300 --warmup 30 --skip-cpu-baseline --skip-other-configs --skip-b512"
for cus in 64 96 128; do
  for rep in 1 2; do
    LIPASR_MFCC_CUS=$cus python $R/bench.py $B > $O/c3_${cus}_$rep.json 2>>$O/err.log
  done
done
python - <<PY
import json,glob
for f in sorted(glob.glob("$O/c3_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d["roofline"]
        print(f.split('/')[-1], d["ms_per_step"], "train_graph", d["train_graph_ms"], r.get("kernel_ms"), d.get("cu_partition"))
    except Exception as e: print(f, "ERR", e)
PY
