#!/bin/bash
# A/B of the headline (config 3) between two settings of one environment variable, same box, interleaved.
# usage: scratch/ab_cfg3.sh OUT VAR VAL_A VAL_B [steps]
out=$1; var=$2; a=$3; b=$4; steps=${5:-200}
for rep in 1 2; do
for v in "$a" "$b"; do
  env $var=$v python bench.py --steps $steps --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 > /tmp/ab3.json 2>/tmp/ab3.err || { tail -5 /tmp/ab3.err; exit 1; }
  python - "$var=$v" /tmp/ab3.json >> "$out" <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
r = d["roofline"]
print(f"{sys.argv[1]}: ms_per_step {d['ms_per_step']}  train_graph_ms {d['train_graph_ms']}  mfcc {r['kernel_ms']}  cu {d['cu_partition']}  loss {d['loss']}")
P
done
done
cat "$out"
