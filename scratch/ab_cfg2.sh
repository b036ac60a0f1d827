#!/bin/bash
# A/B of config 2 (pre-extracted features, classifier alone): fused BatchNorm exchange against the launch chain, same box, back to back
set -e
out=${1:-gpurun_out/ab_cfg2.txt}
for rep in 1 2; do
for fb in 1 0; do
  LIPASR_FUSE_BN=$fb python bench.py --pre-extracted --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 > /tmp/ab_$fb.json
  python - "$fb" /tmp/ab_$fb.json >> "$out" <<'P'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"LIPASR_FUSE_BN={sys.argv[1]}: config 2 ms_per_step {d['ms_per_step']}  train_graph_ms {d['train_graph_ms']}  loss {d['loss']}")
P
done
done
cat "$out"
