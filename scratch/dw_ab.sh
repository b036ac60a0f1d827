#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"), d["roofline"]["kernel_ms"])'
{
for g in 12 13 14 15; do
echo "groups $g, lds min tiles 224"; LIPASR_MFCC_GROUPS=$g timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "groups $g, lds min tiles 128"; LIPASR_MFCC_GROUPS=$g LIPASR_LDS_MIN_TILES=128 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
done
} > gpurun_out/groups.txt 2>&1
cat gpurun_out/groups.txt
