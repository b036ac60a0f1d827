#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"), d["roofline"]["kernel_ms"])'
{
echo "96 | 160"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "128 | 160 (32 shared)"; LIPASR_MFCC_CUS=128 LIPASR_TRAIN_OVERLAP_GROUPS=4 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "96 | 192 (32 shared)"; LIPASR_MFCC_CUS=96 LIPASR_TRAIN_OVERLAP_GROUPS=4 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "128 | 192 (64 shared)"; LIPASR_MFCC_CUS=128 LIPASR_TRAIN_OVERLAP_GROUPS=8 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
} > gpurun_out/overlap_groups.txt 2>&1
cat gpurun_out/overlap_groups.txt
