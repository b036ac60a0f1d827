#!/bin/bash
# the classifier's stream confined to the CUs the MFCC stream does not use (LIPASR_TRAIN_CUS=rest) against the default
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"))'
{
for rep in 1 2 3; do
echo "default"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "rest, mfcc 96"; LIPASR_TRAIN_CUS=rest LIPASR_MFCC_CUS=96 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "shared train stream, mfcc 96"; LIPASR_MFCC_CUS=96 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
done
echo "batch 512: default"; timeout -k 10 200 python bench.py $F --batch-per-gpu 512 2>/dev/null | python -c "$P"
echo "batch 512: rest, mfcc 96"; LIPASR_TRAIN_CUS=rest LIPASR_MFCC_CUS=96 timeout -k 10 200 python bench.py $F --batch-per-gpu 512 2>/dev/null | python -c "$P"
echo "batch 512: rest, mfcc 64"; LIPASR_TRAIN_CUS=rest LIPASR_MFCC_CUS=64 timeout -k 10 200 python bench.py $F --batch-per-gpu 512 2>/dev/null | python -c "$P"
} > gpurun_out/train_mask.txt 2>&1
cat gpurun_out/train_mask.txt
