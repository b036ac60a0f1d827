#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"), d["roofline"]["kernel_ms"])'
{
for cus in 96 128 64; do
echo "mfcc $cus | rest"; LIPASR_MFCC_CUS=$cus timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
done
echo "mfcc 128, classifier everywhere"; LIPASR_TRAIN_CUS=all LIPASR_MFCC_CUS=128 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "batch 512: 64 | rest"; timeout -k 10 200 python bench.py $F --batch-per-gpu 512 2>/dev/null | python -c "$P"
echo "batch 2048: 96 | rest"; timeout -k 10 200 python bench.py $F --batch-per-gpu 2048 --pool-clips 32768 2>/dev/null | python -c "$P"
echo "batch 2048: 128 | rest"; LIPASR_MFCC_CUS=128 timeout -k 10 200 python bench.py $F --batch-per-gpu 2048 --pool-clips 32768 2>/dev/null | python -c "$P"
} > gpurun_out/cu_sweep3.txt 2>&1
cat gpurun_out/cu_sweep3.txt
