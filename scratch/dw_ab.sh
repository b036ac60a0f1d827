#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"))'
{
for rep in 1 2; do
echo "auto"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "lds wherever legal"; LIPASR_GEMM_MODE=2 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
done
echo "pre-extracted: auto"; timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
echo "pre-extracted: lds wherever legal"; LIPASR_GEMM_MODE=2 timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
echo "pre-extracted: fragment dW"; LIPASR_GEMM_MODE=8 timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
} > gpurun_out/dw_ab2.txt 2>&1
cat gpurun_out/dw_ab2.txt
