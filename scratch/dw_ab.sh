#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 20"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"), d.get("loss"), d.get("final_product_norm"))'
{
timeout -k 10 600 python -m pytest tests/test_mlp_gpu.py tests/test_pipeline_gpu.py tests/test_dp_gpu.py tests/test_end_to_end_gpu.py tests/test_fullsize_gpu.py tests/test_sr_gpu.py -x -q -m gpu 2>&1 | tail -5
for rep in 1 2; do
echo "BN applies formed by the consuming GEMM"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "separate BN apply launches"; LIPASR_GEMM_MODE=16 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
done
echo "pre-extracted: formed"; timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
echo "pre-extracted: separate"; LIPASR_GEMM_MODE=16 timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
} > gpurun_out/bn_fuse2.txt 2>&1
cat gpurun_out/bn_fuse2.txt
