#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
F="--skip-cpu-baseline --skip-b512 --skip-other-configs --steps 300 --warmup 30"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d["ms_per_step"], d.get("train_graph_ms"), d.get("loss"))'
{
echo "default"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "deep: stats allowed, K>=512, tiles<=320"; LIPASR_DEEP_STATS=1 LIPASR_DEEP_TILES=320 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "deep: stats allowed, K>=256, tiles<=320"; LIPASR_DEEP_STATS=1 LIPASR_DEEP_TILES=320 LIPASR_DEEP_K=256 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "deep: stats allowed, K>=128, tiles<=320"; LIPASR_DEEP_STATS=1 LIPASR_DEEP_TILES=320 LIPASR_DEEP_K=128 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "deep: stats allowed, K>=128, tiles<=192"; LIPASR_DEEP_STATS=1 LIPASR_DEEP_K=128 timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "default"; timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P"
echo "pre-extracted default"; timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
echo "pre-extracted deep: stats, K>=256, tiles<=320"; LIPASR_DEEP_STATS=1 LIPASR_DEEP_TILES=320 LIPASR_DEEP_K=256 timeout -k 10 200 python bench.py $F --pre-extracted 2>/dev/null | python -c "$P"
} > gpurun_out/deep.txt 2>&1
cat gpurun_out/deep.txt
