#!/bin/bash
# round 3, GPU call A: the tests that changed, the exit probes (plain, profiled, profiled with the round-2 leak), a first bench line
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_end_to_end_gpu.py tests/test_pipeline_gpu.py tests/test_teardown_gpu.py tests/test_mfcc_gpu.py -q -m gpu -s > gpurun_out/ta.log 2>&1
echo "pytest rc=$?" >> gpurun_out/ta.log
grep -E "passed|failed|FAILED|ERROR|end-to-end|accuracy|constrained model|layer [0-5]:" gpurun_out/ta.log | tail -40
python3 scratch/exit_probe.py > gpurun_out/exit_plain.log 2>&1; echo "exit_probe plain rc=$?" | tee -a gpurun_out/exit_plain.log
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/exit_fixed -- python3 scratch/exit_probe.py > gpurun_out/exit_fixed.log 2>&1; echo "exit_probe profiled rc=$?" | tee -a gpurun_out/exit_fixed.log
timeout -k 10 400 python bench.py --steps 50 --warmup 10 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_a.json
rocprofv3 --kernel-trace --stats -d gpurun_out/exit_leak -- python3 scratch/exit_probe.py --leak > gpurun_out/exit_leak.log 2>&1; echo "exit_probe profiled+leak rc=$?" | tee -a gpurun_out/exit_leak.log
exit 0
