#!/bin/bash
# Round-3 evidence in one gpurun call: the bench line (config 3 + the secondary records), the rocprofv3 kernel summary of the
# same command, and the counter passes (instruction mix + HBM-side FETCH_SIZE / WRITE_SIZE) of the stand-alone MFCC stage for
# the default three-kernel path and for the fused resample -> STFT kernel.  Outputs under gpurun_out/eor3/ (copied into
# profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eor3; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 200 --warmup 20 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 50 --warmup 10 --skip-cpu-baseline --skip-b512 --skip-other-configs --pool-clips 16384 > $O/prof.log 2>&1 || exit 1
cd $R
bash scratch/pmc_r3.sh 0 default > $O/pmc_default.txt 2>&1
bash scratch/pmc_r3.sh 1024 fused > $O/pmc_fused.txt 2>&1
python3 scratch/diag_r3.py 1024 > $O/phase_split.txt 2>&1
python3 - <<PY
import json
d = json.load(open("$O/bench_config3.json"))
print("bench", d["value"], d["ms_per_step"], d["train_graph_ms"], d["roofline"]["kernel_ms"], d["roofline"].get("standalone_whole_chip"))
for tag in ("default", "fused"):
    s = json.load(open("$R/gpurun_out/pmc_%s/summary.json" % tag))
    tot = 0
    for k, v in s.items():
        rd, wr = 2 * v.get("FETCH_SIZE", 0) * 1024, v.get("WRITE_SIZE", 0) * 1024
        tot += rd + wr
        print(tag, k, "read MB %.1f write MB %.1f" % (rd / 1e6, wr / 1e6))
    print(tag, "stage bytes per launch of 1024 clips: %.1f MB = %.2f x algorithmic" % (tot / 1e6, tot / (67520 * 1024)))
PY
