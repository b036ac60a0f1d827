#!/bin/bash
# per-kernel times of the classifier step (config 2, pre-extracted features), fused apply (default) or LIPASR_GEMM_MODE=16
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_cfg2_$1
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $R/bench.py --pre-extracted --steps 100 --warmup 10 --skip-cpu-baseline --skip-other-configs --skip-b512 > $R/gpurun_out/prof_cfg2_$1.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/prof_cfg2_$1_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in [r for r in rows if "lipasr" in r["Name"]]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.2f} us  min {float(r['MinNs'])/1e3:6.2f}")
PY
rm -rf $out
