#!/bin/bash
# per-kernel PMC counters of the classifier step (config 2): scratch/pmc_cfg2.sh TAG "COUNTER LIST" [env assignments]
tag=$1; ctrs=$2; shift 2
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/pmc2_$tag
PRE=--pre-extracted
if [ "$PMC_CFG3" = "1" ]; then PRE=; fi   # PMC_CFG3=1: the headline configuration (the classifier on its 128-CU share)
rm -rf $out
rocprofv3 --kernel-trace --pmc $ctrs -d $out --output-format csv -- python3 $R/bench.py $PRE --steps 12 --warmup 4 --skip-cpu-baseline --skip-other-configs --skip-b512 > $R/gpurun_out/pmc2_$tag.log 2>&1
python3 - "$out" > $R/gpurun_out/pmc2_$tag.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "lipasr" not in r["Kernel_Name"]:
            continue
        k = r["Kernel_Name"].replace("lipasr::", "").replace("void ", "")[:60] + " grid " + r.get("Grid_Size", "?")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(acc.items()):
    print(k, {c: round(sum(v[-6:]) / len(v[-6:]), 1) for c, v in d.items()})
PY
cat $R/gpurun_out/pmc2_$tag.txt
rm -rf $out
