#!/bin/bash
# end-of-round evidence: bench lines of configs 3 / 2 / 2-bf16 / 5, the rocprofv3 kernel summary of config 3 and the
# HBM-traffic counters of the MFCC kernels.  Outputs under gpurun_out/eor/ (copied into profiles/ by hand).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eor; mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --steps 100 --warmup 10 > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
timeout -k 10 300 python bench.py --pre-extracted --skip-cpu-baseline --skip-b512 > $O/bench_config2.json 2> $O/c2.err || exit 1
timeout -k 10 300 python bench.py --pre-extracted --bf16 --skip-cpu-baseline --skip-b512 > $O/bench_config2_bf16.json 2> $O/c2b.err || exit 1
timeout -k 10 300 python bench.py --pgd 20 --steps 30 --skip-cpu-baseline --skip-b512 > $O/bench_config5.json 2> $O/c5.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 30 --warmup 5 --skip-cpu-baseline --skip-b512 > $O/prof.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python3 $R/scratch/one_mfcc.py > $O/pmc_$c.log 2>&1 || exit 1
done
cd $R
python3 - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/pmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[c] = {k: [len(v), sum(v[-5:]) / len(v[-5:])] for k, v in acc.items() if "lipasr" in k}
json.dump(out, open("$O/mfcc_pmc_raw.json", "w"), indent=1)
print(json.dumps(out, indent=1))
for n in ("bench_config3", "bench_config2", "bench_config2_bf16", "bench_config5"):
    d = json.load(open("$O/%s.json" % n)); print(n, d["ms_per_step"], d["value"], d.get("mfcc_stream"))
PY
