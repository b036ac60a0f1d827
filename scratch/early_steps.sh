#!/bin/bash
# Per-step kernel time of the classifier's stream from the first step on (why are the driver's 20 timed steps slower than steps 100+?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/early; rm -rf $out
LIPASR_GPU_FLAGS=0 rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $R/bench.py --steps 80 --warmup 0 --skip-cpu-baseline --skip-other-configs --skip-b512 > $R/gpurun_out/early.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/early.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "lipasr" in r["Kernel_Name"] and not any(k in r["Kernel_Name"] for k in ("resample", "stft", "dct_kernel", "flag_", "mfcc"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_nonneg" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    d = collections.OrderedDict()
    for k, r in enumerate(seg):
        name = r["Kernel_Name"].replace("lipasr::", "").replace("void ", "")[:28] + "#%d" % k
        d[name] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    steps.append(d)
print("steps", len(steps))
def avg(lo, hi):
    acc = collections.OrderedDict()
    n = 0
    for s in steps[lo:hi]:
        if len(s) != len(steps[-1]): continue
        n += 1
        for k, v in s.items(): acc[k] = acc.get(k, 0.0) + v
    return {k: v / max(n, 1) for k, v in acc.items()}, n
early, n1 = avg(5, 25)
late, n2 = avg(55, 78)
print("kernel (position in the step: adam first)            steps 5-24   steps 55-77   diff")
for k in late:
    print(f"{k:50s} {early.get(k, 0):9.2f} {late[k]:11.2f} {early.get(k, 0) - late[k]:8.2f}")
print("sum", round(sum(early.values()), 1), round(sum(late.values()), 1), n1, n2)
print("per-step sums:", [round(sum(s.values())) for s in steps[:80]])
PY
cat $R/gpurun_out/early.txt
rm -rf $out
