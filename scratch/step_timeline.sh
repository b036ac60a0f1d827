#!/bin/bash
# One training step of config 2 kernel by kernel (name, grid, duration, gap to the previous kernel's end), from rocprofv3 --kernel-trace.
# usage: scratch/step_timeline.sh <tag> [extra env assignments ...]   (run from the repo root on the GPU box)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/tl_$tag
rm -rf $out
PRE=--pre-extracted
if [ "$TL_CFG3" = "1" ]; then PRE=; fi   # TL_CFG3=1: the headline configuration; the extraction stream's kernels are left out of the listing
rocprofv3 --kernel-trace --output-format csv -d $out -o p -- python3 $R/bench.py $PRE --steps 40 --warmup 10 --skip-cpu-baseline --skip-other-configs --skip-b512 > $R/gpurun_out/tl_$tag.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $R/gpurun_out/tl_$tag.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "lipasr" in r["Kernel_Name"] and not any(k in r["Kernel_Name"] for k in ("resample", "stft", "dct_kernel", "flag_", "mfcc"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by adam_nonneg_kernel; take the median-length step among the last 20
idx = [i for i, r in enumerate(rows) if "adam_nonneg" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    steps.append((int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"]), a, b))
steps = steps[-20:]
steps.sort()
dur, a, b = steps[len(steps) // 2]
seg = rows[a:b]
print(f"median step (adam .. next adam): {dur/1e3:.1f} us, {len(seg)} kernels")
prev_end = None
tk = 0
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    tk += (e - s)
    name = r["Kernel_Name"].replace("lipasr::", "").replace("void ", "")[:70]
    print(f"{name:70s} grid {r.get('Grid_Size_X', r.get('Grid_Size','?')):>7s} wg {r.get('Workgroup_Size_X', r.get('Workgroup_Size','?')):>4s}  {(e - s)/1e3:7.2f} us   gap {gap:6.2f}")
    prev_end = e
print(f"sum of kernel durations {tk/1e3:.1f} us")
PY
cat $R/gpurun_out/tl_$tag.txt
rm -rf $out
