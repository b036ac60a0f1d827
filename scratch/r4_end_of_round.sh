#!/bin/bash
# Round-4 evidence in one gpurun call: the bench line as the driver launches it (short) and a long run, the rocprofv3 kernel summary
# of the same command, the counter passes (instruction mix, matrix-pipe busy, HBM-side FETCH_SIZE / WRITE_SIZE) of the stand-alone
# MFCC stage with the block-DFT STFT kernel (default) and with the Stockham kernel (stage-mask 256), and profiles/r04_mfcc_pmc.json
# (what bench.py's roofline.traffic reads; carries the hash of the kernel sources it was taken at).  Outputs under gpurun_out/eor4/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eor4; mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_default_run.json 2> $O/bench_default_run.err || exit 1
timeout -k 10 500 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 50 --warmup 10 --skip-cpu-baseline --skip-b512 --skip-other-configs --pool-clips 16384 > $O/prof.log 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/config3_kernel_stats.csv
rm -rf $O/prof
cd $R
bash scratch/pmc_r4.sh 0 r4_default > $O/pmc_default.txt 2>&1
bash scratch/pmc_r4.sh 256 r4_stockham > $O/pmc_stockham.txt 2>&1
timeout -k 10 120 python scratch/time_mfcc2.py 1024 0 256 64 > $O/mfcc_standalone.txt 2>&1
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
out = {"batch": 1024, "note": "rocprofv3 --pmc, one pass per counter set (scratch/pmc_r4.sh; --kernel-trace only beside them), scratch/one_mfcc.py on 8 different "
       "batches, averages over the last 4 of 5 dispatches; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE (KiB) as is",
       "paths": {}}
for tag, name in (("r4_default", "default: resample_persist_h2 -> stft_bdft (block-DFT STFT on the matrix pipe) -> dct"), ("r4_stockham", "stage-mask 256: resample_persist_h2 -> stft_mel2 (Stockham FFT) -> dct")):
    s = json.load(open("$R/gpurun_out/pmc_%s/summary.json" % tag))
    tot = 0
    for k, v in s.items():
        v["hbm_read_bytes_corrected"] = 2 * v.get("FETCH_SIZE", 0) * 1024
        v["hbm_write_bytes"] = v.get("WRITE_SIZE", 0) * 1024
        tot += v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
    out["paths"][tag] = {"name": name, "kernels": s, "stage_bytes_per_launch": tot, "stage_bytes_per_utt": tot / 1024, "x_algorithmic": tot / (67520 * 1024)}
    print(tag, "stage bytes per launch of 1024 clips: %.1f MB = %.2f x algorithmic" % (tot / 1e6, tot / (67520 * 1024)))
d = out["paths"]["r4_default"]
out["end_of_round"] = {"stage_bytes_per_utt": d["stage_bytes_per_utt"], "x_algorithmic": d["x_algorithmic"], "source_sha16": bench.mfcc_source_sha()}
k = d["kernels"].get("stft_bdft_kernel", {})
if k:
    # matrix pipe busy per SIMD (256 CUs x 4) over the dispatch's shader cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)
    out["end_of_round"]["stft_bdft_mfma_busy_frac"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (k["GRBM_GUI_ACTIVE"] / 8.0) if k.get("GRBM_GUI_ACTIVE") else None
json.dump(out, open("$O/r04_mfcc_pmc.json", "w"), indent=1)
b = json.load(open("$O/bench_config3.json"))
print("bench long ", b["value"], b["ms_per_step"], b["train_graph_ms"], b["roofline"]["kernel_ms"], b["roofline"].get("standalone_whole_chip"))
b = json.load(open("$O/bench_default_run.json"))
print("bench short", b["value"], b["ms_per_step"], b["train_graph_ms"], b["roofline"]["kernel_ms"])
PY
cat $O/mfcc_standalone.txt | grep mask
