#!/bin/bash
# Round-5 evidence in one gpurun call (outputs under gpurun_out/eor5/): the bench line as the driver launches it and a long run, the
# rocprofv3 kernel summaries of config 3 and config 2, one step of each kernel by kernel, the stand-alone MFCC times, the counter
# passes of the MFCC stage (HBM-side FETCH_SIZE / WRITE_SIZE and the instruction mix) -> profiles/r05_mfcc_pmc.json.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/eor5; mkdir -p $O
cd $R
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench_default_run.json 2> $O/bench_default_run.err || exit 1
timeout -k 10 600 python bench.py --steps 200 --warmup 20 --skip-cpu-baseline > $O/bench_config3.json 2> $O/bench_config3.err || exit 1
timeout -k 10 300 python bench.py --pre-extracted --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 > $O/bench_config2.json 2> $O/bench_config2.err || exit 1
timeout -k 10 300 python bench.py --exact-fp32 --steps 200 --warmup 20 --skip-cpu-baseline --skip-other-configs --skip-b512 > $O/bench_config3_exact_fp32.json 2> $O/bench_config3_exact.err || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py --steps 50 --warmup 10 --skip-cpu-baseline --skip-b512 --skip-other-configs --pool-clips 16384 > $O/prof.log 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/config3_kernel_stats.csv
rm -rf $O/prof
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof2 -o run -- python3 $R/bench.py --pre-extracted --steps 100 --warmup 10 --skip-cpu-baseline --skip-b512 --skip-other-configs > $O/prof2.log 2>&1 || exit 1
cp $(find $O/prof2 -name "*kernel_stats.csv" | head -1) $O/config2_kernel_stats.csv
rm -rf $O/prof2
cd $R
timeout -k 10 250 ./scratch/step_timeline.sh eor5_cfg2 > $O/step_timeline_config2.txt 2>&1
timeout -k 10 250 ./scratch/step_timeline.sh eor5_cfg3 TL_CFG3=1 LIPASR_GPU_FLAGS=0 > $O/step_timeline_config3.txt 2>&1
bash scratch/pmc_r4.sh 0 r5_default > $O/pmc_default.txt 2>&1
timeout -k 10 120 python scratch/time_mfcc2.py 1024 0 256 > $O/mfcc_standalone.txt 2>&1
python3 - <<PY
import json, sys
sys.path.insert(0, "$R")
import bench
out = {"batch": 1024, "note": "rocprofv3 --pmc, one pass per counter set (scratch/pmc_r4.sh; --kernel-trace only beside them), scratch/one_mfcc.py on 8 different "
       "batches, averages over the last 4 of 5 dispatches; FETCH_SIZE (KiB) doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE (KiB) as is",
       "paths": {}}
s = json.load(open("$R/gpurun_out/pmc_r5_default/summary.json"))
tot = 0
for k, v in s.items():
    v["hbm_read_bytes_corrected"] = 2 * v.get("FETCH_SIZE", 0) * 1024
    v["hbm_write_bytes"] = v.get("WRITE_SIZE", 0) * 1024
    tot += v["hbm_read_bytes_corrected"] + v["hbm_write_bytes"]
out["paths"]["r5_default"] = {"name": "default: resample_persist_h2 -> stft_bdft (block-DFT STFT on the matrix pipe) -> dct", "kernels": s, "stage_bytes_per_launch": tot,
                              "stage_bytes_per_utt": tot / 1024, "x_algorithmic": tot / (67520 * 1024)}
print("stage bytes per launch of 1024 clips: %.1f MB = %.2f x algorithmic" % (tot / 1e6, tot / (67520 * 1024)))
d = out["paths"]["r5_default"]
out["end_of_round"] = {"stage_bytes_per_utt": d["stage_bytes_per_utt"], "x_algorithmic": d["x_algorithmic"], "source_sha16": bench.mfcc_source_sha()}
k = d["kernels"].get("stft_bdft_kernel", {})
if k and k.get("GRBM_GUI_ACTIVE"):
    out["end_of_round"]["stft_bdft_mfma_busy_frac"] = k.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / (k["GRBM_GUI_ACTIVE"] / 8.0)
    if k.get("SQ_WAVE_CYCLES"):
        out["end_of_round"]["stft_bdft_active_inst_frac"] = k.get("SQ_ACTIVE_INST_ANY", 0) / k["SQ_WAVE_CYCLES"]
json.dump(out, open("$O/r05_mfcc_pmc.json", "w"), indent=1)
for name in ("bench_default_run", "bench_config3", "bench_config2", "bench_config3_exact_fp32"):
    b = json.loads([l for l in open("$O/%s.json" % name) if l.startswith("{")][-1])
    print(name, b["value"], b["ms_per_step"], b["train_graph_ms"], b["dtype"][:20], b.get("cu_partition"))
PY
cat $O/mfcc_standalone.txt | grep mask
