#!/usr/bin/env python3
"""bench.py -- utterances/sec of the end-to-end training step on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic 1 s @ 16 kHz clips that are already
resident in HBM:  waveform -> K1 MFCC (+ fused standardisation) -> K2 fwd/bwd (fp32 MFMA) -> [RCCL
all-reduce of the flat gradient, N > 1] -> K5 Adam + NonNeg -> K3 simple_norm_constraint(rho = 0.1)
(BASELINE configs 3 and 4; --pgd K adds config 5's PGD-K inner loop).  Per-GPU batch is fixed
(weak scaling): 1024 clips = config 4's 8192/8 shard; rank 0 additionally times the reference's own
batch of 512 at N = 1 and reports it under "reference_batch_512".

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events on the stream the kernels run
on (lipasr_mfcc_profile_*), `cpu_baseline` times the oracle (a NumPy restatement of the reference path;
TensorFlow / librosa are not installable here) on the host cores for a bounded sample.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# this pool's host driver only supports dmabuf IPC: RCCL needs it (harmless when already exported)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MFCC_BYTES_PER_UTT = 64000 + 3520       # 16000 fp32 samples in, 880 fp32 features out
MFCC_FLOP_PER_UTT = 8.7e6               # with the 16 k -> 22.05 k resampler
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md
FP32_PEAK_TFLOPS = 157.3
TRAIN_FLOP_PER_UTT = 9.59e6             # fwd 3.196 + bwd


def _cpu_mfcc_chunk(args):
    from oracle import mfcc_ref as M

    return M.compute_mfcc_batch(args, fast=True)


def cpu_baseline(batch, seconds_budget=20.0):
    """Reference-equivalent CPU path (restated; TF/librosa unavailable offline): per-clip MFCC loop +
    fp32 train step + simple_norm_constraint with LAPACK SVDs, on the host cores."""
    import multiprocessing as mp

    from lipasr.synth import synth_clips_fast
    from oracle import constraints_ref as R, mfcc_ref as M, mlp_ref as P

    cores = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
    waves, labels = synth_clips_fast(batch, seed=99)
    y = P.to_categorical(labels, 10)
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=0, dtype=np.float32, nonneg_init=True)
    st = P.AdamState()
    M.compute_mfcc_batch(waves[:2], fast=True)  # builds the cached tables
    ctx = mp.get_context("fork")
    steps, t_total, t_mfcc = 0, 0.0, 0.0
    with ctx.Pool(cores) as pool:
        chunks = [waves[i::cores] for i in range(cores)]
        pool.map(_cpu_mfcc_chunk, [c[:1] for c in chunks])  # warm the workers
        while t_total < seconds_budget and steps < 4:
            t0 = time.perf_counter()
            feats = np.concatenate(pool.map(_cpu_mfcc_chunk, chunks)).astype(np.float32)
            t1 = time.perf_counter()
            rng = np.random.default_rng(steps)
            masks = [((rng.uniform(size=(batch, s.n_out)) > s.dropout) / (1 - s.dropout)).astype(np.float32) if s.dropout > 0 else None for s in spec]
            P.train_step(spec, p, st, feats, y, masks=masks)
            new_w, _ = R.simple_norm_constraint_pass(p.W, 0.1, [])
            p.W = new_w
            t2 = time.perf_counter()
            steps += 1
            t_total += t2 - t0
            t_mfcc += t1 - t0
    return {"value": round(batch * steps / t_total, 2), "unit": "utterances/sec", "cores": cores, "kind": "port",
            "sample": f"{steps} steps of batch {batch} (oracle: NumPy MFCC loop over {cores} processes, fp32 NumPy/OpenBLAS train step, "
                      f"simple_norm_constraint with LAPACK SVD); {t_mfcc / t_total:.0%} of the time in MFCC",
            "label": "reference-equivalent CPU path (restated; TensorFlow/librosa unavailable offline)"}


def make_pool(n_clips, device, seed):
    from lipasr.synth import synth_clips_fast

    waves, labels = synth_clips_fast(n_clips, seed=seed)
    y = np.zeros((n_clips, 10), dtype=np.float32)
    y[np.arange(n_clips), labels] = 1
    return torch.as_tensor(waves).to(device), torch.as_tensor(y).to(device)


def run_config(args, batch, rank, world, device, steps, warmup, profile):
    """Builds model + pipeline for one per-GPU batch, times `steps` steps; returns (seconds, extras)."""
    import torch.distributed as dist

    import lipasr._native as N
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.keras import CategoricalCrossentropy
    from lipasr.parallel import DataParallel
    from lipasr.pipeline import TrainPipeline
    from lipasr.train_constraints import get_model

    n_batches = 8
    waves, y = make_pool(n_batches * batch, device, seed=1234 + rank)
    # same seed on every rank: replicas start identical
    model = get_model(max_batch=batch, seed=0, compute_dtype="bfloat16" if args.bf16 else "float32")
    model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    # A2 "affine, precomputed": StandardScaler fitted once on MFCCs of the pool
    ex = MfccExtractor(16000, 16000, batch, device)
    feats = torch.cat([ex(waves[i * batch:(i + 1) * batch]) for i in range(min(4, n_batches))])
    sc = StandardScaler().fit(feats)
    dp = DataParallel()
    if world > 1:
        dp.broadcast(sc.mean_, sc.scale_, model._params, model._bnstate)
    pgd = dict(eps=args.pgd_eps, eps_step=0.1, max_iter=args.pgd) if args.pgd > 0 else None
    pipe = TrainPipeline(model, batch=batch, rho=0.1, constraint=args.constraint, affine=(sc.mean_, sc.scale_), pgd=pgd, dp=dp,
                         use_graph=not args.no_graph)

    if os.environ.get("LIPASR_RS_WGS"):
        N.check(N.lib.lipasr_debug_set(pipe.h.h, 1, int(os.environ["LIPASR_RS_WGS"])))
    feat_pool = None
    if args.pre_extracted:
        feat_pool = torch.cat([ex(waves[i * batch:(i + 1) * batch], 44, sc.mean_, sc.scale_) for i in range(n_batches)])

    def one(i):
        s = (i % n_batches) * batch
        if feat_pool is not None:
            pipe.step(None, y[s:s + batch], features=feat_pool[s:s + batch])
        else:
            pipe.step(waves[s:s + batch], y[s:s + batch])

    for i in range(warmup):
        one(i)
    pipe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    standalone = None
    if profile and not args.pre_extracted:
        # the same three kernels alone on the whole chip (untimed, before the measured region): the pipeline confines
        # them to a CU share, which lengthens them by design -- both figures are reported
        s_all = torch.cuda.Stream(device=device)
        n_cu = torch.cuda.get_device_properties(device).multi_processor_count
        N.check(N.lib.lipasr_debug_set(pipe.h.h, 1, n_cu))  # resampler grid for the whole chip
        with torch.cuda.stream(s_all):
            for _ in range(3):
                pipe.ex(waves[:batch], 44, sc.mean_, sc.scale_, out=pipe._feats2[0][:batch])
            s_all.synchronize()
            N.check(N.lib.lipasr_mfcc_profile_begin(pipe.h.h, 10))
            for _ in range(10):
                pipe.ex(waves[:batch], 44, sc.mean_, sc.scale_, out=pipe._feats2[0][:batch])
            ms3, ncalls = (C.c_float * 3)(), C.c_int()
            N.check(N.lib.lipasr_mfcc_profile_end(pipe.h.h, ms3, C.byref(ncalls)))
        standalone = {"resample": round(ms3[0], 4), "stft_mel": round(ms3[1], 4), "dct": round(ms3[2], 4)}
        N.check(N.lib.lipasr_debug_set(pipe.h.h, 1, getattr(pipe, "mfcc_cus", n_cu)))  # back to the pipeline's CU share
        N.check(N.lib.lipasr_mfcc_profile_begin(pipe.h.h, steps))
    tid = C.c_int()
    N.check(N.lib.lipasr_timer_create(pipe.h.h, C.byref(tid)))
    with torch.cuda.stream(pipe.stream):
        N.check(N.lib.lipasr_timer_start(pipe.h.h, tid.value, N.stream_ptr()))
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    with torch.cuda.stream(pipe.stream):
        N.check(N.lib.lipasr_timer_stop(pipe.h.h, tid.value, N.stream_ptr()))
    pipe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = C.c_float()
    N.check(N.lib.lipasr_timer_elapsed_ms(pipe.h.h, tid.value, C.byref(ev_ms)))
    extras = {"event_ms_per_step": ev_ms.value / steps, "final_norm": float(pipe.norms[-1].item()) if args.constraint == "product" else None,
              "loss": float(model._loss_rows[:batch].mean().item())}
    if profile and args.pre_extracted:
        extras["mfcc_ms"] = {"resample": 0.0, "stft_mel": 0.0, "dct": 0.0, "calls": 0}
    elif profile:
        ms3 = (C.c_float * 3)()
        n = C.c_int()
        N.check(N.lib.lipasr_mfcc_profile_end(pipe.h.h, ms3, C.byref(n)))
        extras["mfcc_ms"] = {"resample": ms3[0], "stft_mel": ms3[1], "dct": ms3[2], "calls": n.value}
    extras["mfcc_standalone_ms"] = standalone
    extras["mfcc_cus"] = getattr(pipe, "mfcc_cus", None)
    extras["mfcc_stream"] = pipe.mfcc_stream_kind
    extras["n_cus"] = torch.cuda.get_device_properties(device).multi_processor_count
    assert np.isfinite(extras["loss"]), "training diverged"
    pipe.close()  # the masked stream is a hardware queue: hand it back before the next configuration builds its own
    return dt, extras


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--constraint", default="product", choices=["product", "per_layer", "none"])
    ap.add_argument("--pgd", type=int, default=0, help="PGD iterations per batch (config 5 uses 20)")
    ap.add_argument("--pgd-eps", type=float, default=0.5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--bf16", action="store_true", help="classifier GEMM operands rounded to bf16 at the MFMA, fp32 accumulate (BASELINE config 2's "
                    "arithmetic); the default and the headline are exact fp32")
    ap.add_argument("--pre-extracted", action="store_true", help="BASELINE config 2: train from resident (N,880) features, no MFCC stage")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-b512", action="store_true")
    args = ap.parse_args()
    if args.constraint == "none":
        args.constraint = None

    from lipasr.parallel import init_from_env

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    rank, world = init_from_env("nccl" if world_env > 1 else None)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist

    batch = args.batch_per_gpu
    dt, ex = run_config(args, batch, rank, world, device, args.steps, args.warmup, profile=True)
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    global_batch = batch * world
    value = global_batch * args.steps / dt
    ms = ex["mfcc_ms"]
    stage_ms = ms["resample"] + ms["stft_mel"] + ms["dct"]
    dom = max(("resample", "stft_mel", "dct"), key=lambda k: ms[k])
    achieved = MFCC_BYTES_PER_UTT * batch / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
    # HBM traffic of the stage per launch: PMC counters cannot be read from inside this process, so the value is
    # the committed rocprofv3 measurement of the same three kernels (profiles/r02_mfcc_pmc.json: separate
    # --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md), scaled to this batch.
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r02_mfcc_pmc.json")) as f:
            traffic = round(json.load(f)["end_of_round"]["stage_bytes_per_utt"] * batch)
    except Exception:
        pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": "profiles/r02_mfcc_pmc.json (rocprofv3 --pmc, per launch, scaled by batch)", "stage": "MFCC (K1 = resample + stft_mel + dct kernels)", "dominant_kernel": dom + "_kernel",
                "algorithmic_bytes_per_utt": MFCC_BYTES_PER_UTT, "units_per_launch": batch,
                "kernel_ms": {k: round(ms[k], 4) for k in ("resample", "stft_mel", "dct")},
                "fp32_flop_frac": round(MFCC_FLOP_PER_UTT * batch / (stage_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5) if stage_ms > 0 else 0.0,
                "note": "stage is fp32-compute-bound (8.7 MFLOP/utt vs 67.5 kB/utt): see DESIGN.md"}
    if ex.get("mfcc_cus"):
        # the pipeline confines the MFCC stream to a share of the CUs (DESIGN.md, step level): its kernels take longer by
        # design while the step gets shorter; `frac` above is what the timed region shows, the fields below put it in context
        share = ex["mfcc_cus"] / ex["n_cus"]
        roofline["cu_share"] = round(share, 4)
        roofline["frac_of_share"] = round(achieved / HBM_PEAK_GBS / share, 5)
    if ex.get("mfcc_standalone_ms"):
        sa = ex["mfcc_standalone_ms"]
        sa_ms = sa["resample"] + sa["stft_mel"] + sa["dct"]
        sa_ach = MFCC_BYTES_PER_UTT * batch / (sa_ms * 1e-3) / 1e9
        roofline["standalone_whole_chip"] = {"kernel_ms": sa, "achieved": round(sa_ach, 2), "frac": round(sa_ach / HBM_PEAK_GBS, 5),
                                             "fp32_flop_frac": round(MFCC_FLOP_PER_UTT * batch / (sa_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5)}
    if args.pre_extracted:
        tf = TRAIN_FLOP_PER_UTT * batch / (ex["event_ms_per_step"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 5),
                    "traffic": None, "stage": "dense classifier train step (v_mfma_f32_32x32x2_f32, exact fp32)",
                    "algorithmic_flop_per_utt": TRAIN_FLOP_PER_UTT, "units_per_launch": batch,
                    "note": "whole step incl. BatchNorm, Adam and projection kernels; GEMM-only time is in profiles/"}
    out = {"metric": "utterances/sec (train, 1 s@16 kHz)", "value": round(value, 1), "unit": "utterances/sec", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.bf16 else "f32", "data": "synthetic",
           "config": {"workload": ("pre-extracted standardised (N,880) MFCC features" if args.pre_extracted else "raw 16 kHz waveform -> on-GPU MFCC")
                                  + " -> Lipschitz-constrained MLP train step (Adam+NonNeg, simple_norm_constraint rho=0.1)" + (f" + PGD-{args.pgd} adversarial inner loop" if args.pgd else "")
                                  + (", data-parallel RCCL gradient all-reduce" if world > 1 else ", 1xMI355X"),
                      "baseline_config": 5 if args.pgd else (2 if args.pre_extracted else (4 if world > 1 else 3)), "global_batch": global_batch, "per_gpu_batch": batch,
                      "clip": "1 s @ 16 kHz fp32", "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
           "roofline": roofline, "mfcc_stream": ex.get("mfcc_stream"),
           "mlp_tflops": round(TRAIN_FLOP_PER_UTT * batch / max(1e-9, (ex["event_ms_per_step"] - stage_ms) * 1e-3) / 1e12, 3),
           "event_ms_per_step": round(ex["event_ms_per_step"], 4), "final_product_norm": ex["final_norm"], "loss": round(ex["loss"], 4)}
    if world == 1 and not args.skip_b512 and batch != 512:
        dt5, ex5 = run_config(args, 512, rank, world, device, args.steps, args.warmup, profile=True)
        out["reference_batch_512"] = {"value": round(512 * args.steps / dt5, 1), "ms_per_step": round(dt5 / args.steps * 1e3, 4),
                                      "mfcc_ms": {k: round(v, 4) for k, v in ex5["mfcc_ms"].items() if k != "calls"},
                                      "mfcc_stream": ex5.get("mfcc_stream")}
    if world == 1 and not args.skip_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(512)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
