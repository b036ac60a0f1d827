#!/usr/bin/env python3
"""bench.py -- utterances/sec of the end-to-end training step on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch of synthetic 1 s @ 16 kHz clips that are already
resident in HBM:  waveform -> K1 MFCC (+ fused standardisation) -> K2 fwd/bwd (fp32 MFMA) -> [RCCL
all-reduce of the flat gradient, N > 1] -> K5 Adam + NonNeg -> K3 simple_norm_constraint(rho = 0.1)
(BASELINE configs 3 and 4; --pgd K adds config 5's PGD-K inner loop).  Default: per-GPU batch fixed
(weak scaling): 1024 clips = config 4's 8192/8 shard.  `--scaling strong` fixes the GLOBAL batch instead
(`--global-batch`, default 8192 = config 4's) and gives every rank 8192 / N clips per step: the form
north_star's ">= 6.5x strong scaling at 8 GPUs" is worded in; its N = 1 denominator (one GPU, batch
8192) is also in every default N = 1 line as "reference_global_batch_8192_1gpu".  Rank 0 additionally
times the reference's own batch of 512 at N = 1 ("reference_batch_512").

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


def _cpu_worker_init():
    try:
        from threadpoolctl import threadpool_limits

        threadpool_limits(limits=1)
    except Exception:
        pass


def cpu_baseline(batch, warmup=5, steps=20):
    """Reference-equivalent CPU path (restated; TF/librosa unavailable offline): per-clip MFCC loop +
    fp32 train step + simple_norm_constraint with LAPACK SVDs, on every core of the job's CPU share
    (>= 5 warm-up + >= 20 timed steps; the counts are in the record)."""
    import multiprocessing as mp

    from lipasr.synth import synth_clips_fast
    from oracle import constraints_ref as R, mfcc_ref as M, mlp_ref as P

    host_cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # The GPU box shows every CPU of the host (256) but grants a one-GPU job a 16-CPU share; 256 workers on that share
    # measured 208 utt/s against 455 with 16 (round 3 / round 2).  LIPASR_CPU_WORKERS overrides.
    cores = max(1, min(host_cpus, int(os.environ.get("LIPASR_CPU_WORKERS", "16"))))
    try:  # the train step's GEMMs: OpenBLAS on the same cores (the MFCC workers are single-threaded)
        from threadpoolctl import threadpool_info, threadpool_limits

        threadpool_limits(limits=cores)
        blas_threads = max([i.get("num_threads", 1) for i in threadpool_info() if i.get("user_api") == "blas"] or [1])
    except Exception:
        blas_threads = None
    pool_clips = 4 * batch
    waves, labels = synth_clips_fast(pool_clips, seed=99)
    y = P.to_categorical(labels, 10)
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=0, dtype=np.float32, nonneg_init=True)
    st = P.AdamState()
    # spawn, not fork: this process holds a HIP context by now (tests/helpers.py states the same rule); every worker builds
    # its own tables in the warm-up map below
    ctx = mp.get_context("spawn")
    t_total, t_mfcc = 0.0, 0.0
    with ctx.Pool(cores, initializer=_cpu_worker_init) as pool:
        pool.map(_cpu_mfcc_chunk, [waves[i:i + 1] for i in range(cores)])  # every worker has imported and built its tables
        for it in range(warmup + steps):
            s = (it % 4) * batch
            wb, yb = waves[s:s + batch], y[s:s + batch]
            t0 = time.perf_counter()
            feats = np.concatenate(pool.map(_cpu_mfcc_chunk, np.array_split(wb, min(batch, 2 * cores)))).astype(np.float32)
            t1 = time.perf_counter()
            rng = np.random.default_rng(it)
            masks = [((rng.uniform(size=(batch, sp.n_out)) > sp.dropout) / (1 - sp.dropout)).astype(np.float32) if sp.dropout > 0 else None for sp in spec]
            P.train_step(spec, p, st, feats, yb, masks=masks)
            new_w, _ = R.simple_norm_constraint_pass(p.W, 0.1, [])
            p.W = new_w
            t2 = time.perf_counter()
            if it >= warmup:
                t_total += t2 - t0
                t_mfcc += t1 - t0
    return {"value": round(batch * steps / t_total, 2), "unit": "utterances/sec", "cores": cores, "host_cpus": host_cpus, "blas_threads": blas_threads, "kind": "port",
            "sample": f"{steps} timed steps of batch {batch} after {warmup} warm-up steps ({t_total:.1f} s; oracle: NumPy MFCC loop over {cores} "
                      f"processes, fp32 NumPy/OpenBLAS train step, simple_norm_constraint with LAPACK SVD); {t_mfcc / t_total:.0%} of the time in MFCC",
            "label": "reference-equivalent CPU path (restated; TensorFlow/librosa unavailable offline)"}


# rocprofv3 names of the three timed MFCC slots (what `roofline.dominant_kernel` must match in profiles/*kernel_stats.csv)
MFCC_KERNELS = {"resample": "resample_persist_h2_kernel", "stft_mel": "stft_bdft_kernel", "dct": "dct_kernel"}


def mfcc_kernel_names(fused=False):
    """The three timed slots' kernel names for the ACTIVE stage mask (LIPASR_MFCC_MASK: 256 = the Stockham STFT kernel, 64 = the
    round-2 one, 16 = the fp32 resampler; ADVICE r4: the names used to be hard-coded to the default path)."""
    mask = int(os.environ.get("LIPASR_MFCC_MASK", "0") or 0)
    names = dict(MFCC_KERNELS)
    if fused:
        names["resample"], names["stft_mel"] = "(inside mfcc_fused_kernel)", "mfcc_fused_kernel"
    elif mask & 256:
        names["stft_mel"] = "stft_mel2_kernel"
    elif mask & 64:
        names["stft_mel"] = "stft_mel_kernel"
    if mask & 16 and not fused:
        names["resample"] = "resample_persist_kernel"
    return names
MFCC_SOURCES = ("mfcc.hip", "stft_bdft.hip", "stft.h", "mfcc_tables.h")


def mfcc_source_sha():
    """sha256 over the MFCC kernel sources: profiles/r04_mfcc_pmc.json records the value its counters were taken at, and the
    bench line says `traffic_stale: true` when the sources have changed since."""
    import hashlib

    h = hashlib.sha256()
    for name in MFCC_SOURCES:
        with open(os.path.join(ROOT, "asr-using-robust-nn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


DTYPE_NAMES = {"float32": "f32", "float16x2": "f32 accumulate, fp16 two-plane split products in the training GEMMs (22-bit operands)", "bfloat16": "bf16"}
POOL_CLIPS = 65536                      # throughput runs loop a resident pool of >= 65 536 clips per GPU (4.2 GB)


def make_pool(n_clips, device, seed):
    """Resident synthetic pool, generated on the device (plumbing): waves [n, 16000] fp32, one-hot labels [n, 10]."""
    from lipasr.synth import synth_clips_device

    waves, labels = synth_clips_device(n_clips, seed, device)
    y = torch.zeros(n_clips, 10, device=device)
    y[torch.arange(n_clips, device=device), labels] = 1
    return waves, y


def run_config(opt, pool, batch, rank, world, device, steps, warmup, profile):
    """Builds model + pipeline for one per-GPU batch, times `steps` steps; returns (seconds, extras).
    opt: dict(constraint=, pgd=, pgd_eps=, bf16=, pre_extracted=, graph="auto"|"on"|"off")."""
    import torch.distributed as dist

    import lipasr._native as N
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.keras import CategoricalCrossentropy
    from lipasr.parallel import DataParallel
    from lipasr.pipeline import TrainPipeline
    from lipasr.train_constraints import get_model

    waves, y = pool
    if opt.get("int16"):  # 16-bit PCM pool: the same clips quantised as a wav file holds them
        waves = (waves * 32768.0).round_().clamp_(-32768, 32767).to(torch.int16)
    n_batches = waves.shape[0] // batch
    # same seed on every rank: replicas start identical
    # arithmetic of the classifier's training GEMMs: "float16x2" (default since round 5: fp16 two-plane split products, fp32 accumulate,
    # 2^-21 per product; inference / attack GEMMs and everything else exact fp32), "float32" (--exact-fp32: fp32 MFMA chains) or "bfloat16"
    model = get_model(max_batch=batch, seed=0, compute_dtype="bfloat16" if opt.get("bf16") else opt.get("compute", "float16x2"))
    model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    # A2 "affine, precomputed": StandardScaler fitted once on the MFCCs of the WHOLE resident pool, as the reference fits it on its
    # whole dataset (train_constraints.py:28-35: fit_transform over train + dev + test) -- until round 4 a four-batch shortcut
    ex = MfccExtractor(16000, 16000, batch, device)
    feats = torch.cat([ex(waves[i * batch:(i + 1) * batch]) for i in range(min(int(os.environ.get("LIPASR_SCALER_BATCHES", "1000000")), n_batches))])
    sc = StandardScaler().fit(feats)
    del feats
    dp = DataParallel()
    comm = dp.probe(device)
    if world > 1:
        dp.broadcast(sc.mean_, sc.scale_, model._params, model._bnstate)
    pgd = dict(eps=opt.get("pgd_eps", 0.5), eps_step=0.1, max_iter=opt["pgd"]) if opt.get("pgd", 0) > 0 else None
    pipe = TrainPipeline(model, batch=batch, rho=0.1, constraint=opt.get("constraint", "product"), affine=(sc.mean_, sc.scale_), pgd=pgd, dp=dp,
                         use_graph={"on": True, "off": False}.get(opt.get("graph", "auto"), "auto"), sync_inputs=False,
                         mfcc_cus=(int(os.environ["LIPASR_PRE_PARTITION"]) if os.environ.get("LIPASR_PRE_PARTITION") else None) if opt.get("pre_extracted") else "auto",  # no extraction stream to keep apart: no CU partition (LIPASR_PRE_PARTITION: A/B probe of the classifier alone on the rest of a partition)
                         overlap_buckets=os.environ.get("LIPASR_DP_OVERLAP", "0") == "1")  # the pool is resident and synchronised before the loop

    fused = os.environ.get("LIPASR_MFCC_FUSED", "0") == "1"  # A/B runs: the fused resample -> STFT kernel for every batch
    if fused:
        pipe.ex.set(2, 1)
    if os.environ.get("LIPASR_MFCC_MASK"):  # A/B runs: 64 = the round-2 STFT kernel
        pipe.ex.set(0, int(os.environ["LIPASR_MFCC_MASK"]))
    feat_pool = None
    pre = bool(opt.get("pre_extracted"))
    if pre:
        n_batches = min(n_batches, 16)
        feat_pool = torch.cat([ex(waves[i * batch:(i + 1) * batch], 44, sc.mean_, sc.scale_) for i in range(n_batches)])

    def one(i):
        s = (i % n_batches) * batch
        if feat_pool is not None:
            pipe.step(None, y[s:s + batch], features=feat_pool[s:s + batch])
        else:
            pipe.step(waves[s:s + batch], y[s:s + batch])

    prewarm_ms = float(os.environ.get("LIPASR_BENCH_PREWARM_MS", "0"))
    if prewarm_ms > 0:  # A/B probe (not a default): is the slowness of the first timed steps the device's clock ramp?
        a = torch.randn(4096, 4096, device=device)
        tw = time.perf_counter()
        while (time.perf_counter() - tw) * 1e3 < prewarm_ms:
            for _ in range(8):
                a = (a @ a).clamp_(-1.0, 1.0)
            torch.cuda.synchronize()
        del a
    for i in range(warmup):
        one(i)
    pipe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if profile and not pre:
        pipe.ex.profile_begin(steps)
    tid = C.c_int()
    N.check(N.lib.lipasr_timer_create(pipe.h.h, C.byref(tid)))
    with torch.cuda.stream(pipe.stream):
        N.check(N.lib.lipasr_timer_start(pipe.h.h, tid.value, N.stream_ptr()))
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    with torch.cuda.stream(pipe.stream):
        N.check(N.lib.lipasr_timer_stop(pipe.h.h, tid.value, N.stream_ptr()))
    t_host = time.perf_counter() - t0  # the host is done enqueueing here; the rest of dt is the GPU catching up
    pipe.synchronize()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ev_ms = C.c_float()
    N.check(N.lib.lipasr_timer_elapsed_ms(pipe.h.h, tid.value, C.byref(ev_ms)))
    extras = {"event_ms_per_step": ev_ms.value / steps, "host_enqueue_ms_per_step": t_host / steps * 1e3, "final_norm": float(pipe.norms[-1].item()) if opt.get("constraint", "product") == "product" else None,
              "loss": float(model._loss_rows[:batch].mean().item())}
    if profile and pre:
        extras["mfcc_ms"] = {"resample": 0.0, "stft_mel": 0.0, "dct": 0.0, "calls": 0}
    elif profile:
        ms3, n = pipe.ex.profile_end()
        extras["mfcc_ms"] = dict(ms3, calls=n)
    extras["mfcc_fused"] = fused
    # the classifier part of the step, timed on ITS stream while the MFCC of the next batch runs beside it (untimed extra
    # steps after the measured region, so that the two event records per step do not touch `value`)
    n_prof = min(20, steps)
    pipe.profile_train(n_prof)
    dp.time_collectives(True)  # HIP events around the gradient all-reduce of these untimed steps
    for i in range(n_prof):
        one(warmup + steps + i)
    pipe.synchronize()
    extras["train_graph_ms"] = pipe.train_ms()
    extras["hip_graph"] = bool(pipe.use_graph)
    extras["allreduce_ms"] = dp.collective_ms()
    extras["comm"] = comm
    standalone = None
    if profile and not pre:
        # the same MFCC kernels alone on the whole chip (untimed, AFTER the measured region: thirteen extractions stream 0.8 GB
        # through the Infinity Cache, and run in front of the timed steps they evicted the classifier's state -- the first dozen
        # timed steps then ran 0.437 -> 0.425 ms instead of 0.42, round 4): the pipeline confines them to a CU share, which
        # lengthens them by design -- both figures are reported
        s_all = torch.cuda.Stream(device=device)
        n_cu = torch.cuda.get_device_properties(device).multi_processor_count
        pipe.ex.set(1, n_cu)  # (three-kernel path: resampler grid for the whole chip)
        with torch.cuda.stream(s_all):
            for k in range(3):
                pipe.ex(waves[k * batch:(k + 1) * batch], 44, sc.mean_, sc.scale_, out=pipe._feats2[0][:batch])
            s_all.synchronize()
            pipe.ex.profile_begin(10)
            for k in range(10):
                j = (3 + k) % n_batches
                pipe.ex(waves[j * batch:(j + 1) * batch], 44, sc.mean_, sc.scale_, out=pipe._feats2[0][:batch])
            ms3, ncalls = pipe.ex.profile_end()
        standalone = {k: round(v, 4) for k, v in ms3.items()}
        pipe.ex.set(1, getattr(pipe, "mfcc_cus", n_cu))  # back to the pipeline's CU share
    extras["mfcc_standalone_ms"] = standalone
    extras["mfcc_cus"] = getattr(pipe, "mfcc_cus", None)
    extras["train_cus"] = getattr(pipe, "train_cus", None)
    extras["mfcc_stream"] = pipe.mfcc_stream_kind
    extras["n_cus"] = torch.cuda.get_device_properties(device).multi_processor_count
    extras["pool_clips"] = int(n_batches * batch)
    assert np.isfinite(extras["loss"]), "training diverged"
    pipe.close()  # graphs, then the masked stream (a hardware queue): handed back before the next configuration builds its own
    model.close()
    ex.close()
    return dt, extras


def run_config_dry(batch, rank, world, steps, warmup):
    """--dry-run-dp: this script's world > 1 control flow on the CPU with the gloo backend and a stand-in replica (a two-layer
    torch classifier on random features; NOT the product and NOT a measurement): communicator probe, broadcast of scaler and
    start state from rank 0, per-step shard -> fwd/bwd -> lipasr.parallel.DataParallel gradient all-reduce -> update, the
    barrier + MAX-over-ranks timing, the JSON line's data-parallel fields.  tests/test_dp_gloo.py runs it under
    torch.distributed.run --nproc-per-node 2, so the branches an 8-GPU node takes first are rehearsed here."""
    import torch.distributed as dist

    from lipasr.parallel import DataParallel

    class Replica:
        def __init__(self, seed):
            g = torch.Generator().manual_seed(seed)
            self.w = [torch.randn(880, 64, generator=g) * 0.03, torch.zeros(64), torch.randn(64, 10, generator=g) * 0.1, torch.zeros(10)]
            self.n = sum(t.numel() for t in self.w)
            self.params = torch.cat([t.reshape(-1) for t in self.w])
            self.grads = torch.zeros(self.n)

        def _views(self):
            o, out = 0, []
            for t in self.w:
                out.append(self.params[o:o + t.numel()].view_as(t))
                o += t.numel()
            return out

        def train_fwd_bwd(self, x, y, inv_batch=None, **kw):
            w1, b1, w2, b2 = [v.clone().requires_grad_(True) for v in self._views()]
            logits = torch.relu(x @ w1 + b1) @ w2 + b2
            loss = -(y * torch.log_softmax(logits, 1)).sum() * inv_batch
            gs = torch.autograd.grad(loss, [w1, b1, w2, b2])
            self.grads.copy_(torch.cat([g.reshape(-1) for g in gs]))
            self.loss = float(loss.detach())

        def apply_adam(self):
            self.params.sub_(0.05 * self.grads)

    dp = DataParallel()
    comm = dp.probe("cpu")
    rep = Replica(seed=100 + rank)  # replicas START different: the broadcast must bring them together
    g = torch.Generator().manual_seed(7)
    pool_x = torch.randn(8 * batch * world, 880, generator=g)
    pool_y = torch.nn.functional.one_hot(torch.randint(0, 10, (8 * batch * world,), generator=g), 10).float()
    mean, scale = pool_x.mean(0), pool_x.std(0)
    if rank != 0:
        mean, scale = torch.zeros_like(mean), torch.ones_like(scale)
    dp.broadcast(mean, scale, rep.params)

    def one(i):
        s = (i % 8) * batch * world
        xb, yb = dp.shard(pool_x[s:s + batch * world], pool_y[s:s + batch * world])
        dp.train_step(rep, (xb - mean) / scale, yb, global_batch=batch * world)

    for i in range(warmup):
        one(i)
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        one(warmup + i)
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    dp.time_collectives(True)
    for i in range(min(5, steps)):
        one(warmup + steps + i)
    return dt, {"comm": comm, "allreduce_ms": dp.collective_ms(), "loss": rep.loss, "divergence": dp.max_divergence(rep.params)}


def _short(opt_over, pool, batch, device, steps, warmup):
    """One secondary single-GPU configuration, measured IN THIS PROCESS after the headline (its own model + pipeline on the same
    resident pool).  Round 3 ran these in child processes because the PGD-20 graph replayed 3x slower as the fifth configuration
    of one process; the cause was the training stream sharing a hardware queue (torch pool streams are multiplexed over 4) --
    TrainPipeline now gives it a queue of its own (scratch/pgd_fifth_probe.py: 4.07 ms as the fifth configuration,
    4.09 alone)."""
    opt = {"constraint": "product", "pgd": 0, "pgd_eps": 0.5, "bf16": False, "pre_extracted": False, "graph": "auto", "int16": False,
           "compute": os.environ.get("LIPASR_COMPUTE", "float16x2")}
    opt.update(opt_over)
    try:
        sub = (pool[0][:16 * batch], pool[1][:16 * batch])  # (a pool shorter than 16 batches is used whole)
        dt, ex = run_config(opt, sub, batch, 0, 1, device, steps, warmup, profile=True)
    except Exception as e:  # the headline must not die with a secondary record
        return {"error": f"{type(e).__name__}: {e}"[:200]}
    cls = TRAIN_FLOP_PER_UTT * batch / (ex["train_graph_ms"] * 1e-3) / 1e12 if not opt["pgd"] and ex["train_graph_ms"] > 0 else None
    return {"value": round(batch * steps / dt, 1), "unit": "utterances/sec", "ms_per_step": round(dt / steps * 1e3, 4), "per_gpu_batch": batch, "steps": steps,
            "dtype": "bf16 operands, f32 accumulate" if opt["bf16"] else DTYPE_NAMES[opt["compute"]], "train_graph_ms": round(ex["train_graph_ms"], 4),
            "classifier_tflops": round(cls, 2) if cls else None, "mfcc_stream": ex.get("mfcc_stream"), "measured_in": "this process"}


def run_fit_api(device, batch=512, n=16384, epochs=3):
    """The reference-shaped entry as train_constraints.py:94-105 drives it: model.fit(dataset, callbacks=[simple_norm_constraint])
    on pre-extracted standardised features, eager launches (one optimizer step, one callback, two metric reductions per
    batch; no HIP graph).  Returns utterances/sec of the last epochs."""
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.keras import CategoricalCrossentropy, Dataset
    from lipasr.train_constraints import get_model

    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(n, 880, generator=g).numpy()
    lab = torch.randint(0, 10, (n,), generator=g).numpy()
    y = np.zeros((n, 10), dtype=np.float32)
    y[np.arange(n), lab] = 1
    model = get_model(max_batch=batch, seed=0)
    model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    ds = Dataset.from_tensor_slices((x, y)).batch(batch)
    cb = simple_norm_constraint(rho=0.1, affected_layers_indices=[])
    model.fit(ds, epochs=1, verbose=0, callbacks=[cb])  # warm-up (plans, first launches)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit(ds, epochs=epochs, verbose=0, callbacks=[cb])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    model.close()
    return {"value": round(n * epochs / dt, 1), "unit": "utterances/sec", "ms_per_step": round(dt / (epochs * (n // batch)) * 1e3, 4),
            "per_gpu_batch": batch, "what": "lipasr.keras.Model.fit + simple_norm_constraint callback, eager (the reference's own driver shape)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=1024)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: --batch-per-gpu clips per rank and step (default); strong: "
                    "--global-batch clips per step over ALL ranks, each rank takes global/N (north_star's 8-GPU bar is worded as strong scaling)")
    ap.add_argument("--global-batch", type=int, default=8192, help="--scaling strong: clips per step over all ranks (BASELINE config 4: 8192)")
    ap.add_argument("--pool-clips", type=int, default=POOL_CLIPS, help="resident waveform pool per GPU (SURVEY 8d: >= 65536)")
    ap.add_argument("--constraint", default="product", choices=["product", "per_layer", "none"])
    ap.add_argument("--pgd", type=int, default=0, help="PGD iterations per batch (config 5 uses 20)")
    ap.add_argument("--pgd-eps", type=float, default=0.5)
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"], help="the classifier's launches as HIP graph(s); auto = the library's "
                    "choice (TrainPipeline: graphs for PGD / synchronized BatchNorm, plain launches otherwise)")
    ap.add_argument("--no-graph", action="store_true", help="= --graph off")
    ap.add_argument("--bf16", action="store_true", help="classifier GEMM operands rounded to bf16 at the MFMA, fp32 accumulate (BASELINE config 2's "
                    "arithmetic); the default and the headline are exact fp32")
    ap.add_argument("--exact-fp32", action="store_true", help="the classifier's training GEMMs on exact fp32 MFMA chains (v_mfma_f32_32x32x2_f32) instead of the "
                    "default fp16 two-plane split (v_mfma_f32_32x32x16_f16 x 3, fp32 accumulate)")
    ap.add_argument("--pre-extracted", action="store_true", help="BASELINE config 2: train from resident (N,880) features, no MFCC stage")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-b512", action="store_true")
    ap.add_argument("--int16", action="store_true", help="the resident pool as 16-bit PCM (what the corpus is): 2 bytes per sample in, scaled by 2^-15 on the device")
    ap.add_argument("--fit-api", action="store_true", help="only time the Keras-shaped fit() entry (used by the main run as a child process)")
    ap.add_argument("--skip-other-configs", action="store_true", help="do not add the config 2 / config 2 bf16 / config 5 records (N = 1 only)")
    ap.add_argument("--dry-run-dp", action="store_true", help="rehearse the world > 1 control flow on the CPU (gloo, stand-in replica); prints a line marked dry_run, not a measurement")
    args = ap.parse_args()
    opt = {"constraint": None if args.constraint == "none" else args.constraint, "pgd": args.pgd, "pgd_eps": args.pgd_eps, "bf16": args.bf16,
           "pre_extracted": args.pre_extracted, "graph": "off" if args.no_graph else args.graph, "int16": args.int16,
           "compute": "float32" if args.exact_fp32 else os.environ.get("LIPASR_COMPUTE", "float16x2")}

    from lipasr.parallel import init_from_env

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_env}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.dry_run_dp:
        import torch.distributed as dist

        rank, world = init_from_env("gloo")
        if args.scaling == "strong":  # the global batch is fixed (scaled down for the CPU stand-in), every rank takes its share
            gb = min(args.global_batch, 128)
            if gb % world:
                raise SystemExit(f"--scaling strong: global batch {gb} is not a multiple of {world} ranks")
            b_rank = gb // world
        else:
            b_rank = min(args.batch_per_gpu, 64)
            gb = b_rank * world
        dt, ex = run_config_dry(b_rank, rank, world, args.steps, args.warmup)
        t = torch.tensor([dt], dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if rank == 0:
            print(json.dumps({"dry_run": True, "metric": "utterances/sec (DRY RUN: CPU stand-in replica over gloo, not a measurement)",
                              "value": round(gb * args.steps / float(t.item()), 1), "unit": "utterances/sec", "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": round(float(t.item()) / args.steps * 1e3, 4), "scaling": args.scaling,
                              "rccl_ranks_seen": ex["comm"]["ranks_seen"], "rccl_version": ex["comm"]["rccl_version"], "comm_backend": ex["comm"]["backend"],
                              "allreduce_ms": round(ex["allreduce_ms"], 4) if ex["allreduce_ms"] is not None else None,
                              "replica_divergence": ex["divergence"], "loss": round(ex["loss"], 4),
                              "config": {"global_batch": gb, "per_gpu_batch": b_rank, "parallelism": f"dp{world}"}}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    rank, world = init_from_env("nccl" if world_env > 1 else None)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist

    if args.fit_api:
        print(json.dumps(run_fit_api(device)), flush=True)
        return
    batch = args.batch_per_gpu
    if args.scaling == "strong":
        if args.global_batch % world or args.global_batch < world:
            raise SystemExit(f"--scaling strong: global batch {args.global_batch} is not a multiple of {world} ranks")
        batch = args.global_batch // world
    pool = make_pool(max(args.pool_clips, 8 * batch) // batch * batch, device, seed=1234 + rank)
    dt, ex = run_config(opt, pool, batch, rank, world, device, args.steps, args.warmup, profile=True)
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
    bytes_per_utt = (32000 + 3520) if args.int16 else MFCC_BYTES_PER_UTT  # the 16-bit PCM variant reads 2 B per sample
    achieved = bytes_per_utt * batch / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else 0.0
    # HBM traffic of the stage per launch: PMC counters cannot be read from inside this process, so the value is
    # the committed rocprofv3 measurement of the same kernels (separate --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH
    # doubled per MI355X_MICROARCH.md), scaled to this batch.
    traffic, traffic_src, traffic_fused, traffic_stale = None, None, None, None
    for name in ("r05_mfcc_pmc.json", "r04_mfcc_pmc.json", "r03_mfcc_pmc.json", "r02_mfcc_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                eor = json.load(f)["end_of_round"]
            traffic_stale = eor.get("source_sha16") != mfcc_source_sha()  # (files before round 4 carry no hash: stale)
            key = "fused_stage_bytes_per_utt" if ex.get("mfcc_fused") and "fused_stage_bytes_per_utt" in eor else "stage_bytes_per_utt"
            traffic = round(eor[key] * batch)
            if "fused_stage_bytes_per_utt" in eor:
                traffic_fused = round(eor["fused_stage_bytes_per_utt"] * batch)
            traffic_src = f"profiles/{name} (rocprofv3 --pmc, per launch, scaled by batch)"
            break
        except Exception:
            continue
    if args.int16:  # the committed counter passes ran on float32 clips
        traffic, traffic_src, traffic_fused = None, None, None
    names = mfcc_kernel_names(bool(ex.get("mfcc_fused")))
    roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                "stage": "MFCC (K1 = " + (" + ".join(dict.fromkeys(v for v in names.values() if not v.startswith("(")))) + ")",
                "dominant_kernel": names[dom],
                "kernel_names": names,
                "algorithmic_bytes_per_utt": bytes_per_utt, "units_per_launch": batch,
                "kernel_ms": {k: round(ms[k], 4) for k in ("resample", "stft_mel", "dct")},
                "fp32_flop_frac": round(MFCC_FLOP_PER_UTT * batch / (stage_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5) if stage_ms > 0 else 0.0,
                "traffic_x_algorithmic": round(traffic / (bytes_per_utt * batch), 3) if traffic else None,
                "fused_path_traffic": traffic_fused,  # the resample + STFT kernel (int16 / ragged input, or LIPASR_MFCC_FUSED=1): y stays in LDS
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
        sa_ach = bytes_per_utt * batch / (sa_ms * 1e-3) / 1e9
        roofline["standalone_whole_chip"] = {"kernel_ms": sa, "achieved": round(sa_ach, 2), "frac": round(sa_ach / HBM_PEAK_GBS, 5),
                                             "fp32_flop_frac": round(MFCC_FLOP_PER_UTT * batch / (sa_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 5)}
    cls_tflops = TRAIN_FLOP_PER_UTT * batch / (ex["train_graph_ms"] * 1e-3) / 1e12 if not args.pgd and ex["train_graph_ms"] > 0 else None
    if args.pre_extracted:
        tf = TRAIN_FLOP_PER_UTT * batch / (ex["event_ms_per_step"] * 1e-3) / 1e12
        roofline = {"bound": "mfma", "achieved": round(tf, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / FP32_PEAK_TFLOPS, 5),
                    "traffic": None, "stage": "dense classifier train step (" + ("fp16 two-plane split on v_mfma_f32_32x32x16_f16 x 3: algorithmic fp32 FLOPs priced "
                                                  "against the fp32 matrix peak" if opt["compute"] == "float16x2" and not args.bf16 else "v_mfma_f32_32x32x2_f32, exact fp32") + ")",
                    "algorithmic_flop_per_utt": TRAIN_FLOP_PER_UTT, "units_per_launch": batch,
                    "note": "whole step incl. BatchNorm, Adam and projection kernels; GEMM-only time is in profiles/"}
    out = {"metric": "utterances/sec (train, 1 s@16 kHz)", "value": round(value, 1), "unit": "utterances/sec", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
           "scaling": args.scaling, "vs_baseline": None, "dtype": "bf16" if args.bf16 else DTYPE_NAMES[opt["compute"]], "data": "synthetic",
           # what "dtype" means here: parameters, activations, BatchNorm, loss, Adam, projection and every inference / attack GEMM are fp32;
           # in mode float16x2 the training GEMMs split each fp32 operand into two fp16 planes and run three of the four cross terms on
           # v_mfma_f32_32x32x16_f16 with fp32 accumulation -- 2^-21 per product (the K1 resampler's and STFT's technique).  The same
           # step on exact fp32 MFMA chains is in "reference_exact_fp32_arithmetic"; tests/ hold both to the same oracle bounds
           "arithmetic": {"training_gemms": opt["compute"], "everything_else": "float32"},
           "config": {"workload": ("pre-extracted standardised (N,880) MFCC features" if args.pre_extracted else "raw 16 kHz waveform -> on-GPU MFCC")
                                  + " -> Lipschitz-constrained MLP train step (Adam+NonNeg, simple_norm_constraint rho=0.1)" + (f" + PGD-{args.pgd} adversarial inner loop" if args.pgd else "")
                                  + (", data-parallel RCCL gradient all-reduce" if world > 1 else ", 1xMI355X"),
                      "baseline_config": 5 if args.pgd else (2 if args.pre_extracted else (4 if world > 1 else 3)), "global_batch": global_batch, "per_gpu_batch": batch,
                      "clip": "1 s @ 16 kHz int16 PCM" if args.int16 else "1 s @ 16 kHz fp32", "resident_pool_clips_per_gpu": ex["pool_clips"], "parallelism": f"dp{world}", "hip_graph": ex["hip_graph"]},
           "roofline": roofline, "mfcc_stream": ex.get("mfcc_stream"),
           # CUs of the two CU-masked streams (disjoint): feature extraction | classifier chain
           "cu_partition": {"mfcc": ex.get("mfcc_cus"), "classifier": ex.get("train_cus") or ex.get("n_cus")},
           # the classifier's part of the step (attack + fwd/bwd [+ all-reduce] + Adam + projection), HIP events on the training
           # stream while the next batch's MFCC runs on its own stream; TFLOP/s = 9.59 MFLOP/utt x batch / that time
           "train_graph_ms": round(ex["train_graph_ms"], 4), "host_enqueue_ms_per_step": round(ex["host_enqueue_ms_per_step"], 4), "classifier_tflops": round(cls_tflops, 2) if cls_tflops else None,
           "classifier_fp32_mfma_frac": round(cls_tflops / FP32_PEAK_TFLOPS, 4) if cls_tflops else None,
           "event_ms_per_step": round(ex["event_ms_per_step"], 4), "final_product_norm": ex["final_norm"], "loss": round(ex["loss"], 4),
           # data parallel: how many ranks the collective backend really reduced over (SUM of ones, not WORLD_SIZE echoed), its
           # version, the gradient all-reduce's duration (HIP events, untimed extra steps) and the share of the classifier's part
           # of the step it exposes (the exchange is not overlapped: DESIGN.md 5)
           "rccl_ranks_seen": ex["comm"]["ranks_seen"], "rccl_version": ex["comm"]["rccl_version"], "comm_backend": ex["comm"]["backend"],
           "allreduce_ms": round(ex["allreduce_ms"], 4) if ex.get("allreduce_ms") is not None else None,
           "allreduce_exposed_frac": round(ex["allreduce_ms"] / ex["train_graph_ms"], 4) if ex.get("allreduce_ms") and ex["train_graph_ms"] > 0 else None}
    if world == 1 and not args.skip_b512 and batch != 512:
        dt5, ex5 = run_config(opt, pool, 512, rank, world, device, args.steps, args.warmup, profile=True)
        out["reference_batch_512"] = {"value": round(512 * args.steps / dt5, 1), "ms_per_step": round(dt5 / args.steps * 1e3, 4),
                                      "mfcc_ms": {k: round(v, 4) for k, v in ex5["mfcc_ms"].items() if k != "calls"},
                                      "train_graph_ms": round(ex5["train_graph_ms"], 4), "event_ms_per_step": round(ex5["event_ms_per_step"], 4),
                                      "mfcc_stream": ex5.get("mfcc_stream")}
    if world == 1 and not args.skip_other_configs and not (args.pgd or args.pre_extracted or args.bf16):
        # the other single-GPU BASELINE configurations, in the driver-run record (short runs; each is its own model + pipeline)
        k, w = min(args.steps, 50), min(args.warmup, 10)
        if opt["compute"] != "float32":  # the headline's step with the training GEMMs on exact fp32 MFMA chains (round 4's arithmetic)
            out["reference_exact_fp32_arithmetic"] = _short({"compute": "float32"}, pool, batch, device, k, w)
        if batch != 8192:
            # north_star's 8-GPU bar is STRONG scaling at global batch 8192: this is its denominator -- the same step with all 8192
            # clips on one GPU (same pool, same pipeline, its own CU split).  8 x 1024-clip shards in T8 ms against this record's
            # ms_per_step is the strong-scaling speed-up; `--scaling strong --gpus N` measures the numerator in that form.
            out["reference_global_batch_8192_1gpu"] = _short({}, pool, 8192, device, min(k, 20), min(w, 5))
        out["reference_config_2_pre_extracted_f32"] = _short({"pre_extracted": True}, pool, batch, device, k, w)
        out["reference_config_2_pre_extracted_bf16"] = _short({"pre_extracted": True, "bf16": True}, pool, batch, device, k, w)
        out["reference_config_5_pgd20_1gpu"] = _short({"pgd": 20, "pgd_eps": 0.5}, pool, batch, device, min(k, 20), min(w, 5))
        # config 3 on 16-bit PCM clips (SURVEY 8d's 35 520 B/utt variant): the resampler reads int16 directly
        out["reference_config_3_int16_pcm_input"] = _short({"int16": True}, pool, batch, device, k, w)
        del pool
        torch.cuda.empty_cache()
        try:  # the Keras-shaped fit() entry with the constraint as a callback, as train_constraints.py drives it
            out["reference_fit_api_batch_512"] = dict(run_fit_api(device), measured_in="this process")
        except Exception as e:
            out["reference_fit_api_batch_512"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        pool = None
    if world == 1 and not args.skip_cpu_baseline:
        pool = None
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(512)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
