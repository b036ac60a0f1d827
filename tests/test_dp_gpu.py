"""GPU rehearsal of the data-parallel path on ONE MI355X: two processes share cuda:0, the gradient all-reduce goes
through gloo (host-staged), everything else is the production path: per-rank shards, HIP graphs split around the
collective, redundant Adam + NonNeg + projection.  The 8-GPU RCCL run itself is the driver's (bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _spec():
    from oracle import mlp_ref as P

    # no BatchNorm / dropout: per-replica statistics would (by design) differ from a single big batch
    return [P.LayerSpec(880, 256, False, 0.0, True), P.LayerSpec(256, 64, False, 0.0, True), P.LayerSpec(64, 10, False, 0.0, True)]


def _run(dp, steps=3):
    from helpers import build_model, dev, load_params
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mlp_ref as P

    spec = _spec()
    m = build_model(spec, max_batch=64)
    load_params(m, P.init_params(spec, seed=8, dtype=np.float32, nonneg_init=True))
    waves, labels = synth_clips(64 * steps, seed=61)
    y = P.to_categorical(labels, 10)
    per = 64 // dp.world
    pipe = TrainPipeline(m, batch=per, rho=0.1, constraint="product", dp=dp, use_graph=True)  # one collective per step
    for s in range(steps):
        xb, yb = dp.shard(dev(waves[64 * s:64 * (s + 1)]), dev(y[64 * s:64 * (s + 1)]))
        pipe.step(xb.contiguous(), yb.contiguous())
    pipe.synchronize()
    return m, pipe


def _worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lipasr.parallel import DataParallel, init_from_env

    torch.cuda.set_device(0)
    init_from_env("gloo")
    dp = DataParallel()
    m, pipe = _run(dp)
    div = dp.max_divergence(m._params)
    torch.save({"params": m._params.cpu(), "norms": pipe.norms.cpu(), "div": div, "step": int(m._step.item())}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_one_rank(cuda, tmp_path):
    from lipasr.parallel import DataParallel

    m1, pipe1 = _run(DataParallel())
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["div"] == 0.0 and r1["div"] == 0.0 and r0["step"] == 3
    assert torch.equal(r0["params"], r1["params"])  # replicas bitwise identical
    ref = m1._params.cpu()
    # two half-batch gradients summed by the collective vs one full-batch gradient: fp32 summation order only
    d = (r0["params"] - ref).abs() / ref.abs().max()
    assert float(d.quantile(0.999)) < 1e-4 and float(d.max()) < 5e-2
    torch.testing.assert_close(r0["norms"], pipe1.norms.cpu(), rtol=1e-3, atol=0)


# ------------------------------------------------------------------ the reference's BatchNorm + dropout model under data parallel
def _bn_run(dp, out_dir=None):
    """train_google_dataset.py's classifier (five BatchNorm layers, Dropout 0.4 after each: the model that converges in a
    dozen epochs) on MFCCs of synthetic clips, global batch 128, per-replica BatchNorm statistics, per-rank dropout masks,
    through TrainPipeline (three HIP graphs around the two gradient buckets when world > 1).  Returns test accuracy."""
    from helpers import build_model, dev, load_params
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import mfcc
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mlp_ref as P

    spec = P.vd_unconstrained_spec()
    m = build_model(spec, max_batch=512, seed=1)
    load_params(m, P.init_params(spec, seed=1, dtype=np.float32))
    waves, labels = synth_clips(1536, seed=91)
    feats = torch.cat([mfcc(waves[s:s + 512]) for s in range(0, 1536, 512)])
    sc = StandardScaler().fit(feats)
    x = ((feats.double() - sc.mean_) / sc.scale_).float()
    y = dev(P.to_categorical(labels, 10))
    per = 128 // dp.world
    pipe = TrainPipeline(m, batch=per, rho=0.1, constraint=None, dp=dp, use_graph=True, overlap_buckets=True)  # two buckets
    for epoch in range(12):
        for s in range(0, 1024, 128):
            xb, yb = dp.shard(x[s:s + 128], y[s:s + 128])
            pipe.step(None, yb.contiguous(), features=xb.contiguous())
    pipe.synchronize()
    acc = float((m.predict_device(x[1024:]).argmax(1).cpu().numpy() == labels[1024:]).mean())
    pipe.close()
    return m, acc


def _bn_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lipasr.parallel import DataParallel, init_from_env

    torch.cuda.set_device(0)
    init_from_env("gloo")
    dp = DataParallel()
    m, acc = _bn_run(dp)
    div = dp.max_divergence(m._params)
    torch.save({"acc": acc, "div": div, "bn": m._bnstate.cpu()}, os.path.join(out_dir, f"b{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_batchnorm_model_two_ranks_accuracy_parity(cuda, tmp_path):
    """SURVEY 7 hard part 4 / 8e: per-replica BatchNorm statistics are allowed if accuracy parity is shown.  Two ranks
    (64 rows each) against one rank (128 rows): trainables identical across ranks, moving statistics per replica, final
    top-1 accuracy within +-0.5 pt of the single-process run."""
    from lipasr.parallel import DataParallel

    m1, acc1 = _bn_run(DataParallel())
    mp.spawn(_bn_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "b0.pt")
    r1 = torch.load(tmp_path / "b1.pt")
    print(f"\nBatchNorm + dropout model: 1 rank {acc1:.4f}, 2 ranks {r0['acc']:.4f} / {r1['acc']:.4f}")
    assert r0["div"] == 0.0 and r1["div"] == 0.0
    assert not torch.equal(r0["bn"], r1["bn"])  # moving statistics ARE per replica
    assert acc1 > 0.95 and min(r0["acc"], r1["acc"]) > 0.95
    assert abs(r0["acc"] - acc1) <= 0.005 and abs(r1["acc"] - acc1) <= 0.005


def test_dropout_masks_differ_by_rank(cuda):
    """the Philox key was (seed, local element, layer, step) -- identical on every rank.  The rank is part of
    the key now: same weights, same input, same step -> different masks -> different gradients; same rank -> same."""
    from helpers import build_model, dev, load_params
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=2, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(0)
    x = dev(rng.standard_normal((64, 880)))
    y = dev(P.to_categorical(rng.integers(0, 10, 64), 10))
    grads = []
    for rank in (0, 1, 0):
        m = build_model(spec, max_batch=64, seed=3)
        load_params(m, p)
        m._replica_rank = rank
        m.train_fwd_bwd(x, y)
        torch.cuda.synchronize()
        grads.append(m._grads.clone())
        m.close()
    assert torch.equal(grads[0], grads[2])
    assert not torch.equal(grads[0], grads[1])


def test_head_plus_dw0_equals_one_call(cuda):
    """lipasr_mlp_train_fwd_bwd_head + lipasr_mlp_train_dw0 (the data-parallel pair) against lipasr_mlp_train_fwd_bwd:
    everything past late_floats bit-identical (the same launches), the first layer's [dW | db] from a different GEMM
    tiling (LDS-tiled 64x64 instead of the grouped split-K kernel): fp32 summation order only."""
    from helpers import build_model, dev, load_params
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=5, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(1)
    x = dev(rng.standard_normal((512, 880)))
    y = dev(P.to_categorical(rng.integers(0, 10, 512), 10))
    m = build_model(spec, max_batch=512)
    load_params(m, p)
    m.train_fwd_bwd(x, y, dropout=False)
    whole = m._grads.clone()
    bn = m._bnstate.clone()
    m2 = build_model(spec, max_batch=512)
    load_params(m2, p)
    m2._grads.fill_(float("nan"))
    m2.train_fwd_bwd(x, y, dropout=False, defer_dw0=True)
    late = m2.late_floats
    assert late == 880 * 1024 + 1024
    assert torch.isnan(m2._grads[:late]).all()           # the head leaves [dW_0 | db_0] alone
    untouched = torch.isnan(m2._grads[late:])            # the alignment pads between segments are never written
    assert int(untouched.sum()) < 32 and float(whole[late:][untouched].abs().max() if untouched.any() else 0.0) == 0.0
    if getattr(m, "_compute_dtype", "float32") == "float16x2":
        # arithmetic mode 2 chooses the weight-gradient tile by how many 128 x 128 tiles the GROUP makes: with layer 1 in the group
        # they cover 40 % of the chip (128 x 128 split-pass tiles), without it they do not (64 x 64 ring tiles) -- the same sums in
        # another fp32 order, each path bitwise reproducible in itself
        a, b = m2._grads[late:][~untouched], whole[late:][~untouched]
        assert float((a - b).abs().max()) <= 2e-6 * float(b.abs().max())
    else:
        assert torch.equal(m2._grads[late:][~untouched], whole[late:][~untouched])
    m2.train_dw0(x)
    d = (m2._grads[:late] - whole[:late]).abs().max() / whole[:late].abs().max()
    assert float(d) < 2e-5, float(d)
    assert torch.equal(m2._bnstate, bn)


# ------------------------------------------------------------------ synchronized BatchNorm (opt-in)
def _sync_run(dp, sync_bn, steps=4):
    from helpers import build_model, dev, load_params
    from lipasr.pipeline import TrainPipeline
    from oracle import mlp_ref as P

    # the reference's constrained model with its five BatchNorm layers, dropout off so that runs are comparable bit for bit
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    m = build_model(spec, max_batch=64)
    load_params(m, P.init_params(spec, seed=6, dtype=np.float32, nonneg_init=True))
    rng = np.random.default_rng(17)
    x = dev(rng.standard_normal((64 * steps, 880)))
    y = dev(P.to_categorical(rng.integers(0, 10, 64 * steps), 10))
    per = 64 // dp.world
    pipe = TrainPipeline(m, batch=per, rho=0.1, constraint="product", dp=dp, use_graph=True, sync_bn=sync_bn)
    for s in range(steps):
        xb, yb = dp.shard(x[64 * s:64 * (s + 1)], y[64 * s:64 * (s + 1)])
        pipe.step(None, yb.contiguous(), features=xb.contiguous())
    pipe.synchronize()
    return m, pipe


def _sync_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lipasr.parallel import DataParallel, init_from_env

    torch.cuda.set_device(0)
    init_from_env("gloo")
    dp = DataParallel()
    out = {}
    for sync in (True, False):
        m, pipe = _sync_run(dp, sync)
        out[sync] = {"params": m._params.cpu(), "bn": m._bnstate.cpu(), "div": dp.max_divergence(m._params), "norms": pipe.norms.cpu()}
        pipe.close()
    torch.save(out, os.path.join(out_dir, f"s{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_batchnorm_two_ranks_match_one_rank(cuda, tmp_path):
    """Opt-in synchronized BatchNorm (lipasr_mlp_train_segment + a SUM all-reduce of the column partial sums between
    segments): two ranks of 32 rows reproduce the single process's 64-row run of the five-BatchNorm model -- weights AND
    moving statistics -- to fp32 summation order, which per-replica statistics (the default) do not."""
    from lipasr.parallel import DataParallel

    m1, pipe1 = _sync_run(DataParallel(), False)
    ref, ref_bn = m1._params.cpu(), m1._bnstate.cpu()
    mp.spawn(_sync_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "s0.pt")
    r1 = torch.load(tmp_path / "s1.pt")
    for sync in (True, False):
        assert r0[sync]["div"] == 0.0 and r1[sync]["div"] == 0.0
    assert torch.equal(r0[True]["bn"], r1[True]["bn"])  # synchronized: identical moving statistics on both ranks
    scale = ref.abs().max()
    d_sync = ((r0[True]["params"] - ref).abs() / scale)
    d_rep = ((r0[False]["params"] - ref).abs() / scale)
    bn_sync = (r0[True]["bn"] - ref_bn).abs().max() / ref_bn.abs().max()
    bn_rep = (r0[False]["bn"] - ref_bn).abs().max() / ref_bn.abs().max()
    print(f"\nSyncBN vs 1 rank: params q999 {float(d_sync.quantile(0.999)):.2e} max {float(d_sync.max()):.2e}, moving stats {float(bn_sync):.2e}; "
          f"per-replica BN: params q999 {float(d_rep.quantile(0.999)):.2e}, moving stats {float(bn_rep):.2e}")
    assert float(bn_sync) < 1e-5                      # the statistics ARE the global batch's
    assert float(d_sync.quantile(0.999)) < 1e-4       # as tight as the no-BatchNorm parity test above
    assert float(bn_rep) > 10 * float(bn_sync)        # ... which per-replica statistics are not
    torch.testing.assert_close(r0[True]["norms"], pipe1.norms.cpu(), rtol=1e-3, atol=0)


# ------------------------------------------------------------------ round 4: uneven shards and the PGD inner loop under synchronized BatchNorm
def _sync_uneven_run(dp, split, pgd, steps=3):
    """64 rows per step cut into `split` rows per rank (40 | 24 rows are 2 | 1 row tiles of 32: the partial-sum
    buffers of the two ranks have different shapes; they are reduced to [2][width] before the exchange)."""
    from helpers import build_model, dev, load_params
    from lipasr.pipeline import TrainPipeline
    from oracle import mlp_ref as P

    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    m = build_model(spec, max_batch=64)
    load_params(m, P.init_params(spec, seed=8, dtype=np.float32, nonneg_init=True))
    rng = np.random.default_rng(23)
    x = dev(rng.standard_normal((64 * steps, 880)))
    y = dev(P.to_categorical(rng.integers(0, 10, 64 * steps), 10))
    lo = sum(split[:dp.rank]) if dp.world > 1 else 0
    n = split[dp.rank] if dp.world > 1 else 64
    pipe = TrainPipeline(m, batch=max(split) if dp.world > 1 else 64, rho=0.1, constraint="product", dp=dp, use_graph=True, sync_bn=True,
                         pgd=dict(eps=0.3, eps_step=0.1, max_iter=3) if pgd else None)
    for s in range(steps):
        xb, yb = x[64 * s + lo:64 * s + lo + n], y[64 * s + lo:64 * s + lo + n]
        pipe.step(None, yb.contiguous(), features=xb.contiguous(), global_batch=64)
    pipe.synchronize()
    return m, pipe


def _sync_uneven_worker(rank, world, port, out_dir):
    for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lipasr.parallel import DataParallel, init_from_env

    torch.cuda.set_device(0)
    init_from_env("gloo")
    dp = DataParallel()
    out = {}
    for pgd in (False, True):
        m, pipe = _sync_uneven_run(dp, (40, 24), pgd)
        out[pgd] = {"params": m._params.cpu(), "bn": m._bnstate.cpu(), "div": dp.max_divergence(m._params)}
        pipe.close()
    torch.save(out, os.path.join(out_dir, f"u{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_sync_batchnorm_uneven_shards_and_pgd(cuda, tmp_path):
    """Synchronized BatchNorm with shards of 40 and 24 rows (different row-tile counts per rank) and, second, with the PGD
    inner loop in front of the step (it raised NotImplementedError): both reproduce the single-process
    run of the 64-row batch, every piece between two collectives replayed as a HIP graph."""
    import torch.multiprocessing as mp
    from lipasr.parallel import DataParallel

    ref = {}
    for pgd in (False, True):
        m1, pipe1 = _sync_uneven_run(DataParallel(), (64,), pgd)
        ref[pgd] = (m1._params.cpu(), m1._bnstate.cpu())
        pipe1.close()
    mp.spawn(_sync_uneven_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp_path, "u0.pt")), torch.load(os.path.join(tmp_path, "u1.pt"))
    for pgd in (False, True):
        assert r0[pgd]["div"] == 0.0 and r1[pgd]["div"] == 0.0
        assert torch.equal(r0[pgd]["bn"], r1[pgd]["bn"])
        p_ref, bn_ref = ref[pgd]
        d = ((r0[pgd]["params"] - p_ref).abs() / p_ref.abs().max()).quantile(0.999)
        dbn = (r0[pgd]["bn"] - bn_ref).abs().max() / bn_ref.abs().max()
        print(f"\nSyncBN 40|24 rows, pgd={pgd}: params q999 {float(d):.2e}, moving stats {float(dbn):.2e}")
        assert float(dbn) < 1e-5 and float(d) < 1e-4
