"""World-size-2 gloo tests of the data-parallel path on CPU.

The replica is a CPU stand-in with the Model's DP-facing interface (``grads``, ``train_fwd_bwd(defer_dw0=)``,
``train_dw0``, ``late_floats``, ``apply_adam``) whose arithmetic is the oracle; what is under test is lipasr.parallel:
contiguous sharding, the broadcast of the start state, the SUM all-reduce over the flat gradient buffer -- as one
message and as the two buckets the pipeline overlaps with the first layer's weight-gradient GEMM -- with 1/global_batch
folded into the loss gradient, uneven shards, replicas staying identical; and, for the reference's BatchNorm + dropout
model, that per-replica batch statistics and per-rank dropout masks leave the accuracy where the single process has it.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _paths():
    for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)


class CpuReplica:
    """Oracle-backed replica; flat layout per layer [W | b | gamma | beta] like the native plan's."""

    def __init__(self, spec, seed=4):
        from oracle import mlp_ref as P

        self.P, self.spec = P, spec
        self.p = P.init_params(spec, seed=seed, dtype=np.float64, nonneg_init=any(s.nonneg for s in spec))
        self.st = P.AdamState()
        self._replica_rank = 0
        self.step = 0
        self.n = sum(self._sizes(l)[-1] for l in range(len(spec)))
        self.grads = torch.zeros(self.n, dtype=torch.float64)

    def _sizes(self, l):
        s = self.spec[l]
        w, b = s.n_in * s.n_out, s.n_out
        g = s.n_out if s.bn else 0
        return w, b, g, w + b + 2 * g

    @property
    def late_floats(self):
        w, b, _, _ = self._sizes(0)
        return w + b

    def _flat(self, out, skip_dw0):
        parts = []
        for l, s in enumerate(self.spec):
            zero = skip_dw0 and l == 0
            parts += [np.zeros(s.n_in * s.n_out) if zero else out["dW"][l].ravel(), np.zeros(s.n_out) if zero else out["db"][l].ravel()]
            if s.bn:
                parts += [out["dgamma"][l].ravel(), out["dbeta"][l].ravel()]
        return np.concatenate(parts)

    def train_fwd_bwd(self, x, y, inv_batch=None, defer_dw0=False, **kw):
        # every rank its own dropout masks: the rank is part of the key (as in the native Philox key)
        rng = np.random.default_rng([1234, self._replica_rank, self.step])
        masks = [((rng.uniform(size=(x.shape[0], s.n_out)) > s.dropout) / (1 - s.dropout)) if s.dropout > 0 else None for s in self.spec]
        self.out = self.P.forward_backward(self.spec, self.p, x.numpy(), y.numpy(), masks=masks, training=True)
        self.scale = x.shape[0] * inv_batch  # the oracle divides by the local batch; DP wants 1/global
        flat = self._flat(self.out, defer_dw0) * self.scale
        if defer_dw0:
            self.grads[self.late_floats:].copy_(torch.from_numpy(flat[self.late_floats:]))
        else:
            self.grads.copy_(torch.from_numpy(flat))

    def train_dw0(self, x):
        flat = self._flat(self.out, False) * self.scale
        self.grads[:self.late_floats].copy_(torch.from_numpy(flat[:self.late_floats]))

    def apply_adam(self):
        g = self.grads.numpy()
        P = self.P
        self.st.t += 1
        self.step += 1
        o = 0
        for l, s in enumerate(self.spec):
            items = [("W", self.p.W[l]), ("b", self.p.b[l])] + ([("gamma", self.p.gamma[l]), ("beta", self.p.beta[l])] if s.bn else [])
            for name, arr in items:
                gg = g[o:o + arr.size].reshape(arr.shape); o += arr.size
                key = (name, l)
                if key not in self.st.m:
                    self.st.m[key] = np.zeros_like(arr); self.st.v[key] = np.zeros_like(arr)
                P.adam_update(arr, gg, self.st.m[key], self.st.v[key], self.st.t)
            if s.nonneg:
                self.p.W[l] *= (self.p.W[l] >= 0)
            if s.bn:
                mu, var = self.out["stats"][l]
                self.p.mov_mean[l] = self.p.mov_mean[l] * P.BN_MOMENTUM + mu * (1 - P.BN_MOMENTUM)
                self.p.mov_var[l] = self.p.mov_var[l] * P.BN_MOMENTUM + var * (1 - P.BN_MOMENTUM)

    def flat_weights(self):
        parts = []
        for l, s in enumerate(self.spec):
            parts += [self.p.W[l].ravel(), self.p.b[l].ravel()]
            if s.bn:
                parts += [self.p.gamma[l].ravel(), self.p.beta[l].ravel(), self.p.mov_mean[l].ravel(), self.p.mov_var[l].ravel()]
        return torch.from_numpy(np.concatenate(parts))

    def flat_trainables(self):
        parts = []
        for l, s in enumerate(self.spec):
            parts += [self.p.W[l].ravel(), self.p.b[l].ravel()] + ([self.p.gamma[l].ravel(), self.p.beta[l].ravel()] if s.bn else [])
        return torch.from_numpy(np.concatenate(parts))

    def load_flat_weights(self, t):
        a, o = t.numpy(), 0
        for l, s in enumerate(self.spec):
            names = ["W", "b"] + (["gamma", "beta", "mov_mean", "mov_var"] if s.bn else [])
            for nm in names:
                arr = getattr(self.p, nm)[l]
                arr[...] = a[o:o + arr.size].reshape(arr.shape); o += arr.size


def _plain_spec():
    from oracle import mlp_ref as P

    # no BatchNorm, no dropout: per-replica and global statistics coincide -> exact gradient parity
    return [P.LayerSpec(12, 10, False, 0.0, True), P.LayerSpec(10, 8, False, 0.0, True), P.LayerSpec(8, 4, False, 0.0, True)]


def _bn_spec():
    from oracle import mlp_ref as P

    # the reference model's structure in small: Dense -> BatchNorm -> Dropout blocks, softmax head
    return [P.LayerSpec(16, 32, True, 0.2, False), P.LayerSpec(32, 16, True, 0.2, False), P.LayerSpec(16, 4, False, 0.0, False)]


def _data(n=15, d=12, c=4, seed=9):
    rng = np.random.default_rng(seed)
    x = torch.from_numpy(rng.standard_normal((n, d)))
    y = torch.zeros(n, c, dtype=torch.float64)
    y[torch.arange(n), torch.from_numpy(rng.integers(0, c, n))] = 1
    return x, y


def _blobs(n, seed):
    """4 Gaussian classes in 16 dimensions: a task a small BatchNorm net learns in a few hundred steps."""
    rng = np.random.default_rng(seed)
    centers = np.random.default_rng(77).standard_normal((4, 16)) * 1.5
    lab = rng.integers(0, 4, n)
    x = centers[lab] + rng.standard_normal((n, 16))
    y = np.zeros((n, 4)); y[np.arange(n), lab] = 1
    return torch.from_numpy(x), torch.from_numpy(y), lab


def _worker(rank, world, port, out_dir, mode):
    _paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from lipasr.parallel import DataParallel, init_from_env

    init_from_env("gloo")
    dp = DataParallel()
    if mode == "plain":
        # replicas START DIFFERENT (seed = rank) and are brought together by the broadcast, as bench.py does with the
        # scaler and the parameters; uneven shards (15 rows -> 8 + 7) with the true global batch in the loss gradient
        rep = CpuReplica(_plain_spec(), seed=4 + rank)
        w = rep.flat_weights()
        dp.broadcast(w)
        rep.load_flat_weights(w)
        x, y = _data()
        grads_first = None
        for step in range(3):
            xb, yb = dp.shard(x, y)
            (dp.train_step_overlapped if step % 2 == 0 else dp.train_step)(rep, xb, yb, global_batch=None if step == 1 else x.shape[0])
            if step == 0:
                grads_first = rep.grads.clone()
        div = dp.max_divergence(rep.flat_weights())
        torch.save({"grads": grads_first, "weights": rep.flat_weights(), "div": div, "shard": dp.shard(x).shape[0]}, os.path.join(out_dir, f"r{rank}.pt"))
    else:
        rep = CpuReplica(_bn_spec(), seed=2)
        x, y, _ = _blobs(512, seed=1)
        for epoch in range(12):
            for s in range(0, 512, 64):
                xb, yb = dp.shard(x[s:s + 64], y[s:s + 64])
                dp.train_step_overlapped(rep, xb, yb, global_batch=64)
        div = dp.max_divergence(rep.flat_trainables())  # trainables are synchronous; moving statistics are per replica
        xt, _, lt = _blobs(1000, seed=2)
        acc = float(np.mean(rep.P.forward_infer(rep.spec, rep.p, xt.numpy()).argmax(1) == lt))
        torch.save({"acc": acc, "div": div, "mask_rank": rep._replica_rank}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
def test_two_ranks_equal_one_rank(tmp_path):
    _paths()
    from lipasr.parallel import DataParallel

    # single-process reference, same schedule of step kinds
    rep = CpuReplica(_plain_spec(), seed=4)
    dp = DataParallel()
    x, y = _data()
    g1 = None
    for step in range(3):
        (dp.train_step_overlapped if step % 2 == 0 else dp.train_step)(rep, x, y, global_batch=x.shape[0])
        if step == 0:
            g1 = rep.grads.clone()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "plain"), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["shard"] == 8 and r1["shard"] == 7  # uneven shards
    assert r0["div"] == 0.0 and r1["div"] == 0.0  # replicas bit-identical after broadcast + redundant updates
    torch.testing.assert_close(r0["grads"], r1["grads"], rtol=0, atol=0)
    torch.testing.assert_close(r0["grads"], g1, rtol=1e-12, atol=1e-15)
    torch.testing.assert_close(r0["weights"], rep.flat_weights(), rtol=1e-10, atol=1e-13)


@pytest.mark.timeout(300)
def test_batchnorm_and_dropout_under_data_parallel(tmp_path):
    """Per-replica BatchNorm statistics (32 rows per rank instead of 64) and per-rank dropout masks: the trainables stay
    identical across ranks, and the accuracy lands where the single process has it (+-0.5 pt on 1000 test points...
    the toy task is learned to ~99 % either way)."""
    _paths()
    from lipasr.parallel import DataParallel

    rep = CpuReplica(_bn_spec(), seed=2)
    dp = DataParallel()
    x, y, _ = _blobs(512, seed=1)
    for epoch in range(12):
        for s in range(0, 512, 64):
            dp.train_step_overlapped(rep, x[s:s + 64], y[s:s + 64], global_batch=64)
    xt, _, lt = _blobs(1000, seed=2)
    acc1 = float(np.mean(rep.P.forward_infer(rep.spec, rep.p, xt.numpy()).argmax(1) == lt))
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "bn"), nprocs=2, join=True)
    r0 = torch.load(tmp_path / "r0.pt")
    r1 = torch.load(tmp_path / "r1.pt")
    assert r0["div"] == 0.0 and r1["div"] == 0.0
    assert (r0["mask_rank"], r1["mask_rank"]) == (0, 1)
    assert acc1 > 0.9 and r0["acc"] > 0.9, (acc1, r0["acc"], r1["acc"])
    assert abs(r0["acc"] - acc1) <= 0.005 and abs(r1["acc"] - acc1) <= 0.005, (acc1, r0["acc"], r1["acc"])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_dry_run_dp_under_torchrun(scaling):
    """bench.py --dry-run-dp under torch.distributed.run with two ranks: the script's own world > 1 flow
    -- env rendezvous on 127.0.0.1, communicator probe (SUM of ones = ranks the backend really reduced over), broadcast of
    the scaler and the start state from rank 0 to replicas that start DIFFERENT, shard -> all-reduce -> update steps, barrier
    + MAX-over-ranks timing, one JSON line from rank 0 with the data-parallel fields.  Round 5: in both
    scaling modes -- weak (per-rank batch fixed, the default) and strong (`--scaling strong`: the GLOBAL batch fixed, each rank
    takes global / N; the form north_star's 8-GPU bar is worded in)."""
    import json
    import subprocess

    port = _free_port() if "_free_port" in globals() else 29533
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2", "--dry-run-dp", "--scaling", scaling]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # rank 0 only
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["rccl_ranks_seen"] == 2 and d["comm_backend"] == "gloo"
    assert d["scaling"] == scaling
    # weak: 64 per rank -> 128 global; strong: the global batch (8192, scaled down to 128 for the CPU stand-in) split 64 | 64
    assert d["replica_divergence"] == 0.0 and d["allreduce_ms"] > 0 and d["config"]["global_batch"] == 128 and d["config"]["per_gpu_batch"] == 64
    assert np.isfinite(d["loss"]) and d["value"] > 0
