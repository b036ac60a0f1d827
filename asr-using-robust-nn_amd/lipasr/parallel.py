"""Data parallelism for the training path: one process per GPU, the batch sharded
into contiguous per-rank slices, ONE sum all-reduce per step over the flat fp32 gradient buffer
(torch.distributed backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for tests), the 1/global
batch factor folded into the loss gradient, then Adam + NonNeg + the Lipschitz projection run
redundantly and deterministically on every replica (no second exchange; the kernels use no float
atomics, so replicas stay bit-identical).  The exchange goes in two buckets so that 44 % of it overlaps the
first layer's weight-gradient GEMM (train_step_overlapped / TrainPipeline).  BatchNorm statistics are per
replica by default (tests/test_dp_gloo.py and tests/test_dp_gpu.py check accuracy parity with the single-process run;
``sync_bn=True`` exchanges the column partial sums instead, for runs that must match the single device exactly);
every rank draws its own dropout masks (the rank is folded into the Philox key).

The reference has no counterpart (single process, train_constraints.py:91-105).

A *replica* is anything with ``grads`` (flat tensor), ``train_fwd_bwd(x, y, inv_batch=..., **kw)`` and
``apply_adam()`` -- ``lipasr.keras.Model`` on the GPU, a CPU stand-in in the gloo tests.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun)."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    rank = int(os.environ["RANK"])
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


def shard_bounds(n, rank, world):
    """Contiguous shard [lo, hi) of n items for `rank`; the first n % world ranks get one extra item."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _Done:
    def wait(self):
        return True


class DataParallel:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._timing = None  # time_collectives(): [(start, stop)] per gradient all-reduce

    def shard(self, *tensors):
        """This rank's contiguous slice of each global-batch tensor."""
        out = []
        for t in tensors:
            lo, hi = shard_bounds(t.shape[0], self.rank, self.world)
            out.append(t[lo:hi])
        return out if len(out) > 1 else out[0]

    def broadcast(self, *tensors, src=0):
        if self.world > 1:
            for t in tensors:
                if self._host_staged(t):
                    tmp = t.cpu()
                    dist.broadcast(tmp, src=src, group=self.group)
                    t.copy_(tmp)
                else:
                    dist.broadcast(t, src=src, group=self.group)

    def _host_staged(self, t):
        # gloo cannot reduce device tensors: used only by the single-GPU rehearsal of the DP path (tests)
        return t.is_cuda and dist.get_backend(self.group) == "gloo"

    def time_collectives(self, on=True):
        """Bracket every following allreduce_grads() with timestamps (HIP events on the current stream for device tensors,
        the host clock otherwise); ``collective_ms()`` returns their mean.  bench.py turns this on for its untimed
        profiling steps: how long the exchange is, and so how much of the step it exposes (it is not overlapped)."""
        self._timing = [] if on else None

    def collective_ms(self):
        t = self._timing or []
        self._timing = None
        if not t:
            return None
        if isinstance(t[0][0], float):
            return 1e3 * sum(b - a for a, b in t) / len(t)
        t[-1][1].synchronize()
        return sum(a.elapsed_time(b) for a, b in t) / len(t)

    def probe(self, device):
        """What the communicator really spans: the SUM over ranks of a tensor of ones (= the number of ranks the backend
        reduced over, which is not WORLD_SIZE echoed back), the backend's name and, for RCCL, its version."""
        info = {"world_size_env": self.world, "backend": dist.get_backend(self.group) if dist.is_initialized() else None, "ranks_seen": 1,
                "rccl_version": None}
        if self.world > 1:
            t = torch.ones(8, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
            info["ranks_seen"] = int(round(float(t[0].item())))
            assert bool((t == t[0]).all())
            if info["backend"] == "nccl":
                try:
                    info["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
                except Exception:
                    pass
        return info

    def allreduce_grads(self, flat):
        """The one collective of a step: in-place SUM over the flat gradient buffer."""
        if self._timing is not None and self.world > 1:
            import time

            if flat.is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                self._allreduce(flat)
                e1.record()
                self._timing.append((e0, e1))
            else:
                t0 = time.perf_counter()
                self._allreduce(flat)
                self._timing.append((t0, time.perf_counter()))
            return flat
        return self._allreduce(flat)

    def _allreduce(self, flat):
        if self.world > 1:
            if self._host_staged(flat):
                tmp = flat.cpu()
                dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(tmp)
            else:
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def allreduce_async(self, flat):
        """Start an in-place SUM over ``flat`` (a bucket of the gradient buffer) ordered after the CURRENT stream's work;
        returns a handle whose ``wait()`` orders the current stream after the reduction.  With RCCL the reduction runs on
        the communicator's own stream, beside whatever the caller enqueues next (the first layer's weight-gradient GEMM);
        the host-staged gloo rehearsal and world size 1 complete in place and return a no-op handle."""
        if self.world == 1:
            return _Done()
        if self._host_staged(flat) or dist.get_backend(self.group) != "nccl":
            self.allreduce_grads(flat)
            return _Done()
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def train_step(self, replica, xb, yb, global_batch=None, sync_bn=False, **kw):
        """xb, yb: this rank's shard.  Per-replica gradients carry 1/global_batch, so their SUM is the
        gradient of the mean loss over the global batch.  sync_bn: BatchNorm statistics over the GLOBAL batch
        (``replica.train_fwd_bwd_syncbn``: 2 small all-reduces per BatchNorm layer) instead of per replica."""
        if global_batch is None:
            global_batch = self.global_count(xb.shape[0])
        if hasattr(replica, "_replica_rank"):
            replica._replica_rank = self.rank  # every rank its own dropout masks
        if sync_bn and self.world > 1:
            replica.train_fwd_bwd_syncbn(xb, yb, self, global_batch, **kw)
        else:
            replica.train_fwd_bwd(xb, yb, inv_batch=1.0 / float(global_batch), **kw)
        self.allreduce_grads(replica.grads)
        replica.apply_adam()

    def train_step_overlapped(self, replica, xb, yb, global_batch=None, **kw):
        """The same step with the gradient exchange in two buckets (what TrainPipeline captures as three HIP graphs):
        everything but the first layer's [dW | db] is reduced while that GEMM -- 56 % of the bytes, and the last thing a
        backward pass can start -- still runs.  The replica needs ``train_fwd_bwd(..., defer_dw0=True)``, ``train_dw0(x)``
        and ``late_floats``."""
        if global_batch is None:
            global_batch = self.global_count(xb.shape[0])
        if hasattr(replica, "_replica_rank"):
            replica._replica_rank = self.rank
        replica.train_fwd_bwd(xb, yb, inv_batch=1.0 / float(global_batch), defer_dw0=True, **kw)
        late = replica.late_floats
        ha = self.allreduce_async(replica.grads[late:])
        replica.train_dw0(xb)
        hb = self.allreduce_async(replica.grads[:late])
        ha.wait()
        hb.wait()
        replica.apply_adam()

    def global_count(self, local_n):
        if self.world == 1:
            return local_n
        dev = "cuda" if dist.get_backend(self.group) == "nccl" else "cpu"
        t = torch.tensor([local_n], dtype=torch.int64, device=dev)
        dist.all_reduce(t, group=self.group)
        return int(t.item())

    def max_divergence(self, flat):
        """max |flat - rank0's flat| over ranks (0.0 when the replicas are in sync)."""
        if self.world == 1:
            return 0.0
        mine = flat.cpu() if self._host_staged(flat) else flat
        ref = mine.clone()
        dist.broadcast(ref, src=0, group=self.group)
        d = (mine - ref).abs().max().reshape(1)
        dist.all_reduce(d, op=dist.ReduceOp.MAX, group=self.group)
        return float(d.item())
