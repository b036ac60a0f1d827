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
    pipe = TrainPipeline(m, batch=per, rho=0.1, constraint="product", dp=dp, use_graph=True)
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
