"""GPU: deterministic teardown (DESIGN "exit-time SIGSEGV").  A child process builds a TrainPipeline on the
CU-masked stream, replays its graphs and exits WITHOUT close(): lipasr's atexit teardown must leave rc 0.  The profiled
variant of the same probe (the configuration that crashed in round 2) is run once per round by hand:
profiles/r03_exit_probe.txt."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_exit_without_close_is_clean(cuda):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scratch", "exit_probe.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "exit_probe: stream=" in r.stdout


def test_close_releases_graphs_and_stream(cuda):
    import numpy as np

    import lipasr._native as N
    from helpers import build_model, dev
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    m = build_model(spec, max_batch=32)
    waves, labels = synth_clips(32, seed=5)
    pipe = TrainPipeline(m, batch=32, rho=0.1, use_graph=True)
    pipe.step(dev(waves), dev(P.to_categorical(labels, 10)))
    pipe.synchronize()
    gids = [g for gs in pipe._graphs.values() for g in gs]
    assert gids
    pipe.close()
    pipe.close()  # idempotent
    assert pipe.mfcc_stream_kind == "shared" and not pipe._graphs
    # the executables are gone on the native side: launching one is an argument error, not a crash
    assert N.lib.lipasr_graph_launch(pipe.h.h, gids[0], N.stream_ptr()) == N.EINVAL
    with pytest.raises(RuntimeError):
        pipe.step(dev(waves), dev(P.to_categorical(labels, 10)))
    # a second pipeline on the same handle works after the first was closed
    pipe2 = TrainPipeline(m, batch=32, rho=0.1, use_graph=True)
    pipe2.step(dev(waves), dev(P.to_categorical(labels, 10)))
    pipe2.synchronize()
    assert torch.isfinite(m._params).all()
    pipe2.close()
    m.close()
    m.close()


def test_finaliser_during_capture_does_not_break_it(cuda):
    """A native destroy (hipFree synchronises) invalidates a HIP graph capture of the same thread.  Python finalisers run when
    the garbage collector decides, e.g. in the middle of TrainPipeline._capture (found in round 3 as a once-in-a-while
    'operation failed due to a previous error during capture'): destroys are parked during a capture and run after it."""
    import gc

    import numpy as np

    import lipasr._native as N
    from helpers import build_model, dev
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    m = build_model(spec, max_batch=32)
    waves, labels = synth_clips(32, seed=6)
    pipe = TrainPipeline(m, batch=32, rho=0.1, use_graph=True)
    victims = [build_model(spec, max_batch=32), MfccExtractor(16000, 16000, 4)]
    inner = pipe._attack_and_train

    def noisy(*a, **k):  # what the collector may do at any point of the captured region
        victims.clear()
        gc.collect()
        assert N._deferred, "the destroys should have been parked"
        return inner(*a, **k)

    pipe._attack_and_train = noisy
    pipe.step(dev(waves), dev(P.to_categorical(labels, 10)))
    pipe.synchronize()
    assert not N._deferred and torch.isfinite(m._params).all()
    pipe._attack_and_train = inner
    pipe.step(dev(waves), dev(P.to_categorical(labels, 10)))  # replays the captured graph
    pipe.synchronize()
    pipe.close()
    m.close()


def test_handle_close_closes_its_owners(cuda):
    """lipasr_destroy frees every plan made on the handle; Handle.close() therefore closes the Python owners first
    (newest first), so that a model / extractor / pipeline used afterwards raises instead of handing a freed plan to the
    library.  In a child process: it ends this process's handle."""
    code = r'''
import sys, os
R = sys.argv[1]
for p in (R, os.path.join(R, "asr-using-robust-nn_amd"), os.path.join(R, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import lipasr._native as N
from helpers import build_model, dev
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips
from oracle import mlp_ref as P
spec = P.vd_constrained_spec()
m = build_model(spec, max_batch=32)
ex = MfccExtractor(16000, 16000, 8)
waves, labels = synth_clips(32, seed=7)
pipe = TrainPipeline(m, batch=32, rho=0.1, use_graph=True)
pipe.step(dev(waves), dev(P.to_categorical(labels, 10))); pipe.synchronize()
N.get_handle(0).close()
assert m._plan is None and ex._plan is None and pipe._closed
for call in (lambda: ex(dev(waves[:8])), lambda: pipe.step(dev(waves), dev(P.to_categorical(labels, 10))),
             lambda: m.train_fwd_bwd(torch.zeros(8, 880, device="cuda"), torch.zeros(8, 10, device="cuda"))):
    try:
        call()
    except (RuntimeError, ValueError):
        continue
    raise SystemExit("a closed owner accepted a call")
pipe.synchronize()  # the wrappers of the destroyed queues were replaced
m2 = build_model(spec, max_batch=32)  # a fresh handle serves new owners
ex2 = MfccExtractor(16000, 16000, 8)
assert torch.isfinite(ex2(dev(waves[:8]))).all()
print("handle_close: ok")
'''
    r = subprocess.run([sys.executable, "-c", code, ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "handle_close: ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-2500:])
