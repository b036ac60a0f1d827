"""Exit-path probe for the round-2 exit-time SIGSEGV: builds a TrainPipeline with the CU-masked MFCC stream,
replays its HIP graphs for a few steps and leaves WITHOUT calling pipe.close().

    rocprofv3 --kernel-trace --stats -d gpurun_out/exit_fixed -- python3 scratch/exit_probe.py          # must exit 0
    rocprofv3 --kernel-trace --stats -d gpurun_out/exit_leak  -- python3 scratch/exit_probe.py --leak   # round-2 behaviour

--leak removes lipasr's atexit teardown, i.e. the masked stream, the graph executables and the handle are left to the
HIP runtime's static destructors exactly as before the fix (A/B for the diagnosis; expected to crash under the profiler).
"""
import atexit
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd")):
    sys.path.insert(0, p)

import numpy as np
import torch

import lipasr._native as N
from lipasr.keras import CategoricalCrossentropy
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips_fast
from lipasr.train_constraints import get_model

leak = "--leak" in sys.argv
if leak:
    atexit.unregister(N.shutdown)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
waves, labels = synth_clips_fast(256, seed=1)
y = np.zeros((256, 10), dtype=np.float32)
y[np.arange(256), labels] = 1
wt, yt = torch.as_tensor(waves).to(dev), torch.as_tensor(y).to(dev)
model = get_model(max_batch=128)
model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
pipe = TrainPipeline(model, batch=128, rho=0.1, constraint="product", use_graph=True)
for i in range(6):
    s = (i % 2) * 128
    pipe.step(wt[s:s + 128], yt[s:s + 128])
pipe.synchronize()
print(f"exit_probe: stream={pipe.mfcc_stream_kind} graphs={len(pipe._graphs)} leak={leak} loss={float(model._loss_rows[:128].mean()):.4f}", flush=True)
if leak:
    # keep the objects alive until the interpreter is gone: no __del__-time close either
    import builtins
    builtins._lipasr_probe_keep = (pipe, model)
    TrainPipeline.__del__ = lambda self: None
    type(model).__del__ = lambda self: None
# no pipe.close(), no explicit teardown: the fixed library must leave cleanly on its own
