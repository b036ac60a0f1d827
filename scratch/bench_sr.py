"""Speaker-recognition pipeline throughput: 1-s windows at 22 050 Hz -> 441/220 MFCC (2020) ->
2020-1024-...-20 classifier step with simple_norm_constraint(rho = 1), one MI355X.  Not the headline bench."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "asr-using-robust-nn_amd"))
import torch
from lipasr.keras import CategoricalCrossentropy
from lipasr.pipeline import TrainPipeline
from lipasr.speaker_recognition import WindowMfcc, get_model

for B in (64, 1024):
    torch.manual_seed(0)
    w = 0.1 * torch.randn(4 * B, 22050, device="cuda")
    y = torch.zeros(4 * B, 20, device="cuda"); y[torch.arange(4 * B), torch.randint(0, 20, (4 * B,))] = 1
    m = get_model(max_batch=B); m.compile(optimizer="adam", loss=CategoricalCrossentropy())
    pipe = TrainPipeline(m, batch=B, utterance_length=101, rho=1.0, constraint="product", extractor=WindowMfcc(batch_max=B))
    for i in range(10): pipe.step(w[(i % 4) * B:(i % 4 + 1) * B], y[(i % 4) * B:(i % 4 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 100
    for i in range(n): pipe.step(w[(i % 4) * B:(i % 4 + 1) * B], y[(i % 4) * B:(i % 4 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"SR pipeline batch {B}: {dt * 1e3:.3f} ms/step -> {B / dt:,.0f} windows/s, product norm {float(pipe.norms[-1]):.4f}")
