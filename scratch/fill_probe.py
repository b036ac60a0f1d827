"""Where the fixed cost of a short timed region goes (bench.py: 20 steps 0.457 ms/step, 200 steps 0.417): per-step completion times
of the training stream after a drained pipeline."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'asr-using-robust-nn_amd'))
import torch
import bench
from lipasr.attacks import StandardScaler
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.keras import CategoricalCrossentropy
from lipasr.pipeline import TrainPipeline
from lipasr.train_constraints import get_model
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
B = 1024
waves, y = bench.make_pool(16 * B, dev, seed=1)
model = get_model(max_batch=B, seed=0); model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
ex = MfccExtractor(16000, 16000, B, dev)
sc = StandardScaler().fit(torch.cat([ex(waves[i * B:(i + 1) * B]) for i in range(2)]))
pipe = TrainPipeline(model, batch=B, rho=0.1, constraint="product", affine=(sc.mean_, sc.scale_), sync_inputs=False)
def one(i):
    s = (i % 16) * B
    pipe.step(waves[s:s + B], y[s:s + B])
for i in range(8): one(i)
pipe.synchronize(); torch.cuda.synchronize()
import sys as _s
spin = int(_s.argv[1]) if len(_s.argv) > 1 else 0   # untimed extractions on a side stream before every timed region
side = torch.cuda.Stream(device=dev)
for rep in range(2):
    K = 20
    if spin:
        with torch.cuda.stream(side):
            for k in range(spin):
                ex(waves[(k % 16) * B:(k % 16 + 1) * B])
        side.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    host = []
    with torch.cuda.stream(pipe.stream):
        evs[0].record(pipe.stream)
    t0 = time.perf_counter()
    for i in range(K):
        one(8 + i)
        host.append(time.perf_counter() - t0)
        evs[i + 1].record(pipe.stream)
    pipe.synchronize(); torch.cuda.synchronize()
    total = time.perf_counter() - t0
    gpu = [evs[0].elapsed_time(e) for e in evs[1:]]
    print(f"rep {rep}: total {total * 1e3:.3f} ms = {total / K * 1e3:.4f} ms/step")
    print("  train-stream completion (ms):", [round(g, 3) for g in gpu])
    print("  step intervals (ms):        ", [round(b - a, 3) for a, b in zip([0.0] + gpu[:-1], gpu)])
    print("  host enqueue done (ms):     ", [round(h * 1e3, 3) for h in host])
pipe.close(); model.close(); ex.close()
