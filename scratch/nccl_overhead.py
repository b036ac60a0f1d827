"""Where do the ~40 us per step of the collective path go?  One-rank nccl group (the all-reduce is the identity):
 (a) no collective, one graph  (b) two graphs, collective skipped  (c) two graphs + dist.all_reduce  (d) all_reduce
 captured inside one graph (scratch experiment only)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "asr-using-robust-nn_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29534", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from lipasr.keras import CategoricalCrossentropy
from lipasr.parallel import DataParallel
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips_fast
from lipasr.train_constraints import get_model
B = 1024
w, lab = synth_clips_fast(B * 4, seed=1)
wt = torch.as_tensor(w).cuda(); y = torch.nn.functional.one_hot(torch.as_tensor(lab).long(), 10).float().cuda()

def run(tag, world, skip):
    dp = DataParallel(); dp.world = world
    if skip: dp.allreduce_grads = lambda flat: flat
    m = get_model(max_batch=B); m.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    pipe = TrainPipeline(m, batch=B, rho=0.1, constraint="product", dp=dp)
    for i in range(10): pipe.step(wt[(i % 4) * B:(i % 4 + 1) * B], y[(i % 4) * B:(i % 4 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): pipe.step(wt[(i % 4) * B:(i % 4 + 1) * B], y[(i % 4) * B:(i % 4 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize()
    print(f"{tag}: {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms/step", flush=True)
    pipe.close()

run("(a) one graph, no collective", 1, False)
run("(b) two graphs, collective skipped", 2, True)
run("(c) two graphs + nccl all_reduce", 2, False)
dist.destroy_process_group()
