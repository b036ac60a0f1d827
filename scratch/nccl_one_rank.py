"""RCCL smoke on one GPU: a world-size-1 nccl group next to the pipeline's CU-masked stream (the collective is the
identity here; what is exercised is RCCL's initialisation and its stream interplay with the pipeline)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "asr-using-robust-nn_amd")]
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from lipasr.keras import CategoricalCrossentropy
from lipasr.parallel import DataParallel
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips_fast
from lipasr.train_constraints import get_model
dp = DataParallel()
dp.world = 2  # force the collective code path (SUM over one rank = identity); gradients carry 1/(2 B) as at N = 2
m = get_model(max_batch=512)
m.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
pipe = TrainPipeline(m, batch=512, rho=0.1, constraint="product", dp=dp)
w, lab = synth_clips_fast(512 * 4, seed=1)
wt = torch.as_tensor(w).cuda(); y = torch.nn.functional.one_hot(torch.as_tensor(lab).long(), 10).float().cuda()
for i in range(10): pipe.step(wt[(i % 4) * 512:(i % 4 + 1) * 512], y[(i % 4) * 512:(i % 4 + 1) * 512])
pipe.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): pipe.step(wt[(i % 4) * 512:(i % 4 + 1) * 512], y[(i % 4) * 512:(i % 4 + 1) * 512])
pipe.synchronize(); torch.cuda.synchronize()
print("nccl world-1 pipeline: ms/step", (time.perf_counter() - t0) / 50 * 1e3, "stream", pipe.mfcc_stream_kind, "norm", float(pipe.norms[-1]))
dist.destroy_process_group()
