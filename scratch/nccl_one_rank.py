"""RCCL rehearsal on one GPU: a world-size-1 nccl group next to the pipeline's CU-masked stream.  dp.world is forced to 2, so
the step runs exactly the multi-GPU schedule -- three HIP graphs, bucket A's asynchronous all-reduce on RCCL's stream beside
the dW_0 graph, bucket B, Adam + projection -- with the collective being the identity; what is exercised is RCCL's
initialisation, async_op on slices of the flat gradient buffer, and the stream hand-offs.  Prints ms/step of that schedule
and of the single-graph one."""
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
from lipasr.synth import synth_clips_device
from lipasr.train_constraints import get_model
B = 1024
wt, lab = synth_clips_device(B * 8, 1, torch.device("cuda", 0))
y = torch.nn.functional.one_hot(lab, 10).float()
for forced, overlap in ((1, False), (2, False), (2, True)):
    dp = DataParallel()
    dp.world = forced  # 2: the collective code path (SUM over one rank = identity); gradients carry 1/(2 B) as at N = 2
    m = get_model(max_batch=B)
    m.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    pipe = TrainPipeline(m, batch=B, rho=0.1, constraint="product", dp=dp, sync_inputs=False, overlap_buckets=overlap)
    for i in range(10): pipe.step(wt[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): pipe.step(wt[(i % 8) * B:(i % 8 + 1) * B], y[(i % 8) * B:(i % 8 + 1) * B])
    pipe.synchronize(); torch.cuda.synchronize()
    tag = " with overlapped buckets" if overlap else ""
    print(f"nccl world-1 pipeline, schedule of world {forced}{tag}: ms/step {(time.perf_counter() - t0) / 100 * 1e3:.4f} stream {pipe.mfcc_stream_kind} "
          f"graphs per step {len(next(iter(pipe._graphs.values())))} norm {float(pipe.norms[-1]):.5f} loss {float(m._loss_rows[:B].mean()):.4f}", flush=True)
    pipe.close(); m.close()
dist.destroy_process_group()
