import sys, os, time; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import numpy as np, torch
from lipasr.pipeline import TrainPipeline
from lipasr.train_constraints import get_model
from lipasr.keras import CategoricalCrossentropy
from lipasr.synth import synth_clips_fast
B=1024
w,l=synth_clips_fast(4*B, seed=1)
wt=torch.as_tensor(w).cuda(); y=torch.zeros(4*B,10,device='cuda'); y[torch.arange(4*B), torch.as_tensor(l).long().cuda()]=1
m=get_model(max_batch=B); m.compile(optimizer='adam', loss=CategoricalCrossentropy())
for graph in (True, False):
    pipe=TrainPipeline(m, batch=B, rho=0.1, use_graph=graph)
    for i in range(6): pipe.step(wt[(i%4)*B:(i%4+1)*B], y[(i%4)*B:(i%4+1)*B])
    pipe.synchronize(); torch.cuda.synchronize()
    ts=[]; t0=time.perf_counter()
    for i in range(40):
        a=time.perf_counter(); pipe.step(wt[(i%4)*B:(i%4+1)*B], y[(i%4)*B:(i%4+1)*B]); ts.append(time.perf_counter()-a)
    t_enq=time.perf_counter()-t0
    pipe.synchronize(); torch.cuda.synchronize(); t_all=time.perf_counter()-t0
    ts=np.array(ts)*1e6
    print(f"graph={graph}: host enqueue per step median {np.median(ts):.0f} us (min {ts.min():.0f}, max {ts.max():.0f}); enqueue total {t_enq*1e3:.1f} ms; wall {t_all*1e3:.1f} ms -> {t_all/40*1e6:.0f} us/step")
