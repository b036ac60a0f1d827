"""Unprofiled two-stream timeline of the training pipeline from HIP events (rocprofv3's tracing slows the host enough
to distort it): per step, when the MFCC of that batch and its training graph start and end."""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
from lipasr.pipeline import TrainPipeline
from lipasr.train_constraints import get_model
from lipasr.keras import CategoricalCrossentropy
from lipasr.synth import synth_clips_fast
import lipasr._native as N
B=1024
w,l=synth_clips_fast(4*B, seed=1)
wt=torch.as_tensor(w).cuda(); y=torch.zeros(4*B,10,device='cuda'); y[torch.arange(4*B), torch.as_tensor(l).long().cuda()]=1
m=get_model(max_batch=B); m.compile(optimizer='adam', loss=CategoricalCrossentropy())
pipe=TrainPipeline(m, batch=B, rho=0.1, use_graph=True)
for i in range(8): pipe.step(wt[(i%4)*B:(i%4+1)*B], y[(i%4)*B:(i%4+1)*B])
pipe.synchronize(); torch.cuda.synchronize()
E=lambda: torch.cuda.Event(enable_timing=True)
rec=[]
class ExWrap:
    def __init__(self, ex): self.ex=ex
    def __getattr__(self, k): return getattr(self.ex, k)
    def __call__(self, *a, **kw):
        a0,a1=E(),E(); a0.record(torch.cuda.current_stream()); r=self.ex(*a, **kw); a1.record(torch.cuda.current_stream()); rec.append(('mfcc',a0,a1)); return r
pipe.ex=ExWrap(pipe.ex)
orig=N.lib.lipasr_graph_launch
class LibWrap:
    def __init__(self, lib): self._lib=lib
    def __getattr__(self, k):
        if k=='lipasr_graph_launch':
            def f(*a):
                a0,a1=E(),E(); a0.record(torch.cuda.current_stream()); r=orig(*a); a1.record(torch.cuda.current_stream()); rec.append(('train',a0,a1)); return r
            return f
        return getattr(self._lib, k)
N.lib=LibWrap(N.lib)
import lipasr.pipeline as P; P.N.lib=N.lib
base=E(); base.record(pipe.stream)
for i in range(10): pipe.step(wt[(i%4)*B:(i%4+1)*B], y[(i%4)*B:(i%4+1)*B])
pipe.synchronize(); torch.cuda.synchronize()
for k,a0,a1 in rec:
    print(f"{k:6s} {base.elapsed_time(a0)*1e3:8.0f} -> {base.elapsed_time(a1)*1e3:8.0f}  ({a0.elapsed_time(a1)*1e3:6.0f} us)")
