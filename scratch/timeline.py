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
pipe=TrainPipeline(m, batch=B, rho=0.1, use_graph=True)
for i in range(6): pipe.step(wt[(i%4)*B:(i%4+1)*B], y[(i%4)*B:(i%4+1)*B])
pipe.synchronize(); torch.cuda.synchronize()
# instrument: wrap ex and graph launch with timing events
evs=[]
orig_ex=pipe.ex.__call__
base=torch.cuda.Event(enable_timing=True); base.record(pipe.stream); 
import types
def step_instr(i):
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True); c=torch.cuda.Event(enable_timing=True); d=torch.cuda.Event(enable_timing=True)
    # replicate pipe.step with events
    bsz=B; bb=pipe._i % pipe._nbuf; pipe._i+=1
    waves=wt[(i%4)*B:(i%4+1)*B]; yy=y[(i%4)*B:(i%4+1)*B]
    from lipasr import _native as N
    with torch.cuda.stream(pipe.mfcc_stream):
        if pipe._ev_free[bb] is not None: pipe.mfcc_stream.wait_event(pipe._ev_free[bb])
        a.record(pipe.mfcc_stream)
        pipe.ex(waves, pipe.L, pipe.mean, pipe.scale, out=pipe._feats2[bb][:bsz])
        pipe._labels2[bb][:bsz].copy_(yy)
        b.record(pipe.mfcc_stream)
        pipe._ev_feat[bb].record(pipe.mfcc_stream)
    with torch.cuda.stream(pipe.stream):
        pipe.stream.wait_event(pipe._ev_feat[bb])
        c.record(pipe.stream)
        g=pipe._graphs[(bsz,bb)]
        N.check(N.lib.lipasr_graph_launch(pipe.h.h, g[0], N.stream_ptr()))
        d.record(pipe.stream)
        ev=torch.cuda.Event(); ev.record(pipe.stream); pipe._ev_free[bb]=ev
    evs.append((a,b,c,d))
for i in range(12): step_instr(i)
pipe.synchronize(); torch.cuda.synchronize()
for i,(a,b,c,d) in enumerate(evs):
    print(f"step {i}: mfcc {base.elapsed_time(a)*1e3:8.0f} -> {base.elapsed_time(b)*1e3:8.0f}   train {base.elapsed_time(c)*1e3:8.0f} -> {base.elapsed_time(d)*1e3:8.0f}")
