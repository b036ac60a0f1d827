"""How much of a step is spent BETWEEN graph replays?  Replays the captured training graph back to back
(a) bare, (b) with the event record / wait pair the pipeline puts between steps, and compares with one replay's span."""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
from lipasr.pipeline import TrainPipeline
from lipasr.train_constraints import get_model
from lipasr.keras import CategoricalCrossentropy
from lipasr.synth import synth_clips_fast
import lipasr._native as N
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
w,l=synth_clips_fast(2*B, seed=1)
wt=torch.as_tensor(w).cuda(); y=torch.zeros(2*B,10,device='cuda'); y[torch.arange(2*B), torch.as_tensor(l).long().cuda()]=1
m=get_model(max_batch=B); m.compile(optimizer='adam', loss=CategoricalCrossentropy())
pipe=TrainPipeline(m, batch=B, rho=0.1, use_graph=True)
for i in range(6): pipe.step(wt[(i%2)*B:(i%2+1)*B], y[(i%2)*B:(i%2+1)*B])
pipe.synchronize(); torch.cuda.synchronize()
g=pipe._graphs[(B,0)][0]
s=pipe.stream
E=lambda: torch.cuda.Event(enable_timing=True)
def run(n, between):
    with torch.cuda.stream(s):
        a,b=E(),E(); a.record(s)
        for _ in range(n):
            N.check(N.lib.lipasr_graph_launch(pipe.h.h, g, N.stream_ptr()))
            between()
        b.record(s); s.synchronize()
    return a.elapsed_time(b)/n*1e3
one=run(1, lambda: None)
bare=run(100, lambda: None)
def evs():
    ev=torch.cuda.Event(); ev.record(s); s.wait_event(ev)
with_ev=run(100, evs)
other=torch.cuda.Stream()
def cross():
    ev=torch.cuda.Event(); ev.record(other); s.wait_event(ev); e2=torch.cuda.Event(); e2.record(s)
with_cross=run(100, cross)
print(f"batch {B}: one replay {one:.1f} us; back-to-back {bare:.1f} us/replay; + record/wait on the same stream {with_ev:.1f}; + wait on another stream's event and a record {with_cross:.1f}")
