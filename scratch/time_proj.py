import sys, os, ctypes as C; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd')); sys.path.insert(0,os.path.join(R,'tests'))
import numpy as np, torch
import lipasr._native as N
from lipasr.train_constraints import get_model
m=get_model(max_batch=64)
for l in [l for l in m.layers if 'dense' in l.name]:
    w,b=l.get_weights(); l.set_weights([np.abs(w),b])
norms=torch.zeros(7,device='cuda'); order=N.int_array(list(range(6)))
h=N.get_handle(0)
s=torch.cuda.Stream()
with torch.cuda.stream(s):
    def call(): N.check(N.lib.lipasr_mlp_project_product(m._plan, N.ptr(m._params), 0.1, order, 6, N.ptr(norms), N.stream_ptr()))
    for _ in range(3): call()
    gid=C.c_int()
    N.check(N.lib.lipasr_graph_begin(h.h, N.stream_ptr())); call(); N.check(N.lib.lipasr_graph_end(h.h, N.stream_ptr(), C.byref(gid)))
    s.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(50): N.check(N.lib.lipasr_graph_launch(h.h, gid.value, N.stream_ptr()))
    e1.record(s); s.synchronize()
print('projection (graph replay) us per call:', e0.elapsed_time(e1)/50*1e3, 'norm', norms.cpu().numpy()[[0,-1]])
