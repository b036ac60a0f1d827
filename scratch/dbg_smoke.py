import sys; sys.path.insert(0,'.'); sys.path.insert(0,'asr-using-robust-nn_amd')
import numpy as np, torch
import lipasr._native as N
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips
from lipasr.train_constraints import get_model
from lipasr.keras import CategoricalCrossentropy
from oracle import constraints_ref as R, mfcc_ref as M, mlp_ref as P
waves, labels = synth_clips(16, seed=2)
y = P.to_categorical(labels, 10)
model = get_model(max_batch=16); model.compile(optimizer="adam", loss=CategoricalCrossentropy())
dense = [l for l in model.layers if "dense" in l.name]
for l in dense:
    w,b=l.get_weights(); l.set_weights([np.abs(w),b])
pipe = TrainPipeline(model, batch=16, rho=0.1, use_graph=False)
wt=torch.as_tensor(waves).cuda(); yt=torch.as_tensor(y).cuda()
with torch.cuda.stream(pipe.stream):
    feats=pipe.ex(wt,44)
    model.train_fwd_bwd(feats, yt, dropout=False)
    print('grad absmax', float(model._grads.abs().max()), 'loss', float(model._loss_rows[:16].mean()))
    model.apply_adam()
    w0=[l.get_weights()[0] for l in dense]
    print('w after adam', [float(np.abs(w).max()) for w in w0], [float((w==0).mean()) for w in w0])
    N.check(N.lib.lipasr_mlp_project_product(model._plan, N.ptr(model._params), 0.1, pipe._order, 6, N.ptr(pipe.norms), N.stream_ptr()))
pipe.synchronize()
print('norms', pipe.norms.cpu().numpy())
print('w after proj', [float(np.abs(l.get_weights()[0]).max()) for l in dense])
