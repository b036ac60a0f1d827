import sys; sys.path.insert(0,'.'); sys.path.insert(0,'asr-using-robust-nn_amd'); sys.path.insert(0,'tests')
import numpy as np, torch
from helpers import build_model, dev
from oracle import mlp_ref as P
from lipasr.pipeline import TrainPipeline
from lipasr.synth import synth_clips
from lipasr import _native as N
spec = P.vd_constrained_spec()
waves, labels = synth_clips(64, seed=51)
y = dev(P.to_categorical(labels, 10))
for use_graph in (False, True):
    m = build_model(spec, max_batch=32, seed=3)
    pipe = TrainPipeline(m, batch=32, rho=0.1, pgd=dict(eps=0.5, eps_step=0.1, max_iter=20), use_graph=use_graph)
    pipe.step(dev(waves[:32]), y[:32]); pipe.synchronize()
    print('graph',use_graph,'maxdiff', float((pipe.x_adv-pipe.feats).abs().max()), 'finite', bool(torch.isfinite(pipe.x_adv).all()))
    dx = torch.zeros(32,880,device='cuda')
    N.check(N.lib.lipasr_mlp_input_grad(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(pipe.feats), N.ptr(pipe.labels), 32, N.ptr(dx), N.stream_ptr()))
    torch.cuda.synchronize()
    print(' dx absmax', float(dx.abs().max()), 'nan', int(torch.isnan(dx).sum()), 'nonzero frac', float((dx!=0).float().mean()))
    lg = m.predict_device(pipe.feats, logits=True)
    print(' logits range', float(lg.min()), float(lg.max()), 'acc', float((lg.argmax(1)==pipe.labels.argmax(1)).float().mean()))
