"""Shared helpers for the GPU parity tests: oracle <-> product plumbing (no arithmetic of their own)."""
import numpy as np
import torch

from lipasr import keras as K
from oracle import mlp_ref as P


def build_model(spec, max_batch=1024, seed=0, compute_dtype=None):
    """lipasr.keras.Model with the structure of an oracle LayerSpec list (compute_dtype: None = the library's default, exact fp32)."""
    K.reset_layer_names()
    inp = K.Input((spec[0].n_in,))
    node = inp
    for i, s in enumerate(spec):
        last = i == len(spec) - 1
        node = K.Dense(s.n_out, activation="softmax" if last else "relu", kernel_constraint=K.NonNeg() if s.nonneg else None)(node)
        if not last and s.bn:
            node = K.BatchNormalization()(node)
        if not last and s.dropout > 0:
            node = K.Dropout(s.dropout)(node)
    m = K.Model(inputs=inp, outputs=node, max_batch=max_batch, seed=seed, compute_dtype=compute_dtype)
    m.compile(optimizer="adam", loss=K.CategoricalCrossentropy(), metrics=["accuracy"])
    return m


def load_params(model, p):
    """Copy oracle Params into the model through the layer protocol (set_weights)."""
    dense = [l for l in model.layers if "dense" in l.name]
    bns = {l._index: l for l in model.layers if "batch" in l.name}
    for i, l in enumerate(dense):
        l.set_weights([np.asarray(p.W[i], dtype=np.float32), np.asarray(p.b[i], dtype=np.float32)])
        if i in bns:
            bns[i].set_weights([np.asarray(a, dtype=np.float32) for a in (p.gamma[i], p.beta[i], p.mov_mean[i], p.mov_var[i])])


def read_params(model, spec):
    """Model weights back into an oracle Params (float64)."""
    p = P.Params()
    dense = [l for l in model.layers if "dense" in l.name]
    bns = {l._index: l for l in model.layers if "batch" in l.name}
    for i, l in enumerate(dense):
        w, b = l.get_weights()
        p.W.append(w.astype(np.float64)); p.b.append(b.astype(np.float64))
        if i in bns:
            g, be, mm, mv = (a.astype(np.float64) for a in bns[i].get_weights())
        else:
            g = be = mm = mv = None
        p.gamma.append(g); p.beta.append(be); p.mov_mean.append(mm); p.mov_var.append(mv)
    return p


def grads_of(model, spec):
    """Gradient buffer split like the oracle's dict (float64 numpy)."""
    from lipasr import _native as N

    out = {"dW": [], "db": [], "dgamma": [], "dbeta": []}
    g = model._grads
    for i, s in enumerate(spec):
        def seg(kind):
            off, cnt = model._segs[(i, kind)]
            return g[off:off + cnt].cpu().numpy().astype(np.float64) if cnt else None
        out["dW"].append(seg(N.SEG_W).reshape(s.n_in, s.n_out)); out["db"].append(seg(N.SEG_B))
        out["dgamma"].append(seg(N.SEG_GAMMA)); out["dbeta"].append(seg(N.SEG_BETA))
    return out


def dev(x, device="cuda"):
    return torch.as_tensor(np.asarray(x, dtype=np.float32)).to(device).contiguous()


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _oracle_mfcc_chunk(chunk):
    from oracle import mfcc_ref as M

    return M.compute_mfcc_batch(chunk)


def oracle_mfcc_parallel(waves, workers=None):
    """oracle.mfcc_ref.compute_mfcc_batch (the exact resampler, ~0.1 s per clip) over a process pool.  `spawn`: the
    parent may hold a HIP context, which must not be forked."""
    import multiprocessing as mp
    import os

    n = len(waves)
    workers = workers or max(1, min(16, len(os.sched_getaffinity(0)), (n + 15) // 16))
    if workers == 1:
        return _oracle_mfcc_chunk(waves)
    chunks = np.array_split(np.asarray(waves), workers * 4)
    with mp.get_context("spawn").Pool(workers) as pool:
        return np.concatenate(pool.map(_oracle_mfcc_chunk, [c for c in chunks if len(c)]))
