"""Regenerates tests/golden/*.npz.

    python tests/golden/make_golden.py

constraints.npz, mfcc.npz, mlp.npz are outputs of THIS repo's oracle (parity unpinned: the reference
holds no vectors for the path and cannot run here); they pin the oracle against silent drift and
travel to the GPU box.  ref_labels.npz is reference-held DATA: the label arrays of
"Voice digit recogniton/processed_google_dataset" (split sizes 16566/4733/2366, 10 classes), read
only when /root/reference is present.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import inputs  # noqa: E402
from oracle import constraints_ref as C, mfcc_ref as M, mlp_ref as P  # noqa: E402


def spec_from(cfg):
    w = cfg["widths"]
    return [P.LayerSpec(w[i], w[i + 1], bool(cfg["bn"][i]) and i < len(w) - 2, cfg["dropout"][i], bool(cfg["nonneg"][i])) for i in range(len(w) - 1)]


def main():
    # ---- constraints (A6, A7)
    ws = inputs.nonneg_kernels(inputs.SMALL_WIDTHS)
    nc = C.norm_constraint_pass(ws, 10.0)
    sn, norms = C.simple_norm_constraint_pass(ws, 0.1, [])
    sn_idx, norms_idx = C.simple_norm_constraint_pass(ws, 0.1, [0, 2, 2])
    out = {f"nc_{i}": w for i, w in enumerate(nc)}
    out.update({f"sn_{i}": w for i, w in enumerate(sn)})
    out.update({f"snidx_{i}": w for i, w in enumerate(sn_idx)})
    out["sn_norms"] = np.asarray(norms)
    out["snidx_norms"] = np.asarray(norms_idx)
    out["sigmas"] = C.get_norms(ws)
    np.savez_compressed(os.path.join(HERE, "constraints.npz"), **out)

    # ---- MFCC (A1)
    clips = inputs.test_clips()
    feats = M.compute_mfcc_batch(clips).astype(np.float32)
    y0 = M.librosa_load_resample(clips[0], 16000)
    np.savez_compressed(os.path.join(HERE, "mfcc.npz"), feats=feats, resampled_head=y0[:2048], resampled_tail=y0[-2048:],
                        short=M.compute_mfcc_batch(clips[:2, :7430]).astype(np.float32),
                        long=M.compute_mfcc_batch(np.concatenate([clips[:2], clips[:2, :8000]], axis=1)).astype(np.float32))

    # ---- MLP (A3, A4)
    spec = spec_from(inputs.MLP_SMALL)
    p = P.init_params(spec, seed=5, dtype=np.float32, nonneg_init=True)
    x, y, masks = inputs.mlp_small_case()
    p64 = p.astype(np.float64)
    fb = P.forward_backward(spec, p64, x.astype(np.float64), y.astype(np.float64), masks=masks, training=True, need_dx=True)
    st = P.AdamState()
    p_step = p64.copy()
    P.train_step(spec, p_step, st, x.astype(np.float64), y.astype(np.float64), masks=masks)
    d = {"logits": fb["logits"], "loss": fb["loss"], "dx": fb["dx"], "infer_logits": P.forward_infer(spec, p64, x.astype(np.float64), True)}
    for l in range(len(spec)):
        d[f"W{l}"], d[f"b{l}"] = p.W[l], p.b[l]
        d[f"dW{l}"], d[f"db{l}"] = fb["dW"][l], fb["db"][l]
        d[f"W{l}_after"], d[f"b{l}_after"] = p_step.W[l], p_step.b[l]
        if spec[l].bn:
            d[f"dgamma{l}"], d[f"dbeta{l}"] = fb["dgamma"][l], fb["dbeta"][l]
            d[f"mm{l}_after"], d[f"mv{l}_after"] = p_step.mov_mean[l], p_step.mov_var[l]
    np.savez_compressed(os.path.join(HERE, "mlp.npz"), **d)

    # ---- reference-held data
    ref = "/root/reference/Voice digit recogniton/processed_google_dataset"
    if os.path.isdir(ref):
        np.savez_compressed(os.path.join(HERE, "ref_labels.npz"), train=np.load(os.path.join(ref, "train_label.npy")),
                            dev=np.load(os.path.join(ref, "dev_label.npy")), test=np.load(os.path.join(ref, "test_label.npy")))
        # train_data.npy is a git-LFS pointer; only its declared blob size is kept (a number, not the file)
        with open(os.path.join(ref, "train_data.npy"), "rb") as f:
            size = [int(l.split()[1]) for l in f.read().decode("ascii", "replace").splitlines() if l.startswith("size ")][0]
        with open(os.path.join(HERE, "ref_meta.json"), "w") as f:
            json.dump({"train_data_npy_bytes": size}, f)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
