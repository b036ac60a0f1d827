"""BASELINE's full sizes (per-GPU batch 1024, the 880-1024-512-256-128-64-10 network) through properties that do not
need the oracle to run at that size: batch independence, gain equivariance of the MFCC, row-permutation equivariance of
the classifier, the eps-ball and row independence of PGD, the closed form of the product projection -- plus the oracle
itself on a handful of rows drawn from the full batch.
"""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, load_params
from oracle import constraints_ref as R, mfcc_ref as M, mlp_ref as P

pytestmark = pytest.mark.gpu
B = 1024


def _clips(seed):
    from lipasr.synth import synth_clips_fast

    w, lab = synth_clips_fast(B, seed=seed)
    return np.asarray(w, dtype=np.float32), np.asarray(lab)


def test_mfcc_full_batch_independence_gain_and_oracle_rows(cuda):
    from lipasr.extract_features_construct_dataset import MfccExtractor

    w, _ = _clips(11)
    ex = MfccExtractor(16000, 16000, B)
    full = ex(dev(w)).cpu().numpy()
    assert full.shape == (B, 880) and np.isfinite(full).all()
    # (i) a clip's features do not depend on its neighbours or on the launch size: bit-identical in batches of 64
    ex64 = MfccExtractor(16000, 16000, 64)
    for s in (0, 448, 960):
        np.testing.assert_array_equal(ex64(dev(w[s:s + 64])).cpu().numpy(), full[s:s + 64])
    ex = MfccExtractor(16000, 16000, B)  # (the plan belongs to the last extractor built)
    # (ii) gain equivariance: y -> c y adds 20 log10(c) dB to every mel bin (top_db is relative to the clip's own
    # maximum, so the clamp moves with it), which the orthonormal DCT turns into +20 log10(c) sqrt(128) on coefficient 0
    # and exactly nothing on the others
    c = 0.25
    scaled = ex(dev(c * w)).cpu().numpy()
    d = (scaled - full).reshape(B, 20, 44)
    want0 = 20.0 * np.log10(c) * np.sqrt(128.0)
    assert np.abs(d[:, 0, :] - want0).max() < 5e-3
    assert np.abs(d[:, 1:, :]).max() < 5e-3
    # (iii) the oracle on rows drawn from the full batch
    for i in (0, 257, 511, 1023):
        ref = M.extract_features_wave(w[i], 16000).reshape(-1)
        assert np.abs(full[i] - ref).max() < 2e-2


def _full_model(seed=3):
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=False)
    rng = np.random.default_rng(seed)
    for l, s in enumerate(spec):
        if s.bn:
            p.gamma[l] = (1 + 0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_mean[l] = (0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    m = build_model(spec, max_batch=B)
    load_params(m, p)
    return spec, p, m


def test_classifier_full_batch_row_equivariance_and_oracle_rows(cuda):
    spec, p, m = _full_model()
    rng = np.random.default_rng(0)
    x = rng.standard_normal((B, 880)).astype(np.float32)
    out = m.predict(x)
    assert out.shape == (B, 10) and np.abs(out.sum(axis=1) - 1).max() < 1e-5
    perm = rng.permutation(B)
    np.testing.assert_array_equal(m.predict(x[perm]), out[perm])  # inference treats rows independently, bit for bit
    rows = [0, 31, 32, 500, 1023]
    ref = P.forward_infer(spec, p.astype(np.float64), x[rows].astype(np.float64))
    assert np.abs(out[rows] - ref).max() / np.abs(ref).max() < 1e-4
    np.testing.assert_array_equal(out[rows].argmax(1), ref.argmax(1))


def test_pgd_full_batch_ball_strength_and_row_independence(cuda):
    from lipasr.attacks import ProjectedGradientDescent, TensorFlowV2Classifier

    spec, p, m = _full_model(seed=6)
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, 880)).astype(np.float32)
    eps = 0.5
    atk = ProjectedGradientDescent(estimator=clf, eps=eps, eps_step=0.1, max_iter=20, batch_size=B)
    adv = atk.generate(x=x)
    assert np.abs(adv - x).max() <= eps + 1e-5
    y = clf.predict(x).argmax(1)
    y1 = P.to_categorical(y, 10)

    def ce(z):
        pr = np.clip(clf.predict(z), 1e-30, 1.0)
        return float(-np.log(pr[np.arange(len(z)), y]).mean())

    assert ce(adv) > ce(x) + 0.1  # twenty signed steps against the model's own labels raise its loss
    # rows do not interact: the first 256 rows attacked on their own land where they landed inside the full batch
    # (the smaller launch may take a different GEMM tiling, so near-zero gradient components can flip: compare by mass)
    part = ProjectedGradientDescent(estimator=clf, eps=eps, eps_step=0.1, max_iter=20, batch_size=256).generate(x=x[:256])
    agree = np.abs(part - adv[:256]) < 1e-4
    assert agree.mean() > 0.97, agree.mean()
    assert y1.shape == (B, 10)


def test_projection_full_network_closed_form(cuda):
    """simple_norm_constraint(rho=0.1) on the full network: one pass takes the product norm from n0 to
    n0^((5/6)^6) rho^(1-(5/6)^6) (Constraints.py:158-189 visits the six layers in sequence), and repeated passes
    converge to rho."""
    from lipasr.Constraints import simple_norm_constraint

    spec = P.vd_constrained_spec()
    m = build_model(spec, max_batch=64)
    load_params(m, P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=True))
    ws = [l.get_weights()[0] for l in m.layers if "dense" in l.name]
    n0 = R.sigma_max(R.product_chain(ws))
    cb = simple_norm_constraint(rho=0.1, affected_layers_indices=[])
    cb.set_model(m)
    cb.on_batch_end(0)
    ws1 = [l.get_weights()[0] for l in m.layers if "dense" in l.name]
    n1 = R.sigma_max(R.product_chain(ws1))
    q = (5.0 / 6.0) ** 6
    assert abs(n1 - n0 ** q * 0.1 ** (1 - q)) <= 1e-4 * n1
    for i in range(60):
        cb.on_batch_end(i + 1)
    ws2 = [l.get_weights()[0] for l in m.layers if "dense" in l.name]
    assert abs(R.sigma_max(R.product_chain(ws2)) - 0.1) <= 1e-4
    assert all((w >= 0).all() for w in ws2)


def test_gradient_is_additive_over_shards_at_full_size(cuda):
    """The identity the data-parallel step rests on, at config 4's shard size: with 1/global-batch
    folded into the loss gradient, the flat gradient of 1024 rows is the SUM of the gradients of its two 512-row shards
    (a network without BatchNorm: per-replica statistics are the documented deviation)."""
    widths = [880, 1024, 512, 256, 128, 64, 10]
    spec = [P.LayerSpec(a, b, False, 0.0, True) for a, b in zip(widths[:-1], widths[1:])]
    m = build_model(spec, max_batch=B)
    load_params(m, P.init_params(spec, seed=12, dtype=np.float32, nonneg_init=True))
    rng = np.random.default_rng(2)
    x = dev(rng.standard_normal((B, 880)).astype(np.float32) * 0.1)
    y = dev(P.to_categorical(rng.integers(0, 10, B), 10))

    def grad(lo, hi):
        m.train_fwd_bwd(x[lo:hi].contiguous(), y[lo:hi].contiguous(), inv_batch=1.0 / B, dropout=False)
        torch.cuda.synchronize()
        return m._grads.double().cpu().numpy().copy()

    full = grad(0, B)
    parts = grad(0, 512) + grad(512, B)
    assert np.abs(full).max() > 0
    assert np.abs(full - parts).max() <= 2e-5 * np.abs(full).max()
