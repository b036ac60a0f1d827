"""CPU suite: the oracle against known answers, the committed golden vectors and the reference-held data.

The reference has no tests and no stored outputs for this path and its modules cannot run here
(TensorFlow / librosa / ART absent), so the oracle is PARITY UNPINNED; what pins it against drift is
below: mathematical known answers, cross-checks against SciPy, finite differences, and the fixtures
written by tests/golden/make_golden.py.
"""
import json
import os

import numpy as np
import pytest
import scipy.fft
import scipy.fftpack
import scipy.signal

from golden import inputs
from oracle import attacks_ref as A, constraints_ref as C, mfcc_ref as M, mlp_ref as P


# ------------------------------------------------------------------ reference-held data
def test_reference_label_fixtures(golden_dir):
    lab = np.load(os.path.join(golden_dir, "ref_labels.npz"))
    assert lab["train"].shape == (16566,) and lab["dev"].shape == (4733,) and lab["test"].shape == (2366,)
    for k in ("train", "dev", "test"):
        assert lab[k].dtype == np.int32
        hist = np.bincount(lab[k], minlength=10)
        assert hist.shape == (10,) and hist.min() > 0.8 * hist.mean()  # near-balanced 10 classes
    meta = json.load(open(os.path.join(golden_dir, "ref_meta.json")))
    # float64 (16566, 880) + 128-byte npy header
    assert meta["train_data_npy_bytes"] == 128 + 16566 * 880 * 8


def test_feature_layout_is_coeff_major():
    m = np.arange(20 * 44, dtype=np.float32).reshape(20, 44)
    flat = m.flatten()
    assert flat[3 * 44 + 7] == m[3, 7]  # index = coeff*44 + frame (extract_features...py:145-149)
    assert M.STANDARD_UTTERANCE_LENGTH == 1 + 22050 // 512


# ------------------------------------------------------------------ MFCC oracle
def test_resample_loop_equals_vectorised():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(700).astype(np.float32)
    a = M.resample_kaiser_best(x, 16000, 22050)
    b = M.resample_kaiser_best_loop(x, 16000, 22050)
    assert len(a) == int(700 * 22050 / 16000)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


def test_resample_tone_is_a_tone():
    t = np.arange(16000) / 16000.0
    x = np.sin(2 * np.pi * 1000 * t).astype(np.float32)
    y = M.resample_kaiser_best(x, 16000, 22050)
    t2 = np.arange(len(y)) / 22050.0
    mid = slice(200, -200)
    np.testing.assert_allclose(y[mid], np.sin(2 * np.pi * 1000 * t2)[mid], atol=5e-6)


def test_librosa_fix_length():
    x = np.ones(15999, dtype=np.float32)
    assert len(M.librosa_load_resample(x, 16000)) == 22049  # resampy gives 22048, librosa pads to ceil
    x = np.ones(7430, dtype=np.float32)
    y = M.librosa_load_resample(x, 16000)
    assert len(y) == 10240 and y[-1] == 0.0
    assert M.extract_features_wave(x).shape == (20, 44)
    assert 1 + len(y) // 512 == 21


def test_tables_against_scipy():
    np.testing.assert_allclose(M.hann_periodic(), scipy.signal.get_window("hann", 2048, fftbins=True), atol=1e-15)
    x = np.random.default_rng(1).standard_normal((128, 5))
    np.testing.assert_allclose(M.dct_matrix() @ x, scipy.fftpack.dct(x, type=2, norm="ortho", axis=0)[:20], atol=1e-12)
    fb = M.mel_filterbank()
    assert fb.shape == (128, 1025) and fb.dtype == np.float32 and fb.min() >= 0
    assert np.all((fb > 0).sum(axis=1) >= 1)
    # slaney normalisation: every triangle integrates to ~1 over frequency
    area = fb.sum(axis=1) * (11025.0 / 1024)
    np.testing.assert_allclose(area[5:], 1.0, rtol=0.2)
    assert ((fb > 0).sum(axis=0) <= 2).all()  # at most two filters per bin -> sparse evaluation is exact


def test_mfcc_edge_cases():
    z = M.extract_features_wave(np.zeros(16000, dtype=np.float32))
    assert z.shape == (20, 44)
    np.testing.assert_allclose(z[0], -100.0 * np.sqrt(128.0), rtol=1e-6)  # all bins at 10 log10(1e-10)
    np.testing.assert_allclose(z[1:], 0.0, atol=1e-4)
    short = M.extract_features_wave(inputs.test_clips()[0][:7430])
    assert np.all(short[:, 21:] == 0.0) and np.any(short[:, 20] != 0.0)  # literal zero padding of the matrix
    long = M.extract_features_wave(np.concatenate([inputs.test_clips()[0]] * 2))
    assert long.shape == (20, 44)


def test_oracle_rounding_budget():
    """Bounds the oracle's own float64-vs-float32 choices: the DCT evaluated in
    float32 (scipy.fftpack's dtype) and an all-float32 FFT move the features by far less than the
    standardised-feature budget that the 1e-3 logit tolerance leaves."""
    clips = inputs.test_clips()
    worst_dct, worst_fft = 0.0, 0.0
    for c in clips:
        y = M.librosa_load_resample(c, 16000)
        S = M.power_spectrogram(y)
        db = M.power_to_db(M.mel_filterbank() @ S)
        ref = M.dct_matrix() @ db.astype(np.float64)
        f32 = scipy.fftpack.dct(db.astype(np.float32), type=2, norm="ortho", axis=0)[:20]
        worst_dct = max(worst_dct, np.abs(ref - f32).max())
        yp = M.reflect_pad(y, 1024)
        idx = np.arange(2048)[:, None] + 512 * np.arange(S.shape[1])[None, :]
        fr = (M.hann_periodic().astype(np.float32)[:, None] * yp[idx]).astype(np.float32)
        S32 = (np.abs(scipy.fft.rfft(fr, axis=0)) ** 2).astype(np.float32)
        db32 = M.power_to_db(M.mel_filterbank() @ S32)
        worst_fft = max(worst_fft, np.abs(M.dct_matrix() @ db32.astype(np.float64) - ref).max())
    assert worst_dct < 2e-3, worst_dct
    assert worst_fft < 5e-2, worst_fft


def test_golden_mfcc(golden_dir):
    g = np.load(os.path.join(golden_dir, "mfcc.npz"))
    clips = inputs.test_clips()
    np.testing.assert_allclose(M.compute_mfcc_batch(clips).astype(np.float32), g["feats"], rtol=0, atol=1e-4)
    y0 = M.librosa_load_resample(clips[0], 16000)
    np.testing.assert_allclose(y0[:2048], g["resampled_head"], atol=1e-7)


# ------------------------------------------------------------------ constraints oracle
def test_norm_constraint_known_answer():
    ws = inputs.nonneg_kernels(inputs.FULL_WIDTHS)
    out = C.norm_constraint_pass(ws, 10.0)
    for w in out:
        assert abs(C.sigma_max(w) - 10.0 ** (1 / 6)) < 2e-6  # 1.4677993
        assert w.min() >= 0


def test_simple_norm_closed_form_and_visits():
    ws = inputs.nonneg_kernels(inputs.SMALL_WIDTHS)
    out, norms = C.simple_norm_constraint_pass(ws, 0.1, [])
    m = len(ws)
    for k, n in enumerate(norms):
        assert abs(n - C.simple_norm_closed_form(norms[0], 0.1, m, k)) / n < 1e-5
    # listed indices: visited last-to-first, once per occurrence
    out2, norms2 = C.simple_norm_constraint_pass(ws, 0.1, [0, 2, 2])
    assert len(norms2) == 4
    np.testing.assert_array_equal(out2[1], ws[1])
    s = [(0.1 / (n + C.EPS)) ** (1 / m) for n in norms2[:-1]]
    np.testing.assert_allclose(out2[2], ws[2] * np.float32(s[0]) * np.float32(s[1]), rtol=1e-6)
    np.testing.assert_allclose(out2[0], ws[0] * np.float32(s[2]), rtol=1e-6)


def test_custom_constraint_is_frobenius():
    w = inputs.signed_kernels([30, 20])[0]
    out = C.custom_constraint(w, 3.0)
    assert abs(np.linalg.norm(out) - 3.0) < 1e-5 and out.min() >= 0


def test_fista_runs_and_locates_by_value():
    ws = inputs.nonneg_kernels([24, 16, 12, 6], seed=2)
    out = C.fista_pass(ws, 5.0, 2)
    assert [o.shape for o in out] == [w.shape for w in ws]
    assert all(o.dtype == np.float32 and o.min() >= 0 for o in out)
    dup = [ws[0], ws[1], ws[1].copy()[:, :12] if False else ws[2]]
    first = C.fista_projection(ws[1], ws, 5.0, 2)
    np.testing.assert_allclose(first, C.fista_projection(ws[1].copy(), dup[:2] + [ws[2]], 5.0, 2))


def test_lipschitz_readouts():
    ws = inputs.nonneg_kernels(inputs.SMALL_WIDTHS)
    norms = C.get_norms(ws)
    assert C.get_upper_lipschitz(norms) >= C.sigma_max(C.product_chain(ws)) * (1 - 1e-6)
    bn = [(np.full(64, 2.0), np.full(64, 9.0)), (np.full(32, 1.0), np.full(32, 4.0))]
    assert abs(C.get_lipschitz_constrained(ws, bn) - C.sigma_max(C.product_chain(ws)) / (1.5 * 2.0)) < 1e-6


def test_golden_constraints(golden_dir):
    g = np.load(os.path.join(golden_dir, "constraints.npz"))
    ws = inputs.nonneg_kernels(inputs.SMALL_WIDTHS)
    for i, w in enumerate(C.norm_constraint_pass(ws, 10.0)):
        np.testing.assert_allclose(w, g[f"nc_{i}"], rtol=1e-6)
    out, norms = C.simple_norm_constraint_pass(ws, 0.1, [])
    np.testing.assert_allclose(norms, g["sn_norms"], rtol=1e-6)
    for i, w in enumerate(out):
        np.testing.assert_allclose(w, g[f"sn_{i}"], rtol=1e-6)


# ------------------------------------------------------------------ MLP oracle
def _fd_check(training):
    spec = [P.LayerSpec(12, 9, True, 0.0, True), P.LayerSpec(9, 7, False, 0.0, True), P.LayerSpec(7, 4, False, 0.0, True)]
    p = P.init_params(spec, seed=1, dtype=np.float64)
    for l in range(2):
        if p.gamma[l] is not None:
            p.gamma[l] = p.gamma[l] * 1.3
            p.mov_var[l] = p.mov_var[l] * 0.7
    rng = np.random.default_rng(5)
    x = rng.standard_normal((6, 12))
    y = P.to_categorical(rng.integers(0, 4, 6), 4).astype(np.float64)
    out = P.forward_backward(spec, p, x, y, training=training, need_dx=True)

    def loss_at(pp, xx):
        return P.forward_backward(spec, pp, xx, y, training=training)["loss"]

    eps = 1e-6
    for l in range(3):
        for idx in [(0, 0), (3, 2)]:
            pp = p.copy(); pp.W[l][idx] += eps
            pm = p.copy(); pm.W[l][idx] -= eps
            fd = (loss_at(pp, x) - loss_at(pm, x)) / (2 * eps)
            assert abs(fd - out["dW"][l][idx]) < 1e-6 + 1e-4 * abs(fd)
    xp = x.copy(); xp[2, 5] += eps
    xm = x.copy(); xm[2, 5] -= eps
    fd = (loss_at(p, xp) - loss_at(p, xm)) / (2 * eps)
    assert abs(fd - out["dx"][2, 5]) < 1e-6 + 1e-4 * abs(fd)


def test_finite_differences_train():
    _fd_check(True)


def test_finite_differences_infer():
    _fd_check(False)


def test_loss_is_finite_in_float32_with_saturated_logits():
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=0, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(0)
    x = (5 * rng.standard_normal((16, 880))).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 10, 16), 10)
    out = P.forward_backward(spec, p, x, y, training=True)
    assert np.isfinite(out["loss"])


def test_adam_matches_keras_form():
    w = np.array([1.0, -2.0]); g = np.array([0.5, -0.25]); m = np.zeros(2); v = np.zeros(2)
    P.adam_update(w, g, m, v, 1)
    lr_t = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w, np.array([1.0, -2.0]) - lr_t * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-7), rtol=1e-12)


def test_scaler_constant_columns():
    x = np.random.default_rng(0).standard_normal((50, 4))
    x[:, 1] = 3.0
    x[:, 2] = 1e8 + 1e-9 * np.arange(50)  # round-off variance only
    mean, scale = P.standard_scaler_fit(x)
    assert scale[1] == 1.0 and scale[2] == 1.0 and abs(scale[0] - x[:, 0].std()) < 1e-12


def test_shuffle_structure():
    b = P.tf_shuffle_batches(2000, 512, buffer_size=880, seed=1)
    order = np.concatenate(b)
    assert sorted(order) == list(range(2000)) and [len(x) for x in b] == [512, 512, 512, 464]
    assert all(order[i] < i + 880 for i in range(2000))


def test_golden_mlp(golden_dir):
    from golden.make_golden import spec_from

    g = np.load(os.path.join(golden_dir, "mlp.npz"))
    spec = spec_from(inputs.MLP_SMALL)
    p = P.init_params(spec, seed=5, dtype=np.float32, nonneg_init=True).astype(np.float64)
    x, y, masks = inputs.mlp_small_case()
    fb = P.forward_backward(spec, p, x.astype(np.float64), y.astype(np.float64), masks=masks, training=True, need_dx=True)
    np.testing.assert_allclose(fb["logits"], g["logits"], rtol=1e-10)
    np.testing.assert_allclose(fb["dW"][0], g["dW0"], rtol=1e-9, atol=1e-14)


# ------------------------------------------------------------------ attacks oracle
def test_pgd_stays_in_eps_ball_and_steps():
    spec = [P.LayerSpec(10, 8, True, 0.0, False), P.LayerSpec(8, 3, False, 0.0, False)]
    p = P.init_params(spec, seed=2, dtype=np.float64)
    x = np.random.default_rng(3).standard_normal((5, 10))
    adv1 = A.fgsm(spec, p, x, 0.3, batch_size=2)
    assert np.all(np.isin(np.round(np.abs(adv1 - x), 12), [0.0, 0.3]))
    adv = A.pgd(spec, p, x, eps=0.25, eps_step=0.1, max_iter=7, batch_size=2)
    assert np.abs(adv - x).max() <= 0.25 + 1e-12
    one = A.pgd(spec, p, x, eps=10.0, eps_step=0.1, max_iter=1, batch_size=5)
    np.testing.assert_allclose(one, A.fgsm(spec, p, x, 0.1, batch_size=5), atol=1e-12)
    np.testing.assert_allclose(A.sign_step(x, x, np.ones_like(x), 0.1, np.inf), x + 0.1)


def test_noise_models_statistics():
    rng = np.random.default_rng(0)
    x = np.zeros(200000)
    assert abs(A.add_white_noise(x, 0.05, rng).std() - 0.05) < 1e-3
    n = A.add_noise(x, 0.01, 0.002, rng)
    frac = 2 * (0.5 - 0.49601)  # P(|N(0,1)| < 0.01) ~ 0.00798
    assert abs(n.var() - ((1 - frac) * 0.002 ** 2 + frac * 0.02 ** 2)) < 2e-7
    sig = np.sin(np.arange(200000) * 0.01)
    noisy = A.add_white_noise_with_snr(sig, 10.0, rng)
    snr = 10 * np.log10(np.mean(sig ** 2) / np.mean((noisy - sig) ** 2))
    assert abs(snr - 10.0) < 0.1


# ------------------------------------------------------------------------------------------------------------------
# Independent implementations that ARE installed here (the reference itself is not importable): sklearn's own
# StandardScaler (the class the reference calls, train_constraints.py:28-35), torch.stft and torch autograd.
# They pin the restatement's arithmetic, not the reference's pinned versions: DESIGN.md section 4.
def test_scaler_oracle_equals_sklearn():
    from sklearn.preprocessing import StandardScaler

    rng = np.random.default_rng(0)
    x = rng.standard_normal((500, 37)) * rng.uniform(0.01, 40, 37) + rng.uniform(-100, 100, 37)
    x[:, 5] = 7.5           # constant column: sklearn keeps scale 1
    x[:, 11] = 3.0 + 4.4e-16 * rng.integers(0, 2, 500)  # numerically constant: below the two-pass error bound
    sk = StandardScaler().fit(x)
    mean, scale = P.standard_scaler_fit(x)
    np.testing.assert_allclose(mean, sk.mean_, rtol=0, atol=1e-12)
    assert sk.scale_[5] == 1.0 and sk.scale_[11] == 1.0
    np.testing.assert_allclose(scale, sk.scale_, rtol=1e-12, atol=0)
    np.testing.assert_allclose((x - mean) / scale, sk.transform(x), atol=1e-10)


@pytest.mark.parametrize("n_fft,hop", [(2048, 512), (441, 220)])
def test_stft_oracle_equals_torch_stft(n_fft, hop):
    import torch

    rng = np.random.default_rng(n_fft)
    y = (0.3 * rng.standard_normal(22050)).astype(np.float64)
    S = M.power_spectrogram(y, np.float64, n_fft, hop)
    win = torch.hann_window(n_fft, periodic=True, dtype=torch.float64)
    Z = torch.stft(torch.as_tensor(y), n_fft=n_fft, hop_length=hop, win_length=n_fft, window=win, center=True,
                   pad_mode="reflect", return_complex=True)
    ref = (Z.abs() ** 2).numpy()
    assert S.shape == ref.shape == (1 + n_fft // 2, 1 + 22050 // hop)
    np.testing.assert_allclose(S, ref, rtol=1e-9, atol=1e-12 * ref.max())


def _torch_forward(spec, p, x, training, masks=None):
    """The Keras semantics of A3 written with torch ops (double), for autograd to differentiate."""
    import torch

    h = x
    stats = []
    for l, s in enumerate(spec):
        z = h @ p["W"][l] + p["b"][l]
        if l == len(spec) - 1:
            return z, stats
        a = torch.relu(z)
        if s.bn:
            if training:
                mu, var = a.mean(dim=0), a.var(dim=0, unbiased=False)
            else:
                mu, var = p["mm"][l], p["mv"][l]
            stats.append((mu, var))
            a = (a - mu) / torch.sqrt(var + P.BN_EPS) * p["g"][l] + p["be"][l]
        if masks is not None and masks[l] is not None:
            a = a * masks[l]
        h = a


@pytest.mark.parametrize("training", [True, False])
def test_mlp_oracle_gradients_equal_torch_autograd(training):
    import torch

    spec = [P.LayerSpec(13, 11, True, 0.25, True), P.LayerSpec(11, 7, True, 0.0, True), P.LayerSpec(7, 5, False, 0.0, True)]
    p = P.init_params(spec, seed=4, dtype=np.float64)
    rng = np.random.default_rng(9)
    for l in range(2):
        p.gamma[l] = 1 + 0.3 * rng.standard_normal(spec[l].n_out)
        p.beta[l] = 0.2 * rng.standard_normal(spec[l].n_out)
        p.mov_mean[l] = 0.1 * rng.standard_normal(spec[l].n_out)
        p.mov_var[l] = rng.uniform(0.5, 2.0, spec[l].n_out)
    x = rng.standard_normal((17, 13))
    y = P.to_categorical(rng.integers(0, 5, 17), 5).astype(np.float64)
    masks = [((rng.uniform(size=(17, 11)) > 0.25) / 0.75) if training else None, None, None]
    T = lambda a: None if a is None else torch.tensor(a, dtype=torch.float64, requires_grad=True)
    tp = {"W": [T(w) for w in p.W], "b": [T(b) for b in p.b], "g": [T(g) for g in p.gamma], "be": [T(b) for b in p.beta],
          "mm": [None if a is None else torch.tensor(a) for a in p.mov_mean], "mv": [None if a is None else torch.tensor(a) for a in p.mov_var]}
    xt = torch.tensor(x, requires_grad=True)
    tm = None if not training else [None if k is None else torch.tensor(k) for k in masks]
    logits, _ = _torch_forward(spec, tp, xt, training, tm)
    loss = -(torch.tensor(y) * torch.log_softmax(logits, dim=1)).sum(dim=1).mean()
    loss.backward()
    got = P.forward_backward(spec, p, x, y, masks=masks if training else None, training=training, need_dx=True)
    assert abs(got["loss"] - float(loss.detach())) < 1e-12
    np.testing.assert_allclose(got["logits"], logits.detach().numpy(), atol=1e-12)
    for l in range(3):
        np.testing.assert_allclose(got["dW"][l], tp["W"][l].grad.numpy(), atol=1e-12)
        np.testing.assert_allclose(got["db"][l], tp["b"][l].grad.numpy(), atol=1e-12)
        if spec[l].bn:
            np.testing.assert_allclose(got["dgamma"][l], tp["g"][l].grad.numpy(), atol=1e-12)
            np.testing.assert_allclose(got["dbeta"][l], tp["be"][l].grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(got["dx"], xt.grad.numpy(), atol=1e-12)
    if not training:
        # the vector-Jacobian product behind class_gradient / JSMA / C&W, at the softmax outputs and at the logits
        v = rng.standard_normal((17, 5))
        for on_logits in (False, True):
            xt2 = torch.tensor(x, requires_grad=True)
            lg, _ = _torch_forward(spec, tp, xt2, False)
            out = lg if on_logits else torch.softmax(lg, dim=1)
            (out * torch.tensor(v)).sum().backward()
            dx, prob = P.output_vjp_infer(spec, p, x, v, on_logits=on_logits)
            np.testing.assert_allclose(dx, xt2.grad.numpy(), atol=1e-12)
            np.testing.assert_allclose(prob, torch.softmax(lg, dim=1).detach().numpy(), atol=1e-14)


def test_adam_oracle_equals_torch_adam_up_to_epsilon_placement():
    """Keras-form Adam differs from torch.optim.Adam only in where epsilon sits; with eps -> 0 both agree."""
    import torch

    rng = np.random.default_rng(3)
    w = rng.standard_normal((10, 5))
    m, v = np.zeros_like(w), np.zeros_like(w)
    wt = torch.tensor(w.copy(), requires_grad=True)
    opt = torch.optim.Adam([wt], lr=1e-3, betas=(0.9, 0.999), eps=1e-30)
    for t in range(1, 6):
        g = rng.standard_normal((10, 5))
        P.adam_update(w, g, m, v, t, eps=1e-30)
        wt.grad = torch.tensor(g)
        opt.step()
    np.testing.assert_allclose(w, wt.detach().numpy(), rtol=1e-10, atol=1e-13)


# ------------------------------------------------------------------------------------------------
# The feature stages after the resampler against transformers.audio_utils -- an independent implementation of
# librosa's mel filter bank (norm="slaney", mel_scale="slaney"), centred reflect-padded power spectrogram and
# power_to_db that ships in this image (the Whisper feature extractor's code path), with scipy.fftpack for the DCT:
# together they ARE librosa.feature.mfcc(y, sr=22050) step by step (extract_features_construct_dataset.py:31).
# ------------------------------------------------------------------------------------------------
def _audio_utils():
    try:
        from transformers import audio_utils
    except Exception as e:  # the product does not need it; only this pin does
        pytest.skip(f"transformers.audio_utils is not importable: {e}")
    for name in ("mel_filter_bank", "window_function", "spectrogram", "power_to_db"):
        if not hasattr(audio_utils, name):
            pytest.skip(f"transformers.audio_utils has no {name}")
    return audio_utils


@pytest.mark.parametrize("n_fft", [2048, 441])
def test_mel_bank_and_window_equal_transformers(n_fft):
    AU = _audio_utils()
    fb = AU.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=128, min_frequency=0.0, max_frequency=11025.0,
                            sampling_rate=22050, norm="slaney", mel_scale="slaney")
    mine = M.mel_filterbank(n_fft=n_fft)
    assert mine.shape == fb.T.shape
    assert np.abs(fb.T - mine).max() <= 1e-8  # float32 rounding of a bank whose largest weight is 0.04
    np.testing.assert_array_equal((fb.T != 0), (mine != 0))  # the same two-filters-per-bin sparsity the kernels rely on
    w = AU.window_function(n_fft, "hann", periodic=True)
    assert np.abs(w - M.hann_periodic(n_fft)).max() <= 1e-15


def test_power_to_db_equals_transformers():
    AU = _audio_utils()
    rng = np.random.default_rng(0)
    S = np.abs(rng.standard_normal((128, 44))) ** 2 * 10.0 ** rng.uniform(-14, 3, (128, 44))
    S[3, 5] = 0.0  # below amin
    np.testing.assert_allclose(M.power_to_db(S), AU.power_to_db(S, reference=1.0, min_value=1e-10, db_range=80.0), rtol=0, atol=1e-12)


@pytest.mark.parametrize("n_fft,hop,n", [(2048, 512, 22050), (2048, 512, 9000), (441, 220, 22050)])
def test_mfcc_after_resampling_equals_transformers_plus_scipy(n_fft, hop, n):
    """mfcc_22k (STFT -> power -> 128 Slaney mels -> dB with top_db 80 -> DCT-II ortho, 20 coefficients) for the
    voice-digit window (2048/512) and the Speaker-recognition one (441/220)."""
    import scipy.fftpack

    AU = _audio_utils()
    rng = np.random.default_rng(n_fft + n)
    t = np.arange(n) / 22050.0
    y = (0.3 * np.sin(2 * np.pi * 440 * t * (1 + 0.3 * t)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    fb = AU.mel_filter_bank(num_frequency_bins=1 + n_fft // 2, num_mel_filters=128, min_frequency=0.0, max_frequency=11025.0,
                            sampling_rate=22050, norm="slaney", mel_scale="slaney")
    db = AU.spectrogram(y.astype(np.float64), window=AU.window_function(n_fft, "hann", periodic=True), frame_length=n_fft,
                        hop_length=hop, fft_length=n_fft, power=2.0, center=True, pad_mode="reflect", mel_filters=fb,
                        log_mel="dB", reference=1.0, min_value=1e-10, db_range=80.0)
    want = scipy.fftpack.dct(db, axis=0, type=2, norm="ortho")[:20]
    got = M.mfcc_22k(y, np.float32, n_fft, hop)
    assert got.shape == want.shape
    # the oracle runs the reference's float32 arithmetic, the peer float64: same budget as test_oracle_rounding_budget
    assert np.abs(got - want).max() <= 2e-3
    got64 = M.mfcc_22k(y.astype(np.float64), np.float64, n_fft, hop)
    assert np.abs(got64 - want).max() <= 5e-5  # float64 inside, returned as float32 (values up to ~170: ulp 1.5e-5)


# The resampler against SciPy's polyphase FIR machinery.  resampy's kaiser_best interpolator evaluates, for every output
# sample at input-time tau, sum_n x[n] w(tau - n) with w = the linearly interpolated half window.  For the rational ratio
# L/M every tau - n is a multiple of 1/L, so the whole resampler IS scipy.signal.upfirdn(g, x, up=L, down=M) with the
# prototype g[j] = w(|j - C| / L): the filter here comes from np.interp on the window table (not from the oracle's tap
# gather code), and the filtering from SciPy's compiled upfirdn (not from the oracle's index arithmetic).
def _kaiser_best_prototype(L):
    win, num_table = M.kaiser_best_half_window()
    C = M.KB_NUM_ZEROS * L
    j = np.arange(-C + 1, C)                                  # |d| < 64 zero crossings
    pos = np.abs(j) / L * num_table                           # position in the 512-per-crossing window table
    g = np.interp(pos, np.arange(len(win)), win)
    # resampy's tap count i_max = (len(win) - int(frac * 512)) // 512 keeps the 64th tap of a wing only while the
    # table offset is 0 or 1: the support ends at table position 63 * 512 + 2 (the dropped taps are ~3e-8)
    g[pos >= (M.KB_NUM_ZEROS - 1) * num_table + 2] = 0.0
    return g, C - 1                                           # taps, index of the centre tap


@pytest.mark.parametrize("sr_in,n", [(16000, 16000), (16000, 5003), (8000, 8000), (11025, 3000)])
def test_resampler_equals_scipy_upfirdn(sr_in, n):
    g_ = int(np.gcd(sr_in, 22050))
    L, Mdown = 22050 // g_, sr_in // g_
    x = np.random.default_rng(n).standard_normal(n)
    g, centre = _kaiser_best_prototype(L)
    # upfirdn returns v[Mdown * i] of v = g * upsample(x); y[t] = v[Mdown * t + centre]: shift the centre onto the grid
    lead = (-centre) % Mdown
    out = scipy.signal.upfirdn(np.concatenate([np.zeros(lead), g]), x, up=L, down=Mdown)
    skip = (centre + lead) // Mdown
    n_out = int(n * (22050.0 / sr_in))
    ref = out[skip:skip + n_out]
    for mode in ("accumulate", "multiply"):
        got = M.resample_kaiser_best(x, sr_in, 22050, time_mode=mode).astype(np.float64)
        # float32 rounding of the oracle's output is all that separates them
        assert got.shape == ref.shape and np.abs(got - ref).max() < 2e-7 * max(1.0, np.abs(ref).max()), mode
    fast = M.resample_kaiser_best_fast(x, sr_in, 22050).astype(np.float64)
    assert np.abs(fast - ref).max() < 2e-7 * max(1.0, np.abs(ref).max())


def test_polyphase_table_equals_prototype_slices():
    """The per-phase taps (what the GPU resampler contracts with) are strided slices of the same prototype."""
    h, n_off, L, Mdown, wing = M._polyphase_table(16000, 22050)
    g, centre = _kaiser_best_prototype(L)
    gz = np.concatenate([np.zeros(L), g, np.zeros(L)])   # taps just outside the 64-crossing support are zero
    for p in (0, 1, 17, 220, 440):
        # output L q + p sits at input time M q + p M / L; tap k multiplies x[M q + n_off[p] - (wing - 1) + k]
        k = np.arange(2 * wing)
        j = p * Mdown - (n_off[p] - (wing - 1) + k) * L     # (tau - n) * L
        np.testing.assert_allclose(h[p], gz[L + centre + j], rtol=0, atol=1e-15)
