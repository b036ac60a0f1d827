"""GPU parity for the Speaker-recognition variant: 441/220 short-window MFCC on the DFT-contraction
path, the 1-s window slicing, and the 2020 -> 20 classifier with the product-norm constraint at R = 20 classes.

MFCC tolerance as in test_mfcc_gpu.py (|delta| <= 2e-2 absolute in dB-cepstral units); the oracle evaluates these
windows in float64 like the reference does (``y=np.array(window, dtype=float)``).
"""
import wave

import numpy as np
import pytest
import torch

from golden import inputs
from helpers import build_model, dev, grads_of, load_params, read_params, rel_err
from oracle import constraints_ref as R
from oracle import mfcc_ref as M
from oracle import mlp_ref as P

pytestmark = pytest.mark.gpu
ATOL = 2e-2


def _windows(n=6, seed=3):
    rng = np.random.default_rng(seed)
    clips = inputs.test_clips()
    w = [M.librosa_load_resample(c, 16000) for c in clips]                       # tone / chirp / noise / mixture at 22 050 Hz
    w.append((0.05 * rng.standard_normal(22050)).astype(np.float32))              # white noise
    t = np.arange(22050) / 22050.0
    w.append((0.4 * np.sin(2 * np.pi * 3000 * t) * (t > 0.5)).astype(np.float32))  # half a second of silence, then a tone
    return np.stack(w[:n]).astype(np.float32)


def test_sr_mfcc_windows_match_oracle(cuda):
    from lipasr.speaker_recognition import N_FEATURES, mfcc_windows

    w = _windows()
    got = mfcc_windows(w).cpu().numpy()
    assert got.shape == (len(w), N_FEATURES) == (6, 2020)
    ref = M.sr_mfcc_windows(w)
    assert np.abs(got - ref).max() < ATOL, np.abs(got - ref).max()
    # coefficient-major [20][101]; a second call re-uses the plan and reproduces bit for bit
    one = M.mfcc_22k(w[2].astype(np.float64), np.float64, 441, 220)
    assert one.shape == (20, 101) and np.abs(got[2].reshape(20, 101) - one).max() < ATOL
    assert torch.equal(mfcc_windows(w), torch.as_tensor(got).cuda())


@pytest.mark.parametrize("n_fft,hop,n_samp", [(441, 220, 11025), (400, 160, 16000), (510, 510, 8000), (32, 7, 3000), (256, 64, 22050)])
def test_short_window_path_other_shapes(cuda, n_fft, hop, n_samp):
    """Every legal (n_fft, hop) of lipasr_mfcc_plan_ex's DFT path, ragged row counts and partial last workgroups."""
    from lipasr.speaker_recognition import WindowMfcc

    rng = np.random.default_rng(n_fft + hop)
    b = 5
    t = np.arange(n_samp) / 22050.0
    y = np.stack([0.3 * np.sin(2 * np.pi * (200.0 + 700.0 * i) * t) + 0.02 * rng.standard_normal(n_samp) for i in range(b)]).astype(np.float32)
    ex = WindowMfcc(batch_max=8, n_samp=n_samp, n_fft=n_fft, hop_length=hop)
    assert ex.n_frames == 1 + n_samp // hop
    got = ex(dev(y)).cpu().numpy().reshape(b, 20, ex.n_frames)
    for i in range(b):
        ref = M.mfcc_22k(y[i], np.float32, n_fft, hop)
        assert np.abs(got[i] - ref).max() < ATOL, (i, np.abs(got[i] - ref).max())


def test_plan_ex_argument_checking(cuda):
    import lipasr._native as N

    h = N.get_handle(0)
    assert N.lib.lipasr_mfcc_plan_ex(h.h, 22050, 22050, 4, 1024, 256) == N.EUNSUPPORTED
    assert N.lib.lipasr_mfcc_plan_ex(h.h, 22050, 22050, 4, 441, 442) == N.EUNSUPPORTED
    assert N.lib.lipasr_mfcc_plan_ex(h.h, 22050, 100, 4, 441, 220) == N.EINVAL  # shorter than the reflect padding
    assert N.lib.lipasr_mfcc_plan_ex(h.h, 22050, 22050, 4, 2048, 512) == N.OK  # == lipasr_mfcc_plan


def _write_wav(path, x, sr):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767.0).astype("<i2").tobytes())


def test_load_audio_dataset_and_labels(cuda, tmp_path):
    """SR/extract_features_construct_dataset.py:203-233 end to end on two recordings (22 050 Hz and 16 kHz)."""
    from lipasr.extract_features_construct_dataset import read_wav
    from lipasr.speaker_recognition import load_audio_dataset_and_labels

    rng = np.random.default_rng(8)
    t1 = np.arange(int(22050 * 5.3)) / 22050.0
    x1 = 0.3 * np.sin(2 * np.pi * 440 * t1 * (1 + 0.1 * t1)) + 0.01 * rng.standard_normal(len(t1))
    t2 = np.arange(16000 * 4) / 16000.0
    x2 = 0.2 * np.sin(2 * np.pi * 900 * t2) + 0.01 * rng.standard_normal(len(t2))
    _write_wav(tmp_path / "a.wav", x1, 22050)
    _write_wav(tmp_path / "b.wav", x2, 16000)
    feats, labels = load_audio_dataset_and_labels([tmp_path / "a.wav", tmp_path / "b.wav"], [3, 17])
    d1, _ = read_wav(tmp_path / "a.wav")
    d2, _ = read_wav(tmp_path / "b.wav")
    w1 = M.sr_split_windows(d1)
    w2 = M.sr_split_windows(M.librosa_load_resample(d2, 16000))
    assert len(w1) == 3 and len(w2) == 2
    ref = M.sr_mfcc_windows(np.concatenate([w1, w2]))
    assert feats.shape == (5, 2020) and feats.dtype == np.float64
    np.testing.assert_array_equal(labels, [3, 3, 3, 17, 17])
    assert np.abs(feats - ref).max() < ATOL


def _state(spec, seed):
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(seed + 100)
    for l, s in enumerate(spec):
        p.b[l] = (0.05 * rng.standard_normal(s.n_out)).astype(np.float32)
        if s.bn:
            p.gamma[l] = (1 + 0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.beta[l] = (0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_mean[l] = (0.5 + 0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    return p


@pytest.mark.parametrize("which", ["constrained", "unconstrained"])
def test_sr_model_step_and_constraint(cuda, which):
    """One batch of 64 (SR/train_constraints.py:41): gradients, logits, then simple_norm_constraint(rho=1) at R = 20."""
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.speaker_recognition import get_model, get_model_unconstrained

    spec = P.sr_constrained_spec() if which == "constrained" else P.sr_unconstrained_spec()
    # the product module builds the same structure as the oracle spec
    mine = (get_model if which == "constrained" else get_model_unconstrained)(max_batch=64)
    dense = [l for l in mine.layers if "dense" in l.name]
    assert [tuple(l.get_weights()[0].shape) for l in dense] == [(s.n_in, s.n_out) for s in spec]
    assert sum("batch" in l.name for l in mine.layers) == sum(s.bn for s in spec[:-1])

    p = _state(spec, 5)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    rng = np.random.default_rng(6)
    x = rng.standard_normal((64, 2020)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 20, 64), 20)
    masks = [((rng.uniform(size=(64, s.n_out)) > s.dropout) / (1 - s.dropout)).astype(np.float32) if s.dropout > 0 else None for s in spec]
    # inference first: the training pass below moves the BatchNorm moving statistics
    logits = m.predict_device(dev(x), logits=True).cpu().numpy()
    ref_logits = P.forward_infer(spec, p.astype(np.float64), x.astype(np.float64), return_logits=True)
    rel = np.abs(logits - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
    assert rel.max() <= 1e-3
    np.testing.assert_array_equal(logits.argmax(1), ref_logits.argmax(1))
    m.train_fwd_bwd(dev(x), dev(y), masks=[dev(k) if k is not None else None for k in masks])
    ref = P.forward_backward(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64), masks=masks, training=True)
    got = grads_of(m, spec)
    for l in range(6):
        assert rel_err(got["dW"][l], ref["dW"][l]) < 1e-4, l
        assert rel_err(got["db"][l], ref["db"][l]) < 1e-4, l
    cb = simple_norm_constraint(rho=1, affected_layers_indices=[])  # SR/train_constraints.py:103
    cb.set_model(m)
    cb.on_batch_end(0)
    want, norms = R.simple_norm_constraint_pass([w.astype(np.float64) for w in p.W], 1.0, [])
    after = read_params(m, spec)
    for l in range(6):
        assert rel_err(after.W[l], want[l]) < 2e-5, l
    assert abs(R.sigma_max(R.product_chain(after.W)) - norms[-1]) < 1e-4 * max(1.0, norms[-1])


def test_sr_pipeline_step_matches_oracle(cuda):
    """The end-to-end step with the 441/220 extractor plugged into TrainPipeline: windows -> MFCC (2020) -> fwd/bwd ->
    Adam + NonNeg -> simple_norm_constraint(rho = 1), one batch of 64, dropout off, against the oracle."""
    from lipasr.pipeline import TrainPipeline
    from lipasr.speaker_recognition import WindowMfcc

    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.sr_constrained_spec()]
    p = _state(spec, 11)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    w = np.concatenate([_windows(), _windows(seed=4) * 0.5])[:8]
    w = np.tile(w, (8, 1)).astype(np.float32)  # 64 windows
    w += (0.01 * np.random.default_rng(0).standard_normal(w.shape)).astype(np.float32)
    y = P.to_categorical(np.arange(64) % 20, 20)
    raw = M.sr_mfcc_windows(w)
    mean, scale = P.standard_scaler_fit(raw)  # standardised features, as the reference trains on (train_constraints.py:28-35)
    affine = (torch.as_tensor(mean).cuda(), torch.as_tensor(scale).cuda())
    pipe = TrainPipeline(m, batch=64, utterance_length=101, rho=1.0, constraint="product", extractor=WindowMfcc(batch_max=64), use_graph=True,
                         affine=affine)
    pipe.step(dev(w), dev(y))
    pipe.synchronize()
    feats = pipe.feats.cpu().numpy().astype(np.float64)
    assert np.abs(feats - (raw - mean) / scale).max() < ATOL / scale.min() + 1e-4
    p64, st = p.astype(np.float64), P.AdamState()
    P.train_step(spec, p64, st, feats, y.astype(np.float64))   # from the GPU's features: isolates the classifier step
    want, norms = R.simple_norm_constraint_pass([wk.astype(np.float32) for wk in p64.W], 1.0, [])
    after = read_params(m, spec)
    np.testing.assert_allclose(pipe.norms.cpu().numpy(), norms, rtol=2e-3)
    for l in range(6):
        # Adam's first step moves every weight by ~lr whatever the gradient's size: a sign flip of a ~1e-7 gradient
        # is a 2e-3 move, so the bulk is compared tightly and the tail loosely (as test_pipeline_gpu.py does)
        d = np.abs(after.W[l] - want[l]) / np.abs(want[l]).max()
        assert np.quantile(d, 0.999) < 2e-3 and d.max() < 5e-2, (l, np.quantile(d, 0.999), d.max())
