"""GPU: BASELINE.json's parity statements checked END TO END, from the waveform (SURVEY 8d "Parity procedure").

    reference flow   wav -> librosa MFCC -> StandardScaler -> model.predict
                     (VD/attacks.py:339-345, VD/extract_features_construct_dataset.py:24-39, VD/train_constraints.py:107-111)
    oracle side      oracle.mfcc_ref (exact resampler) -> oracle standard_scaler -> oracle forward_infer, float64
    product side     lipasr MfccExtractor (K1) + fused StandardScaler affine -> Model.predict_device (K2), fp32

Nothing is shared between the two sides except the seeded waveforms and the seeded weights: each side extracts its own
features and fits its own scaler, so the MFCC stage's error budget is inside the logit comparison.
"""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, load_params, oracle_mfcc_parallel
from oracle import mlp_ref as P

pytestmark = pytest.mark.gpu

N_TEST = 2366  # the reference's test split (VD/processed_google_dataset/test_label.npy)


@pytest.fixture(scope="module")
def clips():
    from lipasr.synth import synth_clips

    waves, labels = synth_clips(N_TEST, seed=2366)
    return waves, labels, oracle_mfcc_parallel(waves)


def _state(spec, seed):
    """Seeded constrained-model weights with non-trivial biases and BatchNorm state (a model in mid-training)."""
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(seed + 100)
    for l, s in enumerate(spec):
        p.b[l] = (0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
        if s.bn:
            p.gamma[l] = (1 + 0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.beta[l] = (0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_mean[l] = (0.5 + 0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    return p


def test_logits_from_waveform_2366_clips(cuda, clips):
    """per-utterance logits within 1e-3 relative fp32, argmax labels identical -- waveform in, logits out."""
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor

    waves, _, ref_feats = clips
    spec = P.vd_constrained_spec()
    p = _state(spec, 7)
    # ---- oracle: MFCC (done in the fixture) -> standardise -> predict, float64
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_logits = P.forward_infer(spec, p.astype(np.float64), (ref_feats - mean) / scale, return_logits=True)
    # ---- product: K1 MFCC -> StandardScaler fitted on its own features -> fused affine -> K2 predict
    m = build_model(spec, max_batch=1024)
    load_params(m, p)
    ex = MfccExtractor(16000, 16000, 1024, cuda)
    wt = dev(waves)
    raw = torch.cat([ex(wt[s:s + 1024]) for s in range(0, N_TEST, 1024)])
    sc = StandardScaler().fit(raw)
    feats = torch.cat([ex(wt[s:s + 1024], 44, sc.mean_, sc.scale_) for s in range(0, N_TEST, 1024)])
    got = torch.cat([m.predict_device(feats[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()

    err_feat = np.abs(raw.cpu().numpy() - ref_feats).max()
    err_std = np.abs(feats.cpu().numpy() - (ref_feats - mean) / scale).max()
    rel = np.abs(got - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
    worst = int(rel.argmax())
    print(f"\nend-to-end: mfcc max|d|={err_feat:.2e}  standardised max|d|={err_std:.2e}  logits rel max={rel.max():.2e} (row {worst}), "
          f"median={np.median(rel):.2e}")
    assert err_feat < 2e-2
    assert rel.max() <= 1e-3, (worst, rel[worst], got[worst], ref_logits[worst])
    np.testing.assert_array_equal(got.argmax(1), ref_logits.argmax(1))
    # the softmax the reference's predict() returns
    np.testing.assert_allclose(torch.softmax(torch.as_tensor(got), 1).numpy(), P.softmax(ref_logits), atol=2e-5)


def test_logits_from_waveform_after_training(cuda, clips):
    """Same statement on a model the product trained itself for a few hundred constrained steps (weights that NonNeg,
    the projection and BatchNorm's moving statistics have shaped), read back through get_weights()."""
    from helpers import read_params
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    m = build_model(spec, max_batch=1024, seed=3)
    load_params(m, P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True))
    ex = MfccExtractor(16000, 16000, 1024, cuda)
    wt = dev(waves)
    raw = torch.cat([ex(wt[s:s + 1024]) for s in range(0, N_TEST, 1024)])
    sc = StandardScaler().fit(raw)
    feats = torch.cat([ex(wt[s:s + 1024], 44, sc.mean_, sc.scale_) for s in range(0, N_TEST, 1024)])
    ds = Dataset.from_tensor_slices((feats[:2048].cpu().numpy(), P.to_categorical(labels[:2048], 10))).batch(128)
    m.fit(ds, epochs=20, verbose=0, callbacks=[simple_norm_constraint(0.1, [])])
    p = read_params(m, spec)
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_logits = P.forward_infer(spec, p, (ref_feats - mean) / scale, return_logits=True)
    got = torch.cat([m.predict_device(feats[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()
    rel = np.abs(got - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
    print(f"\nend-to-end after 320 constrained steps: logits rel max={rel.max():.2e}, median={np.median(rel):.2e}")
    assert rel.max() <= 1e-3, rel.max()
    # argmax: identical wherever the oracle's own top-2 margin exceeds the 1e-3 budget (a tie inside the tolerance is not
    # a labelling error); on this model no row is that close
    top2 = np.sort(ref_logits, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 2e-3 * np.abs(ref_logits).max(axis=1)
    np.testing.assert_array_equal(got.argmax(1)[clear], ref_logits.argmax(1)[clear])
    assert clear.mean() > 0.99
