"""Edge cases on the GPU path: the smallest and the raggedest inputs each entry point accepts, against the oracle."""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, grads_of, load_params, rel_err
from oracle import attacks_ref as A
from oracle import mfcc_ref as M
from oracle import mlp_ref as P

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_samp", [2, 37, 512, 1487, 16001])
def test_mfcc_very_short_and_odd_clips(cuda, n_samp):
    """Clips shorter than the 1024-sample reflect padding (np.pad reflects repeatedly), odd lengths (unaligned rows:
    the float4 paths must not be taken) and one sample over a second."""
    from lipasr.extract_features_construct_dataset import mfcc

    rng = np.random.default_rng(n_samp)
    x = (0.2 * rng.standard_normal((3, n_samp))).astype(np.float32)
    got = mfcc(x).cpu().numpy()
    ref = M.compute_mfcc_batch(x)
    assert got.shape == ref.shape == (3, 880)
    assert np.abs(got - ref).max() < 2e-2, np.abs(got - ref).max()


def test_mfcc_single_clip_and_full_plan(cuda):
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.synth import synth_clips

    waves, _ = synth_clips(33, seed=5)
    ex = MfccExtractor(16000, 16000, 33)
    full = ex(dev(waves)).cpu().numpy()                 # batch == batch_max, one clip past a 32-clip tile
    one = ex(dev(waves[32:33])).cpu().numpy()           # batch == 1
    assert np.array_equal(full[32:33], one)             # a clip's features do not depend on its neighbours
    assert np.abs(full - M.compute_mfcc_batch(waves)).max() < 2e-2
    with pytest.raises(ValueError):
        ex(dev(np.concatenate([waves, waves[:1]])))     # one more than the plan holds


@pytest.mark.parametrize("batch", [1, 2, 31, 33])
def test_classifier_tiny_and_ragged_batches(cuda, batch):
    """batch 1 makes every BatchNorm variance exactly 0 (rstd = 1/sqrt(eps)); 31 / 33 straddle the 32-row tiles."""
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=2, dtype=np.float32, nonneg_init=True)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    rng = np.random.default_rng(batch)
    x = rng.standard_normal((batch, 880)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 10, batch), 10)
    ref_logits = P.forward_infer(spec, p.astype(np.float64), x.astype(np.float64), return_logits=True)
    got_logits = m.predict_device(dev(x), logits=True).cpu().numpy()
    assert np.abs(got_logits - ref_logits).max() <= 1e-3 * max(np.abs(ref_logits).max(), 1e-6)
    m.train_fwd_bwd(dev(x), dev(y), dropout=False)
    ref = P.forward_backward(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64), masks=None, training=True)
    got = grads_of(m, spec)
    assert abs(float(m._loss_rows[:batch].mean()) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
    for l in range(6):
        # with batch 1 the BatchNorm backward cancels to exactly 0 in exact arithmetic; compare on an absolute scale
        scale = max(np.abs(ref["dW"][l]).max(), 1e-4)
        assert np.abs(got["dW"][l] - ref["dW"][l]).max() <= 2e-3 * scale, l


def test_attacks_on_ragged_batches(cuda):
    from lipasr.attacks import FastGradientMethod, ProjectedGradientDescent, TensorFlowV2Classifier

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=6, dtype=np.float32)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    x = np.random.default_rng(1).standard_normal((45, 880)).astype(np.float32)   # 32 + 13
    adv = FastGradientMethod(estimator=clf, eps=0.2).generate(x=x)
    assert np.allclose(np.abs(adv - x).max(), 0.2, atol=1e-6)
    p64 = p.astype(np.float64)
    want = A.fgsm(spec, p64, x.astype(np.float64), 0.2)
    assert np.mean(np.sign(adv - x) == np.sign(want - x)) > 0.999
    one = ProjectedGradientDescent(estimator=clf, eps=0.3, max_iter=3).generate(x=x[:1])   # a single sample
    assert one.shape == (1, 880) and np.abs(one - x[:1]).max() <= 0.3 + 1e-6


def test_projection_on_degenerate_stacks(cuda):
    """One layer, one class, and an all-zero kernel (product norm 0: the reference divides by eps = 2.2e-16)."""
    from lipasr.Constraints import simple_norm_constraint
    from oracle import constraints_ref as R
    from test_spectral_gpu import FakeModel

    for ws in ([np.abs(np.random.default_rng(0).standard_normal((7, 3))).astype(np.float32)],
               [np.abs(np.random.default_rng(1).standard_normal((9, 5))).astype(np.float32), np.abs(np.random.default_rng(2).standard_normal((5, 1))).astype(np.float32)]):
        model = FakeModel([w.copy() for w in ws])
        cb = simple_norm_constraint(rho=0.7, affected_layers_indices=[])
        cb.set_model(model)
        cb.on_batch_end(0)
        want, _ = R.simple_norm_constraint_pass(ws, 0.7, [])
        for l, r in zip([l for l in model.layers if "dense" in l.name], want):
            assert rel_err(l.w, r) < 2e-5


@pytest.mark.parametrize("L", [1, 45, 100, 200])
def test_mfcc_long_utterance_lengths_with_affine(cuda, L):
    """utterance_length above the 44 frames a 1-s clip has (zero padding, several DCT frame chunks) with the fused
    (x - mean) / scale, which also applies to the padded zeros (StandardScaler sees them as features)."""
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.synth import synth_clips

    waves, _ = synth_clips(5, seed=9)
    rng = np.random.default_rng(L)
    mean = rng.standard_normal(20 * L) * 10
    scale = rng.uniform(0.5, 20, 20 * L)
    ex = MfccExtractor(16000, 16000, 8)
    got = ex(dev(waves), L, torch.as_tensor(mean).cuda(), torch.as_tensor(scale).cuda()).cpu().numpy()
    ref = (M.compute_mfcc_batch(waves, utterance_length=L) - mean) / scale
    assert got.shape == (5, 20 * L)
    assert np.abs(got - ref).max() < 2e-2 / scale.min() + 1e-4


def test_gemm_with_padded_leading_dimensions(cuda):
    """lda / ldb / ldc larger than the logical widths (sub-matrices of bigger buffers), odd sizes."""
    import lipasr._native as N

    h = N.get_handle(0)
    rng = np.random.default_rng(0)
    M_, Nn, K = 45, 19, 77
    A_full = rng.integers(-3, 4, (M_, K + 5)).astype(np.float32)
    B_full = rng.integers(-3, 4, (K, Nn + 3)).astype(np.float32)
    C_full = np.full((M_, Nn + 7), -1.0, np.float32)
    At, Bt, Ct = dev(A_full), dev(B_full), dev(C_full)
    N.check(N.lib.lipasr_gemm_f32(h.h, 0, 0, M_, Nn, K, N.ptr(At), K + 5, N.ptr(Bt), Nn + 3, N.ptr(Ct), Nn + 7, N.stream_ptr()))
    out = Ct.cpu().numpy()
    np.testing.assert_array_equal(out[:, :Nn], A_full[:, :K].astype(np.float64) @ B_full[:, :Nn].astype(np.float64))
    assert np.all(out[:, Nn:] == -1.0)  # nothing written past the logical width


def test_noise_models_on_odd_lengths(cuda):
    from lipasr.attacks import add_noise, add_white_noise, add_white_noise_with_snr

    x = (0.1 * np.random.default_rng(0).standard_normal(22051)).astype(np.float32)  # odd length
    for fn, args in ((add_white_noise, (0.05,)), (add_noise, (0.1, 0.02)), (add_white_noise_with_snr, (10.0,))):
        y = fn(x, *args)
        y = y.cpu().numpy() if torch.is_tensor(y) else np.asarray(y)
        assert y.shape == x.shape and np.isfinite(y).all() and not np.array_equal(y, x)
    snr = add_white_noise_with_snr(x, 10.0)
    snr = snr.cpu().numpy() if torch.is_tensor(snr) else np.asarray(snr)
    got_db = 10 * np.log10(np.mean(x.astype(np.float64) ** 2) / np.mean((snr - x).astype(np.float64) ** 2))
    assert abs(got_db - 10.0) < 0.3


def test_class_count_limit(cuda):
    """The plan covers 1..32 classes (the class dimension is one MFMA tile in the loss epilogue and the Gram solver);
    32 works end to end, 33 is refused at creation with a message that says so."""
    spec = [P.LayerSpec(50, 48, True, 0.0, True), P.LayerSpec(48, 32, False, 0.0, True)]
    p = P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 50)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 32, 37), 32)
    m.train_fwd_bwd(dev(x), dev(y), dropout=False)
    ref = P.forward_backward(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64), masks=None, training=True)
    got = grads_of(m, spec)
    for l in range(2):
        assert rel_err(got["dW"][l], ref["dW"][l]) < 1e-4
    assert abs(float(m._loss_rows[:37].mean()) - ref["loss"]) < 1e-4 * abs(ref["loss"])
    np.testing.assert_array_equal(m._correct_rows[:37].cpu().numpy(), (ref["prob"].argmax(1) == y.argmax(1)).astype(np.float32))
    with pytest.raises(ValueError, match="32"):
        build_model([P.LayerSpec(50, 48, True, 0.0, True), P.LayerSpec(48, 33, False, 0.0, True)], max_batch=64)
