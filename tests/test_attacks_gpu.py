"""GPU parity: K4 sign step, FGSM / PGD through the ART-shaped classes, against oracle.attacks_ref,
plus the device noise models (statistical: the reference draws from NumPy's unseeded global RNG).

The attack iterates x <- x0 + clip(x + a*sign(g) - x0): a sign flip of one gradient component moves
that component by 2a, so parity is stated as (i) identical sign pattern on components whose float64
gradient magnitude is above fp32 noise and (ii) identical adversarial points wherever (i) holds.
"""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, load_params
from oracle import attacks_ref as A, mlp_ref as P

pytestmark = pytest.mark.gpu


def _setup(seed=4):
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=False)
    rng = np.random.default_rng(seed)
    for l, s in enumerate(spec):
        if s.bn:
            p.gamma[l] = (1 + 0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_mean[l] = (0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    m = build_model(spec)
    load_params(m, p)
    return spec, p, m


def test_sign_step_kernel(cuda):
    from lipasr.attacks import sign_step

    rng = np.random.default_rng(0)
    x0 = rng.standard_normal((37, 880)).astype(np.float32)
    xa = (x0 + rng.uniform(-0.3, 0.3, x0.shape)).astype(np.float32)
    g = rng.standard_normal(x0.shape).astype(np.float32)
    g[0, :5] = [0.0, -0.0, np.nan, np.inf, -np.inf]
    for alpha, eps in [(0.1, 0.25), (0.3, np.inf), (0.0, 0.1)]:
        t = dev(xa)
        sign_step(t, dev(x0), dev(g), alpha, eps)
        gz = np.where(np.isnan(g), 0.0, g)
        ref = A.sign_step(xa, x0, gz, np.float32(alpha), np.float32(eps)) if np.isfinite(eps) else xa + np.float32(alpha) * np.sign(gz)
        np.testing.assert_array_equal(t.cpu().numpy(), ref.astype(np.float32))


def test_input_gradient_matches_oracle(cuda):
    from lipasr.attacks import TensorFlowV2Classifier

    spec, p, m = _setup()
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,), loss_object=None)
    x = np.random.default_rng(1).standard_normal((96, 880)).astype(np.float32)
    y = P.to_categorical(np.arange(96) % 10, 10)
    got = clf.loss_gradient(x, y)
    ref = P.input_gradient_infer(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64))
    assert np.abs(got - ref).max() / np.abs(ref).max() < 5e-5
    np.testing.assert_allclose(clf.predict(x), P.forward_infer(spec, p.astype(np.float64), x.astype(np.float64)), atol=2e-5)


@pytest.mark.parametrize("eps", [0.01, 0.3, 1.0, 30.0])
def test_fgsm_reference_eps_grid(cuda, eps):
    from lipasr.attacks import FastGradientMethod, TensorFlowV2Classifier

    spec, p, m = _setup()
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    x = np.random.default_rng(2).standard_normal((70, 880)).astype(np.float32)
    atk = FastGradientMethod(estimator=clf, eps=eps)
    adv = atk.generate(x=x)
    assert adv is not x and adv.dtype == x.dtype and adv.shape == x.shape
    p64 = p.astype(np.float64)
    ylab = A._own_labels(spec, p64, x.astype(np.float64), 32)
    g = np.concatenate([P.input_gradient_infer(spec, p64, x[i:i + 32].astype(np.float64), ylab[i:i + 32]) for i in range(0, 70, 32)])
    ref = x + np.float32(eps) * np.sign(g).astype(np.float32)
    solid = np.abs(g) > 1e-4 * np.abs(g).max()
    assert solid.mean() > 0.99
    np.testing.assert_array_equal(adv[solid], ref[solid])
    assert np.abs(adv - x).max() <= eps * (1 + 1e-6) + 1e-6  # no component moves by more than eps


@pytest.mark.parametrize("max_iter,eps", [(20, 0.5), (100, 1.0)])
def test_pgd_matches_oracle(cuda, max_iter, eps):
    from lipasr.attacks import ProjectedGradientDescent, TensorFlowV2Classifier

    spec, p, m = _setup(seed=6)
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    x = np.random.default_rng(3).standard_normal((40, 880)).astype(np.float32)
    atk = ProjectedGradientDescent(estimator=clf, eps=eps, max_iter=max_iter)
    assert atk.eps_step == 0.1 and atk.batch_size == 32  # ART defaults the reference relies on
    adv = atk.generate(x=x)
    assert np.abs(adv - x).max() <= eps + 1e-5
    ref = A.pgd(spec, p.astype(np.float64), x.astype(np.float64), eps, 0.1, max_iter, 32)
    # trajectories coincide except where a near-zero gradient component flips sign in fp32
    agree = np.abs(adv - ref) < 1e-4
    assert agree.mean() > 0.97, agree.mean()
    # and the attack is as strong: loss at the adversarial points matches the oracle's
    p64 = p.astype(np.float64)
    y = A._own_labels(spec, p64, x.astype(np.float64), 32)
    def loss(z):
        return P.forward_backward(spec, p64, z.astype(np.float64), y, training=False)["loss"]
    assert abs(loss(adv) - loss(ref)) < 2e-2 * max(1.0, abs(loss(ref)))
    assert loss(adv) > loss(x)


def test_generate_leaves_input_untouched_and_handles_tensors(cuda):
    from lipasr.attacks import FastGradientMethod, TensorFlowV2Classifier

    spec, p, m = _setup()
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    xt = dev(np.random.default_rng(9).standard_normal((33, 880)))
    keep = xt.clone()
    adv = FastGradientMethod(estimator=clf, eps=0.2).generate(x=xt)
    assert torch.equal(xt, keep) and adv.is_cuda and adv.data_ptr() != xt.data_ptr()
    with pytest.raises(TypeError):
        FastGradientMethod(estimator=m, eps=0.1)
    with pytest.raises(ValueError):
        TensorFlowV2Classifier(model=m, nb_classes=11, input_shape=(880,))


def test_device_noise_statistics(cuda):
    from lipasr.attacks import add_noise, add_white_noise, add_white_noise_with_snr, noisy_audio_to_mfcc
    from lipasr.synth import synth_clips
    from oracle import mfcc_ref as M

    z = torch.zeros(8, 22050, device="cuda")
    n = add_white_noise(z, 0.05, seed=1)
    assert abs(float(n.std()) - 0.05) < 5e-4 and abs(float(n.mean())) < 5e-4 and torch.equal(z, torch.zeros_like(z))
    assert not torch.equal(n[0], n[1]) and torch.equal(n, add_white_noise(z, 0.05, seed=1))
    mix = add_noise(z, 0.01, 0.002, seed=2).double()
    frac = 0.0079787
    assert abs(float(mix.var()) - ((1 - frac) * 0.002 ** 2 + frac * 0.02 ** 2)) < 4e-7
    t = torch.arange(22050, device="cuda") * 0.01
    sig = torch.sin(t)[None, :].repeat(4, 1).contiguous()
    noisy = add_white_noise_with_snr(sig, 10.0, seed=3)
    snr = 10 * torch.log10((sig ** 2).mean(dim=1) / ((noisy - sig) ** 2).mean(dim=1))
    assert float((snr - 10.0).abs().max()) < 0.2
    # sigma = 0 path of black_box_attack_on_audio is the clean MFCC
    waves, _ = synth_clips(6, seed=8)
    clean = noisy_audio_to_mfcc(waves, 16000, sigma=0).cpu().numpy()
    assert np.abs(clean - M.compute_mfcc_batch(waves)).max() < 2e-2
    loud = noisy_audio_to_mfcc(waves, 16000, sigma=0.05, seed=4).cpu().numpy()
    assert np.abs(loud - clean).max() > 1.0


def test_file_level_black_box_helpers(cuda, tmp_path):
    """attacks.py:27-45, 89-121, 145-163, 248-274: the per-file forms of the audio attacks, the mixture generator and
    the dataset loader -- the single-file result is the corresponding row of the batched dataset call (same seed)."""
    import wave

    from lipasr.attacks import (black_box_attack_on_audio, black_box_attack_on_audio_dataset, black_box_attack_on_audio_dataset_snr,
                                black_box_attack_on_audio_snr, load_npy_dataset, mixtgauss)
    from lipasr.extract_features_construct_dataset import extract_features

    rng = np.random.default_rng(3)
    files = []
    for k in range(3):
        x = 0.3 * np.sin(2 * np.pi * (300 + 100 * k) * np.arange(16000) / 16000.0) + 0.01 * rng.standard_normal(16000)
        path = tmp_path / f"c{k}.wav"
        with wave.open(str(path), "wb") as f:
            f.setnchannels(1); f.setsampwidth(2); f.setframerate(16000)
            f.writeframes((np.clip(x, -1, 1) * 32767.0).astype("<i2").tobytes())
        files.append(str(path))
    clean = black_box_attack_on_audio(files[0], 44)
    assert clean.shape == (20, 44) and clean.dtype == np.float32
    # sigma = p = alpha = 0: no noise branch.  Not bit-equal: extract_features runs the fused resample -> STFT kernel, the
    # attack path resamples, (adds noise) and runs the STFT kernel on the 22 kHz signal; their FFT twiddles differ in the
    # last bit (table values vs products of table values): ~5e-5 in MFCC units, 400x inside the 2e-2 parity tolerance
    np.testing.assert_allclose(clean, extract_features(files[0], 44), rtol=0, atol=1e-3)
    ds = black_box_attack_on_audio_dataset(files, 0.02, 0, 0, seed=5)
    one = black_box_attack_on_audio(files[0], 44, sigma=0.02, seed=5)
    np.testing.assert_allclose(one.reshape(-1), ds[0], rtol=0, atol=1e-4)
    assert np.abs(one - clean).max() > 0.5
    mix = black_box_attack_on_audio(files[1], 30, p=0.05, alpha=0.01, seed=6)
    assert mix.shape == (20, 30) and np.abs(mix - extract_features(files[1], 30)).max() > 0.1
    snr_ds = black_box_attack_on_audio_dataset_snr(files, 15.0, seed=7)
    snr_one = black_box_attack_on_audio_snr(files[0], 44, 15.0, seed=7)
    np.testing.assert_allclose(snr_one.reshape(-1), snr_ds[0], rtol=0, atol=1e-4)
    g = mixtgauss(400000, 0.01, 0.002, 0.02, seed=8).astype(np.float64)
    frac = 0.0079787  # P(|N(0,1)| < 0.01)
    assert abs(g.var() - ((1 - frac) * 0.002 ** 2 + frac * 0.02 ** 2)) < 4e-7 and abs(g.mean()) < 2e-5
    with pytest.raises(NotImplementedError):
        mixtgauss(10, 0.01, 0.002, 0.5)
    d = tmp_path / "processed"
    d.mkdir()
    arrays = {n: rng.standard_normal((4, 3)) for n in ("train_data", "train_label", "dev_data", "dev_label", "test_data", "test_label")}
    for n, a in arrays.items():
        np.save(d / n, a)
    got = load_npy_dataset(str(d) + "/")
    for a, n in zip(got, ("train_data", "train_label", "dev_data", "dev_label", "test_data", "test_label")):
        np.testing.assert_array_equal(a, arrays[n])


def test_class_gradient_and_output_vjp(cuda):
    """ART class_gradient: gradients of the softmax outputs, and the logits variant of the same VJP."""
    from lipasr.attacks import TensorFlowV2Classifier

    spec, p, m = _setup(7)
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    rng = np.random.default_rng(3)
    x = rng.standard_normal((37, 880)).astype(np.float32)
    labels = rng.integers(0, 10, 37)
    p64 = p.astype(np.float64)
    want = A.class_gradient(spec, p64, x.astype(np.float64), labels)
    got = clf.class_gradient(x, label=labels)
    assert got.shape == (37, 1, 880)
    scale = np.abs(want).max()
    assert np.abs(got[:, 0] - want).max() <= 2e-5 * scale
    # one class for every sample, and the full Jacobian
    got3 = clf.class_gradient(x[:5], label=3)
    assert np.abs(got3[:, 0] - A.class_gradient(spec, p64, x[:5].astype(np.float64), 3)).max() <= 2e-5 * scale
    jac = clf.class_gradient(x[:5])
    assert jac.shape == (5, 10, 880)
    assert np.abs(jac[:, 3] - got3[:, 0]).max() == 0.0
    assert np.abs(jac.sum(axis=1)).max() <= 1e-5 * scale  # probabilities sum to one: their gradients cancel
    # arbitrary upstream vector at the logits
    v = rng.standard_normal((37, 10)).astype(np.float32)
    probs = torch.empty(37, 10, device="cuda")
    gl = clf.output_vjp_device(dev(x), dev(v), on_logits=True, probs_out=probs).cpu().numpy()
    want_l, want_p = P.output_vjp_infer(spec, p64, x.astype(np.float64), v.astype(np.float64), on_logits=True)
    assert np.abs(gl - want_l).max() <= 2e-5 * np.abs(want_l).max()
    np.testing.assert_allclose(probs.cpu().numpy(), want_p, atol=2e-6)


@pytest.mark.parametrize("theta,gamma,bs", [(10.0, 0.1, 1), (2.0, 0.05, 8), (-3.0, 0.1, 4)])
def test_jsma_matches_oracle(cuda, theta, gamma, bs):
    """SaliencyMapMethod (attacks.py:546-550: theta=10, gamma=0.1) against the restated ART loop, targets injected.

    The loop is discrete (pick two features, add theta, re-predict): it matches the oracle exactly unless two
    class-gradient entries tie to within fp32 noise, so samples are compared on the set of perturbed features."""
    from lipasr.attacks import SaliencyMapMethod, TensorFlowV2Classifier, random_targets

    spec, p, m = _setup(9)
    clf = TensorFlowV2Classifier(model=m, nb_classes=10, input_shape=(880,))
    rng = np.random.default_rng(5)
    x = rng.standard_normal((24, 880)).astype(np.float32)
    p64 = p.astype(np.float64)
    preds = P.forward_infer(spec, p64, x.astype(np.float64)).argmax(1)
    y = random_targets(preds, 10, np.random.RandomState(1))
    assert y.shape == (24, 10) and np.all(y.argmax(1) != preds)
    atk = SaliencyMapMethod(classifier=clf, theta=theta, gamma=gamma, batch_size=bs, max_iter=200)
    got = atk.generate(x=x, y=y)
    want = A.jsma(spec, p64, x, y.argmax(1), theta=theta, gamma=gamma, batch_size=bs, max_iter=200)
    assert got.shape == x.shape and got.dtype == x.dtype
    same = [np.array_equal(np.nonzero(got[i] != x[i])[0], np.nonzero(want[i] != x[i])[0]) for i in range(24)]
    assert np.mean(same) >= 0.9, np.mean(same)
    for i in np.nonzero(same)[0]:
        np.testing.assert_allclose(got[i], want[i], atol=1e-4)
    d = got - x
    assert np.all(np.abs(d / theta - np.round(d / theta)) < 1e-4)        # every change is a whole number of thetas
    assert np.all((d != 0).sum(axis=1) / 880.0 <= gamma + 2.0 / 880.0)   # the gamma budget, up to the last pair
    # y=None draws ART's random targets; the input array is left untouched
    x_copy = x.copy()
    adv = SaliencyMapMethod(classifier=clf, theta=theta, gamma=gamma, batch_size=8, max_iter=50).generate(x=x[:8], rng=np.random.RandomState(0))
    assert np.array_equal(x, x_copy) and adv.shape == (8, 880)


def _small_net(seed=21):
    """A small BN network whose decision margins are wide enough that the attacks have something to do."""
    spec = [P.LayerSpec(40, 32, True, 0.0, True), P.LayerSpec(32, 16, True, 0.0, True), P.LayerSpec(16, 5, False, 0.0, True)]
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=False)
    rng = np.random.default_rng(seed)
    for l, s in enumerate(spec):
        p.W[l] = (p.W[l] * 3.0).astype(np.float32)
        if s.bn:
            p.mov_mean[l] = (0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    return spec, p, m


def test_carlini_l2_matches_restated_art(cuda):
    """CarliniL2Method (attacks.py:606-616).  The attack is a sequence of discrete decisions (line search, binary
    search) on float comparisons: parity is stated on what it returns -- which samples are moved, how far, and that
    moved samples change their label -- plus the reference's own regime (confidence >= 1 on softmax outputs), where
    ART's success test cannot hold and generate() returns its input."""
    from lipasr.attacks import CarliniL2Method, TensorFlowV2Classifier

    spec, p, m = _small_net()
    clf = TensorFlowV2Classifier(model=m, nb_classes=5, input_shape=(40,))
    x = np.random.default_rng(2).standard_normal((12, 40)).astype(np.float32)
    p64 = p.astype(np.float64)
    want = A.carlini_l2(spec, p64, x, confidence=0.0, batch_size=4)
    got = CarliniL2Method(classifier=clf, confidence=0.0, batch_size=4).generate(x=x)
    assert got.shape == x.shape and got.dtype == x.dtype
    d_got, d_want = np.linalg.norm(got - x, axis=1), np.linalg.norm(want - x, axis=1)
    moved_got, moved_want = d_got > 0, d_want > 0
    assert moved_want.sum() >= 4, "the fixture should contain samples the attack can move"
    assert (moved_got == moved_want).mean() >= 0.9
    both = moved_got & moved_want
    np.testing.assert_allclose(d_got[both], d_want[both], rtol=0.15)
    lab = np.eye(5)[P.forward_infer(spec, p64, x.astype(np.float64)).argmax(1)]
    margin = A._cw_margin(P.forward_infer(spec, p64, got.astype(np.float64)), lab, 0.0)
    # returned points satisfy ART's success test z_label - z_other <= 0 (minimal-distortion points sit ON the
    # decision boundary, so the float64 re-evaluation is allowed fp32 noise) ...
    assert np.all(margin[moved_got] <= 1e-5)
    assert np.array_equal(got[~moved_got], x[~moved_got])    # ... or are the untouched input
    # the reference's regime
    same = CarliniL2Method(classifier=clf, confidence=150.5, batch_size=6, binary_search_steps=2, max_iter=3).generate(x=x)
    assert np.array_equal(same, x)
    assert np.array_equal(A.carlini_l2(spec, p64, x[:3], confidence=150.5, batch_size=3, binary_search_steps=2, max_iter=3), x[:3].astype(np.float64))


def test_carlini_linf_matches_restated_art(cuda):
    """CarliniLInfMethod (attacks.py:578-582): the iterate stays in the eps box; with confidence 0 samples stop once
    misclassified; with the reference's confidence = 10 every sample keeps descending for max_iter steps."""
    from lipasr.attacks import CarliniLInfMethod, TensorFlowV2Classifier

    spec, p, m = _small_net()
    clf = TensorFlowV2Classifier(model=m, nb_classes=5, input_shape=(40,))
    x = np.random.default_rng(3).standard_normal((16, 40)).astype(np.float32)
    p64 = p.astype(np.float64)
    for conf in (0.0, 10.0):
        # ART iterates in float32 (ART_NUMPY_DTYPE): saturated samples (p_label = 1 - 1e-7) stall there, and so do we
        want = A.carlini_linf(spec, p, x, confidence=conf, eps=0.3, batch_size=8, dtype=np.float32)
        got = CarliniLInfMethod(classifier=clf, confidence=conf, eps=0.3, batch_size=8).generate(x=x)
        assert np.abs(got - x).max() <= 0.3 * (1 + 1e-5)
        # same own-prediction margin after the attack, sample by sample (the path there may differ in float noise)
        y = P.forward_infer(spec, p64, x.astype(np.float64))
        lab = np.eye(5)[y.argmax(1)]
        m_got = A._cw_margin(P.forward_infer(spec, p64, got.astype(np.float64)), lab, conf)
        m_want = A._cw_margin(P.forward_infer(spec, p64, want.astype(np.float64)), lab, conf)
        m_0 = A._cw_margin(y, lab, conf)
        assert np.all(m_got <= m_0 + 1e-6)                       # never worse than the start
        assert np.mean(np.abs(m_got - m_want) <= 0.05 * np.maximum(m_0, 1e-3)) >= 0.8
        if conf == 0.0:
            assert (m_got == 0).sum() >= 1
