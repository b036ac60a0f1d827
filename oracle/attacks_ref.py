"""Oracle A9, A10, A12: FGSM / PGD (ART defaults) and the audio-domain noise models.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: adversarial-robustness-toolbox
(era 1.9-1.10) is absent; update rules restated from its published algorithm.  Call sites:

  VD/attacks.py:500-510  TensorFlowV2Classifier + FastGradientMethod(estimator, eps).generate(x)
  VD/attacks.py:650-661  ProjectedGradientDescent(estimator, eps).generate(x)
  VD/attacks.py:73-86, 145-183, 222-245  noise models (reference uses the unseeded global NumPy RNG;
  here the generator is passed in).
"""
from __future__ import annotations

import numpy as np
from . import mlp_ref


def _own_labels(spec, p, x, batch_size):
    """ART: y=None -> labels := one-hot argmax of the estimator's own prediction."""
    preds = np.concatenate([mlp_ref.forward_infer(spec, p, x[i:i + batch_size]) for i in range(0, len(x), batch_size)])
    y = np.zeros_like(preds)
    y[np.arange(len(preds)), preds.argmax(axis=1)] = 1
    return y


def fgsm(spec, p, x, eps, batch_size=32, y=None):
    """FastGradientMethod(norm=inf, targeted=False, minimal=False, num_random_init=0): x + eps*sign(g); no clipping."""
    x = np.asarray(x)
    y = _own_labels(spec, p, x, batch_size) if y is None else y
    adv = x.copy()
    for i in range(0, len(x), batch_size):
        g = mlp_ref.input_gradient_infer(spec, p, x[i:i + batch_size], y[i:i + batch_size])
        g = np.where(np.isnan(g), 0.0, g)
        adv[i:i + batch_size] = x[i:i + batch_size] + eps * np.sign(g)
    return adv


def pgd(spec, p, x, eps, eps_step=0.1, max_iter=100, batch_size=32, y=None):
    """ProjectedGradientDescent(norm=inf, num_random_init=0): x <- x0 + clip(x + eps_step*sign(g) - x0, +-eps)."""
    x = np.asarray(x)
    y = _own_labels(spec, p, x, batch_size) if y is None else y
    adv = x.copy()
    for i in range(0, len(x), batch_size):
        x0 = x[i:i + batch_size]
        xa = x0.copy()
        yb = y[i:i + batch_size]
        for _ in range(max_iter):
            g = mlp_ref.input_gradient_infer(spec, p, xa, yb)
            g = np.where(np.isnan(g), 0.0, g)
            xa = xa + eps_step * np.sign(g)
            xa = x0 + np.clip(xa - x0, -eps, eps)
        adv[i:i + batch_size] = xa
    return adv


def sign_step(x_adv, x0, g, alpha, eps):
    """One fused PGD update (K4): x0 + clip(x_adv + alpha*sign(g) - x0, -eps, eps); eps=inf gives FGSM."""
    xa = x_adv + alpha * np.sign(g)
    return x0 + np.clip(xa - x0, -eps, eps)


# ------------------------------------------------------------------ (f)-4: class gradients, JSMA
def class_gradient(spec, p, x, labels):
    """ART TensorFlowV2Classifier.class_gradient(x, label=labels)[:, 0, :]: gradient of the OUTPUT (softmax
    probability) of class labels[b] w.r.t. x_b.  VD/attacks.py:538-645 are the call sites of the attacks using it."""
    labels = np.broadcast_to(np.asarray(labels), (len(x),))
    v = np.zeros((len(x), spec[-1].n_out))
    v[np.arange(len(x)), labels] = 1.0
    return mlp_ref.output_vjp_infer(spec, p, x, v, on_logits=False)[0]


def jsma(spec, p, x, targets, theta=0.1, gamma=1.0, batch_size=1, max_iter=None):
    """SaliencyMapMethod(classifier, theta, gamma).generate(x) (VD/attacks.py:546-550: theta=10, gamma=0.1), restated
    from ART 1.9-1.10's published algorithm WITHOUT clip_values (the reference passes none): while the prediction
    differs from the target and at most gamma of the features were touched, add theta to the two features with the
    largest target-class gradient (smallest when theta < 0).  With no clip values the search space never shrinks,
    so ART can loop forever on a sample that never reaches its target; ``max_iter`` (None = ART's behaviour) bounds
    it.  ``targets`` (int array) replaces ART's unseeded random_targets draw."""
    x = np.asarray(x)
    adv = x.astype(np.float64 if x.dtype == np.float64 else np.float32).copy()
    nf = adv.shape[1]
    preds = np.concatenate([mlp_ref.forward_infer(spec, p, x[i:i + batch_size]) for i in range(0, len(x), batch_size)]).argmax(1)
    targets = np.asarray(targets)
    for s0 in range(0, len(adv), batch_size):
        batch = adv[s0:s0 + batch_size]
        tgt = targets[s0:s0 + batch_size]
        cur = preds[s0:s0 + batch_size]
        active = np.where(cur != tgt)[0]
        all_feat = np.zeros_like(batch)
        it = 0
        while active.size != 0 and (max_iter is None or it < max_iter):
            g = class_gradient(spec, p, batch[active], tgt[active])
            if theta > 0:
                ind = np.argpartition(g, -2, axis=1)[:, -2:]
            else:
                ind = np.argpartition(-g, -2, axis=1)[:, -2:]
            all_feat[active, ind[:, 0]] = 1
            all_feat[active, ind[:, 1]] = 1
            tmp = batch[active]
            tmp[np.arange(len(active)), ind[:, 0]] += theta
            tmp[np.arange(len(active)), ind[:, 1]] += theta
            batch[active] = tmp
            cur = mlp_ref.forward_infer(spec, p, batch).argmax(1)
            active = np.where((cur != tgt) & (all_feat.sum(axis=1) / nf <= gamma))[0]
            it += 1
        adv[s0:s0 + batch_size] = batch
    return adv


# ------------------------------------------------------------------ A12
def add_white_noise(array, sigma, rng):
    """VD/attacks.py:73-86."""
    return array + rng.normal(0, sigma, np.array(array).shape[0])


def mixtgauss(N, p, sigma0, sigma1, rng):
    """VD/attacks.py:145-163: u = |q| < p ; x = (sigma0*(1-u) + sigma1*u) * N(0,1)."""
    q = rng.normal(0, 1, N)
    u = np.abs(q) < p
    return (sigma0 * (1 - u) + sigma1 * u) * rng.normal(0, 1, N)


def add_noise(x, p, alpha, rng):
    """VD/attacks.py:166-183: impulse mixture with sigma1 = 10*alpha."""
    return x + mixtgauss(x.shape[0], p, alpha, 10 * alpha, rng)


def add_white_noise_with_snr(audio, target_snr_db, rng):
    """VD/attacks.py:222-245."""
    sample = np.asanyarray(audio)
    signal_avg_watts = np.mean(sample ** 2)
    signal_avg_db = 10 * np.log10(signal_avg_watts)
    noise_avg_watts = 10 ** ((signal_avg_db - target_snr_db) / 10)
    return sample + rng.normal(0, np.sqrt(noise_avg_watts), len(sample))
