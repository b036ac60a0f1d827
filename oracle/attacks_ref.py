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
