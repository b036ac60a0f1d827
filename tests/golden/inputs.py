"""Deterministic inputs shared by make_golden.py and the tests (data generators, no algorithm)."""
import numpy as np


def test_clips(n_samples=16000, sr=16000):
    """Four analytic clips: 1 kHz tone, 200->3500 Hz chirp, white noise, full-scale clipped square-ish wave."""
    t = np.arange(n_samples) / sr
    rng = np.random.default_rng(20221004)
    tone = 0.5 * np.sin(2 * np.pi * 1000.0 * t)
    chirp = 0.4 * np.sin(2 * np.pi * (200.0 * t + 0.5 * (3300.0 / (n_samples / sr)) * t * t))
    noise = 0.1 * rng.standard_normal(n_samples)
    clipped = np.clip(3.0 * np.sin(2 * np.pi * 440.0 * t), -1.0, 1.0)
    return np.stack([tone, chirp, noise, clipped]).astype(np.float32)


def nonneg_kernels(widths, seed=7):
    """Seeded non-negative fp32 Dense kernels (in, out) for the given layer widths."""
    rng = np.random.default_rng(seed)
    ws = []
    for a, b in zip(widths[:-1], widths[1:]):
        lim = np.sqrt(6.0 / (a + b))
        ws.append(np.abs(rng.uniform(-lim, lim, size=(a, b))).astype(np.float32))
    return ws


def signed_kernels(widths, seed=11):
    rng = np.random.default_rng(seed)
    return [rng.uniform(-1, 1, size=(a, b)).astype(np.float32) * np.float32(np.sqrt(6.0 / (a + b))) for a, b in zip(widths[:-1], widths[1:])]


SMALL_WIDTHS = [96, 64, 32, 10]
FULL_WIDTHS = [880, 1024, 512, 256, 128, 64, 10]
MLP_SMALL = dict(widths=[40, 32, 16, 10], bn=[1, 1, 0], dropout=[0.25, 0.0, 0.0], nonneg=[1, 1, 1])


def mlp_small_case(batch=8, seed=3):
    rng = np.random.default_rng(seed)
    w = MLP_SMALL["widths"]
    x = rng.standard_normal((batch, w[0])).astype(np.float32)
    labels = rng.integers(0, w[-1], size=batch)
    y = np.zeros((batch, w[-1]), dtype=np.float32)
    y[np.arange(batch), labels] = 1
    keep = rng.uniform(size=(batch, w[1])) > MLP_SMALL["dropout"][0]
    mask0 = (keep / (1.0 - MLP_SMALL["dropout"][0])).astype(np.float32)
    return x, y, [mask0, None, None]
