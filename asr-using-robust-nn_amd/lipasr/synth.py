"""Deterministic synthetic 'spoken digit' clips: the datasets and models of the
reference are absent (LFS pointers / git-ignored), so parity runs and the bench use these.

Per class c a formant-like triple in 200-3500 Hz, onset jitter, amplitude 0.1-0.5, N(0, 0.01)
noise, clipped to [-1, 1]; 1 s at 16 kHz.  Pure NumPy, shared by tests, bench and examples.
"""
from __future__ import annotations

import numpy as np

_FORMANTS = np.array([[270, 2290, 3010], [390, 1990, 2550], [530, 1840, 2480], [660, 1720, 2410], [730, 1090, 2440],
                      [570, 840, 2410], [440, 1020, 2240], [300, 870, 2240], [640, 1190, 2390], [490, 1350, 1690]], dtype=np.float64)


def synth_clips(n, seed=1234, n_samples=16000, sr=16000):
    """Returns (waves float32 [n, n_samples], labels int32 [n])."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 10, size=n).astype(np.int32)
    t = np.arange(n_samples) / sr
    waves = np.zeros((n, n_samples), dtype=np.float32)
    for i in range(n):
        f = _FORMANTS[labels[i]] * (1.0 + 0.03 * rng.standard_normal(3))
        amp = rng.uniform(0.1, 0.5)
        onset = rng.uniform(0.03, 0.06)
        dur = rng.uniform(0.35, 0.6)
        env = np.clip((t - onset) / 0.02, 0, 1) * np.clip((onset + dur - t) / 0.05, 0, 1)
        s = sum(a * np.sin(2 * np.pi * fj * t + rng.uniform(0, 2 * np.pi)) for a, fj in zip((1.0, 0.5, 0.25), f))
        w = amp * env * s / 1.75 + 0.01 * rng.standard_normal(n_samples)
        waves[i] = np.clip(w, -1, 1).astype(np.float32)
    return waves, labels


def synth_clips_fast(n, seed=1234, n_samples=16000, sr=16000):
    """Vectorised variant for large pools (bench): same recipe, one RNG draw order per array."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 10, size=n).astype(np.int32)
    t = (np.arange(n_samples) / sr).astype(np.float32)[None, :]
    f = (_FORMANTS[labels] * (1.0 + 0.03 * rng.standard_normal((n, 3)))).astype(np.float32)
    amp = rng.uniform(0.1, 0.5, (n, 1)).astype(np.float32)
    onset = rng.uniform(0.03, 0.06, (n, 1)).astype(np.float32)
    dur = rng.uniform(0.35, 0.6, (n, 1)).astype(np.float32)
    ph = rng.uniform(0, 2 * np.pi, (n, 3)).astype(np.float32)
    env = np.clip((t - onset) / 0.02, 0, 1) * np.clip((onset + dur - t) / 0.05, 0, 1)
    s = np.zeros((n, n_samples), dtype=np.float32)
    for j, a in enumerate((1.0, 0.5, 0.25)):
        s += a * np.sin(2 * np.pi * f[:, j:j + 1] * t + ph[:, j:j + 1])
    w = amp * env * s / 1.75 + 0.01 * rng.standard_normal((n, n_samples)).astype(np.float32)
    return np.clip(w, -1, 1).astype(np.float32), labels


def synth_clips_device(n, seed, device, n_samples=16000, sr=16000, chunk=4096):
    """The same recipe generated on the device with torch (plumbing for the bench's resident pool: 65 536 clips would be
    4 GB of host arrays and a minute of NumPy).  Returns (waves float32 [n, n_samples], labels int64 [n]) on ``device``.
    Different random stream than the NumPy generators; same distribution."""
    import torch

    g = torch.Generator(device=device).manual_seed(int(seed))
    formants = torch.as_tensor(_FORMANTS, dtype=torch.float32, device=device)
    t = (torch.arange(n_samples, device=device, dtype=torch.float32) / sr)[None, :]
    waves = torch.empty(n, n_samples, device=device, dtype=torch.float32)
    labels = torch.randint(0, 10, (n,), generator=g, device=device)
    for s in range(0, n, chunk):
        k = min(chunk, n - s)
        u = lambda *shape: torch.rand(*shape, generator=g, device=device)
        f = formants[labels[s:s + k]] * (1.0 + 0.03 * torch.randn(k, 3, generator=g, device=device))
        amp = 0.1 + 0.4 * u(k, 1)
        onset = 0.03 + 0.03 * u(k, 1)
        dur = 0.35 + 0.25 * u(k, 1)
        ph = 2 * np.pi * u(k, 3)
        env = ((t - onset) / 0.02).clamp(0, 1) * ((onset + dur - t) / 0.05).clamp(0, 1)
        sig = torch.zeros(k, n_samples, device=device)
        for j, a in enumerate((1.0, 0.5, 0.25)):
            sig += a * torch.sin(2 * np.pi * f[:, j:j + 1] * t + ph[:, j:j + 1])
        w = amp * env * sig / 1.75 + 0.01 * torch.randn(k, n_samples, generator=g, device=device)
        waves[s:s + k] = w.clamp(-1, 1)
    return waves, labels
