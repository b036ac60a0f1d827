"""NumPy model of the block-DFT STFT kernel (csrc/mfcc.hip: stft_bdft_kernel) -- the arithmetic, stage by stage, with the
fp16 two-plane split emulated, against the float64 oracle.  Not a test: a design tool (error budget per stage).

  frame spectrum  X_f[k] = sum_{b<4} (-i)^(b k) B_{f+b}[k],   B_j[k] = sum_{m<512} ypad[512 j + m] e^(-2 pi i m k / 2048)
  Hann in the frequency domain (periodic Hann = 0.5 - 0.25 e^(+) - 0.25 e^(-)):  Xw[k] = 0.5 X[k] - 0.25 (X[k-1] + X[k+1])
  block DFT in two matrix stages, m = 32 n1 + n2, k = k1 + 64 k2:
      A[n2][k1]  = sum_{n1<16} y[32 n1 + n2] W64^(n1 k1)           (real data x complex 16-point matrix)
      A'[n2][k1] = A[n2][k1] W2048^(n2 k1)                          (fp32 twiddle)
      B[k1 + 64 k2] = sum_{n2<32} A'[n2][k1] W32^(n2 k2),  k2 < 16  (complex x complex, half of the outputs)
"""
import sys

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/asr-using-robust-nn_amd")
import numpy as np

from oracle import mfcc_ref as M

SIG, TAP, MID = 2048.0, 64.0, 1.0 / 1024.0


def split(v):
    v = np.asarray(v, dtype=np.float32)
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def mm3(a, b):
    """three cross terms of a two-plane product, fp32 result"""
    ah, al = split(a)
    bh, bl = split(b)
    return (ah @ bh + ah @ bl + al @ bh).astype(np.float32)


def block_dft(blocks, planes=True):
    """blocks [nb][512] float32 -> B [nb][1025] complex (float32 parts), scaled back to signal units"""
    nb = blocks.shape[0]
    n1 = np.arange(16)[:, None]
    k1 = np.arange(64)[None, :]
    C = np.concatenate([np.cos(2 * np.pi * n1 * k1 / 64), -np.sin(2 * np.pi * n1 * k1 / 64)], axis=1) * TAP  # [16][128]
    Y = blocks.reshape(nb, 16, 32).transpose(0, 2, 1) * np.float32(SIG)  # [nb][n2][n1]
    Y = np.clip(Y, -65000, 65000)
    if planes:
        A = np.stack([mm3(Y[b], C) for b in range(nb)])  # [nb][n2][128]
    else:
        A = (Y.astype(np.float64) @ C).astype(np.float32)
    Are, Aim = A[:, :, :64], A[:, :, 64:]
    n2 = np.arange(32)[:, None]
    tw = np.exp(-2j * np.pi * n2 * k1 / 2048) * MID
    twr, twi = tw.real.astype(np.float32), tw.imag.astype(np.float32)
    Apr = (Are * twr - Aim * twi).astype(np.float32)
    Api = (Are * twi + Aim * twr).astype(np.float32)
    k2 = np.arange(16)[None, :]
    c3, s3 = np.cos(2 * np.pi * n2 * k2 / 32) * TAP, np.sin(2 * np.pi * n2 * k2 / 32) * TAP
    E = np.block([[c3, -s3], [s3, c3]])  # rows (re n2 | im n2), cols (re k2 | im k2)
    out = np.zeros((nb, 1025), dtype=np.complex128)
    for b in range(nb):
        Aop = np.concatenate([Apr[b].T, Api[b].T], axis=1)  # [k1][64]
        D = mm3(Aop, E) if planes else (Aop.astype(np.float64) @ E).astype(np.float32)  # [k1][32]
        Bk = (D[:, :16] + 1j * D[:, 16:]).astype(np.complex64)  # [k1][k2]
        out[b, :1024] = Bk.T.reshape(-1)  # k = k1 + 64 k2
        sgn = np.where(np.arange(512) % 2 == 0, 1.0, -1.0)
        out[b, 1024] = np.float32((blocks[b].astype(np.float32) * sgn).sum(dtype=np.float32)) * (SIG * TAP * TAP * MID)
    return out / (SIG * TAP * TAP * MID)


def log_mel(y, planes=True, return_power=False):
    y = np.asarray(y, dtype=np.float32)
    yp = M.reflect_pad(y, 1024)
    n_frames = 1 + len(y) // 512
    nb = n_frames + 3
    blocks = np.zeros((nb, 512), dtype=np.float32)
    flat = yp[: 512 * nb]
    blocks.reshape(-1)[: len(flat)] = flat
    B = block_dft(blocks, planes)
    k = np.arange(1025)
    X = np.zeros((n_frames, 1025), dtype=np.complex128)
    for b in range(4):
        X += ((-1j) ** ((b * k) % 4))[None, :] * B[b : b + n_frames]
    X = X.astype(np.complex64).astype(np.complex128)
    Xm = np.concatenate([np.conj(X[:, 1:2]), X[:, :-1]], axis=1)  # X[k-1]
    Xp = np.concatenate([X[:, 1:], np.conj(X[:, 1023:1024])], axis=1)  # X[k+1]
    Xw = 0.5 * X - 0.25 * (Xm + Xp)
    P = (Xw.real.astype(np.float32) ** 2 + Xw.imag.astype(np.float32) ** 2).astype(np.float32)
    if return_power:
        return P.T
    mel = M.mel_filterbank() @ P.T
    return M.power_to_db(mel.astype(np.float32))


def mfcc(y, planes=True):
    return (M.dct_matrix() @ log_mel(y, planes).astype(np.float64)).astype(np.float32)


if __name__ == "__main__":
    from lipasr.synth import synth_clips

    rng = np.random.default_rng(0)
    waves, _ = synth_clips(6, seed=11)
    cases = {f"synth{i}": M.librosa_load_resample(waves[i], 16000, fast=True) for i in range(6)}
    t = np.arange(22050) / 22050.0
    cases["tone 1 kHz"] = (0.5 * np.sin(2 * np.pi * 1000.3 * t)).astype(np.float32)
    cases["tone 5 kHz + floor -90 dB"] = (0.5 * np.sin(2 * np.pi * 5000.7 * t) + 1.5e-5 * rng.standard_normal(22050)).astype(np.float32)
    cases["white noise full scale"] = rng.uniform(-1, 1, 22050).astype(np.float32)
    cases["quiet noise 1e-4"] = (1e-4 * rng.standard_normal(22050)).astype(np.float32)
    cases["chirp"] = (0.3 * np.sin(2 * np.pi * (200 * t + 4000 * t * t))).astype(np.float32)
    cases["short 3000"] = (0.2 * rng.standard_normal(3000)).astype(np.float32)
    for name, y in cases.items():
        ref64 = M.dct_matrix() @ M.power_to_db(M.mel_filterbank().astype(np.float64) @ M.power_spectrogram(y, np.float64))
        ref32 = M.mfcc_22k(y)
        S_ref = M.power_spectrogram(y, np.float64)
        for planes in (False, True):
            got = mfcc(y, planes)
            P = log_mel(y, planes, return_power=True)
            relp = np.abs(P - S_ref).max() / S_ref.max()
            print(f"{name:28s} planes={planes!s:5s} mfcc max|d| vs f64 {np.abs(got - ref64).max():.2e}  vs oracle f32 {np.abs(got - ref32).max():.2e}"
                  f"  (oracle f32 vs f64 {np.abs(ref32 - ref64).max():.2e})  power err / max power {relp:.1e}")
