"""Oracle A1/A1b: waveform -> MFCC(20 x L) exactly as the reference obtains it.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED at the librosa
boundary: the arithmetic lives in third-party modules that are neither in
/root/reference nor installable here and that the reference does not pin:

  librosa  (era 0.8.1 - 0.9.2): ``load(path, mono=True)`` + ``feature.mfcc(y, sr)``
  resampy  (era 0.2.2)        : ``resample(y, 16000, 22050, filter='kaiser_best')``
  scipy.fftpack.dct

Call sites restated here:
  VD/extract_features_construct_dataset.py:24-39  (extract_features)
  VD/extract_features_construct_dataset.py:144-150 (compute_mfcc_all_files)
  VD/attacks.py:106-121, 262-274                  (same arithmetic after noise)

Algorithm restated (librosa <= 0.9 defaults):
  1. resample sr_in -> 22050 Hz with resampy ``kaiser_best`` (windowed sinc,
     64 zero crossings, 512 table steps per crossing, Kaiser beta
     14.769656459379492, roll-off 0.9475937167399596, linear table interpolation)
  2. reflect-pad 1024, frame 2048 / hop 512, periodic Hann, rFFT (float64,
     stored complex64), power |.|^2 (float32)
  3. Slaney mel filter bank 128 x 1025 (float32), slaney-normalised
  4. 10*log10(max(1e-10, .)), floor at (max over the clip) - 80 dB
  5. DCT-II ortho over the mel axis, keep 20 coefficients
  6. truncate / zero-pad the frame axis to ``utterance_length``.
"""
from __future__ import annotations

import functools
import numpy as np

SR_TARGET = 22050
N_FFT = 2048
HOP = 512
N_MELS = 128
N_MFCC = 20
STANDARD_UTTERANCE_LENGTH = 44  # VD/extract_features_construct_dataset.py:18

# resampy kaiser_best design constants
KB_NUM_ZEROS = 64
KB_PRECISION = 9  # 2**9 = 512 table entries per zero crossing
KB_ROLLOFF = 0.9475937167399596
KB_BETA = 14.769656459379492


@functools.lru_cache(maxsize=None)
def kaiser_best_half_window():
    """resampy.filters.sinc_window(num_zeros=64, precision=9, kaiser(beta), rolloff).

    Returns (interp_win float64 [32769], num_table=512)."""
    num_bits = 2 ** KB_PRECISION
    n = num_bits * KB_NUM_ZEROS
    sinc_win = KB_ROLLOFF * np.sinc(KB_ROLLOFF * np.linspace(0, KB_NUM_ZEROS, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, KB_BETA)[n:]
    return (taper * sinc_win).astype(np.float64), num_bits


def resample_kaiser_best(x: np.ndarray, sr_orig: int, sr_new: int, time_mode: str = "accumulate") -> np.ndarray:
    """resampy.resample(x, sr_orig, sr_new, filter='kaiser_best') for 1-D x, vectorised.

    ``time_mode='accumulate'`` follows resampy 0.2.x (time_register += 1/ratio in
    float64); ``'multiply'`` follows resampy >= 0.3 (t * (1/ratio)).  Products are
    summed in float64 and rounded once to float32 (the reference rounds the
    running sum to float32 after every tap; the difference is ~1e-7 relative).
    Samples outside [0, n) are treated as absent, exactly like the tap-count
    clamps ``i_max`` / ``k_max`` of resampy's loop.
    """
    x = np.asarray(x)
    if sr_orig == sr_new:
        return x.astype(np.float32, copy=True)
    n_orig = x.shape[0]
    sample_ratio = float(sr_new) / float(sr_orig)
    n_out = int(n_orig * sample_ratio)
    interp_win, num_table = kaiser_best_half_window()
    interp_win = interp_win.copy()
    if sample_ratio < 1:
        interp_win *= sample_ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    nwin = interp_win.shape[0]

    scale = min(1.0, sample_ratio)
    time_increment = 1.0 / sample_ratio
    index_step = int(scale * num_table)
    if time_mode == "accumulate":
        t_reg = np.concatenate(([0.0], np.cumsum(np.full(n_out - 1, time_increment, dtype=np.float64))))
    elif time_mode == "multiply":
        t_reg = np.arange(n_out, dtype=np.float64) * time_increment
    else:
        raise ValueError(time_mode)
    n = t_reg.astype(np.int64)
    xd = x.astype(np.float64)
    max_taps = nwin // index_step + 1
    y = np.zeros(n_out, dtype=np.float64)
    taps = np.arange(max_taps, dtype=np.int64)

    # left wing: x[n - i]
    frac = scale * (t_reg - n)
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    i_max = np.minimum(n + 1, (nwin - offset) // index_step)
    widx = offset[:, None] + taps[None, :] * index_step
    valid = taps[None, :] < i_max[:, None]
    widx_c = np.where(valid, widx, 0)
    w = interp_win[widx_c] + eta[:, None] * interp_delta[widx_c]
    xi = n[:, None] - taps[None, :]
    xv = xd[np.where(valid, xi, 0)]
    y += np.sum(np.where(valid, w * xv, 0.0), axis=1)

    # right wing: x[n + 1 + k]
    frac = scale - frac
    index_frac = frac * num_table
    offset = index_frac.astype(np.int64)
    eta = index_frac - offset
    k_max = np.minimum(n_orig - n - 1, (nwin - offset) // index_step)
    widx = offset[:, None] + taps[None, :] * index_step
    valid = taps[None, :] < k_max[:, None]
    widx_c = np.where(valid, widx, 0)
    w = interp_win[widx_c] + eta[:, None] * interp_delta[widx_c]
    xi = n[:, None] + taps[None, :] + 1
    xv = xd[np.where(valid, xi, 0)]
    y += np.sum(np.where(valid, w * xv, 0.0), axis=1)
    return y.astype(np.float32)


def resample_kaiser_best_loop(x, sr_orig, sr_new):
    """Pure-Python transcription of resampy's ``resample_f`` loop (small inputs only).

    Used by tests to check the vectorised version above."""
    x = np.asarray(x, dtype=np.float64)
    n_orig = x.shape[0]
    sample_ratio = float(sr_new) / float(sr_orig)
    n_out = int(n_orig * sample_ratio)
    interp_win, num_table = kaiser_best_half_window()
    interp_win = interp_win.copy()
    if sample_ratio < 1:
        interp_win *= sample_ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    nwin = interp_win.shape[0]
    scale = min(1.0, sample_ratio)
    time_increment = 1.0 / sample_ratio
    index_step = int(scale * num_table)
    time_register = 0.0
    y = np.zeros(n_out, dtype=np.float64)
    for t in range(n_out):
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        for i in range(i_max):
            w = interp_win[offset + i * index_step] + eta * interp_delta[offset + i * index_step]
            y[t] += w * x[n - i]
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        for k in range(k_max):
            w = interp_win[offset + k * index_step] + eta * interp_delta[offset + k * index_step]
            y[t] += w * x[n + k + 1]
        time_register += time_increment
    return y.astype(np.float32)


def hann_periodic(n: int = N_FFT) -> np.ndarray:
    """scipy.signal.get_window('hann', n, fftbins=True), float64."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


@functools.lru_cache(maxsize=None)
def mel_filterbank(sr: int = SR_TARGET, n_fft: int = N_FFT, n_mels: int = N_MELS) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels=128, fmin=0, fmax=sr/2, htk=False, norm='slaney') -> float32."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0, float(sr) / 2, n_bins, endpoint=True)
    mels = np.linspace(_hz_to_mel(0.0), _hz_to_mel(float(sr) / 2), n_mels + 2)
    mel_f = _mel_to_hz(mels)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2 : n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis].astype(np.float32)
    return weights


@functools.lru_cache(maxsize=None)
def dct_matrix(n_mfcc: int = N_MFCC, n_mels: int = N_MELS) -> np.ndarray:
    """Rows 0..n_mfcc-1 of scipy.fftpack.dct(type=2, norm='ortho') as a matrix, float64."""
    n = np.arange(n_mels)
    k = np.arange(n_mfcc)[:, None]
    d = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_mels)) * np.sqrt(2.0 / n_mels)
    d[0] *= np.sqrt(0.5)
    return d


def reflect_pad(y: np.ndarray, pad: int) -> np.ndarray:
    return np.pad(y, pad, mode="reflect")


def power_spectrogram(y: np.ndarray) -> np.ndarray:
    """|STFT|^2 with librosa defaults (center=True, reflect, periodic Hann).  float32 [1025, T]."""
    y = np.asarray(y, dtype=np.float32)
    yp = reflect_pad(y, N_FFT // 2)
    n_frames = 1 + (len(yp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[:, None] + HOP * np.arange(n_frames)[None, :]
    frames = yp[idx]  # [2048, T] float32
    win = hann_periodic(N_FFT)[:, None]  # float64
    spec = np.fft.rfft(win * frames, axis=0).astype(np.complex64)
    return (np.abs(spec) ** 2.0).astype(np.float32)


def power_to_db(S: np.ndarray, amin: float = 1e-10, top_db: float = 80.0) -> np.ndarray:
    """librosa.power_to_db(S, ref=1.0, amin=1e-10, top_db=80)."""
    S = np.asarray(S, dtype=np.float32)
    log_spec = (10.0 * np.log10(np.maximum(np.float32(amin), S))).astype(np.float32)
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - np.float32(top_db))
    return log_spec


def mfcc_22k(y: np.ndarray) -> np.ndarray:
    """librosa.feature.mfcc(y=y, sr=22050) -> float32 [20, T]."""
    S = power_spectrogram(y)
    mel = mel_filterbank() @ S  # float32 sgemm in the reference
    db = power_to_db(mel)
    return (dct_matrix() @ db.astype(np.float64)).astype(np.float32)


def fix_frames(m: np.ndarray, utterance_length: int) -> np.ndarray:
    """VD/extract_features_construct_dataset.py:33-37: truncate or zero-pad the frame axis."""
    if m.shape[1] > utterance_length:
        return m[:, :utterance_length]
    return np.pad(m, ((0, 0), (0, utterance_length - m.shape[1])), mode="constant", constant_values=0)


def extract_features_wave(x: np.ndarray, sr_in: int = 16000, utterance_length: int = STANDARD_UTTERANCE_LENGTH) -> np.ndarray:
    """extract_features() for an in-memory mono float waveform at ``sr_in``. -> float32 [20, L]."""
    y = resample_kaiser_best(np.asarray(x, dtype=np.float32), sr_in, SR_TARGET)
    return fix_frames(mfcc_22k(y), utterance_length)


def compute_mfcc_batch(waves: np.ndarray, sr_in: int = 16000, utterance_length: int = STANDARD_UTTERANCE_LENGTH) -> np.ndarray:
    """compute_mfcc_all_files() on an in-memory batch [N, n]: per-clip loop, flatten coeff-major, float64 [N, 20*L]."""
    out = np.zeros((len(waves), N_MFCC * utterance_length))
    for i in range(len(waves)):
        out[i] = extract_features_wave(waves[i], sr_in, utterance_length).flatten()
    return out


def synth_clips(n: int, seed: int = 1234, n_samples: int = 16000, sr: int = 16000):
    """Deterministic synthetic 'spoken digit' clips: formant-like triples per class.

    Returns (waves float32 [n, n_samples], labels int32 [n])."""
    rng = np.random.default_rng(seed)
    labels = rng.integers(0, 10, size=n).astype(np.int32)
    base = np.array([[270, 2290, 3010], [390, 1990, 2550], [530, 1840, 2480], [660, 1720, 2410], [730, 1090, 2440],
                     [570, 840, 2410], [440, 1020, 2240], [300, 870, 2240], [640, 1190, 2390], [490, 1350, 1690]], dtype=np.float64)
    t = np.arange(n_samples) / sr
    waves = np.zeros((n, n_samples), dtype=np.float32)
    for i in range(n):
        f = base[labels[i]] * (1.0 + 0.03 * rng.standard_normal(3))
        amp = rng.uniform(0.1, 0.5)
        onset = rng.uniform(0.03, 0.06)
        dur = rng.uniform(0.35, 0.6)
        env = np.clip((t - onset) / 0.02, 0, 1) * np.clip((onset + dur - t) / 0.05, 0, 1)
        s = sum(a * np.sin(2 * np.pi * fj * t + rng.uniform(0, 2 * np.pi)) for a, fj in zip((1.0, 0.5, 0.25), f))
        w = amp * env * s / 1.75 + 0.01 * rng.standard_normal(n_samples)
        waves[i] = np.clip(w, -1, 1).astype(np.float32)
    return waves, labels
