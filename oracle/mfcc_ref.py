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


@functools.lru_cache(maxsize=None)
def _polyphase_table(sr_orig: int, sr_new: int):
    """Per-output-phase taps of the kaiser_best interpolator for an up-sampling ratio L/M (float64).

    Output t = L q + p always sees the fractional position (p M mod L)/L, so resampy's interpolated
    weights can be tabulated once per phase: h[p] applies to x[M q + floor(p M / L) - 63 .. + 64]."""
    g = int(np.gcd(sr_orig, sr_new))
    L, Mdown = sr_new // g, sr_orig // g
    assert sr_new > sr_orig, "table form is used for up-sampling only (exact step of 512)"
    win, num_table = kaiser_best_half_window()
    delta = np.zeros_like(win); delta[:-1] = np.diff(win)
    wing = win.shape[0] // num_table  # 64
    h = np.zeros((L, 2 * wing))
    n_off = np.zeros(L, dtype=np.int64)
    taps = np.arange(wing)
    for p in range(L):
        num = p * Mdown
        n_off[p] = num // L
        frac = (num % L) / L
        idx_f = frac * num_table; off = int(idx_f); eta = idx_f - off
        i_max = (win.shape[0] - off) // num_table
        wl = np.where(taps < i_max, win[np.minimum(off + taps * num_table, win.shape[0] - 1)] + eta * delta[np.minimum(off + taps * num_table, win.shape[0] - 1)], 0.0)
        h[p, wing - 1 - taps] = wl
        idx_f = (1.0 - frac) * num_table; off = int(idx_f); eta = idx_f - off
        k_max = (win.shape[0] - off) // num_table
        wr = np.where(taps < k_max, win[np.minimum(off + taps * num_table, win.shape[0] - 1)] + eta * delta[np.minimum(off + taps * num_table, win.shape[0] - 1)], 0.0)
        h[p, wing + taps] = wr
    return h, n_off, L, Mdown, wing


def resample_kaiser_best_fast(x: np.ndarray, sr_orig: int, sr_new: int) -> np.ndarray:
    """Same result as resample_kaiser_best (to ~1e-7) for up-sampling, organised per phase so that it
    runs at the speed of the reference's compiled resampler (used by bench.py's cpu_baseline leg)."""
    x = np.asarray(x, dtype=np.float64)
    h, n_off, L, Mdown, wing = _polyphase_table(sr_orig, sr_new)
    n_out = int(x.shape[0] * (float(sr_new) / float(sr_orig)))
    nq = (n_out + L - 1) // L
    xp = np.concatenate([np.zeros(wing - 1), x, np.zeros(2 * wing + Mdown * (nq + 1) - 0)])
    win = np.lib.stride_tricks.sliding_window_view(xp, 2 * wing)  # win[i] = x[i-63 .. i+64]
    y = np.empty((nq, L))
    base = Mdown * np.arange(nq)
    for p in range(L):
        y[:, p] = win[base + n_off[p]] @ h[p]
    return y.reshape(-1)[:n_out].astype(np.float32)


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


def power_spectrogram(y: np.ndarray, dtype=np.float32, n_fft: int = N_FFT, hop: int = HOP) -> np.ndarray:
    """|STFT|^2 with librosa defaults (center=True, reflect, periodic Hann, win_length = n_fft).  [1 + n_fft//2, T].

    ``dtype=np.float32`` is the clean path (librosa.load returns float32, the STFT is stored complex64).
    ``dtype=np.float64`` is what librosa <= 0.9 does for the noisy-audio paths of VD/attacks.py, where
    ``raw_w + np.random.normal(...)`` is float64 and the STFT is inferred complex128."""
    y = np.asarray(y, dtype=dtype)
    ctype = np.complex64 if dtype == np.float32 else np.complex128
    yp = reflect_pad(y, n_fft // 2)
    n_frames = 1 + (len(yp) - n_fft) // hop
    idx = np.arange(n_fft)[:, None] + hop * np.arange(n_frames)[None, :]
    frames = yp[idx]  # [n_fft, T]
    win = hann_periodic(n_fft)[:, None]  # float64
    spec = np.fft.rfft(win * frames, axis=0).astype(ctype)
    return (np.abs(spec) ** 2.0).astype(dtype)


def power_to_db(S: np.ndarray, amin: float = 1e-10, top_db: float = 80.0) -> np.ndarray:
    """librosa.power_to_db(S, ref=1.0, amin=1e-10, top_db=80); keeps S's float dtype."""
    S = np.asarray(S)
    dt = S.dtype.type
    log_spec = (10.0 * np.log10(np.maximum(dt(amin), S))).astype(S.dtype)
    if top_db is not None:
        log_spec = np.maximum(log_spec, log_spec.max() - dt(top_db))
    return log_spec


def mfcc_22k(y: np.ndarray, dtype=np.float32, n_fft: int = N_FFT, hop: int = HOP) -> np.ndarray:
    """librosa.feature.mfcc(y=y, sr=22050[, n_fft=, win_length=n_fft, hop_length=]) -> [20, T] in ``dtype``
    (float32 clean path, float64 noisy path and the Speaker-recognition windows).

    The DCT is evaluated in float64 and rounded once; scipy.fftpack.dct ran in the input dtype, the
    difference is bounded in tests/test_oracle_cpu.py::test_oracle_rounding_budget."""
    S = power_spectrogram(y, dtype, n_fft, hop)
    mel = mel_filterbank(n_fft=n_fft).astype(dtype) @ S  # float32 sgemm in the reference's clean path
    db = power_to_db(mel)
    return (dct_matrix() @ db.astype(np.float64)).astype(dtype)


def fix_frames(m: np.ndarray, utterance_length: int) -> np.ndarray:
    """VD/extract_features_construct_dataset.py:33-37: truncate or zero-pad the frame axis."""
    if m.shape[1] > utterance_length:
        return m[:, :utterance_length]
    return np.pad(m, ((0, 0), (0, utterance_length - m.shape[1])), mode="constant", constant_values=0)


def librosa_load_resample(x: np.ndarray, sr_in: int, fast: bool = False) -> np.ndarray:
    """librosa.load's resampling step: resampy.resample then librosa.resample's
    ``util.fix_length(y_hat, int(np.ceil(n * ratio)))`` (resampy yields int(n * ratio) samples; when
    n * ratio is not an integer librosa appends one zero sample)."""
    x = np.asarray(x, dtype=np.float32)
    if sr_in == SR_TARGET:
        return x.copy()
    y = resample_kaiser_best_fast(x, sr_in, SR_TARGET) if (fast and sr_in < SR_TARGET) else resample_kaiser_best(x, sr_in, SR_TARGET)
    n_fixed = int(np.ceil(x.shape[0] * (float(SR_TARGET) / float(sr_in))))
    if len(y) < n_fixed:
        y = np.pad(y, (0, n_fixed - len(y)))
    return y[:n_fixed]


def extract_features_wave(x: np.ndarray, sr_in: int = 16000, utterance_length: int = STANDARD_UTTERANCE_LENGTH, fast: bool = False) -> np.ndarray:
    """extract_features() for an in-memory mono float waveform at ``sr_in``. -> float32 [20, L]."""
    y = librosa_load_resample(x, sr_in, fast)
    return fix_frames(mfcc_22k(y), utterance_length)


def compute_mfcc_batch(waves: np.ndarray, sr_in: int = 16000, utterance_length: int = STANDARD_UTTERANCE_LENGTH, fast: bool = False) -> np.ndarray:
    """compute_mfcc_all_files() on an in-memory batch [N, n]: per-clip loop, flatten coeff-major, float64 [N, 20*L]."""
    out = np.zeros((len(waves), N_MFCC * utterance_length))
    for i in range(len(waves)):
        out[i] = extract_features_wave(waves[i], sr_in, utterance_length, fast).flatten()
    return out


# --- Speaker recognition variant ------------------------------------------------------------------------------
SR_N_FFT, SR_HOP = 441, 220  # SR/extract_features_construct_dataset.py:225-226 (20 ms window at 22 050 Hz)


def sr_split_windows(raw_w: np.ndarray, sampling_rate: int = SR_TARGET) -> np.ndarray:
    """SR/extract_features_construct_dataset.py:207-221: 1-s windows of a recording that is already at 22 050 Hz,
    the first second and the last (more than one) second dropped.  -> [n_windows, sampling_rate]."""
    window_length = 1 * sampling_rate
    audio_length = int(len(raw_w) / window_length)
    raw_w = raw_w[window_length:(audio_length - 1) * window_length]
    audio_length = int(len(raw_w) / window_length)
    return np.array([raw_w[i * window_length:(i + 1) * window_length] for i in range(audio_length)]).reshape(audio_length, window_length)


def sr_mfcc_windows(windows: np.ndarray) -> np.ndarray:
    """SR/extract_features_construct_dataset.py:223-232: librosa.feature.mfcc(y=float64 window, sr, win_length=441,
    n_fft=441, hop_length=220) per window, [20, 101] flattened coefficient-major -> [N, 2020] float64."""
    out = np.zeros((len(windows), N_MFCC * (1 + windows.shape[1] // SR_HOP)))
    for j in range(len(windows)):
        out[j] = mfcc_22k(np.asarray(windows[j], dtype=np.float64), np.float64, SR_N_FFT, SR_HOP).reshape(-1)
    return out
