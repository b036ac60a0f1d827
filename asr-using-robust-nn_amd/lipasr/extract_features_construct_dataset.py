"""extract_features_construct_dataset.py surface (reference :24-39, :144-196) over liblipasr.

* ``extract_features(file_path, utterance_length)`` / ``compute_mfcc_all_files(filenames)`` keep the
  reference signatures; the wav is decoded on the host (stdlib ``wave``: 16-bit PCM, what the Speech
  Commands corpus ships) and everything after decoding -- resampling to 22 050 Hz, STFT, mel, dB,
  DCT, pad/trim, flatten -- runs in the K1 kernels, batched.
* ``mfcc(waveforms, sr_in)`` is the batched tensor entry the GPU pipeline uses.
* ``get_norms`` / ``get_upper_lipschitz`` / ``get_lipschitz_constrained`` are the Lipschitz
  read-outs, computed by the K3 kernels instead of host SVDs.
"""
from __future__ import annotations

import ctypes as C
import wave

import numpy as np
import torch

from . import _native as N
from .keras import Model

STANDARD_UTTERANCE_LENGTH = 44  # reference :18
N_MFCC = 20


class MfccExtractor:
    """One native MFCC plan (lipasr_mfcc_create: tables + intermediates of its own) for clips of up to ``n_samp`` samples
    at ``sr_in`` Hz, ``batch_max`` clips per launch.  Extractors do not share state: a pipeline's and a validation
    pass's extractor, or two streams, coexist on one handle.  ``n_fft`` / ``hop`` other than 2048 / 512 select the
    short-window path (Speaker recognition)."""

    def __init__(self, sr_in=16000, n_samp=16000, batch_max=512, device=None, n_fft=2048, hop=512):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.h = N.get_handle(self.device.index)
        self.sr_in, self.n_samp, self.batch_max = int(sr_in), int(n_samp), int(batch_max)
        plan = N.c_h()
        N.check(N.lib.lipasr_mfcc_create(self.h.h, self.sr_in, self.n_samp, self.batch_max, int(n_fft), int(hop), C.byref(plan)))
        self._plan = plan
        N.register_owner(self)
        ny, nf, fu = C.c_int(), C.c_int(), C.c_int()
        N.check(N.lib.lipasr_mfcc_plan_dims(plan, C.byref(ny), C.byref(nf), C.byref(fu)))
        self.n_y, self.n_frames, self.fused = ny.value, nf.value, bool(fu.value)

    def close(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan and self.h.alive:
            N.destroy_or_defer(N.lib.lipasr_mfcc_destroy, plan)  # (a finaliser may run in the middle of a graph capture)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set(self, key, value):
        """lipasr_mfcc_plan_set: key 0 = stage mask (64 = round-2 STFT kernel, 128 = never fuse), key 1 = resampler workgroups,
        key 2 = 1: the fused resample -> STFT kernel for every batch (default: only where the three kernels cannot read the
        input: int16 / ragged rows that are not a multiple of 4 samples long).  Stage-mask 256 = the Stockham STFT kernel
        (stft_mel2_kernel, the parity reference of the block-DFT kernel); key 3 = frames per workgroup of the block-DFT kernel
        (a multiple of 4, default 44); key 4 = 1: that kernel also applies the top_db floor and the DCT (no dct_kernel launch;
        bit-identical, slower on cache-cold batches)."""
        N.check(N.lib.lipasr_mfcc_plan_set(self._plan, int(key), int(value)))

    def profile_begin(self, max_calls):
        N.check(N.lib.lipasr_mfcc_plan_profile_begin(self._plan, int(max_calls)))

    def profile_end(self):
        """-> ({'resample', 'stft_mel', 'dct'} mean milliseconds, number of extractions timed)"""
        ms3, n = (C.c_float * 3)(), C.c_int()
        N.check(N.lib.lipasr_mfcc_plan_profile_end(self._plan, ms3, C.byref(n)))
        return {"resample": ms3[0], "stft_mel": ms3[1], "dct": ms3[2]}, n.value

    def __call__(self, waves, utterance_length=STANDARD_UTTERANCE_LENGTH, mean=None, scale=None, out=None, n_valid=None):
        """waves: device tensor [B, n_samp], float32 in [-1, 1) or int16 PCM -> [B, 20*utterance_length] (coefficient-major).
        n_valid: int32 device tensor [B]: samples of each row that belong to the clip (clips of different lengths in one
        launch; the rest of a row is ignored)."""
        if self._plan is None:
            raise RuntimeError("MfccExtractor used after close()")
        b = waves.shape[0]
        if waves.shape[1] != self.n_samp or not waves.is_contiguous():
            raise ValueError(f"waves must be contiguous [B, {self.n_samp}], got {tuple(waves.shape)}")
        if waves.dtype == torch.int16:
            fmt = 1
        elif waves.dtype == torch.float32:
            fmt = 0
        else:
            raise ValueError(f"waves must be float32 or int16, got {waves.dtype}")
        if n_valid is not None and (n_valid.dtype != torch.int32 or n_valid.shape[0] != b):
            raise ValueError("n_valid must be an int32 device tensor [B]")
        if out is None:
            out = torch.empty(b, N_MFCC * utterance_length, device=self.device)
        N.check(N.lib.lipasr_mfcc_extract(self._plan, N.ptr(waves), fmt, N.ptr(n_valid), b, utterance_length, N.ptr(mean), N.ptr(scale),
                                          N.ptr(out), N.stream_ptr()))
        return out

    def resample(self, waves, out=None):
        y = torch.empty(waves.shape[0], self.n_y, device=self.device) if out is None else out
        N.check(N.lib.lipasr_mfcc_plan_resample(self._plan, N.ptr(waves), waves.shape[0], N.ptr(y), N.stream_ptr()))
        return y

    def from_22k(self, y, utterance_length=STANDARD_UTTERANCE_LENGTH, mean=None, scale=None, out=None):
        if out is None:
            out = torch.empty(y.shape[0], N_MFCC * utterance_length, device=self.device)
        N.check(N.lib.lipasr_mfcc_plan_from_22k(self._plan, N.ptr(y), y.shape[0], y.shape[1], utterance_length, N.ptr(mean), N.ptr(scale),
                                                N.ptr(out), N.stream_ptr()))
        return out


_extractors = {}


def _extractor(sr_in, n_samp, batch_max):
    key = (sr_in, n_samp, torch.cuda.current_device())
    ex = _extractors.get(key)
    if ex is None or ex.batch_max < batch_max or ex._plan is None or not ex.h.alive:
        if len(_extractors) >= 8:  # plans hold ~0.1 MB per clip of batch_max: keep a handful
            _extractors.pop(next(iter(_extractors))).close()
        ex = MfccExtractor(sr_in, n_samp, batch_max)
        _extractors[key] = ex
    return ex


def mfcc(waveforms, sr_in=16000, utterance_length=STANDARD_UTTERANCE_LENGTH, n_valid=None):
    """Batched entry: [B, n] float32 or int16 PCM (tensor or array) at ``sr_in`` Hz -> device tensor [B, 20*utterance_length].
    n_valid ([B] ints): clips of different lengths, each row zero-padded (or not: the tail is ignored) to n."""
    dev = torch.device("cuda", torch.cuda.current_device())
    if torch.is_tensor(waveforms):
        w = waveforms
    else:
        a = np.asarray(waveforms)
        w = torch.as_tensor(a if a.dtype == np.int16 else a.astype(np.float32))
    if w.dtype != torch.int16:
        w = w.to(dtype=torch.float32)
    w = w.to(device=dev).contiguous()
    nv = None
    if n_valid is not None:
        nv = torch.as_tensor(np.asarray(n_valid, dtype=np.int32)).to(dev) if not torch.is_tensor(n_valid) else n_valid.to(device=dev, dtype=torch.int32)
    return _extractor(int(sr_in), w.shape[1], w.shape[0])(w, utterance_length, n_valid=nv)


def read_wav(file_path, pcm16=False):
    """librosa.load's decode + mono mix (float32 in [-1, 1)); returns (samples, sampling_rate).
    pcm16=True: a 16-bit mono file comes back as its raw int16 samples (the device scales them by 2^-15 while it stages
    them, lipasr_mfcc_i16: half the bytes over PCIe and from HBM); any other file still comes back as float32."""
    with wave.open(str(file_path), "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
    if pcm16 and width == 2 and nch == 1:
        return np.frombuffer(raw, dtype="<i2").copy(), sr
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width} in {file_path}")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1).astype(np.float32)
    return x, sr


def extract_features(file_path, utterance_length):
    """Reference :24-39: MFCC(20 x utterance_length) of one wav file, float32 NumPy."""
    x, sr = read_wav(file_path)
    out = mfcc(x[None, :], sr, utterance_length)
    return out.view(N_MFCC, utterance_length).cpu().numpy()


def compute_mfcc_all_files(filenames, chunk=512):
    """Reference :144-150: (N, 880) float64.  Files are decoded on the host (16-bit mono files stay int16) and go to the
    device in chunks of ``chunk`` clips of ANY lengths: one launch per chunk, each clip processed with its own length
    (n_valid), rows padded to the chunk's longest clip rounded up to a multiple of 4000 samples so that a corpus of
    one-second clips uses one plan."""
    feats = np.zeros((len(filenames), N_MFCC * STANDARD_UTTERANCE_LENGTH))
    by_sr = {}
    for i, fn in enumerate(filenames):
        x, sr = read_wav(fn, pcm16=True)
        by_sr.setdefault((sr, x.dtype == np.int16), []).append((i, x))
    for (sr, is_pcm), items in by_sr.items():
        ragged_ok = sr in (16000, 8000)  # the rates whose plans take per-clip lengths; other rates: one launch per distinct length
        if ragged_ok:
            batches = [items[s:s + chunk] for s in range(0, len(items), chunk)]
        else:
            groups = {}
            for it in items:
                groups.setdefault(len(it[1]), []).append(it)
            batches = [g[s:s + chunk] for g in groups.values() for s in range(0, len(g), chunk)]
        for b in batches:
            lens = np.array([len(x) for _, x in b], dtype=np.int32)
            n_max = int(lens.max())
            n_pad = -(-n_max // 4000) * 4000 if ragged_ok else n_max
            w = np.zeros((len(b), n_pad), dtype=np.int16 if is_pcm else np.float32)
            for r, (_, x) in enumerate(b):
                w[r, :len(x)] = x
            f = mfcc(w, sr, STANDARD_UTTERANCE_LENGTH, n_valid=lens if ragged_ok else None).cpu().numpy()
            for (i, _), row in zip(b, f):
                feats[i] = row
    return feats


# ------------------------------------------------------------------------------------------------ Lipschitz read-outs
def _dense_kernels(model):
    if isinstance(model, Model):
        return model.dense_kernels()
    dev = torch.device("cuda", torch.cuda.current_device())
    return [torch.as_tensor(np.asarray(l.get_weights()[0], dtype=np.float32)).to(dev).contiguous() for l in model.layers if "dense" in l.name]


def get_norms(model, iters=64):
    """Reference :154-161: sigma_max of every Dense kernel (power iteration on the device)."""
    ks = _dense_kernels(model)
    dev = ks[0].device
    h = N.get_handle(dev.index)
    out = torch.zeros(len(ks), device=dev)
    for i, k in enumerate(ks):
        v = torch.zeros(k.shape[1], device=dev)
        N.check(N.lib.lipasr_sigma_max(h.h, N.ptr(k), k.shape[0], k.shape[1], N.ptr(v), 0, iters, 0, N.ptr(out[i:i + 1]), N.stream_ptr()))
    return out.cpu().numpy().astype(np.float64)


def get_upper_lipschitz(norms):
    """Reference :165-166."""
    return np.prod(norms)


def product_norm(model):
    """||W_m^T ... W_1^T||_2 as a device scalar tensor."""
    ks = _dense_kernels(model)
    dev = ks[0].device
    h = N.get_handle(dev.index)
    sig = torch.zeros(1, device=dev)
    ptrs = N.ptr_array([k.data_ptr() for k in ks])
    N.check(N.lib.lipasr_product_norm(h.h, C.cast(ptrs, N.PV), N.int_array([k.shape[0] for k in ks]), N.int_array([k.shape[1] for k in ks]),
                                      len(ks), N.ptr(sig), N.stream_ptr()))
    return sig


def get_lipschitz_constrained(model):
    """Reference :169-196: product norm divided by prod over BatchNorm layers of max_j sqrt(var_j)/gamma_j."""
    sig = product_norm(model)
    dev = sig.device
    h = N.get_handle(dev.index)
    factors = []
    for layer in model.layers:
        if "batch" in layer.name:
            if isinstance(model, Model):
                gamma, _, _, var = layer._tensors()
            else:
                ws = layer.get_weights()
                gamma = torch.as_tensor(np.asarray(ws[0], dtype=np.float32)).to(dev)
                var = torch.as_tensor(np.asarray(ws[3], dtype=np.float32)).to(dev)
            f = torch.zeros(1, device=dev)
            N.check(N.lib.lipasr_bn_correction(h.h, N.ptr(gamma), N.ptr(var), gamma.numel(), N.ptr(f), N.stream_ptr()))
            factors.append(f)
    cst = float(sig.item())
    correction = float(np.prod([float(f.item()) for f in factors])) if factors else 1.0
    return cst / correction


# ------------------------------------------------------------------------------------------------ dataset construction
digit = ['zero', 'one', 'two', 'three', 'four', 'five', 'six', 'seven', 'eight', 'nine']  # reference :120


def get_file_names_and_labels(file_path, classes=None):
    """Reference :118-141: the class folders of ``file_path`` that exist (in the order of ``digit``), every file in
    them, and one integer label per file -- the label is the folder's rank among the folders that are PRESENT."""
    import glob
    import os

    classes = digit if classes is None else classes
    present = set(os.listdir(str(file_path)))
    filenames, labels = [], []
    for i, name in enumerate([c for c in classes if c in present]):
        found = sorted(glob.glob(os.path.join(str(file_path), name, "*")))
        filenames += found
        labels += [i] * len(found)
    return filenames, np.array(labels)


def shuffle(*arrays, random_state=None):
    """sklearn.utils.shuffle (reference :205): one permutation applied to every argument; lists stay lists."""
    if not arrays:
        return None
    n = len(arrays[0])
    if any(len(a) != n for a in arrays):
        raise ValueError("shuffle: arguments of different lengths")
    rng = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
    perm = rng.permutation(n)
    out = [[a[i] for i in perm] if isinstance(a, (list, tuple)) else np.asarray(a)[perm] for a in arrays]
    return out[0] if len(out) == 1 else out


def split_train_dev_test(items):
    """Reference :208-214: [:0.7 n], [0.7 n : 0.9 n], [-0.1 n:] with the reference's int() truncations (for n < 10 the
    last slice is ``[-0:]``, i.e. everything, exactly as the reference's expression evaluates)."""
    n = len(items)
    a, b, c = int(n * 0.7), int(n * 0.9), int(n * 0.1)
    return items[:a], items[a:b], items[-c:]


def main(data_dir="data", save_dir="processed_google_dataset", noise_dir="test_dataset_to_add_noise", random_state=None):
    """Reference :198-232 (the ``__main__`` block): list, shuffle, split 70/20/10, MFCC of every file on the GPU,
    and the eight ``.npy`` files train_constraints.py:16-25 and attacks.py read."""
    import os

    filenames, labels = get_file_names_and_labels(data_dir)
    filenames, labels = shuffle(filenames, labels, random_state=random_state)
    filenames_train, filenames_dev, filenames_test = split_train_dev_test(filenames)
    labels_train, labels_dev, labels_test = split_train_dev_test(labels)
    os.makedirs(noise_dir, exist_ok=True)
    os.makedirs(save_dir, exist_ok=True)
    np.save(os.path.join(noise_dir, "test_label"), labels_test)
    np.save(os.path.join(noise_dir, "test_filenames"), filenames_test)
    for name, files, lab in (("train", filenames_train, labels_train), ("dev", filenames_dev, labels_dev),
                             ("test", filenames_test, labels_test)):
        np.save(os.path.join(save_dir, f"{name}_data"), compute_mfcc_all_files(files))
        np.save(os.path.join(save_dir, f"{name}_label"), lab)
    return save_dir


if __name__ == "__main__":
    main()
