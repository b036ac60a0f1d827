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
    """Plan + launch wrapper of lipasr_mfcc_plan / lipasr_mfcc_f32 for one (sr_in, n_samp, batch_max)."""

    def __init__(self, sr_in=16000, n_samp=16000, batch_max=512, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.h = N.get_handle(self.device.index)
        self.sr_in, self.n_samp, self.batch_max = int(sr_in), int(n_samp), int(batch_max)
        self._plan()

    def _plan(self):
        N.check(N.lib.lipasr_mfcc_plan(self.h.h, self.sr_in, self.n_samp, self.batch_max))
        self.h.mfcc_owner = self
        ny, nf = C.c_int(), C.c_int()
        N.check(N.lib.lipasr_mfcc_dims(self.h.h, C.byref(ny), C.byref(nf)))
        self.n_y, self.n_frames = ny.value, nf.value

    def _own(self):
        # one MFCC plan lives in the handle; re-plan if another extractor replaced it
        if getattr(self.h, "mfcc_owner", None) is not self:
            self._plan()

    def __call__(self, waves, utterance_length=STANDARD_UTTERANCE_LENGTH, mean=None, scale=None, out=None):
        """waves: float32 device tensor [B, n_samp] -> [B, 20*utterance_length] (coefficient-major)."""
        self._own()
        b = waves.shape[0]
        if out is None:
            out = torch.empty(b, N_MFCC * utterance_length, device=self.device)
        N.check(N.lib.lipasr_mfcc_f32(self.h.h, N.ptr(waves), b, utterance_length, N.ptr(mean), N.ptr(scale), N.ptr(out), N.stream_ptr()))
        return out

    def resample(self, waves, out=None):
        self._own()
        y = torch.empty(waves.shape[0], self.n_y, device=self.device) if out is None else out
        N.check(N.lib.lipasr_resample_f32(self.h.h, N.ptr(waves), waves.shape[0], N.ptr(y), N.stream_ptr()))
        return y

    def from_22k(self, y, utterance_length=STANDARD_UTTERANCE_LENGTH, mean=None, scale=None, out=None):
        self._own()
        if out is None:
            out = torch.empty(y.shape[0], N_MFCC * utterance_length, device=self.device)
        N.check(N.lib.lipasr_mfcc_from_22k(self.h.h, N.ptr(y), y.shape[0], y.shape[1], utterance_length, N.ptr(mean), N.ptr(scale),
                                           N.ptr(out), N.stream_ptr()))
        return out


_extractors = {}


def _extractor(sr_in, n_samp, batch_max):
    key = (sr_in, n_samp, torch.cuda.current_device())
    ex = _extractors.get(key)
    if ex is None or ex.batch_max < batch_max:
        ex = MfccExtractor(sr_in, n_samp, batch_max)
        _extractors[key] = ex
    return ex


def mfcc(waveforms, sr_in=16000, utterance_length=STANDARD_UTTERANCE_LENGTH):
    """Batched entry: float32 [B, n] (tensor or array) at ``sr_in`` Hz -> device tensor [B, 20*utterance_length]."""
    w = torch.as_tensor(np.asarray(waveforms, dtype=np.float32) if not torch.is_tensor(waveforms) else waveforms)
    w = w.to(device=torch.device("cuda", torch.cuda.current_device()), dtype=torch.float32).contiguous()
    return _extractor(int(sr_in), w.shape[1], w.shape[0])(w, utterance_length)


def read_wav(file_path):
    """librosa.load's decode + mono mix (float32 in [-1, 1)); returns (samples, sampling_rate)."""
    with wave.open(str(file_path), "rb") as f:
        sr, nch, width, n = f.getframerate(), f.getnchannels(), f.getsampwidth(), f.getnframes()
        raw = f.readframes(n)
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


def compute_mfcc_all_files(filenames):
    """Reference :144-150: (N, 880) float64, files of equal length batched through one launch."""
    feats = np.zeros((len(filenames), N_MFCC * STANDARD_UTTERANCE_LENGTH))
    groups = {}
    for i, fn in enumerate(filenames):
        x, sr = read_wav(fn)
        groups.setdefault((sr, len(x)), []).append((i, x))
    for (sr, n), items in groups.items():
        for s in range(0, len(items), 512):
            chunk = items[s:s + 512]
            w = np.stack([x for _, x in chunk])
            f = mfcc(w, sr, STANDARD_UTTERANCE_LENGTH).cpu().numpy()
            for (i, _), row in zip(chunk, f):
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
