"""Speaker-recognition variant of the path: the reference's second copy of the pipeline,
``Speaker recognition/``, differs from the voice-digit one in three places and only those live here --

* features: recordings already at 22 050 Hz are cut into 1-s windows (first and last second dropped) and each
  window goes through ``librosa.feature.mfcc(win_length=441, n_fft=441, hop_length=220)`` -> 20 x 101 = 2020
  (Speaker recognition/extract_features_construct_dataset.py:203-233);
* classifier: 2020 inputs, 20 speakers, batch 64 (Speaker recognition/train_constraints.py:41,63-88;
  train_no_constraints.py:52-74 for the baseline without BatchNorm / NonNeg);
* the constraint in use is ``simple_norm_constraint(rho=1)`` (train_constraints.py:103).

Constraints.py, the Lipschitz read-outs and the attacks are the same files as the voice-digit ones and are
re-exported.  The window length is not a power of two (441 = 3^2 7^2), so the STFT runs on the short-window
path of the MFCC plan (lipasr_mfcc_plan_ex: windowed real DFT as an fp32 MFMA contraction).
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N
from .Constraints import customConstraint, norm_constraint, norm_constraint_FISTA, simple_norm_constraint  # noqa: F401
from .extract_features_construct_dataset import (  # noqa: F401  (re-exported: the Speaker-recognition copy has the same read-outs)
    N_MFCC,
    MfccExtractor,
    extract_features,
    get_lipschitz_constrained,
    get_norms,
    get_upper_lipschitz,
    read_wav,
)
from .keras import BatchNormalization, Dense, Dropout, Input, Model, NonNeg

# the 20 speaker ids of the RoDigits split (extract_features_construct_dataset.py:11-12)
digit = ['006', '041', '043', '044', '045', '046', '047', '048', '049', '105', '117', '118', '211', '212',
         '213', '214', '215', '260', '261', '420']

SR = 22050
WIN_LENGTH = N_FFT = 441  # 20 ms at 22 050 Hz
HOP_LENGTH = 220
N_FRAMES = 1 + SR // HOP_LENGTH  # 101
N_FEATURES = N_MFCC * N_FRAMES   # 2020
N_SPEAKERS = 20


class WindowMfcc:
    """Plan + launch wrapper for [B, 22050] windows -> [B, 2020] on the short-window MFCC path (a plan of its own:
    ``MfccExtractor`` with n_fft = win_length = 441, hop 220)."""

    def __init__(self, batch_max=512, n_samp=SR, n_fft=N_FFT, hop_length=HOP_LENGTH, device=None):
        self._ex = MfccExtractor(SR, n_samp, batch_max, device, n_fft=n_fft, hop=hop_length)
        self.device, self.h = self._ex.device, self._ex.h
        self.batch_max, self.n_samp, self.n_fft, self.hop = int(batch_max), int(n_samp), int(n_fft), int(hop_length)
        self.n_y, self.n_frames = self._ex.n_y, self._ex.n_frames

    def set(self, key, value):
        self._ex.set(key, value)

    def profile_begin(self, n):
        self._ex.profile_begin(n)

    def profile_end(self):
        return self._ex.profile_end()

    def close(self):
        self._ex.close()

    def __call__(self, windows, mean=None, scale=None, out=None):
        return self._ex.from_22k(windows, self.n_frames, mean, scale, out)


_window_mfcc = {}


def mfcc_windows(windows, chunk=512):
    """Batched tensor entry: float32 [N, 22050] windows (tensor or array) -> device tensor [N, 2020]."""
    dev = torch.device("cuda", torch.cuda.current_device())
    w = torch.as_tensor(np.asarray(windows, dtype=np.float32) if not torch.is_tensor(windows) else windows)
    w = w.to(device=dev, dtype=torch.float32).contiguous()
    key = (w.shape[1], dev.index)
    ex = _window_mfcc.get(key)
    if ex is None:
        ex = _window_mfcc[key] = WindowMfcc(batch_max=chunk, n_samp=w.shape[1])
    out = torch.empty(w.shape[0], N_MFCC * ex.n_frames, device=dev)
    for s in range(0, w.shape[0], ex.batch_max):
        ex(w[s:s + ex.batch_max], out=out[s:s + ex.batch_max])
    return out


def split_windows(raw_w, sampling_rate=SR):
    """extract_features_construct_dataset.py:207-221: 1-s windows, first second and the tail dropped."""
    window_length = 1 * sampling_rate
    audio_length = int(len(raw_w) / window_length)
    raw_w = raw_w[window_length:(audio_length - 1) * window_length]
    audio_length = int(len(raw_w) / window_length)
    return np.asarray(raw_w[:audio_length * window_length], dtype=np.float32).reshape(audio_length, window_length)


def _load_22k(file_path):
    """librosa.load(file_path, mono=True): decode, mono mix, resample to 22 050 Hz on the device if needed."""
    x, sr = read_wav(file_path)
    if sr == SR:
        return x
    ex = MfccExtractor(sr, len(x), 1)
    return ex.resample(torch.as_tensor(x).cuda()[None])[0].cpu().numpy()


def load_audio_dataset_and_labels(filenames, labels):
    """extract_features_construct_dataset.py:203-233 -> (mfcc [num_seconds, 2020] float64, labels [num_seconds])."""
    windows, local_labels = [], []
    for i, file_path in enumerate(filenames):
        w = split_windows(_load_22k(file_path))
        windows.append(w)
        local_labels.extend([labels[i]] * len(w))
    if not windows or sum(len(w) for w in windows) == 0:
        return np.zeros((0, N_FEATURES)), np.array(local_labels)
    feats = mfcc_windows(np.concatenate(windows, axis=0))
    return feats.cpu().numpy().astype(np.float64), np.array(local_labels)


def get_file_names_and_labels(file_path):
    """Speaker recognition/extract_features_construct_dataset.py:114-137: the speaker folders that are present."""
    from .extract_features_construct_dataset import get_file_names_and_labels as _list

    return _list(file_path, classes=digit)


def main(data_dir="dataset/rodigits", save_dir="RoDigits_splitV2", noise_dir="test_dataset_to_add_noise", random_state=None):
    """Speaker recognition/extract_features_construct_dataset.py:236-267: recordings are split 70/20/10 BEFORE
    windowing, every recording becomes 1-s windows with its label repeated, and the windowed MFCCs are saved."""
    import os

    from .extract_features_construct_dataset import shuffle, split_train_dev_test

    filenames, labels = get_file_names_and_labels(data_dir)
    filenames, labels = shuffle(filenames, labels, random_state=random_state)
    parts = list(zip(("train", "dev", "test"), split_train_dev_test(filenames), split_train_dev_test(labels)))
    os.makedirs(noise_dir, exist_ok=True)
    os.makedirs(save_dir, exist_ok=True)
    out = {}
    for name, files, lab in parts:
        out[name] = load_audio_dataset_and_labels(files, lab)
    np.save(os.path.join(noise_dir, "test_label"), out["test"][1])
    np.save(os.path.join(noise_dir, "test_filenames"), parts[2][1])
    for name in ("train", "dev", "test"):
        np.save(os.path.join(save_dir, f"{name}_data"), out[name][0])
        np.save(os.path.join(save_dir, f"{name}_label"), out[name][1])
    return save_dir


def get_model(**kw):
    """Speaker recognition/train_constraints.py:63-88 (the voice-digit network with 2020 inputs, 20 outputs)."""
    inp = Input((N_FEATURES,))
    hdn = inp
    for units, drop in ((1024, 0.1), (512, 0.1), (256, 0.1), (128, 0.0), (64, 0.0)):
        hdn = Dense(units, activation="relu", kernel_constraint=NonNeg())(hdn)
        hdn = BatchNormalization()(hdn)
        if drop:
            hdn = Dropout(drop)(hdn)
    out = Dense(N_SPEAKERS, activation="softmax", kernel_constraint=NonNeg())(hdn)
    return Model(inputs=inp, outputs=out, **kw)


def get_model_unconstrained(**kw):
    """Speaker recognition/train_no_constraints.py:52-74: plain Dense/ReLU stack, no BatchNorm, no Dropout."""
    inp = Input((N_FEATURES,))
    hdn = inp
    for units in (1024, 512, 256, 128, 64):
        hdn = Dense(units, activation="relu")(hdn)
    out = Dense(N_SPEAKERS, activation="softmax")(hdn)
    return Model(inputs=inp, outputs=out, **kw)


def train_no_constraints_main(argv=None):
    """Speaker recognition/train_no_constraints.py:16-43,77-97: the unconstrained baseline on RoDigits_splitV2/*.npy
    (shuffle buffers 2000 / 1000, batch 64, EarlyStopping(patience=10)), then the per-layer norms and their product."""
    import argparse

    from .train_constraints import load_processed_dataset
    from .train_google_dataset import run

    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="RoDigits_splitV2/")
    ap.add_argument("--epochs", type=int, default=10000)
    ap.add_argument("--checkpoint", default="bin/models/baseline_splitV2.h5")
    args = ap.parse_args(argv)
    model, y, results = run(get_model_unconstrained(max_batch=64), load_processed_dataset(args.data), N_SPEAKERS, batch=64,
                            epochs=args.epochs, patience=10, checkpoint=args.checkpoint, max_batch=64, shuffle=(2000, 1000))
    lip = get_upper_lipschitz(get_norms(model))
    print(f"Upper Lipschitz constant for non-constrained model: {lip}")
    return model, results, lip


def train_constraints_main(argv=None):
    """Speaker recognition/train_constraints.py:17-42,91-113: the constrained network with
    simple_norm_constraint(rho=1, affected_layers_indices=[]), batch 64, then the BatchNorm-corrected Lipschitz constant."""
    import argparse

    from .train_constraints import load_processed_dataset
    from .train_google_dataset import run

    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="RoDigits_splitV2/")
    ap.add_argument("--epochs", type=int, default=10000)
    ap.add_argument("--rho", type=float, default=1.0)
    ap.add_argument("--checkpoint", default="bin/models_constrained/model_constrained_Rho1_splitV2_batch_norm.h5")
    args = ap.parse_args(argv)
    model, y, results = run(get_model(max_batch=64), load_processed_dataset(args.data), N_SPEAKERS, batch=64, epochs=args.epochs,
                            patience=2000, checkpoint=args.checkpoint, max_batch=64,
                            callbacks=[simple_norm_constraint(rho=args.rho, affected_layers_indices=[])])
    lip = get_lipschitz_constrained(model)
    print(f"Lipschitz constant for constrained model: {lip}")
    return model, results, lip
