"""The evaluation script at the bottom of attacks.py (:296-693) without its ``input()`` prompts: each prompt is an
argument, each branch one accuracy-vs-strength sweep over a constrained and an unconstrained model.  The sweeps only
orchestrate -- noise, MFCC, gradients and attacks are the kernels behind ``lipasr.attacks``.  Plotting (matplotlib
windows) is left to the caller: every sweep returns ``(grid, {model name: accuracies})``.

    prompt in attacks.py                                   argument here
    "standardized before or after the attack? [B]/[A]"     standardize="before" | "after"          (:320-322)
    "Black-box or white-box attack? [B]/[W]"               attack="black" | "white"                (:324)
    "[S]imple/[M]ixture/[SNR]"                             kind="simple" | "mixture" | "snr"       (:326)
    "noise over [A]udio or [M]FCC"                         over="audio" | "mfcc"                   (:327)
    "[F]GSM/Carlini[L2]/Carlini[Linf]/[P]GD/[J]SMA"        kind="fgsm" | "l2" | "linf" | "pgd" | "jsma"   (:494)
"""
from __future__ import annotations

import argparse

import numpy as np

from . import attacks as A
from .keras import CategoricalCrossentropy, load_model, to_categorical

# the grids the reference hard-codes
AUDIO_SIGMAS = [0, 0.002, 0.004, 0.01, 0.015, 0.02, 0.03, 0.04, 0.05, 0.075, 0.1]  # :329
AUDIO_ALPHAS = np.linspace(0, 0.015, 15)                                              # :313
SNRS = [60, 30, 20, 15, 10, 5, 0]                                                     # :311
MFCC_SIGMAS = np.linspace(0, 100, 20)                                                 # :312
MFCC_ALPHAS = np.linspace(0, 100, 30)                                                 # :452
MIXTURE_P = 0.01                                                                      # :355, :451


def accuracy(predictions, labels_onehot):
    """np.sum(argmax(pred) == argmax(labels)) / len(labels) (:337-338)."""
    return float(np.sum(np.argmax(predictions, axis=1) == np.argmax(labels_onehot, axis=1)) / len(labels_onehot))


def _sweep(models, grid, make_data, labels, what):
    acc = {name: [] for name in models}
    for item in grid:
        for name, model in models.items():
            a = accuracy(model.predict(make_data(name, model, item)), labels)
            acc[name].append(a)
            print(f"Accuracy on {what} test examples{'' if name == 'constrained' else ' ' + name}: {a * 100}% ({item})")
    return list(grid), {k: np.asarray(v) for k, v in acc.items()}


def black_box_sweep(models, train_data, val_data, test_data, test_labels, kind="simple", over="mfcc", standardize="before",
                    test_filenames=None, grid=None, points=None, seed=None):
    """attacks.py:326-491.  ``models``: {"constrained": model, "unconstrained": model}; data as load_npy_dataset returns
    it; ``test_labels`` one-hot.  Noise over audio re-extracts the MFCCs of ``test_filenames`` and standardizes them with
    the statistics of (train, val, noisy test) exactly as :333 does."""
    if standardize == "before" and over == "mfcc":
        train_data, val_data, test_data = A.standardize_dataset(train_data, val_data, test_data)
    if over == "audio":
        if test_filenames is None:
            raise ValueError("noise over audio needs test_filenames (test_dataset_to_add_noise/test_filenames.npy)")
        if standardize == "before":
            train_data, val_data, _ = A.standardize_dataset(train_data, val_data, test_data)  # :320-322 precede the sweep
        grid = grid if grid is not None else {"simple": AUDIO_SIGMAS, "mixture": AUDIO_ALPHAS, "snr": SNRS}[kind]

        def make(name, model, item):
            if kind == "simple":
                d = A.black_box_attack_on_audio_dataset(test_filenames, item, p=0, alpha=0, seed=seed)
            elif kind == "mixture":
                d = A.black_box_attack_on_audio_dataset(test_filenames, sigma=0, p=MIXTURE_P, alpha=item, seed=seed)
            else:
                d = A.black_box_attack_on_audio_dataset_snr(test_filenames, item, seed=seed)
            return A.standardize_dataset(train_data, val_data, d)[2]
    elif over == "mfcc":
        if kind == "snr":
            raise ValueError("the SNR attack is defined on audio only (attacks.py:391)")
        grid = grid if grid is not None else {"simple": MFCC_SIGMAS, "mixture": MFCC_ALPHAS}[kind]

        def make(name, model, item):
            d = (A.add_white_noise_on_dataset(test_data, item, seed=seed) if kind == "simple"
                 else A.add_noise_mixture_on_dataset(dataset=test_data, p=MIXTURE_P, alpha=item, seed=seed))
            return A.standardize_dataset(train_data, val_data, d)[2] if standardize == "after" else d
    else:
        raise ValueError("over must be 'audio' or 'mfcc'")
    return _sweep(models, list(grid)[:points], make, test_labels, "black-box attack")


def white_box_sweep(models, train_data, val_data, test_data, test_labels, kind="fgsm", standardize="before", grid=None,
                    points=None, limit=None, **attack_kw):
    """attacks.py:493-693.  The models are wrapped as TensorFlowV2Classifier(model=, nb_classes=, input_shape=,
    loss_object=) (:500-504) and attacked with ART's constructor keywords; JSMA runs on the first 100 test rows (:552)."""
    if standardize == "before":
        train_data, val_data, test_data = A.standardize_dataset(train_data, val_data, test_data)
    n_classes, n_in = test_labels.shape[1], test_data.shape[1]
    clfs = {name: A.TensorFlowV2Classifier(model=m, nb_classes=n_classes, input_shape=(n_in,), loss_object=CategoricalCrossentropy())
            for name, m in models.items()}
    if kind == "fgsm":
        default = np.linspace(1, 30, 50) if standardize == "after" else np.linspace(0.01, 0.3, 10)  # :497-499
        build = lambda clf, item: A.FastGradientMethod(estimator=clf, eps=item, **attack_kw)
    elif kind == "pgd":
        default = np.linspace(1, 30, 50)                                                             # :648
        build = lambda clf, item: A.ProjectedGradientDescent(estimator=clf, eps=item, **attack_kw)
    elif kind == "jsma":
        default, limit = [10], (100 if limit is None else limit)                                     # :539, :552
        build = lambda clf, item: A.SaliencyMapMethod(classifier=clf, theta=item, gamma=0.1, **attack_kw)
    elif kind == "linf":
        default = [10]                                                                               # :572
        build = lambda clf, item: A.CarliniLInfMethod(classifier=clf, confidence=item, **attack_kw)
    elif kind == "l2":
        default = np.linspace(1, 300, 3)                                                             # :607
        build = lambda clf, item: A.CarliniL2Method(classifier=clf, confidence=item, **attack_kw)
    else:
        raise ValueError(f"unknown white-box attack {kind!r}")
    grid = list(default if grid is None else grid)[:points]
    x = np.asarray(test_data[:limit] if limit else test_data, dtype=np.float32)
    labels = test_labels[:limit] if limit else test_labels

    def make(name, clf, item):
        adv = build(clf, item).generate(x=x)
        return A.standardize_dataset(train_data, val_data, adv)[2] if standardize == "after" else adv

    return _sweep(clfs, grid, make, labels, "adversarial")


def main(argv=None):
    ap = argparse.ArgumentParser(description="attacks.py's evaluation menu as flags")
    ap.add_argument("--path", default="processed_google_dataset/")
    ap.add_argument("--noise-dir", default="test_dataset_to_add_noise")
    ap.add_argument("--constrained", default="bin/models_constrained/model_constrained_Rho01_dropout01.h5")
    ap.add_argument("--unconstrained", default="bin/models/baseline.h5")
    ap.add_argument("--standardize", choices=["before", "after"], default="before")
    ap.add_argument("--attack", choices=["black", "white"], default="black")
    ap.add_argument("--kind", default="simple", help="black: simple|mixture|snr; white: fgsm|l2|linf|pgd|jsma")
    ap.add_argument("--over", choices=["audio", "mfcc"], default="mfcc")
    ap.add_argument("--points", type=int, default=None, help="keep only the first N grid points")
    args = ap.parse_args(argv)
    import os

    train_data, _, val_data, _, test_data, test_label = A.load_npy_dataset(args.path)
    n_classes = int(test_label.max()) + 1
    labels = to_categorical(test_label, n_classes)
    models = {"constrained": load_model(args.constrained), "unconstrained": load_model(args.unconstrained)}
    if args.attack == "black":
        names = np.load(os.path.join(args.noise_dir, "test_filenames.npy")).tolist() if args.over == "audio" else None
        if names is not None:
            labels = to_categorical(np.load(os.path.join(args.noise_dir, "test_label.npy")), n_classes)  # :298-304
        return black_box_sweep(models, train_data, val_data, test_data, labels, kind=args.kind, over=args.over,
                               standardize=args.standardize, test_filenames=names, points=args.points)
    return white_box_sweep(models, train_data, val_data, test_data, labels, kind=args.kind, standardize=args.standardize,
                           points=args.points)


if __name__ == "__main__":
    main()
