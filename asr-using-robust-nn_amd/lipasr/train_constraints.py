"""The reference's constrained-training driver (train_constraints.py:63-111) on lipasr.

``get_model()`` and ``lip_stats_callback`` keep the reference's shape; ``main()`` reads like
train_constraints.py:91-111 but trains on synthetic clips (the reference's ``.npy`` features are LFS
pointers that are absent from the mount) pushed through the on-GPU MFCC kernel.
"""
from __future__ import annotations

import argparse

import numpy as np

from .Constraints import customConstraint, norm_constraint, norm_constraint_FISTA, simple_norm_constraint  # noqa: F401
from .attacks import standardize_dataset
from .extract_features_construct_dataset import get_lipschitz_constrained, get_norms, mfcc
from .keras import (BatchNormalization, Callback, CategoricalCrossentropy, Dataset, Dense, Dropout, EarlyStopping, Input, Model,
                    ModelCheckpoint, NonNeg, TensorBoard, load_model, to_categorical)
from .synth import synth_clips_fast


def tensorboard_callback():
    """train_constraints.py:45-48."""
    import datetime

    logdir = "logs/log_constrained" + datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    return TensorBoard(log_dir=logdir)


class lip_stats_callback(Callback):
    """train_constraints.py:52-60: per-layer spectral norms and the network Lipschitz constant each epoch."""

    def on_epoch_begin(self, epoch, logs=None):
        lip_cst = get_lipschitz_constrained(self.model)
        norms = get_norms(self.model)
        dense = [l for l in self.model.layers if "dense" in l.name]
        for layer, norm in zip(dense, norms):
            print(f"The norm for layer {layer} is : {norm}")
        print(f"The Lipschitz constant on epoch {epoch} is {lip_cst}")


def get_model(n_in=880, n_classes=10, **kw):
    """train_constraints.py:63-88."""
    inp = Input((n_in,))
    hdn = Dense(1024, activation="relu", kernel_constraint=NonNeg())(inp)
    hdn = BatchNormalization()(hdn)
    hdn = Dropout(0.1)(hdn)

    hdn = Dense(512, activation="relu", kernel_constraint=NonNeg())(hdn)
    hdn = BatchNormalization()(hdn)
    hdn = Dropout(0.1)(hdn)

    hdn = Dense(256, activation="relu", kernel_constraint=NonNeg())(hdn)
    hdn = BatchNormalization()(hdn)
    hdn = Dropout(0.1)(hdn)

    hdn = Dense(128, activation="relu", kernel_constraint=NonNeg())(hdn)
    hdn = BatchNormalization()(hdn)

    hdn = Dense(64, activation="relu", kernel_constraint=NonNeg())(hdn)
    hdn = BatchNormalization()(hdn)

    out = Dense(n_classes, activation="softmax", kernel_constraint=NonNeg())(hdn)
    return Model(inputs=inp, outputs=out, **kw)


def get_model_unconstrained(n_in=880, n_classes=10, **kw):
    """train_google_dataset.py:49-74: no NonNeg, Dropout(0.4) after every hidden block."""
    inp = Input((n_in,))
    hdn = inp
    for units in (1024, 512, 256, 128, 64):
        hdn = Dense(units, activation="relu")(hdn)
        hdn = BatchNormalization()(hdn)
        hdn = Dropout(0.4)(hdn)
    out = Dense(n_classes, activation="softmax")(hdn)
    return Model(inputs=inp, outputs=out, **kw)


def synthetic_dataset(n_train=16566, n_dev=4733, n_test=2366, seed=1234):
    """Synthetic stand-in with the reference's split sizes, extracted by the K1 kernel."""
    waves, labels = synth_clips_fast(n_train + n_dev + n_test, seed)
    feats = np.concatenate([mfcc(waves[s:s + 512]).cpu().numpy() for s in range(0, len(waves), 512)]).astype(np.float64)
    a, b = n_train, n_train + n_dev
    return (feats[:a], labels[:a]), (feats[a:b], labels[a:b]), (feats[b:], labels[b:])


def load_processed_dataset(path="processed_google_dataset/"):
    """train_constraints.py:16-25: the six ``.npy`` files extract_features_construct_dataset.main() writes."""
    import os

    def ld(name):
        return np.load(os.path.join(path, name + ".npy"))

    return (ld("train_data"), ld("train_label")), (ld("dev_data"), ld("dev_label")), (ld("test_data"), ld("test_label"))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--rho", type=float, default=0.1)
    ap.add_argument("--small", action="store_true", help="2048/512/512 clips instead of the reference's split sizes")
    ap.add_argument("--data", default=None, help="directory with train/dev/test _data.npy and _label.npy "
                                                 "(train_constraints.py:16-25); default: synthetic clips")
    args = ap.parse_args(argv)
    sizes = (2048, 512, 512) if args.small else (16566, 4733, 2366)
    if args.data:
        (train_data, train_label), (val_data, val_label), (test_data, test_label1) = load_processed_dataset(args.data)
    else:
        (train_data, train_label), (val_data, val_label), (test_data, test_label1) = synthetic_dataset(*sizes)
    train_label, val_label, test_label = (to_categorical(l, 10) for l in (train_label, val_label, test_label1))
    train_data, val_data, test_data = standardize_dataset(train_data, val_data, test_data)

    train_dataset = Dataset.from_tensor_slices((train_data, train_label)).shuffle(880, reshuffle_each_iteration=False).batch(512)
    val_dataset = Dataset.from_tensor_slices((val_data, val_label)).shuffle(880, reshuffle_each_iteration=False).batch(512)

    model = get_model()
    model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    print(model.summary())
    model.fit(train_dataset, epochs=args.epochs, validation_data=val_dataset, verbose=2,
              callbacks=[EarlyStopping(monitor="val_loss", patience=6000, restore_best_weights=False),
                         simple_norm_constraint(rho=args.rho, affected_layers_indices=[]),
                         lip_stats_callback(),
                         ModelCheckpoint("bin/models_constrained/TEST.h5", save_best_only=True, verbose=1)])
    model = load_model("bin/models_constrained/TEST.h5")
    y = np.argmax(model.predict(test_data), axis=1)
    results = model.evaluate(test_data, test_label)
    print(f"Test loss: {results[0]} / Test accuracy: {results[1]} / agreement {np.mean(y == test_label1)}")


if __name__ == "__main__":
    main()
