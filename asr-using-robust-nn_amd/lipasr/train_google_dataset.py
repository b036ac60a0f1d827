"""The reference's unconstrained-baseline driver (train_google_dataset.py:14-99; BASELINE config 1) on lipasr:
pre-extracted MFCCs from ``processed_google_dataset/*.npy`` -> StandardScaler -> Dense/BatchNorm/Dropout(0.4) stack
trained with Adam, EarlyStopping(patience=200) and ModelCheckpoint('bin/models/baselineV2.h5') -> reload, predict,
evaluate, confusion matrix.  Also the shared body of the Speaker-recognition drivers (``run``).
"""
from __future__ import annotations

import argparse

import numpy as np

from .attacks import standardize_dataset
from .keras import Dataset, EarlyStopping, ModelCheckpoint, load_model, to_categorical
from .train_constraints import get_model_unconstrained as get_model  # noqa: F401  (train_google_dataset.py:49-74)
from .train_constraints import load_processed_dataset, synthetic_dataset


def confusion_matrix(labels, predictions, num_classes=None):
    """tf.math.confusion_matrix (train_google_dataset.py:94): rows = true label, columns = prediction, int32 counts."""
    labels = np.asarray(labels).astype(np.int64).ravel()
    predictions = np.asarray(predictions).astype(np.int64).ravel()
    if labels.shape != predictions.shape:
        raise ValueError("labels and predictions differ in length")
    n = int(max(labels.max(initial=-1), predictions.max(initial=-1)) + 1) if num_classes is None else int(num_classes)
    out = np.zeros((n, n), dtype=np.int32)
    np.add.at(out, (labels, predictions), 1)
    return out


def run(model, splits, n_classes, batch, epochs, patience, checkpoint, callbacks=(), max_batch=None, shuffle=(880, 880)):
    """The body every training driver of the reference shares: one-hot labels, joint standardisation
    (:26-33), shuffle(880).batch(B) datasets (:39-40), fit with EarlyStopping + ModelCheckpoint, reload the best
    checkpoint, predict / evaluate on the test split.  Returns (reloaded model, test predictions, [loss, accuracy])."""
    (train_data, train_label), (val_data, val_label), (test_data, test_label1) = splits
    train_label, val_label, test_label = (to_categorical(l, n_classes) for l in (train_label, val_label, test_label1))
    train_data, val_data, test_data = standardize_dataset(train_data, val_data, test_data)
    train_dataset = Dataset.from_tensor_slices((train_data, train_label)).shuffle(shuffle[0], reshuffle_each_iteration=False).batch(batch)
    val_dataset = Dataset.from_tensor_slices((val_data, val_label)).shuffle(shuffle[1], reshuffle_each_iteration=False).batch(batch)
    model.compile(optimizer="adam", loss="categorical_crossentropy", metrics=["accuracy"])
    print(model.summary())
    model.fit(train_dataset, epochs=epochs, validation_data=val_dataset, verbose=2,
              callbacks=[EarlyStopping(monitor="val_loss", patience=patience, restore_best_weights=False), *callbacks,
                         ModelCheckpoint(checkpoint, save_best_only=True, verbose=1)])
    model = load_model(checkpoint, **({"max_batch": max_batch} if max_batch else {}))
    print(model.summary())
    y = np.argmax(model.predict(test_data), axis=1)
    results = model.evaluate(test_data, test_label)
    print(f"Test loss: {results[0]} / Test accuracy: {results[1]}")
    return model, y, results


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=10000)
    ap.add_argument("--data", default=None, help="folder with the six .npy files (train_google_dataset.py:15-24); default: synthetic clips")
    ap.add_argument("--small", action="store_true", help="synthetic 2048/512/512 split")
    ap.add_argument("--checkpoint", default="bin/models/baselineV2.h5")
    args = ap.parse_args(argv)
    splits = load_processed_dataset(args.data) if args.data else synthetic_dataset(*((2048, 512, 512) if args.small else ()))
    model, y, results = run(get_model(), splits, 10, batch=256, epochs=args.epochs, patience=200, checkpoint=args.checkpoint)
    conf_matrix = confusion_matrix(splits[2][1], y, 10)
    print(conf_matrix)
    return model, results, conf_matrix


if __name__ == "__main__":
    main()
