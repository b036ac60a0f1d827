"""Keras' HDF5 checkpoint layout (``model.save('x.h5')`` / ``ModelCheckpoint('...h5')`` / ``load_model`` /
``save_weights`` / ``load_weights``) for the classifier of the path: train_constraints.py:104-107,
train_google_dataset.py:85-87, attacks.py:315-317 write and read exactly these files.

Layout (tf.keras 2.x ``save_format='h5'``; restated from Keras' documented file structure, and read back in the
tests with h5py the way Keras' own loader walks it):

    /                      attrs  keras_version, backend, model_config (JSON), training_config (JSON, if compiled)
    /model_weights         attrs  layer_names [S], backend, keras_version
    /model_weights/<layer> attrs  weight_names [S]   (weight-less layers: an empty array)
    /model_weights/<layer>/<layer>/kernel:0 ...      one float32 dataset per weight, named by weight_names
    /optimizer_weights     attrs  weight_names [S];  datasets Adam/iter:0 (int64), Adam/<layer>/<var>/m:0, .../v:0

A weights-only file (``save_weights``) holds the contents of ``/model_weights`` at the root.  This module is plain host
code over NumPy arrays (the Model methods in lipasr.keras move tensors to and from the device around it).
"""
import json

import numpy as np

from . import _hdf5 as H

KERAS_VERSION = "2.4.0"
BACKEND = "tensorflow"
BN_MOMENTUM, BN_EPSILON = 0.99, 1e-3  # the values liblipasr's BatchNorm kernels carry (Keras defaults)

_ZEROS = {"class_name": "Zeros", "config": {}}
_ONES = {"class_name": "Ones", "config": {}}


def is_hdf5_path(path):
    return str(path).lower().endswith((".h5", ".hdf5", ".keras.h5"))


# ------------------------------------------------------------------------------------------------
# model_config <-> the layer chain [("Input", width, name), ("Dense", units, activation, nonneg, name), ...]
# ------------------------------------------------------------------------------------------------
def model_config(chain, name="model"):
    """The Functional-API ``model_config`` JSON object Keras writes for the chain (train_constraints.py:63-88)."""
    layers, prev = [], None
    for item in chain:
        kind, lname = item[0], item[-1]
        if kind == "Input":
            cfg = {"batch_input_shape": [None, int(item[1])], "dtype": "float32", "sparse": False, "ragged": False, "name": lname}
            cls = "InputLayer"
        elif kind == "Dense":
            cfg = {"name": lname, "trainable": True, "dtype": "float32", "units": int(item[1]), "activation": item[2],
                   "use_bias": True, "kernel_initializer": {"class_name": "GlorotUniform", "config": {"seed": None}},
                   "bias_initializer": _ZEROS, "kernel_regularizer": None, "bias_regularizer": None, "activity_regularizer": None,
                   "kernel_constraint": {"class_name": "NonNeg", "config": {}} if item[3] else None, "bias_constraint": None}
            cls = "Dense"
        elif kind == "BatchNormalization":
            cfg = {"name": lname, "trainable": True, "dtype": "float32", "axis": [1], "momentum": BN_MOMENTUM, "epsilon": BN_EPSILON,
                   "center": True, "scale": True, "beta_initializer": _ZEROS, "gamma_initializer": _ONES,
                   "moving_mean_initializer": _ZEROS, "moving_variance_initializer": _ONES, "beta_regularizer": None,
                   "gamma_regularizer": None, "beta_constraint": None, "gamma_constraint": None}
            cls = "BatchNormalization"
        elif kind == "Dropout":
            cfg = {"name": lname, "trainable": True, "dtype": "float32", "rate": float(item[1]), "noise_shape": None, "seed": None}
            cls = "Dropout"
        else:
            raise ValueError(f"unknown layer kind {kind!r}")
        layers.append({"class_name": cls, "config": cfg, "name": lname,
                       "inbound_nodes": [] if prev is None else [[[prev, 0, 0, {}]]]})
        prev = lname
    return {"class_name": "Functional",
            "config": {"name": name, "layers": layers, "input_layers": [[layers[0]["name"], 0, 0]],
                       "output_layers": [[layers[-1]["name"], 0, 0]]},
            "keras_version": KERAS_VERSION, "backend": BACKEND}


def _constraint_flag(c, lname):
    if c is None:
        return False
    cls = c.get("class_name") if isinstance(c, dict) else str(c)
    if cls in ("NonNeg", "non_neg"):
        return True
    raise NotImplementedError(f"layer {lname}: kernel_constraint {cls!r} is not a checkpointable constraint of this path "
                              "(NonNeg or none; customConstraint is applied as a projection after loading)")


def chain_from_config(cfg):
    """Inverse of model_config for files Keras wrote: Functional ('Model' in older files) or Sequential models made of
    InputLayer / Dense / BatchNormalization / Dropout."""
    if isinstance(cfg, (str, bytes)):
        cfg = json.loads(cfg)
    cls, body = cfg.get("class_name"), cfg.get("config")
    if cls not in ("Functional", "Model", "Sequential"):
        raise NotImplementedError(f"model class {cls!r} is not supported")
    layers = body if isinstance(body, list) else body["layers"]  # Keras < 2.2.3 stored a bare list for Sequential
    chain = []
    for i, layer in enumerate(layers):
        k, c = layer["class_name"], layer["config"]
        lname = c.get("name", layer.get("name"))
        if i == 0 and k != "InputLayer":  # Sequential without an explicit Input: the first layer carries the shape
            shape = c.get("batch_input_shape") or c.get("batch_shape")
            if not shape:
                raise ValueError("the first layer has no batch_input_shape")
            chain.append(("Input", int(shape[-1]), f"{lname}_input"))
        if k == "InputLayer":
            shape = c.get("batch_input_shape") or c.get("batch_shape")  # (Keras 3's legacy-h5 writer says batch_shape)
            if not shape:
                raise ValueError(f"InputLayer {lname} has no batch_input_shape")
            if len(shape) != 2:
                raise NotImplementedError(f"input shape {shape} (flat feature vectors only)")
            chain.append(("Input", int(shape[1]), lname))
        elif k == "Dense":
            if not c.get("use_bias", True):
                raise NotImplementedError(f"layer {lname}: use_bias=False")
            chain.append(("Dense", int(c["units"]), c["activation"], _constraint_flag(c.get("kernel_constraint"), lname), lname))
        elif k == "BatchNormalization":
            if abs(c.get("momentum", BN_MOMENTUM) - BN_MOMENTUM) > 1e-12 or abs(c.get("epsilon", BN_EPSILON) - BN_EPSILON) > 1e-12:
                raise NotImplementedError(f"layer {lname}: BatchNormalization momentum/epsilon other than "
                                          f"{BN_MOMENTUM}/{BN_EPSILON} (the kernels' constants)")
            if not (c.get("center", True) and c.get("scale", True)):
                raise NotImplementedError(f"layer {lname}: center/scale=False")
            chain.append(("BatchNormalization", lname))
        elif k == "Dropout":
            chain.append(("Dropout", float(c["rate"]), lname))
        else:
            raise NotImplementedError(f"layer class {k!r} is outside the path (Dense / BatchNormalization / Dropout chains)")
    return chain


def training_config(lr, beta_1, beta_2, epsilon):
    """model.compile(optimizer='adam', loss=CategoricalCrossentropy(), metrics=['accuracy']) (train_constraints.py:94)."""
    return {"loss": {"class_name": "CategoricalCrossentropy",
                     "config": {"reduction": "auto", "name": "categorical_crossentropy", "from_logits": False, "label_smoothing": 0}},
            "metrics": [[{"class_name": "MeanMetricWrapper", "config": {"name": "accuracy", "dtype": "float32", "fn": "categorical_accuracy"}}]],
            "weighted_metrics": None, "loss_weights": None,
            "optimizer_config": {"class_name": "Adam",
                                 "config": {"name": "Adam", "learning_rate": float(lr), "decay": 0.0, "beta_1": float(beta_1),
                                            "beta_2": float(beta_2), "epsilon": float(epsilon), "amsgrad": False}}}


def adam_from_training_config(tc):
    """(lr, beta_1, beta_2, epsilon) or None."""
    if not tc:
        return None
    if isinstance(tc, (str, bytes)):
        tc = json.loads(tc)
    oc = tc.get("optimizer_config") or {}
    if oc.get("class_name") != "Adam":
        raise NotImplementedError(f"optimizer {oc.get('class_name')!r} (the path trains with Adam)")
    c = oc.get("config", {})
    if c.get("amsgrad"):
        raise NotImplementedError("Adam(amsgrad=True)")
    return (float(c.get("learning_rate", c.get("lr", 1e-3))), float(c.get("beta_1", 0.9)), float(c.get("beta_2", 0.999)),
            float(c.get("epsilon", 1e-7)))


# ------------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------------
_DENSE_VARS, _BN_VARS = ("kernel", "bias"), ("gamma", "beta", "moving_mean", "moving_variance")


def weight_names(kind, lname):
    if kind == "Dense":
        return [f"{lname}/{v}:0" for v in _DENSE_VARS]
    if kind == "BatchNormalization":
        return [f"{lname}/{v}:0" for v in _BN_VARS]
    return []


def _write_weight_group(f, root, layers):
    """layers: [(layer_name, [(weight_name, array), ...]), ...] in model order."""
    f.create_group(root or "/")
    f.write_attr(root or "/", "layer_names", [lname.encode("utf-8") for lname, _ in layers])
    f.write_attr(root or "/", "backend", BACKEND)
    f.write_attr(root or "/", "keras_version", KERAS_VERSION)
    for lname, weights in layers:
        g = f"{root}/{lname}" if root else lname
        f.create_group(g)
        f.write_attr(g, "weight_names", [w.encode("utf-8") for w, _ in weights])
        for wname, arr in weights:
            f.write_dataset(f"{g}/{wname}", np.asarray(arr, dtype=np.float32))


def _read_weight_group(f, root):
    out = []
    for lname in f.read_attr(root or "/", "layer_names"):
        g = f"{root}/{lname}" if root else lname
        names = f.read_attr(g, "weight_names")
        names = [] if isinstance(names, np.ndarray) else list(names)  # weight-less layers: an empty numeric array
        out.append((lname, [(w, f.read_dataset(f"{g}/{w}")) for w in names]))
    return out


def save_model(path, chain, layers, train_cfg=None, optimizer_weights=None, name="model"):
    """Full-model file.  optimizer_weights: [(name, array)] with 'Adam/iter:0' first (int64), or None."""
    with H.File(path, "w") as f:
        f.write_attr("/", "keras_version", KERAS_VERSION)
        f.write_attr("/", "backend", BACKEND)
        f.write_attr("/", "model_config", json.dumps(model_config(chain, name)))
        if train_cfg is not None:
            f.write_attr("/", "training_config", json.dumps(train_cfg))
        _write_weight_group(f, "model_weights", layers)
        if optimizer_weights:
            f.create_group("optimizer_weights")
            f.write_attr("optimizer_weights", "weight_names", [n.encode("utf-8") for n, _ in optimizer_weights])
            for n, arr in optimizer_weights:
                f.write_dataset(f"optimizer_weights/{n}", arr)


def load_model(path):
    """-> dict(chain, layers [(layer, [(weight_name, array)])], adam (lr, b1, b2, eps) | None,
    optimizer_weights {name: array})."""
    with H.File(path, "r") as f:
        if not f.has_attr("/", "model_config"):
            raise ValueError(f"{path}: no model_config attribute (a weights-only file? use Model.load_weights)")
        out = {"chain": chain_from_config(f.read_attr("/", "model_config")),
               "layers": _read_weight_group(f, "model_weights"),
               "adam": adam_from_training_config(f.read_attr("/", "training_config")) if f.has_attr("/", "training_config") else None,
               "optimizer_weights": {}}
        if f.exists("optimizer_weights") and f.has_attr("optimizer_weights", "weight_names"):
            names = f.read_attr("optimizer_weights", "weight_names")
            if not isinstance(names, np.ndarray):
                out["optimizer_weights"] = {n: f.read_dataset(f"optimizer_weights/{n}") for n in names}
        return out


def save_weights(path, layers):
    with H.File(path, "w") as f:
        _write_weight_group(f, "", layers)


def load_weights(path):
    """Weights of a weights-only file, or of the model_weights group of a full-model file."""
    with H.File(path, "r") as f:
        root = "model_weights" if (not f.has_attr("/", "layer_names") and f.exists("model_weights")) else ""
        return _read_weight_group(f, root)
