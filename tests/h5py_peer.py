"""Independent HDF5 peer for tests/test_hdf5_cpu.py, run in a SEPARATE interpreter that has h5py (the image's
/opt/conda/bin/python3.9; lipasr itself talks to libhdf5 through ctypes and never imports h5py).

    h5py_peer.py read  <file.h5> <out.npz>   walk a full-model file the way Keras' loader does
                                             (attrs['model_config'], model_weights.attrs['layer_names'],
                                             group.attrs['weight_names'], np.asarray(group[name])) and dump it
    h5py_peer.py write <file.h5> <in.npz>    write a file the way tf.keras 2.x + h5py 2.x did: JSON attributes as
                                             utf-8 *bytes*, name lists as lists of bytes, [] for weight-less layers
"""
import json
import sys

import h5py
import numpy as np


def _s(x):
    return x.decode("utf-8") if hasattr(x, "decode") else str(x)


def read(path, out):
    res = {}
    with h5py.File(path, "r") as f:
        res["model_config"] = _s(f.attrs["model_config"])
        res["training_config"] = _s(f.attrs["training_config"]) if "training_config" in f.attrs else ""
        res["keras_version"] = _s(f.attrs["keras_version"])
        res["backend"] = _s(f.attrs["backend"])
        g = f["model_weights"]
        layer_names = [_s(n) for n in g.attrs["layer_names"]]
        res["layer_names"] = json.dumps(layer_names)
        wn_all = {}
        for ln in layer_names:
            names = [_s(n) for n in g[ln].attrs["weight_names"]]
            wn_all[ln] = names
            for n in names:
                res["w:" + ln + ":" + n] = np.asarray(g[ln][n])
        res["weight_names"] = json.dumps(wn_all)
        if "optimizer_weights" in f:
            og = f["optimizer_weights"]
            on = [_s(n) for n in og.attrs["weight_names"]]
            res["optimizer_weight_names"] = json.dumps(on)
            for n in on:
                res["o:" + n] = np.asarray(og[n])
    np.savez(out, **res)


def write(path, src):
    d = np.load(src, allow_pickle=False)
    spec = json.loads(str(d["spec"]))
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = spec["keras_version"].encode("utf8")
        f.attrs["backend"] = b"tensorflow"
        f.attrs["model_config"] = json.dumps(spec["model_config"]).encode("utf8")
        if spec.get("training_config"):
            f.attrs["training_config"] = json.dumps(spec["training_config"]).encode("utf8")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = [n.encode("utf8") for n in spec["layer_names"]]
        g.attrs["backend"] = b"tensorflow"
        g.attrs["keras_version"] = spec["keras_version"].encode("utf8")
        for ln in spec["layer_names"]:
            lg = g.create_group(ln)
            names = spec["weight_names"][ln]
            lg.attrs["weight_names"] = [n.encode("utf8") for n in names]
            for n in names:
                lg.create_dataset(n, data=d["w:" + ln + ":" + n])
        if spec.get("optimizer_weight_names"):
            og = f.create_group("optimizer_weights")
            og.attrs["weight_names"] = [n.encode("utf8") for n in spec["optimizer_weight_names"]]
            for n in spec["optimizer_weight_names"]:
                og.create_dataset(n, data=d["o:" + n])


if __name__ == "__main__":
    {"read": read, "write": write}[sys.argv[1]](sys.argv[2], sys.argv[3])
