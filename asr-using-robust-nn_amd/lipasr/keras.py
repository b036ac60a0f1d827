"""Host-side mirror of the small Keras surface the reference's training driver uses
(train_constraints.py:2-10, 37-42, 63-111; train_google_dataset.py:49-99), over liblipasr.

    inp = Input((880,)); hdn = Dense(1024, activation='relu', kernel_constraint=NonNeg())(inp)
    hdn = BatchNormalization()(hdn); hdn = Dropout(0.1)(hdn); ...; out = Dense(10, activation='softmax', ...)(hdn)
    model = Model(inputs=inp, outputs=out)
    model.compile(optimizer='adam', loss=CategoricalCrossentropy(), metrics=['accuracy'])
    model.fit(train_dataset, epochs=..., validation_data=val_dataset, verbose=2, callbacks=[...])

Only the topology the reference builds is supported: a chain of Dense(relu) [-> BatchNormalization]
[-> Dropout] blocks ending in Dense(softmax).  Every arithmetic step (GEMMs, BatchNorm, dropout,
loss, Adam, NonNeg) runs in liblipasr's HIP kernels on flat device buffers; this module only owns the
protocol: layer objects with ``name`` / ``get_weights()`` / ``set_weights()`` (the duck-typed
interface Constraints.py:18-33 and extract_features_construct_dataset.py:176-183 rely on), the
callback loop, datasets and checkpoints.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import time

import numpy as np
import torch

from . import _native as N
from . import keras_h5

_name_counts = {}


def _auto_name(prefix):
    k = _name_counts.get(prefix, 0)
    _name_counts[prefix] = k + 1
    return prefix if k == 0 else f"{prefix}_{k}"


def reset_layer_names():
    """Restart Keras-style auto-naming (dense, dense_1, ...) -- a fresh 'process' for tests."""
    _name_counts.clear()


def to_categorical(labels, num_classes=None):
    """tensorflow.keras.utils.to_categorical (train_constraints.py:20)."""
    labels = np.asarray(labels).astype(np.int64).ravel()
    if num_classes is None:
        num_classes = int(labels.max()) + 1
    y = np.zeros((labels.shape[0], num_classes), dtype=np.float32)
    y[np.arange(labels.shape[0]), labels] = 1.0
    return y


class NonNeg:
    """tensorflow.keras.constraints.NonNeg: w * [w >= 0], applied inside the fused Adam kernel (K5)."""

    def __call__(self, w):
        return w * (w >= 0)

    def get_config(self):
        return {}


class CategoricalCrossentropy:
    """Marker for the only loss the path uses (train_constraints.py:94); computed from logits in softmax_ce_kernel."""

    name = "categorical_crossentropy"


class _Node:
    def __init__(self, layer, parent, width):
        self.layer, self.parent, self.width = layer, parent, width


class Layer:
    def __init__(self, name=None, prefix="layer"):
        self.name = name or _auto_name(prefix)
        self._model = None
        self._index = -1  # block index inside the plan

    def __call__(self, node):
        return _Node(self, node, self._out_width(node.width))

    def _out_width(self, w):
        return w

    def get_weights(self):
        return []

    def set_weights(self, weights):
        if len(weights):
            raise ValueError(f"layer {self.name} has no weights")

    def __repr__(self):
        return f"<lipasr.keras.{type(self).__name__} name={self.name}>"


class InputLayer(Layer):
    def __init__(self, shape, name=None):
        super().__init__(name, "input")
        self.shape = tuple(shape)


def Input(shape, name=None):
    if isinstance(shape, int):
        shape = (shape,)
    if len(shape) != 1:
        raise NotImplementedError("lipasr supports flat feature vectors only (Input((880,)))")
    layer = InputLayer(shape, name)
    return _Node(layer, None, int(shape[0]))


class Dense(Layer):
    def __init__(self, units, activation=None, kernel_constraint=None, name=None):
        super().__init__(name, "dense")
        if activation not in ("relu", "softmax"):
            raise NotImplementedError("Dense activation must be 'relu' (hidden) or 'softmax' (output)")
        if kernel_constraint is not None and not isinstance(kernel_constraint, NonNeg):
            raise NotImplementedError(
                "kernel_constraint must be NonNeg() or None; customConstraint is applied with "
                "lipasr.Constraints.customConstraint as a callback-style projection"
            )
        self.units, self.activation, self.kernel_constraint = int(units), activation, kernel_constraint

    def _out_width(self, w):
        return self.units

    @property
    def kernel(self):
        return self._model._seg(self._index, N.SEG_W).view(self._model._widths[self._index], self.units)

    @property
    def bias(self):
        return self._model._seg(self._index, N.SEG_B)

    def get_weights(self):
        """[W (in, out), b] as host NumPy copies -- the caller owns them (Constraints.py:30-31)."""
        return [self.kernel.detach().cpu().numpy().copy(), self.bias.detach().cpu().numpy().copy()]

    def set_weights(self, weights):
        w, b = weights
        k = self.kernel
        k.copy_(torch.as_tensor(np.asarray(w, dtype=np.float32)).reshape(k.shape))
        self.bias.copy_(torch.as_tensor(np.asarray(b, dtype=np.float32)))


class BatchNormalization(Layer):
    def __init__(self, name=None):
        super().__init__(name, "batch_normalization")

    def _tensors(self):
        m, i = self._model, self._index
        return [m._seg(i, N.SEG_GAMMA), m._seg(i, N.SEG_BETA), m._seg(i, N.SEG_MMEAN, state=True), m._seg(i, N.SEG_MVAR, state=True)]

    def get_weights(self):
        """[gamma, beta, moving_mean, moving_variance] (extract_features_construct_dataset.py:182-183)."""
        return [t.detach().cpu().numpy().copy() for t in self._tensors()]

    def set_weights(self, weights):
        for t, w in zip(self._tensors(), weights):
            t.copy_(torch.as_tensor(np.asarray(w, dtype=np.float32)))


class Dropout(Layer):
    def __init__(self, rate, name=None):
        super().__init__(name, "dropout")
        self.rate = float(rate)


# ------------------------------------------------------------------------------------------------
class Dataset:
    """The slice of tf.data the reference uses: from_tensor_slices(...).shuffle(buffer,
    reshuffle_each_iteration=False).batch(n) (train_constraints.py:37-42)."""

    def __init__(self, x, y, batch_size=None, order=None):
        self.x, self.y, self.batch_size, self.order = x, y, batch_size, order

    @staticmethod
    def from_tensor_slices(tensors):
        x, y = tensors
        return Dataset(np.asarray(x) if not torch.is_tensor(x) else x, np.asarray(y) if not torch.is_tensor(y) else y)

    def shuffle(self, buffer_size, reshuffle_each_iteration=False, seed=None):
        """Sliding shuffle buffer; one fixed order (the reference passes reshuffle_each_iteration=False).
        TensorFlow's RNG stream cannot be matched; the structure (element i moves forward by less than
        `buffer_size`) is."""
        if reshuffle_each_iteration:
            raise NotImplementedError("reshuffle_each_iteration=True is not used by the reference")
        n = len(self.x)
        rng = np.random.default_rng(seed)
        buf = list(range(min(buffer_size, n)))
        nxt = len(buf)
        order = []
        while buf:
            j = int(rng.integers(0, len(buf)))
            order.append(buf[j])
            if nxt < n:
                buf[j] = nxt
                nxt += 1
            else:
                buf[j] = buf[-1]
                buf.pop()
        return Dataset(self.x, self.y, self.batch_size, np.asarray(order, dtype=np.int64))

    def batch(self, batch_size):
        return Dataset(self.x, self.y, int(batch_size), self.order)

    def __len__(self):
        n = len(self.x)
        return n if not self.batch_size else (n + self.batch_size - 1) // self.batch_size

    def materialize(self, device):
        """(x, y) float32 device tensors in iteration order."""
        x = torch.as_tensor(self.x).to(device=device, dtype=torch.float32)
        y = torch.as_tensor(self.y).to(device=device, dtype=torch.float32)
        if self.order is not None:
            idx = torch.as_tensor(self.order, device=device)
            x, y = x.index_select(0, idx).contiguous(), y.index_select(0, idx).contiguous()
        return x.contiguous(), y.contiguous()


# ------------------------------------------------------------------------------------------------
class Callback:
    """tensorflow.keras.callbacks.Callback protocol (the hooks the reference's callbacks implement)."""

    def __init__(self):
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_train_begin(self, logs=None):
        pass

    def on_train_end(self, logs=None):
        pass

    def on_epoch_begin(self, epoch, logs=None):
        pass

    def on_epoch_end(self, epoch, logs=None):
        pass

    def on_batch_end(self, batch, logs=None):
        pass


class EarlyStopping(Callback):
    def __init__(self, monitor="val_loss", patience=0, restore_best_weights=False):
        super().__init__()
        self.monitor, self.patience, self.restore_best_weights = monitor, patience, restore_best_weights
        self.best, self.wait, self.best_state = math.inf, 0, None

    def on_train_begin(self, logs=None):
        self.best, self.wait, self.best_state = math.inf, 0, None

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if cur is None:
            return
        if cur < self.best:
            self.best, self.wait = cur, 0
            if self.restore_best_weights:
                self.best_state = self.model._state_dict()
        else:
            self.wait += 1
            if self.wait >= self.patience:
                self.model.stop_training = True
                if self.restore_best_weights and self.best_state is not None:
                    self.model._load_state_dict(self.best_state)


class ModelCheckpoint(Callback):
    def __init__(self, filepath, monitor="val_loss", save_best_only=False, verbose=0):
        super().__init__()
        self.filepath, self.monitor, self.save_best_only, self.verbose = filepath, monitor, save_best_only, verbose
        self.best = math.inf

    def on_epoch_end(self, epoch, logs=None):
        cur = (logs or {}).get(self.monitor)
        if self.save_best_only:
            if cur is None or not cur < self.best:
                return
            if self.verbose:
                print(f"Epoch {epoch + 1}: {self.monitor} improved from {self.best:.5f} to {cur:.5f}, saving model to {self.filepath}")
            self.best = cur
        self.model.save(self.filepath)


class TensorBoard(Callback):
    """tensorflow.keras.callbacks.TensorBoard(log_dir=) as train_constraints.py:45-48,98 uses it: a passive logger of
    the per-epoch scalars.  TensorBoard's event files are TensorFlow protobuf records (not writable without
    tensorflow/tensorboard); the same scalars go to ``<log_dir>/scalars.jsonl``, one JSON object per epoch."""

    def __init__(self, log_dir="logs", **_ignored):
        super().__init__()
        self.log_dir = str(log_dir)
        self._fh = None

    def on_train_begin(self, logs=None):
        import os

        os.makedirs(self.log_dir, exist_ok=True)
        self._fh = open(os.path.join(self.log_dir, "scalars.jsonl"), "a")

    def on_epoch_end(self, epoch, logs=None):
        import json

        if self._fh is not None:
            self._fh.write(json.dumps({"epoch": int(epoch), **{k: float(v) for k, v in (logs or {}).items()}}) + "\n")
            self._fh.flush()

    def on_train_end(self, logs=None):
        if self._fh is not None:
            self._fh.close()
            self._fh = None


# ------------------------------------------------------------------------------------------------
class Model:
    def __init__(self, inputs, outputs, device=None, seed=0, max_batch=1024, compute_dtype=None):
        """compute_dtype: "float32" (exact fp32 MFMA chains), "float16x2" (round 5: every GEMM operand split into two fp16 planes on the
        fp16 matrix instruction, three of the four cross terms, fp32 accumulate: 22-bit operands, 2^-21 per product -- the
        resampler's and the STFT's arithmetic; parameters, activations, BatchNorm, loss, Adam and the projections stay fp32) or
        "bfloat16" (GEMM operands rounded to bf16: BASELINE config 2's arithmetic, NOT inside the 1e-3 logit bound)."""
        self._compute_from_env = compute_dtype is None
        if compute_dtype is None:  # (LIPASR_COMPUTE: sweep knob for the test suite and A/B runs)
            import os as _os

            compute_dtype = _os.environ.get("LIPASR_COMPUTE", "float32")
        if compute_dtype not in ("float32", "bfloat16", "float16x2"):
            raise ValueError("compute_dtype must be 'float32', 'float16x2' or 'bfloat16'")
        self._compute_dtype = compute_dtype
        chain = []
        node = outputs
        while node is not None:
            chain.append(node.layer)
            node = node.parent
        chain.reverse()
        if not isinstance(chain[0], InputLayer) or chain[0] is not inputs.layer:
            raise ValueError("outputs is not connected to inputs")
        self.layers = chain
        self.stop_training = False
        self._seed = seed
        self._max_batch = int(max_batch)
        self._parse(chain)
        self._compiled = False
        self._dp = None  # lipasr.parallel.DataParallel, optional
        self._dropout_seed = 0x5EED0000 + seed
        self._replica_rank = 0  # data parallel: folded into the dropout key, so that every rank draws its own masks
        self._device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._build()

    # ---- structure
    def _parse(self, chain):
        widths = [chain[0].shape[0]]
        blocks = []  # dict(dense=, bn=, drop=)
        for layer in chain[1:]:
            if isinstance(layer, Dense):
                blocks.append({"dense": layer, "bn": None, "drop": None})
                widths.append(layer.units)
            elif isinstance(layer, BatchNormalization):
                if not blocks or blocks[-1]["bn"] is not None or blocks[-1]["drop"] is not None:
                    raise NotImplementedError("BatchNormalization must directly follow a Dense layer")
                blocks[-1]["bn"] = layer
            elif isinstance(layer, Dropout):
                if not blocks or blocks[-1]["drop"] is not None:
                    raise NotImplementedError("Dropout must follow Dense or BatchNormalization")
                blocks[-1]["drop"] = layer
            else:
                raise NotImplementedError(f"unsupported layer {layer!r}")
        if not blocks:
            raise ValueError("model has no Dense layer")
        for i, b in enumerate(blocks):
            last = i == len(blocks) - 1
            want = "softmax" if last else "relu"
            if b["dense"].activation != want:
                raise NotImplementedError(f"Dense {i} must use activation={want!r}")
            if last and (b["bn"] or b["drop"]):
                raise NotImplementedError("nothing may follow the softmax Dense layer")
        if len(blocks) > N.MAX_LAYERS:
            raise NotImplementedError(f"at most {N.MAX_LAYERS} Dense layers")
        self._blocks, self._widths = blocks, widths

    def _build(self):
        self._h = N.get_handle(self._device.index)
        nb = len(self._blocks)
        bn = [1 if b["bn"] else 0 for b in self._blocks]
        drop = [b["drop"].rate if b["drop"] else 0.0 for b in self._blocks]
        nonneg = [1 if b["dense"].kernel_constraint is not None else 0 for b in self._blocks]
        plan = N.c_h()
        N.check(N.lib.lipasr_mlp_create(self._h.h, nb, N.int_array(self._widths), N.int_array(bn), N.float_array(drop),
                                        N.int_array(nonneg), self._max_batch, C.byref(plan)))
        self._plan = plan
        N.register_owner(self)
        if self._compute_dtype == "bfloat16":
            N.check(N.lib.lipasr_mlp_set_compute(plan, 1))
        elif self._compute_dtype == "float16x2":
            rc = N.lib.lipasr_mlp_set_compute(plan, 2)
            if rc == N.EUNSUPPORTED and self._compute_from_env:
                self._compute_dtype = "float32"  # (the environment's default does not apply to a model without BatchNorm)
            else:
                N.check(rc)
        import os as _os

        if _os.environ.get("LIPASR_FUSE_BN", "1") == "0":  # A/B knob: BatchNorm as launches of its own (the round-4 chain)
            N.check(N.lib.lipasr_mlp_set_fuse_bn(plan, 0))
        if _os.environ.get("LIPASR_GEMM_TILES"):  # A/B knob: 64 x 64 tiles from which a GEMM takes the LDS-tiled / ring kernels (default 224; the pipeline sets 128 on a CU share)
            N.check(N.lib.lipasr_mlp_set_gemm_tiles(plan, int(_os.environ["LIPASR_GEMM_TILES"])))
        n_params, n_state = N.sz(), N.sz()
        N.check(N.lib.lipasr_mlp_sizes(plan, C.byref(n_params), C.byref(n_state)))
        dev = self._device
        self._params = torch.zeros(n_params.value, device=dev)
        self._grads = torch.zeros(n_params.value, device=dev)
        self._adam_m = torch.zeros(n_params.value, device=dev)
        self._adam_v = torch.zeros(n_params.value, device=dev)
        self._bnstate = torch.zeros(max(1, n_state.value), device=dev)
        self._step = torch.zeros(1, dtype=torch.int32, device=dev)
        self._segs = {}
        for i in range(nb):
            for kind in range(6):
                off, cnt = N.sz(), N.sz()
                N.check(N.lib.lipasr_mlp_segment(plan, i, kind, C.byref(off), C.byref(cnt)))
                self._segs[(i, kind)] = (off.value, cnt.value)
        for i, b in enumerate(self._blocks):
            for key in ("dense", "bn", "drop"):
                if b[key] is not None:
                    b[key]._model, b[key]._index = self, i
        self._n_classes = self._widths[-1]
        self._loss_rows = torch.zeros(self._max_batch, device=dev)
        self._correct_rows = torch.zeros(self._max_batch, device=dev)
        self._init_weights()

    def exchange_errors(self):
        """Non-zero if a fused BatchNorm exchange (lipasr_mlp_set_fuse_bn) gave up since the last call; synchronises."""
        e = C.c_int(0)
        if getattr(self, "_plan", None):
            N.check(N.lib.lipasr_mlp_exchange_errors(self._plan, C.byref(e)))
        return e.value

    def close(self):
        """Frees the native classifier plan (its workspace).  Idempotent; the tensors stay readable."""
        plan, self._plan = getattr(self, "_plan", None), None
        if plan and self._h.alive:
            N.destroy_or_defer(N.lib.lipasr_mlp_destroy, plan)  # (a finaliser may run in the middle of a graph capture)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _seg(self, i, kind, state=False):
        off, cnt = self._segs[(i, kind)]
        buf = self._bnstate if state or kind in (N.SEG_MMEAN, N.SEG_MVAR) else self._params
        return buf[off:off + cnt]

    def _init_weights(self):
        """Keras defaults: glorot_uniform kernels, zero biases, gamma 1, beta 0, moving mean 0, moving var 1."""
        g = torch.Generator(device="cpu").manual_seed(self._seed)
        for i, b in enumerate(self._blocks):
            n_in, n_out = self._widths[i], self._widths[i + 1]
            lim = math.sqrt(6.0 / (n_in + n_out))
            w = (torch.rand(n_in, n_out, generator=g, dtype=torch.float64) * 2 - 1) * lim
            self._seg(i, N.SEG_W).copy_(w.reshape(-1).to(torch.float32))
            if b["bn"] is not None:
                self._seg(i, N.SEG_GAMMA).fill_(1.0)
                self._seg(i, N.SEG_MVAR).fill_(1.0)

    # ---- Keras surface
    def compile(self, optimizer="adam", loss=None, metrics=None, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        if optimizer != "adam":
            raise NotImplementedError("only optimizer='adam' (the reference's choice) is implemented")
        if loss is not None and not (isinstance(loss, CategoricalCrossentropy) or loss == "categorical_crossentropy"):
            raise NotImplementedError("only categorical cross-entropy is implemented")
        self._adam = (float(learning_rate), float(beta_1), float(beta_2), float(epsilon))
        self._metrics = list(metrics or [])
        self._compiled = True

    def summary(self):
        lines = ["Model: lipasr dense classifier", "_" * 60]
        total = 0
        for layer in self.layers:
            n = sum(int(np.prod(w.shape)) for w in layer.get_weights()) if not isinstance(layer, InputLayer) else 0
            total += n
            lines.append(f"{layer.name:<28}{type(layer).__name__:<22}{n:>10}")
        lines += ["=" * 60, f"Total params: {total}"]
        return "\n".join(lines)

    @property
    def grads(self):
        """Flat fp32 gradient buffer (the data-parallel all-reduce buffer)."""
        return self._grads

    def dense_kernels(self):
        """Device views of the Dense kernels in layer order (what the native constraints project)."""
        return [b["dense"].kernel for b in self._blocks]

    def _dropout_cfg(self, masks=None, enable=True):
        cfg = N.DropoutCfg()
        if masks is not None:
            cfg.mode = 2
            self._mask_ptrs = N.ptr_array([m.data_ptr() if m is not None else None for m in masks])
            cfg.masks = C.cast(self._mask_ptrs, N.PV)
        elif enable and any(b["drop"] for b in self._blocks):
            cfg.mode = 1
            # Philox key = seed (+ rank): the counter is (local element, layer, step), which is the same on every rank
            cfg.seed = (self._dropout_seed + 0x9E3779B97F4A7C15 * self._replica_rank) & 0xFFFFFFFFFFFFFFFF
            cfg.step_dev = self._step.data_ptr()
        else:
            cfg.mode = 0
        return cfg

    def train_fwd_bwd(self, xb, yb, inv_batch=None, masks=None, dropout=True, probs=None, defer_dw0=False):
        """Enqueue forward + loss + backward for one batch (device tensors); grads land in self._grads.
        defer_dw0 (data parallel): everything except the first layer's [dW | db] -- ``self._grads[self.late_floats:]`` is
        final when this call's kernels are, ``train_dw0(xb)`` then fills ``self._grads[:self.late_floats]``."""
        bsz = xb.shape[0]
        cfg = self._dropout_cfg(masks, dropout)
        inv = 1.0 / bsz if inv_batch is None else inv_batch
        fn = N.lib.lipasr_mlp_train_fwd_bwd_head if defer_dw0 else N.lib.lipasr_mlp_train_fwd_bwd
        N.check(fn(self._plan, N.ptr(self._params), N.ptr(self._bnstate), N.ptr(xb), N.ptr(yb), bsz, inv,
                   C.byref(cfg), N.ptr(self._grads), N.ptr(self._loss_rows), N.ptr(self._correct_rows),
                   N.ptr(probs), N.stream_ptr()))

    def syncbn_segments(self):
        """Number of segments the synchronized-BatchNorm step is cut into (allocates the partial-sum buffer on first use)."""
        if getattr(self, "_part", None) is None:
            n = N.sz()
            N.check(N.lib.lipasr_mlp_part_floats(self._plan, C.byref(n)))
            self._part = torch.zeros(int(n.value), device=self._device)
            ns = C.c_int()
            N.check(N.lib.lipasr_mlp_train_segments(self._plan, C.byref(ns)))
            self._n_segments = ns.value
        return self._n_segments

    def syncbn_segment(self, seg, xb, yb, global_batch, world, masks=None, dropout=True, probs=None):
        """Enqueue segment ``seg`` of the training step (it ends right after a GEMM whose epilogue left BatchNorm column sums,
        reduced to [2][width] in self._part); returns the number of floats of self._part to SUM-all-reduce before the next
        segment (0 after the last one).  The count does not depend on this rank's row count: uneven shards are fine."""
        bsz = xb.shape[0]
        cfg = self._dropout_cfg(masks, dropout)
        self.syncbn_segments()
        N.check(N.lib.lipasr_mlp_train_segment(self._plan, seg, N.ptr(self._params), N.ptr(self._bnstate), N.ptr(xb), N.ptr(yb), bsz,
                                               1.0 / float(global_batch), C.byref(cfg), N.ptr(self._grads), N.ptr(self._loss_rows),
                                               N.ptr(self._correct_rows), N.ptr(probs), N.ptr(self._part), int(global_batch),
                                               1.0 / float(world), N.stream_ptr()))
        n = N.sz()
        N.check(N.lib.lipasr_mlp_train_segment_exchange(self._plan, bsz, seg, C.byref(n)))
        return int(n.value)

    def train_fwd_bwd_syncbn(self, xb, yb, dp, global_batch, masks=None, dropout=True, probs=None):
        """Synchronized BatchNorm (opt-in): the same kernels as train_fwd_bwd, run segment by segment with a SUM all-reduce
        of the BatchNorm column sums between segments (``dp``: lipasr.parallel.DataParallel), so that every rank
        normalises with the statistics of the GLOBAL batch, as the single-device reference does: 2 x (BatchNorm layers)
        small collectives per step (2 x width floats each, whatever the ranks' row counts).  TrainPipeline replays every
        segment as a HIP graph."""
        for seg in range(self.syncbn_segments()):
            n = self.syncbn_segment(seg, xb, yb, global_batch, dp.world, masks=masks, dropout=dropout, probs=probs)
            if n:
                dp.allreduce_grads(self._part[:n])

    def train_dw0(self, xb):
        N.check(N.lib.lipasr_mlp_train_dw0(self._plan, N.ptr(xb), xb.shape[0], N.ptr(self._grads), N.stream_ptr()))

    @property
    def late_floats(self):
        """Length of the leading part of the flat gradient ([dW_0 | db_0]) that ``train_dw0`` produces."""
        n = N.sz()
        N.check(N.lib.lipasr_mlp_grad_split(self._plan, C.byref(n)))
        return int(n.value)

    def apply_adam(self, grad_scale=1.0):
        lr, b1, b2, eps = self._adam
        N.check(N.lib.lipasr_mlp_adam_nonneg(self._plan, N.ptr(self._params), N.ptr(self._grads), N.ptr(self._adam_m), N.ptr(self._adam_v),
                                             N.ptr(self._step), lr, b1, b2, eps, grad_scale, N.stream_ptr()))

    def apply_adam_project_product(self, rho, order, norms_out, grad_scale=1.0):
        """Adam + NonNeg + simple_norm_constraint (``order``: N.int_array of layer visits) as one native call."""
        lr, b1, b2, eps = self._adam
        N.check(N.lib.lipasr_mlp_adam_project_product(self._plan, N.ptr(self._params), N.ptr(self._grads), N.ptr(self._adam_m),
                                                      N.ptr(self._adam_v), N.ptr(self._step), lr, b1, b2, eps, grad_scale, float(rho), order,
                                                      len(order), N.ptr(norms_out), N.stream_ptr()))

    def train_on_batch(self, xb, yb, masks=None, dropout=True):
        """fwd, bwd, (data-parallel gradient all-reduce), Adam + NonNeg; returns nothing (stream-ordered)."""
        if self._dp is not None:
            self._dp.train_step(self, xb, yb, masks=masks, dropout=dropout)
            return
        self.train_fwd_bwd(xb, yb, masks=masks, dropout=dropout)
        self.apply_adam()

    def fit(self, x, y=None, epochs=1, batch_size=32, validation_data=None, verbose=1, callbacks=None):
        if not self._compiled:
            raise RuntimeError("call compile() before fit()")
        ds = x if isinstance(x, Dataset) else Dataset(x, y, batch_size)
        if ds.batch_size is None:
            ds = ds.batch(batch_size)
        if ds.batch_size > self._max_batch:
            raise ValueError(f"batch {ds.batch_size} exceeds the plan's max_batch {self._max_batch}")
        xs, ys = ds.materialize(self._device)
        val = None
        if validation_data is not None:
            vd = validation_data if isinstance(validation_data, Dataset) else Dataset(validation_data[0], validation_data[1], ds.batch_size)
            val = vd.materialize(self._device)
        callbacks = list(callbacks or [])
        for cb in callbacks:
            cb.set_model(self) if hasattr(cb, "set_model") else setattr(cb, "model", self)
        for cb in callbacks:
            cb.on_train_begin()
        self.stop_training = False
        history = {"loss": [], "accuracy": []}
        n, bs = xs.shape[0], ds.batch_size
        for epoch in range(epochs):
            t0 = time.time()
            for cb in callbacks:
                cb.on_epoch_begin(epoch)
            loss_sum = torch.zeros((), device=self._device)
            acc_sum = torch.zeros((), device=self._device)
            for bi, s in enumerate(range(0, n, bs)):
                xb, yb = xs[s:s + bs], ys[s:s + bs]
                self.train_on_batch(xb, yb)
                k = xb.shape[0]
                loss_sum += self._loss_rows[:k].sum()
                acc_sum += self._correct_rows[:k].sum()
                for cb in callbacks:
                    cb.on_batch_end(bi)
            logs = {"loss": float(loss_sum) / n, "accuracy": float(acc_sum) / n}
            if val is not None:
                vl, va = self._evaluate_device(val[0], val[1], bs)
                logs["val_loss"], logs["val_accuracy"] = vl, va
            for k, v in logs.items():
                history.setdefault(k, []).append(v)
            if verbose:
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.2f}s - {msg}")
            for cb in callbacks:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in callbacks:
            cb.on_train_end()
        self.history = history
        return history

    def predict_device(self, x, batch_size=None, logits=False):
        """Inference on a float32 device tensor; returns a device tensor [n, classes]."""
        bs = min(self._max_batch, batch_size or self._max_batch)
        out = torch.empty(x.shape[0], self._n_classes, device=self._device)
        for s in range(0, x.shape[0], bs):
            xb = x[s:s + bs]
            ob = out[s:s + bs]
            N.check(N.lib.lipasr_mlp_predict(self._plan, N.ptr(self._params), N.ptr(self._bnstate), N.ptr(xb), xb.shape[0],
                                             None if logits else N.ptr(ob), N.ptr(ob) if logits else None, N.stream_ptr()))
        return out

    def predict(self, x, batch_size=32, verbose=0):
        xt = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x).to(device=self._device, dtype=torch.float32).contiguous()
        return self.predict_device(xt).cpu().numpy()

    def _evaluate_device(self, x, y, bs):
        p = self.predict_device(x, bs, logits=True)
        logp = torch.log_softmax(p.double(), dim=1)  # metric read-out only (plumbing), not on the training path
        loss = float(-(y.double() * logp).sum(dim=1).mean())
        acc = float((p.argmax(dim=1) == y.argmax(dim=1)).double().mean())
        return loss, acc

    def evaluate(self, x, y, batch_size=32, verbose=0):
        xt = torch.as_tensor(np.asarray(x)).to(device=self._device, dtype=torch.float32).contiguous()
        yt = torch.as_tensor(np.asarray(y)).to(device=self._device, dtype=torch.float32).contiguous()
        loss, acc = self._evaluate_device(xt, yt, self._max_batch)
        return [loss, acc]

    # ---- checkpoints: Keras' HDF5 layout behind .h5 / .hdf5 names (lipasr.keras_h5), a torch archive otherwise
    def _config(self):
        cfg = []
        for layer in self.layers:
            if isinstance(layer, InputLayer):
                cfg.append(("Input", layer.shape[0], layer.name))
            elif isinstance(layer, Dense):
                cfg.append(("Dense", layer.units, layer.activation, layer.kernel_constraint is not None, layer.name))
            elif isinstance(layer, BatchNormalization):
                cfg.append(("BatchNormalization", layer.name))
            elif isinstance(layer, Dropout):
                cfg.append(("Dropout", layer.rate, layer.name))
        return cfg

    def _state_dict(self):
        return {"params": self._params.clone(), "bnstate": self._bnstate.clone(), "adam_m": self._adam_m.clone(),
                "adam_v": self._adam_v.clone(), "step": self._step.clone()}

    def _load_state_dict(self, sd):
        self._params.copy_(sd["params"]); self._bnstate.copy_(sd["bnstate"])
        self._adam_m.copy_(sd["adam_m"]); self._adam_v.copy_(sd["adam_v"]); self._step.copy_(sd["step"])

    def _named_weights(self):
        """[(layer name, [(Keras weight name, host array)])] for every layer, in model order."""
        out = []
        for layer in self.layers:
            kind = type(layer).__name__
            out.append((layer.name, list(zip(keras_h5.weight_names(kind, layer.name), layer.get_weights()))))
        return out

    def _trainable_segments(self):
        """(Keras variable path, offset, count, shape) of the trainable variables in Keras' order (kernel, bias,
        gamma, beta per block): the order tf.keras' Adam creates its m and v slots in."""
        segs = []
        for i, b in enumerate(self._blocks):
            d = b["dense"]
            segs.append((f"{d.name}/kernel", *self._segs[(i, N.SEG_W)], (self._widths[i], d.units)))
            segs.append((f"{d.name}/bias", *self._segs[(i, N.SEG_B)], (d.units,)))
            if b["bn"] is not None:
                segs.append((f"{b['bn'].name}/gamma", *self._segs[(i, N.SEG_GAMMA)], (d.units,)))
                segs.append((f"{b['bn'].name}/beta", *self._segs[(i, N.SEG_BETA)], (d.units,)))
        return segs

    def _optimizer_weights(self):
        """tf.keras Adam.weights: iterations, then the m slot of every variable, then the v slots."""
        m, v = self._adam_m.cpu().numpy(), self._adam_v.cpu().numpy()
        out = [("Adam/iter:0", np.asarray(int(self._step.item()), dtype=np.int64))]
        for slot, buf in (("m", m), ("v", v)):
            for name, off, cnt, shape in self._trainable_segments():
                out.append((f"Adam/{name}/{slot}:0", buf[off:off + cnt].reshape(shape).copy()))
        return out

    def _load_optimizer_weights(self, ow):
        """Slots are matched by name ('Adam/<layer>/<var>/m:0'); a file without them leaves the moments at zero."""
        if not ow:
            return
        it = [a for n, a in ow.items() if n.split("/")[-1].startswith("iter")]
        if it:
            self._step.fill_(int(np.asarray(it[0]).reshape(-1)[0]))
        for name, off, cnt, _ in self._trainable_segments():
            for slot, buf in (("m", self._adam_m), ("v", self._adam_v)):
                hit = [a for n, a in ow.items() if n.endswith(f"/{name}/{slot}:0") or n == f"{name}/{slot}:0"]
                if hit:
                    buf[off:off + cnt].copy_(torch.as_tensor(np.asarray(hit[0], dtype=np.float32).reshape(-1)))

    def save(self, path, include_optimizer=True):
        """Model.save / ModelCheckpoint target (train_constraints.py:104-105): '.h5' / '.hdf5' names get Keras' HDF5
        layout (model_config, model_weights, training_config, optimizer_weights); other names a torch archive."""
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        if keras_h5.is_hdf5_path(path):
            adam = getattr(self, "_adam", None)
            keras_h5.save_model(path, self._config(), self._named_weights(),
                                train_cfg=keras_h5.training_config(*adam) if adam else None,
                                optimizer_weights=self._optimizer_weights() if (adam and include_optimizer) else None)
            return
        sd = {k: v.cpu() for k, v in self._state_dict().items()}
        torch.save({"config": self._config(), "state": sd, "adam": getattr(self, "_adam", None), "max_batch": self._max_batch}, path)

    def get_weights(self):
        out = []
        for layer in self.layers:
            out += layer.get_weights()
        return out

    def set_weights(self, weights):
        """Keras Model.set_weights: the flat list get_weights() returns, layer by layer."""
        weights = list(weights)
        i = 0
        for layer in self.layers:
            n = len(layer.get_weights())
            if n:
                layer.set_weights(weights[i:i + n])
                i += n
        if i != len(weights):
            raise ValueError(f"set_weights: got {len(weights)} arrays, the model holds {i}")

    def _set_named_weights(self, layers, by_name=False):
        """layers: [(layer name, [(weight name, array)])] as read from a file.  Keras' default is topological loading:
        the file's weighted layers are matched, in order, with the model's weighted layers."""
        mine = [l for l in self.layers if l.get_weights()]
        theirs = [(n, w) for n, w in layers if w]
        if by_name:
            lookup = dict(theirs)
            for layer in mine:
                if layer.name in lookup:
                    layer.set_weights([a for _, a in lookup[layer.name]])
            return
        if len(mine) != len(theirs):
            raise ValueError(f"the file holds weights for {len(theirs)} layers, the model has {len(mine)} layers with weights")
        for layer, (lname, ws) in zip(mine, theirs):
            want = [w.shape for w in layer.get_weights()]
            got = [np.asarray(a).shape for _, a in ws]
            if want != got:
                raise ValueError(f"layer {layer.name} <- file layer {lname}: weight shapes {got} do not match {want}")
            layer.set_weights([a for _, a in ws])

    def save_weights(self, path):
        """Weights only (train_constraints.py:96's commented ``load_weights`` counterpart): the trainable and
        BatchNorm state, no optimizer moments.  '.h5' / '.hdf5' names get Keras' HDF5 weights layout."""
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        if keras_h5.is_hdf5_path(path):
            keras_h5.save_weights(path, self._named_weights())
            return
        torch.save({"config": self._config(), "weights": [np.asarray(w) for w in self.get_weights()]}, path)

    def load_weights(self, path, by_name=False):
        if _is_hdf5_file(path):
            self._set_named_weights(keras_h5.load_weights(path), by_name=by_name)
            return
        blob = torch.load(path, map_location="cpu", weights_only=False)
        if "weights" in blob:
            self.set_weights(blob["weights"])
        else:  # a full Model.save archive: take the parameters, leave this model's optimizer state alone
            sd = blob["state"]
            self._params.copy_(sd["params"].to(self._device))
            self._bnstate.copy_(sd["bnstate"].to(self._device))


def _is_hdf5_file(path):
    """By content, not by name: the 8-byte HDF5 signature."""
    with open(path, "rb") as fh:
        return fh.read(8) == b"\x89HDF\r\n\x1a\n"


def model_from_config(cfg, **kw):
    node = None
    inp = None
    for item in cfg:
        kind = item[0]
        if kind == "Input":
            inp = node = Input((item[1],), name=item[2])
        elif kind == "Dense":
            node = Dense(item[1], activation=item[2], kernel_constraint=NonNeg() if item[3] else None, name=item[4])(node)
        elif kind == "BatchNormalization":
            node = BatchNormalization(name=item[1])(node)
        elif kind == "Dropout":
            node = Dropout(item[1], name=item[2])(node)
    return Model(inputs=inp, outputs=node, **kw)


def load_model(path, custom_objects=None, compile=True, **kw):
    """tensorflow.keras.models.load_model (train_constraints.py:107, attacks.py:315-317): Keras' own ``.h5`` files and
    the archives Model.save writes.  ``custom_objects`` is accepted for signature parity; NonNeg is the one
    constraint class a checkpoint of the path carries."""
    if _is_hdf5_file(path):
        blob = keras_h5.load_model(path)
        m = model_from_config(blob["chain"], **kw)
        m._set_named_weights(blob["layers"])
        if compile and blob["adam"]:
            lr, b1, b2, eps = blob["adam"]
            m.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"], learning_rate=lr, beta_1=b1,
                      beta_2=b2, epsilon=eps)
            m._load_optimizer_weights(blob["optimizer_weights"])
        return m
    blob = torch.load(path, map_location="cpu", weights_only=False)
    kw.setdefault("max_batch", blob.get("max_batch", 1024))
    m = model_from_config(blob["config"], **kw)
    m._load_state_dict({k: v.to(m._device) for k, v in blob["state"].items()})
    if blob.get("adam"):
        lr, b1, b2, eps = blob["adam"]
        m.compile(optimizer="adam", loss=CategoricalCrossentropy(), learning_rate=lr, beta_1=b1, beta_2=b2, epsilon=eps)
    return m
