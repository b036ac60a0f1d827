"""attacks.py surface of the reference (attacks.py:48-86, 145-294, 496-536, 647-693) over liblipasr.

White-box: ``TensorFlowV2Classifier`` / ``FastGradientMethod`` / ``ProjectedGradientDescent`` keep
the ART constructor keywords the reference uses (``estimator=``, ``eps=``) plus the ART defaults it
relies on (norm=inf, eps_step=0.1, max_iter=100, batch_size=32, untargeted, y=None -> the model's own
predictions, no clip_values).  Each PGD iteration is ONE native call: inference forward, CE gradient,
backward to the input and the sign step fused into the last backward GEMM's epilogue (K4).

Black-box: ``standardize_dataset`` (A2, fp64-accumulated fit on the device), the audio-domain noise
models on the device (Philox RNG) and the noisy-audio -> MFCC dataset helpers.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _native as N
from .extract_features_construct_dataset import read_wav, _extractor
from .keras import Model, to_categorical


def _dev():
    return torch.device("cuda", torch.cuda.current_device())


def _to_dev(x):
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    return t.to(device=_dev(), dtype=torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------ A2
class StandardScaler:
    """sklearn.preprocessing.StandardScaler restricted to fit / transform / fit_transform, on the device.
    Statistics are fp64 (mean_, scale_ are float64 device tensors, as sklearn's are float64 arrays)."""

    def fit(self, x):
        xt = _to_dev(x)
        h = N.get_handle(xt.device.index)
        self.mean_ = torch.zeros(xt.shape[1], dtype=torch.float64, device=xt.device)
        self.scale_ = torch.zeros(xt.shape[1], dtype=torch.float64, device=xt.device)
        N.check(N.lib.lipasr_scaler_fit(h.h, N.ptr(xt), xt.shape[0], xt.shape[1], N.ptr(self.mean_), N.ptr(self.scale_), N.stream_ptr()))
        return self

    def transform_device(self, xt):
        h = N.get_handle(xt.device.index)
        out = torch.empty_like(xt)
        N.check(N.lib.lipasr_scaler_apply(h.h, N.ptr(xt), xt.shape[0], xt.shape[1], N.ptr(self.mean_), N.ptr(self.scale_), N.ptr(out), N.stream_ptr()))
        return out

    def transform(self, x):
        out = self.transform_device(_to_dev(x))
        return out if torch.is_tensor(x) else out.cpu().numpy()

    def fit_transform(self, x):
        return self.fit(x).transform(x)


def standardize_dataset(train_data, val_data, test_data):
    """attacks.py:48-69 / train_constraints.py:28-35: fit on the concatenation, split back."""
    parts = [_to_dev(train_data), _to_dev(val_data), _to_dev(test_data)]
    all_data = torch.cat(parts, dim=0)
    out = StandardScaler().fit(all_data).transform_device(all_data)
    a, b = parts[0].shape[0], parts[0].shape[0] + parts[1].shape[0]
    res = (out[:a], out[a:b], out[b:])
    if torch.is_tensor(train_data):
        return res
    return tuple(r.cpu().numpy() for r in res)


# ------------------------------------------------------------------------------------------------ A9 / A10
class TensorFlowV2Classifier:
    """ART estimator wrapper (attacks.py:500-504): ``predict`` and ``loss_gradient`` over a lipasr Model."""

    def __init__(self, model, nb_classes, input_shape, loss_object=None, clip_values=None):
        if not isinstance(model, Model):
            raise TypeError("model must be a lipasr.keras.Model")
        if clip_values is not None:
            raise NotImplementedError("the reference passes no clip_values")
        self.model, self.nb_classes, self.input_shape = model, int(nb_classes), tuple(input_shape)
        if model._n_classes != self.nb_classes or model._widths[0] != self.input_shape[0]:
            raise ValueError("nb_classes / input_shape do not match the model")

    def predict(self, x, batch_size=128):
        return self.model.predict(x)

    def loss_gradient(self, x, y):
        """d mean CE(f(x), y) / dx in inference mode, NumPy in / NumPy out."""
        m = self.model
        xt, yt = _to_dev(x), _to_dev(y)
        out = torch.empty_like(xt)
        bs = m._max_batch
        for s in range(0, xt.shape[0], bs):
            xb, yb, ob = xt[s:s + bs], yt[s:s + bs], out[s:s + bs]
            N.check(N.lib.lipasr_mlp_input_grad(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xb), N.ptr(yb), xb.shape[0], N.ptr(ob), N.stream_ptr()))
        return out.cpu().numpy()

    def output_vjp_device(self, xt, vt, on_logits=False, probs_out=None):
        """sum_c v[b, c] d out_c / dx on device tensors ([B, features], [B, classes]) -> [B, features]."""
        m = self.model
        out = torch.empty_like(xt)
        bs = m._max_batch
        for s in range(0, xt.shape[0], bs):
            xb, vb, ob = xt[s:s + bs], vt[s:s + bs], out[s:s + bs]
            pb = None if probs_out is None else probs_out[s:s + bs]
            N.check(N.lib.lipasr_mlp_output_vjp(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xb), N.ptr(vb), 1 if on_logits else 0,
                                                xb.shape[0], N.ptr(pb), N.ptr(ob), N.stream_ptr()))
        return out

    def class_gradient(self, x, label=None):
        """ART class_gradient: gradients of the model OUTPUT (softmax probabilities) w.r.t. x.
        label None -> [B, nb_classes, features]; int or int array [B] -> [B, 1, features]."""
        xt = _to_dev(x)
        b = xt.shape[0]
        if label is None:
            cols = []
            for c in range(self.nb_classes):
                v = torch.zeros(b, self.nb_classes, device=xt.device)
                v[:, c] = 1.0
                cols.append(self.output_vjp_device(xt, v))
            return torch.stack(cols, dim=1).cpu().numpy()
        lab = torch.as_tensor(np.broadcast_to(np.asarray(label), (b,)).astype(np.int64), device=xt.device)
        v = torch.zeros(b, self.nb_classes, device=xt.device)
        v[torch.arange(b, device=xt.device), lab] = 1.0
        return self.output_vjp_device(xt, v)[:, None, :].cpu().numpy()


def random_targets(labels, nb_classes, rng=None):
    """ART utils.random_targets: for every sample a uniformly drawn class different from ``labels`` -> one-hot."""
    rng = np.random if rng is None else rng
    labels = np.asarray(labels)
    if labels.ndim > 1:
        labels = labels.argmax(axis=1)
    result = np.zeros(labels.shape, dtype=np.int64)
    for c in range(nb_classes):
        other = [k for k in range(nb_classes) if k != c]
        sel = labels == c
        result[sel] = rng.choice(other, size=int(sel.sum()))
    return to_categorical(result, nb_classes)


class SaliencyMapMethod:
    """ART SaliencyMapMethod(classifier=, theta=, gamma=) (JSMA; attacks.py:546-550 uses theta=10, gamma=0.1) without
    clip_values, as the reference runs it: while a sample's prediction differs from its target and at most ``gamma``
    of its features were touched, add ``theta`` to the two features with the largest target-class gradient.
    The class gradients and predictions run natively (lipasr_mlp_output_vjp / lipasr_mlp_predict).

    ``max_iter`` bounds the loop: with no clip values ART's search space never shrinks, so a sample that never
    reaches its target would loop forever there; None keeps ART's behaviour."""

    def __init__(self, classifier, theta=0.1, gamma=1.0, batch_size=1, verbose=True, max_iter=None):
        if not isinstance(classifier, TensorFlowV2Classifier):
            raise TypeError("classifier must be a lipasr TensorFlowV2Classifier")
        if not 0 < gamma <= 1:
            raise ValueError("The total perturbation percentage `gamma` must be between 0 and 1.")
        if batch_size <= 0:
            raise ValueError("The batch size `batch_size` has to be positive.")
        self.estimator, self.theta, self.gamma = classifier, float(theta), float(gamma)
        self.batch_size, self.max_iter = int(batch_size), max_iter

    def generate(self, x, y=None, rng=None):
        est = self.estimator
        m = est.model
        xt = _to_dev(x)
        adv = xt.clone()
        nf = adv.shape[1]
        preds = m.predict_device(xt).argmax(dim=1)
        if y is None:
            targets = torch.as_tensor(random_targets(preds.cpu().numpy(), est.nb_classes, rng).argmax(axis=1), device=xt.device)
        else:
            targets = torch.as_tensor(np.asarray(y.cpu() if torch.is_tensor(y) else y).argmax(axis=1), device=xt.device)
        bs = min(self.batch_size, m._max_batch)
        for s0 in range(0, adv.shape[0], bs):
            batch = adv[s0:s0 + bs]
            tgt = targets[s0:s0 + bs]
            active = torch.nonzero(preds[s0:s0 + bs] != tgt)[:, 0]
            all_feat = torch.zeros_like(batch)
            it = 0
            while active.numel() != 0 and (self.max_iter is None or it < self.max_iter):
                v = torch.zeros(active.numel(), est.nb_classes, device=xt.device)
                v[torch.arange(active.numel(), device=xt.device), tgt[active]] = 1.0
                g = est.output_vjp_device(batch[active].contiguous(), v)
                ind = torch.topk(g if self.theta > 0 else -g, 2, dim=1).indices
                rows = active[:, None].expand(-1, 2)
                all_feat[rows, ind] = 1.0
                batch[rows, ind] += self.theta
                cur = m.predict_device(batch.contiguous()).argmax(dim=1)
                active = torch.nonzero((cur != tgt) & (all_feat.sum(dim=1) / nf <= self.gamma))[:, 0]
                it += 1
        return adv if torch.is_tensor(x) else adv.cpu().numpy().astype(np.asarray(x).dtype, copy=False)


_TANH_SMOOTHER = 0.999999
_C_UPPER_BOUND = 10e10


def _to_tanh(x, lo, hi):
    return torch.atanh((torch.minimum(torch.maximum(x, lo), hi) - lo) / (hi - lo) * (2 * _TANH_SMOOTHER) - _TANH_SMOOTHER)


def _from_tanh(xt, lo, hi):
    return (torch.tanh(xt) / _TANH_SMOOTHER + 1.0) / 2.0 * (hi - lo) + lo


class _Carlini:
    """Shared pieces of ART's Carlini & Wagner attacks as the reference calls them (attacks.py:571-645): untargeted,
    y=None (labels := the estimator's own predictions), no clip_values.  ART evaluates the margin on the model
    OUTPUT -- softmax probabilities for this Keras model -- so with the reference's confidence >= 1 the success test
    can never hold; that behaviour is kept.  Predictions and the class-gradient difference run natively
    (lipasr_mlp_predict, lipasr_mlp_output_vjp); the line-search bookkeeping is a handful of elementwise tensor
    ops per iteration.  Restated from ART 1.9-1.10's published implementation (ART is absent: parity unpinned)."""

    def __init__(self, classifier, confidence, targeted, learning_rate, max_iter, max_halving, max_doubling, batch_size):
        if not isinstance(classifier, TensorFlowV2Classifier):
            raise TypeError("classifier must be a lipasr TensorFlowV2Classifier")
        if targeted:
            raise NotImplementedError("the reference runs the untargeted attack")
        if max_iter < 0 or max_halving < 1 or max_doubling < 1 or batch_size < 1:
            raise ValueError("max_iter >= 0, max_halving >= 1, max_doubling >= 1 and batch_size >= 1 are required")
        self.estimator, self.confidence, self.learning_rate = classifier, float(confidence), float(learning_rate)
        self.max_iter, self.max_halving, self.max_doubling, self.batch_size = int(max_iter), int(max_halving), int(max_doubling), int(batch_size)

    def _predict(self, xa):
        return self.estimator.model.predict_device(xa.contiguous())

    def _margin(self, z, target):
        z_target = (z * target).sum(dim=1)
        z_other = (z * (1 - target) + (z.min(dim=1).values - 1)[:, None] * target).max(dim=1).values
        return torch.clamp(z_target - z_other + self.confidence, min=0.0)

    def _grad_diff(self, z, target, xa):
        other = (z * (1 - target) + (z.min(dim=1).values - 1)[:, None] * target).argmax(dim=1)
        v = target.clone()                                   # +1 at the label (i_add) ...
        v[torch.arange(z.shape[0], device=z.device), other] -= 1.0   # ... -1 at the best other class (i_sub)
        return self.estimator.output_vjp_device(xa.contiguous(), v)

    def _labels(self, xt, y):
        if y is not None:
            return _to_dev(y)
        m = self.estimator.model
        yb = torch.empty(xt.shape[0], m._n_classes, device=xt.device)
        bs = m._max_batch
        for s in range(0, xt.shape[0], bs):
            N.check(N.lib.lipasr_mlp_own_labels(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xt[s:s + bs]), xt[s:s + bs].shape[0],
                                                N.ptr(yb[s:s + bs]), N.stream_ptr()))
        return yb

    def _line_search(self, n, active, loss, pert, lr, evaluate):
        """ART's halving / doubling search on the per-sample learning rate; evaluate(sel, step) -> loss of the trial
        points x_tanh[sel] + step * pert_rows.  Returns best_lr (0 where no trial improved the loss)."""
        prev_loss, best_loss = loss.clone(), loss.clone()
        best_lr = torch.zeros(n, device=loss.device)
        halving = torch.zeros(n, device=loss.device)
        idx = torch.nonzero(active)[:, 0]
        for _ in range(self.max_halving):
            do = loss[idx] >= prev_loss[idx]
            if not bool(do.any()):
                break
            sel = idx[do]
            loss[sel] = evaluate(sel, lr[sel], pert[do])
            better = loss < best_loss
            best_lr[better] = lr[better]
            best_loss[better] = loss[better]
            lr[sel] /= 2
            halving[sel] += 1
        lr[idx] *= 2
        for _ in range(self.max_doubling):
            do = (halving[idx] == 1) & (loss[idx] <= best_loss[idx])
            if not bool(do.any()):
                break
            sel = idx[do]
            lr[sel] *= 2
            loss[sel] = evaluate(sel, lr[sel], pert[do])
            better = loss < best_loss
            best_lr[better] = lr[better]
            best_loss[better] = loss[better]
        lr[halving == 1] /= 2
        return best_lr


class CarliniL2Method(_Carlini):
    """ART CarliniL2Method(classifier=, confidence=) (attacks.py:606-616: confidence in linspace(1, 300, 3))."""

    def __init__(self, classifier, confidence=0.0, targeted=False, learning_rate=0.01, binary_search_steps=10, max_iter=10,
                 initial_const=0.01, max_halving=5, max_doubling=5, batch_size=1, verbose=True):
        super().__init__(classifier, confidence, targeted, learning_rate, max_iter, max_halving, max_doubling, batch_size)
        self.binary_search_steps, self.initial_const = int(binary_search_steps), float(initial_const)

    def generate(self, x, y=None):
        xt = _to_dev(x)
        adv = xt.clone()
        lo = torch.tensor(float(xt.min()), device=xt.device)
        hi = torch.tensor(float(xt.max()), device=xt.device)
        yt = self._labels(xt, y)
        bs = min(self.batch_size, self.estimator.model._max_batch)
        for s0 in range(0, xt.shape[0], bs):
            xb, yb = xt[s0:s0 + bs], yt[s0:s0 + bs]
            n = xb.shape[0]
            xb_tanh = _to_tanh(xb, lo, hi)
            c_cur = torch.full((n,), self.initial_const, device=xt.device)
            c_lower = torch.zeros(n, device=xt.device)
            c_double = torch.ones(n, dtype=torch.bool, device=xt.device)
            best_l2 = torch.full((n,), float("inf"), device=xt.device)
            best_adv = xb.clone()

            def loss_fn(sel, xa_sel):
                l2 = ((xb[sel] - xa_sel) ** 2).sum(dim=1)
                z = self._predict(xa_sel)
                return z, l2, c_cur[sel] * self._margin(z, yb[sel]) + l2

            everyone = torch.arange(n, device=xt.device)
            for _bss in range(self.binary_search_steps):
                if not bool((c_cur < _C_UPPER_BOUND).any()):
                    break
                lr = torch.full((n,), self.learning_rate, device=xt.device)
                xa, xa_tanh = xb.clone(), xb_tanh.clone()
                z, l2, loss = loss_fn(everyone, xa)
                success = loss - l2 <= 0
                overall = success.clone()
                for _it in range(self.max_iter):
                    improved = success & (l2 < best_l2)
                    best_l2[improved] = l2[improved]
                    best_adv[improved] = xa[improved]
                    active = (c_cur < _C_UPPER_BOUND) & (lr > 0)
                    if not bool(active.any()):
                        break
                    g = self._grad_diff(z[active], yb[active], xa[active])
                    g = g * c_cur[active][:, None] + 2 * (xa[active] - xb[active])
                    g = g * (hi - lo) * (1 - torch.tanh(xa_tanh[active]) ** 2) / (2 * _TANH_SMOOTHER)
                    pert = -g

                    def evaluate(sel, step, rows):
                        new_x = _from_tanh(xa_tanh[sel] + step[:, None] * rows, lo, hi)
                        _, l2[sel], ls = loss_fn(sel, new_x)
                        return ls

                    best_lr = self._line_search(n, active, loss, pert, lr, evaluate)
                    idx = torch.nonzero(active)[:, 0]
                    upd = best_lr[idx] > 0
                    if bool(upd.any()):
                        sel = idx[upd]
                        xa_tanh[sel] = xa_tanh[sel] + best_lr[sel][:, None] * pert[upd]
                        xa[sel] = _from_tanh(xa_tanh[sel], lo, hi)
                        z[sel], l2[sel], loss[sel] = loss_fn(sel, xa[sel])
                        success = loss - l2 <= 0
                        overall = overall | success
                improved = success & (l2 < best_l2)
                best_l2[improved] = l2[improved]
                best_adv[improved] = xa[improved]
                c_double[overall] = False
                c_old = c_cur.clone()
                c_cur[overall] = c_lower[overall] + (c_cur - c_lower)[overall] / 2
                fail = ~overall
                c_lower[fail] = c_old[fail]
                fd = fail & c_double
                c_cur[fd] = c_cur[fd] * 2
                nd = fail & ~c_double
                c_cur[nd] = c_cur[nd] + (c_cur - c_lower)[nd] / 2
            adv[s0:s0 + bs] = best_adv
        return adv if torch.is_tensor(x) else adv.cpu().numpy().astype(np.asarray(x).dtype, copy=False)


class CarliniLInfMethod(_Carlini):
    """ART CarliniLInfMethod(classifier=, confidence=) (attacks.py:578-582: confidence = 10), eps = 0.3."""

    def __init__(self, classifier, confidence=0.0, targeted=False, learning_rate=0.01, max_iter=10, max_halving=5, max_doubling=5,
                 eps=0.3, batch_size=128, verbose=True):
        super().__init__(classifier, confidence, targeted, learning_rate, max_iter, max_halving, max_doubling, batch_size)
        if eps <= 0:
            raise ValueError("The eps parameter must be strictly positive.")
        self.eps = float(eps)

    def generate(self, x, y=None):
        xt = _to_dev(x)
        adv = xt.clone()
        yt = self._labels(xt, y)
        bs = min(self.batch_size, self.estimator.model._max_batch)
        for s0 in range(0, xt.shape[0], bs):
            xb, yb = xt[s0:s0 + bs], yt[s0:s0 + bs]
            n = xb.shape[0]
            lo, hi = xb - self.eps, xb + self.eps
            xa, xa_tanh = xb.clone(), _to_tanh(xb, lo, hi)
            z = self._predict(xa)
            loss = self._margin(z, yb)
            lr = torch.full((n,), self.learning_rate, device=xt.device)
            for _it in range(self.max_iter):
                active = (loss > 0) & (lr > 0)
                if not bool(active.any()):
                    break
                g = self._grad_diff(z[active], yb[active], xa[active])
                pert = -(g * (hi - lo)[active] * (1 - torch.tanh(xa_tanh[active]) ** 2) / (2 * _TANH_SMOOTHER))

                def evaluate(sel, step, rows):
                    new_x = _from_tanh(xa_tanh[sel] + step[:, None] * rows, lo[sel], hi[sel])
                    return self._margin(self._predict(new_x), yb[sel])

                best_lr = self._line_search(n, active, loss, pert, lr, evaluate)
                idx = torch.nonzero(active)[:, 0]
                upd = best_lr[idx] > 0
                if bool(upd.any()):
                    sel = idx[upd]
                    xa_tanh[sel] = xa_tanh[sel] + best_lr[sel][:, None] * pert[upd]
                    xa[sel] = _from_tanh(xa_tanh[sel], lo[sel], hi[sel])
                z = self._predict(xa)
                loss = self._margin(z, yb)
            adv[s0:s0 + bs] = xa
        return adv if torch.is_tensor(x) else adv.cpu().numpy().astype(np.asarray(x).dtype, copy=False)


class _SignAttack:
    def __init__(self, estimator, eps, eps_step, max_iter, batch_size, norm, targeted, num_random_init):
        if not isinstance(estimator, TensorFlowV2Classifier):
            raise TypeError("estimator must be a lipasr TensorFlowV2Classifier")
        if norm not in (np.inf, "inf", math.inf):
            raise NotImplementedError("only norm=inf (the ART default the reference uses) is implemented")
        if targeted or num_random_init:
            raise NotImplementedError("targeted / random-init variants are not used by the reference")
        self.estimator, self.eps, self.eps_step = estimator, float(eps), float(eps_step)
        self.max_iter, self.batch_size = int(max_iter), int(batch_size)

    def _labels(self, m, xb, y):
        if y is not None:
            return y
        yb = torch.empty(xb.shape[0], m._n_classes, device=xb.device)
        N.check(N.lib.lipasr_mlp_own_labels(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xb), xb.shape[0], N.ptr(yb), N.stream_ptr()))
        return yb

    def generate_device(self, xt, yt=None):
        """x: float32 device tensor; returns a NEW device tensor (the input is left untouched)."""
        m = self.estimator.model
        adv = xt.clone()
        bs = min(self.batch_size, m._max_batch)
        for s in range(0, xt.shape[0], bs):
            x0 = xt[s:s + bs]
            xa = adv[s:s + bs]
            yb = self._labels(m, x0, None if yt is None else yt[s:s + bs])
            self._run(m, xa, x0, yb)
        return adv

    def generate(self, x, y=None):
        xt = _to_dev(x)
        yt = None if y is None else _to_dev(y)
        adv = self.generate_device(xt, yt)
        return adv if torch.is_tensor(x) else adv.cpu().numpy().astype(np.asarray(x).dtype, copy=False)


class FastGradientMethod(_SignAttack):
    """ART FastGradientMethod(estimator=, eps=) (attacks.py:506-510): x + eps * sign(grad), one step, no clipping."""

    def __init__(self, estimator, eps=0.3, batch_size=32, norm=np.inf, targeted=False, num_random_init=0):
        super().__init__(estimator, eps, eps, 1, batch_size, norm, targeted, num_random_init)

    def _run(self, m, xa, x0, yb):
        N.check(N.lib.lipasr_mlp_attack_step(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xa), N.ptr(x0), N.ptr(yb), xa.shape[0],
                                             self.eps, math.inf, N.stream_ptr()))


class ProjectedGradientDescent(_SignAttack):
    """ART ProjectedGradientDescent(estimator=, eps=) (attacks.py:657-661): max_iter steps of
    x <- x0 + clip(x + eps_step * sign(grad) - x0, -eps, eps)."""

    def __init__(self, estimator, eps=0.3, eps_step=0.1, max_iter=100, batch_size=32, norm=np.inf, targeted=False, num_random_init=0):
        super().__init__(estimator, eps, eps_step, max_iter, batch_size, norm, targeted, num_random_init)

    def _run(self, m, xa, x0, yb):
        for _ in range(self.max_iter):
            N.check(N.lib.lipasr_mlp_attack_step(m._plan, N.ptr(m._params), N.ptr(m._bnstate), N.ptr(xa), N.ptr(x0), N.ptr(yb), xa.shape[0],
                                                 self.eps_step, self.eps, N.stream_ptr()))


def sign_step(x_adv, x0, g, alpha, eps):
    """Stand-alone K4 on device tensors, in place on x_adv."""
    h = N.get_handle(x_adv.device.index)
    N.check(N.lib.lipasr_sign_step(h.h, N.ptr(x_adv), N.ptr(x0), N.ptr(g), x_adv.numel(), float(alpha), float(eps), N.stream_ptr()))
    return x_adv


# ------------------------------------------------------------------------------------------------ A12 (device noise)
_noise_calls = [0]


def _noise(arr, mode, p0, p1, seed):
    was_tensor = torch.is_tensor(arr)
    t = _to_dev(arr).clone()
    one_d = t.dim() == 1
    if one_d:
        t = t[None, :]
    if seed is None:
        _noise_calls[0] += 1
        seed = 0xA77AC000 + _noise_calls[0]
    h = N.get_handle(t.device.index)
    N.check(N.lib.lipasr_add_noise_f32(h.h, N.ptr(t), t.shape[0], t.shape[1], mode, float(p0), float(p1), int(seed), N.stream_ptr()))
    if one_d:
        t = t[0]
    return t if was_tensor else t.cpu().numpy()


def add_white_noise(array, sigma, seed=None):
    """attacks.py:73-86: array + N(0, sigma)."""
    return _noise(array, 0, sigma, 0.0, seed)


def add_noise(x, p, alpha, seed=None):
    """attacks.py:166-183 (mixtgauss :145-163): impulse mixture, sigma0 = alpha, sigma1 = 10 alpha, peaks where |N(0,1)| < p."""
    return _noise(x, 1, p, alpha, seed)


def add_white_noise_with_snr(audio, target_snr_db, seed=None):
    """attacks.py:222-245: white noise whose power sits target_snr_db below the clip's mean power."""
    return _noise(audio, 2, target_snr_db, 0.0, seed)


def add_white_noise_on_dataset(dataset, sigma, seed=None):
    """attacks.py:186-201: white noise on every row of an MFCC matrix."""
    return _noise(dataset, 0, sigma, 0.0, seed)


def add_noise_mixture_on_dataset(dataset, p, alpha, seed=None):
    """attacks.py:204-219."""
    return _noise(dataset, 1, p, alpha, seed)


def noisy_audio_to_mfcc(waves, sr_in=16000, sigma=0, p=0, alpha=0, target_snr_db=None, seed=None, utterance_length=44):
    """The one end-to-end audio flow of the reference (attacks.py:89-121, 248-274) for a batch of clips:
    resample -> add noise at 22 050 Hz -> MFCC -> (B, 20*utterance_length) device tensor."""
    w = _to_dev(waves)
    ex = _extractor(int(sr_in), w.shape[1], w.shape[0])
    y = ex.resample(w)
    if target_snr_db is not None:
        y = add_white_noise_with_snr(y, target_snr_db, seed)
    elif sigma != 0:
        y = add_white_noise(y, sigma, seed)
    elif p != 0 and alpha != 0:
        y = add_noise(y, p, alpha, seed)
    return ex.from_22k(y, utterance_length)


def _files_to_batches(filenames):
    groups = {}
    for i, fn in enumerate(filenames):
        x, sr = read_wav(fn)
        groups.setdefault((sr, len(x)), []).append((i, x))
    return groups


def black_box_attack_on_audio_dataset(filenames, sigma, p, alpha, seed=None):
    """attacks.py:124-142: noisy MFCC for a list of wav files, (N, 880) float64."""
    out = np.zeros((len(filenames), 20 * 44))
    for (sr, n), items in _files_to_batches(filenames).items():
        w = np.stack([x for _, x in items])
        f = noisy_audio_to_mfcc(w, sr, sigma=sigma, p=p, alpha=alpha, seed=seed).cpu().numpy()
        for (i, _), row in zip(items, f):
            out[i] = row
    return out


def black_box_attack_on_audio_dataset_snr(filenames, target_snr_db, seed=None):
    """attacks.py:277-294."""
    out = np.zeros((len(filenames), 20 * 44))
    for (sr, n), items in _files_to_batches(filenames).items():
        w = np.stack([x for _, x in items])
        f = noisy_audio_to_mfcc(w, sr, target_snr_db=target_snr_db, seed=seed).cpu().numpy()
        for (i, _), row in zip(items, f):
            out[i] = row
    return out


def black_box_attack_on_audio(file_path, utterance_length, sigma=0, p=0, alpha=0, seed=None):
    """attacks.py:89-121: one file -> noisy MFCC (20, utterance_length), float32 NumPy."""
    x, sr = read_wav(file_path)
    f = noisy_audio_to_mfcc(x[None, :], sr, sigma=sigma, p=p, alpha=alpha, seed=seed, utterance_length=utterance_length)
    return f.view(20, utterance_length).cpu().numpy()


def black_box_attack_on_audio_snr(file_path, utterance_length, target_snr_db, seed=None):
    """attacks.py:248-274."""
    x, sr = read_wav(file_path)
    f = noisy_audio_to_mfcc(x[None, :], sr, target_snr_db=target_snr_db, seed=seed, utterance_length=utterance_length)
    return f.view(20, utterance_length).cpu().numpy()


def mixtgauss(N_, p, sigma0, sigma1, seed=None):
    """attacks.py:145-163: N_ samples of the impulse mixture (sigma1 where |N(0,1)| < p, sigma0 elsewhere), drawn by the
    device generator -- add_noise's noise term on its own."""
    if abs(sigma1 - 10 * sigma0) > 1e-12 * max(1.0, abs(sigma1)):
        raise NotImplementedError("mixtgauss with sigma1 != 10 sigma0 (the reference's only call, attacks.py:178-180, uses 10x)")
    z = torch.zeros(int(N_), device=_dev())
    h = N.get_handle(z.device.index)
    if seed is None:
        _noise_calls[0] += 1
        seed = 0xA77AC000 + _noise_calls[0]
    N.check(N.lib.lipasr_add_noise_f32(h.h, N.ptr(z), 1, z.numel(), 1, float(p), float(sigma0), int(seed), N.stream_ptr()))
    return z.cpu().numpy()


def load_npy_dataset(path):
    """attacks.py:27-45: the six ``.npy`` files of a processed dataset folder (``path`` ends with a separator, as in the
    reference's call sites)."""
    import os

    def ld(name):
        return np.load(os.path.join(path, name) if os.path.isdir(path) else path + name)

    return (ld("train_data.npy"), ld("train_label.npy"), ld("dev_data.npy"), ld("dev_label.npy"), ld("test_data.npy"),
            ld("test_label.npy"))
