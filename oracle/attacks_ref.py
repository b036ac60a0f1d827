"""Oracle A9, A10, A12: FGSM / PGD (ART defaults) and the audio-domain noise models.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: adversarial-robustness-toolbox
(era 1.9-1.10) is absent; update rules restated from its published algorithm.  Call sites:

  VD/attacks.py:500-510  TensorFlowV2Classifier + FastGradientMethod(estimator, eps).generate(x)
  VD/attacks.py:650-661  ProjectedGradientDescent(estimator, eps).generate(x)
  VD/attacks.py:73-86, 145-183, 222-245  noise models (reference uses the unseeded global NumPy RNG;
  here the generator is passed in).
"""
from __future__ import annotations

import numpy as np
from . import mlp_ref


def _own_labels(spec, p, x, batch_size):
    """ART: y=None -> labels := one-hot argmax of the estimator's own prediction."""
    preds = np.concatenate([mlp_ref.forward_infer(spec, p, x[i:i + batch_size]) for i in range(0, len(x), batch_size)])
    y = np.zeros_like(preds)
    y[np.arange(len(preds)), preds.argmax(axis=1)] = 1
    return y


def fgsm(spec, p, x, eps, batch_size=32, y=None):
    """FastGradientMethod(norm=inf, targeted=False, minimal=False, num_random_init=0): x + eps*sign(g); no clipping."""
    x = np.asarray(x)
    y = _own_labels(spec, p, x, batch_size) if y is None else y
    adv = x.copy()
    for i in range(0, len(x), batch_size):
        g = mlp_ref.input_gradient_infer(spec, p, x[i:i + batch_size], y[i:i + batch_size])
        g = np.where(np.isnan(g), 0.0, g)
        adv[i:i + batch_size] = x[i:i + batch_size] + eps * np.sign(g)
    return adv


def pgd(spec, p, x, eps, eps_step=0.1, max_iter=100, batch_size=32, y=None):
    """ProjectedGradientDescent(norm=inf, num_random_init=0): x <- x0 + clip(x + eps_step*sign(g) - x0, +-eps)."""
    x = np.asarray(x)
    y = _own_labels(spec, p, x, batch_size) if y is None else y
    adv = x.copy()
    for i in range(0, len(x), batch_size):
        x0 = x[i:i + batch_size]
        xa = x0.copy()
        yb = y[i:i + batch_size]
        for _ in range(max_iter):
            g = mlp_ref.input_gradient_infer(spec, p, xa, yb)
            g = np.where(np.isnan(g), 0.0, g)
            xa = xa + eps_step * np.sign(g)
            xa = x0 + np.clip(xa - x0, -eps, eps)
        adv[i:i + batch_size] = xa
    return adv


def sign_step(x_adv, x0, g, alpha, eps):
    """One fused PGD update (K4): x0 + clip(x_adv + alpha*sign(g) - x0, -eps, eps); eps=inf gives FGSM."""
    xa = x_adv + alpha * np.sign(g)
    return x0 + np.clip(xa - x0, -eps, eps)


# ------------------------------------------------------------------ (f)-4: class gradients, JSMA
def class_gradient(spec, p, x, labels):
    """ART TensorFlowV2Classifier.class_gradient(x, label=labels)[:, 0, :]: gradient of the OUTPUT (softmax
    probability) of class labels[b] w.r.t. x_b.  VD/attacks.py:538-645 are the call sites of the attacks using it."""
    labels = np.broadcast_to(np.asarray(labels), (len(x),))
    v = np.zeros((len(x), spec[-1].n_out))
    v[np.arange(len(x)), labels] = 1.0
    return mlp_ref.output_vjp_infer(spec, p, x, v, on_logits=False)[0]


def jsma(spec, p, x, targets, theta=0.1, gamma=1.0, batch_size=1, max_iter=None):
    """SaliencyMapMethod(classifier, theta, gamma).generate(x) (VD/attacks.py:546-550: theta=10, gamma=0.1), restated
    from ART 1.9-1.10's published algorithm WITHOUT clip_values (the reference passes none): while the prediction
    differs from the target and at most gamma of the features were touched, add theta to the two features with the
    largest target-class gradient (smallest when theta < 0).  With no clip values the search space never shrinks,
    so ART can loop forever on a sample that never reaches its target; ``max_iter`` (None = ART's behaviour) bounds
    it.  ``targets`` (int array) replaces ART's unseeded random_targets draw."""
    x = np.asarray(x)
    adv = x.astype(np.float64 if x.dtype == np.float64 else np.float32).copy()
    nf = adv.shape[1]
    preds = np.concatenate([mlp_ref.forward_infer(spec, p, x[i:i + batch_size]) for i in range(0, len(x), batch_size)]).argmax(1)
    targets = np.asarray(targets)
    for s0 in range(0, len(adv), batch_size):
        batch = adv[s0:s0 + batch_size]
        tgt = targets[s0:s0 + batch_size]
        cur = preds[s0:s0 + batch_size]
        active = np.where(cur != tgt)[0]
        all_feat = np.zeros_like(batch)
        it = 0
        while active.size != 0 and (max_iter is None or it < max_iter):
            g = class_gradient(spec, p, batch[active], tgt[active])
            if theta > 0:
                ind = np.argpartition(g, -2, axis=1)[:, -2:]
            else:
                ind = np.argpartition(-g, -2, axis=1)[:, -2:]
            all_feat[active, ind[:, 0]] = 1
            all_feat[active, ind[:, 1]] = 1
            tmp = batch[active]
            tmp[np.arange(len(active)), ind[:, 0]] += theta
            tmp[np.arange(len(active)), ind[:, 1]] += theta
            batch[active] = tmp
            cur = mlp_ref.forward_infer(spec, p, batch).argmax(1)
            active = np.where((cur != tgt) & (all_feat.sum(axis=1) / nf <= gamma))[0]
            it += 1
        adv[s0:s0 + batch_size] = batch
    return adv


# ------------------------------------------------------------------ (f)-4: Carlini & Wagner, as ART runs them
_TANH_SMOOTHER = 0.999999
_C_UPPER_BOUND = 10e10


def _to_tanh(x, lo, hi):
    """art.utils.original_to_tanh: arctanh(((x - lo) / (hi - lo) * 2 - 1) * smoother)."""
    return np.arctanh((np.clip(x, lo, hi) - lo) / (hi - lo) * 2 * _TANH_SMOOTHER - _TANH_SMOOTHER)


def _from_tanh(xt, lo, hi):
    """art.utils.tanh_to_original: (tanh(xt) / smoother + 1) / 2 * (hi - lo) + lo."""
    return (np.tanh(xt) / _TANH_SMOOTHER + 1.0) / 2.0 * (hi - lo) + lo


def _cw_margin(z, target, confidence, targeted=False):
    """ART's rearranged C&W term on the model OUTPUT z (softmax probabilities for this Keras model):
    max(z_target - z_other + confidence, 0) untargeted."""
    z_target = np.sum(z * target, axis=1)
    z_other = np.max(z * (1 - target) + (np.min(z, axis=1) - 1)[:, None] * target, axis=1)
    if targeted:
        return np.maximum(z_other - z_target + confidence, 0.0)
    return np.maximum(z_target - z_other + confidence, 0.0)


def _cw_class_grad_diff(spec, p, z, target, x_adv, targeted=False):
    """class_gradient(x_adv, label=i_add) - class_gradient(x_adv, label=i_sub) (ART _loss_gradient)."""
    other = np.argmax(z * (1 - target) + (np.min(z, axis=1) - 1)[:, None] * target, axis=1)
    tgt = np.argmax(target, axis=1)
    i_add, i_sub = (other, tgt) if targeted else (tgt, other)
    v = np.zeros_like(z)
    v[np.arange(len(z)), i_add] += 1.0
    v[np.arange(len(z)), i_sub] -= 1.0
    return mlp_ref.output_vjp_infer(spec, p, x_adv, v, on_logits=False)[0]


def carlini_l2(spec, p, x, confidence=0.0, learning_rate=0.01, binary_search_steps=10, max_iter=10, initial_const=0.01,
               max_halving=5, max_doubling=5, batch_size=1, y=None, dtype=np.float64):
    """CarliniL2Method(classifier, confidence).generate(x) (VD/attacks.py:606-616), untargeted, no clip_values
    (then ART takes clip_min, clip_max = min(x), max(x)), restated from ART 1.9-1.10's published implementation:
    binary search over c, gradient steps in tanh space with a halving / doubling line search on the learning rate.
    PARITY UNPINNED (ART absent): written from the algorithm as published; the bookkeeping may differ in detail."""
    x = np.asarray(x, dtype=dtype)  # ART works in float32 (ART_NUMPY_DTYPE); float64 is the reference-grade evaluation
    x_adv = x.copy()
    lo, hi = dtype(np.amin(x)), dtype(np.amax(x))
    y = _own_labels(spec, p, x, batch_size) if y is None else np.asarray(y, dtype=dtype)
    predict = lambda a: mlp_ref.forward_infer(spec, p, a)

    def loss_fn(xb, xa, yb, c):
        l2 = np.sum(np.square(xb - xa), axis=1)
        z = predict(xa)
        return z, l2, c * _cw_margin(z, yb, confidence) + l2

    for s0 in range(0, len(x), batch_size):
        xb, yb = x[s0:s0 + batch_size], y[s0:s0 + batch_size]
        n = len(xb)
        xb_tanh = _to_tanh(xb, lo, hi)
        c_cur = initial_const * np.ones(n, dtype=dtype)
        c_lower = np.zeros(n, dtype=dtype)
        c_double = np.ones(n) > 0
        best_l2 = np.inf * np.ones(n, dtype=dtype)
        best_adv = xb.copy()
        for _bss in range(binary_search_steps):
            if int(np.sum(c_cur < _C_UPPER_BOUND)) == 0:
                break
            lr = learning_rate * np.ones(n, dtype=dtype)
            xa = xb.copy()
            xa_tanh = xb_tanh.copy()
            z, l2, loss = loss_fn(xb, xa, yb, c_cur)
            success = loss - l2 <= 0
            overall = success.copy()
            for _it in range(max_iter):
                improved = success & (l2 < best_l2)
                best_l2[improved] = l2[improved]
                best_adv[improved] = xa[improved]
                active = (c_cur < _C_UPPER_BOUND) & (lr > 0)
                if int(np.sum(active)) == 0:
                    break
                # gradient of the loss in tanh space, negated
                g = _cw_class_grad_diff(spec, p, z[active], yb[active], xa[active])
                g = g * c_cur[active][:, None] + 2 * (xa[active] - xb[active])
                g = g * (hi - lo) * (1 - np.square(np.tanh(xa_tanh[active]))) / (2 * _TANH_SMOOTHER)
                pert = -g
                prev_loss, best_loss = loss.copy(), loss.copy()
                best_lr = np.zeros(n, dtype=dtype)
                halving = np.zeros(n)
                for _h in range(max_halving):
                    do = loss[active] >= prev_loss[active]
                    if int(np.sum(do)) == 0:
                        break
                    sel = active.copy()
                    sel[active] = do
                    new_tanh = xa_tanh[sel] + lr[sel][:, None] * pert[do]
                    new_x = _from_tanh(new_tanh, lo, hi)
                    _, l2[sel], loss[sel] = loss_fn(xb[sel], new_x, yb[sel], c_cur[sel])
                    better = loss < best_loss
                    best_lr[better] = lr[better]
                    best_loss[better] = loss[better]
                    lr[sel] /= 2
                    halving[sel] += 1
                lr[active] *= 2
                for _d in range(max_doubling):
                    do = (halving[active] == 1) & (loss[active] <= best_loss[active])
                    if int(np.sum(do)) == 0:
                        break
                    sel = active.copy()
                    sel[active] = do
                    lr[sel] *= 2
                    new_tanh = xa_tanh[sel] + lr[sel][:, None] * pert[do]
                    new_x = _from_tanh(new_tanh, lo, hi)
                    _, l2[sel], loss[sel] = loss_fn(xb[sel], new_x, yb[sel], c_cur[sel])
                    better = loss < best_loss
                    best_lr[better] = lr[better]
                    best_loss[better] = loss[better]
                lr[halving == 1] /= 2
                upd = best_lr[active] > 0
                if int(np.sum(upd)) > 0:
                    sel = active.copy()
                    sel[active] = upd
                    xa_tanh[sel] = xa_tanh[sel] + best_lr[sel][:, None] * pert[upd]
                    xa[sel] = _from_tanh(xa_tanh[sel], lo, hi)
                    z[sel], l2[sel], loss[sel] = loss_fn(xb[sel], xa[sel], yb[sel], c_cur[sel])
                    success = loss - l2 <= 0
                    overall = overall | success
            improved = success & (l2 < best_l2)
            best_l2[improved] = l2[improved]
            best_adv[improved] = xa[improved]
            # binary search on c: halve towards the lower bound after a success, else raise the lower bound and double
            c_double[overall] = False
            c_old = c_cur.copy()
            c_cur[overall] = c_lower[overall] + (c_cur - c_lower)[overall] / 2
            fail = ~overall
            c_lower[fail] = c_old[fail]
            c_cur[fail & c_double] = c_cur[fail & c_double] * 2
            nd = fail & ~c_double
            c_cur[nd] = c_cur[nd] + (c_cur - c_lower)[nd] / 2
        x_adv[s0:s0 + batch_size] = best_adv
    return x_adv


def carlini_linf(spec, p, x, confidence=0.0, learning_rate=0.01, max_iter=10, max_halving=5, max_doubling=5, eps=0.3,
                 batch_size=128, y=None, dtype=np.float64):
    """CarliniLInfMethod(classifier, confidence).generate(x) (VD/attacks.py:578-582), untargeted, no clip_values:
    minimise the C&W margin alone inside the box [x - eps, x + eps] (the tanh change of variables makes the box
    the constraint), same line search as the L2 attack; samples stop moving once the margin reaches 0.
    PARITY UNPINNED (ART absent), restated from the published implementation."""
    x = np.asarray(x, dtype=dtype)
    x_adv = x.copy()
    y = _own_labels(spec, p, x, batch_size) if y is None else np.asarray(y, dtype=dtype)
    eps = dtype(eps)
    predict = lambda a: mlp_ref.forward_infer(spec, p, a)
    for s0 in range(0, len(x), batch_size):
        xb, yb = x[s0:s0 + batch_size], y[s0:s0 + batch_size]
        n = len(xb)
        lo, hi = xb - eps, xb + eps
        xa = xb.copy()
        xa_tanh = _to_tanh(xb, lo, hi)
        z = predict(xa)
        loss = _cw_margin(z, yb, confidence)
        lr = learning_rate * np.ones(n, dtype=dtype)
        for _it in range(max_iter):
            active = (loss > 0) & (lr > 0)
            if int(np.sum(active)) == 0:
                break
            g = _cw_class_grad_diff(spec, p, z[active], yb[active], xa[active])
            g = g * (hi - lo)[active] * (1 - np.square(np.tanh(xa_tanh[active]))) / (2 * _TANH_SMOOTHER)
            pert = -g
            prev_loss, best_loss = loss.copy(), loss.copy()
            best_lr = np.zeros(n, dtype=dtype)
            halving = np.zeros(n)
            for _h in range(max_halving):
                do = loss[active] >= prev_loss[active]
                if int(np.sum(do)) == 0:
                    break
                sel = active.copy()
                sel[active] = do
                new_x = _from_tanh(xa_tanh[sel] + lr[sel][:, None] * pert[do], lo[sel], hi[sel])
                loss[sel] = _cw_margin(predict(new_x), yb[sel], confidence)
                better = loss < best_loss
                best_lr[better] = lr[better]
                best_loss[better] = loss[better]
                lr[sel] /= 2
                halving[sel] += 1
            lr[active] *= 2
            for _d in range(max_doubling):
                do = (halving[active] == 1) & (loss[active] <= best_loss[active])
                if int(np.sum(do)) == 0:
                    break
                sel = active.copy()
                sel[active] = do
                lr[sel] *= 2
                new_x = _from_tanh(xa_tanh[sel] + lr[sel][:, None] * pert[do], lo[sel], hi[sel])
                loss[sel] = _cw_margin(predict(new_x), yb[sel], confidence)
                better = loss < best_loss
                best_lr[better] = lr[better]
                best_loss[better] = loss[better]
            lr[halving == 1] /= 2
            upd = best_lr[active] > 0
            if int(np.sum(upd)) > 0:
                sel = active.copy()
                sel[active] = upd
                xa_tanh[sel] = xa_tanh[sel] + best_lr[sel][:, None] * pert[upd]
                xa[sel] = _from_tanh(xa_tanh[sel], lo[sel], hi[sel])
            z = predict(xa)
            loss = _cw_margin(z, yb, confidence)
        x_adv[s0:s0 + batch_size] = xa
    return x_adv


# ------------------------------------------------------------------ A12
def add_white_noise(array, sigma, rng):
    """VD/attacks.py:73-86."""
    return array + rng.normal(0, sigma, np.array(array).shape[0])


def mixtgauss(N, p, sigma0, sigma1, rng):
    """VD/attacks.py:145-163: u = |q| < p ; x = (sigma0*(1-u) + sigma1*u) * N(0,1)."""
    q = rng.normal(0, 1, N)
    u = np.abs(q) < p
    return (sigma0 * (1 - u) + sigma1 * u) * rng.normal(0, 1, N)


def add_noise(x, p, alpha, rng):
    """VD/attacks.py:166-183: impulse mixture with sigma1 = 10*alpha."""
    return x + mixtgauss(x.shape[0], p, alpha, 10 * alpha, rng)


def add_white_noise_with_snr(audio, target_snr_db, rng):
    """VD/attacks.py:222-245."""
    sample = np.asanyarray(audio)
    signal_avg_watts = np.mean(sample ** 2)
    signal_avg_db = 10 * np.log10(signal_avg_watts)
    noise_avg_watts = 10 ** ((signal_avg_db - target_snr_db) / 10)
    return sample + rng.normal(0, np.sqrt(noise_avg_watts), len(sample))
