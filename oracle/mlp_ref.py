"""Oracle A2-A5: StandardScaler, the Keras MLP, categorical CE, Keras-form Adam, NonNeg.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: TensorFlow/Keras and
scikit-learn semantics are restated (TF 2.7-2.9 era defaults), call sites:

  VD/train_constraints.py:28-35   StandardScaler().fit_transform(concat(train,dev,test))
  VD/train_constraints.py:63-88   get_model(): Dense(relu,NonNeg) -> BN -> Dropout ... Dense(softmax)
  VD/train_constraints.py:94      compile(optimizer='adam', loss=CategoricalCrossentropy())
  VD/train_constraints.py:37-42   shuffle(880, reshuffle_each_iteration=False).batch(512)
  VD/train_google_dataset.py:49-74 unconstrained variant (Dropout .4 everywhere, no NonNeg)

Everything here is plain NumPy in a caller-chosen dtype (float64 for the "exact"
reference, float32 for the reference-equivalent CPU baseline).
"""
from __future__ import annotations

from dataclasses import dataclass, field
import numpy as np

BN_MOMENTUM = 0.99
BN_EPS = 1e-3
ADAM_LR = 1e-3
ADAM_B1 = 0.9
ADAM_B2 = 0.999
ADAM_EPS = 1e-7


# ----------------------------------------------------------------------------- A2
def standard_scaler_fit(all_data: np.ndarray):
    """sklearn StandardScaler.fit: per-feature mean and population std; zero std -> scale 1."""
    all_data = np.asarray(all_data, dtype=np.float64)
    mean = all_data.mean(axis=0)
    var = all_data.var(axis=0)
    scale = np.sqrt(var)
    # sklearn >= 1.0 _handle_zeros_in_scale + _is_constant_feature: (near-)constant columns keep scale 1
    n = all_data.shape[0]
    eps = np.finfo(np.float64).eps
    constant = var <= n * eps * var + (n * mean * eps) ** 2
    scale[(scale < 10 * eps) | constant] = 1.0
    return mean, scale


def standardize_dataset(train, val, test):
    """VD/attacks.py:48-69 / VD/train_constraints.py:28-35: fit on the concatenation, split back."""
    all_data = np.concatenate((train, val, test), axis=0)
    mean, scale = standard_scaler_fit(all_data)
    all_data = (all_data - mean) / scale
    a, b = train.shape[0], train.shape[0] + val.shape[0]
    return all_data[:a], all_data[a:b], all_data[b:]


def to_categorical(labels, num_classes):
    y = np.zeros((len(labels), num_classes), dtype=np.float32)
    y[np.arange(len(labels)), np.asarray(labels).astype(np.int64)] = 1.0
    return y


# ----------------------------------------------------------------------------- A3
@dataclass
class LayerSpec:
    n_in: int
    n_out: int
    bn: bool
    dropout: float
    nonneg: bool


def vd_constrained_spec():
    """get_model() of VD/train_constraints.py:63-88."""
    w = [880, 1024, 512, 256, 128, 64, 10]
    drop = [0.1, 0.1, 0.1, 0.0, 0.0, 0.0]
    return [LayerSpec(w[i], w[i + 1], i < 5, drop[i], True) for i in range(6)]


def vd_unconstrained_spec():
    """get_model() of VD/train_google_dataset.py:49-74."""
    w = [880, 1024, 512, 256, 128, 64, 10]
    return [LayerSpec(w[i], w[i + 1], i < 5, 0.4 if i < 5 else 0.0, False) for i in range(6)]


def sr_constrained_spec():
    """get_model() of SR/train_constraints.py:63-88: the same stack with 2020 inputs and 20 speakers."""
    w = [2020, 1024, 512, 256, 128, 64, 20]
    drop = [0.1, 0.1, 0.1, 0.0, 0.0, 0.0]
    return [LayerSpec(w[i], w[i + 1], i < 5, drop[i], True) for i in range(6)]


def sr_unconstrained_spec():
    """get_model() of SR/train_no_constraints.py:52-74: Dense/ReLU only (BatchNorm and Dropout commented out)."""
    w = [2020, 1024, 512, 256, 128, 64, 20]
    return [LayerSpec(w[i], w[i + 1], False, 0.0, False) for i in range(6)]


@dataclass
class Params:
    W: list = field(default_factory=list)       # (in, out)
    b: list = field(default_factory=list)
    gamma: list = field(default_factory=list)   # None where no BN
    beta: list = field(default_factory=list)
    mov_mean: list = field(default_factory=list)
    mov_var: list = field(default_factory=list)

    def copy(self):
        c = lambda l: [None if a is None else a.copy() for a in l]
        return Params(c(self.W), c(self.b), c(self.gamma), c(self.beta), c(self.mov_mean), c(self.mov_var))

    def astype(self, dt):
        c = lambda l: [None if a is None else a.astype(dt) for a in l]
        return Params(c(self.W), c(self.b), c(self.gamma), c(self.beta), c(self.mov_mean), c(self.mov_var))


def init_params(spec, seed=0, dtype=np.float32, nonneg_init=False) -> Params:
    """glorot_uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1 (Keras defaults)."""
    rng = np.random.default_rng(seed)
    p = Params()
    for s in spec:
        lim = np.sqrt(6.0 / (s.n_in + s.n_out))
        W = rng.uniform(-lim, lim, size=(s.n_in, s.n_out))
        if nonneg_init:
            W = np.abs(W)
        p.W.append(W.astype(dtype))
        p.b.append(np.zeros(s.n_out, dtype=dtype))
        if s.bn:
            p.gamma.append(np.ones(s.n_out, dtype=dtype)); p.beta.append(np.zeros(s.n_out, dtype=dtype))
            p.mov_mean.append(np.zeros(s.n_out, dtype=dtype)); p.mov_var.append(np.ones(s.n_out, dtype=dtype))
        else:
            p.gamma.append(None); p.beta.append(None); p.mov_mean.append(None); p.mov_var.append(None)
    return p


def softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def forward_infer(spec, p: Params, x, return_logits=False):
    """training=False: BN uses moving statistics, Dropout is identity."""
    h = x
    for l, s in enumerate(spec):
        z = h @ p.W[l] + p.b[l]
        if l == len(spec) - 1:
            return z if return_logits else softmax(z)
        h = np.maximum(z, 0)
        if s.bn:
            h = (h - p.mov_mean[l]) / np.sqrt(p.mov_var[l] + BN_EPS) * p.gamma[l] + p.beta[l]
    raise AssertionError


def forward_backward(spec, p: Params, x, y_onehot, masks=None, training=True, need_dx=False):
    """One training-mode forward + backward.

    masks[l]: inverted-dropout multiplier array (0 or 1/(1-p)) for layer l, or None.
    Returns dict with logits, loss (mean over batch), grads (dW, db, dgamma, dbeta),
    batch statistics (mean, var) per BN layer and optionally dx.
    Gradient at the logits is (softmax - y)/B (Keras recovers logits from the softmax op).
    """
    L = len(spec)
    B = x.shape[0]
    cache = []
    h = x
    for l, s in enumerate(spec):
        inp = h
        z = inp @ p.W[l] + p.b[l]
        if l == L - 1:
            cache.append(dict(inp=inp))
            logits = z
            break
        a = np.maximum(z, 0)
        c = dict(inp=inp, a=a)
        if s.bn:
            if training:
                mu = a.mean(axis=0); var = a.var(axis=0)
            else:
                mu = p.mov_mean[l]; var = p.mov_var[l]
            rstd = 1.0 / np.sqrt(var + BN_EPS)
            xhat = (a - mu) * rstd
            h = xhat * p.gamma[l] + p.beta[l]
            c.update(mu=mu, var=var, rstd=rstd, xhat=xhat)
        else:
            h = a
        if masks is not None and masks[l] is not None:
            h = h * masks[l]
            c["mask"] = masks[l]
        cache.append(c)
    prob = softmax(logits)
    # Keras recovers the logits behind the softmax op and uses the log-softmax form, which stays finite
    # when a probability underflows (log(prob) gave 0 * -inf = NaN in float32).
    zs = logits - logits.max(axis=1, keepdims=True)
    logp = zs - np.log(np.exp(zs).sum(axis=1, keepdims=True))
    loss = float(-(np.where(y_onehot != 0, y_onehot * logp, 0.0)).sum(axis=1).mean())
    g = (prob - y_onehot) / B
    dW = [None] * L; db = [None] * L; dgamma = [None] * L; dbeta = [None] * L
    dx = None
    for l in reversed(range(L)):
        c = cache[l]
        s = spec[l]
        if l < L - 1:
            if "mask" in c:
                g = g * c["mask"]
            if s.bn:
                dgamma[l] = (g * c["xhat"]).sum(axis=0)
                dbeta[l] = g.sum(axis=0)
                if training:
                    g = p.gamma[l] * c["rstd"] * (g - dbeta[l] / B - c["xhat"] * dgamma[l] / B)
                else:
                    g = p.gamma[l] * c["rstd"] * g
            g = g * (c["a"] > 0)
        dW[l] = c["inp"].T @ g
        db[l] = g.sum(axis=0)
        if l > 0 or need_dx:
            g = g @ p.W[l].T
            if l == 0:
                dx = g
    stats = [(c.get("mu"), c.get("var")) for c in cache]
    return dict(logits=logits, prob=prob, loss=loss, dW=dW, db=db, dgamma=dgamma, dbeta=dbeta, stats=stats, dx=dx)


def input_gradient_infer(spec, p: Params, x, y_onehot):
    """d CE(f(x), y)/dx in inference mode (what ART's loss_gradient returns, up to the 1/B mean factor)."""
    return forward_backward(spec, p, x, y_onehot, masks=None, training=False, need_dx=True)["dx"]


def output_vjp_infer(spec, p: Params, x, v, on_logits=False):
    """sum_c v[b, c] * d out_c(x_b)/dx in inference mode; out = softmax probabilities (what ART's class_gradient
    differentiates for a Keras model ending in softmax) or the logits.  Returns (dx, probs)."""
    L = len(spec)
    cache = []
    h = x
    for l, s in enumerate(spec):
        z = h @ p.W[l] + p.b[l]
        if l == L - 1:
            logits = z
            break
        a = np.maximum(z, 0)
        if s.bn:
            rstd = 1.0 / np.sqrt(p.mov_var[l] + BN_EPS)
            h = (a - p.mov_mean[l]) * rstd * p.gamma[l] + p.beta[l]
            cache.append((a, p.gamma[l] * rstd))
        else:
            h = a
            cache.append((a, None))
    prob = softmax(logits)
    g = np.asarray(v, dtype=prob.dtype)
    if not on_logits:
        g = prob * (g - (prob * g).sum(axis=1, keepdims=True))
    for l in reversed(range(L)):
        if l < L - 1:
            a, scale = cache[l]
            if scale is not None:
                g = g * scale
            g = g * (a > 0)
        g = g @ p.W[l].T
    return g, prob


# ----------------------------------------------------------------------------- A4
@dataclass
class AdamState:
    m: dict = field(default_factory=dict)
    v: dict = field(default_factory=dict)
    t: int = 0


def adam_update(w, g, m, v, t, lr=ADAM_LR, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    """Keras optimizer_v2 Adam: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); w -= lr_t*m/(sqrt(v)+eps)."""
    dt = w.dtype
    lr_t = dt.type(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    # the same operations in the same order as the plain expressions
    #   m = m*b1 + g*(1-b1);  v = v*b2 + (g*g)*(1-b2);  w = w - lr_t*m/(sqrt(v)+eps)
    # written with in-place ufuncs (one temporary instead of seven: this is most of a CPU step's time)
    tmp = g * dt.type(1 - b1)
    np.multiply(m, dt.type(b1), out=m)
    np.add(m, tmp, out=m)
    np.multiply(g, g, out=tmp)
    np.multiply(tmp, dt.type(1 - b2), out=tmp)
    np.multiply(v, dt.type(b2), out=v)
    np.add(v, tmp, out=v)
    np.sqrt(v, out=tmp)
    np.add(tmp, dt.type(eps), out=tmp)
    np.divide(lr_t * m, tmp, out=tmp)
    np.subtract(w, tmp, out=w)


def train_step(spec, p: Params, st: AdamState, x, y_onehot, masks=None, grads_override=None):
    """fwd, bwd, Adam on every trainable, NonNeg on constrained kernels, BN moving-stat update.

    Order follows Keras: moving stats are updated during the forward pass, the optimizer applies
    all updates, kernel constraints run right after each variable's update; callbacks come later."""
    out = forward_backward(spec, p, x, y_onehot, masks=masks, training=True)
    grads = grads_override if grads_override is not None else out
    st.t += 1
    for l, s in enumerate(spec):
        for name, arr, g in (("W", p.W[l], grads["dW"][l]), ("b", p.b[l], grads["db"][l]),
                             ("gamma", p.gamma[l], grads["dgamma"][l]), ("beta", p.beta[l], grads["dbeta"][l])):
            if arr is None:
                continue
            key = (name, l)
            if key not in st.m:
                st.m[key] = np.zeros_like(arr); st.v[key] = np.zeros_like(arr)
            adam_update(arr, g.astype(arr.dtype), st.m[key], st.v[key], st.t)
        if s.nonneg:
            p.W[l] *= (p.W[l] >= 0)
        if s.bn:
            mu, var = out["stats"][l]
            dt = p.mov_mean[l].dtype
            p.mov_mean[l] = p.mov_mean[l] * dt.type(BN_MOMENTUM) + mu.astype(dt) * dt.type(1 - BN_MOMENTUM)
            p.mov_var[l] = p.mov_var[l] * dt.type(BN_MOMENTUM) + var.astype(dt) * dt.type(1 - BN_MOMENTUM)
    return out


# ----------------------------------------------------------------------------- A5
def tf_shuffle_batches(n, batch, buffer_size=880, seed=0):
    """Index batches in the spirit of Dataset.shuffle(buffer, reshuffle_each_iteration=False).batch(b).

    TF's RNG stream cannot be matched; what IS reproduced is the structure: a sliding shuffle
    buffer of ``buffer_size`` elements (element i can only move forward by < buffer), one fixed
    order reused every epoch, partial last batch kept."""
    rng = np.random.default_rng(seed)
    buf = list(range(min(buffer_size, n)))
    nxt = len(buf)
    order = []
    while buf:
        j = int(rng.integers(0, len(buf)))
        order.append(buf[j])
        if nxt < n:
            buf[j] = nxt; nxt += 1
        else:
            buf[j] = buf[-1]; buf.pop()
    order = np.asarray(order, dtype=np.int64)
    return [order[i:i + batch] for i in range(0, n, batch)]
