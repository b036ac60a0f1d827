"""Oracle A6-A8b, A11: the four Lipschitz constraints and the Lipschitz read-outs.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference module imports
TensorFlow at VD/Constraints.py:2 and cannot be run here (no stand-in import was made), and the
reference holds no tests or stored outputs.  Each function cites the reference lines it follows
and is checked in tests/ by known answers (sigma_max == rho^(1/m); the product-norm law).
Note the reference lines ``if cst == []`` / ``if A == []`` raise on ndarrays under NumPy >= 2
(2022 NumPy evaluated them to False with a DeprecationWarning, i.e. took the else branch); the
restatement uses ``is None``.

Weight convention everywhere: Keras Dense kernel W is (in, out), y = x @ W.
"""
from __future__ import annotations

import numpy as np

EPS = float(np.spacing(1))  # 2.220446049250313e-16, VD/Constraints.py:25


def sigma_max(w) -> float:
    """np.linalg.norm(w, ord=2): largest singular value (LAPACK gesdd)."""
    return float(np.linalg.norm(np.asarray(w), ord=2))


# ------------------------------------------------------------------ A7  VD/Constraints.py:9-33
def norm_constraint_projection(w, rho, m):
    """get_projection (VD/Constraints.py:22-25): clamp >= 0, scale so sigma_max = rho^(1/m)."""
    w = np.asarray(w)
    w = w * np.greater_equal(w, 0)
    norm = np.linalg.norm(w, ord=2)
    return (w * np.power(rho, 1 / m) / (norm + np.spacing(1))).astype(np.float32)


def norm_constraint_pass(w_list, rho):
    """on_batch_end (VD/Constraints.py:27-33) over all dense kernels; m = number of dense layers (:15-20)."""
    m = len(w_list)
    return [norm_constraint_projection(w, rho, m) for w in w_list]


# ------------------------------------------------------------------ A8  VD/Constraints.py:38-49
def custom_constraint(w, rho):
    """customConstraint.__call__: tf.norm(w, ord=2) with axis=None is the FROBENIUS norm (2-norm of the
    flattened tensor), not the spectral norm."""
    w = np.asarray(w, dtype=np.float32)
    w = w * (w >= 0).astype(w.dtype)
    norm = np.sqrt(np.sum(w.astype(np.float64) ** 2))
    return (w * (rho / (norm + EPS))).astype(np.float32)


# ------------------------------------------------------------------ A6  VD/Constraints.py:135-189
def product_chain(w_list):
    """cst = W_m^T . W_{m-1}^T ... W_1^T  (VD/Constraints.py:160-166), left-to-right association."""
    cst = None
    for index in reversed(range(len(w_list))):
        wt = np.array(w_list[index]).transpose()
        cst = wt if cst is None else np.matmul(cst, wt)
    return cst


def simple_norm_projection(w, w_list, rho):
    """get_projection (VD/Constraints.py:158-169): w * (rho / (||cst||_2 + eps))^(1/len(w_list))."""
    cst = product_chain(w_list)
    s = np.power(rho / (np.linalg.norm(cst, ord=2) + np.spacing(1)), 1 / len(w_list))
    return (np.asarray(w) * np.float32(s)).astype(np.float32)


def simple_norm_constraint_pass(w_list, rho, affected_layers_indices=()):
    """on_batch_end (VD/Constraints.py:171-189).

    Empty index list: every dense layer, SEQUENTIALLY -- each projection sees the already rescaled
    earlier kernels because get_w_list() is re-evaluated inside get_projection (:159).
    Non-empty: only the listed indices (position among dense layers), visited in reverse order (:181)."""
    w_list = [np.asarray(w, dtype=np.float32).copy() for w in w_list]
    norms = []
    if len(affected_layers_indices) == 0:
        order = list(range(len(w_list)))
    else:
        # :181-189: for index in reversed(range(m)): for layer_index in affected: if layer_index == index -> project
        # (a layer listed twice is projected twice)
        order = [index for index in reversed(range(len(w_list))) for layer_index in affected_layers_indices if layer_index == index]
    for i in order:
        norms.append(sigma_max(product_chain(w_list)))
        w_list[i] = simple_norm_projection(w_list[i], w_list, rho)
    norms.append(sigma_max(product_chain(w_list)))
    return w_list, norms


def simple_norm_closed_form(n0, rho, m, k):
    """Product norm after k of m sequential rescalings: n_k = n0^((1-1/m)^k) * rho^(1-(1-1/m)^k)."""
    q = (1.0 - 1.0 / m) ** k
    return n0 ** q * rho ** (1.0 - q)


# ------------------------------------------------------------------ A8b VD/Constraints.py:54-130
def fista_constraint(w, Y0, A, B, nit, rho):
    """Constraint_Fista (VD/Constraints.py:69-94), line by line.  ``w`` is the TRANSPOSED kernel (out, in)."""
    Y = Y0
    Yold = Y0
    gam = 1 / ((np.linalg.norm(A, ord=2) * np.linalg.norm(B, ord=2) + np.spacing(1)) ** 2)
    alpha = 2.1
    w_new = w
    for i in range(nit):
        eta = i / (i + 1 + alpha)
        Z = Y + eta * (Y - Yold)
        Yold = Y
        w_new = w - A.T @ Z @ B.T
        w_new = w_new * np.greater_equal(w_new, 0)
        T = A @ w_new @ B
        s = np.linalg.svd(T, compute_uv=False)
        criterion = np.linalg.norm(w_new - w, ord="fro")
        constraint = np.linalg.norm(s[s > rho] - rho, ord=2)
        Yt = Z + gam * T
        u1, s1, v1 = np.linalg.svd(Yt / gam, full_matrices=False)
        s1 = np.clip(s1, 0, rho)
        Y = Yt - gam * np.dot(u1 * s1, v1)
        if criterion < 30 and constraint < 0.01:
            return w_new
    return w_new


def fista_projection(w, w_list, rho, nit):
    """get_projection (VD/Constraints.py:96-122) for the kernel ``w`` (Keras layout (in, out)).

    The layer is located by VALUE equality, lowest matching index wins (:100-102).
    A = product of the LATER layers' transposes (identity for the last layer, :116-117),
    B = product of the EARLIER layers' transposes (identity for the first layer, :114-115).
    dtypes follow the reference: kernels (and hence A, B, w.T) are float32 as get_weights() returns
    them, the identities and Y0 are float64 (np.eye / np.zeros defaults), so the iteration promotes to
    float64 from the first ``A.T @ Z @ B.T`` on.  Returns the new kernel in Keras layout (in, out)
    (the caller transposes back, :130)."""
    w = np.asarray(w)
    w_index = None
    for index in reversed(range(len(w_list))):
        if np.array_equal(w, w_list[index]):
            w_index = index
    if w_index is None:
        raise ValueError("kernel not found in w_list")
    A = None
    Bm = None
    for index in reversed(range(len(w_list))):
        wt = np.array(w_list[index]).transpose()
        if index > w_index:
            A = wt if A is None else np.matmul(A, wt)
        elif index < w_index:
            Bm = wt if Bm is None else np.matmul(Bm, wt)
    if w_index == 0:
        Bm = np.eye(w.shape[0], w.shape[0])
    if w_index == len(w_list) - 1:
        A = np.eye(w.shape[1], w.shape[1])
    Y0 = np.zeros([A.shape[0], Bm.shape[1]])
    w_new = fista_constraint(w.T, Y0, A, Bm, nit, rho)
    return w_new.T.astype(np.float32)


def fista_pass(w_list, rho, nit):
    """on_batch_end (VD/Constraints.py:124-130): every dense layer in order, each seeing earlier updates."""
    w_list = [np.asarray(w, dtype=np.float32).copy() for w in w_list]
    for i in range(len(w_list)):
        w_list[i] = fista_projection(w_list[i], w_list, rho, nit)
    return w_list


# ------------------------------------------------------------------ A11
def get_norms(w_list):
    """VD/extract_features_construct_dataset.py:154-161."""
    return np.array([sigma_max(w) for w in w_list])


def get_upper_lipschitz(norms):
    """VD/extract_features_construct_dataset.py:165-166."""
    return float(np.prod(norms))


def get_lipschitz_constrained(w_list, bn_list):
    """VD/extract_features_construct_dataset.py:169-196.  bn_list: [(gamma, moving_variance), ...]."""
    correction_factor = 1.0
    cfs = [np.sqrt(np.asarray(var, dtype=np.float64)) / np.asarray(gamma, dtype=np.float64) for gamma, var in bn_list]
    if cfs:
        correction_factor = float(np.prod([np.max(c) for c in cfs]))
    return sigma_max(product_chain(w_list)) / correction_factor
