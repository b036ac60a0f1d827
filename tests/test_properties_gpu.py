"""Property tests (hypothesis) of the kernel-level entry points on random shapes: ragged sizes, unaligned leading
dimensions and degenerate inputs that the fixed-shape parity tests do not reach.  Each property is one the
reference's arithmetic has regardless of size (linearity of the GEMM, idempotence of the projections, the
singular-value clip as a contraction, PGD's box), checked against NumPy on the same inputs.
"""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from helpers import dev

pytestmark = pytest.mark.gpu
COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _native():
    import lipasr._native as N

    return N, N.get_handle(0)


@settings(max_examples=40, **COMMON)
@given(M=st.integers(1, 200), Nn=st.integers(1, 200), K=st.integers(1, 300), ta=st.booleans(), tb=st.booleans(),
       seed=st.integers(0, 2**31 - 1))
def test_gemm_random_shapes(cuda, M, Nn, K, ta, tb, seed):
    N, h = _native()
    rng = np.random.default_rng(seed)
    a = rng.integers(-4, 5, (M, K)).astype(np.float32)   # small integers: the product is exact in fp32
    b = rng.integers(-4, 5, (K, Nn)).astype(np.float32)
    A = dev(np.ascontiguousarray(a.T if ta else a))
    B = dev(np.ascontiguousarray(b.T if tb else b))
    out = torch.full((M, Nn), float("nan"), device="cuda")
    N.check(N.lib.lipasr_gemm_f32(h.h, int(ta), int(tb), M, Nn, K, N.ptr(A), A.shape[1], N.ptr(B), B.shape[1], N.ptr(out), Nn, N.stream_ptr()))
    np.testing.assert_array_equal(out.cpu().numpy(), a.astype(np.float64) @ b.astype(np.float64))


@settings(max_examples=25, **COMMON)
@given(R_=st.integers(1, 32), n=st.integers(1, 400), hi=st.floats(0.0, 5.0), seed=st.integers(0, 2**31 - 1))
def test_sv_clip_properties(cuda, R_, n, hi, seed):
    """U min(S, hi) V^T: singular values of the result are min(s, hi); clipping twice changes nothing; a level above
    sigma_max returns the input."""
    N, h = _native()
    x = np.random.default_rng(seed).standard_normal((R_, n)).astype(np.float32)
    xt = dev(x)
    out, sv = torch.empty_like(xt), torch.empty(R_, device="cuda")
    N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(xt), R_, n, hi, N.ptr(out), N.ptr(sv), N.stream_ptr()))
    s_ref = np.linalg.svd(x.astype(np.float64), compute_uv=False)
    k = min(R_, n)
    tol = 3e-6 * max(s_ref[0], 1.0)
    assert np.abs(sv.cpu().numpy()[:k] - s_ref).max() <= tol
    s_out = np.linalg.svd(out.cpu().numpy().astype(np.float64), compute_uv=False)
    assert np.abs(s_out - np.minimum(s_ref, hi)).max() <= 3 * tol
    again = torch.empty_like(xt)
    N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(out), R_, n, hi, N.ptr(again), None, N.stream_ptr()))
    assert np.abs((again - out).cpu().numpy()).max() <= 3 * tol
    same = torch.empty_like(xt)
    N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(xt), R_, n, float(s_ref[0] * 1.01 + 1e-3), N.ptr(same), None, N.stream_ptr()))
    assert np.abs((same - xt).cpu().numpy()).max() <= 3 * tol


@settings(max_examples=25, **COMMON)
@given(widths=st.lists(st.integers(1, 96), min_size=1, max_size=5), n_cls=st.integers(1, 32), rho=st.floats(0.05, 20.0),
       seed=st.integers(0, 2**31 - 1))
def test_product_projection_reaches_rho_and_is_idempotent(cuda, widths, n_cls, rho, seed):
    """simple_norm_constraint (Constraints.py:158-189) on random stacks: after the sequential pass the product norm
    follows the closed form n_m = n0^((1-1/m)^m) rho^(1-(1-1/m)^m); a stack already at rho is left alone."""
    import ctypes as C

    from oracle import constraints_ref as R

    N, h = _native()
    dims = list(widths) + [n_cls]
    rng = np.random.default_rng(seed)
    ws = [np.abs(rng.standard_normal((a, b))).astype(np.float32) + 0.01 for a, b in zip(dims[:-1], dims[1:])]
    m = len(ws)
    n0 = R.sigma_max(R.product_chain(ws))
    wt = [dev(w) for w in ws]
    ptrs = N.ptr_array([t.data_ptr() for t in wt])
    rows, cols = N.int_array([w.shape[0] for w in ws]), N.int_array([w.shape[1] for w in ws])
    order = N.int_array(list(range(m)))
    norms = torch.zeros(m + 1, device="cuda")

    def project():
        N.check(N.lib.lipasr_project_product(h.h, ptrs, rows, cols, m, C.c_float(rho), order, m, N.ptr(norms), N.stream_ptr()))
        return norms.cpu().numpy().astype(np.float64)

    got = project()
    assert abs(got[0] - n0) <= 2e-5 * n0
    q = (1.0 - 1.0 / m) ** m
    assert abs(got[-1] - n0 ** q * rho ** (1 - q)) <= 1e-4 * got[-1]
    after = R.sigma_max(R.product_chain([t.cpu().numpy() for t in wt]))
    assert abs(after - got[-1]) <= 1e-4 * after
    # drive it to the fixed point: every further pass moves the norm towards rho and rho itself is a fixed point
    for _ in range(60):
        got = project()
    assert abs(got[-1] - rho) <= 1e-3 * rho
    before = [t.clone() for t in wt]
    project()
    for a, b in zip(before, wt):
        assert torch.allclose(a, b, rtol=2e-3, atol=0)


@settings(max_examples=30, **COMMON)
@given(n=st.integers(1, 5000), alpha=st.floats(0.0, 2.0), eps=st.one_of(st.floats(0.0, 3.0), st.just(float("inf"))),
       seed=st.integers(0, 2**31 - 1))
def test_sign_step_box_and_oracle(cuda, n, alpha, eps, seed):
    from lipasr.attacks import sign_step
    from oracle import attacks_ref as A

    rng = np.random.default_rng(seed)
    x0 = rng.standard_normal(n).astype(np.float32)
    xa = (x0 + rng.uniform(-1, 1, n)).astype(np.float32)
    g = rng.standard_normal(n).astype(np.float32)
    g[rng.integers(0, n)] = 0.0
    got = sign_step(dev(xa), dev(x0), dev(g), alpha, eps).cpu().numpy()
    want = A.sign_step(xa.astype(np.float32), x0, g, np.float32(alpha), np.float32(eps))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)
    if np.isfinite(eps):
        assert np.all(np.abs(got - x0) <= eps * (1 + 1e-6) + 1e-6)


@settings(max_examples=20, **COMMON)
@given(rows=st.integers(2, 300), cols=st.integers(1, 130), seed=st.integers(0, 2**31 - 1))
def test_scaler_matches_sklearn_rule(cuda, rows, cols, seed):
    """StandardScaler().fit_transform (train_constraints.py:28-35) incl. constant columns (scale 1)."""
    from lipasr.attacks import StandardScaler
    from oracle import mlp_ref as P

    rng = np.random.default_rng(seed)
    x = (rng.standard_normal((rows, cols)) * rng.uniform(0.1, 30, cols) + rng.uniform(-50, 50, cols)).astype(np.float32)
    x[:, rng.integers(0, cols)] = 3.25  # a constant feature
    sc = StandardScaler().fit(dev(x))
    mean, scale = P.standard_scaler_fit(x.astype(np.float64))
    np.testing.assert_allclose(sc.mean_.cpu().numpy(), mean, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sc.scale_.cpu().numpy(), scale, rtol=1e-5, atol=1e-7)
    got = sc.transform(dev(x)).cpu().numpy()
    np.testing.assert_allclose(got, (x.astype(np.float64) - mean) / scale, atol=2e-5)
