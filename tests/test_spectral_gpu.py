"""GPU parity: K3 Lipschitz projections (HIP) against oracle.constraints_ref (LAPACK SVD on the host).

Tolerances: sigma and projected kernels agree to 2e-5 relative.  The HIP path sums in fp32 with a
different association than LAPACK and takes sigma from a Gram eigenvalue (product) or a converged
power iteration (per layer); the reference's own float32 SVD is good to ~1e-6.
"""
import os

import numpy as np
import pytest
import torch

from golden import inputs
from helpers import dev, rel_err
from oracle import constraints_ref as R

pytestmark = pytest.mark.gpu
RTOL = 2e-5


class FakeLayer:
    """The 10-line duck-typed layer protocol of Constraints.py: name / get_weights / set_weights."""

    def __init__(self, name, w, b=None):
        self.name, self.w, self.b = name, np.array(w, dtype=np.float32), np.zeros(w.shape[1], np.float32) if b is None else b

    def get_weights(self):
        return [self.w.copy(), self.b.copy()]

    def set_weights(self, ws):
        self.w, self.b = np.array(ws[0], dtype=np.float32), np.array(ws[1], dtype=np.float32)


class FakeModel:
    def __init__(self, ws):
        self.layers = []
        for i, w in enumerate(ws):
            self.layers.append(FakeLayer("dense" if i == 0 else f"dense_{i}", w))
            if i < len(ws) - 1:  # a non-dense layer in between, as Dropout sits in the reference model
                self.layers.append(type("Drop", (), {"name": f"dropout_{i}", "get_weights": lambda self: []})())


@pytest.mark.parametrize("widths", [inputs.SMALL_WIDTHS, inputs.FULL_WIDTHS, [50, 20], [33, 17, 5]])
@pytest.mark.parametrize("rho", [0.1, 10.0])
def test_simple_norm_constraint_all_layers(cuda, widths, rho):
    from lipasr.Constraints import simple_norm_constraint

    ws = inputs.nonneg_kernels(widths)
    ref, norms = R.simple_norm_constraint_pass(ws, rho, [])
    model = FakeModel(ws)
    cb = simple_norm_constraint(rho, [])
    cb.set_model(model)
    cb.on_batch_end(0)
    got_norms = cb.last_norms.cpu().numpy()
    np.testing.assert_allclose(got_norms, norms, rtol=RTOL)
    for l, r in zip([l for l in model.layers if "dense" in l.name], ref):
        assert rel_err(l.w, r) < RTOL
    # closed form of SURVEY 3.1
    m = len(ws)
    for k, n in enumerate(got_norms):
        assert abs(n - R.simple_norm_closed_form(norms[0], rho, m, k)) / n < 5e-5


def test_simple_norm_constraint_listed_indices_and_signed_kernels(cuda):
    from lipasr.Constraints import simple_norm_constraint

    ws = inputs.signed_kernels(inputs.SMALL_WIDTHS)
    for idx in ([1], [0, 2, 2], [2, 0]):
        ref, norms = R.simple_norm_constraint_pass(ws, 0.1, idx)
        model = FakeModel(ws)
        cb = simple_norm_constraint(0.1, idx)
        cb.set_model(model)
        cb.on_batch_end(0)
        np.testing.assert_allclose(cb.last_norms.cpu().numpy(), norms, rtol=RTOL)
        for l, r in zip([l for l in model.layers if "dense" in l.name], ref):
            assert rel_err(l.w, r) < RTOL
        # get_projection keeps the reference's signature: one kernel in, scaled kernel out
        model2 = FakeModel(ws)
        cb.set_model(model2)
        np.testing.assert_allclose(cb.get_projection(ws[1]), R.simple_norm_projection(ws[1], ws, 0.1), rtol=RTOL)


def test_product_norm_golden_and_zero(cuda, golden_dir):
    from lipasr.extract_features_construct_dataset import get_lipschitz_constrained, product_norm

    g = np.load(os.path.join(golden_dir, "constraints.npz"))
    ws = inputs.nonneg_kernels(inputs.SMALL_WIDTHS)
    model = FakeModel(ws)
    assert abs(float(product_norm(model).item()) - g["sn_norms"][0]) / g["sn_norms"][0] < RTOL
    zero = FakeModel([np.zeros_like(w) for w in ws])
    assert float(product_norm(zero).item()) == 0.0
    assert abs(get_lipschitz_constrained(FakeModel(ws)) - R.get_lipschitz_constrained(ws, [])) / g["sn_norms"][0] < RTOL


@pytest.mark.parametrize("widths", [inputs.SMALL_WIDTHS, inputs.FULL_WIDTHS])
def test_norm_constraint_cold_then_warm(cuda, widths, golden_dir):
    from lipasr.Constraints import norm_constraint

    rho = 10.0
    ws = inputs.nonneg_kernels(widths)
    model = FakeModel(ws)
    cb = norm_constraint(rho)
    cb.set_model(model)
    cb.on_train_begin()
    assert cb.m == len(ws)
    cb.on_batch_end(0)  # cold start: 48 round trips
    ref = R.norm_constraint_pass(ws, rho)
    sig = cb.last_sigmas.cpu().numpy()
    np.testing.assert_allclose(sig, R.get_norms(ws), rtol=RTOL)
    dense = [l for l in model.layers if "dense" in l.name]
    for l, r in zip(dense, ref):
        assert rel_err(l.w, r) < RTOL
        assert abs(R.sigma_max(l.w) - rho ** (1 / len(ws))) < 5e-5  # every kernel ends at rho^(1/m)
    if widths == inputs.SMALL_WIDTHS:
        g = np.load(os.path.join(golden_dir, "constraints.npz"))
        for i, l in enumerate(dense):
            assert rel_err(l.w, g[f"nc_{i}"]) < RTOL
    # perturb (a training step would), then one warm pass of 4 round trips
    rng = np.random.default_rng(0)
    for l in dense:
        l.w = (l.w * (1 + 0.02 * rng.standard_normal(l.w.shape))).astype(np.float32) - np.float32(1e-3)
    pert = [l.w.copy() for l in dense]
    cb.on_batch_end(1)
    ref2 = R.norm_constraint_pass(pert, rho)
    for l, r in zip(dense, ref2):
        assert rel_err(l.w, r) < 1e-4
        assert l.w.min() >= 0


def test_sigma_max_signed_matrix(cuda):
    from lipasr import _native as N

    h = N.get_handle(0)
    for shape in [(880, 1024), (64, 10), (7, 300)]:
        rng = np.random.default_rng(shape[0])
        u, _ = np.linalg.qr(rng.standard_normal((shape[0], min(shape))))
        v, _ = np.linalg.qr(rng.standard_normal((shape[1], min(shape))))
        s = np.concatenate([[3.0], np.linspace(2.4, 0.1, min(shape) - 1)])  # gap ratio 0.8
        w = ((u * s) @ v.T).astype(np.float32)
        wt, vs, out = dev(w), torch.zeros(shape[1], device="cuda"), torch.zeros(1, device="cuda")
        N.check(N.lib.lipasr_sigma_max(h.h, N.ptr(wt), shape[0], shape[1], N.ptr(vs), 0, 200, 0, N.ptr(out), N.stream_ptr()))
        assert abs(float(out.item()) - R.sigma_max(w)) / R.sigma_max(w) < RTOL


def test_custom_constraint_frobenius(cuda):
    from lipasr.Constraints import customConstraint

    w = inputs.signed_kernels([880, 1024])[0]
    c = customConstraint(5.0)
    assert c.get_config() == {"rho": 5.0}
    assert rel_err(c(w), R.custom_constraint(w, 5.0)) < 1e-5
    t = dev(w)
    out = c(t)
    assert out.is_cuda and out.data_ptr() != t.data_ptr() and torch.equal(t.cpu(), torch.as_tensor(w))  # pure function

@pytest.mark.parametrize("R_,n,hi", [(10, 880, 3.0), (6, 24, 0.5), (32, 1000, 20.0), (1, 5, 0.1), (7, 3, 0.4), (10, 880, 0.0)])
def test_sv_clip_matches_lapack(cuda, R_, n, hi):
    """lipasr_sv_clip against numpy's SVD (Constraints.py:86-89 arithmetic) on seeded matrices."""
    import lipasr._native as N

    rng = np.random.default_rng(100 + R_ + n)
    x = rng.standard_normal((R_, n)).astype(np.float32)
    x[0] *= 4.0  # spread the spectrum so that some values are clipped and some are not
    u, sv, vt = np.linalg.svd(x.astype(np.float64), full_matrices=False)
    want = (u * np.clip(sv, 0, hi)) @ vt
    xt = torch.as_tensor(x).cuda()
    out = torch.empty_like(xt)
    got_sv = torch.empty(R_, device="cuda")
    h = N.get_handle(0)
    N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(xt), R_, n, hi, N.ptr(out), N.ptr(got_sv), N.stream_ptr()))
    k = min(R_, n)
    np.testing.assert_allclose(got_sv.cpu().numpy()[:k], sv[:k], rtol=2e-6, atol=1e-6 * sv[0])
    assert np.all(np.abs(got_sv.cpu().numpy()[k:]) <= 1e-6 * sv[0])
    assert np.max(np.abs(out.cpu().numpy() - want)) <= 2e-6 * max(sv[0], 1.0)
    # in place, singular values only, and argument checking
    N.check(N.lib.lipasr_sv_clip(h.h, N.ptr(xt), R_, n, hi, N.ptr(xt), None, N.stream_ptr()))
    assert torch.equal(xt, out)
    assert N.lib.lipasr_sv_clip(h.h, N.ptr(xt), 33, n, hi, N.ptr(xt), None, N.stream_ptr()) == N.EINVAL
    assert N.lib.lipasr_sv_clip(h.h, N.ptr(xt), R_, n, hi, None, None, N.stream_ptr()) == N.EINVAL


def test_fista_surface(cuda):
    from lipasr.Constraints import norm_constraint_FISTA

    ws = inputs.nonneg_kernels([24, 16, 12, 6], seed=2)
    model = FakeModel(ws)
    cb = norm_constraint_FISTA(rho=5.0, nit=2)
    cb.set_model(model)
    cb.on_batch_end(0)
    ref = R.fista_pass(ws, 5.0, 2)
    for l, r in zip([l for l in model.layers if "dense" in l.name], ref):
        assert rel_err(l.w, r) < 2e-3  # fp32 products + GPU SVD vs the float64 host iteration


def test_fista_reference_size(cuda):
    """The reference's own setting (train_constraints.py:100: norm_constraint_FISTA(rho=5, nit=2)) on the reference's
    kernels 880-1024-512-256-128-64-10, against the float64 host iteration of oracle.constraints_ref.fista_pass
    (Constraints.py:54-130 line by line); also the wall time of one on_batch_end, for DESIGN.md."""
    import time

    from lipasr.Constraints import norm_constraint_FISTA

    ws = inputs.nonneg_kernels(inputs.FULL_WIDTHS, seed=4)
    ref = R.fista_pass(ws, 5.0, 2)
    model = FakeModel(ws)
    cb = norm_constraint_FISTA(rho=5.0, nit=2)
    cb.set_model(model)
    cb.on_batch_end(0)
    dense = [l for l in model.layers if "dense" in l.name]
    errs = [rel_err(l.w, r) for l, r in zip(dense, ref)]
    moved = [rel_err(r, w) for r, w in zip(ref, ws)]
    assert max(moved) > 1e-2  # the pass is not a no-op at this size
    assert max(errs) < 2e-3, errs
    # a second pass from the projected kernels (the callback runs after every batch): its wall time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    cb.on_batch_end(1)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(np.isfinite(l.w).all() and l.w.min() >= 0 for l in dense)
    print(f"\nFISTA rho=5 nit=2 at 880-1024-512-256-128-64-10: max rel err {max(errs):.2e}, one on_batch_end {dt * 1e3:.1f} ms")


def test_lip_readouts_on_native_model(cuda):
    from helpers import build_model, load_params
    from lipasr.extract_features_construct_dataset import get_lipschitz_constrained, get_norms, get_upper_lipschitz
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(1)
    for l in range(5):
        p.gamma[l] = (1 + 0.1 * rng.standard_normal(spec[l].n_out)).astype(np.float32)
        p.mov_var[l] = rng.uniform(0.5, 2.0, spec[l].n_out).astype(np.float32)
    m = build_model(spec)
    load_params(m, p)
    norms = get_norms(m)
    np.testing.assert_allclose(norms, R.get_norms(p.W), rtol=RTOL)
    assert abs(get_upper_lipschitz(norms) - R.get_upper_lipschitz(R.get_norms(p.W))) / get_upper_lipschitz(norms) < 1e-4
    bn = [(p.gamma[l], p.mov_var[l]) for l in range(5)]
    ref = R.get_lipschitz_constrained(p.W, bn)
    assert abs(get_lipschitz_constrained(m) - ref) / ref < 5e-5


@pytest.mark.parametrize("widths", [inputs.FULL_WIDTHS, [2020, 1024, 512, 256, 128, 64, 20], [300, 100, 64, 32, 16, 5], [300, 130, 70, 33, 17, 5],
                                    [300, 256, 128, 64, 16], [200, 96, 256, 48, 16, 3]])
def test_chain_head_launch_against_one_launch_per_step(cuda, widths):
    """Round 4: the leading small steps of the product chain W_m^T ... W_1^T run as one launch (chain_head_kernel: every
    workgroup recomputes the small products in its LDS on the fp32 matrix instruction).  Exact fp32 with another association
    of the sums than chain_step_kernel's: norms and projected kernels agree with one launch per step (lipasr_debug_chain_head(0))
    to 2e-6 and with the LAPACK oracle to RTOL, and repeat bit for bit.  Reference widths (2 steps fused by default, 3 on request), Speaker-recognition
    widths (20 classes: not fused), a ragged last fused step (100 rows), widths that are no multiples of 16 (not fused), 16 classes (the tile's columns full) and a
    256-column panel (the widest the fused launch takes)."""
    from lipasr import _native as N
    from lipasr.Constraints import simple_norm_constraint

    ws = inputs.nonneg_kernels(widths)
    res = {}
    try:
        for n in (0, -1, 3, -1):
            N.lib.lipasr_debug_chain_head(n)
            model = FakeModel(ws)
            cb = simple_norm_constraint(0.1, [])
            cb.set_model(model)
            cb.on_batch_end(0)
            res.setdefault(n, []).append((cb.last_norms.cpu().numpy().copy(), [l.w.copy() for l in model.layers if "dense" in l.name]))
    finally:
        N.lib.lipasr_debug_chain_head(-1)
    ref, norms = R.simple_norm_constraint_pass(ws, 0.1, [])
    for n in (0, -1, 3):
        np.testing.assert_allclose(res[n][0][0], norms, rtol=RTOL)
        np.testing.assert_allclose(res[n][0][0], res[0][0][0], rtol=2e-6)
        for a, b, r in zip(res[n][0][1], res[0][0][1], ref):
            np.testing.assert_allclose(a, b, rtol=2e-6, atol=0)
            assert rel_err(a, r) < RTOL
    np.testing.assert_array_equal(res[-1][0][0], res[-1][1][0])
    for a, b in zip(res[-1][0][1], res[-1][1][1]):
        np.testing.assert_array_equal(a, b)
