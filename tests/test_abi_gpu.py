"""GPU: liblipasr.so driven through raw ctypes (no lipasr Python classes), with error-code checks --
the binding a maintainer of the reference would write."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_raw_ctypes_projection_and_errors(cuda):
    from golden import inputs
    from oracle import constraints_ref as R

    lib = C.CDLL(os.path.join(ROOT, "asr-using-robust-nn_amd", "lipasr", "liblipasr.so"))
    lib.lipasr_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.lipasr_create(99, C.byref(h)) == -1 and b"out of range" in lib.lipasr_last_error()
    assert lib.lipasr_create(0, C.byref(h)) == 0
    ws = inputs.nonneg_kernels(inputs.FULL_WIDTHS)
    dws = [torch.as_tensor(w).cuda().contiguous() for w in ws]
    n = len(dws)
    ptrs = (C.c_void_p * n)(*[w.data_ptr() for w in dws])
    rows = (C.c_int * n)(*[w.shape[0] for w in dws])
    cols = (C.c_int * n)(*[w.shape[1] for w in dws])
    order = (C.c_int * n)(*range(n))
    norms = torch.zeros(n + 1, device="cuda")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.lipasr_project_product(h, ptrs, rows, cols, n, C.c_float(0.1), order, n, C.c_void_p(norms.data_ptr()), stream)
    assert rc == 0, lib.lipasr_last_error()
    ref, ref_norms = R.simple_norm_constraint_pass(ws, 0.1, [])
    np.testing.assert_allclose(norms.cpu().numpy(), ref_norms, rtol=2e-5)
    for d, r in zip(dws, ref):
        np.testing.assert_allclose(d.cpu().numpy(), r, rtol=2e-5)
    # error conventions: negative code + message, nothing thrown, nothing launched
    bad_rows = (C.c_int * n)(*([7] + [w.shape[0] for w in dws[1:]]))
    assert lib.lipasr_project_product(h, ptrs, rows, cols, n, C.c_float(-1.0), order, n, C.c_void_p(norms.data_ptr()), stream) == -1
    assert b"rho" in lib.lipasr_last_error()
    bad_order = (C.c_int * n)(*([9] * n))
    assert lib.lipasr_project_product(h, ptrs, rows, cols, n, C.c_float(0.1), bad_order, n, C.c_void_p(norms.data_ptr()), stream) == -1
    cols_bad = (C.c_int * n)(*([w.shape[1] for w in dws[:-1]] + [40]))
    assert lib.lipasr_project_product(h, ptrs, rows, cols_bad, n, C.c_float(0.1), order, n, C.c_void_p(norms.data_ptr()), stream) == -4
    assert lib.lipasr_project_product(h, ptrs, bad_rows, cols, 0, C.c_float(0.1), order, 0, C.c_void_p(norms.data_ptr()), stream) == -1
    assert lib.lipasr_sign_step(h, None, None, None, C.c_size_t(4), C.c_float(0.1), C.c_float(0.1), stream) == -1
    assert lib.lipasr_destroy(h) == 0


def test_timer_and_graph_replay(cuda):
    """lipasr_graph_*: a captured projection replays with device-resident scalars; lipasr_timer_* brackets it."""
    from golden import inputs
    from lipasr import _native as N

    h = N.get_handle(0)
    ws = [torch.as_tensor(w).cuda().contiguous() for w in inputs.nonneg_kernels(inputs.SMALL_WIDTHS)]
    n = len(ws)
    ptrs = N.ptr_array([w.data_ptr() for w in ws])
    rows, cols = N.int_array([w.shape[0] for w in ws]), N.int_array([w.shape[1] for w in ws])
    order = N.int_array(list(range(n)))
    norms = torch.zeros(n + 1, device="cuda")
    s = torch.cuda.Stream()
    tid, gid = C.c_int(), C.c_int()
    N.check(N.lib.lipasr_timer_create(h.h, C.byref(tid)))
    with torch.cuda.stream(s):
        N.check(N.lib.lipasr_graph_begin(h.h, N.stream_ptr()))
        N.check(N.lib.lipasr_project_product(h.h, C.cast(ptrs, N.PV), rows, cols, n, 0.1, order, n, N.ptr(norms), N.stream_ptr()))
        N.check(N.lib.lipasr_graph_end(h.h, N.stream_ptr(), C.byref(gid)))
        seq = []
        N.check(N.lib.lipasr_timer_start(h.h, tid.value, N.stream_ptr()))
        for _ in range(3):
            N.check(N.lib.lipasr_graph_launch(h.h, gid.value, N.stream_ptr()))
            s.synchronize()
            seq.append(norms.cpu().numpy().copy())
        N.check(N.lib.lipasr_timer_stop(h.h, tid.value, N.stream_ptr()))
    ms = C.c_float()
    N.check(N.lib.lipasr_timer_elapsed_ms(h.h, tid.value, C.byref(ms)))
    assert ms.value > 0
    # capture itself launched nothing; each replay advances the product norm towards rho by the (5/6)-law of SURVEY 3.1
    for a, b in zip(seq[:-1], seq[1:]):
        assert abs(b[0] - a[-1]) / a[-1] < 1e-5
    assert seq[-1][-1] < seq[0][0]
    N.check(N.lib.lipasr_graph_destroy(h.h, gid.value))
    assert N.lib.lipasr_graph_launch(h.h, gid.value, N.stream_ptr()) == N.EINVAL


def test_adam_project_product_equals_the_two_calls(cuda):
    """lipasr_mlp_adam_project_product == lipasr_mlp_adam_nonneg then lipasr_mlp_project_product, bit for bit,
    including the step counter (advanced inside the projection's kernel instead of its own launch)."""
    import lipasr._native as N
    from helpers import build_model
    from oracle import mlp_ref as P

    spec = P.vd_constrained_spec()
    rng = np.random.default_rng(0)
    x = torch.as_tensor(rng.standard_normal((96, 880)).astype(np.float32)).cuda()
    y = torch.as_tensor(P.to_categorical(rng.integers(0, 10, 96), 10).astype(np.float32)).cuda()
    results = []
    for fused in (False, True):
        m = build_model(spec, max_batch=128, seed=3)
        norms = torch.zeros(7, device="cuda")
        order = N.int_array(list(range(6)))
        for _ in range(3):
            m.train_fwd_bwd(x, y, dropout=False)
            if fused:
                m.apply_adam_project_product(0.1, order, norms)
            else:
                m.apply_adam()
                N.check(N.lib.lipasr_mlp_project_product(m._plan, N.ptr(m._params), 0.1, order, 6, N.ptr(norms), N.stream_ptr()))
        torch.cuda.synchronize()
        results.append((m._params.clone(), norms.clone(), int(m._step.item())))
    assert results[0][2] == results[1][2] == 3
    assert torch.equal(results[0][0], results[1][0])
    assert torch.equal(results[0][1], results[1][1])
