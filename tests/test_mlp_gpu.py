"""GPU parity: K2 (fp32 MFMA GEMM, BatchNorm, dropout, softmax-CE) and K5 (Adam + NonNeg) against
oracle.mlp_ref evaluated in float64.

Tolerances: the HIP path is exact-fp32 fma chains (v_mfma_f32_32x32x2_f32) in a different summation
order than the oracle; activations/gradients agree to ~1e-5 relative to the tensor's max, logits to
well inside BASELINE's 1e-3 relative bound with identical argmax.
"""
import os

import numpy as np
import pytest
import torch

from golden import inputs
from helpers import build_model, dev, grads_of, load_params, read_params, rel_err
from lipasr import keras as K
from oracle import mlp_ref as P

pytestmark = pytest.mark.gpu


def _gemm(a, b, ta, tb):
    from lipasr import _native as N

    h = N.get_handle(0)
    A = dev(a.T if ta else a)
    B = dev(b.T if tb else b)
    M, K = a.shape
    Nn = b.shape[1]
    out = torch.full((M, Nn), float("nan"), device="cuda")
    N.check(N.lib.lipasr_gemm_f32(h.h, int(ta), int(tb), M, Nn, K, N.ptr(A), A.shape[1], N.ptr(B), B.shape[1], N.ptr(out), Nn, N.stream_ptr()))
    return out.cpu().numpy()


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("shape", [(512, 1024, 880), (182, 10, 64), (64, 10, 182), (33, 47, 21), (1, 1, 1), (512, 880, 10), (880, 1024, 512)])
def test_gemm_all_layouts(cuda, ta, tb, shape):
    M, N_, K = shape
    rng = np.random.default_rng(M + 7 * N_ + 13 * K)
    a = rng.standard_normal((M, K)).astype(np.float32)
    b = rng.standard_normal((K, N_)).astype(np.float32)
    # asymmetric integer check first: catches a transposed C map exactly
    ai = rng.integers(-3, 4, (M, K)).astype(np.float32)
    bi = rng.integers(-3, 4, (K, N_)).astype(np.float32)
    np.testing.assert_array_equal(_gemm(ai, bi, ta, tb), ai.astype(np.float64) @ bi.astype(np.float64))
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bound = 4e-7 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-30
    assert np.all(np.abs(_gemm(a, b, ta, tb) - ref) <= bound)


def test_gemm_rejects_bad_arguments(cuda):
    from lipasr import _native as N

    h = N.get_handle(0)
    t = torch.zeros(4, 4, device="cuda")
    assert N.lib.lipasr_gemm_f32(h.h, 0, 0, 4, 4, 0, N.ptr(t), 4, N.ptr(t), 4, N.ptr(t), 4, N.stream_ptr()) == N.EINVAL
    assert N.lib.lipasr_gemm_f32(h.h, 0, 0, 4, 4, 4, None, 4, N.ptr(t), 4, N.ptr(t), 4, N.stream_ptr()) == N.EINVAL
    assert N.lib.lipasr_gemm_f32(h.h, 0, 0, 4, 4, 4, N.ptr(t), 2, N.ptr(t), 4, N.ptr(t), 4, N.stream_ptr()) == N.EINVAL


def test_small_case_against_golden(cuda, golden_dir):
    from golden.make_golden import spec_from

    g = np.load(os.path.join(golden_dir, "mlp.npz"))
    spec = spec_from(inputs.MLP_SMALL)
    p = P.init_params(spec, seed=5, dtype=np.float32, nonneg_init=True)
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    x, y, masks = inputs.mlp_small_case()
    xt, yt = dev(x), dev(y)
    mt = [dev(k) if k is not None else None for k in masks]
    np.testing.assert_allclose(m.predict(x), P.softmax(g["infer_logits"]), atol=1e-5)  # inference mode, before any update
    probs = torch.zeros(8, 10, device="cuda")
    m.train_fwd_bwd(xt, yt, masks=mt, probs=probs)
    got = grads_of(m, spec)
    for l in range(3):
        assert rel_err(got["dW"][l], g[f"dW{l}"]) < 2e-5
        assert rel_err(got["db"][l], g[f"db{l}"]) < 2e-5
        if spec[l].bn:
            assert rel_err(got["dgamma"][l], g[f"dgamma{l}"]) < 2e-5
            assert rel_err(got["dbeta"][l], g[f"dbeta{l}"]) < 2e-5
    assert abs(float(m._loss_rows[:8].mean()) - float(g["loss"])) < 1e-5
    m.apply_adam()
    after = read_params(m, spec)
    for l in range(3):
        assert rel_err(after.W[l], g[f"W{l}_after"]) < 1e-5
        assert rel_err(after.b[l], g[f"b{l}_after"]) < 1e-5 or np.abs(g[f"b{l}_after"]).max() < 1e-2
        if spec[l].bn:
            assert rel_err(after.mov_mean[l], g[f"mm{l}_after"]) < 1e-5
            assert rel_err(after.mov_var[l], g[f"mv{l}_after"]) < 1e-5
    assert int(m._step.item()) == 1


def _random_state(spec, seed):
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(seed + 100)
    for l, s in enumerate(spec):
        p.b[l] = (0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
        if s.bn:
            p.gamma[l] = (1 + 0.2 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.beta[l] = (0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_mean[l] = (0.5 + 0.1 * rng.standard_normal(s.n_out)).astype(np.float32)
            p.mov_var[l] = rng.uniform(0.5, 1.5, s.n_out).astype(np.float32)
    return p


@pytest.mark.parametrize("batch", [512, 182])
def test_full_model_forward_backward(cuda, batch):
    spec = P.vd_constrained_spec()
    p = _random_state(spec, 1)
    m = build_model(spec)
    load_params(m, p)
    rng = np.random.default_rng(batch)
    x = rng.standard_normal((batch, 880)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 10, batch), 10)
    masks = [((rng.uniform(size=(batch, s.n_out)) > s.dropout) / (1 - s.dropout)).astype(np.float32) if s.dropout > 0 else None for s in spec]
    m.train_fwd_bwd(dev(x), dev(y), masks=[dev(k) if k is not None else None for k in masks])
    ref = P.forward_backward(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64), masks=masks, training=True)
    got = grads_of(m, spec)
    for l in range(6):
        assert rel_err(got["dW"][l], ref["dW"][l]) < 5e-5, l
        assert rel_err(got["db"][l], ref["db"][l]) < 5e-5, l
        if spec[l].bn:
            # dgamma/dbeta are sums of `batch` signed terms that cancel to ~1e-6: fp32 accumulation noise
            # relative to the largest entry is a few 1e-5
            assert rel_err(got["dgamma"][l], ref["dgamma"][l]) < 3e-4, l
            assert rel_err(got["dbeta"][l], ref["dbeta"][l]) < 3e-4, l
    assert abs(float(m._loss_rows[:batch].mean()) - ref["loss"]) < 1e-4 * max(1.0, abs(ref["loss"]))
    pred_ok = (ref["prob"].argmax(1) == y.argmax(1)).astype(np.float32)
    np.testing.assert_array_equal(m._correct_rows[:batch].cpu().numpy(), pred_ok)


def test_inference_logits_2366_clips(cuda):
    """BASELINE parity statement: per-utterance logits within 1e-3 relative, argmax identical."""
    spec = P.vd_constrained_spec()
    p = _random_state(spec, 2)
    m = build_model(spec)
    load_params(m, p)
    x = np.random.default_rng(5).standard_normal((2366, 880)).astype(np.float32)
    ref = P.forward_infer(spec, p.astype(np.float64), x.astype(np.float64), return_logits=True)
    got = m.predict_device(dev(x), logits=True).cpu().numpy()
    rel = np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-6)
    assert rel.max() <= 1e-3, rel.max()
    np.testing.assert_array_equal(got.argmax(1), ref.argmax(1))
    np.testing.assert_allclose(m.predict(x), P.softmax(ref), atol=2e-5)


def test_n_train_steps_track_the_oracle(cuda):
    """5 optimizer steps (fwd, bwd, Adam, NonNeg, BN moving statistics) with injected dropout masks."""
    spec = P.vd_constrained_spec()
    p = _random_state(spec, 3)
    m = build_model(spec)
    load_params(m, p)
    p64 = p.astype(np.float64)
    st = P.AdamState()
    rng = np.random.default_rng(11)
    for step in range(5):
        bsz = 512 if step < 4 else 182
        x = rng.standard_normal((bsz, 880)).astype(np.float32)
        y = P.to_categorical(rng.integers(0, 10, bsz), 10)
        masks = [((rng.uniform(size=(bsz, s.n_out)) > s.dropout) / (1 - s.dropout)).astype(np.float32) if s.dropout > 0 else None for s in spec]
        P.train_step(spec, p64, st, x.astype(np.float64), y.astype(np.float64), masks=masks)
        m.train_on_batch(dev(x), dev(y), masks=[dev(k) if k is not None else None for k in masks])
    after = read_params(m, spec)
    for l in range(6):
        # Adam normalises the update, so after 5 steps every weight moved by <= 5e-3; compare the movement
        d = np.abs(after.W[l] - p64.W[l])
        assert np.quantile(d, 0.9999) < 2e-4 and d.max() < 3e-3, (l, np.quantile(d, 0.9999), d.max())
        assert after.W[l].min() >= 0
        if spec[l].bn:
            assert rel_err(after.mov_mean[l], p64.mov_mean[l]) < 1e-4
            assert rel_err(after.mov_var[l], p64.mov_var[l]) < 1e-4
            assert np.abs(after.gamma[l] - p64.gamma[l]).max() < 2e-4
    assert int(m._step.item()) == 5


def test_philox_dropout_statistics_and_backward_consistency(cuda):
    spec = [P.LayerSpec(64, 256, False, 0.4, False), P.LayerSpec(256, 10, False, 0.0, False)]
    m = build_model(spec, max_batch=1024)
    x = np.abs(np.random.default_rng(0).standard_normal((1024, 64))).astype(np.float32)
    y = P.to_categorical(np.zeros(1024, dtype=int), 10)
    dense = [l for l in m.layers if "dense" in l.name]
    dense[0].set_weights([np.full((64, 256), 0.01, np.float32), np.ones(256, np.float32)])
    m.train_fwd_bwd(dev(x), dev(y))
    from lipasr import _native as N
    # H of block 0 lives in the plan workspace; recover the mask from dW of layer 1 instead: rows of the mask
    # are visible through predict-vs-train difference, so check the keep rate via the bias gradient path:
    g1 = m._grads.clone()
    m.train_fwd_bwd(dev(x), dev(y))
    assert torch.equal(g1, m._grads)  # same (seed, step) -> same mask -> bitwise same gradients
    m.apply_adam()  # step counter advances -> new mask
    m.train_fwd_bwd(dev(x), dev(y))
    assert not torch.equal(g1, m._grads)
    m.train_fwd_bwd(dev(x), dev(y), dropout=False)
    ref = P.forward_backward(spec, read_params(m, spec), x.astype(np.float64), y.astype(np.float64), training=True)
    assert rel_err(grads_of(m, spec)["dW"][0], ref["dW"][0]) < 5e-5


def test_layer_protocol_and_checkpoint(cuda, tmp_path):
    from lipasr import keras as K

    spec = P.vd_constrained_spec()
    m = build_model(spec)
    names = [l.name for l in m.layers]
    assert sum("dense" in n for n in names) == 6 and sum("batch" in n for n in names) == 5
    d0 = [l for l in m.layers if "dense" in l.name][0]
    w, b = d0.get_weights()
    assert w.shape == (880, 1024) and w.dtype == np.float32 and b.shape == (1024,)
    w[0, 0] = 123.0  # caller owns the copy
    assert d0.get_weights()[0][0, 0] != 123.0
    lim = np.sqrt(6.0 / (880 + 1024))
    assert abs(w[1:].max() - lim) < 1e-3 and abs(w[1:].min() + lim) < 1e-3  # glorot_uniform
    bn = [l for l in m.layers if "batch" in l.name][0].get_weights()
    assert [a.shape for a in bn] == [(1024,)] * 4 and np.all(bn[0] == 1) and np.all(bn[3] == 1) and np.all(bn[1] == 0)
    x = np.random.default_rng(0).standard_normal((40, 880)).astype(np.float32)
    before = m.predict(x)
    path = str(tmp_path / "ckpt" / "TEST.pt")
    m.save(path)
    m2 = K.load_model(path)
    np.testing.assert_array_equal(m2.predict(x), before)
    y = P.to_categorical(np.arange(40) % 10, 10)
    loss, acc = m.evaluate(x, y)
    assert np.isfinite(loss) and 0 <= acc <= 1


def test_weight_io_and_tensorboard_callback(cuda, tmp_path):
    """Model.save_weights / load_weights / set_weights (train_constraints.py:96) and the TensorBoard callback slot
    (train_constraints.py:45-48,98): scalars land in <log_dir>/scalars.jsonl."""
    import json

    from lipasr import keras as K

    spec = [P.LayerSpec(12, 8, True, 0.0, True), P.LayerSpec(8, 4, False, 0.0, True)]
    m = build_model(spec, max_batch=32, seed=1)
    m2 = build_model(spec, max_batch=32, seed=2)
    x = np.random.default_rng(0).standard_normal((20, 12)).astype(np.float32)
    assert not np.array_equal(m.predict(x), m2.predict(x))
    path = str(tmp_path / "w" / "weights.pt")
    m.save_weights(path)
    m2.load_weights(path)
    np.testing.assert_array_equal(m.predict(x), m2.predict(x))
    m3 = build_model(spec, max_batch=32, seed=3)
    m3.set_weights(m.get_weights())
    np.testing.assert_array_equal(m.predict(x), m3.predict(x))
    with pytest.raises(ValueError):
        m3.set_weights(m.get_weights()[:-1])
    full = str(tmp_path / "full.pt")
    m.save(full)
    m4 = build_model(spec, max_batch=32, seed=4)
    m4.load_weights(full)
    np.testing.assert_array_equal(m.predict(x), m4.predict(x))
    # TensorBoard slot
    y = P.to_categorical(np.arange(20) % 4, 4)
    tb = K.TensorBoard(log_dir=str(tmp_path / "logs"))
    m.fit(K.Dataset.from_tensor_slices((x, y)).batch(10), epochs=3, verbose=0, callbacks=[tb])
    lines = [json.loads(l) for l in open(tmp_path / "logs" / "scalars.jsonl")]
    assert [l["epoch"] for l in lines] == [0, 1, 2] and all("loss" in l for l in lines)


def test_keras_h5_checkpoint(cuda, tmp_path):
    """ModelCheckpoint('...TEST.h5') -> load_model (train_constraints.py:104-107): the .h5 file is Keras' HDF5 layout.
    A reloaded model predicts bit-identically AND resumes training bit-identically (Adam iterations, m and v come
    back from optimizer_weights); save_weights/load_weights go through the same format."""
    from lipasr import _hdf5 as H
    from lipasr import keras as K

    spec = [P.LayerSpec(24, 16, True, 0.0, True), P.LayerSpec(16, 8, True, 0.0, True), P.LayerSpec(8, 5, False, 0.0, True)]
    m = build_model(spec, max_batch=32, seed=7)
    m.compile(optimizer="adam", loss=K.CategoricalCrossentropy(), metrics=["accuracy"], learning_rate=3e-3)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((32, 24)).astype(np.float32)
    y = P.to_categorical(np.arange(32) % 5, 5)
    xt, yt = dev(x), dev(y)
    for _ in range(3):
        m.train_on_batch(xt, yt, dropout=False)
    path = str(tmp_path / "bin" / "models_constrained" / "TEST.h5")
    m.save(path)
    with open(path, "rb") as fh:
        assert fh.read(8) == b"\x89HDF\r\n\x1a\n"
    with H.File(path) as f:
        names = f.read_attr("model_weights", "layer_names")
        assert names == [l.name for l in m.layers]
        d0 = [l for l in m.layers if "dense" in l.name][0]
        np.testing.assert_array_equal(f.read_dataset(f"model_weights/{d0.name}/{d0.name}/kernel:0"), d0.get_weights()[0])
        assert int(f.read_dataset("optimizer_weights/Adam/iter:0")) == 3
    m2 = K.load_model(path, max_batch=32)
    assert [type(l).__name__ for l in m2.layers] == [type(l).__name__ for l in m.layers]
    assert m2._adam == m._adam
    np.testing.assert_array_equal(m2.predict(x), m.predict(x))
    for _ in range(2):
        m.train_on_batch(xt, yt, dropout=False)
        m2.train_on_batch(xt, yt, dropout=False)
    assert int(m2._step.item()) == 5
    np.testing.assert_array_equal(m2._params.cpu().numpy(), m._params.cpu().numpy())
    np.testing.assert_array_equal(m2._bnstate.cpu().numpy(), m._bnstate.cpu().numpy())
    # weights-only .h5, into a differently initialised model; and load_weights pointed at the full-model file
    wpath = str(tmp_path / "w.h5")
    m.save_weights(wpath)
    for src in (wpath, path):
        m3 = build_model(spec, max_batch=32, seed=9)
        m3.load_weights(src)
        if src == wpath:
            np.testing.assert_array_equal(m3.predict(x), m.predict(x))
        else:
            np.testing.assert_array_equal(m3.predict(x), K.load_model(path, max_batch=32).predict(x))
    wrong = build_model([P.LayerSpec(24, 16, True, 0.0, True), P.LayerSpec(16, 5, False, 0.0, True)], max_batch=32)
    with pytest.raises(ValueError):
        wrong.load_weights(wpath)
    # the callback the reference passes to fit()
    ck = str(tmp_path / "ck" / "best.h5")
    m.fit(K.Dataset.from_tensor_slices((x, y)).batch(16), epochs=2, validation_data=K.Dataset.from_tensor_slices((x, y)).batch(16),
          verbose=0, callbacks=[K.ModelCheckpoint(ck, save_best_only=True)])
    assert K.load_model(ck, max_batch=32).predict(x).shape == (32, 5)


def test_bf16_operand_mode(cuda):
    """lipasr_mlp_set_compute(1) / Model(compute_dtype="bfloat16"): BASELINE config 2's arithmetic -- GEMM operands
    rounded to bf16 at the MFMA, fp32 accumulation.  Not the parity path: the results must sit within bf16's 2^-8
    operand rounding of the fp32 ones (and far outside fp32 noise, or the mode did nothing), the fp32 path must be
    untouched, and a model must still train."""
    from lipasr import keras as K

    spec = P.vd_constrained_spec()
    p = _random_state(spec, 5)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((200, 880)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 10, 200), 10)
    ref = P.forward_infer(spec, p.astype(np.float64), x.astype(np.float64), return_logits=True)
    out = {}
    for dt in ("float32", "bfloat16"):
        K.reset_layer_names()
        m = build_model(spec, max_batch=256)
        if dt == "bfloat16":
            from lipasr import _native as N
            N.check(N.lib.lipasr_mlp_set_compute(m._plan, 1))
        load_params(m, p)
        lg = m.predict_device(dev(x), logits=True).cpu().numpy()
        m.train_fwd_bwd(dev(x), dev(y), dropout=False)
        out[dt] = (lg, grads_of(m, spec))
    rel32 = np.abs(out["float32"][0] - ref).max() / np.abs(ref).max()
    rel16 = np.abs(out["bfloat16"][0] - ref).max() / np.abs(ref).max()
    assert rel32 < 1e-4 and 1e-4 < rel16 < 3e-2, (rel32, rel16)
    assert np.mean(out["bfloat16"][0].argmax(1) == ref.argmax(1)) > 0.97
    # gradients: every backward GEMM rounds its operands and the BatchNorm backward subtracts batch means, so the
    # relative error grows from the last layer (bf16 level) towards the first; the direction is what training needs
    errs, coss = [], []
    for l in range(6):
        g32, g16 = out["float32"][1]["dW"][l], out["bfloat16"][1]["dW"][l]
        errs.append(rel_err(g16, g32))
        coss.append(float((g16 * g32).sum() / np.sqrt((g16 * g16).sum() * (g32 * g32).sum())))
    assert 1e-5 < errs[5] < 2e-2, errs
    assert min(coss) > 0.9, (errs, coss)
    # a small model trains to the same place in either arithmetic
    accs = {}
    xs = rng.standard_normal((512, 32)).astype(np.float32)
    w_true = rng.standard_normal((32, 4))
    ys = P.to_categorical((xs @ w_true).argmax(1), 4)
    for dt in ("float32", "bfloat16"):
        K.reset_layer_names()
        inp = K.Input((32,))
        h = K.Dense(64, activation="relu")(inp)
        h = K.BatchNormalization()(h)
        o = K.Dense(4, activation="softmax")(h)
        mm = K.Model(inputs=inp, outputs=o, max_batch=128, seed=3, compute_dtype=dt)
        mm.compile(optimizer="adam", loss=K.CategoricalCrossentropy(), metrics=["accuracy"])
        mm.fit(K.Dataset.from_tensor_slices((xs, ys)).batch(128), epochs=30, verbose=0)
        accs[dt] = float(np.mean(mm.predict(xs).argmax(1) == ys.argmax(1)))
    assert accs["float32"] > 0.9 and abs(accs["float32"] - accs["bfloat16"]) <= 0.03, accs
    with pytest.raises(ValueError):
        K.Model(inputs=inp, outputs=o, compute_dtype="float16")


# ------------------------------------------------------------------------------------------------
# Round 5: training-mode BatchNorm inside the producing GEMM (the exchange epilogue)
# ------------------------------------------------------------------------------------------------
def _train_state(m):
    return m._grads.clone(), m._bnstate.clone(), m._loss_rows.clone()


@pytest.mark.parametrize("widths,batch,drop", [
    ((880, 1024, 512, 256, 128, 64, 10), 1024, True),    # the reference's model at the bench's batch: the 64x64 LDS kernel and the fragment kernel
    ((880, 1024, 512, 256, 128, 64, 10), 512, True),     # the reference's own batch
    ((880, 1024, 512, 256, 128, 64, 10), 182, False),    # its last, partial batch: a ragged last row tile
    ((100, 72, 50, 33, 10), 77, True),                    # widths that are no multiple of 4 or 32, ragged rows
    ((2020, 1024, 512, 256, 128, 64, 20), 2048, False),  # the speaker-recognition shape, 64 row tiles of 32 (the regions' limit)
])
def test_batchnorm_inside_the_gemm_equals_the_launch_chain(cuda, widths, batch, drop):
    """lipasr_mlp_set_fuse_bn: the same training step with BatchNorm finished inside the producing GEMMs (column sums exchanged
    between the row tiles of a column block during the launch) and as GEMM + bn_apply launches.  The partial sums are the same
    fp32 numbers; they are added in fp64 in a different fixed order, so the two agree to rounding level (not bitwise), and each
    path is bitwise reproducible from launch to launch.  Also checks that no exchange gave up (error word)."""
    from lipasr import _native as N

    spec = [P.LayerSpec(widths[i], widths[i + 1], i + 2 < len(widths), (0.1 if (drop and i < 3 and i + 2 < len(widths)) else 0.0), True)
            for i in range(len(widths) - 1)]
    p = _random_state(spec, 4)
    rng = np.random.default_rng(batch + len(widths))
    x = dev(rng.standard_normal((batch, widths[0])).astype(np.float32))
    y = dev(P.to_categorical(rng.integers(0, widths[-1], batch), widths[-1]))
    out = {}
    for mode in (0, 1):
        m = build_model(spec, max_batch=batch)
        load_params(m, p)
        N.check(N.lib.lipasr_mlp_set_fuse_bn(m._plan, mode))
        m.train_fwd_bwd(x, y)                     # Philox dropout: same key on both paths
        first = _train_state(m)
        m.apply_adam()
        after = (m._params.clone(), m._bnstate.clone())
        # again from the same state: bitwise equal to the first launch (and the generation words have moved on)
        load_params(m, p)
        m._step.zero_()
        m.train_fwd_bwd(x, y)
        second = _train_state(m)
        for a, b in zip(first, second):
            assert torch.equal(a, b), f"mode {mode}: not reproducible"
        assert m.exchange_errors() == 0
        out[mode] = (first, after)
        m.close()
    (g0, s0, l0), (p0, b0) = out[0]
    (g1, s1, l1), (p1, b1) = out[1]
    scale = float(g0.abs().max())
    assert float((g0 - g1).abs().max()) <= 2e-6 * scale + 1e-12, float((g0 - g1).abs().max()) / scale
    torch.testing.assert_close(s1, s0, rtol=1e-6, atol=1e-7)     # moving statistics
    torch.testing.assert_close(l1[:batch], l0[:batch], rtol=1e-6, atol=1e-7)
    assert float(b1.sub(b0).abs().max()) <= 1e-6 * float(b0.abs().max())


@pytest.mark.parametrize("batch,drop,width1", [(1024, True, 1024), (1000, False, 1024), (800, True, 1024), (1024, True, 1000)])
def test_exchange_tiles_of_128_rows_on_a_cu_share_equal_the_launch_chain(cuda, batch, drop, width1):
    """Arithmetic mode 2 on a CU share (lipasr_mlp_set_cu_budget(plan, 128), what the pipeline sets beside the extraction stream):
    layer 1's forward GEMM and the input-gradient GEMM into it run on 128 x 64 tiles with the split pass (gemm_ring2_tile), the
    weight gradients on 128 x 128 split-pass tiles.  Same step as the launch chain (GEMM + bn_apply, 64 x 64 tiles) to rounding
    level, bitwise reproducible, no exchange gave up -- and the launch counters say those kernels really ran.  Ragged batches:
    1000 = 7 full row tiles + 104 rows, 800 = 6 + 32; a first hidden layer of 1000 units: a ragged last column block (40 of 64 columns)."""
    from lipasr import _native as N

    widths = (880, width1, 512, 256, 128, 64, 10)
    spec = [P.LayerSpec(widths[i], widths[i + 1], i + 2 < len(widths), (0.1 if (drop and i < 3 and i + 2 < len(widths)) else 0.0), True)
            for i in range(len(widths) - 1)]
    p = _random_state(spec, 4)
    rng = np.random.default_rng(batch)
    x = dev(rng.standard_normal((batch, widths[0])).astype(np.float32))
    y = dev(P.to_categorical(rng.integers(0, widths[-1], batch), widths[-1]))
    out = {}
    for mode in (0, 1):
        m = build_model(spec, max_batch=batch, compute_dtype="float16x2")
        load_params(m, p)
        N.check(N.lib.lipasr_mlp_set_fuse_bn(m._plan, mode))
        N.check(N.lib.lipasr_mlp_set_cu_budget(m._plan, 128))
        N.check(N.lib.lipasr_mlp_set_gemm_tiles(m._plan, 128))  # (as the pipeline sets it on a CU share)
        c0 = (N.lib.lipasr_debug_launch_count(0), N.lib.lipasr_debug_launch_count(1))
        m.train_fwd_bwd(x, y)
        c1 = (N.lib.lipasr_debug_launch_count(0), N.lib.lipasr_debug_launch_count(1))
        if mode == 1:
            assert c1[0] - c0[0] == 2, "layer 1 forward and the input gradient into layer 1 were meant to take the 128 x 64 exchange tile"
        assert c1[1] - c0[1] == 1, "the weight gradients were meant to take the 128 x 128 split-pass tile"
        first = _train_state(m)
        m.apply_adam()
        after = (m._params.clone(), m._bnstate.clone())
        load_params(m, p)
        m._step.zero_()
        m.train_fwd_bwd(x, y)
        second = _train_state(m)
        for a, b in zip(first, second):
            assert torch.equal(a, b), f"mode {mode}: not reproducible"
        assert m.exchange_errors() == 0
        out[mode] = (first, after)
        m.close()
    (g0, s0, l0), (p0, b0) = out[0]
    (g1, s1, l1), (p1, b1) = out[1]
    scale = float(g0.abs().max())
    # (the two tilings add a k-step's products in different orders -- 64 x 64 tiles in two K halves, these tiles in one -- so the
    # fp32 accumulations differ at the 1e-6 level before any BatchNorm: a wider band than the same-tiling test above)
    assert float((g0 - g1).abs().max()) <= 5e-6 * scale + 1e-12, float((g0 - g1).abs().max()) / scale
    torch.testing.assert_close(s1, s0, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(l1[:batch], l0[:batch], rtol=2e-5, atol=1e-6)
    assert float(b1.sub(b0).abs().max()) <= 1e-5 * float(b0.abs().max())


def test_first_step_of_a_fresh_process_takes_the_exchange_tiles(cuda):
    """The exchange instances are chosen by an occupancy query that used to run BEFORE their first launch had raised the kernel's
    dynamic-LDS limit: the runtime answered 0, the answer was cached, and a fresh process trained through the launch chain until
    something else had launched those kernels (the tests above passed in this file and failed in the full suite).  One mode-2 step
    on a 128-CU budget in a NEW interpreter: both large layer-1 GEMMs on 128 x 64 exchange tiles, the weight gradients on split-pass
    tiles, no exchange error."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    code = (
        "import sys, numpy as np, torch\n"
        f"sys.path[:0] = [{here!r}, {os.path.join(os.path.dirname(here), 'asr-using-robust-nn_amd')!r}, {os.path.dirname(here)!r}]\n"
        "from helpers import build_model, dev, load_params\n"
        "from oracle import mlp_ref as P\n"
        "from lipasr import _native as N\n"
        "w = (880, 1024, 512, 256, 128, 64, 10)\n"
        "spec = [P.LayerSpec(w[i], w[i + 1], i + 2 < len(w), 0.0, True) for i in range(len(w) - 1)]\n"
        "p = P.init_params(spec, seed=1, dtype=np.float32, nonneg_init=True)\n"
        "rng = np.random.default_rng(0)\n"
        "x = dev(rng.standard_normal((1024, 880)).astype(np.float32)); y = dev(P.to_categorical(rng.integers(0, 10, 1024), 10))\n"
        "m = build_model(spec, max_batch=1024, compute_dtype='float16x2'); load_params(m, p)\n"
        "N.check(N.lib.lipasr_mlp_set_cu_budget(m._plan, 128)); N.check(N.lib.lipasr_mlp_set_gemm_tiles(m._plan, 128))\n"
        "m.train_fwd_bwd(x, y); torch.cuda.synchronize()\n"
        "print('COUNTS', N.lib.lipasr_debug_launch_count(0), N.lib.lipasr_debug_launch_count(1), m.exchange_errors())\n"
    )
    env = {k: v for k, v in os.environ.items() if k not in ("LIPASR_FUSE_BN", "LIPASR_GEMM_MODE", "LIPASR_GEMM_TILES")}  # (A/B knobs that change the tile choice)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("COUNTS")][-1].split()
    assert line[1:] == ["2", "1", "0"], out.stdout


def test_batchnorm_inside_the_gemm_over_thirty_steps(cuda):
    """VERDICT r4 item 1's bar for a restructured step: parameters within 1e-6 (of the largest weight) of the launch-chain path
    after 30 training steps with the constraint, dropout on."""
    from lipasr import _native as N
    from lipasr.Constraints import simple_norm_constraint

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=21, dtype=np.float32, nonneg_init=True)
    rng = np.random.default_rng(8)
    xs = dev(rng.standard_normal((3 * 256, 880)).astype(np.float32))
    ys = dev(P.to_categorical(rng.integers(0, 10, 3 * 256), 10))
    res = []
    for mode in (0, 1):
        m = build_model(spec, max_batch=256)
        load_params(m, p)
        N.check(N.lib.lipasr_mlp_set_fuse_bn(m._plan, mode))
        cst = simple_norm_constraint(0.1, [])
        cst.set_model(m)
        for i in range(30):
            s = (i % 3) * 256
            m.train_fwd_bwd(xs[s:s + 256], ys[s:s + 256])
            m.apply_adam()
            cst.on_batch_end(i)
        assert m.exchange_errors() == 0
        res.append((m._params.clone(), m._bnstate.clone()))
        m.close()
    wmax = float(res[0][0].abs().max())
    d = float((res[0][0] - res[1][0]).abs().max())
    print(f"\nparameters after 30 steps: max |fused - chain| = {d:.3e} ({d / wmax:.2e} of the largest weight)")
    assert d <= 1e-6 * wmax + 1e-9, (d, wmax)
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * float(res[0][1].abs().max())


# ------------------------------------------------------------------------------------------------
# Round 5: fp16 two-plane split arithmetic (lipasr_mlp_set_compute(plan, 2), Model(compute_dtype="float16x2"))
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("batch", [1024, 182])
def test_fp16_two_plane_mode_is_fp32_accurate(cuda, batch):
    """Every GEMM operand split into two fp16 planes (hi = RNE(x), lo = RNE(x - hi)), three of the four cross terms on
    v_mfma_f32_32x32x16_f16, fp32 accumulation: products of fp16 numbers are exact in fp32, the error is the 2^-22 of the
    representation and the dropped lo lo term -- the resampler's and the STFT's arithmetic, now for the classifier's GEMMs.
    Held to the SAME bounds against the float64 oracle as the exact-fp32 mode in test_full_model_forward_backward (gradients 5e-5
    of the tensor's maximum, loss 1e-5), and the two modes are compared with each other: logits within 2e-6 relative of the
    largest logit, gradients within 5e-5 -- two orders inside BASELINE's 1e-3, where the bf16 mode is 1e-2."""
    spec = P.vd_constrained_spec()
    p = _random_state(spec, 1)
    rng = np.random.default_rng(batch)
    x = rng.standard_normal((batch, 880)).astype(np.float32)
    y = P.to_categorical(rng.integers(0, 10, batch), 10)
    masks = [((rng.uniform(size=(batch, s.n_out)) > s.dropout) / (1 - s.dropout)).astype(np.float32) if s.dropout > 0 else None for s in spec]
    ref = P.forward_backward(spec, p.astype(np.float64), x.astype(np.float64), y.astype(np.float64), masks=masks, training=True)
    got, logits = {}, {}
    for mode in ("float32", "float16x2"):
        K.reset_layer_names()
        m = _build_with_dtype(spec, mode, max_batch=batch)
        load_params(m, p)
        m.train_fwd_bwd(dev(x), dev(y), masks=[dev(k) if k is not None else None for k in masks])
        got[mode] = grads_of(m, spec)
        assert abs(float(m._loss_rows[:batch].mean()) - ref["loss"]) < 1e-5 * max(1.0, abs(ref["loss"]))
        logits[mode] = m.predict_device(dev(x), logits=True).cpu().numpy()
        assert m.exchange_errors() == 0
        m.close()
    for l in range(6):
        for k in ("dW", "db"):
            assert rel_err(got["float16x2"][k][l], ref[k][l]) < 5e-5, (k, l, rel_err(got["float16x2"][k][l], ref[k][l]))
            assert rel_err(got["float16x2"][k][l], got["float32"][k][l]) < 5e-5, (k, l)
        if spec[l].bn:
            assert rel_err(got["float16x2"]["dgamma"][l], ref["dgamma"][l]) < 5e-5 and rel_err(got["float16x2"]["dbeta"][l], ref["dbeta"][l]) < 5e-5
    d = np.abs(logits["float16x2"] - logits["float32"]).max() / np.abs(logits["float32"]).max()
    print(f"\nfloat16x2 against float32, batch {batch}: logits {d:.2e} of the largest logit; dW0 {rel_err(got['float16x2']['dW'][0], got['float32']['dW'][0]):.2e}")
    assert d < 2e-6 and (logits["float16x2"].argmax(1) == logits["float32"].argmax(1)).all()


def _build_with_dtype(spec, compute_dtype, max_batch=1024):
    from lipasr import keras as K2

    inp = K2.Input((spec[0].n_in,))
    node = inp
    for i, s in enumerate(spec):
        last = i == len(spec) - 1
        node = K2.Dense(s.n_out, activation="softmax" if last else "relu", kernel_constraint=K2.NonNeg() if s.nonneg else None)(node)
        if not last and s.bn:
            node = K2.BatchNormalization()(node)
        if not last and s.dropout > 0:
            node = K2.Dropout(s.dropout)(node)
    m = K2.Model(inputs=inp, outputs=node, max_batch=max_batch, compute_dtype=compute_dtype)
    m.compile(optimizer="adam", loss=K2.CategoricalCrossentropy(), metrics=["accuracy"])
    return m


def _gemm_split(a, b, ta, tb, sa=16.0, sb=16.0):
    from lipasr import _native as N

    h = N.get_handle(0)
    A = dev(a.T if ta else a)
    B = dev(b.T if tb else b)
    M, K_ = a.shape
    Nn = b.shape[1]
    out = torch.full((M, Nn), float("nan"), device="cuda")
    N.check(N.lib.lipasr_gemm_f16x2(h.h, int(ta), int(tb), M, Nn, K_, N.ptr(A), A.shape[1], N.ptr(B), B.shape[1], N.ptr(out), Nn, sa, sb, N.stream_ptr()))
    return out.cpu().numpy()


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("shape", [(1024, 1024, 896), (1024, 512, 1024), (512, 256, 64), (192, 128, 32), (1024, 64, 128),   # LDS-DMA ring kernel
                                   (1024, 1024, 880), (256, 128, 100), (128, 64, 36),                                        # ... with a K tail (zeros source)
                                   (512, 1022, 880), (182, 10, 64), (33, 47, 21), (100, 72, 50), (512, 880, 10)])              # the other tiles
def test_gemm_f16x2_all_layouts(cuda, ta, tb, shape):
    """lipasr_gemm_f16x2: every layout on both kernels of arithmetic mode 2 -- the LDS-DMA ring (K-contiguous operands through the
    XOR-swizzled [i][32 k] slot image, k-major operands through [32 k][64 i]) and the register-staged / fragment tiles for shapes
    the ring does not take.  Small integers are exact (they are fp16 numbers: catches a wrong swizzle or lane map exactly);
    random operands are held to 6e-7 of sum |a||b| per element -- the two-plane split's 2^-21 per product."""
    M, N_, K_ = shape
    rng = np.random.default_rng(M + 7 * N_ + 13 * K_)
    ai = rng.integers(-3, 4, (M, K_)).astype(np.float32)
    bi = rng.integers(-3, 4, (K_, N_)).astype(np.float32)
    np.testing.assert_array_equal(_gemm_split(ai, bi, ta, tb), ai.astype(np.float64) @ bi.astype(np.float64))
    a = rng.standard_normal((M, K_)).astype(np.float32)
    b = rng.standard_normal((K_, N_)).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    bound = 6e-7 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-30
    got = _gemm_split(a, b, ta, tb)
    assert np.all(np.abs(got - ref) <= bound), float((np.abs(got - ref) / bound).max())
