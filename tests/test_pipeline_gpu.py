"""GPU: the drop-in driver (reads like train_constraints.py:94-105) and the fused training pipeline
(waveform -> MFCC -> fwd/bwd -> Adam+NonNeg -> projection), eager vs HIP-graph replay."""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, load_params, read_params
from oracle import constraints_ref as R, mlp_ref as P

pytestmark = pytest.mark.gpu


def test_driver_reads_like_the_reference(cuda, tmp_path, capsys):
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.attacks import standardize_dataset
    from lipasr.extract_features_construct_dataset import get_lipschitz_constrained, mfcc
    from lipasr.keras import CategoricalCrossentropy, Dataset, EarlyStopping, ModelCheckpoint, load_model, to_categorical
    from lipasr.synth import synth_clips
    from lipasr.train_constraints import get_model, lip_stats_callback

    waves, labels = synth_clips(768, seed=21)
    feats = mfcc(waves[:512]).cpu().numpy().astype(np.float64)
    feats = np.concatenate([feats, mfcc(waves[512:]).cpu().numpy().astype(np.float64)])
    train_data, val_data, test_data = standardize_dataset(feats[:512], feats[512:640], feats[640:])
    train_label, val_label, test_label = (to_categorical(l, 10) for l in (labels[:512], labels[512:640], labels[640:]))
    train_dataset = Dataset.from_tensor_slices((train_data, train_label)).shuffle(880, reshuffle_each_iteration=False).batch(128)
    val_dataset = Dataset.from_tensor_slices((val_data, val_label)).shuffle(880, reshuffle_each_iteration=False).batch(128)

    model = get_model()
    model.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
    ckpt = str(tmp_path / "bin" / "models_constrained" / "TEST.pt")
    cst = simple_norm_constraint(rho=0.1, affected_layers_indices=[])
    hist = model.fit(train_dataset, epochs=6, validation_data=val_dataset, verbose=2,
                     callbacks=[EarlyStopping(monitor="val_loss", patience=6000, restore_best_weights=False), cst, lip_stats_callback(),
                                ModelCheckpoint(ckpt, save_best_only=True, verbose=1)])
    out = capsys.readouterr().out
    assert "The Lipschitz constant on epoch 0 is" in out and "The norm for layer" in out
    assert hist["loss"][-1] < hist["loss"][0] and np.isfinite(hist["val_loss"]).all()
    # the constraint drives ||W6^T..W1^T|| to rho: after 24 batches the log-distance shrank by 0.335^24
    norms = cst.last_norms.cpu().numpy()
    assert abs(norms[-1] - 0.1) < 2e-2  # Adam moves the product norm between projections; it hovers just above rho
    ws = [l.get_weights()[0] for l in model.layers if "dense" in l.name]
    assert abs(R.sigma_max(R.product_chain(ws)) - norms[-1]) / norms[-1] < 1e-4
    assert all(w.min() >= 0 for w in ws)  # NonNeg held
    assert abs(get_lipschitz_constrained(model) - R.get_lipschitz_constrained(ws, [(l.get_weights()[0], l.get_weights()[3]) for l in model.layers if "batch" in l.name])) < 1e-3 * norms[-1] + 1e-6
    model2 = load_model(ckpt)
    y = np.argmax(model2.predict(test_data), axis=1)
    results = model2.evaluate(test_data, test_label)
    assert y.shape == (128,) and np.isfinite(results[0])


def test_pipeline_graph_equals_eager(cuda):
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(192, seed=31)
    wt = dev(waves)
    yt = dev(P.to_categorical(labels, 10))
    results = []
    for use_graph in (False, True):
        m = build_model(spec, max_batch=64)
        load_params(m, p)
        pipe = TrainPipeline(m, batch=64, rho=0.1, constraint="product", use_graph=use_graph)
        for s in range(0, 192, 64):
            pipe.step(wt[s:s + 64], yt[s:s + 64])
        for s in range(0, 128, 64):  # graph replays on the second pass
            pipe.step(wt[s:s + 64], yt[s:s + 64])
        pipe.synchronize()
        results.append((m._params.clone(), m._bnstate.clone(), pipe.norms.clone(), int(m._step.item())))
    assert results[0][3] == results[1][3] == 5
    assert torch.equal(results[0][0], results[1][0])  # bitwise: same kernels, same order, no atomics
    assert torch.equal(results[0][1], results[1][1])
    assert torch.equal(results[0][2], results[1][2])


def test_pipeline_matches_oracle_training_steps(cuda):
    """3 end-to-end steps (MFCC -> train step -> simple_norm_constraint) against the oracle, dropout off."""
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mfcc_ref as M

    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    p = P.init_params(spec, seed=10, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(96, seed=41)
    y = P.to_categorical(labels, 10)
    m = build_model(spec, max_batch=32)
    load_params(m, p)
    pipe = TrainPipeline(m, batch=32, rho=0.1, constraint="product", use_graph=True)
    p64, st = p.astype(np.float64), P.AdamState()
    ref_feats = M.compute_mfcc_batch(waves)
    for s in range(0, 96, 32):
        pipe.step(dev(waves[s:s + 32]), dev(y[s:s + 32]))
        pipe.synchronize()
        feats = pipe.feats.cpu().numpy().astype(np.float64)
        assert np.abs(feats - ref_feats[s:s + 32]).max() < 2e-2
        # the oracle steps from the checked device features: Adam turns a sign flip of a ~1e-7 gradient
        # into a 1e-3 move, which would measure MFCC rounding rather than the step's arithmetic
        P.train_step(spec, p64, st, feats, y[s:s + 32].astype(np.float64))
        new_w, norms = R.simple_norm_constraint_pass([w.astype(np.float32) for w in p64.W], 0.1, [])
        p64.W = [w.astype(np.float64) for w in new_w]
    pipe.synchronize()
    after = read_params(m, spec)
    np.testing.assert_allclose(pipe.norms.cpu().numpy(), norms, rtol=2e-3)
    for l in range(6):
        d = np.abs(after.W[l] - p64.W[l]) / np.abs(p64.W[l]).max()
        assert np.quantile(d, 0.999) < 2e-3 and d.max() < 5e-2, (l, np.quantile(d, 0.999), d.max())


def test_pgd_adversarial_training_step(cuda):
    """Config 5's inner loop: PGD-20 on the standardised features inside the captured step."""
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips

    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import mfcc

    spec = P.vd_constrained_spec()
    m = build_model(spec, max_batch=32)
    load_params(m, P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True))
    waves, labels = synth_clips(64, seed=51)
    y = dev(P.to_categorical(labels, 10))
    sc = StandardScaler().fit(mfcc(waves))
    pipe = TrainPipeline(m, batch=32, rho=0.1, affine=(sc.mean_, sc.scale_), pgd=dict(eps=0.5, eps_step=0.1, max_iter=20), use_graph=True)
    before = m._params.clone()
    moved = []
    for s in (0, 32, 0):
        pipe.step(dev(waves[s:s + 32]), y[s:s + 32])
        pipe.synchronize()
        d = (pipe.x_adv - pipe.feats).abs()
        assert float(d.max()) <= 0.5 + 1e-5  # inside the eps ball around the clean features
        moved.append(float(d.max()))
    assert moved[0] > 0.3  # 20 steps of 0.1 saturate the ball on the untrained net
    assert not torch.equal(before, m._params) and torch.isfinite(m._params).all()
    assert int(m._step.item()) == 3
