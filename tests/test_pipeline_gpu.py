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
    ckpt = str(tmp_path / "bin" / "models_constrained" / "TEST.h5")
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


@pytest.mark.parametrize("compute", ["float32", "float16x2"])
def test_pipeline_matches_oracle_training_steps(cuda, compute):
    """(compute: the arithmetic of the training GEMMs -- exact fp32 chains, or round 5's fp16 two-plane split, held to the same bounds.)
    3 end-to-end steps (MFCC + fused standardisation -> train step -> simple_norm_constraint) against the oracle,
    dropout off.  The features are standardised (as train_constraints.py:28-35 does before fit): raw MFCCs are mostly
    negative, and behind non-negative kernels they leave every ReLU dead and every gradient exactly zero -- which is what
    this test compared in round 2."""
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import mfcc_ref as M

    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    p = P.init_params(spec, seed=10, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(96, seed=41)
    y = P.to_categorical(labels, 10)
    ref_feats = M.compute_mfcc_batch(waves)
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_std = (ref_feats - mean) / scale
    m = build_model(spec, max_batch=32, compute_dtype=compute)
    load_params(m, p)
    pipe = TrainPipeline(m, batch=32, rho=0.1, constraint="product", use_graph=True,
                         affine=(torch.as_tensor(mean).cuda(), torch.as_tensor(scale).cuda()))
    p64, st = p.astype(np.float64), P.AdamState()
    solid = [None] * 6
    for s in range(0, 96, 32):
        pipe.step(dev(waves[s:s + 32]), dev(y[s:s + 32]))
        pipe.synchronize()
        feats = pipe.feats.cpu().numpy().astype(np.float64)
        assert np.abs(feats - ref_std[s:s + 32]).max() < 2e-3  # standardised units
        # the oracle steps from the checked device features: Adam turns a sign flip of a ~1e-7 gradient
        # into a 1e-3 move, which would measure MFCC rounding rather than the step's arithmetic
        out = P.train_step(spec, p64, st, feats, y[s:s + 32].astype(np.float64))
        # Adam's first steps move a weight by ~lr * sign(g): where |g| is at rounding level its sign -- and with it a
        # 1e-3-sized move -- is not determined.  Entries whose oracle gradient stayed clear of that level in every step
        # are held to the tight bound, the undetermined rest only to the size of the moves themselves.
        for l in range(6):
            g = np.abs(out["dW"][l])
            assert g.max() > 0, l  # a live network
            ok = g > 1e-3 * g.max()  # GPU dW agrees to 5e-5 of the max (test_mlp_gpu): >= 20x clear of a sign flip
            solid[l] = ok if solid[l] is None else (solid[l] & ok)
        new_w, norms = R.simple_norm_constraint_pass([w.astype(np.float32) for w in p64.W], 0.1, [])
        p64.W = [w.astype(np.float64) for w in new_w]
    pipe.synchronize()
    after = read_params(m, spec)
    np.testing.assert_allclose(pipe.norms.cpu().numpy(), norms, rtol=2e-3)
    for l in range(6):
        d = np.abs(after.W[l] - p64.W[l]) / np.abs(p64.W[l]).max()
        print(f"layer {l}: solid fraction {solid[l].mean():.3f}, max rel diff on solid {d[solid[l]].max() if solid[l].any() else 0:.2e}, overall q999 {np.quantile(d, 0.999):.2e} max {d.max():.2e}")
        assert solid[l].mean() > 0.05, (l, solid[l].mean())
        assert d[solid[l]].max() < 2e-3, (l, d[solid[l]].max())
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


def test_adversarial_training_steps_match_the_composed_oracle(cuda):
    """BASELINE config 5 as a whole: three adversarial-training steps at batch 32, dropout off --
    waveform -> MFCC + standardisation -> PGD-20 (eps 0.5, eps_step 0.1, the update rule of ART's ProjectedGradientDescent,
    attacks.py:647-661, driven from the training loop with the BATCH's labels, inference-mode network) -> training step on x_adv
    -> Adam + NonNeg -> simple_norm_constraint -- against the same composition of the oracle's pieces:
    oracle.attacks_ref.pgd -> oracle.mlp_ref.train_step -> oracle.constraints_ref.simple_norm_constraint_pass.
    x_adv is compared by mass, as test_pgd_matches_oracle does (a near-zero gradient component may flip its sign in fp32 and move
    that one coordinate by up to 2 eps_step per iteration); the oracle then takes ITS training step from the device's x_adv, so
    that what the parameter comparison measures is the step's arithmetic and not those coordinates, and is held to the tolerances
    of test_pipeline_matches_oracle_training_steps."""
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips
    from oracle import attacks_ref as A, mfcc_ref as M

    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    p = P.init_params(spec, seed=10, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(96, seed=41)
    y = P.to_categorical(labels, 10)
    ref_feats = M.compute_mfcc_batch(waves)
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_std = (ref_feats - mean) / scale
    m = build_model(spec, max_batch=32)
    load_params(m, p)
    eps, eps_step, iters = 0.5, 0.1, 20
    pipe = TrainPipeline(m, batch=32, rho=0.1, constraint="product", use_graph=True, pgd=dict(eps=eps, eps_step=eps_step, max_iter=iters),
                         affine=(torch.as_tensor(mean).cuda(), torch.as_tensor(scale).cuda()))
    p64, st = p.astype(np.float64), P.AdamState()
    solid = [None] * 6
    for k, s in enumerate(range(0, 96, 32)):
        yb = y[s:s + 32].astype(np.float64)
        pipe.step(dev(waves[s:s + 32]), dev(y[s:s + 32]))
        pipe.synchronize()
        feats = pipe.feats.cpu().numpy().astype(np.float64)
        x_adv = pipe.x_adv.cpu().numpy().astype(np.float64)
        assert np.abs(feats - ref_std[s:s + 32]).max() < 2e-3
        assert np.abs(x_adv - feats).max() <= eps + 1e-5
        # the attack, from the oracle's parameters as they stand before this step's update
        ref_adv = A.pgd(spec, p64, feats, eps, eps_step, iters, 32, y=yb)
        agree = np.abs(x_adv - ref_adv) < 1e-4
        assert agree.mean() > 0.97, (k, agree.mean())

        def loss_at(z):
            return P.forward_backward(spec, p64, z, yb, training=False)["loss"]

        l_dev, l_ref, l_clean = loss_at(x_adv), loss_at(ref_adv), loss_at(feats)
        # (after two projected steps this network's inference-mode output no longer depends on its input -- BatchNorm's moving
        # statistics lag the batch statistics by design, momentum 0.99 -- so the attack has nothing to climb: equality is allowed
        # from the second step on, the first step must be a real ascent)
        assert abs(l_dev - l_ref) < 2e-2 * max(1.0, abs(l_ref)) and l_dev >= l_clean and (k > 0 or l_dev > l_clean), (k, l_dev, l_ref, l_clean)
        # the training step on x_adv, then the callback
        out = P.train_step(spec, p64, st, x_adv, yb)
        for l in range(6):
            g = np.abs(out["dW"][l])
            assert g.max() > 0, l
            ok = g > 1e-3 * g.max()
            solid[l] = ok if solid[l] is None else (solid[l] & ok)
        new_w, norms = R.simple_norm_constraint_pass([w.astype(np.float32) for w in p64.W], 0.1, [])
        p64.W = [w.astype(np.float64) for w in new_w]
    after = read_params(m, spec)
    assert int(m._step.item()) == 3
    np.testing.assert_allclose(pipe.norms.cpu().numpy(), norms, rtol=2e-3)
    for l in range(6):
        d = np.abs(after.W[l] - p64.W[l]) / np.abs(p64.W[l]).max()
        print(f"layer {l}: solid fraction {solid[l].mean():.3f}, max rel diff on solid {d[solid[l]].max() if solid[l].any() else 0:.2e}, q999 {np.quantile(d, 0.999):.2e} max {d.max():.2e}")
        assert solid[l].mean() > 0.05, (l, solid[l].mean())
        assert d[solid[l]].max() < 2e-3, (l, d[solid[l]].max())
        assert np.quantile(d, 0.999) < 2e-3 and d.max() < 5e-2, (l, np.quantile(d, 0.999), d.max())
    for l in range(5):  # BatchNorm moving statistics saw the adversarial batch on both sides
        mm, mv = after.mov_mean[l], after.mov_var[l]
        np.testing.assert_allclose(mm, p64.mov_mean[l], rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(mv, p64.mov_var[l], rtol=2e-3, atol=2e-5)
    pipe.close()


def test_training_accuracy_parity(cuda):
    """BASELINE's accuracy statement: same data, same order, same init -> the GPU-trained and the oracle-trained
    classifier agree on held-out top-1 accuracy within +-0.5 pt.  Checked to convergence on the unconstrained
    model (train_google_dataset.py's, which learns the synthetic task in 10 epochs); the constrained model
    (rho = 0.1 needs thousands of epochs, as in the reference) is compared after 40 steps on loss, product norm
    and prediction agreement.  Dropout masks cannot match TensorFlow's or NumPy's: deterministic runs use
    dropout 0, the GPU's own Philox-dropout run is checked statistically."""
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.attacks import standardize_dataset
    from lipasr.extract_features_construct_dataset import mfcc
    from lipasr.keras import Dataset
    from lipasr.synth import synth_clips

    waves, labels = synth_clips(1536, seed=71)
    feats = np.concatenate([mfcc(waves[s:s + 512]).cpu().numpy() for s in range(0, 1536, 512)]).astype(np.float64)
    tr, _, te = standardize_dataset(feats[:1024], feats[1024:1025], feats[1024:])
    ytr, yte = P.to_categorical(labels[:1024], 10), labels[1024:]
    ds = Dataset.from_tensor_slices((tr, ytr)).batch(128)

    # ---- A: unconstrained model to convergence
    spec_u = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, False) for s in P.vd_unconstrained_spec()]
    pu = P.init_params(spec_u, seed=12, dtype=np.float32)
    m = build_model(spec_u, max_batch=128)
    load_params(m, pu)
    m.fit(ds, epochs=10, verbose=0)
    acc_gpu = float(np.mean(m.predict(te).argmax(1) == yte))
    p64, st = pu.astype(np.float64), P.AdamState()
    for _ in range(10):
        for s in range(0, 1024, 128):
            P.train_step(spec_u, p64, st, tr[s:s + 128], ytr[s:s + 128].astype(np.float64))
    acc_ref = float(np.mean(P.forward_infer(spec_u, p64, te).argmax(1) == yte))
    assert acc_ref > 0.95 and acc_gpu > 0.95, (acc_gpu, acc_ref)
    assert abs(acc_gpu - acc_ref) <= 0.005, (acc_gpu, acc_ref)

    # ---- B: constrained model (NonNeg + simple_norm_constraint rho = 0.1), 40 deterministic steps
    spec_c = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    pc = P.init_params(spec_c, seed=12, dtype=np.float32, nonneg_init=True)
    mc = build_model(spec_c, max_batch=128)
    load_params(mc, pc)
    cst = simple_norm_constraint(0.1, [])
    hist = mc.fit(ds, epochs=5, verbose=0, callbacks=[cst])
    p64, st = pc.astype(np.float64), P.AdamState()
    losses = []
    for _ in range(5):
        ep = []
        for s in range(0, 1024, 128):
            ep.append(P.train_step(spec_c, p64, st, tr[s:s + 128], ytr[s:s + 128].astype(np.float64))["loss"])
            new_w, norms = R.simple_norm_constraint_pass([w.astype(np.float32) for w in p64.W], 0.1, [])
            p64.W = [w.astype(np.float64) for w in new_w]
        losses.append(float(np.mean(ep)))
    np.testing.assert_allclose(hist["loss"], losses, rtol=2e-3)
    assert abs(float(cst.last_norms[-1]) - norms[-1]) / norms[-1] < 2e-3
    agree = np.mean(mc.predict(te).argmax(1) == P.forward_infer(spec_c, p64, te).argmax(1))
    assert agree >= 0.97, agree

    # ---- C: the GPU's own dropout (reference rates 0.1): same ball park as the deterministic run
    md = build_model(P.vd_constrained_spec(), max_batch=128)
    load_params(md, pc)
    hd = md.fit(ds, epochs=5, verbose=0, callbacks=[simple_norm_constraint(0.1, [])])
    assert np.isfinite(hd["loss"]).all() and abs(hd["loss"][-1] - losses[-1]) < 0.15 * losses[-1] + 0.05


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_accuracy_parity_three_seeds(cuda, seed):
    """top-1 accuracy within +-0.5 pt, over three seeds (data, split and initialisation all change with the
    seed).  (i) dropout off: GPU and oracle run the same arithmetic, so the accuracies must agree to the half point;
    (ii) the reference's dropout (0.4 on every hidden block): the GPU draws Philox masks, the oracle NumPy masks --
    different streams, as TensorFlow's would be -- and both must land in the same place."""
    from lipasr.attacks import standardize_dataset
    from lipasr.extract_features_construct_dataset import mfcc
    from lipasr.keras import Dataset
    from lipasr.synth import synth_clips

    waves, labels = synth_clips(1536, seed=100 + seed)
    feats = np.concatenate([mfcc(waves[s:s + 512]).cpu().numpy() for s in range(0, 1536, 512)]).astype(np.float64)
    tr, _, te = standardize_dataset(feats[:1024], feats[1024:1025], feats[1024:])
    ytr, yte = P.to_categorical(labels[:1024], 10), labels[1024:]
    ds = Dataset.from_tensor_slices((tr, ytr)).batch(128)
    accs = {}
    for drop in (0.0, 0.4):
        spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, drop if i < 5 else 0.0, False) for i, s in enumerate(P.vd_unconstrained_spec())]
        p0 = P.init_params(spec, seed=seed, dtype=np.float32)
        m = build_model(spec, max_batch=128, seed=seed)
        load_params(m, p0)
        m.fit(ds, epochs=12, verbose=0)
        acc_gpu = float(np.mean(m.predict(te).argmax(1) == yte))
        p64, st = p0.astype(np.float64), P.AdamState()
        rng = np.random.default_rng(seed)
        for _ in range(12):
            for s in range(0, 1024, 128):
                masks = None
                if drop > 0:
                    masks = [((rng.uniform(size=(128, sp.n_out)) > sp.dropout) / (1 - sp.dropout)) if sp.dropout > 0 else None for sp in spec]
                P.train_step(spec, p64, st, tr[s:s + 128], ytr[s:s + 128].astype(np.float64), masks=masks)
        acc_ref = float(np.mean(P.forward_infer(spec, p64, te).argmax(1) == yte))
        accs[drop] = (acc_gpu, acc_ref)
    assert abs(accs[0.0][0] - accs[0.0][1]) <= 0.005, accs
    assert min(accs[0.4]) > 0.9 and abs(accs[0.4][0] - accs[0.4][1]) <= 0.02, accs  # independent mask streams: 512 test clips, 1 clip = 0.2 pt


def test_pgd_pipeline_is_not_slower_as_a_later_pipeline_of_the_process(cuda):
    """VERDICT r3 item 5.  Round 3: the PGD-20 graph replayed 3x slower (12.5 against 4.2 ms) as the FIFTH pipeline of one
    process.  Cause (round 4, scratch/pgd_fifth_probe.py): the training stream was a stream of torch's pool, and pool streams
    are multiplexed over GPU_MAX_HW_QUEUES = 4 hardware queues; once enough streams exist it shared a queue with other work and
    its ~440 dependent kernel nodes waited node by node.  TrainPipeline now makes that stream with a full CU mask, which owns a
    hardware queue.  Here: a PGD pipeline timed first, then again after the four kinds of pipeline bench.py builds before it
    (two CU partitions, pre-extracted fp32 / bf16) have come and gone: 4.08 against 4.05 ms when written; asserted within 30 %
    (the fault was a factor of three; a shared box moves a 4 ms step by a few per cent)."""
    import time

    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.keras import CategoricalCrossentropy
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips_device
    from lipasr.train_constraints import get_model

    device = torch.device("cuda", 0)
    B = 1024
    waves, labels = synth_clips_device(4 * B, 5, device)
    y = torch.zeros(4 * B, 10, device=device)
    y[torch.arange(4 * B, device=device), labels] = 1
    ex = MfccExtractor(16000, 16000, B, device)
    feats = torch.cat([ex(waves[i * B:(i + 1) * B]) for i in range(2)])
    sc = StandardScaler().fit(feats)
    feat_std = (feats - sc.mean_.float()) / sc.scale_.float()
    ex.close()

    def run(batch, pgd=None, pre=False, bf16=False, steps=12):
        m = get_model(max_batch=batch, seed=0, compute_dtype="bfloat16" if bf16 else "float32")
        m.compile(optimizer="adam", loss=CategoricalCrossentropy(), metrics=["accuracy"])
        pipe = TrainPipeline(m, batch=batch, rho=0.1, constraint="product", affine=(sc.mean_, sc.scale_), pgd=pgd, sync_inputs=False,
                             mfcc_cus=None if pre else "auto")
        nb = (2 * B) // batch if pre else (4 * B) // batch

        def one(i):
            s = (i % nb) * batch
            if pre:
                pipe.step(None, y[s:s + batch], features=feat_std[s:s + batch])
            else:
                pipe.step(waves[s:s + batch], y[s:s + batch])

        for i in range(4):
            one(i)
        pipe.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            one(4 + i)
        pipe.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        pipe.close()
        m.close()
        return dt

    pgd = dict(eps=0.5, eps_step=0.1, max_iter=20)
    first = run(B, pgd=pgd)
    others = [run(B), run(512), run(B, pre=True), run(B, pre=True, bf16=True)]
    fifth = run(B, pgd=pgd)
    print(f"\nPGD-20 step: {first:.3f} ms as the first pipeline, {fifth:.3f} ms after four others ({[round(o, 3) for o in others]})")
    assert fifth <= 1.30 * first, (first, fifth)


def test_device_flag_handoffs_equal_event_handoffs_and_report_a_missing_signal(cuda, monkeypatch):
    """Round 4: the two hand-offs between the extraction stream and the classifier stream are device-side counters
    (lipasr_flag_signal / lipasr_flag_wait) by default.  Same kernels in the same order on each stream => the same bits as with
    hipEventRecord + hipStreamWaitEvent (LIPASR_GPU_FLAGS=0), also when a buffer is reused many times.  And the wait is bounded:
    a counter nobody raises makes lipasr_flag_wait give up after its timeout and set the error word instead of hanging the queue;
    a counter that is raised later on another stream releases it."""
    import time

    from lipasr import _native as N
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips

    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(192, seed=31)
    wt = dev(waves)
    yt = dev(P.to_categorical(labels, 10))
    for pgd in (None, dict(eps=0.5, eps_step=0.1, max_iter=5)):  # (PGD: the classifier's stream keeps every CU, graphs)
        results = []
        for flags in ("0", "1"):
            monkeypatch.setenv("LIPASR_GPU_FLAGS", flags)
            m = build_model(spec, max_batch=64)
            load_params(m, p)
            pipe = TrainPipeline(m, batch=64, rho=0.1, constraint="product", sync_inputs=False, pgd=pgd)
            assert (pipe._flags is not None) == (flags == "1")
            for rep in range(4):
                for s in range(0, 192, 64):
                    pipe.step(wt[s:s + 64], yt[s:s + 64])
            pipe.synchronize()
            results.append((m._params.clone(), m._bnstate.clone(), pipe.norms.clone(), int(m._step.item())))
            pipe.close()
        assert results[0][3] == results[1][3] == 12
        for a, b in zip(results[0][:3], results[1][:3]):
            assert torch.equal(a, b)

    # ---- the primitive alone, on two CU-masked streams: each owns a hardware queue, so the waiter can never sit in FRONT of the
    # signaller in one queue (two streams of torch's pool can: they are multiplexed over 4 hardware queues -- DESIGN.md 3 --
    # and round 4's version of this test used them)
    import ctypes as C

    h = N.get_handle(0)
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (n_cu + 31) // 32

    def masked(lo, hi):
        mask = (C.c_uint32 * words)()
        for b in range(lo, hi):
            mask[b // 32] |= 1 << (b % 32)
        st = N.c_s()
        N.check(N.lib.lipasr_stream_create_masked(h.h, mask, words, C.byref(st)))
        return st, torch.cuda.ExternalStream(st.value, device=torch.device("cuda", 0))

    raw1, s1 = masked(0, n_cu // 2)
    raw2, s2 = masked(n_cu // 2, n_cu)
    flag = torch.zeros(1, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32).pin_memory()  # the report word in pinned host memory, as the pipeline keeps it
    torch.cuda.synchronize()
    try:
        # (1) raised later, on the other stream: the waiting stream goes on, no report
        with torch.cuda.stream(s1):
            N.check(N.lib.lipasr_flag_wait(h.h, flag.data_ptr(), 3, 20000, err.data_ptr(), N.stream_ptr()))
            after = torch.ones(1, device="cuda") * 2
        time.sleep(0.05)
        assert not s1.query()
        with torch.cuda.stream(s2):
            N.check(N.lib.lipasr_flag_signal(h.h, flag.data_ptr(), 3, N.stream_ptr()))
        s1.synchronize()
        assert flag.tolist() == [3] and int(err[0]) == 0 and float(after.item()) == 2.0
        # (2) raised AFTER the timeout: the wait reports (1) without synchronising anything and is STILL waiting -- nothing behind
        # it has run -- until the signal comes (round 4 let the stream go on at the timeout)
        with torch.cuda.stream(s1):
            N.check(N.lib.lipasr_flag_wait(h.h, flag.data_ptr(), 4, 50, err.data_ptr(), N.stream_ptr()))
            after2 = torch.ones(1, device="cuda") * 5
        t0 = time.perf_counter()
        while int(err[0]) == 0 and time.perf_counter() - t0 < 5.0:
            time.sleep(0.005)
        assert int(err[0]) == 1, "the overdue wait was not reported to the pinned word"
        assert not s1.query(), "the wait let its stream go on at the soft timeout"
        with torch.cuda.stream(s2):
            N.check(N.lib.lipasr_flag_signal(h.h, flag.data_ptr(), 4, N.stream_ptr()))
        s1.synchronize()
        assert int(err[0]) == 1 and float(after2.item()) == 5.0
        err[0] = 0
        # (3) never raised: after 4 x the timeout it gives up (2), so the queue drains instead of hanging
        t0 = time.perf_counter()
        with torch.cuda.stream(s1):
            N.check(N.lib.lipasr_flag_wait(h.h, flag.data_ptr(), 9, 50, err.data_ptr(), N.stream_ptr()))
        s1.synchronize()
        dt = time.perf_counter() - t0
        assert int(err[0]) == 2 and 0.19 < dt < 3.0, (int(err[0]), dt)
        assert N.lib.lipasr_flag_wait(h.h, flag.data_ptr(), 4, 0, err.data_ptr(), N.stream_ptr()) == N.EINVAL
    finally:
        torch.cuda.synchronize()
        N.check(N.lib.lipasr_stream_destroy(h.h, raw1))
        N.check(N.lib.lipasr_stream_destroy(h.h, raw2))


def test_pipeline_reports_a_missing_handoff_without_synchronize_and_still_closes(cuda, monkeypatch):
    """VERDICT r4 item 7 / ADVICE r4: a device-side wait whose signal never comes must (a) reach a caller that never calls
    pipe.synchronize() -- the next step() raises from the pinned report word --, (b) not be raised for ever (the word is cleared
    when it is raised) and (c) leave close() able to drain and give the two hardware queues back."""
    import time

    from lipasr import _native as N
    from lipasr.pipeline import TrainPipeline
    from lipasr.synth import synth_clips

    monkeypatch.setenv("LIPASR_GPU_FLAGS", "1")
    monkeypatch.setenv("LIPASR_FLAG_TIMEOUT_MS", "40")
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=True)
    waves, labels = synth_clips(64, seed=5)
    wt, yt = dev(waves), dev(P.to_categorical(labels, 10))
    m = build_model(spec, max_batch=64)
    load_params(m, p)
    pipe = TrainPipeline(m, batch=64, rho=0.1, constraint="product", sync_inputs=False)
    assert pipe._flags is not None and pipe._masked_stream is not None and pipe._masked_train_stream is not None
    pipe.step(wt, yt)
    pipe.synchronize()
    real_signal = N.lib.lipasr_flag_signal
    monkeypatch.setattr(N.lib, "lipasr_flag_signal", lambda *a: 0)  # this step's "features ready" / "buffer free" never go out
    pipe.step(wt, yt)
    monkeypatch.setattr(N.lib, "lipasr_flag_signal", real_signal)
    t0 = time.perf_counter()
    with pytest.raises(RuntimeError, match="device-side wait"):
        while time.perf_counter() - t0 < 5.0:   # no synchronize(): the report arrives through the pinned word
            time.sleep(0.02)
            pipe.step(wt, yt)
    assert getattr(pipe, "_failed", False)
    try:
        pipe.close()                             # may re-raise a second report (the wait has given up by now) ...
    except RuntimeError as e:
        assert "device-side wait" in str(e)
    assert pipe._closed and pipe._masked_stream is None and pipe._masked_train_stream is None and not pipe._graphs  # ... after the teardown
    m.close()
    # the device and the handle are fine afterwards: a new pipeline trains
    m2 = build_model(spec, max_batch=64)
    load_params(m2, p)
    pipe2 = TrainPipeline(m2, batch=64, rho=0.1, constraint="product", sync_inputs=False)
    pipe2.step(wt, yt)
    pipe2.synchronize()
    assert int(m2._step.item()) == 1
    pipe2.close()
    m2.close()
