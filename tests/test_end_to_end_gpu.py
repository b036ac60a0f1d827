"""GPU: BASELINE.json's parity statements checked END TO END, from the waveform (SURVEY 8d "Parity procedure").

    reference flow   wav -> librosa MFCC -> StandardScaler -> model.predict
                     (VD/attacks.py:339-345, VD/extract_features_construct_dataset.py:24-39, VD/train_constraints.py:107-111)
    oracle side      oracle.mfcc_ref (exact resampler) -> oracle standard_scaler -> oracle forward_infer, float64
    product side     lipasr MfccExtractor (K1) + fused StandardScaler affine -> Model.predict_device (K2), fp32

Nothing is shared between the two sides except the seeded waveforms and the seeded weights: each side extracts its own
features and fits its own scaler, so the MFCC stage's error budget is inside the logit comparison.
"""
import numpy as np
import pytest
import torch

from helpers import build_model, dev, load_params, oracle_mfcc_parallel
from oracle import mlp_ref as P

pytestmark = pytest.mark.gpu

N_TEST = 2366  # the reference's test split (VD/processed_google_dataset/test_label.npy)


@pytest.fixture(scope="module")
def clips():
    from lipasr.synth import synth_clips

    waves, labels = synth_clips(N_TEST, seed=2366)
    return waves, labels, oracle_mfcc_parallel(waves)


def _state(spec, seed):
    """Seeded constrained-model weights with non-trivial biases and BatchNorm state (a model in mid-training)."""
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


def _product_side(cuda, waves, m):
    """K1 MFCC -> StandardScaler fitted on the product's own features -> fused affine -> K2 logits (all on the device)."""
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor

    ex = MfccExtractor(16000, 16000, 1024, cuda)
    wt = dev(waves)
    raw = torch.cat([ex(wt[s:s + 1024]) for s in range(0, N_TEST, 1024)])
    sc = StandardScaler().fit(raw)
    feats = torch.cat([ex(wt[s:s + 1024], 44, sc.mean_, sc.scale_) for s in range(0, N_TEST, 1024)])
    ex.close()
    return raw, feats


def test_logits_from_waveform_2366_clips(cuda, clips):
    """BASELINE's statement on the reference's model -- the Lipschitz-CONSTRAINED classifier (train_constraints.py:63-105:
    NonNeg kernels, simple_norm_constraint(rho = 0.1) after every batch): per-utterance logits within 1e-3 relative fp32,
    argmax labels identical, waveform in, logits out.  The weights are the product's own after 320 constrained steps
    (what NonNeg, the projection and BatchNorm's moving statistics shape), read back through get_weights() and handed to
    the oracle, which runs ITS OWN feature extraction and standardisation on the same 2 366 waveforms."""
    from helpers import read_params
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    m = build_model(spec, max_batch=1024, seed=3)
    load_params(m, P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True))
    raw, feats = _product_side(cuda, waves, m)
    ds = Dataset.from_tensor_slices((feats[:2048].cpu().numpy(), P.to_categorical(labels[:2048], 10))).batch(128)
    cst = simple_norm_constraint(0.1, [])
    m.fit(ds, epochs=20, verbose=0, callbacks=[cst])
    assert abs(float(cst.last_norms[-1]) - 0.1) < 2e-2  # the model sits on the constraint
    p = read_params(m, spec)
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_logits = P.forward_infer(spec, p, (ref_feats - mean) / scale, return_logits=True)
    got = torch.cat([m.predict_device(feats[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()
    rel = np.abs(got - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
    worst = int(rel.argmax())
    print(f"\nend-to-end, constrained model after 320 steps: mfcc max|d|={np.abs(raw.cpu().numpy() - ref_feats).max():.2e}  "
          f"logits rel max={rel.max():.2e} (row {worst}), median={np.median(rel):.2e}")
    assert rel.max() <= 1e-3, (worst, rel[worst], got[worst], ref_logits[worst])
    np.testing.assert_array_equal(got.argmax(1), ref_logits.argmax(1))
    probs = torch.cat([m.predict_device(feats[s:s + 1024]) for s in range(0, N_TEST, 1024)]).cpu().numpy()
    np.testing.assert_allclose(probs, P.softmax(ref_logits), atol=2e-5)


def test_logits_from_waveform_stress_unprojected_weights(cuda, clips):
    """The same chain on weights nothing has constrained: seeded non-negative Glorot kernels with random BatchNorm state,
    never projected, never trained.  Their product has a Lipschitz constant of ~5e4 (SURVEY 3.1 measured 51 464: every
    kernel is positive, so the all-ones direction is amplified at every layer), i.e. this network turns the MFCC stage's
    rounding (float32 FFT against the oracle's float64 FFT: <= 2.5e-4 absolute on features of spread 5-60) into logit
    differences four orders of magnitude larger than the constrained model does.  Finding (round 3): the median row agrees
    to 8e-8 relative and the argmax is identical on every row, but 6 of the 2 366 rows exceed 1e-3 and the worst reaches 5e-3 -- the
    1e-3 budget is met for the reference's constrained model (test above, 6e-6) and for 99.7 % of the rows here, not for
    the worst rows of an unconstrained random network."""
    waves, _, ref_feats = clips
    spec = P.vd_constrained_spec()
    p = _state(spec, 7)
    mean, scale = P.standard_scaler_fit(ref_feats)
    ref_logits = P.forward_infer(spec, p.astype(np.float64), (ref_feats - mean) / scale, return_logits=True)
    m = build_model(spec, max_batch=1024)
    load_params(m, p)
    raw, feats = _product_side(cuda, waves, m)
    got = torch.cat([m.predict_device(feats[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()
    err_feat = np.abs(raw.cpu().numpy() - ref_feats).max()
    err_std = np.abs(feats.cpu().numpy() - (ref_feats - mean) / scale).max()
    rel = np.abs(got - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
    worst = int(rel.argmax())
    print(f"\nend-to-end, unprojected random weights: mfcc max|d|={err_feat:.2e}  standardised max|d|={err_std:.2e}  logits rel max={rel.max():.2e} "
          f"(row {worst}), median={np.median(rel):.2e}, rows > 1e-3: {(rel > 1e-3).sum()} of {N_TEST}")
    assert err_feat < 2e-2
    np.testing.assert_array_equal(got.argmax(1), ref_logits.argmax(1))
    assert (rel <= 1e-3).mean() >= 0.995 and rel.max() <= 2e-2, (worst, rel[worst])


def test_logit_bound_along_the_constrained_trajectory(cuda, clips):
    """Where the 1e-3 logit bound starts to hold.  train_constraints.py starts from glorot_uniform kernels
    (:67), the first optimizer step applies NonNeg, and from the first batch on simple_norm_constraint(0.1) rescales the kernels
    (:102).  Logits from the waveform, product against oracle, for the weights after 0, 1, 5 and 33 constrained steps (33 = one
    epoch of the reference: 16 566 clips / 512) of the reference's batch size -- the state the reference's model is really in
    after its first batch, its first few, its first epoch.  Measured (round 4, 2 366 rows; the product-chain norm is the
    model's Lipschitz bound, each constrained step shrinks its log-distance to rho by 0.335):

        steps  ||W6^T..W1^T||   worst row    median   rows > 1e-3   argmax equal
            0       3.1e+06      4.7e-02    4.3e-06        3          yes      (unprojected random network, the stress case)
            1          8.09      1.6e-03    2.4e-06        2          yes
            5         0.127      8.8e-05    2.1e-07        0          yes
           33         0.103      1.4e-05    7.3e-08        0          yes

    i.e. the logit error tracks the Lipschitz constant of the network (it multiplies the MFCC stage's fp32 rounding), BASELINE's
    1e-3 bound holds on EVERY row from the fifth constrained step on -- the first hundredth of the reference's first epoch --
    and before that on all but 2-3 of 2 366 rows, with every argmax equal throughout."""
    from helpers import read_params
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    mean, scale = P.standard_scaler_fit(ref_feats)
    x_ref = (ref_feats - mean) / scale
    rows = []
    for n_steps in (0, 1, 5, 33):
        m = build_model(spec, max_batch=1024, seed=9)
        # glorot_uniform as Keras draws it (signed); NonNeg only acts through the optimizer step, as in the reference
        load_params(m, P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=(n_steps == 0)))
        raw, feats = _product_side(cuda, waves, m)
        if n_steps:
            x, y = feats[:2048].cpu().numpy(), P.to_categorical(labels[:2048], 10)
            ds = Dataset.from_tensor_slices((np.tile(x, (9, 1))[:512 * n_steps], np.tile(y, (9, 1))[:512 * n_steps])).batch(512)
            m.fit(ds, epochs=1, verbose=0, callbacks=[simple_norm_constraint(0.1, [])])
        p = read_params(m, spec)
        ref_logits = P.forward_infer(spec, p, x_ref, return_logits=True)
        got = torch.cat([m.predict_device(feats[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()
        rel = np.abs(got - ref_logits).max(axis=1) / np.maximum(np.abs(ref_logits).max(axis=1), 1e-6)
        lip = float(np.linalg.norm(R_chain(p.W), ord=2))
        rows.append((n_steps, lip, float(rel.max()), float(np.median(rel)), int((rel > 1e-3).sum()), bool((got.argmax(1) == ref_logits.argmax(1)).all())))
        m.close()
    print("\nsteps  ||W6^T..W1^T||   worst row    median   rows > 1e-3   argmax equal")
    for r in rows:
        print(f"{r[0]:5d}  {r[1]:14.4g}  {r[2]:10.2e}  {r[3]:8.1e}  {r[4]:6d}        {r[5]}")
    for r in rows:
        assert r[5], r  # every argmax equal, even for the unprojected network
        if r[0] >= 5:
            assert r[2] <= 1e-3, r  # the bound, on every row
        else:
            assert r[4] <= 0.005 * N_TEST and r[2] <= 0.1, r  # a few rows of a network whose Lipschitz constant is 1e1 .. 1e6
    assert rows[3][2] < rows[2][2] < rows[1][2] < rows[0][2]  # the error follows the product norm down


def test_logit_error_split_by_stage(cuda, clips):
    """which stage carries the logit error of the early-training states?  For the weights after 0, 1 and 5
    constrained steps (the states of test_logit_bound_along_the_constrained_trajectory) two crossed pairings against the all-oracle
    logits (oracle features -> oracle classifier, float64):

        K2 alone   oracle features (cast to float32) -> PRODUCT classifier
        K1 alone   PRODUCT features -> oracle classifier                       with the block-DFT STFT kernel (default) and
        both       PRODUCT features -> PRODUCT classifier                      with the Stockham STFT kernel (stage mask 256)

    Measured on the MI355X (round 5; worst row of 2 366 rows / rows over 1e-3):

        steps  ||W6^T..W1^T||        K2         K1 block-DFT     K1 Stockham     both block-DFT   both Stockham
            0     3.1e+06      3.5e-03 / 1    4.4e-02 / 4      3.8e-02 / 4      4.7e-02 / 4      4.8e-02 / 4
            1        8.09      8.2e-06 / 0    1.7e-03 / 2      1.8e-03 / 2      1.7e-03 / 2      1.8e-03 / 2
            5       0.127      1.2e-06 / 0    8.5e-05 / 0      7.9e-05 / 0      8.5e-05 / 0      7.9e-05 / 0

    Reading: the classifier alone (fp32 MFMA chains against float64) is 200 times inside the bound as soon as the network's gain
    is O(1); the 1.6e-3 of step 1 comes with the product's FEATURES, equally with either STFT kernel, and "both" is "K1".  The
    features of the two rows over the bound differ from the oracle's by at most 2.5e-5 STANDARDISED units (row median 5e-7) -- in
    coefficient 0 (the sum of the 128 mel dB values / sqrt(128)) of late frames (36-42: the decay of the clip, rms 1e-2 under a
    peak of 0.16), i.e. fp32 rounding of 128 logarithms added up, ~1e-3 dB -- and the whole set's worst standardised difference is
    3.9e-5.  That is the MFCC stage at its fp32 floor (the oracle computes the same quantities in float64); 1.7e-3 is that 2.5e-5
    times the network's gain of 8.1 relative to logits that are themselves small.  No cheap fix exists short of a float64 dB / DCT
    tail, and none is needed from the fifth step on (gain 0.13: 8.5e-5)."""
    from helpers import read_params
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.attacks import StandardScaler
    from lipasr.extract_features_construct_dataset import MfccExtractor
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, s.nonneg) for s in P.vd_constrained_spec()]
    mean, scale = P.standard_scaler_fit(ref_feats)
    x_ref = (ref_feats - mean) / scale
    wt = dev(waves)

    def product_features(mask):
        ex = MfccExtractor(16000, 16000, 1024, cuda)
        ex.set(0, mask)
        raw = torch.cat([ex(wt[s:s + 1024]) for s in range(0, N_TEST, 1024)])
        sc = StandardScaler().fit(raw)
        f = torch.cat([ex(wt[s:s + 1024], 44, sc.mean_, sc.scale_) for s in range(0, N_TEST, 1024)])
        ex.close()
        return f

    feats = {"block-DFT": product_features(0), "Stockham": product_features(256)}
    x_ref_dev = dev(x_ref.astype(np.float32))

    def rel_rows(got, ref):
        return np.abs(got - ref).max(axis=1) / np.maximum(np.abs(ref).max(axis=1), 1e-6)

    def logits_dev(m, x):
        return torch.cat([m.predict_device(x[s:s + 1024], logits=True) for s in range(0, N_TEST, 1024)]).cpu().numpy()

    table = []
    for n_steps in (0, 1, 5):
        m = build_model(spec, max_batch=1024, seed=9)
        load_params(m, P.init_params(spec, seed=9, dtype=np.float32, nonneg_init=(n_steps == 0)))
        if n_steps:  # the same trajectory as the test above: trained on the default path's features
            x, y = feats["block-DFT"][:2048].cpu().numpy(), P.to_categorical(labels[:2048], 10)
            ds = Dataset.from_tensor_slices((np.tile(x, (9, 1))[:512 * n_steps], np.tile(y, (9, 1))[:512 * n_steps])).batch(512)
            m.fit(ds, epochs=1, verbose=0, callbacks=[simple_norm_constraint(0.1, [])])
        p = read_params(m, spec)
        ref = P.forward_infer(spec, p, x_ref, return_logits=True)
        lip = float(np.linalg.norm(R_chain(p.W), ord=2))
        row = {"steps": n_steps, "lip": lip, "K2": rel_rows(logits_dev(m, x_ref_dev), ref)}
        for name, f in feats.items():
            row[f"K1 {name}"] = rel_rows(P.forward_infer(spec, p, f.cpu().numpy().astype(np.float64), return_logits=True), ref)
            row[f"both {name}"] = rel_rows(logits_dev(m, f), ref)
        table.append(row)
        if n_steps == 1:
            # where the K1 error of the worst rows comes from
            r = row["K1 block-DFT"]
            worst = np.argsort(r)[-3:][::-1]
            fd = np.abs(feats["block-DFT"].cpu().numpy().astype(np.float64) - x_ref)
            for w in worst:
                j = int(fd[w].argmax())
                print(f"step 1, row {int(w)}: logit error {r[w]:.2e}; largest standardised feature difference {fd[w].max():.2e} at coefficient {j // 44}, frame {j % 44}; "
                      f"row median feature difference {np.median(fd[w]):.1e}; clip peak |y| {np.abs(waves[w]).max():.3f}, "
                      f"rms of that frame's samples {np.sqrt(np.mean(waves[w][max(0, (j % 44) * 372 - 743):(j % 44) * 372 + 743] ** 2)):.2e}")
            print(f"all rows: standardised feature difference max {fd.max():.2e}, median row max {np.median(fd.max(axis=1)):.2e}")
        m.close()
    cols = ["K2", "K1 block-DFT", "K1 Stockham", "both block-DFT", "both Stockham"]
    print("\nsteps  ||W6^T..W1^T||  " + "  ".join(f"{c:>22s}" for c in cols) + "      (worst row / rows > 1e-3)")
    for row in table:
        print(f"{row['steps']:5d}  {row['lip']:14.4g}  " + "  ".join(f"{row[c].max():14.2e} / {int((row[c] > 1e-3).sum()):5d}" for c in cols))
    for row in table:
        if row["lip"] < 10:  # the classifier alone is never what exceeds the bound once the network's gain is O(1)
            assert row["K2"].max() <= 1e-4, (row["steps"], row["K2"].max())
        if row["steps"] >= 5:
            for c in cols:
                assert row[c].max() <= 1e-3, (row["steps"], c, row[c].max())
    # the two STFT kernels are interchangeable for this question
    for row in table:
        a, b = row["both block-DFT"], row["both Stockham"]
        assert (a > 1e-3).sum() <= 0.005 * N_TEST and (b > 1e-3).sum() <= 0.005 * N_TEST


def R_chain(w_list):
    from oracle import constraints_ref as R

    return R.product_chain([np.asarray(w, dtype=np.float64) for w in w_list])


# ------------------------------------------------------------------ accuracy parity from the waveform
def _split(seed):
    perm = np.random.default_rng(seed).permutation(N_TEST)
    return perm[:1400], perm[1400:1700], perm[1700:]  # train / validation / test (666 clips: one clip = 0.15 pt)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_accuracy_parity_from_waveform(cuda, clips, seed):
    """Final top-1 accuracy within +-0.5 pt, each side from the WAVEFORM: the oracle trains on the oracle's MFCCs (its own
    resampler, float64 FFT, its own StandardScaler), the product on the product's (K1 + A2 on the device), same split,
    same initial weights, same batch order, dropout off.  Model: train_google_dataset.py's (the one that converges in a
    dozen epochs on the synthetic task).  Split, initialisation and data order change with the seed."""
    from lipasr.attacks import StandardScaler
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    tr, _, te = _split(seed)
    spec = [P.LayerSpec(s.n_in, s.n_out, s.bn, 0.0, False) for s in P.vd_unconstrained_spec()]
    p0 = P.init_params(spec, seed=seed, dtype=np.float32)
    y = P.to_categorical(labels, 10)
    # ---- product
    m = build_model(spec, max_batch=128, seed=seed)
    load_params(m, p0)
    raw, feats = _product_side(cuda, waves, m)
    x_gpu = feats.cpu().numpy()
    m.fit(Dataset.from_tensor_slices((x_gpu[tr], y[tr])).batch(128), epochs=12, verbose=0)
    acc_gpu = float(np.mean(m.predict(x_gpu[te]).argmax(1) == labels[te]))
    # ---- oracle
    mean, scale = P.standard_scaler_fit(ref_feats)
    x_ref = (ref_feats - mean) / scale
    p64, st = p0.astype(np.float64), P.AdamState()
    for _ in range(12):
        for s in range(0, len(tr), 128):
            idx = tr[s:s + 128]
            P.train_step(spec, p64, st, x_ref[idx], y[idx].astype(np.float64))
    acc_ref = float(np.mean(P.forward_infer(spec, p64, x_ref[te]).argmax(1) == labels[te]))
    print(f"\naccuracy from the waveform, seed {seed}: product {acc_gpu:.4f}  oracle {acc_ref:.4f}")
    assert acc_ref > 0.95 and acc_gpu > 0.95, (acc_gpu, acc_ref)
    assert abs(acc_gpu - acc_ref) <= 0.005, (acc_gpu, acc_ref)


def _train_constrained(spec, x, y, labels, seed, max_batch=512, compute=None):
    """train_constraints.py:91-111 with the protocol of tests/golden/make_constrained_acc.py: 400 epochs of 11 batches of
    128 in order, simple_norm_constraint(0.1) after every step, validation loss every 10 epochs, test accuracy at the
    best-validation-loss evaluation.  Returns (test accuracy at the best checkpoint, the model)."""
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.keras import Dataset

    tr, va, te = _split(seed)
    m = build_model(spec, max_batch=max_batch, seed=seed, compute_dtype=compute)
    load_params(m, P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=True))
    ds = Dataset.from_tensor_slices((x[tr], y[tr])).batch(128)
    xv, yv, xt = dev(x[va]), dev(y[va]), dev(x[te])
    cst = simple_norm_constraint(0.1, [])
    best = (np.inf, None)
    for _ in range(40):  # 40 x 10 epochs x 11 batches = 4 400 steps
        m.fit(ds, epochs=10, verbose=0, callbacks=[cst])
        vl, _ = m._evaluate_device(xv, yv, 512)
        if vl < best[0]:
            best = (vl, float(np.mean(m.predict_device(xt).argmax(1).cpu().numpy() == labels[te])))
    return best[1], m


@pytest.mark.parametrize("compute", ["float32", "float16x2"])
def test_constrained_accuracy_product_vs_oracle(cuda, clips, compute):
    """(compute: the training GEMMs on exact fp32 chains, or on round 5's fp16 two-plane split -- the bench's default -- held to the
    same statement.)  BASELINE.json: "final top-1 accuracy within +-0.5 pt" for the model the north star names -- get_model() + NonNeg +
    simple_norm_constraint(0.1) after every batch (VD/train_constraints.py:63-111) -- PRODUCT against ORACLE.

    Oracle side: tests/golden/constrained_acc.npz, written on the CPU by tests/golden/make_constrained_acc.py: oracle.mlp_ref +
    oracle.constraints_ref.simple_norm_constraint_pass trained to convergence (4 400 steps, float32, the reference's dropout) on
    the ORACLE's MFCCs of the 2 366 synthetic clips, five seeds (split, initialisation), each with two independent dropout
    streams A and B.  Product side, here: the same five seeds, same split / initial weights / batch order / schedule / model
    selection, trained by the HIP kernels from the WAVEFORM (K1 features, A2 on the device, K2/K5/K3 steps, Philox dropout).
    Nothing is shared but the waveforms, the initial weights and the protocol.

    Dropout cannot be matched draw for draw, so single runs scatter: the oracle against ITSELF with another dropout
    stream moves by up to `scatter` (read from the fixture, 1.2 pt when it was written; 0.3 pt typical).  Statement asserted:
    |mean_product - mean_oracle| <= 0.5 pt over the five seeds, and every product run within the oracle's own same-seed
    scatter (+ one test clip) of that seed's oracle runs."""
    import os

    fx = np.load(os.path.join(os.path.dirname(__file__), "golden", "constrained_acc.npz"))
    seeds = [int(s) for s in fx["seeds"]]
    oa, ob = fx["acc_stream_a"], fx["acc_stream_b"]
    scatter = float(np.abs(oa - ob).max())
    waves, labels, _ = clips
    y = P.to_categorical(labels, 10)
    spec = P.vd_constrained_spec()
    m0 = build_model(spec, max_batch=128, seed=0)
    raw, feats = _product_side(cuda, waves, m0)
    m0.close()
    x = feats.cpu().numpy()
    acc = []
    for seed in seeds:
        a, m = _train_constrained(spec, x, y, labels, seed, compute=compute)
        m.close()
        acc.append(a)
    acc = np.array(acc)
    o_mean = 0.5 * (oa.mean() + ob.mean())
    print(f"\nconstrained model [{compute}], seeds {seeds}: test accuracy at the best-validation checkpoint\n  product (from the waveform) {acc} mean {acc.mean():.4f}"
          f"\n  oracle stream A {oa} mean {oa.mean():.4f}\n  oracle stream B {ob} mean {ob.mean():.4f}\n  oracle same-seed scatter max {scatter:.4f}")
    assert acc.min() > 0.95
    assert abs(acc.mean() - o_mean) <= 0.005  # +-0.5 pt, product mean against oracle mean
    one_clip = 1.0 / 666
    lo, hi = np.minimum(oa, ob) - scatter - one_clip, np.maximum(oa, ob) + scatter + one_clip
    assert np.all((acc >= lo) & (acc <= hi)), (acc, lo, hi)


@pytest.mark.parametrize("compute", ["float32", "float16x2"])
def test_constrained_training_is_bitwise_reproducible(cuda, clips, compute):
    """DESIGN: no float atomics anywhere, Philox dropout keyed by (seed, element, layer, step) => the same seed on the same
    features gives the SAME bits.  Two fresh models, 330 constrained steps each with the reference's dropout: every parameter,
    BatchNorm statistic and Adam moment must be bit-identical, so run-to-run accuracy scatter of this model is chaos
    (sensitivity to 1e-7 feature changes and to the dropout stream), never nondeterminism of the kernels."""
    from lipasr.Constraints import simple_norm_constraint
    from lipasr.keras import Dataset

    waves, labels, ref_feats = clips
    mean, scale = P.standard_scaler_fit(ref_feats)
    x = ((ref_feats - mean) / scale).astype(np.float32)
    y = P.to_categorical(labels, 10)
    tr, _, _ = _split(3)
    spec = P.vd_constrained_spec()
    states = []
    for _ in range(2):
        m = build_model(spec, max_batch=512, seed=3, compute_dtype=compute)
        load_params(m, P.init_params(spec, seed=3, dtype=np.float32, nonneg_init=True))
        ds = Dataset.from_tensor_slices((x[tr], y[tr])).batch(128)
        m.fit(ds, epochs=30, verbose=0, callbacks=[simple_norm_constraint(0.1, [])])
        states.append(m._state_dict())
        m.close()
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), k
