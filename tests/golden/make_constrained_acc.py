"""Writes tests/golden/constrained_acc.npz: the ORACLE side of BASELINE.json's "+-0.5 pt top-1 accuracy" statement for
the model the north star names -- get_model() + NonNeg + simple_norm_constraint(0.1) after every batch
(VD/train_constraints.py:63-111) -- trained on the CPU by oracle.mlp_ref / oracle.constraints_ref alone.

    waveforms     lipasr.synth.synth_clips(2366, seed=2366)          (the reference's test-split size)
    features      oracle.mfcc_ref.compute_mfcc_batch (exact resampler), oracle StandardScaler fitted on all 2 366 rows
    split         np.random.default_rng(seed).permutation(2366) -> 1400 train / 300 validation / 666 test
    model         vd_constrained_spec() (dropout 0.1 after layers 1-3), glorot_uniform made non-negative, float32
    schedule      400 epochs x 11 batches of 128 (in order, no reshuffle: VD/train_constraints.py:41) = 4 400 steps,
                  simple_norm_constraint_pass(rho = 0.1) after every step (the sequential six-projection pass,
                  VD/Constraints.py:171-189), validation loss every 10 epochs
    reported      test accuracy at the best-validation-loss evaluation (ModelCheckpoint(save_best_only) + evaluate,
                  VD/train_constraints.py:104-111)

Each seed is run with two independent dropout streams (A and B), so the file also holds the oracle's OWN run-to-run
scatter for an identical seed, split and initialisation -- the bound a single product-vs-oracle pair can be held to.

tests/test_end_to_end_gpu.py::test_constrained_accuracy_product_vs_oracle trains the product from the waveform with the
same protocol and compares.  Run here (CPU, about 10 min per run on one core):

    python tests/golden/make_constrained_acc.py            # all seeds, a process per run
    python tests/golden/make_constrained_acc.py --one 3 A  # a single run, prints its record
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "asr-using-robust-nn_amd"))

import numpy as np

SEEDS = (3, 4, 5, 6, 7)
STREAMS = {"A": 1000, "B": 2000}
N = 2366
RHO = 0.1
EPOCH_BLOCKS, EPOCHS_PER_BLOCK, BATCH = 40, 10, 128
FEATS = os.environ.get("LIPASR_ORACLE_FEATS", "/tmp/oracle_feats_2366_exact.npy")


def split(seed):
    perm = np.random.default_rng(seed).permutation(N)
    return perm[:1400], perm[1400:1700], perm[1700:]


def _chunk(c):
    from oracle import mfcc_ref as M

    return M.compute_mfcc_batch(c)


def features():
    from lipasr.synth import synth_clips

    waves, labels = synth_clips(N, seed=2366)
    if os.path.exists(FEATS):
        return np.load(FEATS), labels
    import multiprocessing as mp

    with mp.get_context("spawn").Pool(min(8, os.cpu_count())) as pool:
        f = np.concatenate(pool.map(_chunk, [c for c in np.array_split(waves, 64) if len(c)]))
    np.save(FEATS, f)
    return f, labels


def run(seed, stream):
    from oracle import constraints_ref as R
    from oracle import mlp_ref as P

    f, labels = features()
    mean, scale = P.standard_scaler_fit(f)
    x = ((f - mean) / scale).astype(np.float32)
    y = P.to_categorical(labels, 10)
    tr, va, te = split(seed)
    spec = P.vd_constrained_spec()
    p = P.init_params(spec, seed=seed, dtype=np.float32, nonneg_init=True)
    st = P.AdamState()
    drop = np.random.default_rng(STREAMS[stream] + seed)

    def masks(b):
        return [None if s.dropout == 0 else ((drop.random((b, s.n_out)) >= s.dropout) / (1.0 - s.dropout)).astype(np.float32)
                for s in spec]

    def evaluate(idx):
        lg = P.forward_infer(spec, p, x[idx], return_logits=True).astype(np.float64)
        lp = lg - lg.max(1, keepdims=True)
        lp = lp - np.log(np.exp(lp).sum(1, keepdims=True))
        return float(-(y[idx] * lp).sum(1).mean()), float((lg.argmax(1) == labels[idx]).mean())

    best = (np.inf, None, -1)
    hist = []
    t0 = time.time()
    norm = None
    for blk in range(EPOCH_BLOCKS):
        for _ in range(EPOCHS_PER_BLOCK):
            for s in range(0, len(tr), BATCH):
                idx = tr[s:s + BATCH]
                P.train_step(spec, p, st, x[idx], y[idx], masks=masks(len(idx)))
                p.W, norms = R.simple_norm_constraint_pass(p.W, RHO)
                norm = norms[-1]
        vl, vacc = evaluate(va)
        tl, tacc = evaluate(te)
        hist.append((vl, vacc, tl, tacc, norm))
        if vl < best[0]:
            best = (vl, tacc, blk)
        print(f"seed {seed}{stream} block {blk:2d}: val loss {vl:.4f} acc {vacc:.4f} | test acc {tacc:.4f} | product norm {norm:.5f} | {time.time() - t0:.0f} s",
              file=sys.stderr, flush=True)
    sig = [float(R.sigma_max(w)) for w in p.W]
    return dict(seed=seed, stream=stream, best_val_loss=best[0], test_acc_at_best=best[1], best_block=best[2],
                final_test_acc=hist[-1][3], history=hist, final_sigmas=sig, final_product_norm=float(norm), steps=st.t)


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--one":
        print(json.dumps(run(int(sys.argv[2]), sys.argv[3])))
        return
    import subprocess

    features()  # cache before the children start
    env = dict(os.environ, OPENBLAS_NUM_THREADS="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    jobs = [(sd, s) for s in STREAMS for sd in SEEDS]
    width = int(os.environ.get("LIPASR_ORACLE_JOBS", "5"))
    recs = []
    while jobs:
        batch, jobs = jobs[:width], jobs[width:]
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--one", str(sd), s], stdout=subprocess.PIPE, env=env)
                 for sd, s in batch]
        for pr in procs:
            out, _ = pr.communicate()
            assert pr.returncode == 0
            recs.append(json.loads(out.decode().strip().splitlines()[-1]))
    recs.sort(key=lambda r: (r["stream"], r["seed"]))
    acc = {s: np.array([r["test_acc_at_best"] for r in recs if r["stream"] == s]) for s in STREAMS}
    np.savez(os.path.join(HERE, "constrained_acc.npz"),
             seeds=np.array(SEEDS), rho=RHO, steps=np.array([r["steps"] for r in recs]),
             acc_stream_a=acc["A"], acc_stream_b=acc["B"],
             final_acc_stream_a=np.array([r["final_test_acc"] for r in recs if r["stream"] == "A"]),
             final_acc_stream_b=np.array([r["final_test_acc"] for r in recs if r["stream"] == "B"]),
             best_block_a=np.array([r["best_block"] for r in recs if r["stream"] == "A"]),
             best_block_b=np.array([r["best_block"] for r in recs if r["stream"] == "B"]),
             history=np.array([r["history"] for r in recs], dtype=np.float64),  # [run][block][val loss, val acc, test loss, test acc, product norm]
             final_sigmas=np.array([r["final_sigmas"] for r in recs]),
             final_product_norm=np.array([r["final_product_norm"] for r in recs]))
    print("stream A", acc["A"], "mean", acc["A"].mean())
    print("stream B", acc["B"], "mean", acc["B"].mean())
    print("same-seed scatter |A - B|", np.abs(acc["A"] - acc["B"]))


if __name__ == "__main__":
    main()
