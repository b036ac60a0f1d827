"""The callers on the input side of K1: extract_features_construct_dataset.py's ``__main__`` block (:198-232) and
the Speaker-recognition ``main()`` (SR/extract_features_construct_dataset.py:236-267) -- folder listing, the joint
shuffle, the 70/20/10 split, MFCC of every file on the GPU, and the ``.npy`` files train_constraints.py:16-25 loads --
followed by the training driver reading those files back.
"""
import os
import wave

import numpy as np
import pytest

from oracle import mfcc_ref as M

ATOL = 2e-3  # dB-domain MFCC tolerance of tests/test_mfcc_gpu.py


def _write_wav(path, x, sr):
    with wave.open(str(path), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes((np.clip(x, -1, 1) * 32767.0).astype("<i2").tobytes())


def _tone(rng, n, sr, f0):
    t = np.arange(n) / sr
    return 0.3 * np.sin(2 * np.pi * f0 * t * (1 + 0.2 * t)) + 0.02 * rng.standard_normal(n)


def _make_corpus(root, classes, per_class, rng, sr=16000, lengths=(16000,)):
    for ci, c in enumerate(classes):
        os.makedirs(root / c, exist_ok=True)
        for k in range(per_class):
            n = lengths[(ci + k) % len(lengths)]
            _write_wav(root / c / f"{k:03d}_nohash_{ci}.wav", _tone(rng, n, sr, 200.0 + 90 * ci + 7 * k), sr)


def test_listing_shuffle_split_cpu(tmp_path):
    """No GPU involved: the reference's listing rule (labels rank the folders that are PRESENT, foreign folders are
    ignored), sklearn's shuffle restated (checked against sklearn itself when it is installed) and the split slices."""
    from lipasr.extract_features_construct_dataset import digit, get_file_names_and_labels, shuffle, split_train_dev_test

    rng = np.random.default_rng(0)
    present = ["one", "three", "nine", "zero"]
    _make_corpus(tmp_path, present + ["_background_noise_", "bed"], 3, rng, lengths=(800,))
    files, labels = get_file_names_and_labels(tmp_path)
    order = [d for d in digit if d in present]  # zero, one, three, nine
    assert len(files) == 12 and labels.tolist() == [0] * 3 + [1] * 3 + [2] * 3 + [3] * 3
    assert [os.path.basename(os.path.dirname(f)) for f in files] == [c for c in order for _ in range(3)]
    f2, l2 = shuffle(files, labels, random_state=3)
    assert sorted(f2) == sorted(files) and isinstance(f2, list)
    assert all(order[l] == os.path.basename(os.path.dirname(f)) for f, l in zip(f2, l2))
    try:
        from sklearn.utils import shuffle as sk_shuffle
    except Exception:
        sk_shuffle = None
    if sk_shuffle is not None:
        f3, l3 = sk_shuffle(files, labels, random_state=3)
        assert f3 == f2 and np.array_equal(l3, l2)
    # the reference's slice arithmetic, incl. its truncations: 16 566 / 4 733 / 2 366 of 23 666 files
    a, b, c = split_train_dev_test(list(range(23666)))
    assert (len(a), len(b), len(c)) == (16566, 4733, 2366) and a[-1] + 1 == b[0] and c[-1] == 23665
    a, b, c = split_train_dev_test(np.arange(12))
    assert (len(a), len(b), len(c)) == (8, 2, 1)


@pytest.mark.gpu
def test_voice_digit_dataset_construction_and_training(cuda, tmp_path, capsys):
    from lipasr import extract_features_construct_dataset as E
    from lipasr import train_constraints as T

    rng = np.random.default_rng(1)
    data = tmp_path / "data"
    _make_corpus(data, E.digit, 6, rng, lengths=(16000, 16000, 12345, 16000, 9000, 16000))  # some clips shorter than 1 s
    save, noise = str(tmp_path / "processed_google_dataset"), str(tmp_path / "test_dataset_to_add_noise")
    E.main(data_dir=str(data), save_dir=save, noise_dir=noise, random_state=11)
    files, labels = E.get_file_names_and_labels(data)
    files, labels = E.shuffle(files, labels, random_state=11)
    n = len(files)
    assert n == 60
    got = {k: np.load(os.path.join(save, k + ".npy")) for k in ("train_data", "train_label", "dev_data", "dev_label", "test_data", "test_label")}
    assert got["train_data"].shape == (42, 880) and got["dev_data"].shape == (12, 880) and got["test_data"].shape == (6, 880)
    assert got["train_data"].dtype == np.float64  # the reference's np.zeros(...) container (:145-146)
    np.testing.assert_array_equal(got["train_label"], labels[:42])
    np.testing.assert_array_equal(got["dev_label"], labels[42:54])
    np.testing.assert_array_equal(got["test_label"], labels[-6:])
    np.testing.assert_array_equal(np.load(os.path.join(noise, "test_label.npy")), labels[-6:])
    assert np.load(os.path.join(noise, "test_filenames.npy")).tolist() == files[-6:]
    # features: the oracle on the decoded files, a short clip among them (zero-padded MFCC columns, :33-37)
    allf = np.concatenate([got["train_data"], got["dev_data"], got["test_data"]])  # rows line up with `files`
    lens = set()
    for i in list(range(0, 54, 5)) + [54, 59]:
        x, sr = E.read_wav(files[i])
        lens.add(len(x))
        ref = M.extract_features_wave(x, sr).reshape(-1)
        assert np.abs(allf[i] - ref).max() < ATOL, i
    assert min(lens) < 16000  # a short clip was among the checked ones
    # the training driver on those files (train_constraints.py:16-42 -> :91-111), checkpoint in Keras' .h5
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        T.main(["--epochs", "2", "--data", save])
    finally:
        os.chdir(cwd)
    out = capsys.readouterr().out
    assert "Test loss" in out and os.path.exists(tmp_path / "bin" / "models_constrained" / "TEST.h5")


@pytest.mark.gpu
def test_speaker_dataset_construction(cuda, tmp_path):
    from lipasr import speaker_recognition as S
    from lipasr.extract_features_construct_dataset import shuffle, split_train_dev_test

    rng = np.random.default_rng(2)
    data = tmp_path / "rodigits"
    speakers = S.digit[:5]
    # 4 recordings per speaker, 3.4 .. 5.2 s at 22 050 Hz -> 1 .. 3 windows each once the first and last second go
    _make_corpus(data, speakers, 4, rng, sr=22050, lengths=(int(22050 * 3.4), int(22050 * 4.1), int(22050 * 5.2)))
    save, noise = str(tmp_path / "RoDigits_splitV2"), str(tmp_path / "noise")
    S.main(data_dir=str(data), save_dir=save, noise_dir=noise, random_state=4)
    files, labels = S.get_file_names_and_labels(data)
    assert len(files) == 20 and labels.max() == 4
    files, labels = shuffle(files, labels, random_state=4)
    parts_f, parts_l = split_train_dev_test(files), split_train_dev_test(labels)
    for name, fs, ls in zip(("train", "dev", "test"), parts_f, parts_l):
        feats, lab = np.load(os.path.join(save, f"{name}_data.npy")), np.load(os.path.join(save, f"{name}_label.npy"))
        wins, want = [], []
        for f, l in zip(fs, ls):
            w = M.sr_split_windows(S.read_wav(f)[0])
            wins.append(w)
            want += [l] * len(w)
        assert feats.shape == (len(want), 2020) and feats.dtype == np.float64
        np.testing.assert_array_equal(lab, want)
        ref = M.sr_mfcc_windows(np.concatenate(wins))
        assert np.abs(feats - ref).max() < ATOL
    assert np.load(os.path.join(noise, "test_filenames.npy")).tolist() == list(parts_f[2])


@pytest.mark.gpu
def test_attack_evaluation_driver(cuda, tmp_path, capsys):
    """attacks.py:296-693 as flags (lipasr.attack_eval): wav corpus -> .npy dataset -> two trained .h5 models ->
    black-box sweeps over audio and over MFCC, white-box FGSM / PGD sweeps.  Strength 0 must reproduce the clean
    accuracy, and accuracies are probabilities."""
    from lipasr import attack_eval as V
    from lipasr import attacks as A
    from lipasr import extract_features_construct_dataset as E
    from lipasr import keras as K
    from lipasr import train_constraints as T

    rng = np.random.default_rng(5)
    data = tmp_path / "data"
    _make_corpus(data, E.digit, 8, rng)
    save, noise = str(tmp_path / "processed_google_dataset") + "/", str(tmp_path / "test_dataset_to_add_noise")
    E.main(data_dir=str(data), save_dir=save, noise_dir=noise, random_state=2)
    train_data, train_label, val_data, val_label, test_data, test_label = A.load_npy_dataset(save)
    tr, va, _ = A.standardize_dataset(train_data, val_data, test_data)
    paths = {}
    for name, build in (("constrained", T.get_model), ("unconstrained", T.get_model_unconstrained)):
        K.reset_layer_names()
        m = build(max_batch=64)
        m.compile(optimizer="adam", loss=K.CategoricalCrossentropy(), metrics=["accuracy"])
        m.fit(K.Dataset.from_tensor_slices((tr, K.to_categorical(train_label, 10))).batch(56), epochs=6, verbose=0)
        paths[name] = str(tmp_path / "bin" / f"{name}.h5")
        m.save(paths[name])
    base = ["--path", save, "--noise-dir", noise, "--constrained", paths["constrained"], "--unconstrained", paths["unconstrained"]]
    models = {k: K.load_model(v, max_batch=64) for k, v in paths.items()}
    labels = K.to_categorical(test_label, 10)
    clean = {k: V.accuracy(m.predict(A.standardize_dataset(train_data, val_data, test_data)[2]), labels) for k, m in models.items()}

    grid, acc = V.main(base + ["--attack", "black", "--kind", "simple", "--over", "mfcc", "--points", "3"])
    assert len(grid) == 3 and grid[0] == 0 and set(acc) == {"constrained", "unconstrained"}
    for k in acc:
        assert acc[k][0] == clean[k] and np.all((acc[k] >= 0) & (acc[k] <= 1))
    grid, acc = V.main(base + ["--attack", "black", "--kind", "mixture", "--over", "mfcc", "--standardize", "after", "--points", "2"])
    assert len(grid) == 2 and acc["constrained"][0] == clean["constrained"]  # alpha = 0: no noise
    for kind, n in (("simple", 2), ("mixture", 2), ("snr", 2)):
        grid, acc = V.main(base + ["--attack", "black", "--kind", kind, "--over", "audio", "--points", str(n)])
        assert len(grid) == n and all(len(v) == n for v in acc.values())
    grid, acc = V.main(base + ["--attack", "white", "--kind", "fgsm", "--points", "2"])
    assert np.allclose(grid, np.linspace(0.01, 0.3, 10)[:2]) and all(len(v) == 2 for v in acc.values())
    grid, acc = V.white_box_sweep(models, train_data, val_data, test_data, labels, kind="pgd", grid=[0.5, 4.0], max_iter=5)
    for k in acc:
        assert acc[k][1] <= acc[k][0] <= clean[k] + 1e-12  # a stronger attack never helps the model it was made for
    with pytest.raises(ValueError):
        V.black_box_sweep(models, train_data, val_data, test_data, labels, kind="snr", over="mfcc")
    assert "Accuracy on black-box attack test examples" in capsys.readouterr().out


@pytest.mark.gpu
def test_remaining_training_drivers(cuda, tmp_path, capsys):
    """train_google_dataset.py:76-99 (baseline + confusion matrix) and the two Speaker-recognition drivers
    (SR/train_no_constraints.py:77-97, SR/train_constraints.py:91-113) on small .npy datasets they load themselves."""
    from lipasr import speaker_recognition as S
    from lipasr import train_google_dataset as G

    rng = np.random.default_rng(6)

    def fake_dataset(folder, n_feat, n_cls, sizes=(192, 64, 64)):
        os.makedirs(folder, exist_ok=True)
        centers = rng.standard_normal((n_cls, n_feat)) * 2.0
        for name, n in zip(("train", "dev", "test"), sizes):
            lab = rng.integers(0, n_cls, n)
            lab[:n_cls] = np.arange(n_cls)  # every class present
            np.save(os.path.join(folder, f"{name}_data"), centers[lab] + rng.standard_normal((n, n_feat)))
            np.save(os.path.join(folder, f"{name}_label"), lab)

    vd, sr = str(tmp_path / "processed_google_dataset"), str(tmp_path / "RoDigits_splitV2")
    fake_dataset(vd, 880, 10)
    fake_dataset(sr, 2020, 20)
    cm = G.confusion_matrix([0, 1, 1, 2, 2, 2], [0, 1, 2, 2, 2, 0], 3)
    np.testing.assert_array_equal(cm, [[1, 0, 0], [0, 1, 1], [1, 0, 2]])
    assert cm.dtype == np.int32
    model, results, conf = G.main(["--epochs", "12", "--data", vd, "--checkpoint", str(tmp_path / "bin" / "models" / "baselineV2.h5")])
    assert conf.shape == (10, 10) and conf.sum() == 64 and np.trace(conf) == round(results[1] * 64)
    assert results[1] > 0.5  # well-separated classes: the baseline learns them in a few epochs
    m2, r2, lip2 = S.train_no_constraints_main(["--epochs", "6", "--data", sr, "--checkpoint", str(tmp_path / "bin" / "sr_base.h5")])
    assert len(m2.layers) == 7 and np.isfinite(lip2) and lip2 > 0
    m3, r3, lip3 = S.train_constraints_main(["--epochs", "6", "--data", sr, "--checkpoint", str(tmp_path / "bin" / "sr_con.h5")])
    assert sum("batch" in l.name for l in m3.layers) == 5 and np.isfinite(lip3) and lip3 > 0
    out = capsys.readouterr().out
    assert "Upper Lipschitz constant for non-constrained model" in out and "Lipschitz constant for constrained model" in out
