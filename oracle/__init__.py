"""CPU oracle for the lipasr hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy restatement of the reference algorithm for the path
named in BASELINE.json (MFCC -> Lipschitz-constrained MLP -> FGSM/PGD).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / timed baseline.  The
product package (``asr-using-robust-nn_amd``) never imports it and has no CPU
fallback: it raises if ``liblipasr.so`` is missing.

Pinning status (DESIGN.md section "Oracle"): PARITY UNPINNED for every module.

The reference has no tests, golden vectors or stored outputs for this path, and none of its
modules can be run here: each imports TensorFlow and/or librosa at top level, which are absent
and not installable.  Importing ``Constraints.py`` behind fake ``tensorflow.keras`` modules was
NOT done: a stand-in for a library the image lacks is not a reference run.  So:

* ``constraints_ref`` -- restated line by line from VD/Constraints.py; checked by mathematical
  known answers (sigma_max == rho^(1/m) after ``norm_constraint``; the product-norm law
  n_k = n0^((5/6)^k) rho^(1-(5/6)^k) for ``simple_norm_constraint``) and LAPACK SVD.
* ``mfcc_ref``   -- librosa/resampy/scipy.fftpack restated from their published algorithm
  (early-2022 defaults); checked by analytic signals (tone resampling, filter-bank shape), SciPy's
  Hann / DCT, ``torch.stft`` for both window shapes (2048/512 and the Speaker-recognition 441/220), and
  ``transformers.audio_utils`` (an independent re-implementation of librosa's Slaney mel bank, power spectrogram
  and power_to_db) + ``scipy.fftpack.dct`` for the whole chain after the resampler (``mfcc_22k``).
* ``mlp_ref``    -- TensorFlow/Keras restated; checked by finite-difference gradients, by torch autograd
  for every gradient it produces (training and inference mode, the output VJP), by scikit-learn's own
  ``StandardScaler`` and by ``torch.optim.Adam`` in the eps -> 0 limit.
* ``attacks_ref``-- ART restated (FGSM, PGD, JSMA, Carlini-Wagner L2 / Linf) from its published algorithms.

These are independent implementations installed in the image, not the reference's pinned versions:
they check the restatement's arithmetic, the status stays PARITY UNPINNED.

The known-answer checks live in tests/test_oracle_cpu.py.  ``tests/golden/*.npz`` are produced by
THIS oracle (tests/golden/make_golden.py), not by the reference; they are regression vectors that
also travel to the GPU box.  The only reference-held data are the label arrays
(``tests/golden/ref_labels.npz``: split sizes and class histograms).
"""
