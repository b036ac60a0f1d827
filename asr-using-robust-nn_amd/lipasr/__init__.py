"""lipasr -- the MI355X-native hot path of fmazilu/ASR-using-robust-NN.

waveform -> MFCC (K1) -> dense classifier fwd/bwd on fp32 MFMA (K2) -> Keras-form Adam + NonNeg (K5)
-> Lipschitz projection (K3) -> FGSM / PGD sign step (K4), data-parallel over RCCL.

Modules mirror the reference's files: ``Constraints``, ``extract_features_construct_dataset``,
``attacks``, ``train_constraints``; ``keras`` is the slice of the Keras API those files use.
Importing the package loads liblipasr.so and raises if it is missing (there is no CPU fallback).
"""
from . import _native  # noqa: F401  (fails loudly when liblipasr.so is absent)

__all__ = ["Constraints", "attacks", "extract_features_construct_dataset", "keras", "parallel", "pipeline", "synth",
           "train_constraints"]
__version__ = "0.2.0"
