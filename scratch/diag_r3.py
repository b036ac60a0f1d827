"""Round-3 diagnostics: phase split of mfcc_fused_kernel (stage-mask bits 8 / 9) and the three-kernel path, stand-alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "asr-using-robust-nn_amd")):
    sys.path.insert(0, p)
import torch
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_device

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dev = torch.device("cuda", 0)
waves, _ = synth_clips_device(8 * B, 1, dev)
ex = MfccExtractor(16000, 16000, B)
out = torch.empty(B, 880, device=dev)
for name, mask, pf in (("fused", 0, 1), ("fused, no frames (stage+resample)", 256, 1), ("fused, staging only", 256 | 512, 1), ("fused, no resample MFMAs", 512, 1), ("three-kernel, dual-FFT STFT", 0, 0), ("three-kernel, round-2 STFT", 64, 0), ):
    ex.set(0, mask); ex.set(2, pf)
    for k in range(3):
        ex(waves[k * B:(k + 1) * B], out=out)
    torch.cuda.synchronize()
    ex.profile_begin(20)
    for k in range(20):
        j = k % 8
        ex(waves[j * B:(j + 1) * B], out=out)
    ms, n = ex.profile_end()
    print(f"{name:40s} resample {ms['resample']*1e3:7.1f}  stft/fused {ms['stft_mel']*1e3:7.1f}  dct {ms['dct']*1e3:6.1f} us", flush=True)
