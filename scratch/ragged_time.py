"""Stage times of the MFCC path for the four input forms (float32 / int16, one length / per-clip lengths), three kernels vs
the fused kernel, batch 1024 of 1 s clips on the whole chip."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "asr-using-robust-nn_amd")]
import numpy as np, torch
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_device
B = 1024
wt, _ = synth_clips_device(B, 1, torch.device("cuda", 0))
pcm = (wt * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
rng = np.random.default_rng(0)
lens = torch.as_tensor(rng.integers(4000, 16001, size=B).astype(np.int32)).cuda()
full = torch.full((B,), 16000, dtype=torch.int32, device="cuda")
ex = MfccExtractor(16000, 16000, B)
out = torch.empty(B, 880, device="cuda")
for fused in (0, 1):
    ex.set(2, fused)
    for name, w, nv in (("f32", wt, None), ("i16", pcm, None), ("f32 ragged(full)", wt, full), ("i16 ragged(full)", pcm, full),
                        ("i16 ragged(4k..16k)", pcm, lens)):
        for _ in range(5): ex(w, out=out, n_valid=nv)
        torch.cuda.synchronize()
        ex.profile_begin(50)
        for _ in range(50): ex(w, out=out, n_valid=nv)
        ms, n = ex.profile_end()
        print(f"{'fused' if fused else 'three'} {name:22s} resample {ms['resample']*1e3:7.1f} stft {ms['stft_mel']*1e3:7.1f} dct {ms['dct']*1e3:6.1f} us  total {sum(ms.values())*1e3:7.1f} ({n} calls)", flush=True)
