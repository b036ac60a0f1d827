"""First-light check of stft_bdft_kernel: new default vs stft_mel2_kernel (stage-mask 256) vs the float64 oracle, and kernel times."""
import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/asr-using-robust-nn_amd'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips
from oracle import mfcc_ref as M

def dev(x): return torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda().contiguous()
rng = np.random.default_rng(0)
for n in (22050, 9000, 1500, 23000, 2, 600):
    w = (0.2 * rng.standard_normal((3, n))).astype(np.float32)
    ex = MfccExtractor(22050, n, 4)
    L = 1 + n // 512
    new = ex(dev(w), L).cpu().numpy()
    ex.set(0, 256); old = ex(dev(w), L).cpu().numpy(); ex.set(0, 0)
    ref = M.compute_mfcc_batch(w, sr_in=22050, utterance_length=L)
    print(f"22.05k n={n:6d} frames {L:3d}: new-oracle {np.abs(new-ref).max():.2e} old-oracle {np.abs(old-ref).max():.2e} new-old {np.abs(new-old).max():.2e}", flush=True)
    ex.close()
waves, _ = synth_clips(64, seed=11)
ex = MfccExtractor(16000, 16000, 1024)
new = ex(dev(waves)).cpu().numpy()
ex.set(0, 256); old = ex(dev(waves)).cpu().numpy(); ex.set(0, 0)
ref = M.compute_mfcc_batch(waves[:8])
print(f"16k synth: new-oracle {np.abs(new[:8]-ref).max():.2e} old-oracle {np.abs(old[:8]-ref).max():.2e} new-old {np.abs(new-old).max():.2e}", flush=True)
a = ex(dev(waves)).cpu().numpy(); assert np.array_equal(a, new), "nondeterministic"
# timing at batch 1024
big = torch.as_tensor(np.tile(waves, (16, 1))).cuda()
for mask, seg, name in ((256, 44, "stft_mel2"), (0, 44, "bdft seg44"), (0, 24, "bdft seg24"), (0, 12, "bdft seg12"), (0, 44, "bdft seg44")):
    ex.set(0, mask); ex.set(3, seg)
    for _ in range(3): ex(big)
    torch.cuda.synchronize()
    ex.profile_begin(50)
    for _ in range(50): ex(big)
    torch.cuda.synchronize()
    ms = ex.profile_end()
    print(name, "resample/stft/dct ms:", ms, flush=True)
