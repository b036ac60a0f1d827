import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "asr-using-robust-nn_amd"))
import torch
from lipasr.speaker_recognition import WindowMfcc
B = 1024
ex = WindowMfcc(batch_max=B)
w = (0.1 * torch.randn(B, 22050, device="cuda"))
out = torch.empty(B, 2020, device="cuda")
for _ in range(3): ex(w, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ex(w, out=out)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"SR mfcc 441/220: {ms*1e3:.1f} us per {B} windows -> {B/ms*1e3:.0f} windows/s")
