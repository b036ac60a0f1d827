import sys, ctypes as C; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import numpy as np, torch
import lipasr._native as N
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_fast
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
w,_=synth_clips_fast(B, seed=3)
wt=torch.as_tensor(w).cuda()
ex=MfccExtractor(16000,16000,B)
out=torch.empty(B,880,device='cuda')
s=torch.cuda.Stream()
for mask in [int(a) for a in sys.argv[2:]] or [0]:
    N.check(N.lib.lipasr_debug_set(ex.h.h,0,mask))
    with torch.cuda.stream(s):
        for _ in range(3): ex(wt,44,out=out)
        s.synchronize()
        N.check(N.lib.lipasr_mfcc_profile_begin(ex.h.h, 20))
        for _ in range(20): ex(wt,44,out=out)
        ms=(C.c_float*3)(); n=C.c_int()
        N.check(N.lib.lipasr_mfcc_profile_end(ex.h.h, ms, C.byref(n)))
    print(f"mask {mask}: resample {ms[0]*1e3:8.1f} us  stft_mel {ms[1]*1e3:8.1f} us  dct {ms[2]*1e3:8.1f} us  (n={n.value})")
N.check(N.lib.lipasr_debug_set(ex.h.h,0,0))
