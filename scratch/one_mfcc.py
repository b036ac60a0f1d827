"""Five stand-alone MFCC extractions of 1024 clips for the rocprofv3 counter passes.  argv[1]: 0 = default (three-kernel path, dual-FFT STFT
kernel), 64 = three-kernel path with the round-2 STFT kernel, 1024 = fused resample -> STFT kernel."""
import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_device
B=1024
mask=int(sys.argv[1]) if len(sys.argv)>1 else 0
wt,_=synth_clips_device(8*B, 3, torch.device('cuda',0))   # 8 different batches: nothing is served from a warm cache
ex=MfccExtractor(16000,16000,B)
ex.set(0,mask & ~1024); ex.set(2, 1 if mask & 1024 else 0)   # 1024: the fused kernel for every batch
out=torch.empty(B,880,device='cuda')
for k in range(5): ex(wt[k*B:(k+1)*B],44,out=out)
torch.cuda.synchronize()
