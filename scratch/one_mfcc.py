import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_fast
B=1024
w,_=synth_clips_fast(B, seed=3)
wt=torch.as_tensor(w).cuda()
ex=MfccExtractor(16000,16000,B)
out=torch.empty(B,880,device='cuda')
for _ in range(5): ex(wt,44,out=out)
torch.cuda.synchronize()
