"""VERDICT r3 item 3, the cheap form of the question: does the resampled signal y stay in L2 between the resampler and the STFT kernel when
the batch is cut into chunks whose y fits the L2s (8 x 4 MiB)?  argv[1] = clips per chunk (1024 = one launch per kernel, the default path).
Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (scratch/pmc_chunk.sh); prints the stage time per 1024 clips as well."""
import sys; import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_device
B=1024
C=int(sys.argv[1]) if len(sys.argv)>1 else 1024
wt,_=synth_clips_device(8*B, 3, torch.device('cuda',0))
ex=MfccExtractor(16000,16000,B)
out=torch.empty(B,880,device='cuda')
def batch(k):
    for c0 in range(0,B,C):
        ex(wt[k*B+c0:k*B+c0+C],44,out=out[c0:c0+C])
batch(7); torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(5): batch(k)
e1.record(); torch.cuda.synchronize()
print(f"chunk {C}: {e0.elapsed_time(e1)/5*1e3:.1f} us per 1024 clips")
