"""Stand-alone MFCC stage timing (HIP events inside the library): argv = batch, then stage masks (0 = default, 256 = Stockham STFT,
512 = keep dct_kernel).  8 different batches of a device-generated pool; 30 timed extractions per mask."""
import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_device
B=int(sys.argv[1]) if len(sys.argv)>1 else 1024
wt,_=synth_clips_device(8*B, 3, torch.device('cuda',0))
ex=MfccExtractor(16000,16000,B)
out=torch.empty(B,880,device='cuda')
for mask in [int(a) for a in sys.argv[2:]] or [0]:
    ex.set(0,mask)
    for k in range(4): ex(wt[k*B:(k+1)*B],44,out=out)
    torch.cuda.synchronize()
    ex.profile_begin(30)
    for k in range(30): ex(wt[(k%8)*B:(k%8+1)*B],44,out=out)
    torch.cuda.synchronize()
    ms,n=ex.profile_end()
    print(f"mask {mask:4d}: resample {ms['resample']*1e3:7.1f} us  stft {ms['stft_mel']*1e3:7.1f} us  dct {ms['dct']*1e3:6.1f} us  total {(ms['resample']+ms['stft_mel']+ms['dct'])*1e3:7.1f} us")
