"""How does hipExtStreamCreateWithCUMask map mask bits to CUs on this part?  Times one MFCC extraction (1024 clips)
on streams with different masks: the kernels' durations show how much of the chip each mask leaves them."""
import sys, os, ctypes as C; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N
from lipasr.extract_features_construct_dataset import MfccExtractor
from lipasr.synth import synth_clips_fast
B=1024
w,_=synth_clips_fast(B, seed=3); wt=torch.as_tensor(w).cuda()
ex=MfccExtractor(16000,16000,B); out=torch.empty(B,880,device='cuda')
h=ex.h
n_cu=torch.cuda.get_device_properties(0).multi_processor_count
print("CUs", n_cu)
def mask_from(fn):
    words=(n_cu+31)//32; m=(C.c_uint32*words)()
    for i in range(n_cu):
        if fn(i): m[i//32] |= 1<<(i%32)
    return m, words
def run(name, fn):
    m,words=mask_from(fn)
    st=N.c_s(); N.check(N.lib.lipasr_stream_create_masked(h.h, m, words, C.byref(st)))
    s=torch.cuda.ExternalStream(st.value)
    with torch.cuda.stream(s):
        for _ in range(3): ex(wt,44,out=out)
        s.synchronize()
        N.check(N.lib.lipasr_mfcc_profile_begin(h.h, 10))
        for _ in range(10): ex(wt,44,out=out)
        ms=(C.c_float*3)(); n=C.c_int()
        N.check(N.lib.lipasr_mfcc_profile_end(h.h, ms, C.byref(n)))
    bits=sum(bin(x).count('1') for x in m)
    print(f"{name:28s} bits {bits:3d}: resample {ms[0]*1e3:7.1f} stft {ms[1]*1e3:7.1f} dct {ms[2]*1e3:6.1f} us   mask {' '.join(f'{x:08x}' for x in m)}")
    N.check(N.lib.lipasr_stream_destroy(h.h, st))
run("all", lambda i: True)
run("first 128", lambda i: i<128)
run("first 64", lambda i: i<64)
run("first 32", lambda i: i<32)
run("even CUs", lambda i: i%2==0)
run("i%4==0", lambda i: i%4==0)
run("i%8==0", lambda i: i%8==0)
run("i%8<4", lambda i: i%8<4)
run("i%16<8", lambda i: i%16<8)
run("i%32<16", lambda i: i%32<16)
run("i%64<32", lambda i: i%64<32)
run("word0 only low 16", lambda i: i<16)
