import sys, os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N
h=N.get_handle(0)
shapes=[('fwd0',0,0,1024,1024,880),('x2',0,0,1024,2048,880),('x4',0,0,2048,2048,880),('x8',0,0,4096,2048,880),('dX1',0,1,1024,1024,512),('dW0',1,0,880,1024,1024),('fwd1',0,0,1024,512,1024),('dW1',1,0,1024,512,1024)]
def run(ta,tb,M,Nn,K,mode):
    N.lib.lipasr_debug_gemm_mode(mode)
    A=torch.randn((K,M) if ta else (M,K),device='cuda'); Bm=torch.randn((Nn,K) if tb else (K,Nn),device='cuda'); Cc=torch.empty(M,Nn,device='cuda')
    def call(): N.check(N.lib.lipasr_gemm_f32(h.h,ta,tb,M,Nn,K,N.ptr(A),A.shape[1],N.ptr(Bm),Bm.shape[1],N.ptr(Cc),Nn,N.stream_ptr()))
    for _ in range(5): call()
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): call()
    e1.record(); torch.cuda.synchronize()
    ref=(A.t() if ta else A).double()@(Bm.t() if tb else Bm).double()
    err=float((Cc.double()-ref).abs().max()/ref.abs().max())
    return e0.elapsed_time(e1)/50*1e3, err
for name,ta,tb,M,Nn,K in shapes:
    t,e=run(ta,tb,M,Nn,K,2)
    print(f"{name:5s} {M}x{Nn}x{K}: lds kernel {t:6.1f} us ({2*M*Nn*K/t/1e6:5.1f} TFLOP/s)  rel err {e:.1e}")
N.lib.lipasr_debug_gemm_mode(0)
