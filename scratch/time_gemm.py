import sys, os, ctypes as C; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N
h=N.get_handle(0)
B=1024
W=[880,1024,512,256,128,64,10]
shapes=[]
for l in range(6):
    shapes.append(('fwd%d'%l,0,0,B,W[l+1],W[l]))      # X[B,in] @ W[in,out]: transA=0, transB=0
    if l>0: shapes.append(('dX%d'%l,0,1,B,W[l],W[l+1]))  # dZ[B,out] @ W^T: B stored [N=in][K=out] -> transB=1
    shapes.append(('dW%d'%l,1,0,W[l],W[l+1],B))       # X^T @ dZ: A stored [K=B][M=in] -> transA=1
def run(ta,tb,M,Nn,K,mode):
    N.lib.lipasr_debug_gemm_mode(mode)
    A=torch.randn((K,M) if ta else (M,K),device='cuda'); Bm=torch.randn((Nn,K) if tb else (K,Nn),device='cuda'); Cc=torch.empty(M,Nn,device='cuda')
    s=torch.cuda.current_stream()
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
tot={1:0,2:0}
for name,ta,tb,M,Nn,K in shapes:
    t1,e1=run(ta,tb,M,Nn,K,1); 
    legal = M>=64 and Nn>=64 and K>=32
    t2,e2=run(ta,tb,M,Nn,K,2) if legal else (float('nan'),0)
    print(f"{name:6s} M{M:5d} N{Nn:5d} K{K:5d}  splitK {t1:7.1f} us   lds {t2:7.1f} us   err {e1:.1e} {e2:.1e}")
N.lib.lipasr_debug_gemm_mode(0)
