"""Does a destroyed CU-masked stream leave its mask on a hardware queue that later streams inherit?  (the
PGD-20 graph replayed 3x slower as the fifth configuration of one process.)  Times a throughput-bound kernel (fp32 matmul) on
fresh streams before / while / after masked streams exist, and a graph replay of it."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'asr-using-robust-nn_amd'))
import torch
import lipasr._native as N

dev = torch.device('cuda', 0)
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
def t_on(stream, n=10):
    with torch.cuda.stream(stream):
        for _ in range(3): torch.mm(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n): torch.mm(a, b)
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / n
h = N.get_handle(0)
print("default stream  ", round(t_on(torch.cuda.current_stream()), 3), "ms")
fresh = [torch.cuda.Stream(device=dev) for _ in range(4)]
print("4 pool streams  ", [round(t_on(s), 3) for s in fresh])
def masked(ncu):
    words = 8
    m = (C.c_uint32 * words)()
    for bit in range(ncu):
        m[bit // 32] |= 1 << (bit % 32)
    st = N.c_s()
    N.check(N.lib.lipasr_stream_create_masked(h.h, m, words, C.byref(st)))
    return st
for rnd in range(3):
    sts = [masked(96), masked(64)]
    ext = [torch.cuda.ExternalStream(s.value, device=dev) for s in sts]
    print(f"round {rnd}: masked 96 / 64 CUs", [round(t_on(s), 3) for s in ext])
    print("   pool streams while masked exist", [round(t_on(s), 3) for s in fresh])
    for s in sts:
        N.check(N.lib.lipasr_stream_destroy(h.h, s))
    new = [torch.cuda.Stream(device=dev) for _ in range(6)]
    print("   6 NEW pool streams after destroy", [round(t_on(s), 3) for s in new], " old pool streams", [round(t_on(s), 3) for s in fresh])
    # a graph captured and replayed on a fresh stream
    gs = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(gs):
        torch.mm(a, b); gs.synchronize()
        with torch.cuda.graph(g, stream=gs):
            for _ in range(10): torch.mm(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay(); e0.record(gs); g.replay(); e1.record(gs)
    e1.synchronize()
    print("   graph of 10 matmuls on a new stream:", round(e0.elapsed_time(e1) / 10, 3), "ms per matmul")
print("env GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
