"""Main-loop ablation of the NT LDS-DMA kernel (results garbage in modes 1/2): where does a K-step's time go?"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd import _lib
lib = _lib.load()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

for (M, N, K) in ((14336, 2304, 6144), (14336, 2304, 768), (8192, 8192, 8192)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    run = lambda: _lib.check(lib.rgqa_op_linear(P(A), P(W), None, P(Cc), M, N, K, K, K, N, 0, 1, S()))
    lib.rgqa_debug_set(1, 8)
    res = []
    for rep in range(2):
        for mode in (0, 1, 2, 5):
            lib.rgqa_debug_set(3, mode)
            res.append(timeit(run))
    lib.rgqa_debug_set(3, 0); lib.rgqa_debug_set(1, 0)
    print("M=%d N=%d K=%d MT8: full %.1f / %.1f us | no-DMA %.1f / %.1f | DMA only %.1f / %.1f | DMA only, 2x bytes in flight %.1f / %.1f   (full = %.0f TF)" % (M, N, K, res[0], res[4], res[1], res[5], res[2], res[6], res[3], res[7], 2.0*M*N*K/res[0]/1e6), flush=True)
