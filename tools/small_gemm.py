"""Latency of the small-M GEMMs of the BUTD path (GRU step: M=256, N=3072, K=1024/300) under the two NT kernels."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd import _lib
lib = _lib.load()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def timeit(fn, iters=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for (M, N, K) in ((256, 3072, 1024), (256, 3072, 320), (256, 1024, 1024), (256, 2048, 1024), (256, 1842 // 8 * 8, 2048), (9216, 1024, 2112)):
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16(); b = torch.randn(N, device="cuda")
    Cc = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    run = lambda: _lib.check(lib.rgqa_op_linear(P(A), P(W), P(b), P(Cc), M, N, K, K, K, N, 0, 1, S()))
    res = []
    for mt in (-1, 2, 4, 8, 0):
        if mt < 0: lib.rgqa_debug_set(0, 1)
        else: lib.rgqa_debug_set(0, 0); lib.rgqa_debug_set(1, mt)
        res.append(timeit(run))
    lib.rgqa_debug_set(1, 0)
    print("M=%d N=%d K=%d: 128sq %.1f | MT2 %.1f MT4 %.1f MT8 %.1f auto %.1f us" % ((M, N, K) + tuple(res)), flush=True)
