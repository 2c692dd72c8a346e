"""GEMM micro-benchmark on the encoder's shapes (interleaved A/B of the 128x128 and 256x256 NT kernels, TN wgrad)."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd import _lib

lib = _lib.load()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def nt(M, N, K, epi=0):
    A = torch.randn(M, K, device="cuda").bfloat16()
    W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    b = torch.randn(N, device="cuda")
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    if epi:
        print("epilogue %d:" % epi, end=" ")
    res = []
    run = lambda: _lib.check(lib.rgqa_op_linear(P(A), P(W), P(b), P(Cc), M, N, K, K, K, N, epi, 1, S()))
    lib.rgqa_debug_set(0, 1)
    res.append(2.0 * M * N * K / timeit(run) / 1e12)
    lib.rgqa_debug_set(0, 0)
    for mt in (8, 7, 6, 5, 4, 2, 0):
        lib.rgqa_debug_set(1, mt)
        res.append(2.0 * M * N * K / timeit(run) / 1e12)
    print("NT  M=%6d N=%5d K=%5d : 128sq %6.0f | dma MT8 %6.0f MT7 %6.0f MT6 %6.0f MT5 %6.0f MT4 %6.0f MT2 %6.0f | auto %6.0f TF" % ((M, N, K) + tuple(res)), flush=True)


def tn(M, N, K):
    A = torch.randn(K, M, device="cuda").bfloat16()
    B = torch.randn(K, N, device="cuda").bfloat16()
    Cc = torch.empty(M, N, device="cuda")
    run = lambda: _lib.check(lib.rgqa_op_matmul_tn(P(A), P(B), P(Cc), M, N, K, M, N, N, 1, S()))
    lib.rgqa_debug_set(0, 1)
    t0 = timeit(run)
    lib.rgqa_debug_set(0, 0)
    t = timeit(run)
    print("TN  M=%6d N=%5d K=%5d       : 128sq %7.1f TF   dma %7.1f TF" % (M, N, K, 2.0 * M * N * K / t0 / 1e12, 2.0 * M * N * K / t / 1e12), flush=True)


if __name__ == "__main__" and "--ksweep" not in sys.argv:
    for M in (() if "--tn" in sys.argv else (14336, 9216, 5120)):
        for N, K in ((768, 768), (2304, 768), (3072, 768), (768, 3072)):
            nt(M, N, K)
    nt(14336, 3072, 768, epi=1)
    nt(8192, 8192, 8192)
    nt(4096, 4096, 4096)
    for M, N in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        for K in (14336, 5120):
            tn(M, N, K)
    tn(4096, 4096, 4096)


def ksweep(M=14336, N=2304):
    print("K sweep, M=%d N=%d (auto tile): time_us" % (M, N))
    for K in (64, 128, 256, 512, 768, 1536, 3072, 6144):
        A = torch.randn(M, K, device="cuda").bfloat16()
        W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
        for mt in (8, 6, 2):
            lib.rgqa_debug_set(1, mt)
            t = timeit(lambda: _lib.check(lib.rgqa_op_linear(P(A), P(W), None, P(Cc), M, N, K, K, K, N, 0, 1, S())), iters=30)
            print("  K=%5d MT%d: %8.1f us  %7.1f TF" % (K, mt, t * 1e6, 2.0 * M * N * K / t / 1e12), flush=True)
    lib.rgqa_debug_set(1, 0)


if "--ksweep" in sys.argv:
    ksweep()
    ksweep(14336, 768)
