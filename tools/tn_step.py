"""Per-K-step time of the TN (weight-gradient) kernel at full occupancy: one 256x256 tile per CU."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch, ctypes as C
import gemm_bench as g
from rgqa_amd import _lib
lib = g.lib
for plan in (0, 1):
    lib.rgqa_debug_set(6, plan)
    for (M, N, K) in ((4096, 4096, 8192), (4096, 4096, 2048), (3072, 768, 9216), (3072, 3072, 9216), (8192, 8192, 4096)):
        A = torch.randn(K, M, device="cuda").bfloat16(); B = torch.randn(K, N, device="cuda").bfloat16()
        Cc = torch.empty(M, N, device="cuda")
        for mtw in (8, 4):
            lib.rgqa_debug_set(4, mtw)
            run = lambda: _lib.check(lib.rgqa_op_matmul_tn(g.P(A), g.P(B), g.P(Cc), M, N, K, M, N, N, 1, g.S()))
            t = g.timeit(run)
            tiles = (M // (32 * mtw)) * (N // 256)
            rounds = -(-tiles // 256)
            print("plan %d TN M=%5d N=%5d K=%5d mtw %d: %7.1f us  %6.0f TF  tiles %4d  us/K-step/round %.2f" % (plan, M, N, K, mtw, t * 1e6, 2.0 * M * N * K / t / 1e12, tiles, t * 1e6 / (K / 64) / rounds), flush=True)
lib.rgqa_debug_set(4, 0)
