"""A/B of an experimental NT main loop (rgqa_debug_set key given on the command line, default 5) on the encoder's shapes:
bit-equality against the shipped loop + interleaved timing."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd import _lib
lib = _lib.load()
KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 5
VAL = int(os.environ.get("AB_VAL", "1"))
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

def timeit(fn, iters=40):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

shapes = [(12356, 768, 768), (12356, 2304, 768), (12356, 3072, 768), (12356, 768, 3072), (12356, 768, 2304), (3140, 2304, 768), (9216, 768, 2048), (8192, 8192, 8192)]
if "--small" in sys.argv:
    shapes = [(3140, 768, 3072), (3140, 3072, 768), (3140, 2304, 768), (3140, 768, 768), (3140, 768, 2304), (256, 3072, 1024), (256, 1024, 3072), (256, 1536, 768), (12356, 768, 768), (12356, 768, 3072), (9216, 1024, 2112)]
for epi in (0, 1):
    for (M, N, K) in shapes:
        A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
        b = torch.randn(N, device="cuda")
        outs = {}
        line = "epi %d M=%5d N=%4d K=%4d:" % (epi, M, N, K)
        for mt in ((0, 2, 4, 5) if "--small" in sys.argv else ((0, 8, 7, 6, 5, 4) if "--allmt" in sys.argv else (0, 8, 6))):
            lib.rgqa_debug_set(1, mt)
            t = {}
            for rep in range(2):
                for var in (0, 1):
                    lib.rgqa_debug_set(KEY, var * VAL)
                    Cc = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
                    run = lambda: _lib.check(lib.rgqa_op_linear(P(A), P(W), P(b), P(Cc), M, N, K, K, K, N, epi, 1, S()))
                    t.setdefault(var, []).append(timeit(run))
                    outs[(mt, var)] = Cc
            same = torch.equal(outs[(mt, 0)], outs[(mt, 1)])
            line += "  MT%s old %.1f new %.1f us (%+.0f%%)%s" % (mt or "auto", min(t[0]), min(t[1]), 100 * (min(t[1]) / min(t[0]) - 1), "" if same else " MISMATCH")
        print(line, flush=True)
lib.rgqa_debug_set(1, 0); lib.rgqa_debug_set(KEY, 0)
