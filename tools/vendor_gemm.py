"""Yardstick, never the product path: the vendor bf16 GEMM (torch.matmul -> hipBLASLt) on the hot shapes of the train step, timed
interleaved with this repo's kernels (rgqa_op_linear / rgqa_op_matmul_tn) on the same box, same random operands.

    python3 tools/vendor_gemm.py                 # table: per-launch time in a stream of back-to-back launches, TFLOP/s, ratio
    python3 tools/vendor_gemm.py --names         # three vendor launches per shape and nothing else: run under
                                                 #   rocprofv3 --kernel-trace, then tools/vendor_gemm.py --parse <kernel_trace.csv>

Two operand regimes per shape: "hot" = the same operands every launch (L2 / Infinity-Cache resident), "rot" = 8 operand sets taken in
turn (A and W come from beyond the 4-MB L2s, as inside the train step where every launch reads what another kernel wrote).  Grouped
launches of the step (language 3,140 rows + vision 9,216 rows, their own weights) are given to the library as ONE 12,356-row problem:
an upper bound for it, since it then streams one weight panel instead of two."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (tag, M, N, K): C[M,N] = A[M,K] W[N,K]^T  (forward / dgrad launches of profiles/r03_launch_table_bf16.txt)
NT_SHAPES = [
    ("FFN1 paired", 12356, 3072, 768), ("FFN2 paired", 12356, 768, 3072), ("QKV paired/shared", 12356, 2304, 768),
    ("QKV dgrad", 12356, 768, 2304), ("attn-out", 12356, 768, 768),
    ("FFN1 lang", 3140, 3072, 768), ("FFN2 lang", 3140, 768, 3072), ("QKV lang", 3140, 2304, 768), ("QKV dgrad lang", 3140, 768, 2304),
    ("attn-out lang", 3140, 768, 768), ("visn_fc", 9216, 768, 2048), ("tail FFN2 M=256", 256, 768, 3072),
]
# (tag, M, N, K): dW[M,N] = dY[K,M]^T X[K,N]   (weight gradients)
TN_SHAPES = [("wgrad FFN", 3072, 768, 12356), ("wgrad QKV", 2304, 768, 12356), ("wgrad out", 768, 768, 12356), ("wgrad FFN2", 768, 3072, 12356)]
NSET = 8


def stream_time(fn, n=40):
    """per-launch microseconds of n back-to-back launches between two events (the event pair's own ~5 us is amortised)"""
    for _ in range(3):
        fn(0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    from rgqa_amd import _lib
    lib = _lib.load()
    S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    names_only = "--names" in sys.argv
    torch.manual_seed(0)
    rows = []
    print("# box: %s, torch %s; per-launch us in a stream of 40 launches, median of 5 interleaved rounds" % (torch.cuda.get_device_name(0), torch.__version__))
    print("%-20s %6s %5s %6s | %9s %9s %6s | %9s %9s %6s" % ("shape", "M", "N", "K", "lib hot", "ours hot", "ratio", "lib rot", "ours rot", "ratio"))
    for kind, shapes in (("NT", NT_SHAPES), ("TN", TN_SHAPES)):
        for tag, M, N, K in shapes:
            if kind == "NT":
                A = [torch.randn(M, K, device="cuda").bfloat16() for _ in range(NSET)]
                W = [(torch.randn(N, K, device="cuda") * 0.05).bfloat16() for _ in range(NSET)]
                Cv = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                Co = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                vend = lambda i, r: torch.matmul(A[(i % NSET) * r], W[(i % NSET) * r].t(), out=Cv)
                ours = lambda i, r: _lib.check(lib.rgqa_op_linear(P(A[(i % NSET) * r]), P(W[(i % NSET) * r]), None, P(Co), M, N, K, K, K, N, 0, 1, S()))
            else:
                A = [torch.randn(K, M, device="cuda").bfloat16() for _ in range(NSET)]
                W = [torch.randn(K, N, device="cuda").bfloat16() for _ in range(NSET)]
                Cv = torch.empty(M, N, dtype=torch.float32, device="cuda")
                Co = torch.empty(M, N, dtype=torch.float32, device="cuda")
                Cb = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
                # the library has no bf16 x bf16 -> f32 entry in torch; its bf16-output form is the yardstick (fewer bytes written: in its favour)
                vend = lambda i, r: torch.matmul(A[(i % NSET) * r].t(), W[(i % NSET) * r], out=Cb)
                ours = lambda i, r: _lib.check(lib.rgqa_op_matmul_tn(P(A[(i % NSET) * r]), P(W[(i % NSET) * r]), P(Co), M, N, K, M, N, N, 1, S()))
            if names_only:
                for i in range(3):
                    vend(i, 1)
                torch.cuda.synchronize()
                continue
            # agreement first: a yardstick must compute the same thing
            vend(0, 1); ours(0, 1); torch.cuda.synchronize()
            ref = (Cb if kind == "TN" else Cv).float(); got = Co.float()
            err = float((got - ref).abs().max() / ref.abs().max())
            assert err < 2e-2, (tag, err)
            t = {k: [] for k in ("vh", "oh", "vr", "or")}
            for _ in range(5):
                t["vh"].append(stream_time(lambda i: vend(i, 0)))
                t["oh"].append(stream_time(lambda i: ours(i, 0)))
                t["vr"].append(stream_time(lambda i: vend(i, 1)))
                t["or"].append(stream_time(lambda i: ours(i, 1)))
            m = {k: statistics.median(v) for k, v in t.items()}
            fl = 2.0 * M * N * K
            tf = lambda us: fl / us / 1e6
            print("%-20s %6d %5d %6d | %5.1f %4.0f %5.1f %4.0f %5.2f | %5.1f %4.0f %5.1f %4.0f %5.2f" % (
                tag, M, N, K, m["vh"], tf(m["vh"]), m["oh"], tf(m["oh"]), m["oh"] / m["vh"], m["vr"], tf(m["vr"]), m["or"], tf(m["or"]), m["or"] / m["vr"]), flush=True)
            rows.append((tag, m))
            del A, W
    if not names_only:
        print("# columns: us TF/s (library) us TF/s (this repo) ratio = ours / library time (> 1: the library is faster)")


def parse(csv_path):
    """kernel names of the --names run: the i-th group of three library launches belongs to shape i"""
    import csv
    shapes = [("NT",) + s for s in NT_SHAPES] + [("TN",) + s for s in TN_SHAPES]
    ks = []
    for r in csv.DictReader(open(csv_path)):
        n = r["Kernel_Name"]
        if n.startswith("Cijk_"):
            ks.append((int(r["Start_Timestamp"]), n, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                       r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?")), r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?")))
    ks.sort()
    for i, s in enumerate(shapes):
        grp = ks[3 * i:3 * i + 3]
        if not grp:
            break
        print("%s %-20s %6d %5d %6d : last of 3 launches %.1f us, workgroup %s grid %s LDS %s VGPR %s\n    %s" % (s[0], s[1], s[2], s[3], s[4], grp[-1][2], grp[-1][3], grp[-1][4], grp[-1][5], grp[-1][6], grp[-1][1]))


if __name__ == "__main__":
    if "--parse" in sys.argv:
        parse(sys.argv[sys.argv.index("--parse") + 1])
    else:
        main()
