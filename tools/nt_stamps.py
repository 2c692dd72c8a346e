"""In-kernel phase stamps of the bf16 NT GEMM kernels (rgqa_probe_gemm: stamped instantiations of the product kernels).
Per shape: median over the blocks of entry -> first operands landed (prologue), the first tile's K loop, its epilogue (stores issued), the
block's life, the shader clock, and the launch's duration by HIP events.  `cold`: the operands were evicted from L2 / the Infinity Cache before
the launch (a 1-GiB write), `chain`: A was written by the previous kernel and W is cold - the situation inside the train step.
usage: python tools/nt_stamps.py > profiles/r03_nt_stamps.txt"""
import ctypes as C
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__file__), ".."))
from rgqa_amd._lib import check, load, ptr      # noqa: E402

SHAPES = [  # name, M, N, K, mt, gelu      (mt = what pick_mt chooses for the launch in the train step)
    ("N=768 K=768 projection, 12356 rows (deep ring, 160-row tiles, 234 tiles)", 12356, 768, 768, 5, 0),
    ("cross QKV 12356 x 2304 x 768 (persistent, 224-row tiles, 504 tiles)", 12356, 2304, 768, 7, 0),
    ("QKV 12356 x 2304 x 768 at 256-row tiles (persistent, 441 tiles)", 12356, 2304, 768, 8, 0),
    ("FFN1 12356 x 3072 x 768, GELU (persistent, 224-row tiles, 672 tiles)", 12356, 3072, 768, 7, 1),
    ("FFN2 12356 x 768 x 3072 (deep ring, 160-row tiles, 234 tiles)", 12356, 768, 3072, 5, 0),
    ("language-only projection 3140 x 768 x 768 (deep ring, 64-row tiles, 150 tiles)", 3140, 768, 768, 2, 0),
    ("language-only QKV 3140 x 2304 x 768 (deep ring, 64-row tiles, 450 tiles)", 3140, 2304, 768, 2, 0),
]


def main():
    lib = load()
    dev = torch.device("cuda:0")
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    evict = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    print("# tools/nt_stamps.py - phase stamps of the bf16 NT GEMM kernels (us; medians over the blocks of one launch, 5 launches each)")
    print("# launch = HIP events around the launch (includes the events' own cost); start skew = entry time of the 95th-percentile block - the first block's")
    print("# prologue = block entry -> first operands landed; kloop = first tile's K loop; epi = its epilogue (stores issued); life = block entry -> exit")
    one = torch.zeros(64, device=dev)
    fl = []
    for _ in range(20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); one.add_(1.0); e1.record()
        torch.cuda.synchronize()
        fl.append(e0.elapsed_time(e1) * 1e3)
    print("# floor of the event pair: a one-wave torch kernel between the two records reads %.1f us (median of 20) - of the 'outside any block' column that much is the method's own" % float(np.median(fl)))
    for name, M, N, K, mt, gelu in SHAPES:
        A = torch.randn(M, K, device=dev).bfloat16()
        A2 = torch.empty_like(A)
        W = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
        Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        C2 = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        tiles = -(-M // (32 * mt)) * -(-N // 256)
        nblk = min(tiles, 256) if mt >= 7 else tiles
        st = torch.zeros(nblk * 8, dtype=torch.int64, device=dev)
        print("\n## %s" % name)
        for mode in ("cold", "chain", "hot"):
            rows = []
            for rep in range(5):
                if mode != "hot":
                    evict.fill_(float(rep))
                if mode == "chain":
                    A2.copy_(A)                 # the A operand written by the kernel right before the GEMM
                src = A2 if mode == "chain" else A
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.rgqa_probe_gemm(ptr(src), ptr(W), ptr(Cc), ptr(C2), M, N, K, mt, gelu, 1, ptr(st), s))
                e1.record()
                torch.cuda.synchronize()
                v = st.view(-1, 8).cpu().numpy().astype(np.float64)
                v = v[v[:, 3] > v[:, 1]]
                t = lambda a, b: float(np.median((v[:, a] - v[:, b]) * 0.01))      # 100-MHz ticks -> us
                mhz = float(np.median((v[:, 2] - v[:, 0]) / (v[:, 3] - v[:, 1]) * 100.0))
                span = float((v[:, 3].max() - v[:, 1].min()) * 0.01)                 # first block in -> last block out
                skew_in = float((np.percentile(v[:, 1], 95) - v[:, 1].min()) * 0.01)   # how long the dispatcher takes to start 95 % of the blocks
                skew_out = float((v[:, 3].max() - np.percentile(v[:, 3], 5)) * 0.01)   # first 5 % of the blocks done -> last block done
                rows.append((t(4, 1), t(5, 4), t(6, 5), t(3, 1), mhz, e0.elapsed_time(e1) * 1e3, float(np.median(v[:, 7])), span, skew_in, skew_out))
            r = np.median(np.array(rows), axis=0)
            b2b = None
            if mode == "hot":       # 50 launches back to back, events at both ends only: what a launch costs IN a stream
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                check(lib.rgqa_probe_gemm(ptr(A), ptr(W), ptr(Cc), ptr(C2), M, N, K, mt, gelu, 50, ptr(st), s))
                e1.record()
                torch.cuda.synchronize()
                b2b = e0.elapsed_time(e1) * 1e3 / 50
            nk = K // 64
            print("%-5s prologue %5.2f  kloop %6.2f (%4.2f per K-tile)  epi %5.2f  life %6.2f  clock %4.0f MHz  tiles/block %.0f | launch %6.1f us (%4.0f TFLOP/s) = "
                  "first-in..last-out %6.2f (start skew p95 %5.2f, finish spread %5.2f) + %5.2f outside any block"
                  % (mode, r[0], r[1], r[1] / nk, r[2], r[3], r[4], r[6], r[5], 2.0 * M * N * K / r[5] * 1e-6, r[7], r[8], r[9], r[5] - r[7]))
            if b2b is not None:
                print("      50 launches back to back: %6.2f us per launch = first-in..last-out + %5.2f (%4.0f TFLOP/s)" % (b2b, b2b - r[7], 2.0 * M * N * K / b2b * 1e-6))


if __name__ == "__main__":
    main()
