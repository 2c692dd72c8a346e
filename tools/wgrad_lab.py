"""Lab: one cross-modality layer's weight-gradient launch (10 problems, 252 tiles of 256 x 256, contraction lengths 12356 / 9216 / 3140) through
rgqa_op_matmul_tn_group: one layer per launch vs two layers merged into one launch vs two launches side by side on two streams, and each
contraction class alone.
python3 tools/wgrad_lab.py [bf16|x3]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rgqa_amd import _lib
lib = _lib.load()
X3 = len(sys.argv) > 1 and sys.argv[1] == "x3"
RL, RV = 3140, 9216
H, I = 768, 3072


def operand(rows, cols):
    if X3:
        t = torch.randn(rows, cols, device="cuda"); o = torch.empty(rows, cols, dtype=torch.int32, device="cuda")
        _lib.check(lib.rgqa_split_f32(C.c_void_p(t.data_ptr()), C.c_void_p(o.data_ptr()), t.numel(), None)); return o
    return torch.randn(rows, cols, device="cuda").bfloat16()


def layer():
    """(dY, X, M, N, K) of one cross layer: shared cross-attention (all rows), self-attention and FFN of both modalities"""
    P = []
    for rows in (RL + RV, RL, RV):
        P.append((operand(rows, 3 * H), operand(rows, H), 3 * H, H, rows))       # qkv
        P.append((operand(rows, H), operand(rows, H), H, H, rows))               # attention output
    for rows in (RL, RV):
        P.append((operand(rows, I), operand(rows, H), I, H, rows))               # FFN up
        P.append((operand(rows, H), operand(rows, I), H, I, rows))               # FFN down
    return P


def launch(P, Cs, stream):
    n = len(P)
    vp = C.c_void_p * n; ia = C.c_int * n
    args = (n, vp(*[p[0].data_ptr() for p in P]), vp(*[p[1].data_ptr() for p in P]), vp(*[c.data_ptr() for c in Cs]), None,
            ia(*[p[2] for p in P]), ia(*[p[3] for p in P]), ia(*[p[4] for p in P]), ia(*[p[2] for p in P]), ia(*[p[3] for p in P]), ia(*[p[3] for p in P]))
    def go(epoch):
        _lib.check(lib.rgqa_op_matmul_tn_group(*args, 0, 2 if X3 else 1, C.c_void_p(stream.cuda_stream)))
    return go


def timeit(fns, reps=20, warm=3):
    ep = [1]
    def once():
        for f in fns:
            f(ep[0]); ep[0] += 1
    for _ in range(warm): once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): once()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


L1, L2 = layer(), layer()
C1 = [torch.empty(p[2], p[3], device="cuda") for p in L1]; C2 = [torch.empty(p[2], p[3], device="cuda") for p in L2]
s0 = torch.cuda.current_stream()
flop = lambda P: sum(2.0 * p[2] * p[3] * p[4] for p in P)
rows = []
rows.append(("one layer per launch", timeit([launch(L1, C1, s0), launch(L2, C2, s0)]) / 2, flop(L1)))
rows.append(("two layers merged in one launch", timeit([launch(L1 + L2, C1 + C2, s0)]) / 2, flop(L1)))
# two launches side by side on two streams
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
f1, f2 = launch(L1, C1, s1), launch(L2, C2, s2)
def both(ep):
    f1(ep); f2(ep)
for _ in range(3): both(0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); s1.wait_stream(s0); s2.wait_stream(s0)
for _ in range(20): both(0)
s0.wait_stream(s1); s0.wait_stream(s2); b.record(); torch.cuda.synchronize()
rows.append(("two launches side by side on two streams", a.elapsed_time(b) / 40 * 1e3, flop(L1)))
# by contraction class, alone: what a balanced launch of equal tiles does
for name, sel in (("K=12356 problems alone (36 tiles)", [0, 1]), ("K=9216 problems alone (108 tiles)", [4, 5, 8, 9]), ("K=3140 problems alone (108 tiles)", [2, 3, 6, 7])):
    P = [L1[i] for i in sel]; Cs = [C1[i] for i in sel]
    rows.append((name, timeit([launch(P, Cs, s0)]), flop(P)))
print("# %s operands; us per layer-launch, TFLOP/s" % ("split-f32" if X3 else "bf16"))
for n, us, f in rows:
    print("%-66s %8.1f us %7.0f TF/s" % (n, us, f / us * 1e-6))
