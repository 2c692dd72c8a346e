import ctypes as C, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd import _lib
lib = _lib.load()
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
def split(x):
    x = x.contiguous().float(); out = torch.empty(x.shape, dtype=torch.int32, device="cuda")
    _lib.check(lib.rgqa_split_f32(P(x), P(out), x.numel(), S())); return out
def unsplit(x):
    out = torch.empty(x.shape, dtype=torch.float32, device="cuda")
    _lib.check(lib.rgqa_unsplit_f32(P(x), P(out), x.numel(), S())); return out
M, N, K = 128, 256, 64
A = (torch.arange(M * K).reshape(M, K) % 97).float().cuda()
W = torch.zeros(N, K, device="cuda")
for n in range(N): W[n, n % K] = 1.0
Cs = split(torch.zeros(M, N, device="cuda"))
As, Ws = split(A), split(W)
_lib.check(lib.rgqa_op_linear(P(As), P(Ws), None, P(Cs), M, N, K, K, K, N, 0, 2, S()))
got = unsplit(Cs); ref = A @ W.t()
bad = (got != ref)
print("bad count", int(bad.sum()), "of", bad.numel())
rows = bad.any(1).nonzero().flatten()[:10].tolist(); cols = bad.any(0).nonzero().flatten()[:40].tolist()
print("bad rows", rows, "bad cols", cols)
print("got[0,:16]", got[0, :16].tolist()); print("ref[0,:16]", ref[0, :16].tolist())
print("got[0,32:48]", got[0, 32:48].tolist()); print("ref[0,32:48]", ref[0, 32:48].tolist())
# TN
K2, M2, N2 = 64, 256, 256
A2 = (torch.arange(K2 * M2).reshape(K2, M2) % 13).float().cuda(); B2 = torch.zeros(K2, N2, device="cuda")
for n in range(N2): B2[n % K2, n] = 1.0
Cc = torch.zeros(M2, N2, device="cuda")
A2s, B2s = split(A2), split(B2)
_lib.check(lib.rgqa_op_matmul_tn(P(A2s), P(B2s), P(Cc), M2, N2, K2, M2, N2, N2, 2, S()))
ref2 = A2.t() @ B2
print("TN bad", int((Cc != ref2).sum()), "of", Cc.numel())
print("TN got[0,:8]", Cc[0, :8].tolist(), "ref", ref2[0, :8].tolist())
