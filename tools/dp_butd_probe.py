"""one-rank RCCL group around a BUTD engine (arch = 1): does the exchange drive it?  python3 tools/dp_butd_probe.py [mode] [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import torch.distributed as dist
from rgqa_amd.engine import Engine
from rgqa_amd.parallel import make_exchange
from rgqa_amd import synth
mode = sys.argv[1] if len(sys.argv) > 1 else "sharded"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29641", rank=0, world_size=1, device_id=torch.device("cuda", 0))
B, L, O, NA, NT = 64, 40, 36, 1842, 3000
def make():
    e = Engine(arch=1, vocab_size=NT + 1, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=NA, precision=prec,
               hidden_dropout=0.0, attn_dropout=0.0, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    e.ensure_shape(B, L, O); e.sync_weights()
    return e
b = synth.synth_batch(B, L, seed=5, uq_frac=0.25, vocab=NT)
import bench
toks = torch.from_numpy(bench.butd_tokens(B, L, NT, seed=3)).cuda()
f, p, t = (torch.from_numpy(b[k]).cuda() for k in ("feats", "boxes", "target"))
def step(e, comm):
    e.forward(f, p, toks, toks, None, train=False)
    e.loss_backward(t)
    if comm is not None:
        comm.exchange(); comm.step(1e-3, max_norm=5.0)
    else:
        e.adam_step(1e-3, max_norm=5.0)
e0 = make()
for _ in range(3): step(e0, None)
ref = e0.params.clone()
e1 = make()
comm = make_exchange(e1, dist, mode, overlap=True, chunk_mb=4, bucket_mb=4)
print("exchange:", comm.describe(), "gather_overlap", getattr(comm, "gather_overlap", None), "segments", len(e1.grad_segments()) if hasattr(e1, "grad_segments") else None)
for _ in range(3): step(e1, comm)
comm.gather_master(); torch.cuda.synchronize()
d = (e1.params - ref).abs()
print("butd dp %s/%s: |params - plain| max %.3e mean %.3e" % (mode, prec, float(d.max()), float(d.mean())))
lg0 = e0.forward(f, p, toks, toks, None, train=False)[0].clone(); lg1 = e1.forward(f, p, toks, toks, None, train=False)[0].clone()
print("logits diff after 3 steps: %.3e (|logits| max %.3e)" % (float((lg0 - lg1).abs().max()), float(lg0.abs().max())))
dist.destroy_process_group()
