"""Where does the host wait inside a steady-state step?  Per call (forward / loss_backward / adam_step) host time WITHOUT synchronising, 30 steps:
a call that blocks on the device shows up as ~step-time instead of enqueue-time.  usage: python tools/host_block_probe.py [precision=bf16x3_fwd]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as Bn
from rgqa_amd.engine import Engine
from rgqa_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3_fwd"
e = Engine(precision=prec, **Bn.FULL).allocate("cuda"); Bn.init_params(e, seed=0)
b = synth.synth_batch(256, 20, seed=1)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e.ensure_shape(256, 20, 36); e.sync_weights()
def step(i, rec=None):
    t0 = time.perf_counter()
    e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=i, lengths=lengths)
    t1 = time.perf_counter()
    e.loss_backward(dev["target"])
    t2 = time.perf_counter()
    e.adam_step(1e-5, max_norm=5.0)
    t3 = time.perf_counter()
    if rec is not None:
        rec.append((t1 - t0, t2 - t1, t3 - t2))
for i in range(5): step(i)
torch.cuda.synchronize()
rec = []
t0 = time.perf_counter()
for i in range(30): step(i, rec)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
r = np.array(rec) * 1e3
print("host enqueue %.2f ms/step; total %.2f ms/step" % ((t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3))
print("per call, median / max ms: forward %.3f / %.3f   loss_backward %.3f / %.3f   adam_step %.3f / %.3f" % (
    np.median(r[:, 0]), r[:, 0].max(), np.median(r[:, 1]), r[:, 1].max(), np.median(r[:, 2]), r[:, 2].max()))
print("steps 0..9 forward:", " ".join("%.2f" % x for x in r[:10, 0]))
print("steps 0..9 backward:", " ".join("%.2f" % x for x in r[:10, 1]))
print("steps 0..9 adam:", " ".join("%.2f" % x for x in r[:10, 2]))
