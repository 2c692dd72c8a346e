import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as Bn
from rgqa_amd.engine import Engine
from rgqa_amd import synth
e = Engine(precision="bf16", **Bn.FULL).allocate("cuda"); Bn.init_params(e, seed=0)
b = synth.synth_batch(256, 20, seed=1)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e.ensure_shape(256, 20, 36); e.sync_weights()
def step(i):
    e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=i, lengths=lengths)
    e.loss_backward(dev["target"]); e.adam_step(1e-5, max_norm=5.0)
for i in range(5): step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms/step; total %.2f ms/step" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
ts = []
for i in range(8):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); step(i); t1 = time.perf_counter()
    ts.append((t1 - t0) * 1e3)
print("host enqueue of ONE step into an empty queue: " + " ".join("%.2f" % t for t in ts) + " ms")
