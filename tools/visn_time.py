"""Times the VisualFeatEncoder tail kernels through the engine's profiler categories (quick check after kernel edits)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as Bn
from rgqa_amd.engine import Engine
from rgqa_amd import synth
e = Engine(precision="bf16", **Bn.FULL).allocate("cuda"); Bn.init_params(e, seed=0)
b = synth.synth_batch(256, 20, seed=1)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights()
os.environ["RGQA_PROF_DUMP"] = "/tmp/prof_dump.txt"
e.profile(True)
for i in range(3):
    e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=i)
    e.loss_backward(dev["target"])
print({k: round(v["ms"] / 3, 3) for k, v in e.profile_read().items()})
print({k: round(v["ms"] / 3, 3) for k, v in e.profile_blocks().items()})
