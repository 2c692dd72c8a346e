"""per-launch records of the B=256 train step (HIP events around every launch, wgrad launches serialised onto the main stream):
RGQA_PROF_DUMP=<file> python tools/prof_dump.py [bf16|bf16x3] [steps]   then   python tools/launch_table.py <file> <steps> [x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rgqa_amd.engine import Engine
from rgqa_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
for _ in range(5):
    step()
torch.cuda.synchronize()
e.profile(True)
for _ in range(steps):
    step()
p = e.profile_read()
e.profile(False)
print({k: round(v["ms"] / steps, 3) for k, v in p.items()})
