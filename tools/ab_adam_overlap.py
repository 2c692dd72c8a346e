"""interleaved A/B in ONE process: the optimizer pass on the step's stream vs beside the next forward pass (Engine.adam_overlap).
python3 tools/ab_adam_overlap.py [rounds] [precision] [steps]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
res = {0: [], 1: []}
for r in range(rounds):
    for v in (0, 1):
        e.adam_overlap = bool(v)
        for _ in range(8): step()
        torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps): step()
        e.join_update()
        z.record(); torch.cuda.synchronize()
        res[v].append(a.elapsed_time(z) / steps)
for v in (0, 1):
    print("adam_overlap = %d  median %.3f ms/step   rounds: %s" % (v, statistics.median(res[v]), " ".join("%.3f" % x for x in res[v])))
