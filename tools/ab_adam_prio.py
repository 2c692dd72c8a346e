"""does a HIGH-priority stream for the step (forward / backward) change how the update beside the forward shares the chip?  one process, interleaved:
step loop on the default stream vs on a priority -1 stream; the update always on a default-priority stream of its own"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
e = Engine(precision="bf16", **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
hi = torch.cuda.Stream(priority=-1)
res = {}
for r in range(3):
    for name, st in (("default", torch.cuda.current_stream()), ("high", hi)):
        for ov in (0, 1):
            e.adam_overlap = bool(ov)
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                for _ in range(8): step()
                torch.cuda.synchronize()
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(40): step()
                e.join_update()
                z.record(); torch.cuda.synchronize()
            res.setdefault((name, ov), []).append(a.elapsed_time(z) / 40)
for k, v in res.items():
    print("step stream %-7s adam_overlap %d  median %.3f  rounds %s" % (k[0], k[1], statistics.median(v), " ".join("%.3f" % x for x in v)))
