"""interleaved A/B of rgqa_debug_set values in ONE process with the per-category kernel times beside the step time:
python3 tools/ab_debug_cats.py KEY "v0 v1 ..." [rounds] [precision] [steps]
per value: ms per train step (B=256, lean loop of bench.py; medians over the rounds) and, from 3 profiled steps (HIP events around every launch, every stream
folded into the launch stream), the kernel ms per step by category"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgqa_amd import _lib, synth
from rgqa_amd.engine import Engine
key = int(sys.argv[1]); vals = [int(v) for v in sys.argv[2].split()]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3_fwd"
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 40
lib = _lib.load()
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
res = {v: [] for v in vals}
cats = {}
for r in range(rounds):
    for v in vals:
        _lib.check(lib.rgqa_debug_set(key, v))
        for _ in range(8): step()
        torch.cuda.synchronize()
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps): step()
        z.record(); torch.cuda.synchronize()
        res[v].append(a.elapsed_time(z) / steps)
        if r == rounds - 1:
            e.join_update(); torch.cuda.synchronize()
            e.profile(True); ov, e.adam_overlap = e.adam_overlap, False
            for _ in range(3): step()
            e.adam_overlap = ov
            p = e.profile_read(); e.profile(False)
            cats[v] = {k: round(x["ms"] / 3, 3) for k, x in p.items() if x["launches"]}
for v in vals:
    print("key %d = %d  median %.3f ms/step   rounds: %s   kernel ms/step: %s" % (key, v, statistics.median(res[v]), " ".join("%.3f" % x for x in res[v]), cats.get(v)))
