"""The clip norm's sum(g^2): taken segment by segment beside backward (Engine.enable_segment_sumsq) against one pass over the arena inside
adam_step - the same train step, one process, interleaved rounds.  usage: python3 tools/sumsq_probe.py [precision=bf16] [rounds=4]"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lens = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
e.ensure_shape(256, 20, 36); e.sync_weights()
fn = bench.engine_step_fn(e, dev, lens)
res = {"segment sums on": [], "segment sums off": []}
for r in range(rounds):
    for name, on in (("segment sums on", True), ("segment sums off", False)):
        e.enable_segment_sumsq(on)
        res[name].append(bench.time_steps(fn, 30, 3))
for k, v in res.items():
    print("%-18s median %.3f ms/step   rounds: %s" % (k, statistics.median(v), " ".join("%.3f" % x for x in v)))
