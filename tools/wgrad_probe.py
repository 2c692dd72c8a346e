"""What the weight-gradient launches cost the train step, measured in ONE process with interleaved rounds (boxes differ by a few per cent):
  base        the shipped configuration (wgrad launches on a side stream)
  tn8 / tn4   the wgrad kernel forced to its older shapes (rgqa_debug_set key 4: 8 = 256-row tiles, two 64-row slots; 4 = 128-row tiles)
  serial      wgrad launches on the main stream (key 2)
  skip        wgrad launches not issued at all (key 5; gradients wrong - a floor: the step if wgrad cost nothing)
usage: python3 tools/wgrad_probe.py [steps=30] [rounds=4] [precision=bf16]"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rgqa_amd import _lib, synth
from rgqa_amd.engine import Engine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
L = _lib.load()
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lens = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
fn = bench.engine_step_fn(e, dev, lens)
variants = [("base", None), ("tn8", (4, 8)), ("tn4", (4, 4)), ("serial", (2, 1)), ("serial_tn8", (2, 1, 4, 8)), ("skip", (5, 1))]
res = {v[0]: [] for v in variants}
for r in range(rounds):
    for name, key in variants:
        if key:
            for i in range(0, len(key), 2):
                L.rgqa_debug_set(key[i], key[i + 1])
        res[name].append(bench.time_steps(fn, steps, 3))
        if key:
            for i in range(0, len(key), 2):
                L.rgqa_debug_set(key[i], 0)
for name, v in res.items():
    print("%-10s median %.3f ms/step   rounds: %s" % (name, statistics.median(v), " ".join("%.3f" % x for x in v)))
