"""What the weight-gradient launches cost the train step, measured in ONE process with interleaved rounds (boxes differ by a few per cent):
  base        the shipped configuration (wgrad launches on a side stream of default priority)
  prio_low    the side stream created with the LOWEST priority HIP offers (main-stream blocks are dispatched first)
  prio_high   ... with the highest
  serial      wgrad launches on the main stream (rgqa_debug_set key 2)
  skip        wgrad launches not issued at all (key 5; gradients wrong - a floor: the step if wgrad cost nothing)
usage: python3 tools/wgrad_probe.py [steps=30] [rounds=4] [precision=bf16]"""
import ctypes as C
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
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi))
print("# stream priority range: least %d, greatest %d" % (lo.value, hi.value))
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lens = np.ascontiguousarray(b["lengths"], dtype=np.int32)


def make(prio):
    L.rgqa_debug_set(3, prio)          # read when the engine creates its side stream (first backward)
    e = Engine(precision=prec, **bench.FULL).allocate("cuda")
    bench.init_params(e, 0)
    e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
    fn = bench.engine_step_fn(e, dev, lens)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    L.rgqa_debug_set(3, 0)
    return fn


fns = {"base": make(0), "prio_low": make(lo.value), "prio_high": make(hi.value)}
variants = [("base", "base", None), ("prio_low", "prio_low", None), ("prio_high", "prio_high", None), ("serial", "base", (2, 1)), ("skip", "base", (5, 1))]
res = {v[0]: [] for v in variants}
for r in range(rounds):
    for name, fk, key in variants:
        if key:
            L.rgqa_debug_set(key[0], key[1])
        res[name].append(bench.time_steps(fns[fk], steps, 3))
        if key:
            L.rgqa_debug_set(key[0], 0)
for name, v in res.items():
    print("%-10s median %.3f ms/step   rounds: %s" % (name, statistics.median(v), " ".join("%.3f" % x for x in v)))
