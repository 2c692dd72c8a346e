"""the drop-in trainer step of bench.py on its own, next to the engine-direct step in the SAME process (boxes of the pool differ by a few
per cent): python tools/dropin_profile.py [steps]     (RGQA_DROPIN_ONLY=1: only the drop-in leg, e.g. under rocprofv3)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
prec = os.environ.get("RGQA_PRECISION", "bf16")
if not os.environ.get("RGQA_DROPIN_ONLY"):
    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    e = Engine(precision=prec, **bench.FULL).allocate("cuda")
    bench.init_params(e, 0)
    b = synth.synth_batch(256, 20, seed=1234)
    dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
    e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
    ms0 = bench.time_steps(bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32)), n, 5)
    print("engine-direct step: %.3f ms/step" % ms0)
    del e
    torch.cuda.empty_cache()
t0 = time.perf_counter()
ms = bench.dropin_step_leg(256, 20, n, prec)
print("dropin step: %.3f ms/step (%d steps, set-up + run %.1f s)" % (ms, n, time.perf_counter() - t0))
