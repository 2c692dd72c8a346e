"""Times the fused BertAdam launch over the 212 M-element arena (HIP events): us per optimizer step and TB/s of its 30 B per parameter.
A/B through the environment: RGQA_ADAM_NT=1 (non-temporal f32 state), RGQA_ADAM_BLOCKS=<n>."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd.engine import Engine
from bench import FULL
e = Engine(precision="bf16", **FULL).allocate("cuda")
e.ensure_shape(8, 20, 36)
e.params.normal_(0, 0.02); e.grads.normal_(0, 0.01)
e.sync_weights()
for _ in range(3): e.adam_step(1e-5, max_norm=5.0)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): e.adam_step(1e-5, max_norm=5.0)
b.record(); torch.cuda.synchronize()
n = sum(hi - lo for lo, hi in e.live_ranges())
us = a.elapsed_time(b) * 1000 / 20
print("NT=%s BLOCKS=%s: adam_step (norm + adam + transposes) %.1f us / step, %d live params, %.2f TB/s of 30 B/param (transposes 0.8 GB extra)" % (os.environ.get("RGQA_ADAM_NT", "0"), os.environ.get("RGQA_ADAM_BLOCKS", "2048"), us, n, 30.0 * n / us / 1e6))
