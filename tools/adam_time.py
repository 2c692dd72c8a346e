"""Times the fused BertAdam kernel over a 212 M-element arena (HIP events)."""
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
for _ in range(10): e.adam_step(1e-5, max_norm=5.0)
b.record(); torch.cuda.synchronize()
n = e.params.numel()
print("adam_step (sumsq + adam + transposes): %.1f us / step, %d params" % (a.elapsed_time(b) * 100, n))
