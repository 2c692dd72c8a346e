"""queue-mapping probe: communicator first, then K extra streams (each used once), then the engine; lean step with the update beside the forward (1) / on the step's stream (0)"""
import os, sys, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.set_device(0)
import torch.distributed as dist
K = int(sys.argv[1]) if len(sys.argv) > 1 else 0
pg = os.environ.get("NO_PG") != "1"
if pg:
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29657", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    x = torch.ones(1 << 20, device="cuda"); dist.all_reduce(x); torch.cuda.synchronize()
extra = []
for k in range(K):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        torch.ones(16, device="cuda").add_(1)
    extra.append(s)
torch.cuda.synchronize()
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
e = Engine(precision="bf16", **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
out = []
for ov in (1, 0, 1):
    e.adam_overlap = bool(ov)
    out.append("overlap %d: %.3f" % (ov, bench.time_steps(step, 40, 8)))
print("queues %s  pg %d  extra streams %d   %s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), pg, K, "   ".join(out)), flush=True)
