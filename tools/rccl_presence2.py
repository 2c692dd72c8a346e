"""as rccl_presence.py, but the communicator is made FIRST (as bench.py does), then the engine and its streams"""
import os, sys, time, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
torch.cuda.set_device(0)
import torch.distributed as dist
if os.environ.get("NO_PG") != "1":
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29656", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    x = torch.ones(1 << 20, device="cuda"); dist.all_reduce(x); y = torch.empty_like(x); dist.all_to_all_single(y, x); torch.cuda.synchronize()
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
e = Engine(precision="bf16", **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
for i in range(2):
    print("%-50s %.3f ms/step" % ("PG first" if os.environ.get("NO_PG") != "1" else "no PG", bench.time_steps(step, 60, 10)), flush=True)
