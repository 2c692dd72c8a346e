"""does a live RCCL communicator slow the plain train step?  same process: lean step before init_process_group('nccl'), after it, after collectives"""
import os, sys, time, datetime
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
torch.cuda.set_device(0)
e = Engine(precision="bf16", **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
step = bench.engine_step_fn(e, dev, np.ascontiguousarray(b["lengths"], dtype=np.int32))
def t(tag):
    print("%-60s %.3f ms/step" % (tag, bench.time_steps(step, 60, 10)), flush=True)
t("plain")
t("plain again")
import torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29655", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
t("after init_process_group (eager communicator)")
x = torch.ones(1 << 20, device="cuda")
dist.all_reduce(x); torch.cuda.synchronize()
t("after one all_reduce")
y = torch.empty_like(x); dist.all_to_all_single(y, x); torch.cuda.synchronize()
t("after all_to_all_single")
from rgqa_amd.parallel import make_exchange
comm = make_exchange(e, dist, mode="sharded")
t("exchange object made (staging buffers), not used")
dist.destroy_process_group()
t("after destroy_process_group")
