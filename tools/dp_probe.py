"""Where the data-parallel step's overhead at N = 1 comes from: the same train step timed, in one process, (a) before any process group
exists, (b) with a one-rank RCCL group alive, (c) with the exchange object built but unused, (d) through the exchange, (e) after release().
usage: python3 tools/dp_probe.py [mode=sharded] [precision=bf16]"""
import datetime, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
from rgqa_amd.parallel import make_exchange

mode = sys.argv[1] if len(sys.argv) > 1 else "sharded"
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
b = synth.synth_batch(256, 20, seed=1234)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lens = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e = Engine(precision=prec, **bench.FULL).allocate("cuda")
bench.init_params(e, 0)
e.ensure_shape(256, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
state = dict(i=0, comm=None)


def step():
    i = state["i"]
    e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=4321 + 1000003 * i, lengths=lens)
    e.loss_backward(dev["target"])
    c = state["comm"]
    if c is not None:
        c.exchange(); c.step(1e-6, max_norm=5.0)
    else:
        e.adam_step(1e-6, max_norm=5.0)
    state["i"] = i + 1


def t(tag):
    print("%-44s %.3f ms/step" % (tag, bench.time_steps(step, 30, 5)), flush=True)


t("(a) no process group")
import torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % bench._free_port(), rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
t("(b) one-rank RCCL group alive")
comm = make_exchange(e, dist, mode=mode)
t("(c) exchange built, unused")
e.enable_segment_sumsq(False)
t("(c') ... and backward's per-segment sums off")
state["comm"] = comm
t("(d) through the exchange: " + mode)
comm.release()
state["comm"] = None
t("(e) after release(), segment sums off")
e.enable_segment_sumsq(True)
t("(f) after release(), segment sums on")
dist.destroy_process_group()
