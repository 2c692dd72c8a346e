"""Device-side timing of the step boundary WITHOUT a profiler (torch events on the caller's stream): A = end of backward, B = a one-element kernel
enqueued right after adam_step, C = end of the next forward.  A->B says whether main-stream work is held up once the optimizer pass starts beside it,
A->C how much of the optimizer pass the forward hides (compare overlap on / off).  usage: python tools/step_boundary_probe.py [precision]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench as Bn
from rgqa_amd.engine import Engine
from rgqa_amd import synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3_fwd"
e = Engine(precision=prec, **Bn.FULL).allocate("cuda"); Bn.init_params(e, seed=0)
b = synth.synth_batch(256, 20, seed=1)
dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
lengths = np.ascontiguousarray(b["lengths"], dtype=np.int32)
e.ensure_shape(256, 20, 36); e.sync_weights()
one = torch.zeros(1, device="cuda")
def run(overlap, n=24):
    e.adam_overlap = overlap
    evs = []
    for i in range(n):
        e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=i, lengths=lengths)
        c = torch.cuda.Event(enable_timing=True); c.record()
        e.loss_backward(dev["target"])
        a = torch.cuda.Event(enable_timing=True); a.record()
        e.adam_step(1e-5, max_norm=5.0)
        one.add_(1.0)
        bb = torch.cuda.Event(enable_timing=True); bb.record()
        evs.append((c, a, bb))
    torch.cuda.synchronize()
    ab = [evs[i][1].elapsed_time(evs[i][2]) for i in range(4, n - 1)]
    ac = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(4, n - 1)]
    st = [evs[i][0].elapsed_time(evs[i + 1][0]) for i in range(4, n - 1)]
    print("overlap=%d: A->B %.3f ms (median; min %.3f max %.3f)   A->C (optimizer + next forward) %.3f ms   step %.3f ms" % (
        overlap, np.median(ab), min(ab), max(ab), np.median(ac), np.median(st)))
for ov in (True, False, True, False):
    run(ov)
