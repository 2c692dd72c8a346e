#!/bin/bash
# round-5 GPU session 17: the optimizer pass beside the next forward pass - equality with the serial step, then the interleaved A/B
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 300 python3 - <<'PY'
import sys; sys.path.insert(0, '.')
import numpy as np, torch, bench
from rgqa_amd import synth
from rgqa_amd.engine import Engine
out = {}
for prec in ("bf16", "bf16x3_fwd"):
  for ov in (0, 1):
    e = Engine(precision=prec, **bench.FULL).allocate("cuda")
    bench.init_params(e, 0)
    b = synth.synth_batch(64, 20, seed=7)
    dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
    e.ensure_shape(64, 20, 36); e.sync_weights(); e.enable_segment_sumsq(True)
    e.adam_overlap = bool(ov)
    lens = np.ascontiguousarray(b["lengths"], dtype=np.int32)
    losses = []
    for i in range(4):
        e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=11 + i, lengths=lens)
        losses.append(e.loss_backward(dev["target"]).clone())
        e.adam_step(1e-4, max_norm=5.0)
    torch.cuda.synchronize()
    out[(prec, ov)] = (torch.cat([l.reshape(1) for l in losses]).cpu(), e.params.cpu().clone(), e.params_lp.cpu().clone(), e.params_lp_t.cpu().clone())
    del e
for prec in ("bf16", "bf16x3_fwd"):
    a, b = out[(prec, 0)], out[(prec, 1)]
    first = 30522 * 768 + 600000       # behind the embedding tables (float-atomic scatter: to rounding)
    print(prec, "losses", a[0].tolist(), b[0].tolist())
    print(prec, "params equal behind the tables:", bool(torch.equal(a[1][25417728:], b[1][25417728:])), "max diff in tables %.3e" % float((a[1][:25417728] - b[1][:25417728]).abs().max()),
          "operand copies equal:", bool(torch.equal(a[2][25417728:], b[2][25417728:])), bool(torch.equal(a[3][25417728:], b[3][25417728:])))
PY
echo "eq rc=$?"
timeout -k 10 400 python3 tools/ab_adam_overlap.py 4 bf16 40 2>/dev/null | grep adam_overlap
timeout -k 10 400 python3 tools/ab_adam_overlap.py 2 bf16x3_fwd 30 2>/dev/null | grep adam_overlap
