#!/usr/bin/env python3
"""Timeline view of a rocprofv3 --kernel-trace CSV: for two steady-state steps (delimited by the first kernel of a forward pass) print
per-queue busy time, time with NO kernel resident on any queue, and the largest idle gaps of the main queue with the
kernels either side. usage: tools/timeline.py <kernel_trace.csv> [n_gaps]"""
import csv, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
ngaps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 1          # optimizer launches per step (the drop-in BertAdam: one per contiguous run = 3)
# steps are delimited by the first kernel of a forward pass (round 5: the optimizer no longer runs as one long launch - it is cut at the gradient
# segments and runs beside the next forward)
fw = [i for i, r in enumerate(rows) if "embed_fwd" in r[3]]
if len(fw) < 3:
    sys.exit("need >= 3 forward passes in the trace")
t0, t1 = rows[fw[-3]][0], rows[fw[-1]][0]          # two full steps: forward start of step n-2 .. forward start of step n
span = [r for r in rows if r[0] >= t0 and r[1] <= t1]
nsteps = 2
print("window: %.3f ms over %d steps = %.3f ms/step, %d launches/step" % ((t1 - t0) / 1e6, nsteps, (t1 - t0) / 1e6 / nsteps, len(span) / nsteps))
byq = defaultdict(list)
for r in span:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(r[1] - r[0] for r in rs)
    print("queue %d: %d launches/step, busy %.3f ms/step" % (q, len(rs) / nsteps, busy / 1e6 / nsteps))
# union busy
ev = sorted((r[0], r[1]) for r in span)
cur_s, cur_e = ev[0]
union = 0
gaps = []
for s, e in ev[1:]:
    if s > cur_e:
        union += cur_e - cur_s
        gaps.append((s - cur_e, cur_e, s))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
print("any-queue busy %.3f ms/step; idle (no kernel anywhere) %.3f ms/step in %d gaps/step" % (union / 1e6 / nsteps, (t1 - t0 - union) / 1e6 / nsteps, len(gaps) / nsteps))
hist = defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    k = 1 if g < 2000 else 2 if g < 5000 else 5 if g < 10000 else 10 if g < 20000 else 20
    hist[k][0] += 1; hist[k][1] += g
for k in sorted(hist):
    print("  gaps >= %2d us class: %5.1f /step, %.3f ms/step" % (k if k > 1 else 0, hist[k][0] / nsteps, hist[k][1] / 1e6 / nsteps))
mainq = max(byq, key=lambda q: len(byq[q]))
rs = sorted(byq[mainq])
mg = []
for a, b in zip(rs, rs[1:]):
    if b[0] > a[1]:
        mg.append((b[0] - a[1], a[3][:60], b[3][:60]))
mg.sort(reverse=True)
print("main queue %d: idle %.3f ms/step between its own kernels; largest gaps:" % (mainq, sum(g[0] for g in mg) / 1e6 / nsteps))
for g in mg[:ngaps]:
    print("  %7.1f us  after %-60s before %s" % (g[0] / 1e3, g[1], g[2]))
if len(sys.argv) > 3:        # detail: for each listed main-queue gap > 8 us, what the other queues were doing
    print("gap detail (other queues at the gap):")
    oth = [r for r in span if r[2] != mainq]
    for a, b in zip(rs, rs[1:]):
        g = b[0] - a[1]
        if g > 8000:
            ov = [(o[3][:40], (o[0] - a[1]) / 1e3, (o[1] - a[1]) / 1e3) for o in oth if o[1] > a[1] - 20000 and o[0] < b[0] + 20000]
            print("  gap %.1f us @%.3f ms: " % (g / 1e3, (a[1] - t0) / 1e6) + "; ".join("%s [%.0f..%.0f]" % o for o in ov))
