#!/usr/bin/env python3
"""What runs between backward and the next forward: every kernel (all queues) from START_PAT's last launch before the big optimizer launch of a
steady-state step until END_PAT's first launch after it.  usage: tools/step_tail.py <kernel_trace.csv> [steps-from-the-end=2] [tn-launches-back=1] [queue-filter]
(default window: the last weight-gradient GEMM of backward .. the first embedding kernel of the next forward)"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
emb = [i for i, r in enumerate(rows) if "embed_fwd" in r[3] or "butd_embed_fwd" in r[3]]
if len(emb) < back + 1:
    sys.exit("need more forward passes in the trace")
i1 = emb[-back]                                               # the first embedding kernel of a steady-state forward pass
tnb = int(sys.argv[3]) if len(sys.argv) > 3 else 1
qf = int(sys.argv[4]) if len(sys.argv) > 4 else None
tns = [i for i in range(i1) if "gemm_tn" in rows[i][3]]
i0 = tns[-1]                                                  # the last weight-gradient launch of the backward pass before it
ifrom = tns[-tnb]
ads = [i for i in range(i0, i1) if "bertadam" in rows[i][3]]
ia = max(ads, key=lambda i: rows[i][1] - rows[i][0]) if ads else i0
t0 = rows[i0][1]
print("window: end of the last weight-gradient launch -> first embedding kernel of the next forward: %.1f us; optimizer launches %.1f us of it" % (
    (rows[i1][0] - t0) / 1e3, sum(rows[i][1] - rows[i][0] for i in ads) / 1e3))
print("%9s %9s %3s  %s" % ("start us", "dur us", "q", "kernel"))
for s, e, q, n in rows[ifrom:i1 + 1]:
    if qf is not None and q != qf and "gemm_tn" not in n:
        continue
    print("%9.1f %9.1f %3d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:110]))
