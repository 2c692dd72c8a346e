#!/bin/bash
# round-5 GPU session 21: how the update and the next forward interleave (kernel trace of the lean loop)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
rm -rf $OUT/s21_p
rocprofv3 --kernel-trace --output-format csv -d $OUT/s21_p -- python3 bench.py --lean --steps 6 --warmup 3 > $OUT/s21.log 2>&1; echo "rc=$?"
T=$(ls $OUT/s21_p/*/*kernel_trace.csv | head -1)
python3 - $T > $OUT/s21_update_vs_forward.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
emb = [i for i, r in enumerate(rows) if "embed_fwd" in r[3]]
# window: from the last weight-gradient launch before the 2nd-last forward until 4.5 ms into that forward
i1 = emb[-2]
i0 = max(i for i in range(i1) if "gemm_tn" in rows[i][3])
t0 = rows[i0][1]
print("t = 0: end of backward's last weight-gradient launch; q = queue; the update runs on a queue of its own")
for s, e, q, n in rows[i0:]:
    if s - t0 > 4.6e6: break
    short = n.replace("void ", "")[:64]
    print("%8.1f %7.1f q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, short))
PY
head -120 $OUT/s21_update_vs_forward.txt
rm -rf $OUT/s21_p
