#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 600 python3 -m pytest tests/test_gpu_engine.py -q -x -k "segment or sumsq or golden or varlen or optimizer_pass" > $OUT/s34_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/s34_pytest.log
rm -rf $OUT/s34_p
rocprofv3 --kernel-trace --output-format csv -d $OUT/s34_p -- python3 bench.py --lean --steps 6 --warmup 3 > $OUT/s34.log 2>&1; echo "rc=$?"
T=$(ls $OUT/s34_p/*/*kernel_trace.csv | head -1)
python3 - $T <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
emb = [i for i, r in enumerate(rows) if "embed_fwd" in r[3]]
i1 = emb[-2]
i0 = max(i for i in range(i1) if "gemm_tn" in rows[i][3])
t0 = rows[i0][1]
for s, e, q, n in rows[i0 - 6:i1 + 1]:
    if (s - t0) > 400e3: break
    print("%8.1f %7.1f q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n.replace("void ", "")[:60]))
PY
rm -rf $OUT/s34_p
for i in 1 2 3; do python3 bench.py --lean --steps 100 --warmup 20 2>/dev/null | python3 -c "import sys,json; print('lean step', json.loads(sys.stdin.readline())['ms_per_step'])"; done
