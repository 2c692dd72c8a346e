#!/bin/bash
# round-5 GPU session 15: the step's tail under the sharded exchange on a one-rank RCCL group, kernel by kernel
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
rm -rf $OUT/s15_p
RGQA_DP_MODE=sharded RGQA_BENCH_RCCL_REHEARSAL=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/s15_p -- python3 bench.py --lean --steps 6 --warmup 3 > $OUT/s15.log 2>&1; echo "rc=$?"
T=$(ls $OUT/s15_p/*/*kernel_trace.csv | head -1)
python3 tools/step_tail.py $T 2 6 > $OUT/s15_tail_all.txt 2>&1
python3 - $T > $OUT/s15_side_queue.txt <<'PY'
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
emb = [i for i, r in enumerate(rows) if "embed_fwd" in r[3]]
a, b = emb[-3], emb[-2]
t0 = rows[a][0]
print("one steady-state step, forward's first kernel at 0: exchange-related kernels and weight-gradient launches")
for s, e, q, n in rows[a:b + 1]:
    if any(k in n for k in ("gemm_tn", "sum_parts", "cast_bf16", "rccl", "bertadam", "sumsq", "sum_partials", "embed_fwd", "fillBuffer", "cast_transpose", "bce_kernel")):
        print("%9.1f %8.1f q%d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:70]))
PY
rm -rf $OUT/s15_p
