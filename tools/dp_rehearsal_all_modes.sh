#!/bin/bash
# round-5 GPU session 14: every bench.py workload under a one-rank RCCL group (the exchange's code path at world 1)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for m in "--mixup" "--uniter" "--butd" "--precision bf16x3_fwd" "--precision bf16x3" "--padded" ""; do
  for dp in sharded allreduce; do
    tag=$(echo "$m" | tr -d ' -'); tag=${tag:-headline}
    RGQA_DP_MODE=$dp RGQA_BENCH_RCCL_REHEARSAL=1 timeout -k 10 300 python3 bench.py $m --steps 12 --warmup 4 --no-cpu-baseline --no-extra-legs > $OUT/s14_${tag}_$dp.json 2> $OUT/s14_${tag}_$dp.err; rc=$?
    echo "$tag/$dp rc=$rc $(python3 -c "
import json,sys
try:
    d=json.loads(open('$OUT/s14_${tag}_$dp.json').readline()); print(d['ms_per_step'], d.get('dp_mode'), (d.get('dp_exchange') or {}).get('exposed_comm_ms'))
except Exception as e: print('no json', e)
")"
    [ $rc -eq 0 ] || tail -5 $OUT/s14_${tag}_$dp.err
  done
done
