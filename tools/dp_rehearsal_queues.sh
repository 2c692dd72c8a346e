#!/bin/bash
# hardware-queue count vs the DP harness (one-rank RCCL rehearsal, both modes) and the plain step
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for q in ${QUEUES:-4 8 16 24}; do
  for dp in sharded allreduce; do
    GPU_MAX_HW_QUEUES=$q RGQA_DP_MODE=$dp RGQA_BENCH_RCCL_REHEARSAL=1 timeout -k 10 300 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs > $OUT/s27_$dp.json 2> $OUT/s27_$dp.err
    python3 -c "
import json; d=json.loads(open('$OUT/s27_$dp.json').readline()); x=d['dp_exchange']; print('queues $q', '$dp', d['ms_per_step'], 'alt', x.get('alt_ms_per_step'), 'no exchange', x['no_exchange_ms_per_step'], 'padded', d['padded_layout']['ms_per_step'])"
  done
  GPU_MAX_HW_QUEUES=$q python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('queues $q plain', d['ms_per_step'], 'padded', d['padded_layout']['ms_per_step'])"
done
