#!/bin/bash
# round-5 GPU session 24: one-rank RCCL rehearsal of the headline after the optimizer moved beside the forward (both exchange modes)
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for dp in sharded allreduce; do
  RGQA_DP_MODE=$dp RGQA_BENCH_RCCL_REHEARSAL=1 timeout -k 10 300 python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra-legs > $OUT/s24_rehearsal_$dp.json 2> $OUT/s24_rehearsal_$dp.err; echo "$dp rc=$?"
  python3 -c "
import json; d=json.loads(open('$OUT/s24_rehearsal_$dp.json').readline()); x=d['dp_exchange']; print('$dp', d['ms_per_step'], 'no exchange', x['no_exchange_ms_per_step'], 'exposed', x['exposed_comm_ms'], 'alt', x.get('alt_exposed_comm_ms'))"
done
python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra-legs 2>/dev/null | python3 -c "import sys,json; print('plain single-GPU step', json.loads(sys.stdin.readline())['ms_per_step'])"
