#!/bin/bash
# two ranks on the one GPU of the box over gloo (the supervised self-launch path of bench.py --gpus 2), both exchange modes
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for dp in sharded allreduce; do
  RGQA_DP_MODE=$dp timeout -k 10 400 python3 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/s33_$dp.json 2> $OUT/s33_$dp.err; echo "$dp rc=$?"
  python3 -c "
import json; d=json.loads(open('$OUT/s33_$dp.json').readline()); print('$dp', 'ms', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'ranks', d.get('n_ranks_seen'), 'mode', d.get('dp_mode'), 'fallback', d.get('dp_fallback'), 'value', d['value'])" || tail -5 $OUT/s33_$dp.err
done
