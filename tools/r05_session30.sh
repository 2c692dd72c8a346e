#!/bin/bash
# picked streams (rgqa_amd/streams.py): the queue-mapping probe again, the plain bench with every leg, the one-rank RCCL rehearsal - default queues and 8
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for q in default 8; do for k in 0 2 3 4; do
  if [ $q = default ]; then timeout -k 10 120 python3 tools/rccl_presence3.py $k 2>/dev/null | grep queues; else GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 tools/rccl_presence3.py $k 2>/dev/null | grep queues; fi
done; done
python3 bench.py --no-cpu-baseline > $OUT/s30_bench.json 2> $OUT/s30_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s30_bench.json 2>/dev/null | cut -c1-140 | grep "ms_per_step\|dropin\|other_work\|tolerance\|forward_only\|padded"
QUEUES="4 8" bash tools/r05_session27.sh
