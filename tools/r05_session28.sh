#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for q in 8 4; do for k in 0 1 2 3 4 6; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python3 tools/rccl_presence3.py $k 2>/dev/null | grep queues
done; done
NO_PG=1 GPU_MAX_HW_QUEUES=8 timeout -k 10 120 python3 tools/rccl_presence3.py 6 2>/dev/null | grep queues
