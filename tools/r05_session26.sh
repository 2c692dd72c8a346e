#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
echo "default HW queues"; NO_PG=1 timeout -k 10 200 python3 tools/rccl_presence2.py 2>/dev/null | grep "ms/step"; timeout -k 10 200 python3 tools/rccl_presence2.py 2>/dev/null | grep "ms/step"
echo "GPU_MAX_HW_QUEUES=8"; NO_PG=1 GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tools/rccl_presence2.py 2>/dev/null | grep "ms/step"; GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python3 tools/rccl_presence2.py 2>/dev/null | grep "ms/step"
echo "GPU_MAX_HW_QUEUES=2"; NO_PG=1 GPU_MAX_HW_QUEUES=2 timeout -k 10 200 python3 tools/rccl_presence2.py 2>/dev/null | grep "ms/step"
