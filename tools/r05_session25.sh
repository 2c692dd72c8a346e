#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 300 python3 tools/rccl_presence.py 2>/dev/null | grep "ms/step"
