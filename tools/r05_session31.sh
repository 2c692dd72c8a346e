#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 500 python3 tools/ab_debug.py 20 "2048 1024 512 256 128" 3 bf16 40 2>/dev/null | grep key
