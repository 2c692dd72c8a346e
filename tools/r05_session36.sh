#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
echo "overlap on";  RGQA_ADAM_OVERLAP=1 timeout -k 10 300 python3 tools/x3_repro.py 8 2>/dev/null | grep trial
echo "overlap off"; RGQA_ADAM_OVERLAP=0 timeout -k 10 300 python3 tools/x3_repro.py 8 2>/dev/null | grep trial
