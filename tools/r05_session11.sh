#!/bin/bash
# round-5 GPU session 11: where the drop-in step's 0.6 ms over the engine-direct step go: timed-window length, fresh batch vs a held one
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for n in 20 60; do echo "steps=$n"; timeout -k 10 200 python3 tools/dropin_profile.py $n 2>/dev/null | grep step; done
echo "held batch"; RGQA_DROPIN_REUSE_BATCH=1 RGQA_DROPIN_ONLY=1 timeout -k 10 200 python3 tools/dropin_profile.py 60 2>/dev/null | grep step
echo "fresh batch"; RGQA_DROPIN_ONLY=1 timeout -k 10 200 python3 tools/dropin_profile.py 60 2>/dev/null | grep step
