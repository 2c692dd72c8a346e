#!/bin/bash
# a subset of the GPU suite: tools/gpu.sh --timeout 900 -- "bash tools/gpu_quick.sh <pytest args>"
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 800 python3 -m pytest "$@" > $OUT/quick_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/quick_pytest.log
