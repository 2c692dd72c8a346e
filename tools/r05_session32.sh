#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_dropin.py -q -x > $OUT/s32_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/s32_pytest.log
timeout -k 10 200 python3 tools/dropin_profile.py 40 2>/dev/null | grep step
