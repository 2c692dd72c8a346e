#!/bin/bash
# the whole GPU suite, then the round's profile set (tools/gpu.sh --timeout 1200 -- "bash tools/gpu_suite.sh")
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=8 > $OUT/suite_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $OUT/suite_pytest.log
[ $rc -eq 0 ] || exit 1
bash tools/profile_round.sh r05
