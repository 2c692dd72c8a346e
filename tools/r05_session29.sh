#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=10 > $OUT/s29_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/s29_pytest.log
python3 bench.py --no-cpu-baseline > $OUT/s29_bench.json 2> $OUT/s29_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s29_bench.json 2>/dev/null | cut -c1-260 | grep "ms_per_step\|dropin\|other_work\|roofline"
