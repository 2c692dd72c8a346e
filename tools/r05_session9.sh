#!/bin/bash
# round-5 GPU session 9: the persistent GRU with 32-unit slices (weights in registers): BUTD tests, the forms side by side, kernel stats
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 500 python3 -m pytest tests/test_gpu_butd.py -q --maxfail=10 -s > $OUT/s9_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep "butd\|passed\|failed\|Error\|assert" $OUT/s9_pytest.log | tail -30
[ $rc -eq 0 ] || exit 1
for v in 3 1 3 1; do
  echo "butd GRU mode=$v"; timeout -k 10 120 python3 - <<PY 2>/dev/null
import sys; sys.path.insert(0, '.')
from rgqa_amd import _lib
lib = _lib.load(); lib.rgqa_debug_set(18, $v)
import bench, torch
for r in range(3):
    print("  butd step %.3f ms" % bench.butd_leg(256, 30))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s9_p_butd -- python3 bench.py --butd --lean --steps 20 --warmup 5 > $OUT/s9_butd.log 2>&1; echo "butd prof rc=$?"
python3 tools/prof_summary.py $(ls $OUT/s9_p_butd/*/*kernel_stats.csv | head -1) 25 $OUT/s9_butd_kernel_stats.md > /dev/null; head -30 $OUT/s9_butd_kernel_stats.md | tail -22
rm -rf $OUT/s9_p_butd
