#!/bin/bash
# round-5 GPU session 4: BUTD engine after the grouped weight-norm / weight-gradient launches and the 8-wave persistent GRU
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests/test_gpu_butd.py -q --maxfail=10 -s > $OUT/s4_pytest.log 2>&1; echo "pytest rc=$?"; grep "butd\|passed\|failed\|Error" $OUT/s4_pytest.log | tail -30
for v in 1 2 0; do
  echo "butd GRU mode=$v"; python3 - <<PY 2>/dev/null
import sys; sys.path.insert(0, '.')
from rgqa_amd import _lib
lib = _lib.load(); lib.rgqa_debug_set(18, $v)
import bench, torch
for r in range(3):
    print("  butd step %.3f ms" % bench.butd_leg(256, 30))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s4_p_butd -- python3 bench.py --butd --lean --steps 20 --warmup 5 > $OUT/s4_butd.log 2>&1; echo "butd prof rc=$?"
python3 tools/prof_summary.py $(ls $OUT/s4_p_butd/*/*kernel_stats.csv | head -1) 25 $OUT/s4_butd_kernel_stats.md > /dev/null; head -48 $OUT/s4_butd_kernel_stats.md | tail -38
rm -rf $OUT/s4_p_butd
