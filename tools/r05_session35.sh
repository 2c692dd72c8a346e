#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_engine.py tests/test_gpu_dropin.py tests/test_gpu_ops.py -q -x > $OUT/s35_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/s35_pytest.log
for i in 1 2 3; do python3 bench.py --lean --steps 100 --warmup 20 2>/dev/null | python3 -c "import sys,json; print('lean step', json.loads(sys.stdin.readline())['ms_per_step'])"; done
rm -rf $OUT/s35_p
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s35_p -- python3 bench.py --lean --steps 5 --warmup 2 > $OUT/s35.log 2>&1
grep -h "sumsq" $OUT/s35_p/*/*kernel_stats.csv | cut -c1-120
rm -rf $OUT/s35_p
