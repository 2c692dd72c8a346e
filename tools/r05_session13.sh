#!/bin/bash
# round-5 GPU session 13: the exchange around the BUTD engine (tests), the DP GPU tests, bench --butd under a one-rank RCCL group
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_dp.py -q -x -s > $OUT/s13_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; grep "dp \|passed\|failed\|Error" $OUT/s13_pytest.log | tail -30
[ $rc -eq 0 ] || { tail -40 $OUT/s13_pytest.log; exit 1; }
RGQA_BENCH_RCCL_REHEARSAL=1 timeout -k 10 300 python3 bench.py --butd --steps 30 --warmup 5 --no-cpu-baseline > $OUT/s13_butd_rccl.json 2> $OUT/s13_butd_rccl.err; echo "butd rccl rc=$?"; tail -3 $OUT/s13_butd_rccl.err; cut -c1-600 $OUT/s13_butd_rccl.json
