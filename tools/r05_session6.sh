#!/bin/bash
# round-5 GPU session 6: dgrad on the weights as they lie ([K, N] operand form): suite, interleaved A/B against the transposed copies
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=8 > $OUT/s6_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/s6_pytest.log
timeout -k 10 400 python3 tools/ab_debug.py 14 "1 0" 4 bf16 40 2>/dev/null | grep key > $OUT/s6_dgrad_nn_ab.txt; cat $OUT/s6_dgrad_nn_ab.txt
python3 bench.py --no-cpu-baseline --no-extra-legs --steps 60 > $OUT/s6_bench.json 2> $OUT/s6_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s6_bench.json 2>/dev/null | cut -c1-400
