#!/bin/bash
# round-5 GPU session 7: LayerNorm inside the projection launches (key 19): parity tests first, then interleaved A/B on the step, then a lean bench
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 300 python3 -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "layernorm_inside or dgrad_on_the_weights" > $OUT/s7_pytest_ln.log 2>&1; rc=$?; echo "ln tests rc=$rc"; tail -15 $OUT/s7_pytest_ln.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python3 tools/ab_debug.py 19 "0 1" 4 bf16 40 2>/dev/null | grep key > $OUT/s7_ln_fuse_ab.txt; cat $OUT/s7_ln_fuse_ab.txt
python3 bench.py --no-cpu-baseline --no-extra-legs --steps 60 > $OUT/s7_bench.json 2> $OUT/s7_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s7_bench.json 2>/dev/null | cut -c1-400
