#!/bin/bash
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 400 python3 -m pytest tests/test_gpu_engine.py -m gpu -q -x -k "optimizer_pass_beside" > $OUT/s20_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $OUT/s20_pytest.log
for i in 1 2; do
python3 bench.py --no-cpu-baseline --no-extra-legs > $OUT/s20_bench.json 2> $OUT/s20_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s20_bench.json 2>/dev/null | grep "ms_per_step\|roofline\|kernel_ms" | cut -c1-200
RGQA_ADAM_OVERLAP=0 python3 bench.py --no-cpu-baseline --no-extra-legs > $OUT/s20_bench0.json 2> $OUT/s20_bench0.err; echo "bench (overlap off) rc=$?"
python3 tools/show_bench.py $OUT/s20_bench0.json 2>/dev/null | grep "ms_per_step\|roofline\|kernel_ms" | cut -c1-200
done
