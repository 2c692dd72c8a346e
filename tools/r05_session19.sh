#!/bin/bash
# round-5 GPU session 19: the optimizer pass beside the next forward as the default - full GPU suite, A/B, bench
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=10 > $OUT/s19_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $OUT/s19_pytest.log
[ $rc -eq 0 ] || { grep -n "Error\|assert\|FAILED" $OUT/s19_pytest.log | head -40; }
timeout -k 10 400 python3 tools/ab_adam_overlap.py 4 bf16 40 2>/dev/null | grep adam_overlap > $OUT/s19_adam_overlap_ab.txt; cat $OUT/s19_adam_overlap_ab.txt
timeout -k 10 200 python3 tools/dropin_profile.py 40 2>/dev/null | grep step
RGQA_ADAM_OVERLAP=0 timeout -k 10 200 python3 tools/dropin_profile.py 40 2>/dev/null | grep step
python3 bench.py --no-cpu-baseline > $OUT/s19_bench.json 2> $OUT/s19_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s19_bench.json 2>/dev/null | cut -c1-330
