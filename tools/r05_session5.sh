#!/bin/bash
# round-5 GPU session 5: the whole GPU suite + smoke + the default bench line on the current tree
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=8 > $OUT/s5_pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/s5_pytest.log
python3 __graft_entry__.py --smoke > $OUT/s5_smoke.log 2>&1; echo "smoke rc=$?"
python3 bench.py > $OUT/s5_bench.json 2> $OUT/s5_bench.err; echo "bench rc=$?"; grep "bench.py \[" $OUT/s5_bench.err | tail -25
python3 tools/show_bench.py $OUT/s5_bench.json 2>/dev/null | cut -c1-700
