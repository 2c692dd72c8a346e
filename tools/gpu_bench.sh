#!/bin/bash
# the default bench.py line on one GPU box, stamped with the box: tools/gpu.sh --timeout 900 -- "bash tools/gpu_bench.sh r05"
set -u
TAG="${1:-r05}"; OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
BOXMS=$(python3 bench.py --lean --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
BOXID=$( (rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | grep -i "unique id" | head -1 | sed 's/.*: *//') || true)
BOX="gpu ${BOXID:-unknown} host $(hostname) bf16 lean step ${BOXMS:-?} ms ($(date -u +%Y-%m-%dT%H:%MZ))"
python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; echo "bench rc=$?"
python3 tools/stamp_box.py "$BOX" $OUT/${TAG}_bench_n1.json
python3 tools/show_bench.py $OUT/${TAG}_bench_n1.json 2>/dev/null | cut -c1-200
