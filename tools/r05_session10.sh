#!/bin/bash
# round-5 GPU session 10: after the GRU's second form - full GPU suite, then the PMC traffic passes + the bench line + the BUTD / headline kernel stats on the final sources
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=8 > $OUT/s10_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $OUT/s10_pytest.log
[ $rc -eq 0 ] || exit 1
python3 __graft_entry__.py --smoke > $OUT/s10_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $OUT/s10_smoke.log
TAG=r05
BENCH="python3 bench.py --steps 5 --warmup 2 --lean"
BOXMS=$(python3 bench.py --lean --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
BOXID=$( (rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | grep -i "unique id" | head -1 | sed 's/.*: *//') || true)
BOX="gpu ${BOXID:-unknown} host $(hostname) bf16 lean step ${BOXMS:-?} ms ($(date -u +%Y-%m-%dT%H:%MZ))"
echo "box: $BOX"
rm -f $OUT/${TAG}_pmc_gemm_nt.json
pmc_ok=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_p_$c
  RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_p_$c -- $BENCH > $OUT/${TAG}_pmc_$c.log 2>&1; rc=$?; echo "pmc $c rc=$rc"
  [ $rc -eq 0 ] || pmc_ok=0
done
if [ $pmc_ok -eq 1 ] && python3 tools/pmc_summary.py "$(ls $OUT/${TAG}_p_FETCH_SIZE/*/*counter_collection.csv | head -1)" "$(ls $OUT/${TAG}_p_WRITE_SIZE/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_gemm_nt.json \
   && [ -s $OUT/${TAG}_pmc_gemm_nt.json ]; then
  python3 tools/stamp_box.py "$BOX" $OUT/${TAG}_pmc_gemm_nt.json
  cp $OUT/${TAG}_pmc_gemm_nt.json profiles/${TAG}_pmc_gemm_nt.json
fi
rm -rf $OUT/${TAG}_p_FETCH_SIZE $OUT/${TAG}_p_WRITE_SIZE
python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; echo "bench rc=$?"
python3 tools/stamp_box.py "$BOX" $OUT/${TAG}_bench_n1.json
python3 tools/show_bench.py $OUT/${TAG}_bench_n1.json 2>/dev/null | cut -c1-300
