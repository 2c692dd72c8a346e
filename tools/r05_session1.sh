#!/bin/bash
# round-5 GPU session 1: the test suite on the round's first build, the default bench line, where BUTD's and the drop-in step's time goes
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests -m gpu -q --maxfail=8 > $OUT/s1_pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/s1_pytest.log
python3 __graft_entry__.py --smoke > $OUT/s1_smoke.log 2>&1; echo "smoke rc=$?"
python3 bench.py > $OUT/s1_bench.json 2> $OUT/s1_bench.err; echo "bench rc=$?"
python3 tools/show_bench.py $OUT/s1_bench.json 2>/dev/null | head -60
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1_p_butd -- python3 bench.py --butd --lean --steps 20 --warmup 5 > $OUT/s1_butd.log 2>&1; echo "butd prof rc=$?"
python3 tools/prof_summary.py $(ls $OUT/s1_p_butd/*/*kernel_stats.csv | head -1) 25 $OUT/s1_butd_kernel_stats.md > /dev/null; echo "butd summary rc=$?"
python3 tools/dropin_profile.py 30 > $OUT/s1_dropin.txt 2>&1; echo "dropin rc=$?"; grep "ms/step" $OUT/s1_dropin.txt
RGQA_DROPIN_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s1_p_dropin -- python3 tools/dropin_profile.py 20 > $OUT/s1_dropin_prof.log 2>&1; echo "dropin prof rc=$?"
python3 tools/prof_summary.py $(ls $OUT/s1_p_dropin/*/*kernel_stats.csv | head -1) 23 $OUT/s1_dropin_kernel_stats.md > /dev/null
cp $(ls $OUT/s1_p_dropin/*/*kernel_stats.csv | head -1) $OUT/s1_dropin_kernel_stats.csv
timeout -k 10 300 python3 tools/ab_debug.py 17 "6 8" 3 bf16 40 2>/dev/null | grep key > $OUT/s1_wgrad_sets_ab.txt; echo "sets ab rc=$?"; cat $OUT/s1_wgrad_sets_ab.txt
rm -rf $OUT/s1_p_butd $OUT/s1_p_dropin
