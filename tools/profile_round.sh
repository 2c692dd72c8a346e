#!/bin/bash
# Reproduces the committed profiles of a round on a GPU box: tools/profile_round.sh r02   (writes gpurun_out/<tag>_*, copy what is to be
# judged into profiles/).  Counter passes are separate runs, each with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -u
TAG="${1:-r02}"; OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --lean"
STEPS=7
run() { name=$1; shift; ( "$@" ) > $OUT/${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
# 1. the headline line (live HIP-event roofline + CPU baseline)
python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; echo "bench rc=$?"
# 2. kernel trace + stats: shipped two-stream configuration, and wgrad serialised (what bench.py's live timing sees)
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats -- $BENCH
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats_serial -- $BENCH > $OUT/${TAG}_stats_serial.log 2>&1; echo "stats_serial rc=$?"
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_serial/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_serial.md > /dev/null
cp $(ls $OUT/${TAG}_p_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_b256.csv
cp $(ls $OUT/${TAG}_p_stats_serial/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_b256_serial.csv
python3 tools/timeline.py $(ls $OUT/${TAG}_p_stats/*/*kernel_trace.csv | head -1) 12 > $OUT/${TAG}_timeline_b256.txt 2>&1
# 3. counters, one pass each
for c in FETCH_SIZE WRITE_SIZE; do
  RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_p_$c -- $BENCH > $OUT/${TAG}_pmc_$c.log 2>&1; echo "pmc $c rc=$?"
done
python3 tools/pmc_summary.py $(ls $OUT/${TAG}_p_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $OUT/${TAG}_p_WRITE_SIZE/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_gemm_nt.json
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_p_mfma -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
python3 tools/mfma_util.py $(ls $OUT/${TAG}_p_mfma/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_mfma_util.json
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_p_sq -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
python3 tools/sq_breakdown.py $(ls $OUT/${TAG}_p_sq/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_sq_wave_breakdown.json "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU, RGQA_WGRAD_SERIAL=1 $BENCH; fractions of wave cycles"
# keep the merge small: the raw traces stay on the box
rm -rf $OUT/${TAG}_p_stats $OUT/${TAG}_p_stats_serial $OUT/${TAG}_p_FETCH_SIZE $OUT/${TAG}_p_WRITE_SIZE $OUT/${TAG}_p_mfma $OUT/${TAG}_p_sq
ls $OUT | grep "^${TAG}_"
