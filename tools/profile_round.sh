#!/bin/bash
# Reproduces the committed profiles of a round on ONE GPU box: tools/profile_round.sh r04   (writes gpurun_out/<tag>_*; copy what is to be
# judged into profiles/).  Counter passes are separate runs, each with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
# "Serial" runs fold every other stream into the launch stream (RGQA_WGRAD_SERIAL=1: the weight-gradient side stream; RGQA_ADAM_OVERLAP=0: the optimizer
# pass that otherwise runs beside the next forward): per-kernel durations and counters of a kernel alone on the chip.
# Every file it writes is stamped with the box (GPU unique id, host) and the lean single-GPU step measured on that box first, so that
# figures from different files - different boxes of the pool differ by +-3 % - can be put side by side (tools/stamp_box.py).
set -u
TAG="${1:-r04}"; OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --lean"
STEPS=7
# 0. the box and its step
BOXMS=$(python3 bench.py --lean --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
BOXID=$( (rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | grep -i "unique id" | head -1 | sed 's/.*: *//') || true)
BOX="gpu ${BOXID:-unknown} host $(hostname) bf16 lean step ${BOXMS:-?} ms ($(date -u +%Y-%m-%dT%H:%MZ))"
echo "$BOX" > $OUT/${TAG}_box.txt; echo "box: $BOX"
stamp() { python3 tools/stamp_box.py "$BOX" "$@"; }
# 1. fabric traffic of the NT launches (separate FETCH_SIZE / WRITE_SIZE passes): bench.py reports `roofline.traffic` from
#    profiles/<tag>_pmc_gemm_nt.json only while its kernel-source digest matches the build it runs on
rm -f $OUT/${TAG}_pmc_gemm_nt.json
pmc_ok=1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_p_$c
  RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_p_$c -- $BENCH > $OUT/${TAG}_pmc_$c.log 2>&1; rc=$?; echo "pmc $c rc=$rc"
  [ $rc -eq 0 ] || pmc_ok=0
done
# the committed profile is replaced only by a complete, fresh one: both passes and the summary must have succeeded
if [ $pmc_ok -eq 1 ] && python3 tools/pmc_summary.py "$(ls $OUT/${TAG}_p_FETCH_SIZE/*/*counter_collection.csv | head -1)" "$(ls $OUT/${TAG}_p_WRITE_SIZE/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_gemm_nt.json \
   && [ -s $OUT/${TAG}_pmc_gemm_nt.json ]; then
  stamp $OUT/${TAG}_pmc_gemm_nt.json
  cp $OUT/${TAG}_pmc_gemm_nt.json profiles/${TAG}_pmc_gemm_nt.json
else
  echo "pmc traffic: a pass failed - profiles/${TAG}_pmc_gemm_nt.json left as it was"
fi
# 2. the headline line (live HIP-event roofline, tolerance_compliant / tolerance_compliant_fwd / forward-only / drop-in legs, CPU baseline)
python3 bench.py > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; echo "bench rc=$?"
stamp $OUT/${TAG}_bench_n1.json
# 3. kernel trace + stats: shipped two-stream configuration, and wgrad serialised (what bench.py's live timing sees), three precisions
run() { name=$1; shift; ( "$@" ) > $OUT/${TAG}_$name.log 2>&1; echo "$name rc=$?"; }
run stats rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats -- $BENCH
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats_serial -- $BENCH > $OUT/${TAG}_stats_serial.log 2>&1; echo "stats_serial rc=$?"
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats_x3 -- $BENCH --precision bf16x3 > $OUT/${TAG}_stats_x3.log 2>&1; echo "stats_x3 rc=$?"
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats_x3f -- $BENCH --precision bf16x3_fwd > $OUT/${TAG}_stats_x3f.log 2>&1; echo "stats_x3_fwd rc=$?"
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_serial/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_serial.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_x3/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_bf16x3_serial.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_x3f/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_bf16x3_fwd_serial.md > /dev/null
cp $(ls $OUT/${TAG}_p_stats_serial/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_b256_serial.csv
python3 tools/timeline.py $(ls $OUT/${TAG}_p_stats/*/*kernel_trace.csv | head -1) 12 > $OUT/${TAG}_timeline_b256.txt 2>&1
stamp $OUT/${TAG}_kernel_stats_b256.md $OUT/${TAG}_kernel_stats_b256_serial.md $OUT/${TAG}_kernel_stats_b256_bf16x3_serial.md $OUT/${TAG}_kernel_stats_b256_bf16x3_fwd_serial.md $OUT/${TAG}_timeline_b256.txt
# 4. per-launch tables (HIP events around every launch; GEMM launches grouped by shape)
rm -f $OUT/${TAG}_pd.txt $OUT/${TAG}_pdx.txt
RGQA_PROF_DUMP=$PWD/$OUT/${TAG}_pd.txt python3 tools/prof_dump.py bf16 3 > /dev/null 2>&1 && python3 tools/launch_table.py $OUT/${TAG}_pd.txt 3 > $OUT/${TAG}_launch_table_bf16.txt
RGQA_PROF_DUMP=$PWD/$OUT/${TAG}_pdx.txt python3 tools/prof_dump.py bf16x3 3 > /dev/null 2>&1 && python3 tools/launch_table.py $OUT/${TAG}_pdx.txt 3 x3 > $OUT/${TAG}_launch_table_bf16x3.txt
rm -f $OUT/${TAG}_pd.txt $OUT/${TAG}_pdx.txt
stamp $OUT/${TAG}_launch_table_bf16.txt $OUT/${TAG}_launch_table_bf16x3.txt
# 5. counters, one pass each: MFMA busy, wave-cycle breakdown, L2 hit rates
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_p_mfma -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
python3 tools/mfma_util.py $(ls $OUT/${TAG}_p_mfma/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_mfma_util.json
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/${TAG}_p_sq -- $BENCH > $OUT/${TAG}_pmc_sq.log 2>&1; echo "pmc sq rc=$?"
python3 tools/sq_breakdown.py $(ls $OUT/${TAG}_p_sq/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_sq_wave_breakdown.json "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU, RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 $BENCH; fractions of wave cycles"
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_p_tcc -- $BENCH > $OUT/${TAG}_pmc_tcc.log 2>&1; echo "pmc tcc rc=$?"
python3 tools/tcc_hit.py "$(ls $OUT/${TAG}_p_tcc/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_tcc_hit.json "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum, RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 $BENCH" > /dev/null
stamp $OUT/${TAG}_pmc_mfma_util.json $OUT/${TAG}_pmc_sq_wave_breakdown.json $OUT/${TAG}_pmc_tcc_hit.json
# keep the merge small: the raw traces stay on the box
rm -rf $OUT/${TAG}_p_stats $OUT/${TAG}_p_stats_serial $OUT/${TAG}_p_stats_x3 $OUT/${TAG}_p_stats_x3f $OUT/${TAG}_p_FETCH_SIZE $OUT/${TAG}_p_WRITE_SIZE $OUT/${TAG}_p_mfma $OUT/${TAG}_p_sq $OUT/${TAG}_p_tcc
# 6. in-kernel phase stamps of the NT kernels; the vendor-GEMM yardstick; what wgrad and the clip norm cost the step
timeout -k 10 200 python3 tools/nt_stamps.py > $OUT/${TAG}_nt_stamps.txt 2>/dev/null; echo "stamps rc=$?"
timeout -k 10 300 python3 tools/vendor_gemm.py > $OUT/${TAG}_vendor_gemm.txt 2>/dev/null; echo "vendor rc=$?"
timeout -k 10 200 python3 tools/wgrad_probe.py 30 3 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_wgrad_probe.txt; echo "wgrad probe rc=$?"
timeout -k 10 200 python3 tools/wgrad_lab.py 2>/dev/null | grep -v amdgpu > $OUT/${TAG}_wgrad_lab.txt; echo "wgrad lab rc=$?"
timeout -k 10 300 python3 tools/ab_debug.py 6 "1 2 3 4" 3 bf16 2>/dev/null | grep key > $OUT/${TAG}_wgrad_merge_ab.txt; echo "wgrad merge ab rc=$?"
timeout -k 10 200 python3 tools/sumsq_probe.py bf16 3 2>/dev/null | grep median > $OUT/${TAG}_sumsq_probe.txt; echo "sumsq probe rc=$?"
stamp $OUT/${TAG}_nt_stamps.txt $OUT/${TAG}_vendor_gemm.txt $OUT/${TAG}_wgrad_probe.txt $OUT/${TAG}_sumsq_probe.txt $OUT/${TAG}_wgrad_lab.txt $OUT/${TAG}_wgrad_merge_ab.txt
ls $OUT | grep "^${TAG}_"
