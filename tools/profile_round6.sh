#!/bin/bash
# Round 6: reproduces the committed profiles of the round on ONE GPU box (tools/profile_round6.sh [tag]; writes gpurun_out/<tag>_*, copies what is judged into
# profiles/).  The headline is the bf16x3_fwd precision (bench.py's default); the bf16 engine's figures are taken beside it.  Counter passes are separate runs,
# each with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).  "Serial" runs fold every other stream into the launch stream
# (RGQA_WGRAD_SERIAL=1, RGQA_ADAM_OVERLAP=0): per-kernel durations and counters of a kernel alone on the chip.
set -u
TAG="${1:-r06}"; OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 2 --lean"
STEPS=7
BOXMS=$(python3 bench.py --lean --steps 60 --warmup 15 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
BOXID=$( (rocm-smi --showuniqueid 2>/dev/null | grep "GPU\[" | grep -i "unique id" | head -1 | sed 's/.*: *//') || true)
BOX="gpu ${BOXID:-unknown} host $(hostname) bf16x3_fwd lean step ${BOXMS:-?} ms ($(date -u +%Y-%m-%dT%H:%MZ))"
echo "$BOX" > $OUT/${TAG}_box.txt; echo "box: $BOX"
stamp() { python3 tools/stamp_box.py "$BOX" "$@"; }
# 1. fabric traffic of the NT launches: the headline's family (split-f32 forward launches) and the bf16 engine's (all NT launches)
for fam in x3fwd bf16; do
  if [ $fam = x3fwd ]; then PREC=bf16x3_fwd; FILT=x3fwd; else PREC=bf16; FILT=all; fi
  ok=1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/${TAG}_p_${fam}_$c
    RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_p_${fam}_$c -- $BENCH --precision $PREC > $OUT/${TAG}_pmc_${fam}_$c.log 2>&1; rc=$?; echo "pmc $fam $c rc=$rc"
    [ $rc -eq 0 ] || ok=0
  done
  if [ $ok -eq 1 ] && python3 tools/pmc_summary.py "$(ls $OUT/${TAG}_p_${fam}_FETCH_SIZE/*/*counter_collection.csv | head -1)" "$(ls $OUT/${TAG}_p_${fam}_WRITE_SIZE/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_gemm_nt_${fam}.json $FILT > /dev/null \
     && [ -s $OUT/${TAG}_pmc_gemm_nt_${fam}.json ]; then
    stamp $OUT/${TAG}_pmc_gemm_nt_${fam}.json
    cp $OUT/${TAG}_pmc_gemm_nt_${fam}.json profiles/${TAG}_pmc_gemm_nt_${fam}.json
  else
    echo "pmc traffic $fam: a pass failed - profiles/${TAG}_pmc_gemm_nt_${fam}.json left as it was"
  fi
  rm -rf $OUT/${TAG}_p_${fam}_FETCH_SIZE $OUT/${TAG}_p_${fam}_WRITE_SIZE
done
# 2. the driver's line (live HIP-event roofline, every leg, CPU baseline)
python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1.json 2> $OUT/${TAG}_bench_n1.err; echo "bench rc=$?"
stamp $OUT/${TAG}_bench_n1.json
# 3. kernel trace + stats: the shipped multi-stream configuration and the serialised one, three precisions
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats -- $BENCH > $OUT/${TAG}_stats.log 2>&1; echo "stats rc=$?"
for v in "x3f:bf16x3_fwd" "bf16:bf16" "x3:bf16x3"; do
  n=${v%%:*}; p=${v##*:}
  RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_p_stats_$n -- $BENCH --precision $p > $OUT/${TAG}_stats_$n.log 2>&1; echo "stats_$n rc=$?"
done
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_x3f/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_serial.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_bf16/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_bf16_serial.md > /dev/null
python3 tools/prof_summary.py $(ls $OUT/${TAG}_p_stats_x3/*/*kernel_stats.csv | head -1) $STEPS $OUT/${TAG}_kernel_stats_b256_bf16x3_serial.md > /dev/null
cp $(ls $OUT/${TAG}_p_stats_x3f/*/*kernel_stats.csv | head -1) $OUT/${TAG}_kernel_stats_b256_serial.csv
python3 tools/timeline.py $(ls $OUT/${TAG}_p_stats/*/*kernel_trace.csv | head -1) 12 > $OUT/${TAG}_timeline_b256.txt 2>&1
stamp $OUT/${TAG}_kernel_stats_b256.md $OUT/${TAG}_kernel_stats_b256_serial.md $OUT/${TAG}_kernel_stats_b256_bf16_serial.md $OUT/${TAG}_kernel_stats_b256_bf16x3_serial.md $OUT/${TAG}_timeline_b256.txt
# 4. per-launch tables (HIP events around every launch; GEMM launches grouped by shape)
for p in bf16x3_fwd bf16; do
  rm -f $OUT/${TAG}_pd.txt
  RGQA_PROF_DUMP=$PWD/$OUT/${TAG}_pd.txt python3 tools/prof_dump.py $p 3 > /dev/null 2>&1 && python3 tools/launch_table.py $OUT/${TAG}_pd.txt 3 > $OUT/${TAG}_launch_table_$p.txt
  rm -f $OUT/${TAG}_pd.txt
  stamp $OUT/${TAG}_launch_table_$p.txt
done
# 5. counters, one pass each (the headline precision): MFMA busy, L2 hit rates
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_p_mfma -- $BENCH > $OUT/${TAG}_pmc_mfma.log 2>&1; echo "pmc mfma rc=$?"
python3 tools/mfma_util.py $(ls $OUT/${TAG}_p_mfma/*/*counter_collection.csv | head -1) $OUT/${TAG}_pmc_mfma_util.json
RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_p_tcc -- $BENCH > $OUT/${TAG}_pmc_tcc.log 2>&1; echo "pmc tcc rc=$?"
python3 tools/tcc_hit.py "$(ls $OUT/${TAG}_p_tcc/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_tcc_hit.json "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum, RGQA_WGRAD_SERIAL=1 RGQA_ADAM_OVERLAP=0 $BENCH" > /dev/null
stamp $OUT/${TAG}_pmc_mfma_util.json $OUT/${TAG}_pmc_tcc_hit.json
rm -rf $OUT/${TAG}_p_stats $OUT/${TAG}_p_stats_x3f $OUT/${TAG}_p_stats_bf16 $OUT/${TAG}_p_stats_x3 $OUT/${TAG}_p_mfma $OUT/${TAG}_p_tcc
ls $OUT | grep "^${TAG}_"
