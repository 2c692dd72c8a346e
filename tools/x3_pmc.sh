set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
BENCH="python3 bench.py --steps 3 --warmup 2 --lean --precision bf16x3"
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_]*LDS[A-Z_]*" $OUT/counters.txt | sort -u > $OUT/lds_counters.txt
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/x3_sq -- $BENCH > $OUT/x3_sq.log 2>&1 && \
python3 tools/sq_breakdown.py $(ls $OUT/x3_sq/*/*counter_collection.csv | head -1) $OUT/x3_sq.json x3 && \
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/x3_mfma -- $BENCH > $OUT/x3_mfma.log 2>&1 && \
python3 tools/mfma_util.py $(ls $OUT/x3_mfma/*/*counter_collection.csv | head -1) $OUT/x3_mfma.json && \
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/x3_lds -- $BENCH > $OUT/x3_lds.log 2>&1
python3 - <<'PY'
import csv,collections,glob
f=glob.glob('gpurun_out/x3_lds/*/*counter_collection.csv')
if f:
    acc=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"].split('(')[0][:60]
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    for k,a in sorted(acc.items(), key=lambda x:-x[1].get("SQ_WAVE_CYCLES",0))[:12]:
        print(k, dict(a))
PY
rm -rf $OUT/x3_sq $OUT/x3_mfma $OUT/x3_lds
