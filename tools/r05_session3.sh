#!/bin/bash
# round-5 GPU session 3: persistent GRU (tests, A/B, trace), deferred-clip drop-in (timeline of the two queues), sharded-exchange rehearsal on RCCL,
# GELU epilogue with / without the compiler's packed-f32 (SLP) instructions
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -m pytest tests/test_gpu_butd.py tests/test_gpu_dropin.py tests/test_gpu_dp.py -q --maxfail=10 > $OUT/s3_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/s3_pytest.log
for v in 1 0; do
  echo "butd GRU persistent=$v"; python3 - <<PY 2>/dev/null
import sys; sys.path.insert(0, '.')
from rgqa_amd import _lib
lib = _lib.load(); lib.rgqa_debug_set(18, $v)
import bench, torch
for r in range(3):
    print("  butd step %.3f ms" % bench.butd_leg(256, 30))
PY
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s3_p_butd -- python3 bench.py --butd --lean --steps 20 --warmup 5 > $OUT/s3_butd.log 2>&1; echo "butd prof rc=$?"
python3 tools/prof_summary.py $(ls $OUT/s3_p_butd/*/*kernel_stats.csv | head -1) 25 $OUT/s3_butd_kernel_stats.md > /dev/null; head -40 $OUT/s3_butd_kernel_stats.md | tail -28
python3 tools/dropin_profile.py 30 > $OUT/s3_dropin.txt 2>&1; grep "ms/step" $OUT/s3_dropin.txt
RGQA_DROPIN_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/s3_p_dropin -- python3 tools/dropin_profile.py 12 > $OUT/s3_dropin_prof.log 2>&1; echo "dropin trace rc=$?"
python3 tools/timeline.py $(ls $OUT/s3_p_dropin/*/*kernel_trace.csv | head -1) 14 3 > $OUT/s3_dropin_timeline.txt 2>&1; head -30 $OUT/s3_dropin_timeline.txt
RGQA_BENCH_RCCL_REHEARSAL=1 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 40 > $OUT/s3_rccl.json 2> $OUT/s3_rccl.err; echo "rccl rehearsal rc=$?"
python3 -c "
import json; d=json.load(open('$OUT/s3_rccl.json')); print('rccl rehearsal:', d['ms_per_step'], d.get('exposed_comm_ms'), json.dumps(d.get('dp_exchange')))"
RGQA_DP_GATHER_OVERLAP=0 RGQA_BENCH_RCCL_REHEARSAL=1 python3 bench.py --no-cpu-baseline --no-extra-legs --steps 40 > $OUT/s3_rccl_nogo.json 2> $OUT/s3_rccl_nogo.err; echo "rccl rehearsal (gather on the step's stream) rc=$?"
python3 -c "
import json; d=json.load(open('$OUT/s3_rccl_nogo.json')); print('rccl rehearsal, no gather overlap:', d['ms_per_step'], d.get('exposed_comm_ms'), json.dumps(d.get('dp_exchange')))"
bash tools/ab_kernels.sh "rgqa_amd/lib/librgqa_hip.so rgqa_amd/lib/librgqa_hip_noslp.so" 3 > $OUT/s3_slp_ab.txt 2>&1; cat $OUT/s3_slp_ab.txt
rm -rf $OUT/s3_p_butd $OUT/s3_p_dropin
