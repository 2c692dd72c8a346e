#!/bin/bash
# round-5 GPU session 16: the exchange on two side streams - DP tests, then the one-rank RCCL rehearsal with one and with two exchange streams
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
timeout -k 10 900 python3 -m pytest tests/test_gpu_dp.py tests/test_gpu_dropin.py -q -x > $OUT/s16_pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $OUT/s16_pytest.log
[ $rc -eq 0 ] || { tail -40 $OUT/s16_pytest.log; exit 1; }
for r in 1 2; do for n in 1 2; do
  RGQA_DP_EXCHANGE_STREAMS=$n RGQA_BENCH_RCCL_REHEARSAL=1 timeout -k 10 300 python3 bench.py --steps 60 --warmup 15 --no-cpu-baseline --no-extra-legs > $OUT/s16_rehearsal_$n.json 2> $OUT/s16_rehearsal_$n.err
  echo "exchange streams=$n: $(python3 -c "
import json; d=json.loads(open('$OUT/s16_rehearsal_$n.json').readline()); x=d['dp_exchange']; print(d['ms_per_step'], 'no exchange', x['no_exchange_ms_per_step'], 'exposed', x['exposed_comm_ms'], 'alt', x.get('alt_exposed_comm_ms'))")"
done; done
