#!/bin/bash
# interleaved A/B of one environment switch on the headline bench: tools/ab_env.sh VAR "v1 v2 .." [rounds] [extra bench args]
VAR=$1; VALS=$2; R=${3:-2}; shift 3
for r in $(seq $R); do for v in $VALS; do
  ms=$(env $VAR=$v python3 bench.py --lean --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); k=d.get('kernel_ms_per_step',{}); print(d['ms_per_step'], k.get('gemm_nt'), k.get('gemm_tn'))")
  echo "$VAR=$v $* : $ms"
done; done
