#!/bin/bash
# interleaved A/B of two builds of the library on the headline bench: tools/ab_lib.sh "libA.so libB.so" [rounds] [extra bench args]
LIBS=$1; R=${2:-2}; shift 2
for r in $(seq $R); do for l in $LIBS; do
  ms=$(RGQA_LIB=$PWD/$l python3 bench.py --lean --steps 60 --warmup 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])")
  echo "$l $* : $ms"
done; done
