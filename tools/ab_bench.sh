#!/bin/bash
# A/B of two environment settings on one box, interleaved: tools/ab_bench.sh "ENV_A=1" "ENV_B=1" [rounds] [bench args...]
A="$1"; B="$2"; R="${3:-3}"; shift 3 || true
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "round $i $tag [$E]: $ms ms"
  done
done
