#!/bin/bash
# interleaved A/B of two builds of the library with the live per-kernel-family timing: tools/ab_kernels.sh "libA.so libB.so" [rounds] [extra bench args]
LIBS=$1; R=${2:-2}; shift; [ $# -gt 0 ] && shift
for r in $(seq $R); do for l in $LIBS; do
  RGQA_LIB=$PWD/$l python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extra-legs "$@" 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.readline()); k=d.get('kernel_ms_per_step',{}); print('$l', d['ms_per_step'], 'nt', k.get('gemm_nt'), 'tn', k.get('gemm_tn'), 'attn', k.get('attn_fwd'), k.get('attn_bwd'), 'ln', k.get('layernorm'), 'other', k.get('other'), 'roof', d['roofline']['frac'])"
done; done
