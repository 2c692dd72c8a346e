#!/bin/bash
# a second build of the library for interleaved A/B runs (tools/ab_kernels.sh, tools/ab_lib.sh; RGQA_LIB selects it at load time):
#   tools/build_variant.sh <name> [extra hipcc flags ...]          -> rgqa_amd/lib/librgqa_hip_<name>.so   (objects under /tmp; git-ignored, ships to the GPU box)
#   PER_FILE="gemm_mfma256.hip:-fno-slp-vectorize" tools/build_variant.sh noslp      (flags for one source only)
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
OBJ=/tmp/rgqa_variant_$NAME; mkdir -p $OBJ
pids=()
for src in rgqa_amd/csrc/*.hip; do
  b=$(basename $src .hip)
  extra=""
  case $b in attn_mfma|attn_x3) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  for pf in ${PER_FILE:-}; do [ "${pf%%:*}" = "$b.hip" ] && extra="$extra ${pf#*:}"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result $extra "$@" -c $src -o $OBJ/$b.o &
  pids+=($!)
  if [ ${#pids[@]} -ge 8 ]; then wait ${pids[0]}; pids=("${pids[@]:1}"); fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o rgqa_amd/lib/librgqa_hip_$NAME.so $OBJ/*.o
echo built rgqa_amd/lib/librgqa_hip_$NAME.so
