#!/bin/bash
# round-5 GPU session 12: the gradient exchange around a BUTD engine (config 5 under DP), one-rank RCCL
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
for m in "sharded bf16" "allreduce bf16" "sharded f32" "sharded bf16x3"; do
  echo "== $m"; timeout -k 10 200 python3 tools/dp_butd_probe.py $m 2>&1 | grep -v "amdgpu.ids" | tail -8
done
