#!/bin/bash
# round-5 GPU session 18: stream priority for the optimizer pass beside the forward
set -u
OUT=gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
python3 -c "
import torch
print('priority range', torch.cuda.Stream.priority_range())
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print(p, '->', s.priority)
    except Exception as e: print(p, 'error', e)
" 2>&1 | grep -v amdgpu
for p in 0 1 -1; do echo "update stream priority $p"; RGQA_ADAM_STREAM_PRIORITY=$p timeout -k 10 300 python3 tools/ab_adam_overlap.py 3 bf16 40 2>/dev/null | grep adam_overlap; done
