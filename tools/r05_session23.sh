#!/bin/bash
set -u
export TMPDIR=/tmp
python3 -c "from rgqa_amd import _lib; _lib.load()" || { echo "stale library in the snapshot"; exit 1; }
bash tools/profile_round.sh r05
