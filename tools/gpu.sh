#!/bin/bash
# local wrapper around gpurun: the library in the tree must be the one built from the tree's sources (a stale .so travels to the box as it is)
#   tools/gpu.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python3 -c "import __graft_entry__ as g; g.build()"
python3 -c "from rgqa_amd import _lib; _lib.load(); print('library ok: every symbol of _lib.SIGNATURES resolves')"
exec /usr/local/graft/bin/gpurun "$@"
