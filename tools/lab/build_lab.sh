#!/bin/bash
# builds the lab library against the product library in the tree: tools/lab/build_lab.sh   (-> tools/lab/libffn_chain.so; ships to the GPU box, git-ignored)
set -e
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -shared tools/lab/ffn_chain.hip -I rgqa_amd/csrc \
    -L rgqa_amd/lib -lrgqa_hip -Wl,-rpath,'$ORIGIN/../../rgqa_amd/lib' -o tools/lab/libffn_chain.so "$@"
echo built tools/lab/libffn_chain.so
