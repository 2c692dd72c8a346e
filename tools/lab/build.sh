#!/bin/bash
# builds tools/lab/gemm_lab against rgqa_amd/lib/librgqa_hip.so (cross-compiles here, runs on the GPU box)
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 gemm_lab.cpp -o gemm_lab -L../../rgqa_amd/lib -lrgqa_hip -Wl,-rpath,'$ORIGIN/../../rgqa_amd/lib'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-result stamp_lab.cpp -o stamp_lab -L../../rgqa_amd/lib -lrgqa_hip -Wl,-rpath,'$ORIGIN/../../rgqa_amd/lib'
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -std=c++17 ln_lab.cpp -o ln_lab -L../../rgqa_amd/lib -lrgqa_hip -Wl,-rpath,'$ORIGIN/../../rgqa_amd/lib'
