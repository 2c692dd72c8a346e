#!/bin/bash
# register / scratch use of the kernels of one built object: tools/kernel_regs.sh gemm_mfma256 [name-filter]
# (unbundles the gfx950 code object out of rgqa_amd/csrc/build/<name>.o and reads its metadata notes)
set -e
o=/root/repo/rgqa_amd/csrc/build/$1.o
t=$(mktemp -d)
/opt/rocm/lib/llvm/bin/llvm-objcopy --dump-section .hip_fatbin=$t/fat.bin $o
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$t/fat.bin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$t/k.co
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $t/k.co | python3 -c "
import sys, re
txt = sys.stdin.read()
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for k in re.split(r'\n\s+- \.agpr_count', txt)[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    if flt and flt not in name: continue
    g = lambda f: re.search(r'\.' + f + r':\s+(\d+)', k).group(1)
    print('%-90s vgpr %3s spill %3s sgpr %3s scratch %4s lds %6s' % (name[:90], g('vgpr_count'), g('vgpr_spill_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size')))
" "$2"
rm -rf $t
