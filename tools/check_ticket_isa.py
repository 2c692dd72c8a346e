#!/usr/bin/env python3
"""The persistent NT GEMM requests its next tile ticket with an inline-asm `global_atomic_add` whose result register hipcc believes
is valid at once; the kernel reads it only after the K loop (whose first step executes s_waitcnt vmcnt(0)).  This checks, on the
compiled ISA of every instantiation, that no instruction reads or writes that register between the atomic and the next
`s_waitcnt vmcnt(0)`.  usage: tools/check_ticket_isa.py  (compiles rgqa_amd/csrc/gemm_mfma256.hip to assembly under /tmp)"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "rgqa_amd", "csrc", "gemm_mfma256.hip")
out = os.path.join(tempfile.gettempdir(), "rgqa_ticket_check.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-result", "-I" + os.path.join(root, "include"),
                       "-I" + os.path.dirname(src), "-S", "--cuda-device-only", src, "-o", out], stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
kern, bad, checked = None, [], 0
i = 0
while i < len(lines):
    l = lines[i]
    m = re.match(r"^(_Z\w*gemm_nt256_kernel\w*):", l)
    if m:
        kern = m.group(1)
    if kern and "s_endpgm" in l:
        kern = None
    if kern and "global_atomic_add" in l and "s_waitcnt" not in lines[i + 1]:       # the spare-block atomic carries its own wait
        reg = l.split()[1].rstrip(",")
        j = i + 1
        while j < len(lines) and "s_waitcnt vmcnt(0)" not in lines[j]:
            code = lines[j].split(";")[0]
            if re.search(r"\b%s\b" % re.escape(reg), code) or re.search(r"v\[(\d+):(\d+)\]", code) and any(int(a) <= int(reg[1:]) <= int(b) for a, b in re.findall(r"v\[(\d+):(\d+)\]", code)):
                bad.append((kern, j + 1, lines[j].strip()))
            if "s_endpgm" in lines[j]:
                bad.append((kern, j + 1, "no wait before the end of the kernel"))
                break
            j += 1
        checked += 1
    i += 1
print("%d ticket atomics checked, %d violations" % (checked, len(bad)))
for b in bad:
    print("  %s line %d: %s" % b)
sys.exit(1 if bad or checked == 0 else 0)
