"""Builds rgqa_amd/lib/librgqa_hip.so from rgqa_amd/csrc/*.hip with hipcc for gfx950 (cross-compiles without a GPU).

    python -m rgqa_amd.build [--force]
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib", "librgqa_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result"] + os.environ.get("RGQA_EXTRA_HIPCC_FLAGS", "").split()
# Per-file additions.  attn_mfma.hip: MFMA results straight into VGPRs - hipcc otherwise parks the small accumulators of these one-wave kernels in
# AGPRs and copies every one out with v_accvgpr_read (216 of 3197 instructions in attn_bwd1<3,3>, a kernel at the instruction-issue limit);
# with the flag that kernel also fits three waves per SIMD (164 registers instead of 158 + 16).
PER_FILE_FLAGS = {"attn_mfma.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attn_x3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _newest_header():
    hs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
            and os.path.getmtime(obj) >= _newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + PER_FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def source_digest():
    """sha1 over the kernel sources (csrc/*.hip, csrc/*.h, include/*.h): ties a committed counter profile to the code it measured"""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    # objects whose source is gone (a deleted or renamed unit) would still travel to the GPU box: prune them
    live = {os.path.basename(s)[:-4] + ".o" for s in srcs}
    for o in glob.glob(os.path.join(OBJ, "*.o")):
        if os.path.basename(o) not in live:
            os.remove(o)
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB) or force:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("linked", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
