"""CPU-only: the library's host translation units under AddressSanitizer + UBSan (GPU sanitizers are not available on
this pool, and this file is listed in .gpurunignore so that it never travels to a GPU box)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_units_under_address_sanitizer(golden_dir, tmp_path):
    """The library's host translation units (tokenizer.hip, hostio.hip, runtime.hip) built with -fsanitize=address,undefined (host
    side only: GPU ASan is not available on this pool) and driven through the C ABI: the golden tokenizer vectors (G4) come out
    identical, edge inputs (empty sentence, 300-character word, control characters, 400 tokens into T=20, T < 2, a missing
    vocabulary, an out-of-range store row) raise no sanitizer report and no leak."""
    import shutil
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    exe = str(tmp_path / "host_asan")
    r = subprocess.run([hipcc, "-x", "hip", "--offload-host-only", "--offload-arch=gfx950", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                        "-fno-omit-frame-pointer", "-Wno-unused-result", os.path.join(ROOT, "tests", "asan", "host_asan.cpp"), "-o", exe],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    g = json.load(open(os.path.join(golden_dir, "g4_tokenizer.json")))
    idx = [i for i, sn in enumerate(g["sentences"]) if all(ord(c) < 128 for c in sn) and "\n" not in sn]
    sf = tmp_path / "sents.txt"
    sf.write_text("\n".join(g["sentences"][i] for i in idx) + "\n")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    for T in (20, 30):
        r = subprocess.run([exe, os.path.join(golden_dir, "g4_vocab.txt"), str(sf), str(T)], capture_output=True, text=True, env=env, timeout=120)
        assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
        lines = r.stdout.strip().split("\n")
        assert lines[-1].startswith("gather ok") and len(lines) == len(idx) + 4
        native = 0
        for k, i in enumerate(idx):
            f = lines[k].split()
            if f[0] == "0":      # encoded natively (not handed back to the Python rules)
                assert [int(x) for x in f[2:]] == g["T%d" % T]["input_ids"][i], g["sentences"][i]
                assert int(f[1]) == sum(g["T%d" % T]["input_mask"][i])
                native += 1
        assert native >= len(idx) - 2
        for extra in lines[len(idx):len(idx) + 3]:       # the three edge sentences: well-formed rows, length within [2, T]
            f = extra.split()
            assert 2 <= int(f[1]) <= T and len(f) == 2 + T
