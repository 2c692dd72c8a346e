"""Condenses two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command, as
MI355X_MICROARCH.md prescribes: TCC counters do not fit one pass) into the per-launch HBM-side traffic of the NT GEMM kernels.
FETCH_SIZE is in KiB-like units of 1024 B and, on gfx950, reports half of a wide coalesced streaming read: doubled here.
usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [all|x3fwd]"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rgqa_amd.build import source_digest


def per_kernel(path, counter, match):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and match(r["Kernel_Name"]):
            tot += float(r["Counter_Value"])
            n += 1
    return tot, n


def main(fetch_csv, write_csv, out, family="all"):
    # family: "all" = every NT GEMM launch (the bf16 engine: one kernel family computes forward and dgrad launches); "x3fwd" = the split-f32 forward launches only
    # (gemm_nt256*_kernel<sf32, ...> / <float, ..., X3 = true>: the dominant kernel of the bf16x3_fwd headline); the per-kernel table below holds every NT instantiation
    def is_nt(k):
        if "gemm_nt" not in k:
            return False
        if family == "x3fwd":
            return "sf32" in k or "4sf32" in k or ("<float" in k and ", true," in k)
        return True
    f, nf = per_kernel(fetch_csv, "FETCH_SIZE", is_nt)
    w, nw = per_kernel(write_csv, "WRITE_SIZE", is_nt)
    fetch = f * 1024.0 * 2.0 / max(nf, 1)
    write = w * 1024.0 / max(nw, 1)
    # per kernel instantiation: launches and bytes per launch (VERDICT r5 #3b: which launch class moves more than its algorithmic bytes)
    per = {}
    for path, counter, scale in ((fetch_csv, "FETCH_SIZE", 2048.0), (write_csv, "WRITE_SIZE", 1024.0)):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter and "gemm_nt" in r["Kernel_Name"]:
                e = per.setdefault(r["Kernel_Name"][:110], {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
                e[counter][0] += float(r["Counter_Value"]) * scale
                e[counter][1] += 1
    table = {k: dict(launches=v["FETCH_SIZE"][1], fetch_mb_per_launch=round(v["FETCH_SIZE"][0] / max(1, v["FETCH_SIZE"][1]) / 1e6, 2),
                     write_mb_per_launch=round(v["WRITE_SIZE"][0] / max(1, v["WRITE_SIZE"][1]) / 1e6, 2)) for k, v in sorted(per.items())}
    res = dict(kernel="gemm_nt (%s NT GEMM launches of bench.py, B=256, T=20, packed language rows)" % ("the split-f32 forward" if family == "x3fwd" else "all"), launches_sampled=nf, per_kernel=table,
               kernel_source_digest=source_digest(), fetch_bytes_per_launch=fetch, write_bytes_per_launch=write, traffic_bytes_per_launch=fetch + write,
               note="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; x1024 -> bytes; FETCH_SIZE doubled per the gfx950 "
                    "correction for 16-B/lane coalesced streams; WRITE_SIZE as read; Infinity-Cache hits are counted (fabric-side requests)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(*sys.argv[1:5])
