"""Condenses a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into the per-round summary kept under profiles/:
per-kernel rows plus the same category buckets bench.py's live HIP-event timing uses (gemm_nt = every NT GEMM kernel)."""
import csv
import json
import sys


def category(name):
    if "gemm_nt" in name:
        return "gemm_nt"
    if "gemm_tn" in name:
        return "gemm_tn"
    if "attn_fwd" in name:
        return "attn_fwd"
    if "attn_bwd" in name:
        return "attn_bwd"
    if "ln_fwd" in name or "ln_bwd" in name:
        return "layernorm"
    return "other"


def main(path, steps, out_md):
    rows = list(csv.DictReader(open(path)))
    # one-off set-up work of the process is not part of a step: the spin kernels of the stream-placement test (rgqa_amd/streams.py)
    setup = [r for r in rows if "spin_kernel" in r["Name"]]
    rows = [r for r in rows if "spin_kernel" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    cats = {}
    for r in rows:
        c = cats.setdefault(category(r["Name"]), dict(ns=0.0, calls=0))
        c["ns"] += float(r["TotalDurationNs"])
        c["calls"] += int(r["Calls"])
    with open(out_md, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (%d bench steps incl. warm-up)\n\n" % steps)
        f.write("source: `%s`; command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --lean`\n\n" % path)
        f.write("Sum of kernel durations per step: **%.3f ms** (kernels of the side stream - the deferred weight-gradient GEMMs - run BESIDE the main "
                "stream's LayerNorm/attention/dgrad kernels, so durations overlap in time and a kernel that waits for CUs shows a longer span: the sum "
                "exceeds the measured step time; `bench.py`'s live HIP-event figures are taken with the side stream folded into the main one)\n\n## categories (as in bench.py)\n\n| category | launches/step | ms/step | avg launch us | share |\n|---|---|---|---|---|\n" % (tot / 1e6 / steps))
        for k, c in sorted(cats.items(), key=lambda kv: -kv[1]["ns"]):
            f.write("| %s | %.1f | %.3f | %.2f | %.1f%% |\n" % (k, c["calls"] / steps, c["ns"] / 1e6 / steps, c["ns"] / c["calls"] / 1e3, 100 * c["ns"] / tot))
        if setup:
            f.write("\n(left out: %d spin kernels of the stream-placement test at start-up, %.1f ms in all - rgqa_amd/streams.py)\n" % (
                sum(int(r["Calls"]) for r in setup), sum(float(r["TotalDurationNs"]) for r in setup) / 1e6))
        f.write("\n## kernels\n\n| kernel | calls/step | ms/step | avg us | min us | max us | share |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            if float(r["TotalDurationNs"]) / tot < 0.0005:
                continue
            f.write("| `%s` | %.1f | %.3f | %.1f | %.1f | %.1f | %.1f%% |\n" % (r["Name"][:90], int(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
                                                                     float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print(json.dumps({k: dict(ms_per_step=c["ns"] / 1e6 / steps, avg_us=c["ns"] / c["calls"] / 1e3) for k, c in cats.items()}))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3])
