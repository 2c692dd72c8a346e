"""Where the waves of each kernel family spend their cycles, from a rocprofv3
`--pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU` pass (MI355X_MICROARCH.md, PMC slots:
WAIT_ANY = parked at s_waitcnt / s_barrier, WAIT_INST_ANY = issue stalls, ACTIVE_INST_ANY = issuing; fractions of SQ_WAVE_CYCLES).
usage: sq_breakdown.py <counter_collection.csv> <out.json> [note]"""
import collections, csv, json, sys


def family(k):
    for pat, name in (("gemm_nt8p", "gemm_nt8p (phase-interleaved, persistent)"), ("gemm_nt256d", "gemm_nt256d (deep ring, one tile per block)"),
                      ("gemm_nt256", "gemm_nt256 (two-slot, persistent)"), ("gemm_tn_dma", "gemm_tn_dma (wgrad)"), ("gemm_tn_x3", "gemm_tn_x3 (split-f32 wgrad)"), ("attn_fwd", "attn_fwd_mfma"),
                      ("attn_bwd", "attn_bwd_mfma"), ("ln_bwd", "ln_bwd"), ("ln_fwd", "ln_fwd"), ("bertadam", "bertadam")):
        if pat in k:
            return name
    return None


disp = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[r["Dispatch_Id"]]["k"] = r["Kernel_Name"]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    f = family(d["k"])
    if f and d.get("SQ_WAVE_CYCLES", 0) > 0:
        acc[f]["n"] += 1
        for c in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU"):
            acc[f][c] += d.get(c, 0.0)
out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "kernels": {}}
for f, a in acc.items():
    w = a["SQ_WAVE_CYCLES"]
    out["kernels"][f] = {"launches_sampled": int(a["n"]), "sq_active_inst_any": round(a["SQ_ACTIVE_INST_ANY"] / w, 4), "sq_active_inst_valu": round(a["SQ_ACTIVE_INST_VALU"] / w, 4),
                         "sq_wait_any": round(a["SQ_WAIT_ANY"] / w, 4), "sq_wait_inst_any": round(a["SQ_WAIT_INST_ANY"] / w, 4), "sq_wait_inst_lds": round(a["SQ_WAIT_INST_LDS"] / w, 4)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["kernels"]))
