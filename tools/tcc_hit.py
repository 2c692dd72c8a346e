"""L2 (TCC) hit rate per kernel family from a rocprofv3 `--pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum` pass of bench.py
(MI355X_MICROARCH.md, L2 section: hit rate = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)).  GEMM launches are also listed by grid size,
which separates the paired / shared launches (persistent, 256 blocks) from the language-only ones.
usage: tcc_hit.py <counter_collection.csv> <out.json> [note]"""
import collections, csv, json, sys


def family(k):
    for f in ("gemm_nt256d", "gemm_nt256", "gemm_nt", "gemm_tn_dma", "gemm_tn_x3", "gemm_tn", "attn_fwd", "attn_bwd", "ln_fwd", "ln_bwd", "bertadam"):
        if f in k:
            return f
    return None


disp = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    d = disp[r["Dispatch_Id"]]
    d[r["Counter_Name"]] = float(r["Counter_Value"])
    d["k"] = r["Kernel_Name"]
    d["grid"] = r.get("Grid_Size", r.get("Grid_Size_X", "0"))
fam = collections.defaultdict(lambda: collections.defaultdict(float))
for d in disp.values():
    f = family(d["k"])
    if not f:
        continue
    keys = [f]
    if f.startswith("gemm_nt256"):
        keys.append("%s grid=%s" % (f, d["grid"]))
    for k in keys:
        a = fam[k]
        a["launches"] += 1
        for c in ("TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_REQ_sum"):
            a[c] += d.get(c, 0.0)
out = {"note": sys.argv[3] if len(sys.argv) > 3 else "", "families": {}}
for f, a in sorted(fam.items()):
    h, m = a["TCC_HIT_sum"], a["TCC_MISS_sum"]
    if h + m == 0:
        continue
    n = a["launches"]
    out["families"][f] = {"launches_sampled": int(n), "hit_rate": round(h / (h + m), 4), "hits_per_launch": round(h / n), "misses_per_launch": round(m / n),
                          "ea_rdreq_per_launch": round(a["TCC_EA0_RDREQ_sum"] / n), "req_per_launch": round(a["TCC_REQ_sum"] / n)}
json.dump(out, open(sys.argv[2], "w"), indent=1)
for f, v in out["families"].items():
    print(f, v)
