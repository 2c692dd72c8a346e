"""MFMA-pipe utilisation per kernel family from a rocprofv3 `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` pass.
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs (16 cycles per v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is summed
over the 8 XCDs, so utilisation = busy / (active / 8 * 1024). usage: mfma_util.py <counter_collection.csv> <out.json>"""
import collections, csv, json, sys


def family(k):
    if "gemm_nt" in k:      # (round 6: the split-f32 forward launches apart from the bf16 launches - under bf16x3_fwd they are different kernel families)
        return "gemm_nt_x3" if ("sf32" in k or ("<float" in k and ", true," in k)) else "gemm_nt_bf16"
    if "gemm_tn" in k:
        return "gemm_tn"
    if "attn_fwd" in k:
        return "attn_fwd"
    if "attn_bwd" in k:
        return "attn_bwd"
    return None


disp = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    disp[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
    disp[r["Dispatch_Id"]]["k"] = r["Kernel_Name"]
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for d in disp.values():
    f = family(d["k"])
    if f and "GRBM_GUI_ACTIVE" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        fam[f][0] += 1; fam[f][1] += d["SQ_VALU_MFMA_BUSY_CYCLES"]; fam[f][2] += d["GRBM_GUI_ACTIVE"]
out = {"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, RGQA_WGRAD_SERIAL=1 python3 bench.py --steps 5 --warmup 2 (B=256, T=20, packed rows); "
               "utilisation = busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "families": {}}
for f, (n, busy, act) in sorted(fam.items()):
    out["families"][f] = {"launches_sampled": n, "mfma_busy_cycles": busy, "gui_active_cycles_sum_xcd": act, "mfma_util": round(busy / (act / 8.0 * 1024.0), 4),
                          "bf16_flop_from_busy_cycles": busy / 16.0 * 16384.0}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out["families"]))
