"""Per-launch table from an RGQA_PROF_DUMP file (bench.py's profiled steps): GEMM launches grouped by shape tag, with time per step
and TFLOP/s, slowest classes first.  usage: tools/launch_table.py dump.txt steps [x3]"""
import sys
from collections import defaultdict
steps = int(sys.argv[2]); mult = 3.0 if len(sys.argv) > 3 else 1.0
agg = defaultdict(lambda: [0, 0.0, 0.0])
for ln in open(sys.argv[1]):
    p = ln.split()
    if len(p) < 6: continue
    cat, blk, fl, by, ms, tag = int(p[0]), int(p[1]), float(p[2]), float(p[3]), float(p[4]), p[5]
    if cat > 1 and cat != 6: continue      # 0 = NT forward, 1 = TN (wgrad), 6 = NT dgrad (one category with 0 until round 5)
    a = agg[(cat, tag)]; a[0] += 1; a[1] += ms; a[2] += fl
tot = {0: 0.0, 1: 0.0, 6: 0.0}
for (cat, tag), a in agg.items(): tot[cat] += a[1]
print("NT forward %.3f ms/step, NT dgrad %.3f ms/step, TN %.3f ms/step" % (tot[0] / steps, tot[6] / steps, tot[1] / steps))
print("%-3s %-34s %6s %9s %9s %8s" % ("cat", "tag", "n/step", "us/launch", "ms/step", "TF/s"))
for (cat, tag), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-3s %-34s %6.1f %9.1f %9.3f %8.0f" % ({0: "NT", 1: "TN", 6: "ND"}[cat], tag, a[0] / steps, a[1] / a[0] * 1e3, a[1] / steps, a[2] * mult / (a[1] * 1e-3) / 1e12 if a[1] > 0 else 0))
