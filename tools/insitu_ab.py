"""Per-launch in-situ comparison of two RGQA_PROF_DUMP files (same launch sequence): groups NT launches by (GFLOP, MB)."""
import collections, sys
def grp(path, cat):
    g = collections.OrderedDict()
    for c, blk, f, by, ms in (l.split() for l in open(path)):
        if c != cat: continue
        k = (round(float(f) / 1e9, 1), round(float(by) / 1e6))
        e = g.setdefault(k, [0, 0.0]); e[0] += 1; e[1] += float(ms)
    return g
a, b = sys.argv[1], sys.argv[2]
for cat, name in (("0", "NT"), ("1", "TN")):
    ga, gb = grp(a, cat), grp(b, cat)
    ta, tb = sum(v[1] for v in ga.values()), sum(v[1] for v in gb.values())
    print("%s total: A %.3f ms  B %.3f ms  (%+.1f%%)" % (name, ta, tb, 100 * (tb / ta - 1)))
    for k in ga:
        if k in gb and ga[k][0] == gb[k][0]:
            n = ga[k][0]
            print("  %7.1f GF %5d MB x%-3d A %7.1f us  B %7.1f us  %+6.1f%%" % (k[0], k[1], n, ga[k][1] / n * 1e3, gb[k][1] / n * 1e3, 100 * (gb[k][1] / ga[k][1] - 1)))
