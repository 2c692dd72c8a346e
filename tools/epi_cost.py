import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import gemm_bench as g
for M in (12288,):
    for epi in (0, 1):
        g.nt(M, 3072, 768, epi)
    g.nt(M, 768, 3072, 0)
    g.nt(M, 768, 768, 0)
    g.nt(M, 2304, 768, 0)
