"""host-side profile of the drop-in trainer step of bench.py (cProfile over the un-synchronised loop): where the Python time of a step goes.
usage: RGQA_PRECISION=bf16x3_fwd python tools/dropin_hostprof.py [steps]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
prec = os.environ.get("RGQA_PRECISION", "bf16x3_fwd")


def time_steps(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(steps):
        fn()
    pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.3f ms/step (under cProfile); wall incl. GPU drain %.3f ms/step" % ((t1 - t0) / steps * 1e3, (t2 - t0) / steps * 1e3))
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(28)
    return (t2 - t0) / steps * 1e3


bench.time_steps = time_steps
bench.dropin_step_leg(256, 20, n, prec)
