"""LAB harness for tools/lab/ffn_chain.hip (VERDICT r4 #1a): the FFN1 -> FFN2 pair of one encoder stage as ONE chained persistent launch against
the two product launches, on chain-cold operands - the activation rows written by the kernel before (a copy), the weights of a set that was last
touched 12 pairs ago (12 x 19 MB of weights + 76-MB outputs in between: cold in the L2s, partly in the Infinity Cache, as in the train step).

    tools/lab/build_lab.sh && python tools/ffn_chain_lab.py [rounds]

Prints, per shape: bit-identity of the outputs, the hand-off's timeout word, and the time of the pair both ways - between HIP events (each
bracket adds ~5 us) and in a stream of 24 back-to-back pairs (copy + pair, minus the copies alone).  Kill criterion of the brief: the chained
paired-layer launch must be >= 8 % faster than the two launches, else record and stop."""
import ctypes as C
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rgqa_amd import _lib          # noqa: E402

prod = _lib.load()
lab = C.CDLL(os.path.join(ROOT, "tools", "lab", "libffn_chain.so"))
VP = C.c_void_p
PP = C.POINTER(VP)
for fn in (lab.lab_ffn_two_launches, lab.lab_ffn_chain):
    fn.restype = C.c_int


def arr(ts):
    return (VP * len(ts))(*[VP(t.data_ptr()) if t is not None else None for t in ts])


def run(tag, rows, epi1, epi2, mt1, mt2, rounds, nsets=12, drop_p=0.1):
    H, I = 768, 3072
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(7)
    R = sum(rows)
    np_ = len(rows)
    off = [sum(rows[:i]) for i in range(np_)]
    Xm = (torch.randn(R, H, device=dev, generator=g)).bfloat16()
    sets = []
    for s in range(nsets):
        sets.append(dict(X=torch.empty_like(Xm),
                         W1=[(torch.randn(I, H, device=dev, generator=g) * 0.03).bfloat16() for _ in rows],
                         W2=[(torch.randn(H, I, device=dev, generator=g) * 0.02).bfloat16() for _ in rows],
                         b1=[torch.randn(I, device=dev, generator=g) * 0.1 for _ in rows], b2=[torch.randn(H, device=dev, generator=g) * 0.1 for _ in rows]))
    aux1 = (torch.randn(R, I, device=dev, generator=g)).bfloat16() if epi1 == 4 else None        # gelu' for the dgrad pair
    outs = {v: dict(Hh=torch.zeros(R, I, dtype=torch.bfloat16, device=dev), Hp=torch.zeros(R, I, dtype=torch.bfloat16, device=dev),
                    Z=torch.zeros(R, H, dtype=torch.bfloat16, device=dev)) for v in ("two", "chain")}
    counters = torch.zeros(256, dtype=torch.int32, device=dev)
    rows_c = (C.c_int * np_)(*rows)
    st = VP(torch.cuda.current_stream().cuda_stream)

    def call(variant, s):
        o, S = outs[variant], sets[s]
        X = [S["X"][off[i]:off[i] + rows[i]] for i in range(np_)]
        Hh = [o["Hh"][off[i]:off[i] + rows[i]] for i in range(np_)]
        Hp = [o["Hp"][off[i]:off[i] + rows[i]] for i in range(np_)] if epi1 == 1 else None
        Z = [o["Z"][off[i]:off[i] + rows[i]] for i in range(np_)]
        a1 = [aux1[off[i]:off[i] + rows[i]] for i in range(np_)] if aux1 is not None else None
        a2 = X                                                                                     # residual (forward) / residual-path gradient (backward)
        args = [np_, rows_c, arr(X), arr(S["W1"]), arr(S["b1"]) if epi1 == 1 else None, arr(Hh), arr(Hp) if Hp else None, arr(S["W2"]),
                arr(S["b2"]) if epi2 == 3 else None, arr(Z), H, I, epi1, epi2, arr(a1) if a1 else None, arr(a2), C.c_float(drop_p if epi2 == 3 else 0.0)]
        if variant == "two":
            rc = lab.lab_ffn_two_launches(*args, st)
        else:
            rc = lab.lab_ffn_chain(*args, mt1, mt2, VP(counters.data_ptr()), st)
        if rc:
            raise RuntimeError("lab: %s" % prod.rgqa_last_error_string().decode())

    # ---- correctness: same bits, no timeout, every counter complete
    sets[0]["X"].copy_(Xm)
    call("two", 0)
    call("chain", 0)
    torch.cuda.synchronize()
    same = all(torch.equal(outs["two"][k], outs["chain"][k]) for k in ("Hh", "Z")) and (epi1 != 1 or torch.equal(outs["two"]["Hp"], outs["chain"]["Hp"]))
    cnt = counters.cpu()
    nb = sum(-(-r // (32 * mt1)) for r in rows)
    print("%s: outputs bit-identical %s; timeout word %d; row-block counters %s (want %d everywhere)" % (
        tag, same, int(cnt[255]), "complete" if bool((cnt[:nb] == I // 256).all()) else "INCOMPLETE " + str(cnt[:nb].tolist()), I // 256))
    ref = (torch.nn.functional.gelu(sets[0]["X"][:64].float() @ sets[0]["W1"][0].float().t() + sets[0]["b1"][0])) if epi1 == 1 else None
    if ref is not None:
        print("   spot check vs torch (64 rows of H): max |diff| %.3e" % float((outs["chain"]["Hh"][:64].float() - ref).abs().max()))
    if not same or int(cnt[255]):
        return None

    # ---- timing 1: one event pair around each pair of launches / each chained launch, interleaved
    ev = {v: [] for v in ("two", "chain")}
    it = 0
    for r in range(rounds):
        for v in ("two", "chain"):
            for _ in range(nsets):
                s = it % nsets
                it += 1
                sets[s]["X"].copy_(Xm)                    # "written by the kernel before"
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                call(v, s)
                z.record()
                ev[v].append((a, z))
    torch.cuda.synchronize()
    t_ev = {v: statistics.median(a.elapsed_time(z) * 1e3 for a, z in ev[v][nsets:]) for v in ev}
    # ---- timing 2: streams of 24 [copy, pair] back to back, minus the copies alone
    def stream(v, n=24):
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for i in range(n):
            sets[i % nsets]["X"].copy_(Xm)
            if v is not None:
                call(v, i % nsets)
        z.record()
        torch.cuda.synchronize()
        return a.elapsed_time(z) * 1e3 / n
    t_st = {}
    for v in ("two", "chain"):
        xs = []
        for r in range(rounds):
            base = stream(None)
            xs.append(stream(v) - base)
        t_st[v] = statistics.median(xs)
    fl = 2.0 * R * H * I * 2
    print("   %-34s  two launches %7.1f us   chained %7.1f us   (%+.1f %%)   [%.0f -> %.0f TFLOP/s]" % (
        "between HIP events (median):", t_ev["two"], t_ev["chain"], (t_ev["chain"] / t_ev["two"] - 1) * 100, fl / t_ev["two"] / 1e6, fl / t_ev["chain"] / 1e6))
    print("   %-34s  two launches %7.1f us   chained %7.1f us   (%+.1f %%)   [%.0f -> %.0f TFLOP/s]" % (
        "in a stream, copies subtracted:", t_st["two"], t_st["chain"], (t_st["chain"] / t_st["two"] - 1) * 100, fl / t_st["two"] / 1e6, fl / t_st["chain"] / 1e6))
    return t_st


if __name__ == "__main__":
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    print("FFN1 -> FFN2 as one chained persistent launch vs two product launches (bf16, H=768, I=3072; chain-cold operands)")
    run("paired layer forward (3140 + 9216 rows), GELU -> dropout+residual, tiles 224 / 160 rows", [3140, 9216], 1, 3, 7, 5, rounds)
    run("paired layer forward, tiles 192 / 160 rows", [3140, 9216], 1, 3, 6, 5, rounds)
    run("paired layer forward, tiles 224 / 128 rows", [3140, 9216], 1, 3, 7, 4, rounds)
    run("paired layer backward (dFFN2 -> dFFN1), .gelu' -> +residual gradient, tiles 224 / 160 rows", [3140, 9216], 4, 5, 7, 5, rounds)
    run("language-only layer forward (3140 rows), tiles 224 / 160 rows", [3140], 1, 3, 7, 5, rounds)
