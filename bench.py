"""Headline benchmark: QA-pairs/sec of one LXMERT-GQA train step (BASELINE.json), B=256 per GPU, T=20, O=36,
synthetic inputs already resident in HBM.

    python bench.py                                          (1 GPU, 100 timed steps)
    python bench.py --gpus N --steps K --warmup W            (N > 1 without WORLD_SIZE: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward (train mode, dropout 0.1) + BCE x NA loss + backward + [RCCL gradient exchange] +
clip_grad_norm_(5.) + BertAdam + re-cast of the weight operand copies: everything tasks/gqa_conf.py:174-202 does per batch.
Rank 0 prints ONE JSON line.  The headline (`value`, `dtype` "bf16x3_fwd") is the mode whose logits stay inside the north star's
1e-3 bound - the forward pass on split-f32 operands (three bf16 MFMA products per f32 product), the backward pass in bf16 as
BASELINE config 3 prescribes - with its logits error re-measured in-run against the CPU oracle (`parity_in_run`).  The same line
carries `config3_bf16` (bf16 forward too: faster, logits OUTSIDE the bound, its error next to it), `tolerance_compliant` (bf16x3
throughout), `seq30` (T = 30, what tasks/gqa_model.py:11 pads to), `forward_only_b256` (BASELINE config 2), `dropin_step` (the
reference trainer's own statements through GQAModel / BertAdam on fresh batches from the device batcher), BASELINE configs 4 and 5
(`other_workloads`, each with a leg inside the bound) and the roofline of the headline's dominant kernel - the split-f32 NT GEMM
launches of the forward pass - against the datasheet peak, counting one product per f32 product (and, beside it, the 3x figure of
MFMA work actually issued).

Multi-GPU runs are supervised: the process that the launcher (or the user) started never touches the GPU; it runs the ranks as
child processes and, when the default gradient exchange ("sharded": all-to-all + all-gather) fails or hangs, starts a FRESH set
once with RGQA_DP_MODE=allreduce and reports `dp_fallback` in the JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_BWD_GFLOP = {20: 30.3388, 30: 37.0403}   # per QA pair, SURVEY.md §8 D3
FWD_GFLOP = {20: 10.5827, 30: 12.8330}
PEAK_BF16_TFLOPS = 2500.0                    # dense MFMA bf16, MI355X_MICROARCH.md (256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz)
PMC_PROFILES = {"bf16x3_fwd": "r06_pmc_gemm_nt_x3fwd.json", "bf16": "r06_pmc_gemm_nt_bf16.json"}      # tools/pmc_summary.py, per headline precision
FULL = dict(vocab_size=30522, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=9, x_layers=5,
            r_layers=5, feat_dim=2048, pos_dim=4, num_answers=1842)


def warmup_linear(x, warmup):
    return x / warmup if x < warmup else max((x - 1.0) / (warmup - 1.0), 0.0)


def init_params(e, seed):
    """random-init weights of the reference architecture (init_bert_weights: N(0, 0.02), LN = 1/0, bias = 0)."""
    g = torch.Generator(device=e.device).manual_seed(seed)
    e.params.normal_(0.0, 0.02, generator=g)
    # LayerNorm weights 1, every other vector 0: ONE index build on the host and two scatter launches (until round 6 one fill launch per 1-D tensor: the ~290 fill
    # kernels of a run's first seconds showed up in the kernel-trace summaries as "42 fills per step")
    ones, zeros = [], []
    for sp in e.specs:
        if len(sp.shape) == 1:
            (ones if ("LayerNorm.weight" in sp.name or "layer_norm.weight" in sp.name or sp.name == "logit_fc.2.weight") else zeros).append(np.arange(sp.offset, sp.offset + sp.numel, dtype=np.int64))
    for idx, val in ((ones, 1.0), (zeros, 0.0)):
        if idx:
            e.params[torch.from_numpy(np.concatenate(idx)).to(e.device)] = val


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


_T0 = time.time()


def note(msg):
    """progress on stderr (the one JSON line owns stdout): which leg of the run is on, with the seconds since start"""
    sys.stderr.write("bench.py [%6.1f s] %s\n" % (time.time() - _T0, msg))
    sys.stderr.flush()


def cpu_budget():
    """(CPUs in this process's affinity mask, CPUs the cgroup's quota allows or None, CPUs of the machine): a container may see every CPU of the
    host in its mask and still be throttled to a few of them - threads beyond the quota only contend"""
    try:
        aff = len(os.sched_getaffinity(0))
    except Exception:
        aff = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2: "max 100000" or "<quota> <period>"
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    return aff, quota, os.cpu_count() or aff


def cpu_baseline(T, sample_b, iters, warm=2, extra=True, check=None):
    """The oracle (CPU restatement pinned to the reference's golden vectors) timed on this box's host cores: `warm` untimed +
    `iters` timed full train steps at B=sample_b (the headline entry), plus SURVEY §8 D4's other cases in `cases`: BASELINE config 1
    (B=4 train step), the B=64 and B=256 train steps and the B=256 eval forward.
    check(P, batch, logits): called with the oracle's weights, the B=256 eval batch and the oracle's logits on it - the one place
    where the oracle is the CHECKER of the GPU engines' logits (the `tolerance_compliant` figures); it is never what is measured there."""
    from oracle import lxmert_ref as R
    from rgqa_amd import synth
    cfg = R.RefConfig(**FULL)
    # every CPU this process may USE (the host's cores, stated): the affinity mask, cut to the cgroup's CPU quota when there is one;
    # the JSON carries mask, quota and the machine's CPU count next to the thread count (rounds 1-4 capped the threads at 16)
    ncpu, quota, host = cpu_budget()
    use = ncpu if quota is None else max(1, min(ncpu, int(quota + 0.5)))
    torch.set_num_threads(max(1, int(os.environ.get("RGQA_BENCH_CPU_THREADS", use))))
    note("cpu_baseline: %d threads (affinity mask %d CPUs, cgroup quota %s, machine %d)" % (torch.get_num_threads(), ncpu, "none" if quota is None else "%.1f" % quota, host))
    # the deterministic filler of the golden fixtures (biases / LayerNorm parameters non-trivial, |logit| ~ 1)
    P = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.fill_state_dict(R.param_shapes(cfg)).items()}
    opt = R.BertAdamRef(list(P.values()), lr=1e-5, warmup=0.1, t_total=1000)

    def timed(fn, n_warm, n):
        for _ in range(n_warm):
            fn()
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))

    def train_case(bsz, n_warm, n):
        b = synth.synth_batch(bsz, T, seed=99)
        batch = {k: torch.from_numpy(v) for k, v in b.items() if k != "lengths"}
        return timed(lambda: R.train_step(P, cfg, batch, opt), n_warm, n)

    out_cases, checked = {}, None
    if extra:
        # the eval forward first, on the untouched weights: its logits are what the GPU engines are checked against
        b = synth.synth_batch(256, T, seed=98)
        batch = {k: torch.from_numpy(v) for k, v in b.items() if k != "lengths"}
        with torch.no_grad():
            Pd = {k: v.detach() for k, v in P.items()}
            lg = []
            te = timed(lambda: lg.append(R.gqa_forward(Pd, cfg, batch["feats"], batch["boxes"], batch["input_ids"], batch["input_mask"], batch["segment_ids"])[0]), 0, 2)
        out_cases["eval_forward_B256"] = dict(value=round(256 / te, 2), unit="QA-pairs/s", sample="BASELINE config 2: eval forward B=256 T=%d, 2 timed, median %.2fs" % (T, te))
        note("cpu_baseline: eval forward B=256 %.2f s" % te)
        if check is not None:
            checked = check({k: v.detach().clone() for k, v in P.items()}, b, lg[-1].detach())
    slow_host = extra and te > 25.0          # (4 s on the 16 host cores of a GPU box: a host this slow would spend minutes on the larger cases below)
    t = train_case(sample_b, warm if not slow_host else 1, iters if not slow_host else 2)
    note("cpu_baseline: train step B=%d %.2f s" % (sample_b, t))
    out = dict(value=sample_b / t, unit="QA-pairs/s", cores=torch.get_num_threads(), affinity_cpus=ncpu, cgroup_quota_cpus=quota, host_cpus=host, kind="port", cpu_model=_cpu_model(),
               sample="full train step (fwd+BCE+bwd+clip+BertAdam), B=%d T=%d, fp32, %d warm-up + %d timed iters, median %.2fs" % (sample_b, T, warm, iters, t))
    if extra:
        t4 = train_case(4, 1, 3)
        out_cases["train_step_B4"] = dict(value=round(4 / t4, 2), unit="QA-pairs/s", sample="BASELINE config 1: full train step B=4 T=%d, 1 warm-up + 3 timed, median %.2fs" % (T, t4))
        note("cpu_baseline: train step B=4 %.2f s" % t4)
        if slow_host:
            out_cases["skipped"] = "train steps at B=64 and B=256: the B=256 eval forward took %.1f s on this host (bounded sample)" % te
        else:
            t64 = train_case(64, 1, 3)
            out_cases["train_step_B64"] = dict(value=round(64 / t64, 2), unit="QA-pairs/s", sample="full train step B=64 T=%d, 1 warm-up + 3 timed, median %.2fs" % (T, t64))
            note("cpu_baseline: train step B=64 %.2f s" % t64)
            t256 = train_case(256, 0, 2)
            out_cases["train_step_B256"] = dict(value=round(256 / t256, 2), unit="QA-pairs/s", sample="full train step B=256 T=%d (the headline's batch), 2 timed, median %.2fs" % (T, t256))
        out["cases"] = out_cases
    return out, checked


_JSON_FD = None


def emit_json(obj):
    """the one stdout line of a run (see main(): fd 1 itself is routed to stderr while the ranks run)"""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_JSON_FD, line)


# ------------------------------------------------------------------------------------------------ supervision of the rank processes
def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run_children(envs, deadline_s):
    """Starts one child of this same command line per environment (child 0 inherits stdout: rank 0's JSON line), waits for all of them;
    -> (rc, reason).  A failing or overdue child ends the attempt: the others are terminated by PID."""
    procs = []
    for i, env in enumerate(envs):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if i == 0 else sys.stderr))
    deadline = time.time() + deadline_s
    rc, reason = 0, ""
    live = list(procs)
    while live and rc == 0:
        for pr in list(live):
            c = pr.poll()
            if c is not None:
                live.remove(pr)
                if c != 0:
                    rc = c if c > 0 else 1
                    reason = "a rank process exited with code %d" % c
        if time.time() > deadline and rc == 0 and live:
            rc, reason = 124, "the rank processes did not finish within %d s" % int(deadline_s)
        if live and rc == 0:
            time.sleep(0.05)
    for pr in live:             # end exactly the processes started here
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
    return rc, reason


def supervise(n_self_launch):
    """The process the launcher started never touches the GPU (nothing that has initialised HIP is ever forked or exec'd): it runs the
    rank(s) as children and exits with their code.
      n_self_launch = N: `python bench.py --gpus N` without a launcher - N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
                         (what lxrt/entry.py:102-103's nn.DataParallel switch is to the reference);
      n_self_launch = 0: this process IS one rank of a torch.distributed.run launch - one child with this rank's environment and a
                         rendezvous of its own (MASTER_PORT + 1 + attempt; the launcher's agent store is not reused, so a second
                         attempt starts from an empty store).
    If the attempt fails or hangs under the default exchange (RGQA_DP_MODE unset or 'sharded'), ONE fresh attempt runs with
    RGQA_DP_MODE=allreduce and RGQA_BENCH_DP_FALLBACK=<reason>: the JSON line then carries `dp_fallback`."""
    mode = os.environ.get("RGQA_DP_MODE", "sharded")
    total = float(os.environ.get("RGQA_BENCH_LAUNCH_TIMEOUT", "1500"))
    base_port = int(os.environ.get("MASTER_PORT", "0"))

    def envs_for(attempt, dp_mode, reason):
        common = dict(os.environ, RGQA_BENCH_CHILD="1", RGQA_DP_MODE=dp_mode, MASTER_ADDR="127.0.0.1")
        common.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        common.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        if reason:
            common["RGQA_BENCH_DP_FALLBACK"] = reason
        if n_self_launch:
            port = _free_port()
            return [dict(common, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_self_launch), LOCAL_WORLD_SIZE=str(n_self_launch),
                         MASTER_PORT=str(port)) for r in range(n_self_launch)]
        return [dict(common, MASTER_PORT=str(base_port + 1 + attempt))]

    retry = mode.partition("_")[0] in ("sharded", "peer")          # the modes with a fallback: the plain all-reduce, the most trodden RCCL path
    first_deadline = total if not retry else min(total, max(120.0, 0.6 * total))
    t0 = time.time()
    rc, reason = _run_children(envs_for(0, mode, ""), first_deadline)
    if rc != 0 and retry:
        sys.stderr.write("bench.py: %s under RGQA_DP_MODE=%s; starting a fresh set of rank processes with RGQA_DP_MODE=allreduce\n" % (reason, mode))
        rc, reason = _run_children(envs_for(1, "allreduce", "%s exchange failed (%s); re-run with allreduce" % (mode, reason)),
                                   max(120.0, total - (time.time() - t0)))
    if rc != 0:
        sys.stderr.write("bench.py: %s\n" % reason)
    return rc


def dp_selfcheck(dist, mode, device, precision="bf16"):
    """Every collective the chosen exchange needs, once, on 1 MB, checked numerically - so that a broken or hanging collective is a fast
    non-zero exit (the process group carries a short timeout) before any warm-up step, not a silent hang of the timed region."""
    W, r = dist.get_world_size(), dist.get_rank()
    n = 262144
    want = float(sum(range(1, W + 1)))
    t = torch.full((n,), float(r + 1), device=device)
    dist.all_reduce(t)
    ok = bool((t == want).all())
    from rgqa_amd.parallel import exchange_payload
    kind = mode.partition("_")[0]
    pdt = exchange_payload(mode, precision)          # what this exchange puts on the wire (sharded: by the engine's precision; all-reduce: f32 unless _bf16)
    if kind == "allreduce" and pdt == torch.bfloat16:
        tb = torch.full((n,), float(r + 1), device=device, dtype=torch.bfloat16)
        dist.all_reduce(tb)
        ok = ok and bool((tb.float() == want).all())
    if kind == "sharded":
        s = n // W
        send = torch.full((W * s,), float(r + 1), device=device, dtype=pdt)
        recv = torch.zeros(W * s, device=device, dtype=pdt)
        dist.all_to_all_single(recv, send)
        ok = ok and bool((recv.view(W, s).float() == torch.arange(1, W + 1, device=device, dtype=torch.float32)[:, None]).all())
        full = torch.zeros(W * s, device=device, dtype=torch.bfloat16)
        full[r * s:(r + 1) * s] = float(r + 1)
        dist.all_gather_into_tensor(full, full[r * s:(r + 1) * s])          # in place, as ShardedExchange.step issues it
        ok = ok and bool((full.view(W, s).float() == torch.arange(1, W + 1, device=device, dtype=torch.float32)[:, None]).all())
    if device.type == "cuda":
        torch.cuda.synchronize()
    if not ok:
        raise SystemExit("bench.py: collective self-check failed under RGQA_DP_MODE=%s" % mode)


XGMI_LINKS, XGMI_GBS_PER_LINK_DIR = 7, 76.8      # MI355X: 7 Infinity Fabric links per GPU, 153.6 GB/s each bidirectional = 76.8 GB/s per direction (MI355X_MICROARCH.md)


def dp_wire_probe(dist, comm, device, iters=10):
    """OUTSIDE the timed region, before the warm-up steps: the collectives of the gradient exchange on the REAL chunk sizes, timed with device events,
    as achieved GB/s per GPU and direction - so that the first record of a multi-GPU run tells whether RCCL's all-to-all / all-gather drive all seven
    xGMI links of a GPU at once (the assumption behind DESIGN.md §5's prediction; a ring-shaped schedule would show 1/7 of it).  Bytes on the wire per
    GPU and direction: (W - 1) / W of the buffer for all-to-all and all-gather, 2 (W - 1) / W for an all-reduce (its bus bandwidth).  At W = 1 (the
    one-rank rehearsal) nothing is on a wire: the figures are then the library's local copy rate, reported as such."""
    W = dist.get_world_size()
    out = {"world": W, "links_per_gpu": XGMI_LINKS, "link_peak_gbs_per_direction": XGMI_GBS_PER_LINK_DIR,
           "all_links_peak_gbs_per_direction": XGMI_LINKS * XGMI_GBS_PER_LINK_DIR}
    if W == 1:
        out["note"] = "one rank: nothing on a wire - local copy rates of the library, not link rates"

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        t = torch.tensor([a.elapsed_time(b) / iters], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) * 1e-3

    wire = (W - 1) / W if W > 1 else 1.0
    peak = XGMI_LINKS * XGMI_GBS_PER_LINK_DIR
    try:
        if hasattr(comm, "chunks"):          # sharded exchange: all-to-all of the largest chunk in the payload type, all-gather of the same chunk of weights
            s_ = comm.smax
            send, recv = comm._send[:W * s_], comm._recv[:W * s_]
            nbytes = W * s_ * send.element_size()
            t = timed(lambda: dist.all_to_all_single(recv, send))
            out["all_to_all"] = dict(buffer_mb=round(nbytes / 2**20, 1), dtype=str(send.dtype).replace("torch.", ""), ms=round(t * 1e3, 3), gbs_per_gpu_per_direction=round(nbytes * wire / t / 1e9, 1),
                                     frac_of_all_links=round(nbytes * wire / t / 1e9 / peak, 3))
            part = recv[:s_].clone()
            t = timed(lambda: dist.all_gather_into_tensor(recv, part))
            out["all_gather"] = dict(buffer_mb=round(nbytes / 2**20, 1), dtype=str(recv.dtype).replace("torch.", ""), ms=round(t * 1e3, 3), gbs_per_gpu_per_direction=round(nbytes * wire / t / 1e9, 1),
                                     frac_of_all_links=round(nbytes * wire / t / 1e9 / peak, 3))
        buf = torch.zeros(64 * (1 << 20) // 4, device=device)       # the all-reduce mode's bucket: 64 MB f32
        t = timed(lambda: dist.all_reduce(buf))
        bus = buf.numel() * 4 * (2 * wire if W > 1 else 1.0) / t / 1e9
        out["all_reduce_f32_64mb"] = dict(ms=round(t * 1e3, 3), bus_gbs_per_gpu=round(bus, 1), frac_of_all_links=round(bus / peak, 3))
        if W > 1 and "all_to_all" in out:
            out["all_links_in_use"] = bool(out["all_to_all"]["frac_of_all_links"] >= 0.35)      # a one-link-at-a-time schedule cannot exceed 1/7 = 0.14
    except Exception as ex:          # a diagnostic: never lose the run over it
        out["error"] = repr(ex)
    return out


def launch_check(world, rank, fail_rank, fail_mode):
    """rehearsal of the launch path without a GPU (tests/test_host.py): rendezvous over gloo, the exchange's self-check, one JSON line"""
    import torch.distributed as dist
    mode = os.environ.get("RGQA_DP_MODE", "sharded")
    if rank == fail_rank and (fail_mode == "" or fail_mode == mode):
        raise SystemExit(3)
    if world > 1:
        import datetime
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=60))
    t = torch.ones(1)
    if world > 1:
        dp_selfcheck(dist, "allreduce" if mode.startswith("sharded") else mode, torch.device("cpu"), "f32")      # (gloo has no all_to_all on CPU tensors in every build)
        dist.all_reduce(t)
        seen = dist.get_world_size()
    else:
        seen = 1
    if rank == 0:
        emit_json({"launch_check": True, "n_gpus": world, "n_ranks_seen": seen, "sum": float(t.item()), "dp_mode": mode,
                   "dp_fallback": os.environ.get("RGQA_BENCH_DP_FALLBACK")})
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ measurement helpers (rank 0, one GPU)
def time_steps(fn, n, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def probe_sustained_clock(e):
    """In-kernel shader clock under the dense bf16 GEMM loop: >= 1 s of back-to-back launches of the stamped instantiation of the
    persistent NT kernel on random operands (the paired QKV projection's shape), median over the blocks of the last launch."""
    import ctypes as C
    from rgqa_amd._lib import check, ptr
    M, N, K = 12356, 2304, 768
    A = torch.randn(M, K, device=e.device).bfloat16()
    W = (torch.randn(N, K, device=e.device) * 0.05).bfloat16()
    Cc = torch.empty(M, N, device=e.device, dtype=torch.bfloat16)
    st = torch.zeros(256 * 8, dtype=torch.int64, device=e.device)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t0 = time.perf_counter()
    launches = 0
    while time.perf_counter() - t0 < 1.2:
        check(e.lib.rgqa_probe_gemm(ptr(A), ptr(W), ptr(Cc), None, M, N, K, 8, 0, 200, ptr(st), s))
        torch.cuda.synchronize()
        launches += 200
    v = st.view(-1, 8).cpu().numpy().astype(np.float64)
    v = v[(v[:, 3] > v[:, 1])]
    mhz = (v[:, 2] - v[:, 0]) / (v[:, 3] - v[:, 1]) * 100.0
    return float(np.median(mhz)), launches


def engine_step_fn(e, dev, lengths, rank=0, t_total=10000):
    state = dict(i=0)

    def step():
        i = state["i"]
        e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=4321 + rank + 1000003 * i, lengths=lengths)
        e.loss_backward(dev["target"])
        e.adam_step(1e-5 * warmup_linear(i / t_total, 0.1), max_norm=5.0)
        state["i"] = i + 1
    return step


def dropin_step_leg(B, T, n_steps, precision):
    """The reference trainer's own statements (tasks/gqa_conf.py:148-202) through the drop-in surface at B=256: a fresh batch from the
    device batcher (feature store on disk -> pinned gather -> H2D -> device preparation), `model(feats, boxes, list_of_str)` (native
    tokenizer), BCE x NA, `loss.backward()`, `nn.utils.clip_grad_norm_`, `BertAdam.step()` - everything the headline leaves out."""
    import tempfile
    import types
    from rgqa_amd.data import FeatureStore, DeviceBatcher
    tmp = tempfile.mkdtemp(prefix="rgqa_bench_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        # vocabulary file in bert-base-uncased's layout (specials at their ids), synthetic words; questions built from it
        words = ["what", "is", "the", "color", "of", "dog", "man", "woman", "left", "right", "to", "on", "in", "front", "behind", "table", "holding", "bottle",
                 "red", "blue", "green", "who", "wearing", "shirt", "are", "there", "any", "cars", "photo", "side", "which", "kind", "animal", "standing", "near", "tree"]
        vocab = ["[unused%d]" % i for i in range(30522)]
        vocab[0], vocab[100], vocab[101], vocab[102], vocab[103] = "[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]"
        vocab[1029] = "?"
        for i, w in enumerate(words):
            vocab[2000 + i] = w
        vpath = os.path.join(tmp, "vocab.txt")
        with open(vpath, "w") as f:
            f.write("\n".join(vocab) + "\n")
        os.environ["RGQA_BERT_VOCAB"] = vpath
        os.environ["RGQA_PRECISION"] = precision
        N, O, F, NA = 2 * B, 36, 2048, 1842
        rng = np.random.RandomState(5)
        prefix = os.path.join(tmp, "store")
        np.maximum(rng.standard_normal((N, O, F)), 0).astype(np.float16).tofile(prefix + ".feats.bin")
        xy = np.sort(rng.uniform(0, 400, (N, O, 2, 2)), axis=3)                 # [.., axis (x|y), (lo, hi)]
        np.stack([xy[:, :, 0, 0], xy[:, :, 1, 0], xy[:, :, 0, 1], xy[:, :, 1, 1]], -1).astype(np.float32).tofile(prefix + ".boxes.bin")
        json.dump({"img_ids": ["i%d" % i for i in range(N)], "img_h": [480] * N, "img_w": [640] * N, "O": O, "F": F, "dtype": "f16"}, open(prefix + ".meta.json", "w"))
        answers = ["a%d" % i for i in range(NA)]
        ans2label = {a: i for i, a in enumerate(answers)}
        data = []
        for i in range(4 * B):
            L = 3 + int(rng.randint(0, 16))
            sent = " ".join(words[int(rng.randint(0, len(words)))] for _ in range(L)) + " ?"
            label = {} if rng.uniform() < 0.25 else {answers[int(rng.randint(0, NA))]: 1.0}
            data.append({"img_id": "i%d" % int(rng.randint(0, N)), "question_id": i, "sent": sent, "label": label})
        from rgqa_amd.tasks.gqa_model import GQAModel
        from rgqa_amd.lxrt.optimization import BertAdam
        model = GQAModel(NA, max_seq_length=T, model_args=types.SimpleNamespace(llayers=9, xlayers=5, rlayers=5, from_scratch=True)).cuda()
        model.train()
        optim = BertAdam(list(model.parameters()), lr=1e-5, warmup=0.1, t_total=10000)
        bce = torch.nn.BCEWithLogitsLoss()
        batcher = DeviceBatcher(FeatureStore(prefix), ans2label, NA, B)
        state = dict(i=0)

        reuse = os.environ.get("RGQA_DROPIN_REUSE_BATCH") == "1"          # tools/dropin_profile.py only: what the fresh batch costs the step
        held = batcher.batch([data[k % len(data)] for k in range(B)]) if reuse else None

        def step():
            i = state["i"]
            if reuse:
                ques_id, feats, boxes, sent, target = held
            else:
                batch = [data[(i * B + k) % len(data)] for k in range(B)]
                ques_id, feats, boxes, sent, target = batcher.batch(batch)
            optim.zero_grad()
            logit = model(feats, boxes, sent)
            loss = bce(logit, target) * logit.size(1)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 5.)
            optim.step()
            state["i"] = i + 1

        ms = time_steps(step, n_steps, 8)
        del model, optim, batcher
        torch.cuda.empty_cache()
        return ms
    finally:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)


def mixup_leg(e, dev, lengths_1x, B, T, O, n_steps, rank=0):
    """BASELINE config 4 on one GPU: the RoI-mixup finetune step (tasks/gqa_mixup_vis.py:134-181, 250-259) - every loader batch of B QA pairs is
    doubled on the device (mixup_v1, Beta(1, 5): run/gqa_mixup_vis_finetune.bash), the model sees 2B rows.  Reuses the headline's engine, re-bound
    to 2B rows.  -> ms per step (n_steps timed after 3 warm-up steps)."""
    import random as _random
    from rgqa_amd.mixup import RoIMixup
    _random.seed(777 + rank)
    np.random.seed(777 + rank)
    mixer = RoIMixup("mixup_v1", alpha=1.0, beta=5.0)
    img_ids = list(range(B))                                  # every synthetic sample is its own image
    ids2 = torch.cat([dev["input_ids"], dev["input_ids"]], 0).contiguous()       # sent = sent + sent (gqa_mixup_vis.py:181)
    mask2 = torch.cat([dev["input_mask"], dev["input_mask"]], 0).contiguous()
    seg2 = torch.cat([dev["segment_ids"], dev["segment_ids"]], 0).contiguous()
    lens2 = None if lengths_1x is None else np.ascontiguousarray(np.tile(lengths_1x, 2), dtype=np.int32)
    e.ensure_shape(2 * B, T, O)
    e.sync_weights()
    e.enable_segment_sumsq(True)
    state = dict(i=0)

    def step():
        i = state["i"]
        f2, b2, t2 = mixer(dev["feats"], dev["boxes"], dev["target"], img_ids)       # host draws + ONE device gather + target scaling
        e.forward(f2, b2, ids2, mask2, seg2, train=True, seed=991 + rank + 1000003 * i, lengths=lens2)
        e.loss_backward(t2)
        e.adam_step(1e-5 * warmup_linear(i / 10000, 0.1), max_norm=5.0)
        state["i"] = i + 1
    return time_steps(step, n_steps, 3)


def butd_tokens(B, L=40, ntoken=3000, seed=1234):
    """BUTD question tokens as GQABUTD.tokenize builds them (butd/butd.py:180-193): dictionary indices of the words, FRONT-padded to 40 with the
    padding index ntoken (whose embedding row is zero and receives no gradient, butd.py:36); question length ~ U{5..40}."""
    rng = np.random.RandomState(seed)
    toks = np.full((B, L), ntoken, dtype=np.int64)
    for r in range(B):
        n = int(rng.randint(5, L + 1))
        toks[r, L - n:] = rng.randint(0, ntoken, size=n)
    return toks


def butd_leg(B, n_steps, precision="bf16", rank=0):
    """BASELINE config 5 on one GPU: the BUTD backbone's train step (butd/butd.py:195-221), B QA pairs, 40 tokens, dictionary of 3000 words.
    -> ms per step (n_steps timed after 3 warm-up steps)."""
    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    T, O = 40, 36
    eb = Engine(arch=1, vocab_size=3001, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=1842, precision=precision,
                hidden_dropout=0.5, attn_dropout=0.2, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0).allocate("cuda")
    g = torch.Generator(device=eb.device).manual_seed(0)
    eb.params.uniform_(-0.03, 0.03, generator=g)
    for sp in eb.specs:
        if sp.name.endswith("weight_g"):
            eb.view(eb.params, sp).fill_(1.0)
    b = synth.synth_batch(B, T, seed=1234 + rank, vocab=3000)
    b["input_ids"] = butd_tokens(B, T, 3000, seed=1234 + rank)
    d = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
    eb.ensure_shape(B, T, O)
    eb.sync_weights()
    ms = time_steps(engine_step_fn(eb, d, None, rank), n_steps, 3)
    del eb
    torch.cuda.empty_cache()
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100, help="timed steps (default long enough for the clock to be in its loaded steady state)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="QA pairs per GPU per step")
    ap.add_argument("--seq", type=int, default=20)
    ap.add_argument("--precision", default="bf16x3_fwd", help="bf16x3_fwd (the headline: split-f32 forward pass inside the 1e-3 logits bound, bf16 backward pass as BASELINE config 3 prescribes) | bf16 (bf16 forward too: outside the bound) | bf16x3 (split f32 throughout) | f32 (exact, vector ALU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--padded", action="store_true", help="compute every padded token position (the reference's layout) instead of packing the real tokens")
    ap.add_argument("--butd", action="store_true", help="BASELINE config 5: BUTD backbone (butd/butd.py) train step, B per GPU, 40 tokens, dictionary 3000")
    ap.add_argument("--uniter", action="store_true", help="UNITER backbone (uniter/uniter.py GQAUNITER): 12 BertLayers over [20 text ; 36 region] sequences, bert-base-cased sizes")
    ap.add_argument("--mixup", action="store_true",
                    help="BASELINE config 4: RoI-mixup finetune (gqa_mixup_vis.py:134-181): every loader batch is doubled on the device "
                         "(mixup_v1, Beta(1,5)); the model sees 2x rows per QA pair; value still counts loader QA pairs")
    ap.add_argument("--lean", action="store_true", help="only the warm-up and timed steps (no extra legs, no live kernel timing, no CPU baseline): what runs under rocprofv3")
    ap.add_argument("--no-extra-legs", action="store_true", help="headline + live kernel timing only (no bf16x3 / forward-only / drop-in legs): quick A/B runs")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)          # CPU rehearsal of the N-rank launch path
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--launch-check-fail-mode", default="", help=argparse.SUPPRESS)        # fail only under this RGQA_DP_MODE
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline, args.profile_steps = True, 0

    child = os.environ.get("RGQA_BENCH_CHILD") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(supervise(args.gpus))       # before any torch.cuda / HIP call
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not child:
        raise SystemExit(supervise(0))               # a launcher's rank: the work runs in a child, so a second attempt stays possible
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # stdout carries the ONE JSON line and nothing else: native libraries (gloo / RCCL rendezvous banners, HIP runtime notes) write to
    # file descriptor 1 behind Python's back, so every rank sends fd 1 to stderr now and rank 0 emits its line through the saved descriptor
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.launch_check:
        return launch_check(world, rank, args.launch_check_fail_rank, args.launch_check_fail_mode)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda unavailable); there is no CPU path")
    local %= max(1, torch.cuda.device_count())      # rehearsal of N ranks on a box with fewer GPUs (the driver's node has one per rank)
    torch.cuda.set_device(local)
    dist = None
    dp_mode = os.environ.get("RGQA_DP_MODE", "sharded")
    if world > 1:
        import datetime
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=int(os.environ.get("RGQA_DP_TIMEOUT", "240")))      # a hung collective is a non-zero exit within minutes
        if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        else:       # rehearsal only: RCCL refuses two ranks on one device; gloo stages the same collectives through the host
            dist.init_process_group("gloo", timeout=tmo)
        if dist.get_backend() == "nccl":
            dp_selfcheck(dist, dp_mode, torch.device("cuda", local), args.precision)
    elif os.environ.get("RGQA_BENCH_RCCL_REHEARSAL") == "1":
        # one-GPU rehearsal of the N > 1 path ON RCCL: a process group of ONE rank, so every collective of the chosen exchange runs through
        # the library (group creation with device_id and timeout, the bf16 all-to-all, the in-place all-gather, the scalar all-reduce) and
        # the exchange's local arithmetic (casts, f32 sums of the shards, sharded BertAdam, copy refresh) is timed at full size
        import datetime
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=120))
        dp_selfcheck(dist, dp_mode, torch.device("cuda", local), args.precision)

    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    from rgqa_amd.parallel import make_exchange

    B, T, O = args.batch, args.seq, 36
    if args.butd:
        T = 40
        e = Engine(arch=1, vocab_size=3001, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=1842, precision=args.precision,
                   hidden_dropout=0.5, attn_dropout=0.2, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0).allocate("cuda")
        g = torch.Generator(device=e.device).manual_seed(0)
        e.params.uniform_(-0.03, 0.03, generator=g)
        for sp in e.specs:
            if sp.name.endswith("weight_g"):
                e.view(e.params, sp).fill_(1.0)
    elif args.uniter:
        e = Engine(precision=args.precision, arch=2, vocab_size=28996, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=12, x_layers=0,
                   r_layers=0, feat_dim=2048, pos_dim=7, num_answers=1842).allocate("cuda")
        init_params(e, seed=0)
    else:
        e = Engine(precision=args.precision, **FULL).allocate("cuda")
        init_params(e, seed=0)       # identical replica on every rank
    b = synth.synth_batch(B, T, seed=1234 + rank, vocab=3000 if args.butd else (28996 if args.uniter else 30522))
    if args.butd:
        b["input_ids"] = butd_tokens(B, T, 3000, seed=1234 + rank)      # dictionary indices, front-padded with the padding index (not BERT word pieces)
    if args.uniter:      # 7-d region position features (x1, y1, x2, y2, w, h, area; tasks/gqa_data.py:240-250)
        bx = b["boxes"]
        w_, h_ = bx[:, :, 2] - bx[:, :, 0], bx[:, :, 3] - bx[:, :, 1]
        b["boxes"] = np.ascontiguousarray(np.stack([bx[:, :, 0], bx[:, :, 1], bx[:, :, 2], bx[:, :, 3], w_, h_, w_ * h_], 2).astype(np.float32))
    dev = {k: torch.from_numpy(v).cuda() for k, v in b.items() if k != "lengths"}
    # real token count of every question (host side, as the tokenizer knows it): the engine packs the language rows
    lengths = None if (args.padded or args.butd) else np.ascontiguousarray(np.tile(b["lengths"], 2 if args.mixup else 1), dtype=np.int32)
    MB = B                       # rows the model sees per step
    if args.mixup:
        from rgqa_amd.mixup import RoIMixup
        MB = 2 * B
        import random as _random
        _random.seed(777 + rank)
        np.random.seed(777 + rank)
        mixer = RoIMixup("mixup_v1", alpha=1.0, beta=5.0)       # run/gqa_mixup_vis_finetune.bash: mixup_v1, Beta(1, 5)
        img_ids = list(range(B))                                  # every synthetic sample is its own image
        loader = {k: dev[k] for k in ("feats", "boxes", "target")}
        ids2 = torch.cat([dev["input_ids"], dev["input_ids"]], 0).contiguous()       # sent = sent + sent (gqa_mixup_vis.py:181)
        mask2 = torch.cat([dev["input_mask"], dev["input_mask"]], 0).contiguous()
        seg2 = torch.cat([dev["segment_ids"], dev["segment_ids"]], 0).contiguous()

        def mixup_batch():
            """the reference's host draws (partner != own image, prop ~ Beta(1,5), int(prop*36) shuffled RoI indices), then ONE
            device gather + target scaling (rgqa_amd.mixup)"""
            f2, b2, t2 = mixer(loader["feats"], loader["boxes"], loader["target"], img_ids)
            dev.update(feats=f2, boxes=b2, target=t2, input_ids=ids2, input_mask=mask2, segment_ids=seg2)
    e.ensure_shape(MB, T, O)
    e.sync_weights()
    comm = make_exchange(e, dist, mode=dp_mode) if dist is not None else None      # RGQA_DP_MODE: sharded (default, payload by precision) | allreduce (f32; _bf16 opts in) (rgqa_amd/parallel.py)
    dp_wire = None
    if dist is not None and dist.get_backend() == "nccl" and not args.lean:
        note("dp_wire: collectives of the exchange on the real chunk sizes")
        dp_wire = dp_wire_probe(dist, comm, torch.device("cuda", local))
    if (dp_wire is not None and (world > 1 or os.environ.get("RGQA_BENCH_RCCL_REHEARSAL") == "1") and dp_mode.partition("_")[0] == "sharded" and "all_to_all" in dp_wire and os.environ.get("RGQA_DP_PEER_PROBE", "1") != "0"
            and dp_wire["all_to_all"]["frac_of_all_links"] < float(os.environ.get("RGQA_DP_PEER_THRESHOLD", "0.7"))):
        # RCCL's all-to-all runs below 70 % of what seven links carry (a ring-shaped schedule shows 1/7): measure the hand-written peer-to-peer exchange
        # (hipIpc buffers, every rank pulling from all of its peers at once: rgqa_amd.parallel.PeerShardedExchange) on the same chunk and take it if it
        # is clearly faster.  Every rank takes the same decision: times are maxima over the ranks, a rank that cannot set the buffers up vetoes.
        note("dp_wire: all-to-all at %.2f of seven links - probing the peer-to-peer exchange" % dp_wire["all_to_all"]["frac_of_all_links"])
        peer, ok = None, torch.ones(1, device="cuda")
        try:
            from rgqa_amd.parallel import PeerShardedExchange
            peer = PeerShardedExchange(e, dist, payload=(dp_mode.partition("_")[2] or None))
            if not peer.selfcheck():
                raise RuntimeError("peer exchange self-check: the pulled data is not what the peers staged")
        except Exception as exn:
            ok.zero_()
            dp_wire["peer_error"] = repr(exn)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) > 0:
            W_, s_ = world, peer.smax
            send, recv = peer._stage[0][0][:W_ * s_], peer._stage[0][1][:W_ * s_]

            def timed_peer(fn, iters=10):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                dist.barrier()
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record()
                for _ in range(iters):
                    fn()
                b_.record()
                torch.cuda.synchronize()
                t_ = torch.tensor([a_.elapsed_time(b_) / iters], device="cuda", dtype=torch.float64)
                dist.all_reduce(t_, op=dist.ReduceOp.MAX)
                return float(t_.item()) * 1e-3
            tp = timed_peer(lambda: peer._a2a(recv, send))
            nbytes = W_ * s_ * send.element_size()
            wire = (W_ - 1) / W_
            dp_wire["peer_all_to_all"] = dict(ms=round(tp * 1e3, 3), gbs_per_gpu_per_direction=round(nbytes * wire / tp / 1e9, 1),
                                              frac_of_all_links=round(nbytes * wire / tp / 1e9 / (XGMI_LINKS * XGMI_GBS_PER_LINK_DIR), 3),
                                              note="stage + barrier + pull from every peer + barrier (two barriers are part of every chunk's exchange)")
            take = tp * 1.1 < dp_wire["all_to_all"]["ms"] * 1e-3
            dp_wire["peer_selected"] = bool(take)
            if take:
                comm, dp_mode = peer, "peer" + dp_mode[len("sharded"):]
                note("dp_wire: the peer-to-peer exchange is faster (%.3f ms vs %.3f): selected" % (tp * 1e3, dp_wire["all_to_all"]["ms"]))
            else:
                peer.close()
        elif peer is not None:
            peer.close()
    if dist is None and not args.butd:
        e.enable_segment_sumsq(True)        # the clip norm's sum(g^2) is taken segment by segment beside backward
    # (under an exchange the norm belongs to the REDUCED gradients: the sharded exchange takes each owner's share while it sums the shards,
    # so backward's per-segment sums of the local gradients are left off - 21 launches of side-stream time; the collective-free legs below
    # switch them on)
    t_total = 10000
    state = dict(step=0, lengths=lengths, comm=comm)

    def step(exchange=True):
        i = state["step"]
        if args.mixup:
            mixup_batch()
        e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=4321 + rank + 1000003 * i, lengths=state["lengths"])
        e.loss_backward(dev["target"])
        lr_t = 1e-5 * warmup_linear(i / t_total, 0.1)
        c = state["comm"]
        if c is not None and exchange:
            c.exchange()
            c.step(lr_t, max_norm=5.0)
        else:       # single GPU, or the collective-free legs after the timed region (local gradients, whole arena)
            e.adam_step(lr_t, max_norm=5.0, grad_prescale=1.0 / world)
        state["step"] = i + 1

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    note("warm-up (%d steps) + timed region (%d steps)" % (args.warmup, args.steps))
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    note("timed region: %.3f ms per step" % (dt / args.steps * 1e3))
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt
    n_ranks_seen = dist.get_world_size() if dist is not None else 1

    def timed_leg(n, **kw):
        """n more steps, bracketed like the timed region; max over ranks, ms per step"""
        fence()
        t1 = time.perf_counter()
        for _ in range(n):
            step(**kw)
        fence()
        d = time.perf_counter() - t1
        if dist is not None:
            t2 = torch.tensor([d], device="cuda", dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            d = float(t2.item())
        return d / n * 1e3

    # diagnostics outside the timed region (every rank takes part): the same exchange issued after backward instead of beside it, and
    # the step without any exchange -> what the exchange costs beyond what backward hides; the reference's padded layout
    dp_legs = None
    if dist is not None and not args.lean:
        n2 = max(3, min(args.steps, 40))       # (10 until round 5: a leg that short carries its own start-up and drain, +0.3..0.6 ms per step)
        dp_legs = {"mode": comm.describe(), "ms_per_step": round(ms, 3)}
        comm.release()                               # every rank holds the full optimizer state again
        if hasattr(comm, "overlap"):
            alt = make_exchange(e, dist, mode=dp_mode, overlap=not comm.overlap)
            state["comm"] = alt
            for _ in range(2):
                step()
            dp_legs["alt_mode"] = alt.describe()
            dp_legs["alt_ms_per_step"] = round(timed_leg(n2), 3)
            alt.release()
        state["comm"] = None
        if not args.butd:
            e.enable_segment_sumsq(True)          # local gradients, whole-arena optimizer: the single-GPU step
        for _ in range(2):
            step(exchange=False)
        local_ms = timed_leg(n2, exchange=False)
        dp_legs["no_exchange_ms_per_step"] = round(local_ms, 3)
        dp_legs["exposed_comm_ms"] = round(ms - local_ms, 3)
        if "alt_ms_per_step" in dp_legs:
            dp_legs["alt_exposed_comm_ms"] = round(dp_legs["alt_ms_per_step"] - local_ms, 3)
    state["comm"] = None          # everything below runs on rank-local state (replicas may diverge from here on: nothing is exchanged again)
    padded_leg = None
    if lengths is not None and not (args.butd or args.uniter or args.mixup or args.lean):
        n2 = max(3, min(args.steps, 40))       # (10 until round 5: a leg that short carries its own start-up and drain, +0.3..0.6 ms per step)
        state["lengths"] = None
        for _ in range(2):
            step(exchange=False)
        pms = timed_leg(n2, exchange=False)
        state["lengths"] = lengths
        step(exchange=False)
        padded_leg = {"ms_per_step": round(pms, 3), "value": round(B * world / pms * 1e3, 1), "unit": "QA-pairs/s",
                      "note": "same build, all %d token positions computed as the reference does (bench.py --padded); %d steps outside the timed region%s" % (
                          MB * T, n2, "" if world == 1 else ", no gradient exchange")}

    # live roofline of the dominant kernel (the NT GEMM family): HIP events around every launch, on the launch stream
    roof = None
    prof = None
    blocks = None
    if rank == 0:
        note("live per-launch timing (%d steps)" % args.profile_steps)
        if hasattr(e, "join_update"):
            e.join_update()
        torch.cuda.synchronize()
        e.profile(True)
        ov, e.adam_overlap = getattr(e, "adam_overlap", False), False      # per-kernel timing with every other stream folded into the launch stream (as the
        for _ in range(args.profile_steps):                                # weight-gradient side stream is under profiling): a kernel's own duration, not its neighbours'
            step(exchange=False)      # rank-0-only kernel timing AFTER the timed region: no collective (the other ranks are at the barrier below)
        e.adam_overlap = ov
        prof = e.profile_read()
        blocks = e.profile_blocks() if hasattr(e, "profile_blocks") and not (args.butd or args.uniter) else None
        e.profile(False)
        x3f = args.precision in ("bf16x3_fwd", "bf16x3")      # the forward launches of the NT family are split-f32 kernels (3 MFMA products per f32 product)
        fw, dg = prof["gemm_nt"], prof.get("gemm_nt_dgrad", dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
        obs = e.profile_operand_bytes() if hasattr(e, "profile_operand_bytes") else {}
        # the dominant kernel: under bf16 ONE kernel family computes forward and dgrad launches alike; under bf16x3_fwd the forward launches
        # (gemm_nt256*_kernel<sf32, ...>) are a family of their own - 60 % of the NT time - and the dgrad launches are the bf16 kernels
        if args.precision == "bf16x3_fwd":
            nt, ob_tot, kname = fw, obs.get("gemm_nt", 0.0), "gemm_nt256_kernel / gemm_nt256d_kernel<sf32, EPI, MT, X3 = true>: every forward GEMM launch (split-f32 operands)"
        else:
            nt = {k: fw[k] + dg[k] for k in ("ms", "flops", "bytes", "launches")}
            ob_tot = obs.get("gemm_nt", 0.0) + obs.get("gemm_nt_dgrad", 0.0)
            kname = "gemm_nt (gemm_nt256_kernel / gemm_nt256d_kernel<OutT,EPI,MT,X3> + gemm_nt_kernel: every forward/dgrad GEMM launch)"
        if nt["launches"]:
            per_launch_flops = nt["flops"] / nt["launches"]
            avg_ms = nt["ms"] / nt["launches"]
            ach = per_launch_flops / (avg_ms * 1e-3) / 1e12
            traffic, traffic_note = None, "no PMC profile for this workload"
            pmc_name = PMC_PROFILES.get(args.precision)
            pmc = os.path.join(ROOT, "profiles", pmc_name) if pmc_name else ""    # separate rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), tools/pmc_summary.py
            if pmc and os.path.exists(pmc) and B == 256 and T == 20 and not (args.mixup or args.uniter or args.butd or args.padded):
                from rgqa_amd.build import source_digest
                pj = json.load(open(pmc))
                if pj.get("kernel_source_digest") == source_digest():
                    traffic, traffic_note = round(pj["traffic_bytes_per_launch"]), "profiles/%s (same kernel sources)" % pmc_name
                else:       # the kernels changed since the counters were collected: a stale figure is worse than none
                    traffic_note = "profiles/%s was collected on other kernel sources (digest mismatch): not reported" % pmc_name
            alg_b, op_b = nt["bytes"] / nt["launches"], (ob_tot / nt["launches"] if ob_tot else None)
            roof = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4),
                        traffic=traffic, traffic_unit="HBM bytes per launch (PMC)", traffic_source=traffic_note, algorithmic_bytes_per_launch=round(alg_b),
                        # the two denominators: since round 4 `algorithmic_bytes_per_launch` counts the fused epilogues' operands
                        # (the residual / gelu' a launch reads, the second output it writes) beside A, W and C; rounds 1-3 counted A, W, C alone
                        gemm_operand_bytes_per_launch=None if op_b is None else round(op_b),
                        traffic_over_algorithmic=None if traffic is None else round(traffic / alg_b, 3),
                        traffic_over_gemm_operands=None if (traffic is None or not op_b) else round(traffic / op_b, 3),
                        kernel=kname, launches_per_step=nt["launches"] // args.profile_steps,
                        avg_launch_us=round(avg_ms * 1e3, 2), gflop_per_launch=round(per_launch_flops / 1e9, 3))
            if x3f:
                # `achieved` counts ONE product per f32 product (2 M N K): what the reference's f32 GEMM computes.  The matrix pipe issues three bf16
                # products for each (hi*hi + hi*lo + lo*hi): the MFMA work the kernel really does is 3x, priced against the same dense-bf16 peak
                roof["mfma_issued"] = dict(achieved=round(3 * ach, 2), unit="TFLOP/s", frac=round(3 * ach / PEAK_BF16_TFLOPS, 4),
                                           note="3 bf16 MFMA products per counted f32 product")
            if args.precision == "bf16x3_fwd" and dg["launches"]:
                dms = dg["ms"] / dg["launches"]
                dach = dg["flops"] / dg["launches"] / (dms * 1e-3) / 1e12
                roof["dgrad_family"] = dict(kernel="gemm_nt256*_kernel<bf16, ...>: the dgrad launches (bf16 operands)", launches_per_step=dg["launches"] // args.profile_steps,
                                            avg_launch_us=round(dms * 1e3, 2), gflop_per_launch=round(dg["flops"] / dg["launches"] / 1e9, 3), achieved=round(dach, 2),
                                            unit="TFLOP/s", frac=round(dach / PEAK_BF16_TFLOPS, 4), algorithmic_bytes_per_launch=round(dg["bytes"] / dg["launches"]))
            if not (args.butd or args.lean):
                try:
                    mhz, nl = probe_sustained_clock(e)
                    sus = 256 * 4 * 1024 * mhz * 1e6 / 1e12
                    roof["peak_sustained"] = dict(value=round(sus, 1), unit="TFLOP/s", shader_clock_mhz=round(mhz, 1), frac=round(ach / sus, 4),
                                                  how="256 CUs x 4 SIMDs x 1024 FLOP/clk x the in-kernel clock (s_memtime / s_memrealtime over every block's life) of the stamped "
                                                      "persistent NT kernel, %d back-to-back launches of 12356x2304x768 on random operands" % nl)
                except Exception as ex:          # the probe is a diagnostic: never lose the line over it
                    roof["peak_sustained"] = dict(error=str(ex))
    if dist is not None:
        dist.barrier()

    HEAD = "bf16x3_fwd"
    extra_legs = rank == 0 and world == 1 and not (args.no_extra_legs or args.lean or args.butd or args.uniter or args.mixup or args.padded) and args.precision == HEAD
    cfg3 = tol = seq30 = fwd_only = dropin = other = parity = None
    engines = {args.precision: e}
    if extra_legs:
        n2 = max(5, min(args.steps, 40))        # (20 until round 5: a leg's start-up and drain are 1-2 % of twenty 11-ms steps)

        def second_engine(prec):
            en = Engine(precision=prec, **FULL).allocate("cuda")
            en.params.copy_(e.params)
            en.ensure_shape(B, T, O)
            en.sync_weights()
            en.enable_segment_sumsq(True)
            engines[prec] = en
            return en
        # ---- BASELINE config 3 with a bf16 FORWARD pass as well (rounds 1-5's headline): faster, but its logits are outside the north star's bound
        note("leg: bf16 train step (config3_bf16)")
        eb = second_engine("bf16")
        bms = time_steps(engine_step_fn(eb, dev, lengths), n2, 3)
        cfg3 = dict(precision="bf16", ms_per_step=round(bms, 3), value=round(B / bms * 1e3, 1), unit="QA-pairs/s", steps=n2,
                    note="same workload, weights and batch as the headline with bf16 operands in the forward pass too (rgqa.h RGQA_PRECISION_BF16): NOT inside the 1e-3 logits "
                         "bound (logits_max_err below, re-measured in-run) - reported for what the bound costs, never as `value`")
        # ---- split-f32 operands throughout (forward and backward)
        note("leg: bf16x3 train step")
        ex = second_engine("bf16x3")
        xms = time_steps(engine_step_fn(ex, dev, lengths), n2, 3)
        tol = dict(precision="bf16x3", ms_per_step=round(xms, 3), value=round(B / xms * 1e3, 1), unit="QA-pairs/s", steps=n2,
                   note="same workload, weights and batch as the headline; split-f32 operands and 3 bf16 MFMA products per f32 product in the backward pass as well "
                        "(rgqa.h RGQA_PRECISION_BF16X3): gradients at f32-class accuracy")
        # ---- BASELINE config 2: forward-only inference at B=256
        fwd_only = {}
        note("leg: forward-only B=256, three precisions")
        for name, en in engines.items():
            fms = time_steps(lambda en=en: en.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=False, lengths=lengths), n2, 3)
            fwd_only[name] = dict(ms=round(fms, 3), value=round(B / fms * 1e3, 1), unit="QA-pairs/s",
                                  reference_equivalent_tflops=round(B / fms * 1e3 * FWD_GFLOP.get(T, FWD_GFLOP[20]) / 1e3, 1))
        # ---- T = 30: what the GQA wrapper pads every question to (tasks/gqa_model.py:11 MAX_GQA_LENGTH; SURVEY §8 D1 asks for it alongside T = 20)
        try:
            note("leg: T=30 train step (seq30)")
            b30 = synth.synth_batch(B, 30, seed=1234 + rank)
            d30 = {k: torch.from_numpy(v).cuda() for k, v in b30.items() if k != "lengths"}
            l30 = np.ascontiguousarray(b30["lengths"], dtype=np.int32)
            e.ensure_shape(B, 30, O)
            e.sync_weights()
            e.enable_segment_sumsq(True)
            s30 = time_steps(engine_step_fn(e, d30, l30), n2, 3)
            p30 = time_steps(engine_step_fn(e, d30, None), max(5, n2 // 2), 2)
            seq30 = dict(precision=args.precision, seq_len=30, ms_per_step=round(s30, 3), value=round(B / s30 * 1e3, 1), unit="QA-pairs/s", steps=n2,
                         language_rows="packed: %d real tokens of %d positions (question length ~ U{5..30})" % (int(l30.sum()), B * 30),
                         padded_ms_per_step=round(p30, 3), padded_value=round(B / p30 * 1e3, 1),
                         reference_equivalent_tflops=round(B / s30 * 1e3 * FWD_BWD_GFLOP[30] / 1e3, 1),
                         note="same engine and weights re-bound to T = 30 (tasks/gqa_model.py:11); padded_* computes all 30 positions of every question as the reference does")
            del d30
            e.ensure_shape(B, T, O)
            e.sync_weights()
            e.enable_segment_sumsq(True)
        except Exception as exn:
            seq30 = dict(error=repr(exn))
        # ---- the reference trainer's statements through the drop-in modules, in the headline's precision
        try:
            note("leg: drop-in trainer step")
            dms = dropin_step_leg(B, T, n2, args.precision)
            dropin = dict(ms_per_step=round(dms, 3), value=round(B / dms * 1e3, 1), unit="QA-pairs/s", steps=n2, vs_headline=round(dms / ms, 3), precision=args.precision,
                          what="GQAModel(feats, boxes, list_of_str) + BCE x NA + backward + nn.utils.clip_grad_norm_ + BertAdam.step (tasks/gqa_conf.py:174-202) on a fresh "
                               "batch per step from the f16 feature store (pinned gather + H2D + device preparation); the headline's precision")
        except Exception as exn:
            dropin = dict(error=repr(exn))
        if dist is not None:
            # one-rank RCCL rehearsal of the drop-in trainer's data-parallel step (RGQA_BENCH_RCCL_REHEARSAL=1): the unchanged loop with the exchange inside
            # backward() - `allreduce` (every rank steps every parameter) and `sharded` (round 6: BertAdam.step owns 1/N of the arena) - through the library
            dropin["rccl_rehearsal"] = {}
            for dmode in ("allreduce", "sharded"):
                keep = {k: os.environ.get(k) for k in ("RGQA_DP_MODE", "RGQA_DP_REHEARSAL")}
                try:
                    os.environ["RGQA_DP_MODE"], os.environ["RGQA_DP_REHEARSAL"] = dmode, "1"
                    note("leg: drop-in trainer step under a one-rank RCCL group, %s" % dmode)
                    rms = dropin_step_leg(B, T, n2, args.precision)
                    dropin["rccl_rehearsal"][dmode] = dict(ms_per_step=round(rms, 3), vs_headline=round(rms / ms, 3))
                except Exception as exn:
                    dropin["rccl_rehearsal"][dmode] = dict(error=repr(exn))
                finally:
                    for k, v in keep.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
        # ---- BASELINE configs 4 and 5 under the same clock, each in a mode inside the bound and in bf16: the RoI-mixup step (2B model rows per B loader
        # pairs) on the engines above re-bound to 2B rows, and the BUTD backbone's step
        other = {}
        n3 = 10
        for key, en, inb in (("roi_mixup_b256", e, True), ("roi_mixup_b256_bf16", eb, False)):
            try:
                note("leg: RoI-mixup step (config 4), 2 x %d rows, %s" % (B, en.precision))
                mms = mixup_leg(en, dev, lengths, B, T, O, n3, rank)
                other[key] = dict(ms_per_step=round(mms, 3), loader_qa_per_s=round(B / mms * 1e3, 1), model_rows_per_s=round(2 * B / mms * 1e3, 1), steps=n3, dtype=en.precision,
                                  within_logits_bound=inb,
                                  what="BASELINE config 4 on one GPU: RoI-mixup finetune step (tasks/gqa_mixup_vis.py:134-181, 250-259; mixup_v1, Beta(1,5)): host draws + one "
                                       "device gather, 2 x %d model rows per %d loader QA pairs, fwd+BCE+bwd+clip+BertAdam; packed language rows" % (B, B))
                en.ensure_shape(B, T, O)              # back to the B-row binding (the oracle check below runs on it)
                en.sync_weights()
            except Exception as exn:
                other[key] = dict(error=repr(exn))
        for key, prec, inb in (("butd_b256", "bf16x3", True), ("butd_b256_bf16", "bf16", False)):
            try:
                note("leg: BUTD step (config 5), %s" % prec)
                bms_ = butd_leg(B, 10, prec, rank)
                other[key] = dict(ms_per_step=round(bms_, 3), value=round(B / bms_ * 1e3, 1), unit="QA-pairs/s", steps=10, dtype=prec, within_logits_bound=inb,
                                  what="BASELINE config 5 on one GPU: BUTD backbone train step (butd/butd.py:195-221): GRU over 40 tokens x 1024, region attention over 36 RoIs, "
                                       "classifier; dictionary 3000 words; fwd+BCE+bwd+clip+BertAdam" + ("; split-f32 operands (logits within 1e-3 of the reference: tests/test_gpu_butd.py)" if inb else
                                       "; bf16 operands (logits gated at 1e-2: outside the bound)"))
            except Exception as exn:
                other[key] = dict(error=repr(exn))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.uniter:
        def check(P, cb, ref_logits):
            """the GPU engines on the oracle's weights and eval batch: max / mean |logit - oracle logit| over all 256 x 1842 entries"""
            if args.butd or args.mixup or args.padded or T != 20 or B != 256:
                return None
            d2 = {k: torch.from_numpy(v).cuda() for k, v in cb.items() if k != "lengths"}
            res = {}
            for name, en in engines.items():
                keep = en.params.clone()
                for sp in en.specs:
                    en.view(en.params, sp).copy_(P[sp.name])
                en.sync_weights()
                lg = en.forward(d2["feats"], d2["boxes"], d2["input_ids"], d2["input_mask"], d2["segment_ids"], train=False,
                                lengths=np.ascontiguousarray(cb["lengths"], dtype=np.int32))[0]
                err = (lg.cpu() - ref_logits).abs()
                res[name] = dict(logits_max_err=float(err.max()), logits_mean_err=float(err.mean()))
                en.params.copy_(keep)
                en.sync_weights()
            return res
        note("leg: CPU baseline (the oracle on the host cores) + logits check of the GPU engines")
        cpu, checked = cpu_baseline(T, args.cpu_sample, 5, extra=not args.butd, check=check)
        note("CPU baseline done")
        if checked:
            on = "B=256 eval forward, golden-fixture filler weights, all 256 x 1842 logits against the CPU oracle (f32)"
            hd = checked[args.precision]
            parity = dict(precision=args.precision, logits_max_err=hd["logits_max_err"], logits_mean_err=hd["logits_mean_err"], bound=1e-3,
                          within_bound=bool(hd["logits_max_err"] <= 1e-3), checked_on=on)
            for leg, name in ((cfg3, "bf16"), (tol, "bf16x3")):
                if leg is not None and name in checked:
                    leg.update(logits_max_err=checked[name]["logits_max_err"], logits_mean_err=checked[name]["logits_mean_err"], bound=1e-3,
                               within_bound=bool(checked[name]["logits_max_err"] <= 1e-3), checked_on=on)
            if fwd_only is not None:
                for name in fwd_only:
                    if name in checked:
                        fwd_only[name]["logits_max_err"] = checked[name]["logits_max_err"]

    if rank == 0:
        # UNITER, padded: 12 layers x (56 x 7,077,888 + 2 x 56^2 x 768) MAC + 36 x 2048 x 768 + head = 4.880 GMAC fwd per QA pair; x2 FLOP, x3 fwd+bwd
        per_pair = 1.5 if args.butd else (29.28 if args.uniter else FWD_BWD_GFLOP.get(T, FWD_BWD_GFLOP[20]))
        step_tflops = value * per_pair / 1e3 * (MB // B)     # BUTD ~0.5 GFLOP fwd / QA pair
        out = {
            "metric": "QA-pairs/sec (train step) BUTD-GQA B=256" if args.butd else ("QA-pairs/sec (train step) UNITER-GQA B=256" if args.uniter else "QA-pairs/sec (train step) LXMERT-GQA B=256"), "value": round(value, 1), "unit": "QA-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BUTD-GQA finetune train step (GRU 40x1024 + region attention + classifier, fwd+BCE+bwd+clip+BertAdam)" if args.butd else "UNITER-GQA finetune train step (fwd+BCE+bwd+clip+BertAdam), 12 BertLayers over [text ; 36 regions], H=768" if args.uniter else ("LXMERT-GQA RoI-mixup finetune train step (device mixup + fwd+BCE+bwd+clip+BertAdam), 9/5/5 layers, H=768" if args.mixup else
                                    "LXMERT-GQA RP finetune train step (fwd+BCE+bwd+clip+BertAdam), 9/5/5 layers, H=768"), "model_rows_per_gpu": MB,
                       "per_gpu_batch": B, "global_batch": B * world, "seq_len": T, "rois": O, "feat_dim": 2048,
                       "num_answers": 1842, "parallelism": "dp%d" % world, "dropout": 0.1,
                       "precision_note": {"bf16": "bf16 operands in the forward pass too; logits outside the north star's 1e-3 bound",
                                          "bf16x3": "split-f32 operands throughout: inside the 1e-3 logits bound",
                                          "bf16x3_fwd": "forward pass on split-f32 operands, 3 bf16 MFMA products per f32 product: logits inside the 1e-3 bound (parity_in_run); "
                                                        "backward pass in bf16 as BASELINE config 3 prescribes (gradients carry bf16 rounding; weights, moments, optimizer f32)",
                                          "f32": "exact f32 on the vector ALU"}.get(args.precision, ""),
                       "language_rows": ("padded: all %d token positions computed" % (MB * T)) if lengths is None else
                                        ("packed: %d real tokens of %d positions (question length ~ U{5..%d}); padding rows are not computed, results identical" % (int(lengths.sum()), MB * T, T))},
            "n_ranks_seen": n_ranks_seen,
            "roofline": roof, "cpu_baseline": cpu,
            # NOT a utilisation figure: the FLOPs the REFERENCE's padded computation would need for the same QA-pairs/s
            # (30.339 GFLOP per QA pair); with packed language rows part of them is never executed here
            "reference_equivalent_tflops_per_gpu": round(step_tflops / world, 1),
        }
        if parity is not None:
            out["parity_in_run"] = parity
        if cfg3 is not None:
            out["config3_bf16"] = cfg3
        if tol is not None:
            out["tolerance_compliant"] = tol
        if seq30 is not None:
            out["seq30"] = seq30
        if fwd_only is not None:
            out["forward_only_b256"] = fwd_only
        if dropin is not None:
            out["dropin_step"] = dropin
        if other:
            out["other_workloads"] = other
        if dist is not None:
            out["dp_mode"] = dp_mode
            out["dp_fallback"] = os.environ.get("RGQA_BENCH_DP_FALLBACK")
            if world == 1:
                out["rccl_rehearsal"] = "one rank on RCCL: every collective of the exchange runs through the library; the exchange's local arithmetic is timed at full size"
        if dp_wire is not None:
            out["dp_wire"] = dp_wire
        if dp_legs is not None:
            out["exposed_comm_ms"] = dp_legs["exposed_comm_ms"]      # step time minus the time of the same step with no exchange and a local whole-arena optimizer
            out["dp_exchange"] = dp_legs
        if padded_leg is not None:
            out["padded_layout"] = padded_leg
        if prof is not None and args.profile_steps > 0:
            out["kernel_ms_per_step"] = {k: round(v["ms"] / args.profile_steps, 3) for k, v in prof.items() if v["launches"]}
            if "gemm_nt_dgrad" in out["kernel_ms_per_step"]:       # (rounds 1-5 reported the forward and dgrad launches of the NT family as one figure)
                out["kernel_ms_per_step"]["gemm_nt_fwd_plus_dgrad"] = round(out["kernel_ms_per_step"].get("gemm_nt", 0.0) + out["kernel_ms_per_step"]["gemm_nt_dgrad"], 3)
            # the utilisation figure: FLOPs the engine actually executed (GEMMs + attention, packed rows) over the measured step time
            ex_fl = sum(v["flops"] for v in prof.values()) / max(1, args.profile_steps)
            out["step_executed_tflops_per_gpu"] = round(ex_fl / (ms * 1e-3) / 1e12, 1)
            out["step_executed_frac_of_bf16_peak"] = round(ex_fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
            # the same figure for the legs that satisfy the 1e-3 logits bound (same workload: the same FLOPs are credited; the bf16x3 kernels issue
            # three MFMA products per credited product, which `executed` does not count)
            for leg in (cfg3, tol):
                if leg is not None and leg.get("ms_per_step"):
                    leg["executed_tflops_per_gpu"] = round(ex_fl / (leg["ms_per_step"] * 1e-3) / 1e12, 1)
                    leg["executed_frac"] = round(ex_fl / (leg["ms_per_step"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
            if blocks is not None:
                # the block the north-star target names: the five LXRTXLayers (cross-attention + self-attention + FFN of both
                # modalities, forward + backward incl. their weight gradients): executed GEMM + attention FLOPs over the sum of
                # ALL kernel durations of those layers (LayerNorm, attention and epilogue time included)
                out["block_ms_per_step"] = {k: round(v["ms"] / args.profile_steps, 3) for k, v in blocks.items()}
                xb = blocks["cross_modality_layers"]
                if xb["ms"] > 0:
                    tf = xb["flops"] / (xb["ms"] * 1e-3) / 1e12
                    out["cross_attention_block"] = {"ms_per_step": round(xb["ms"] / args.profile_steps, 3), "gflop_per_step": round(xb["flops"] / args.profile_steps / 1e9, 1),
                                                    "achieved": round(tf, 1), "unit": "TFLOP/s", "frac_of_bf16_peak": round(tf / PEAK_BF16_TFLOPS, 4)}
        emit_json(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
