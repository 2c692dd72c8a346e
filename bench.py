"""Headline benchmark: QA-pairs/sec of one LXMERT-GQA train step (BASELINE.json), B=256 per GPU, T=20, O=36,
bf16 MFMA operands / f32 accumulate, synthetic inputs already resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W            (N > 1 without WORLD_SIZE: starts its own N rank processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = forward (train mode, dropout 0.1) + BCE x NA loss + backward + [RCCL gradient all-reduce] +
clip_grad_norm_(5.) + BertAdam + bf16 weight re-cast: everything tasks/gqa_conf.py:174-202 does per batch.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_BWD_GFLOP = {20: 30.3388, 30: 37.0403}   # per QA pair, SURVEY.md §8 D3
PEAK_BF16_TFLOPS = 2500.0                    # dense MFMA bf16, MI355X_MICROARCH.md
PMC_PROFILE = "r02_pmc_gemm_nt.json"
FULL = dict(vocab_size=30522, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=9, x_layers=5,
            r_layers=5, feat_dim=2048, pos_dim=4, num_answers=1842)


def warmup_linear(x, warmup):
    return x / warmup if x < warmup else max((x - 1.0) / (warmup - 1.0), 0.0)


def init_params(e, seed):
    """random-init weights of the reference architecture (init_bert_weights: N(0, 0.02), LN = 1/0, bias = 0)."""
    g = torch.Generator(device=e.device).manual_seed(seed)
    e.params.normal_(0.0, 0.02, generator=g)
    for sp in e.specs:
        v = e.view(e.params, sp)
        if len(sp.shape) == 1:
            v.fill_(1.0 if ("LayerNorm.weight" in sp.name or "layer_norm.weight" in sp.name or sp.name == "logit_fc.2.weight") else 0.0)


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(T, sample_b, iters, warm=2, extra=True):
    """The oracle (CPU restatement pinned to the reference's golden vectors) timed on this box's host cores: `warm` untimed +
    `iters` timed full train steps at B=sample_b (the headline entry), plus SURVEY §8 D4's other cases in `cases`:
    BASELINE config 1 (B=4 train step) and the B=256 eval forward."""
    from oracle import lxmert_ref as R
    from rgqa_amd import synth
    cfg = R.RefConfig(**FULL)
    try:
        ncpu = len(os.sched_getaffinity(0))
    except Exception:
        ncpu = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncpu, 16)))     # the box's CPU share for one GPU
    torch.manual_seed(0)
    P = {}
    for k, shp in R.param_shapes(cfg).items():
        t = torch.randn(shp) * 0.02 if len(shp) > 1 else (torch.ones(shp) if "LayerNorm.weight" in k or "layer_norm.weight" in k else torch.zeros(shp))
        P[k] = t.requires_grad_(True)
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

    t = train_case(sample_b, warm, iters)
    out = dict(value=sample_b / t, unit="QA-pairs/s", cores=torch.get_num_threads(), kind="port", cpu_model=_cpu_model(),
               sample="full train step (fwd+BCE+bwd+clip+BertAdam), B=%d T=%d, fp32, %d warm-up + %d timed iters, median %.2fs" % (sample_b, T, warm, iters, t))
    if extra:
        cases = {}
        t4 = train_case(4, 1, 3)
        cases["train_step_B4"] = dict(value=round(4 / t4, 2), unit="QA-pairs/s", sample="BASELINE config 1: full train step B=4 T=%d, 1 warm-up + 3 timed, median %.2fs" % (T, t4))
        b = synth.synth_batch(256, T, seed=98)
        batch = {k: torch.from_numpy(v) for k, v in b.items() if k != "lengths"}
        with torch.no_grad():
            Pd = {k: v.detach() for k, v in P.items()}
            te = timed(lambda: R.gqa_forward(Pd, cfg, batch["feats"], batch["boxes"], batch["input_ids"], batch["input_mask"], batch["segment_ids"]), 0, 2)
        cases["eval_forward_B256"] = dict(value=round(256 / te, 2), unit="QA-pairs/s", sample="eval forward B=256 T=%d, 2 timed, median %.2fs" % (T, te))
        out["cases"] = cases
    return out


_JSON_FD = None


def emit_json(obj):
    """the one stdout line of a run (see main(): fd 1 itself is routed to stderr while the ranks run)"""
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_JSON_FD, line)


def self_launch(n):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this parent - which never touches the GPU, so nothing
    that has initialised HIP is ever exec'd or forked - starts N fresh rank processes of this same command line with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (what lxrt/entry.py:102-103's nn.DataParallel switch is to the reference),
    relays rank 0's JSON line and returns non-zero when any rank fails (the survivors are then ended by PID)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), RGQA_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))     # only rank 0 writes to stdout
    deadline = time.time() + float(os.environ.get("RGQA_BENCH_LAUNCH_TIMEOUT", "1500"))
    rc = 0
    live = list(procs)
    while live and rc == 0:
        for pr in list(live):
            c = pr.poll()
            if c is not None:
                live.remove(pr)
                if c != 0:
                    rc = c if c > 0 else 1
        if time.time() > deadline:
            rc = 124
        if live and rc == 0:
            time.sleep(0.05)
    for pr in live:             # a rank failed (or the launch timed out): end exactly the processes started here
        pr.terminate()
    for pr in live:
        try:
            pr.wait(timeout=20)
        except subprocess.TimeoutExpired:
            pr.kill()
    if rc != 0:
        sys.stderr.write("bench.py: a rank process failed (exit code %d); %d rank(s) were ended\n" % (rc, len(live)))
    return rc


def launch_check(world, rank, fail_rank):
    """rehearsal of the launch path without a GPU (tests/test_host.py): rendezvous over gloo, one all-reduce, one JSON line"""
    import torch.distributed as dist
    if rank == fail_rank:
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group("gloo")
    t = torch.ones(1)
    if world > 1:
        dist.all_reduce(t)
        seen = dist.get_world_size()
    else:
        seen = 1
    if rank == 0:
        emit_json({"launch_check": True, "n_gpus": world, "n_ranks_seen": seen, "sum": float(t.item())})
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256, help="QA pairs per GPU per step")
    ap.add_argument("--seq", type=int, default=20)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32)
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--padded", action="store_true", help="compute every padded token position (the reference's layout) instead of packing the real tokens")
    ap.add_argument("--butd", action="store_true", help="BASELINE config 5: BUTD backbone (butd/butd.py) train step, B per GPU, 40 tokens, dictionary 3000")
    ap.add_argument("--uniter", action="store_true", help="UNITER backbone (uniter/uniter.py GQAUNITER): 12 BertLayers over [20 text ; 36 region] sequences, bert-base-cased sizes")
    ap.add_argument("--mixup", action="store_true",
                    help="BASELINE config 4: RoI-mixup finetune (gqa_mixup_vis.py:134-181): every loader batch is doubled on the device "
                         "(mixup_v1, Beta(1,5)); the model sees 2x rows per QA pair; value still counts loader QA pairs")
    ap.add_argument("--lean", action="store_true", help="only the warm-up and timed steps (no padded-layout / exchange-free legs, no live kernel timing, no CPU baseline): what runs under rocprofv3")
    ap.add_argument("--launch-check", action="store_true", help=argparse.SUPPRESS)          # CPU rehearsal of the N-rank launch path
    ap.add_argument("--launch-check-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline, args.profile_steps = True, 0

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))     # before any torch.cuda / HIP call
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
        return launch_check(world, rank, args.launch_check_fail_rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda unavailable); there is no CPU path")
    local %= max(1, torch.cuda.device_count())      # rehearsal of N ranks on a box with fewer GPUs (the driver's node has one per rank)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if torch.cuda.device_count() >= int(os.environ.get("LOCAL_WORLD_SIZE", world)):
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:       # rehearsal only: RCCL refuses two ranks on one device; gloo stages the same all-reduce calls through the host
            dist.init_process_group("gloo")

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
    comm = make_exchange(e, dist) if world > 1 else None      # RGQA_DP_MODE: sharded (default) | allreduce | allreduce_bf16
    if world == 1 and not args.butd and os.environ.get("RGQA_SEG_SUMSQ", "1") != "0":
        e.enable_segment_sumsq(True)        # the clip norm's sum(g^2) is taken segment by segment beside backward
    t_total = 10000
    state = dict(step=0, lengths=lengths)

    def step(exchange=True):
        i = state["step"]
        if args.mixup:
            mixup_batch()
        e.forward(dev["feats"], dev["boxes"], dev["input_ids"], dev["input_mask"], dev["segment_ids"], train=True, seed=4321 + rank + 1000003 * i, lengths=state["lengths"])
        e.loss_backward(dev["target"])
        lr_t = 1e-5 * warmup_linear(i / t_total, 0.1)
        if comm is not None and exchange:
            comm.exchange()
            comm.step(lr_t, max_norm=5.0)
        else:       # single GPU, or the collective-free legs after the timed region (local gradients, whole arena)
            e.adam_step(lr_t, max_norm=5.0, grad_prescale=1.0 / world)
        state["step"] = i + 1

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
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

    # diagnostics outside the timed region (every rank takes part): the step without the gradient exchange -> what the exchange
    # costs beyond what backward hides; the reference's padded layout (all B*T token positions computed)
    exposed_comm_ms = None
    if dist is not None and not args.lean:
        n2 = max(3, min(args.steps, 10))
        exposed_comm_ms = round(ms - timed_leg(n2, exchange=False), 3)
    padded_leg = None
    if lengths is not None and not (args.butd or args.uniter or args.mixup or args.lean):
        n2 = max(3, min(args.steps, 10))
        state["lengths"] = None
        for _ in range(2):
            step()
        pms = timed_leg(n2)
        state["lengths"] = lengths
        step()
        padded_leg = {"ms_per_step": round(pms, 3), "value": round(B * world / pms * 1e3, 1), "unit": "QA-pairs/s",
                      "note": "same build, all %d token positions computed as the reference does (bench.py --padded); %d steps outside the timed region" % (MB * T, n2)}

    # live roofline of the dominant kernel (the bf16 MFMA NT GEMM): HIP events around every launch, on the launch stream
    roof = None
    prof = None
    blocks = None
    if rank == 0:
        e.profile(True)
        for _ in range(args.profile_steps):
            step(exchange=False)      # rank-0-only kernel timing AFTER the timed region: no collective (the other ranks are at the barrier below)
        prof = e.profile_read()
        blocks = e.profile_blocks() if hasattr(e, "profile_blocks") and not (args.butd or args.uniter) else None
        e.profile(False)
        nt = prof["gemm_nt"]
        if nt["launches"]:
            per_launch_flops = nt["flops"] / nt["launches"]
            avg_ms = nt["ms"] / nt["launches"]
            ach = per_launch_flops / (avg_ms * 1e-3) / 1e12
            traffic, traffic_note = None, "no PMC profile for this workload"
            pmc = os.path.join(ROOT, "profiles", PMC_PROFILE)    # separate rocprofv3 --pmc passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), tools/pmc_summary.py
            if os.path.exists(pmc) and B == 256 and T == 20 and not (args.mixup or args.uniter or args.butd or args.padded):
                from rgqa_amd.build import source_digest
                pj = json.load(open(pmc))
                if pj.get("kernel_source_digest") == source_digest():
                    traffic, traffic_note = round(pj["traffic_bytes_per_launch"]), "profiles/%s (same kernel sources)" % PMC_PROFILE
                else:       # the kernels changed since the counters were collected: a stale figure is worse than none
                    traffic_note = "profiles/%s was collected on other kernel sources (digest mismatch): not reported" % PMC_PROFILE
            roof = dict(bound="mfma", achieved=round(ach, 2), peak=PEAK_BF16_TFLOPS, unit="TFLOP/s", frac=round(ach / PEAK_BF16_TFLOPS, 4),
                        traffic=traffic, traffic_unit="HBM bytes per launch (PMC)", traffic_source=traffic_note, algorithmic_bytes_per_launch=round(nt["bytes"] / nt["launches"]), kernel="gemm_nt (gemm_nt8p_kernel / gemm_nt256_kernel / gemm_nt256d_kernel<EPI,MT> + gemm_nt_kernel: every forward/dgrad GEMM launch)", launches_per_step=nt["launches"] // args.profile_steps,
                        avg_launch_us=round(avg_ms * 1e3, 2), gflop_per_launch=round(per_launch_flops / 1e9, 3))
    if dist is not None:
        dist.barrier()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.uniter:
        cpu = cpu_baseline(T, args.cpu_sample, 5)

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
                       "language_rows": ("padded: all %d token positions computed" % (MB * T)) if lengths is None else
                                        ("packed: %d real tokens of %d positions (question length ~ U{5..%d}); padding rows are not computed, results identical" % (int(lengths.sum()), MB * T, T))},
            "n_ranks_seen": n_ranks_seen,
            "roofline": roof, "cpu_baseline": cpu,
            # NOT a utilisation figure: the FLOPs the REFERENCE's padded computation would need for the same QA-pairs/s
            # (30.339 GFLOP per QA pair); with packed language rows part of them is never executed here
            "reference_equivalent_tflops_per_gpu": round(step_tflops / world, 1),
        }
        if exposed_comm_ms is not None:
            out["exposed_comm_ms"] = exposed_comm_ms      # step time minus the time of the same step with no exchange and a local whole-arena optimizer
            out["dp_exchange"] = comm.describe() if hasattr(comm, "describe") else "all_reduce"
        if padded_leg is not None:
            out["padded_layout"] = padded_leg
        if prof is not None:
            out["kernel_ms_per_step"] = {k: round(v["ms"] / args.profile_steps, 3) for k, v in prof.items() if v["launches"]}
            # the utilisation figure: FLOPs the engine actually executed (GEMMs + attention, packed rows) over the measured step time
            ex = sum(v["flops"] for v in prof.values()) / max(1, args.profile_steps)
            out["step_executed_tflops_per_gpu"] = round(ex / (ms * 1e-3) / 1e12, 1)
            out["step_executed_frac_of_bf16_peak"] = round(ex / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4)
            if blocks is not None and args.profile_steps > 0:
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
