"""Data-parallel step == single-process step on the concatenated batch, two ranks sharing the one
GPU of the test box over gloo (RCCL needs one device per rank; the code path — segment events, side stream, bucketed
async all-reduce, grad_prescale — is the same one bench.py runs over RCCL)."""
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MED = dict(vocab_size=512, hidden=128, heads=2, inter=256, max_pos=64, type_vocab=2, l_layers=3, x_layers=2, r_layers=2,
           feat_dim=64, pos_dim=4, num_answers=70)
B, T, O = 4, 12, 10


def _make(batch, precision="f32"):
    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    e = Engine(precision=precision, hidden_dropout=0.0, attn_dropout=0.0, **MED).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    d = {k: torch.from_numpy(v).cuda() for k, v in batch.items() if k != "lengths"}
    e.ensure_shape(d["feats"].shape[0], T, O)
    e.sync_weights()
    return e, d


def _step(e, d, comm, world):
    e.forward(d["feats"], d["boxes"], d["input_ids"], d["input_mask"], d["segment_ids"], train=False)
    e.loss_backward(d["target"])
    if comm is not None:
        comm.exchange()
        comm.step(1e-3, max_norm=5.0)
    else:
        e.adam_step(1e-3, max_norm=5.0, grad_prescale=1.0 / world)


def _full_batch():
    from rgqa_amd import synth
    return synth.synth_batch(2 * B, T, O=O, F=MED["feat_dim"], NA=MED["num_answers"], vocab=MED["vocab_size"], seed=31, min_len=2)


def _worker(rank, world, port, mode, overlap, precision, q):
    import torch.distributed as dist
    from rgqa_amd.parallel import make_exchange
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    full = _full_batch()
    shard = {k: v[rank * B:(rank + 1) * B] for k, v in full.items()}
    e, d = _make(shard, precision)
    kw = dict(bucket_mb=1, overlap=overlap) if not mode.startswith("sharded") else dict(chunk_mb=1, bucket_mb=1, overlap=overlap)
    if mode == "sharded" and precision == "bf16" and overlap:
        kw["f32_chunk_elems"] = 1 << 14        # the word-embedding table (64 k elements here) then takes the f32-chunk path, the rest the packed one
    comm = make_exchange(e, dist, mode, **kw)
    for _ in range(2):
        _step(e, d, comm, world)
    # what the NEXT forward would read from the f32 masters (biases, LayerNorm, embedding tables): current on every rank without any gather
    f32_read = torch.cat([e.params[sp.offset:sp.offset + sp.numel] for sp in e.specs if sp.f32_read and not sp.dead]).cpu().numpy()
    comm.gather_master()
    torch.cuda.synchronize()
    f32_after = torch.cat([e.params[sp.offset:sp.offset + sp.numel] for sp in e.specs if sp.f32_read and not sp.dead]).cpu().numpy()
    assert np.array_equal(f32_read, f32_after), "rank %d: f32-read parameters were stale before gather_master()" % rank
    nb = len(comm.buckets) if hasattr(comm, "buckets") else len(comm.chunks)
    q.put((rank, e.params.cpu().numpy(), None if e.params_lp is None else (e.params_lp.float() if e.params_lp.dtype == torch.bfloat16 else e.params_lp).cpu().numpy(), nb))
    dist.destroy_process_group()


_MODES = ["allreduce", "allreduce_bf16", "sharded", "sharded_bf16", "allreduce_f32", "peer"]
_PRECS = ["f32", "bf16", "bf16x3", "bf16x3_fwd"]


@pytest.mark.parametrize("mode,overlap,precision", [("allreduce", True, "f32"), ("allreduce", False, "f32"), ("allreduce_bf16", True, "f32"),
                                                     ("sharded", False, "f32"), ("sharded", False, "bf16"), ("sharded", True, "f32"), ("sharded", True, "bf16"),
                                                     ("sharded", True, "bf16x3"), ("sharded_bf16", True, "bf16x3"), ("sharded", True, "bf16x3_fwd"), ("allreduce", True, "bf16")])
def test_two_rank_step_equals_single_rank_on_concatenated_batch(mode, overlap, precision):
    """every exchange mode of rgqa_amd.parallel: two ranks, two optimizer steps == one rank on the concatenated batch.  The sharded exchange's
    payload follows the engine's precision (f32 under f32 / bf16x3 engines, bf16 under bf16 / bf16x3_fwd), an all-reduce is f32 unless the mode
    says _bf16 (parallel.exchange_payload): f32 payloads
    agree to f32 rounding (a bf16x3 engine's two-rank run then stays within 1e-6 of the single-rank run ON AVERAGE - round 3 shipped its
    gradients as bf16 and was 1e-5 off), bf16 payloads to the rounding of the exchanged gradients (2^-9 relative per element)."""
    import torch.multiprocessing as mp
    e, d = _make(_full_batch(), precision)
    for _ in range(2):
        _step(e, d, None, 1)
    ref = e.params.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + 7 * (_MODES.index(mode) * 8 + int(overlap) * 4 + _PRECS.index(precision))
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, overlap, precision, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res, lp, nbs = {}, {}, []
    try:
        for _ in range(2):
            r, params, params_lp, nb = q.get(timeout=240)
            res[r], lp[r] = params, params_lp
            nbs.append(nb)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert all(nb >= (3 if (overlap or mode.startswith("sharded")) else 1) for nb in nbs)
    assert np.array_equal(res[0], res[1])                      # replicas stay bit-identical (sharded: after gather_master)
    if precision != "f32":
        assert np.array_equal(lp[0], lp[1])                    # the forward's weight copy is identical without any gather
    from rgqa_amd.parallel import exchange_payload
    f32_payload = exchange_payload(mode, precision) == torch.float32
    exact = f32_payload and precision == "f32"
    diff = np.abs(res[0] - ref)
    print("dp %s/%s/%s: |params - single-rank| max %.3e mean %.3e" % (mode, overlap, precision, diff.max(), diff.mean()))
    if exact:
        np.testing.assert_allclose(res[0], ref, rtol=2e-4, atol=2e-6)
    elif f32_payload and precision == "bf16x3":        # f32 on the wire: only the split operands of the two batch splits round differently
        assert diff.max() < 1.3e-2 and diff.mean() < 1e-6, (diff.max(), diff.mean())
    elif f32_payload:        # bf16 / bf16x3_fwd engine under the (f32) all-reduce: the engine's own bf16 rounding under the other batch split
        assert diff.max() < 1.3e-2 and diff.mean() < 1e-5, (diff.max(), diff.mean())
    else:
        # bf16 payload: a gradient element moves by up to 2^-9 of itself.  BertAdam without bias correction moves an element by up to
        # lr * 0.1 / sqrt(0.001) = 3.2 lr per step whatever the gradient's size, so an element whose tiny gradient changes sign
        # between the two runs differs by up to 2 steps x 2 x 3.2e-3 = 1.3e-2; the bulk differs by ~1e-6 (f32 engine) / ~1e-5 (bf16
        # engine, whose activations are also rounded differently under the other batch split): the MEAN is the meaningful bound.
        assert diff.max() < 1.3e-2 and diff.mean() < (1e-5 if precision in ("bf16", "bf16x3_fwd", "bf16x3") else 4e-6)      # observed means: 1.6e-6 / 9.4e-7 (f32 engine), 2.3e-6 (bf16 engine;
        # 6.3e-5 while the non-owners' biases / LayerNorm parameters / embedding tables were stale - round 3 fix)


def _rccl_worker(port, mode, precision, q):
    import torch.distributed as dist
    from rgqa_amd.parallel import make_exchange
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    e, d = _make(_full_batch(), precision)
    kw = dict(bucket_mb=1) if not mode.startswith("sharded") else dict(chunk_mb=1, bucket_mb=1, f32_chunk_elems=1 << 14)
    comm = make_exchange(e, dist, mode, overlap=True, **kw)
    assert not getattr(comm, "_host_staged", False)            # device tensors straight into the library
    for _ in range(2):
        _step(e, d, comm, 1)
    comm.gather_master()
    torch.cuda.synchronize()
    q.put(e.params.cpu().numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,precision", [("sharded", "bf16"), ("sharded", "bf16x3"), ("sharded", "bf16x3_fwd"), ("allreduce_f32", "bf16"), ("allreduce", "bf16"), ("allreduce_bf16", "f32")])
def test_exchange_on_rccl_single_rank_group(mode, precision):
    """RCCL itself (torch.distributed backend "nccl"), as far as a one-GPU box allows: a process group of ONE rank - group creation with
    device_id, the bf16 all_to_all_single, the in-place all_gather_into_tensor, the packed f32 / scalar all_reduce, each on device tensors from
    the exchange's side stream - and the result against the plain single-process step (a bf16 payload rounds the gradients once)."""
    import torch.multiprocessing as mp
    e, d = _make(_full_batch(), precision)
    for _ in range(2):
        _step(e, d, None, 1)
    ref = e.params.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 11 * _MODES.index(mode) + 3 * _PRECS.index(precision)
    p = ctx.Process(target=_rccl_worker, args=(port, mode, precision, q), daemon=True)
    p.start()
    try:
        got = q.get(timeout=240)
        p.join(60)
    finally:
        if p.is_alive():
            p.terminate()
    assert p.exitcode == 0
    diff = np.abs(got - ref)
    print("rccl world-1 %s/%s: |params - plain step| max %.3e mean %.3e" % (mode, precision, diff.max(), diff.mean()))
    if mode == "allreduce_f32":
        np.testing.assert_array_equal(got, ref)                  # f32 payload, SUM over one rank: the identity
    elif mode == "sharded" and precision == "bf16x3":            # f32 payload; the clip norm is folded in another order (shard sums): last-bit differences
        assert diff.max() < 1e-4 and diff.mean() < 1e-7, (diff.max(), diff.mean())
    else:
        assert diff.max() < 1.3e-2 and diff.mean() < 1e-5


# ---------------------------------------------------------------------------------------------- the hand-written peer-to-peer exchange (hipIpc)
def _peer_worker(rank, world, port, backend, precision, overlap, q):
    """the SAME two steps under mode 'sharded' and under mode 'peer' (fresh engines): identical arithmetic - only the two collectives differ -
    so parameters, Adam moments and the forward's operand copy must agree bit for bit"""
    import torch.distributed as dist
    from rgqa_amd.parallel import make_exchange
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, **kw)
    full = _full_batch()
    n = 2 * B // world
    shard = {k: v[rank * n:(rank + 1) * n] for k, v in full.items()}
    out = {}
    for mode in ("sharded", "peer"):
        e, d = _make(shard, precision)
        comm = make_exchange(e, dist, mode, chunk_mb=1, bucket_mb=1, overlap=overlap, f32_chunk_elems=1 << 14)
        if mode == "peer":
            assert type(comm).__name__ == "PeerShardedExchange" and len(comm.chunks) >= 3
            assert comm.selfcheck()          # what bench.py's wire probe runs before it times this exchange
        for _ in range(3):
            _step(e, d, comm, world)
        lp = None if e.params_lp is None else e.params_lp.clone()
        comm.gather_master()
        comm.gather_master(e.adam_m)
        torch.cuda.synchronize()
        out[mode] = (e.params.clone(), e.adam_m.clone(), lp)
        if hasattr(comm, "close"):
            dist.barrier()
            comm.close()
    same = all((a is None and c is None) or torch.equal(a, c) for a, c in zip(out["sharded"], out["peer"]))
    q.put((rank, same, out["peer"][0].cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world,precision,overlap", [("gloo", 2, "bf16", True), ("gloo", 2, "f32", False), ("gloo", 2, "bf16x3_fwd", True), ("gloo", 4, "bf16", True), ("nccl", 1, "bf16", True)])
def test_peer_exchange_equals_the_collective_exchange(backend, world, precision, overlap):
    """VERDICT r5 #6 / SURVEY §8 B6: the sharded exchange with its all-to-all and all-gather hand-written over hipIpc peer buffers (rgqa_peer_*:
    every rank stages into ONE exported buffer and pulls its share out of every peer's buffer, workgroups dealt over the peers).  Two PROCESSES
    sharing this GPU map each other's staging buffer (IPC handles work intra-device; their barrier runs over gloo) - bf16 and f32 payloads, chunks
    beside backward on two streams and after it, the weight gather beside the next forward - and one rank on RCCL (the stream-ordered barrier is a
    tiny all-reduce there): three optimizer steps end bit-identical to mode 'sharded', replicas identical."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 400) + 5 * (_PRECS.index(precision) * 4 + (world - 1))
    procs = [ctx.Process(target=_peer_worker, args=(r, world, port, backend, precision, overlap, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, same, params = q.get(timeout=300)
            res[r] = (same, params)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    assert all(v[0] for v in res.values()), "mode 'peer' differs from mode 'sharded'"
    for r in range(1, world):
        assert np.array_equal(res[0][1], res[r][1])          # replicas identical (four ranks: every rank pulls from three peers)


def _peer_fail_worker(rank, world, port, q):
    import torch.distributed as dist
    from rgqa_amd.parallel import PeerShardedExchange, make_exchange
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    full = _full_batch()
    n = 2 * B // world
    shard = {k: v[rank * n:(rank + 1) * n] for k, v in full.items()}
    e, d = _make(shard, "bf16")
    if rank == 1:          # this rank cannot create its staging buffer
        class _Lib:
            def __init__(self, lib):
                self._lib = lib

            def __getattr__(self, k):
                if k == "rgqa_peer_comm_create":
                    return lambda *a: 1
                return getattr(self._lib, k)
        e.lib = _Lib(e.lib)
    msg = None
    try:
        PeerShardedExchange(e, dist, chunk_mb=1, bucket_mb=1, f32_chunk_elems=1 << 14)
    except RuntimeError as exn:
        msg = str(exn)
    if rank == 1:
        e.lib = e.lib._lib
    # the process group is still in step: the collective exchange works right after
    comm = make_exchange(e, dist, "sharded", chunk_mb=1, bucket_mb=1, f32_chunk_elems=1 << 14)
    _step(e, d, comm, world)
    comm.gather_master()
    torch.cuda.synchronize()
    q.put((rank, msg, bool(torch.isfinite(e.params).all())))
    dist.destroy_process_group()


def test_peer_exchange_setup_fails_on_every_rank_together():
    """A rank that cannot create (or map) its staging buffer must not leave its peers inside a collective it never joins: set-up gathers every rank's
    status with the handles and every rank raises the same error (bench.py's probe then falls back to the collective exchange on all ranks)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 400) + 97
    procs = [ctx.Process(target=_peer_fail_worker, args=(r, 2, port, q), daemon=True) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r, msg, finite = q.get(timeout=300)
            res[r] = (msg, finite)
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    for r in (0, 1):
        assert res[r][0] is not None and "rank 1 could not create" in res[r][0], res[r][0]
        assert res[r][1]


# ---------------------------------------------------------------------------------------------- BASELINE config 5 under the exchange
BUTD = dict(arch=1, vocab_size=201, hidden=1024, emb_dim=300, feat_dim=2048, pos_dim=4, num_answers=70, heads=1, inter=8, l_layers=0, x_layers=0, r_layers=0)
BB, BL, BO = 6, 14, 36


def _butd_batch():
    from rgqa_amd import synth
    b = synth.synth_batch(2 * BB, BL, O=BO, F=BUTD["feat_dim"], NA=BUTD["num_answers"], vocab=200, seed=77, min_len=2)
    rng = np.random.RandomState(8)
    toks = np.full((2 * BB, BL), 200, dtype=np.int64)            # dictionary indices, front-padded with the padding index (butd.py:180-193)
    for r in range(2 * BB):
        n = int(rng.randint(1, BL + 1))
        toks[r, BL - n:] = rng.randint(0, 200, size=n)
    return dict(feats=b["feats"], boxes=b["boxes"], target=b["target"], toks=toks)


def _make_butd(batch, precision):
    from rgqa_amd.engine import Engine
    from rgqa_amd import synth
    e = Engine(precision=precision, hidden_dropout=0.0, attn_dropout=0.0, **BUTD).allocate("cuda")
    for sp in e.specs:
        e.view(e.params, sp).copy_(torch.from_numpy(synth.fill_value(sp.name, sp.shape)))
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in batch.items()}
    e.ensure_shape(d["feats"].shape[0], BL, BO)
    e.sync_weights()
    return e, d


def _butd_step(e, d, comm, world):
    e.forward(d["feats"], d["boxes"], d["toks"], d["toks"], None, train=False)
    e.loss_backward(d["target"])
    if comm is not None:
        comm.exchange()
        comm.step(1e-3, max_norm=5.0)
    else:
        e.adam_step(1e-3, max_norm=5.0, grad_prescale=1.0 / world)


def _butd_worker(rank, world, port, backend, mode, precision, q):
    import torch.distributed as dist
    from rgqa_amd.parallel import make_exchange
    torch.cuda.set_device(0)
    kw = dict(device_id=torch.device("cuda", 0)) if backend == "nccl" else {}
    dist.init_process_group(backend, init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, **kw)
    full = _butd_batch()
    n = 2 * BB // world
    e, d = _make_butd({k: v[rank * n:(rank + 1) * n] for k, v in full.items()}, precision)
    comm = make_exchange(e, dist, mode, overlap=True, **(dict(chunk_mb=8, bucket_mb=8) if mode.startswith("sharded") else dict(bucket_mb=8)))
    assert not getattr(comm, "gather_overlap", False)        # this engine's forward waits for no weight event: the gather stays on the step's stream
    for _ in range(2):
        _butd_step(e, d, comm, world)
    comm.gather_master()
    torch.cuda.synchronize()
    q.put((rank, e.params.cpu().numpy()))
    dist.destroy_process_group()


@pytest.mark.parametrize("backend,world,mode,precision", [("nccl", 1, "sharded", "bf16"), ("nccl", 1, "allreduce", "bf16"), ("gloo", 2, "sharded", "bf16"),
                                                           ("gloo", 2, "sharded", "f32"), ("gloo", 2, "allreduce", "bf16x3")])
def test_butd_engine_under_the_exchange(backend, world, mode, precision):
    """BASELINE config 5 is a data-parallel run (8 GPUs): the gradient exchange around the BUTD engine (rgqa_config.arch = 1) - ONE gradient
    segment, its event recorded by the last launch of backward; no per-segment weight events (the engine re-derives its effective weights from
    the masters at the start of every pass), so the sharded mode's weight gather stays on the step's stream.  One rank on RCCL, two ranks on
    gloo: two optimizer steps == the plain step on the whole batch (round 5: until now the exchange raised at the first step - the event had
    never been recorded)."""
    import torch.multiprocessing as mp
    e, d = _make_butd(_butd_batch(), precision)
    for _ in range(2):
        _butd_step(e, d, None, 1)
    ref = e.params.cpu().numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + (os.getpid() % 500) + 13 * (["sharded", "allreduce"].index(mode) * 8 + _PRECS.index(precision) * 2 + (world - 1))
    procs = [ctx.Process(target=_butd_worker, args=(r, world, port, backend, mode, precision, q), daemon=True) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(world):
            r, params = q.get(timeout=240)
            res[r] = params
        for p in procs:
            p.join(60)
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
    assert all(p.exitcode == 0 for p in procs)
    if world == 2:
        assert np.array_equal(res[0], res[1])
    diff = np.abs(res[0] - ref)
    print("butd dp %s x%d %s/%s: |params - plain step| max %.3e mean %.3e" % (backend, world, mode, precision, diff.max(), diff.mean()))
    from rgqa_amd.parallel import exchange_payload
    if exchange_payload(mode, precision) == torch.float32 and precision == "f32":
        np.testing.assert_allclose(res[0], ref, rtol=2e-4, atol=2e-6)
    else:
        assert diff.max() < 1.3e-2 and diff.mean() < 1e-5, (diff.max(), diff.mean())
