"""Data parallelism: one process per GPU, the gradient arena exchanged over RCCL/xGMI (torch.distributed backend "nccl" == RCCL
on ROCm; "gloo" in the CPU / single-GPU tests).

Replaces the reference's single-process nn.DataParallel (lxrt/entry.py:102-103): samples are independent through the whole
forward, so each rank runs its own shard of the batch and the only exchange per step is the gradient arena.
THE SHARDED MODE'S PAYLOAD FOLLOWS THE ENGINE'S PRECISION (payload_dtype): bf16 under `bf16` and `bf16x3_fwd` engines, whose gradients carry
bf16 rounding anyway (410 MB); f32 under `f32` and `bf16x3` engines, whose gradients are exact to 1e-5 and must not be rounded to 8 bits on
the wire (819 MB) - the owner adds the N shards in f32 in rank order, so a bf16 payload costs one rounding per contribution and no more.  An
ALL-REDUCE keeps its running sum in the payload type, so plain `allreduce` is f32 for every engine (what the reference's nn.DataParallel
reduces); `allreduce_bf16` opts in.  The suffixes _bf16 / _f32 force a payload in either mode.  Modes (`make_exchange`, env RGQA_DP_MODE; RGQA_DP_OVERLAP=0 issues any of them
after backward instead of beside it):

  allreduce        SUM all-reduce of the live arena ranges, buckets of >= 64 MB issued from a side stream as backward finalises them;
                   every rank then clips and runs BertAdam over the whole arena (round 1).  The drop-in modules' default: the
                   unchanged trainer owns the optimizer object, which steps every parameter.
  sharded          (bench.py's default) reduce-scatter as ONE all-to-all per chunk: rank r receives every rank's copy
                   of the 1/N range it owns and accumulates them in f32 in rank order (rgqa_sum_parts: deterministic, no
                   low-precision running sum) - the same pass leaves the owner's share of sum(g^2) behind, so the clip norm costs no second
                   read; clip + BertAdam then touch only that 1/N (the shares meet in one scalar
                   all-reduce), and the updated weights are all-gathered into every rank's forward copy (bf16 engines: the bf16
                   copy, plus - in f32 - what the forward reads from the masters: biases, LayerNorm parameters, embedding tables;
                   f32 / bf16x3 engines: the f32 masters).  The chunks are the gradient segments backward finalises (merged
                   to >= 64 MB), each exchanged on a side stream as soon as its event fires, so only the last one is exposed.  On
                   the 8-GPU xGMI mesh an all-to-all uses all 7 links of a GPU at once, the wire carries 2 x 7/8 x 410 MB per GPU
                   per step instead of 2 x 7/8 x 819 MB, and the optimizer's 6 GB of HBM traffic shrinks 8x.  While this mode is
                   active the f32 master copy and the Adam moments of a range are current only on its owner: the engine refuses
                   `adam_step` (Engine._sharded_owner), `gather_master()` refreshes the masters (checkpoints, state_dict) and
                   `release()` gathers masters + moments and hands the optimizer back to the engine.

The dead range (x_layers.<last>.visn_*: never receives gradients in mode 'x') is in no bucket.  The exchange entry points
return immediately; ordering is by streams and events, never by host synchronisation."""
import ctypes as C
import os

import torch

from ._lib import check, ptr


def payload_dtype(precision, forced=None):
    """what goes on the wire: bf16 for engines whose gradients carry bf16 rounding anyway, f32 for the others; forced = 'bf16' / 'f32'"""
    if forced == "bf16":
        return torch.bfloat16
    if forced == "f32":
        return torch.float32
    return torch.bfloat16 if precision in ("bf16", "bf16x3_fwd") else torch.float32


def exchange_payload(mode, precision):
    """dtype on the wire for a full RGQA_DP_MODE string: the sharded exchange follows the engine's precision, an all-reduce is f32 unless the
    mode says _bf16 (its running sum lives in the payload type); a suffix forces either"""
    kind, _, forced = mode.partition("_")
    if kind in ("sharded", "peer"):
        return payload_dtype(precision, forced or None)
    return torch.bfloat16 if forced == "bf16" else torch.float32


def bucket_ranges(ranges, bucket_elems):
    """Split [(a,b), ...] element ranges into consecutive buckets of at most bucket_elems elements."""
    out = []
    for a, b in ranges:
        while a < b:
            n = min(bucket_elems, b - a)
            out.append((a, a + n))
            a += n
    return out


def merge_segments(segs, min_elems):
    """segs: [(begin, end, event)] in completion order. Grows buckets downwards in the arena (layers complete from the
    last to the first) until they hold >= min_elems elements. Returns [(begin, end, event_to_wait)] in flush order."""
    open_by_begin, done = {}, []
    for b, e, ev in segs:
        cur = open_by_begin.pop(e, None)        # an open bucket starting exactly where this segment ends
        if cur is not None:
            cur = (b, cur[1], max(ev, cur[2]))
        else:
            cur = (b, e, ev)
        if cur[1] - cur[0] >= min_elems:
            done.append(cur)
        else:
            open_by_begin[cur[0]] = cur
    done.extend(sorted(open_by_begin.values(), key=lambda c: c[2]))
    return sorted(done, key=lambda c: c[2])


def shard_layout(ranges, world, chunk_elems, align=8):
    """The sharded mode's partition of the live arena: every (a, b) range is cut into chunks of at most chunk_elems elements,
    every chunk into `world` parts of S = ceil(len / world / align) * align elements (the last parts of a ragged chunk are
    short or empty).  Returns [(a, b, S)]; rank r owns [a + r*S, min(a + (r+1)*S, b)) of each chunk."""
    out = []
    for a, b in bucket_ranges(ranges, chunk_elems):
        n = b - a
        s = -(-n // world)
        s = -(-s // align) * align
        out.append((a, b, s))
    return out


def owned(chunk, rank):
    a, b, s = chunk
    lo = min(a + rank * s, b)
    return lo, min(lo + s, b)


class _HipOps:
    """The exchange's local arithmetic on the engine's device, through the C ABI."""

    def __init__(self, lib):
        self.lib = lib

    @staticmethod
    def _s():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def cast_bf16(self, dst, src):
        check(self.lib.rgqa_cast_bf16(ptr(src), ptr(dst), src.numel(), self._s()))

    def split_f32(self, dst, src):
        check(self.lib.rgqa_split_f32(ptr(src), ptr(dst), src.numel(), self._s()))

    def sum_parts(self, dst, parts, stride, nparts, sq_ws=None, sq_out=None):
        check(self.lib.rgqa_sum_parts(ptr(parts), 1 if parts.dtype == torch.float32 else 0, stride, nparts, ptr(dst), dst.numel(),
                                      ptr(sq_ws) if sq_out is not None else None, ptr(sq_out), self._s()))


_DP_STREAMS = {}


def _engine_stream(engine, k, device):
    """the k-th exchange stream of an engine: ONE set per engine, shared by every exchange object made for it (bench.py makes three in a run:
    the mode under test, its counterpart, the fallback) - streams are mapped onto a handful of hardware queues in creation order, and every
    further stream is one more chance to share a queue with the launch stream (csrc/engine.hip make_side_stream)"""
    pool = _DP_STREAMS.setdefault((device.type, device.index), {})
    if k not in pool:
        # stream 0 IS the engine's update stream (the optimizer pass beside the forward; made when the engine was bound): the exchange uses it
        # during backward and for the weight gather, the update runs between the two - one stream fewer to place
        upd = getattr(engine, "_upd_stream", None) if k == 0 else None
        if upd is None and device.type == "cuda":
            from . import streams
            upd = streams.pick(device, 3)[1 if k == 0 else 2]      # picked to run beside the caller's stream and beside each other
        pool[k] = upd if upd is not None else torch.cuda.Stream(device=device)
    return pool[k]


class GradAllReduce:
    """modes 'allreduce' / 'allreduce_bf16'"""

    def release(self):
        pass

    def __init__(self, engine, dist, bucket_mb=64, overlap=None, bf16=None, ops=None):
        if bf16 is None:           # f32 on the wire, as the reference's nn.DataParallel reduces f32 gradients; bf16 is opt-in (allreduce_bf16)
            bf16 = False
        self.e, self.dist, self.bf16 = engine, dist, bf16
        self.ops = ops if ops is not None else (_HipOps(engine.lib) if bf16 else None)
        if overlap is None:
            overlap = os.environ.get("RGQA_DP_OVERLAP", "1") != "0"
        self.overlap = overlap and hasattr(engine, "grad_segments")
        if self.overlap:
            self.buckets = merge_segments(engine.grad_segments(), bucket_mb * (1 << 20) // 4)
        else:
            self.buckets = [(a, b, -1) for a, b in bucket_ranges(engine.live_ranges(), 256 * (1 << 20) // 4)]
        self.side = None
        self._stage = None

    def describe(self):
        return "%s, %d buckets, %s" % ("allreduce_bf16" if self.bf16 else "allreduce", len(self.buckets), "overlapped with backward" if self.overlap else "after backward")

    def _reduce_bucket(self, g, a, b):
        dist = self.dist
        if not self.bf16:
            return dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True), None
        st = self._stage[a:b]
        self.ops.cast_bf16(st, g[a:b])
        return dist.all_reduce(st, op=dist.ReduceOp.SUM, async_op=True), (a, b)

    def all_reduce(self, grads=None):
        """Call right after the engine's backward returned (its work is enqueued, not necessarily finished)."""
        g = self.e.grads if grads is None else grads
        if hasattr(self.e, "invalidate_segment_sumsq"):
            self.e.invalidate_segment_sumsq()          # the norm must come from the REDUCED gradients
        if self.bf16 and self._stage is None:
            self._stage = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
        hs = []
        if not self.overlap or not g.is_cuda:
            hs = [self._reduce_bucket(g, a, b) for a, b, _ in self.buckets]
        else:
            if self.side is None:
                self.side = _engine_stream(self.e, 0, g.device)
            for a, b, ev in self.buckets:
                with torch.cuda.stream(self.side):
                    self.e.wait_grad_event(ev, self.side)
                    hs.append(self._reduce_bucket(g, a, b))
        for h, back in hs:
            h.wait()        # the CURRENT stream waits for the collective; no host synchronisation
            if back is not None:
                g[back[0]:back[1]].copy_(self._stage[back[0]:back[1]])

    exchange = all_reduce

    def step(self, lr_t, **kw):
        """clip + BertAdam over the whole (identical) arena on every rank"""
        self.e.adam_step(lr_t, grad_prescale=1.0 / self.dist.get_world_size(), **kw)

    def gather_master(self):
        pass


class ShardedExchange:
    """mode 'sharded' (module docstring).  exchange(): bf16 all-to-all reduce-scatter with f32 accumulation at the owner;
    step(): sharded clip + BertAdam, bf16 weight all-gather, transposed-copy refresh."""

    def __init__(self, engine, dist, chunk_mb=256, bucket_mb=64, overlap=None, ops=None, f32_chunk_elems=1 << 18, payload=None):
        self.e, self.dist = engine, dist
        self.payload = payload_dtype(getattr(engine, "precision", "f32"), payload)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.ops = ops if ops is not None else _HipOps(engine.lib)
        if overlap is None:
            overlap = os.environ.get("RGQA_DP_OVERLAP", "1") != "0"
        self.overlap = bool(overlap) and hasattr(engine, "grad_segments") and hasattr(engine, "wait_grad_event")
        if self.overlap:       # chunks = the gradient segments in the order backward finalises them, each with the event to wait for
            self.chunks, self.events = [], []
            for a, b, ev in merge_segments(engine.grad_segments(), bucket_mb * (1 << 20) // 4):
                for c in shard_layout([(a, b)], self.world, chunk_mb * (1 << 20) // 2):
                    self.chunks.append(c)
                    self.events.append(ev)
        else:
            self.chunks = shard_layout(engine.live_ranges(), self.world, chunk_mb * (1 << 20) // 2)
            self.events = [-1] * len(self.chunks)
        self.smax = max(s for _, _, s in self.chunks)
        # gradient-segment ids whose weights each chunk holds (the engine's forward waits per segment: step() below)
        segs = engine.grad_segments() if hasattr(engine, "grad_segments") else []
        self.chunk_segs = [sorted({ev for sb, se, ev in segs if sb < b and se > a}) for a, b, _ in self.chunks]
        dev = engine.grads.device
        self._send = torch.zeros(self.world * self.smax, dtype=self.payload, device=dev)
        self._recv = torch.zeros(self.world * self.smax, dtype=self.payload, device=dev)
        self._stage = [(self._send, self._recv)]       # one staging set per exchange stream
        self.nstreams = 2 if os.environ.get("RGQA_DP_EXCHANGE_STREAMS", "2") != "1" else 1
        self.side2 = self._sumsq2 = self._sqws2 = None
        # Everything the exchange streams touch is made HERE, not on first use inside exchange(): a torch.zeros() there put its fill kernel on
        # the caller's stream BEHIND the whole backward pass while a side stream - which waits for a gradient-segment event in the middle of
        # backward only - was already casting into / summing from the buffer: the fill could land between the cast and the shard sum of a chunk in
        # flight.  `_alloc_ready` orders the side streams' first use behind these fills.
        self._sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        self._sqws = torch.zeros(2048, dtype=torch.float32, device=dev)
        if self.overlap and self.nstreams > 1 and len(self.chunks) > 1:
            self._stage.append((torch.zeros_like(self._send), torch.zeros_like(self._recv)))
            self._sumsq2 = torch.zeros(1, dtype=torch.float32, device=dev)
            self._sqws2 = torch.zeros(2048, dtype=torch.float32, device=dev)
        self._alloc_ready = None
        if dev.type == "cuda":
            self._alloc_ready = torch.cuda.Event()
            self._alloc_ready.record(torch.cuda.current_stream(dev))
        self._sumsq_from_exchange = False
        self._norm_reduced = False       # _sumsq already holds the sum over ranks (global_sumsq(): the drop-in clip_grad_norm_ asks before step())
        self._norm_read = None
        self.side = None
        backend = dist.get_backend()
        self._host_staged = backend != "nccl"      # rehearsal on one device over gloo: collectives run on host copies
        # what is all-gathered after the update: the bf16 operand copy (bf16 engines), else the f32 masters (a split-f32 copy cannot be
        # cut at arbitrary element offsets: a 128-byte line holds the hi and the lo parts of 32 elements; it is re-made from the masters)
        self.lp = engine.precision == "bf16"
        # The weight all-gather beside the NEXT forward pass: chunks gathered on the side stream in FORWARD order (embeddings first,
        # head last), one event per chunk; the engine's forward waits, per layer, for the event of the chunk that holds the layer's weights, its
        # backward for the transposed copies refreshed behind the last chunk.  bf16 engines on a real device only (a split-f32 operand copy is
        # re-made from the gathered masters as a whole); RGQA_DP_GATHER_OVERLAP=0 keeps the gather on the step's stream.
        # ... and only around an engine whose forward waits per segment (the BUTD engine re-derives its effective weights from the masters at the
        # start of every pass: it waits for nothing, its gather stays on the step's stream)
        # round 6: the split-f32 engines too (bf16x3_fwd is bench.py's headline) - their chunks gather the f32 masters and re-make the split operand copy of
        # the chunk's own range behind it (chunks begin and end on tensor boundaries: 64-element aligned, whole 128-byte lines of the split layout)
        self.x3 = engine.precision in ("bf16x3", "bf16x3_fwd") and getattr(engine, "_params_lp", None) is not None
        split_ok = self.x3 and all(a % 32 == 0 and b % 32 == 0 for a, b, _ in self.chunks)
        self.gather_overlap = (self.overlap and (self.lp or split_ok) and dev.type == "cuda" and hasattr(engine, "set_weight_event")
                               and hasattr(engine, "num_weight_segments") and engine.num_weight_segments() > 0
                               and os.environ.get("RGQA_DP_GATHER_OVERLAP", "1") != "0")
        self._wevents = []
        # bf16 engines: the forward reads biases, LayerNorm parameters, the embedding tables and the K = 4 box projection from the F32 masters
        # (ParamSpec.f32_read), which the bf16 all-gather does not carry.  Chunks that hold a large such tensor (the embedding tables) are
        # gathered in f32 and their bf16 copy re-made from the result; the small tensors elsewhere travel as one packed f32 all-reduce in which
        # every rank contributes the elements it owns and zeros for the rest.
        self.f32_chunks, self._small_idx, self._small_own = set(), None, None
        if self.lp and hasattr(engine, "specs"):
            live = [sp for sp in engine.specs if getattr(sp, "f32_read", False) and not sp.dead]
            big = [sp for sp in live if sp.numel >= f32_chunk_elems]
            for ci, (a, b, _) in enumerate(self.chunks):
                if any(sp.offset < b and sp.offset + sp.numel > a for sp in big):
                    self.f32_chunks.add(ci)
            idx = []
            for sp in live:
                lo, hi = sp.offset, sp.offset + sp.numel
                covered = any(self.chunks[ci][0] <= lo and hi <= self.chunks[ci][1] for ci in self.f32_chunks)
                if not covered:
                    idx.append(torch.arange(lo, hi, dtype=torch.int64))
            if idx:
                idx = torch.cat(idx)
                own = torch.zeros(idx.numel(), dtype=torch.bool)
                for c in self.chunks:
                    lo, hi = owned(c, self.rank)
                    if hi > lo:
                        own |= (idx >= lo) & (idx < hi)
                # an element outside every chunk (no gradient segment covers it) is never updated: nobody owns it, it keeps its value
                in_chunk = torch.zeros(idx.numel(), dtype=torch.bool)
                for a, b, _ in self.chunks:
                    in_chunk |= (idx >= a) & (idx < b)
                self._small_idx = idx[in_chunk].to(dev)
                self._small_own = own[in_chunk].to(dev)

    def describe(self):
        return "sharded: %s all-to-all reduce-scatter (clip-norm share taken while summing) + sharded BertAdam + %s weight all-gather, %d chunk(s) of <= %d MB, %s" % (
            "bf16" if self.payload == torch.bfloat16 else "f32", "bf16" if self.lp else "f32", len(self.chunks), self.smax * self.world * self._send.element_size() >> 20,
            "overlapped with backward" if self.overlap else "after backward")

    # -- collectives (RCCL on device tensors; host-staged under gloo, where device tensors are not supported by every op)
    def _a2a(self, recv, send):
        if self._host_staged and send.is_cuda:
            r, s = torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
            self.dist.all_to_all_single(r, s)
            recv.copy_(r)
        else:
            self.dist.all_to_all_single(recv, send)

    def _ag(self, out, inp):
        if self._host_staged and inp.is_cuda:
            o = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_gather_into_tensor(o, inp.cpu())
            out.copy_(o)
        else:
            self.dist.all_gather_into_tensor(out, inp)

    def exchange(self, grads=None):
        """After backward: leaves, in the gradient arena, the SUM over ranks of the ranges this rank owns (other ranges keep the
        local gradients and are not read again)."""
        g = self.e.grads if grads is None else grads
        W = self.world
        if hasattr(self.e, "invalidate_segment_sumsq"):
            self.e.invalidate_segment_sumsq()

        def one(a, b, s, k=0):
            n = b - a
            send, recv = self._stage[k][0][:W * s], self._stage[k][1][:W * s]
            if self.payload == torch.bfloat16:
                self.ops.cast_bf16(send[:n], g[a:b])        # part r of the chunk at send[r*s : (r+1)*s]; the ragged tail is never read
            elif n == W * s:
                send = g[a:b]                               # f32 payload, whole parts: straight from the gradient arena
            else:
                send[:n].copy_(g[a:b])
            self._a2a(recv, send)
            lo, hi = owned((a, b, s), self.rank)
            if hi > lo:
                self.ops.sum_parts(g[lo:hi], recv, s, W, self._sqws if k == 0 else self._sqws2, self._sumsq if k == 0 else self._sumsq2)    # f32 accumulation in rank order + this range's sum(g^2)

        self._sumsq_from_exchange = True
        self._norm_reduced = False
        if not (self.overlap and g.is_cuda):
            self._sumsq.zero_()
            for c in self.chunks:
                one(*c)
            return
        # Chunks alternate between TWO side streams, each with a staging set and a norm accumulator of its own, each chunk behind the event of the
        # gradient segment it holds: the exchange of the layers backward has finished runs beside the layers it is still computing, and a chunk's
        # cast and shard sum run beside the neighbouring chunk's all-to-all (the weight gradients of a step come from six grouped launches, so
        # segments become final in bursts of several chunks - the last bursts are what backward cannot hide; round 5: one stream walked them
        # cast -> wire -> sum, chunk after chunk).  RGQA_DP_EXCHANGE_STREAMS=1 restores the single stream.
        if self.side is None:
            self.side = _engine_stream(self.e, 0, g.device)
        two = len(self._stage) > 1
        if two and self.side2 is None:
            self.side2 = _engine_stream(self.e, 1, g.device)
        cur = torch.cuda.current_stream()
        streams = [self.side, self.side2] if two else [self.side]
        for k, st in enumerate(streams):
            if self._alloc_ready is not None:
                st.wait_event(self._alloc_ready)        # the staging buffers' and accumulators' fills (made on the constructor's stream)
            if self._norm_read is not None:
                st.wait_event(self._norm_read)          # the previous step's optimizer has read the norm before it is cleared
            with torch.cuda.stream(st):
                (self._sumsq if k == 0 else self._sumsq2).zero_()
        for i, (c, ev) in enumerate(zip(self.chunks, self.events)):
            k = i % len(streams)
            with torch.cuda.stream(streams[k]):
                self.e.wait_grad_event(ev, streams[k])
                one(*c, k=k)
        for st in streams:
            cur.wait_stream(st)
        if two:
            self._sumsq += self._sumsq2                 # (fixed chunk -> stream assignment: the norm is the same number every run)

    all_reduce = exchange

    def global_sumsq(self):
        """The squared norm of the REDUCED gradients on every rank (device scalar): the owners' shares of sum(g^2) - left behind by exchange(), or taken
        now over the owned ranges - summed over the ranks by one scalar all-reduce.  What the unchanged trainer's `clip_grad_norm_` returns between
        backward() and BertAdam.step() (tasks/gqa_conf.py:201); step(clip=True) then reuses it instead of reducing again."""
        if self._norm_reduced:
            return self._sumsq
        dev = self.e.grads.device
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev.type == "cuda" else None
        if not self._sumsq_from_exchange:
            self._sumsq.zero_()
            for c in self.chunks:
                lo, hi = owned(c, self.rank)
                if hi > lo:
                    self._local_sumsq(lo, hi, s)
            self._sumsq_from_exchange = True
        if self._host_staged and self._sumsq.is_cuda:
            t = self._sumsq.cpu()
            self.dist.all_reduce(t)
            self._sumsq.copy_(t)
        else:
            self.dist.all_reduce(self._sumsq)
        self._norm_reduced = True
        return self._sumsq

    def step(self, lr_t, max_norm=5.0, b1=0.9, b2=0.999, eps=1e-6, weight_decay=0.01, clip=True, grad_prescale=None):
        """grad_prescale: factor on the exchanged gradient SUMS inside the update kernel; default 1 / world (the engine-direct loop backpropagates the
        local mean loss); the drop-in autograd path has already scaled dL/dlogits by 1 / world and passes 1."""
        e, W = self.e, self.world
        dev = e.grads.device
        if hasattr(e, "_sharded_owner"):
            if e._sharded_owner is not None and e._sharded_owner is not self:
                raise RuntimeError("ShardedExchange.step: another sharded exchange owns this engine's optimizer state; release() it first")
            e._sharded_owner = self
        if e.adam_m is None:
            e.adam_m = torch.zeros_like(e.params)        # only the owned ranges are ever touched (288 GB HBM: no need to compact)
            e.adam_v = torch.zeros_like(e.params)
        mine = [owned(c, self.rank) for c in self.chunks]
        mine = [(lo, hi) for lo, hi in mine if hi > lo]
        s = C.c_void_p(torch.cuda.current_stream().cuda_stream) if dev.type == "cuda" else None
        if clip:
            self.global_sumsq()                      # sum over ranks of the shards' sum(g^2) = the global norm^2 (already there if clip_grad_norm_ asked)
        self._sumsq_from_exchange = self._norm_reduced = False      # consumed by THIS step whatever `clip` says (a stale flag would make a later step
        pre = 1.0 / W if grad_prescale is None else float(grad_prescale)                   # reuse an old norm: ADVICE r4)
        for lo, hi in mine:
            self._local_adam(lo, hi, lr_t, b1, b2, eps, weight_decay, clip, max_norm, pre, s)
        if dev.type == "cuda":
            if self._norm_read is None:
                self._norm_read = torch.cuda.Event()
            self._norm_read.record()
        # updated weights -> every rank's forward copy
        if self.gather_overlap:
            return self._gather_beside_forward(s)
        for ci in range(len(self.chunks)):
            self._gather_chunk(ci)
        if self._small_idx is not None and self._small_idx.numel():
            pack = e.params[self._small_idx] * self._small_own
            if self._host_staged and pack.is_cuda:
                t = pack.cpu()
                self.dist.all_reduce(t)
                pack.copy_(t)
            else:
                self.dist.all_reduce(pack)                  # exactly one rank contributes a non-zero to each element: the sum IS the owner's value
            e.params[self._small_idx] = pack
        self._after_weights(s)

    def _gather_chunk(self, ci):
        """all-gather of chunk ci's updated weights into this rank's forward copy, on the current stream"""
        e, W, dev = self.e, self.world, self.e.grads.device
        a, b, sz = self.chunks[ci]
        f32c = ci in self.f32_chunks                  # bf16 engine, chunk with a large f32-read tensor: gather the masters, re-make the copy
        wts = e.params_lp if (self.lp and not f32c) else e.params
        staged = self.lp and not f32c                  # the bf16 staging buffers fit a bf16 chunk only
        n = b - a
        lo, hi = owned((a, b, sz), self.rank)
        if n == W * sz:
            self._ag(wts[a:b], wts[lo:hi])            # in place: part r of the output is this rank's own input
        else:                                           # ragged chunk: through a padded buffer
            buf = self._recv[:W * sz] if staged else torch.empty(W * sz, dtype=wts.dtype, device=dev)
            mine_pad = self._send[:sz] if staged else torch.zeros(sz, dtype=wts.dtype, device=dev)
            if hi > lo:
                mine_pad[:hi - lo].copy_(wts[lo:hi])
            self._ag(buf, mine_pad)
            wts[a:b].copy_(buf[:n])
        if f32c:
            self.ops.cast_bf16(e.params_lp[a:b], e.params[a:b])
        elif self.x3 and self.gather_overlap:      # the chunk's split-f32 operand copy from the gathered masters (the non-overlapped path re-makes the whole copy at the end)
            self.ops.split_f32(e.params_lp[a:b], e.params[a:b])

    def _gather_beside_forward(self, s):
        """step()'s tail with the gather off the critical path.  On the caller's stream only what EVERY layer of the next forward reads from the
        f32 masters (biases, LayerNorm parameters: one small packed all-reduce); then, on the side stream and behind the optimizer, the chunks in
        forward order - the last chunk backward finished (embeddings, first layers) first, the head's last - each followed by an event the
        engine's forward waits for where it first reads that chunk's weights, and the transposed copies (backward's operand) at the very end."""
        e = self.e
        cur = torch.cuda.current_stream()
        if self._small_idx is not None and self._small_idx.numel():
            pack = e.params[self._small_idx] * self._small_own
            if self._host_staged and pack.is_cuda:
                t = pack.cpu()
                self.dist.all_reduce(t)
                pack.copy_(t)
            else:
                self.dist.all_reduce(pack)
            e.params[self._small_idx] = pack
        if self.side is None:
            self.side = _engine_stream(e, 0, e.grads.device)
        updated = torch.cuda.Event()
        updated.record(cur)
        evs = []
        with torch.cuda.stream(self.side):
            self.side.wait_event(updated)
            for ci in reversed(range(len(self.chunks))):          # chunks are in backward's completion order: forward reads them back to front
                self._gather_chunk(ci)
                ev = torch.cuda.Event()
                ev.record(self.side)
                evs.append(ev)
                for seg in self.chunk_segs[ci]:
                    e.set_weight_event(seg, ev)
            check(e.lib.rgqa_engine_sync_transposed(e.h, C.c_void_p(self.side.cuda_stream)))
            ev_t = torch.cuda.Event()
            ev_t.record(self.side)
            evs.append(ev_t)
            e.set_backward_event(ev_t)
        self._wevents = evs            # alive until the passes that wait for them have been enqueued (the next step() replaces them)
        self._gather_done = ev_t
        # a caller that reads the weights outside the engine (state_dict, a checkpoint, another exchange object) first joins the side stream:
        # gather_master() / release() below do; the engine's own passes wait per segment

    def _join_gather(self):
        ev = getattr(self, "_gather_done", None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self._gather_done = None

    # -- local arithmetic (HIP, through the C ABI)
    def _after_weights(self, s):
        if self.lp:
            check(self.e.lib.rgqa_engine_sync_transposed(self.e.h, s))     # dgrad operand: transposed bf16 copies, from the gathered bf16 arena
        elif getattr(self.e, "params_lp", None) is not None:
            check(self.e.lib.rgqa_engine_sync_weights(self.e.h, s))        # bf16x3: both split-f32 copies re-made from the gathered f32 masters

    def _local_sumsq(self, lo, hi, s):
        e = self.e
        check(e.lib.rgqa_grad_sumsq(ptr(e.grads[lo:hi]), hi - lo, ptr(self._sqws), ptr(self._sumsq), 1, s))

    def _local_adam(self, lo, hi, lr_t, b1, b2, eps, wd, clip, max_norm, prescale, s):
        e = self.e
        lp = ptr(e.params_lp[lo:hi]) if self.lp else None
        check(e.lib.rgqa_bertadam_step(ptr(e.params[lo:hi]), ptr(e.grads[lo:hi]), ptr(e.adam_m[lo:hi]), ptr(e.adam_v[lo:hi]), lp, 0, hi - lo,
                                       lr_t, b1, b2, eps, wd, ptr(self._sumsq) if clip else None, max_norm, prescale, s))

    def release(self):
        """Gathers the f32 masters AND the Adam moments to every rank and hands the optimizer back to the engine (Engine.adam_step works
        again; a later step() of this exchange re-shards without loss: every rank then holds the full state)."""
        self.gather_master()
        if self.e.adam_m is not None:
            self.gather_master(self.e.adam_m)
            self.gather_master(self.e.adam_v)
        if hasattr(self.e, "_sharded_owner"):
            self.e._sharded_owner = None

    def gather_master(self, arena=None):
        """All-gathers the f32 master weights (each range is current only on its owner): before state_dict() / checkpoints."""
        self._join_gather()
        p = self.e.params if arena is None else arena
        for a, b, sz in self.chunks:
            n = b - a
            lo, hi = owned((a, b, sz), self.rank)
            if n == self.world * sz:
                self._ag(p[a:b], p[lo:hi])
            else:
                buf = torch.empty(self.world * sz, dtype=p.dtype, device=p.device)
                mine_pad = torch.zeros(sz, dtype=p.dtype, device=p.device)
                if hi > lo:
                    mine_pad[:hi - lo].copy_(p[lo:hi])
                self._ag(buf, mine_pad)
                p[a:b].copy_(buf[:n])


class _RawDeviceBytes:
    """library-owned device memory as seen by torch (the __cuda_array_interface__ protocol): torch.as_tensor(...) aliases it, nothing is copied"""

    def __init__(self, ptr_, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr_), False), "version": 2}


class PeerShardedExchange(ShardedExchange):
    """mode 'peer': the sharded exchange with its two collectives hand-written over hipIpc peer buffers (include/rgqa.h rgqa_peer_*, csrc/peer.hip) -
    the fallback SURVEY §5 / §8 B6 planned for a node where RCCL's all-to-all / all-gather do not drive all seven xGMI links of a GPU (bench.py's
    `dp_wire` probe measures both and picks).  Every rank stages what the others need in ONE library-owned, IPC-exported buffer and PULLS its share out of
    every peer's buffer with one launch whose workgroups are dealt over the peers (all links at once); everything else - chunks beside backward on two
    streams, f32 accumulation of the parts in rank order with the norm share, sharded BertAdam, the weight gather beside the next forward - is the
    parent's, so the results are bit-identical to mode 'sharded'.  Ranks are ordered by a stream-ordered barrier (one tiny collective of the process
    group on the exchange's stream) between a rank's staging kernel and its peers' pulls, and again before the staging area is overwritten: no kernel
    ever spins on a flag."""

    def __init__(self, engine, dist, **kw):
        super().__init__(engine, dist, **kw)
        lib = engine.lib
        W, es = self.world, self._send.element_size()
        self._send_bytes = -(-W * self.smax * es // 256) * 256
        self._ag_bytes = -(-self.smax * 4 // 256) * 256                  # one owned part, f32 at worst
        self._set_bytes = self._send_bytes + self._ag_bytes
        nsets = 2
        # Set-up is collective: a rank that cannot create, export or map a buffer still takes part in both object gathers below and every rank
        # raises together - a rank that left early would leave its peers inside a collective it never joins
        self._comm = None
        dev = engine.grads.device
        hb = C.create_string_buffer(64)
        err = None
        try:
            h = C.c_void_p()
            check(lib.rgqa_peer_comm_create(self.rank, W, nsets * self._set_bytes, C.byref(h)))
            self._comm = h
            p, nb = C.c_void_p(), C.c_size_t()
            check(lib.rgqa_peer_comm_stage(self._comm, C.byref(p), C.byref(nb)))
            self._stage_raw = torch.as_tensor(_RawDeviceBytes(p.value, nb.value), device=dev)
            self._stage_raw.zero_()
            check(lib.rgqa_peer_comm_export(self._comm, hb))
        except Exception as exn:
            err = repr(exn)
        handles = [None] * W
        dist.all_gather_object(handles, (err, bytes(hb.raw)))
        self._raise_together([h_[0] for h_ in handles], "create / export its staging buffer")
        try:
            check(lib.rgqa_peer_comm_connect(self._comm, C.create_string_buffer(b"".join(h_[1] for h_ in handles), 64 * W)))
        except Exception as exn:
            err = repr(exn)
        mapped = [None] * W
        dist.all_gather_object(mapped, err)
        self._raise_together(mapped, "map its peers' staging buffers")
        # the parent's staging sets, their send halves now inside the exported buffer
        sets = []
        for k in range(nsets):
            send = self._stage_raw[k * self._set_bytes:k * self._set_bytes + W * self.smax * es].view(self.payload)
            recv = self._stage[k][1] if k < len(self._stage) else torch.zeros_like(self._recv)
            sets.append((send, recv))
        self._stage = sets[:max(1, len(self._stage))] if len(self._stage) < 2 else sets
        self._send = self._stage[0][0]
        self._tick = torch.zeros(1, dtype=torch.float32, device=dev)
        if dev.type == "cuda":
            self._alloc_ready = torch.cuda.Event()
            self._alloc_ready.record(torch.cuda.current_stream(dev))
        self._barrier()            # every rank has mapped every buffer before anyone pulls

    def _raise_together(self, errors, what):
        bad = [(r, e_) for r, e_ in enumerate(errors) if e_ is not None]
        if bad:
            comm, self._comm = self._comm, None
            self._stage_raw = None
            if comm is not None:
                self.e.lib.rgqa_peer_comm_destroy(comm)
            raise RuntimeError("peer exchange: rank %d could not %s: %s" % (bad[0][0], what, bad[0][1]))

    def describe(self):
        return "peer: " + super().describe()[len("sharded: "):].replace("all-to-all reduce-scatter", "reduce-scatter by hipIpc peer pulls").replace("weight all-gather", "weight all-gather by peer pulls")

    def _k(self):
        """which staging set the calling stream uses (the second exchange stream has its own)"""
        return 1 if (self.side2 is not None and len(self._stage) > 1 and torch.cuda.current_stream() == self.side2) else 0

    def _barrier(self):
        if self._host_staged:          # rehearsal over gloo (two processes on one device): the collective is a host call
            if self._tick.is_cuda:
                torch.cuda.current_stream().synchronize()
            self.dist.barrier()
        else:
            self.dist.all_reduce(self._tick)      # on the current stream: completes only when every rank's stream has reached it

    def _pull(self, src_off, nbytes, dst, stride):
        check(self.e.lib.rgqa_peer_pull(self._comm, src_off, nbytes, ptr(dst), stride, C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def _a2a(self, recv, send):
        k = self._k()
        W = self.world
        s = send.numel() // W
        mine = self._stage[k][0]
        if send.data_ptr() != mine.data_ptr():      # an f32 payload straight from the gradient arena: into the exported buffer first
            mine[:send.numel()].copy_(send)
        es = mine.element_size()
        self._barrier()                             # every rank's payload is staged
        self._pull(k * self._set_bytes + self.rank * s * es, s * es, recv, s * es)
        self._barrier()                             # every rank has pulled: the staging area may be overwritten

    def _ag(self, out, inp):
        k = self._k()
        nbytes = inp.numel() * inp.element_size()
        off = k * self._set_bytes + self._send_bytes
        area = self._stage_raw[off:off + nbytes].view(inp.dtype)
        area.copy_(inp)
        self._barrier()
        self._pull(off, nbytes, out, nbytes)
        self._barrier()

    def selfcheck(self):
        """One all-to-all and one all-gather by peer pulls on a small range, checked numerically: rank q stages the value q * W + r + 1 for rank r, so after
        the exchange part q of this rank's buffer must read q * W + rank + 1 (exact in bf16 up to W = 16).  -> bool.  bench.py's wire probe runs it before it
        times (and possibly selects) this exchange: a mapping that returns stale or foreign bytes must veto, not win on speed."""
        W, s = self.world, int(min(self.smax, 1 << 16))
        send, recv = self._stage[0][0][:W * s], self._stage[0][1][:W * s]
        for r in range(W):
            send[r * s:(r + 1) * s] = float(self.rank * W + r + 1)
        recv.zero_()
        self._a2a(recv, send)
        want = torch.tensor([q * W + self.rank + 1 for q in range(W)], dtype=torch.float32, device=recv.device)
        ok = bool((recv.view(W, s).float() == want[:, None]).all())
        part = torch.full((s,), float(100 + self.rank), dtype=recv.dtype, device=recv.device)
        recv.zero_()
        self._ag(recv, part)
        want = torch.arange(100, 100 + W, dtype=torch.float32, device=recv.device)
        return ok and bool((recv.view(W, s).float() == want[:, None]).all())

    def close(self):
        c, self._comm = getattr(self, "_comm", None), None
        if c is not None:
            self._stage_raw = None
            self.e.lib.rgqa_peer_comm_destroy(c)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


MODES = ("sharded", "sharded_bf16", "sharded_f32", "allreduce", "allreduce_bf16", "allreduce_f32", "peer", "peer_bf16", "peer_f32")


def make_exchange(engine, dist, mode=None, default="sharded", **kw):
    """mode: 'sharded' | 'allreduce', payload by the engine's precision (payload_dtype), or with a forcing suffix _bf16 / _f32; env RGQA_DP_MODE
    overrides `default` (bench.py: 'sharded'; the drop-in modules: 'allreduce' - the trainer's own optimizer steps every parameter)."""
    mode = mode or os.environ.get("RGQA_DP_MODE", default)
    if mode not in MODES:
        raise ValueError("RGQA_DP_MODE must be one of %s (got %r)" % (", ".join(MODES), mode))
    kind, _, forced = mode.partition("_")
    if kind == "sharded":
        return ShardedExchange(engine, dist, payload=forced or None, **kw)
    if kind == "peer":
        return PeerShardedExchange(engine, dist, payload=forced or None, **kw)
    # all-reduce: RCCL would keep a bf16 RUNNING sum across the ranks (error grows with the world size and depends on the reduction order), where
    # the sharded mode adds the N bf16 shards in f32 in rank order.  So the plain 'allreduce' is f32 whatever the engine's precision; the bf16
    # payload is opt-in by name.
    return GradAllReduce(engine, dist, bf16=(forced == "bf16"), **kw)
