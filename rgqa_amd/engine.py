"""Python handle on the C++/HIP engine (include/rgqa.h): flat parameter arenas, workspace, forward / backward.

torch is plumbing only here: it owns device memory (arenas, workspace, I/O tensors) and the HIP stream.
Every tensor operation of the hot path happens inside librgqa_hip.so.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib
from ._lib import Config, check, ptr, PREC_BF16, PREC_BF16X3, PREC_BF16X3_FWD, PREC_F32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_UPD_STREAMS = {}
_SIDE_SET = {}


def _update_stream(device):
    """ONE update stream per device for every engine of the process (like the library's weight-gradient side stream, csrc/engine.hip
    make_side_stream): engines run one after the other, and streams are a scarce resource - HIP maps them onto a handful of hardware queues
    in creation order, and a stream that ends up sharing a queue with the launch stream serialises with it (bench.py's legs make six
    engines: with a stream each, a later leg ran at 33 ms per step instead of 23)."""
    key = (device.type, device.index)
    st = _UPD_STREAMS.get(key)
    if st is None:
        from . import streams
        st = _UPD_STREAMS[key] = streams.pick(device, 3)[1]      # [0] is the library's weight-gradient side stream, [2] the second exchange stream
    return st


_RAW_GRAD = torch.Tensor.grad          # the C-level .grad descriptor


def raw_grad(p):
    """p.grad WITHOUT side effects: what the engine-side bookkeeping (arena bindings, the drop-in optimizer and clip) reads.  The `.grad` property
    of rgqa_amd's parameters (lxrt.modeling.ArenaParameter) first materialises a deferred clip_grad_norm_."""
    return _RAW_GRAD.__get__(p, type(p))


# engines by the address of their f32 parameter arena: lets the drop-in optimizer / clip helpers (rgqa_amd.lxrt.optimization) recognise
# parameters that are views of an engine's arena and use its fused kernels and operand copies
ARENAS = weakref.WeakValueDictionary()


def engine_of(ptr_bytes):
    """the engine whose parameter arena contains the device address, or None"""
    for base, e in list(ARENAS.items()):
        if e._params is not None and base <= ptr_bytes < base + 4 * e.arena_elems:
            return e
    return None


class ParamSpec:
    __slots__ = ("name", "offset", "shape", "linear", "dead", "f32_read")

    def __init__(self, name, offset, shape, flags):
        self.name, self.offset, self.shape = name, offset, tuple(shape)
        self.linear, self.dead = bool(flags & 1), bool(flags & 2)
        self.f32_read = bool(flags & 4)      # the forward reads the f32 master of this tensor whatever the precision

    @property
    def numel(self):
        n = 1
        for d in self.shape:
            n *= d
        return n


class Engine:
    """One engine = one model replica on one GPU."""

    def __init__(self, vocab_size=30522, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=9,
                 x_layers=5, r_layers=5, feat_dim=2048, pos_dim=4, num_answers=1842, precision="bf16",
                 ln_eps=1e-12, hidden_dropout=0.1, attn_dropout=0.1, arch=0, emb_dim=300):
        self.lib = _lib.load()
        self.precision = precision
        # "bf16x3": split-f32 operands, three bf16 MFMA products per f32 product - the fast mode inside the reference's 1e-3 logits bound;
        # "bf16x3_fwd": that forward pass (same kernels, same logits) with the bf16 backward pass - BASELINE config 3 prescribes a bf16 backward;
        # "bf16": BASELINE config 3's mode (outside that bound); "f32": exact f32 FMA arithmetic on the vector ALU (slow; the on-device reference)
        precision = {"fp32": "f32", "x3": "bf16x3", "x3fwd": "bf16x3_fwd"}.get(precision, precision)
        self.precision = precision
        prec = {"bf16": PREC_BF16, "f32": PREC_F32, "bf16x3": PREC_BF16X3, "bf16x3_fwd": PREC_BF16X3_FWD}[precision]
        self.cfg = Config(vocab_size, hidden, heads, inter, max_pos, type_vocab, l_layers, x_layers, r_layers, feat_dim,
                          pos_dim, num_answers, prec, ln_eps, hidden_dropout, attn_dropout, arch, emb_dim)
        h = C.c_void_p()
        check(self.lib.rgqa_engine_create(C.byref(self.cfg), C.byref(h)))
        self.h = h
        n = C.c_size_t()
        check(self.lib.rgqa_engine_arena_elems(self.h, C.byref(n)))
        self.arena_elems = n.value
        cnt = C.c_int()
        check(self.lib.rgqa_engine_num_params(self.h, C.byref(cnt)))
        self.specs = []
        buf = C.create_string_buffer(256)
        off, shape, nd, fl = C.c_size_t(), (C.c_int64 * 2)(), C.c_int(), C.c_int()
        for i in range(cnt.value):
            check(self.lib.rgqa_engine_param_info(self.h, i, buf, 256, C.byref(off), shape, C.byref(nd), C.byref(fl)))
            self.specs.append(ParamSpec(buf.value.decode(), off.value, [shape[k] for k in range(nd.value)], fl.value))
        b, e = C.c_size_t(), C.c_size_t()
        check(self.lib.rgqa_engine_dead_range(self.h, C.byref(b), C.byref(e)))
        self.dead_range = (b.value, e.value)
        self.device = None
        self._params = self._grads = self._params_lp = self._params_lp_t = self.workspace = None
        self._adam_m = self._adam_v = None
        self._seg_sumsq = None
        self._seg_sumsq_valid = False
        self._upd_stream = self._upd_done = self._upd_order = None
        self.adam_overlap = os.environ.get("RGQA_ADAM_OVERLAP", "1") != "0"      # adam_step's default for `overlap` (see _update_beside_forward)
        self._pending_clip = None        # max_norm of a deferred clip_grads_ (lxrt.optimization.clip_grad_norm_ -> BertAdam.step), else None
        self._dp_sharded = None          # the drop-in module's sharded exchange (lxrt.modeling._dp_exchange under RGQA_DP_MODE=sharded): clip / BertAdam go through it
        self._sharded_owner = None       # set by parallel.ShardedExchange while the f32 masters / Adam moments are valid on their owner rank only
        self.shape = None
        self._io = {}
        self._wev_refs = {}              # events handed to the C++ engine by raw handle (set_weight_event / set_backward_event), alive until the pass that waits is enqueued

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.rgqa_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # The arenas as the world outside this class sees them: reading (or replacing) one first makes the caller's stream wait for an optimizer pass
    # that may still be running beside the forward pass (adam_step, _update_beside_forward); the class itself works on the underscore fields.
    def _arena_property(name):
        raw = "_" + name

        def get(self):
            self.join_update()
            return getattr(self, raw)

        def put(self, value):
            self.join_update()
            setattr(self, raw, value)
        return property(get, put)

    params = _arena_property("params")
    grads = _arena_property("grads")
    params_lp = _arena_property("params_lp")
    params_lp_t = _arena_property("params_lp_t")
    adam_m = _arena_property("adam_m")
    adam_v = _arena_property("adam_v")
    del _arena_property

    # ------------------------------------------------------------------ memory
    def allocate(self, device):
        """Allocates the parameter / gradient arenas on `device` (a CUDA==HIP device)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("rgqa_amd runs on an MI355X (torch device 'cuda'); got %s. There is no CPU path." % device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        n = self.arena_elems
        for k in [k for k, v in list(ARENAS.items()) if v is self]:
            del ARENAS[k]
        self._params = torch.zeros(n, dtype=torch.float32, device=device)
        self._grads = torch.zeros(n, dtype=torch.float32, device=device)
        ARENAS[self._params.data_ptr()] = self
        if self.precision == "bf16":
            self._params_lp = torch.zeros(n, dtype=torch.bfloat16, device=device)
            self._params_lp_t = torch.zeros(n, dtype=torch.bfloat16, device=device)
        elif self.precision == "bf16x3":     # split-f32 operand copies: 4 bytes per element slot (csrc/common.h sf32), opaque to torch
            self._params_lp = torch.zeros(n, dtype=torch.int32, device=device)
            self._params_lp_t = torch.zeros(n, dtype=torch.int32, device=device)
        elif self.precision == "bf16x3_fwd":     # forward operand copy split f32, the backward's (transposed) copy bf16
            self._params_lp = torch.zeros(n, dtype=torch.int32, device=device)
            self._params_lp_t = torch.zeros(n, dtype=torch.bfloat16, device=device)
        self.shape = None
        return self

    def view(self, arena, spec):
        return arena[spec.offset:spec.offset + spec.numel].view(spec.shape)

    def ensure_shape(self, B, T, O):
        if self.shape == (B, T, O):
            return
        # An optimizer pass may still be running beside the forward (adam_step(overlap), the sharded exchange's gather): its last kernel reads the
        # transpose descriptor table out of the workspace, which the re-plan below moves or frees: the caller's stream - on which the
        # re-bind's uploads and the next pass run - first joins it, and a sharded exchange joins its gather
        self.join_update()
        owner = self._sharded_owner
        if owner is not None and hasattr(owner, "_join_gather"):
            owner._join_gather()
        need = C.c_size_t()
        check(self.lib.rgqa_engine_workspace_bytes(self.h, B, T, O, C.byref(need)))
        if self.workspace is None or self.workspace.numel() < need.value:
            self.workspace = None
            self.workspace = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        if self.device.type == "cuda" and (self.device.type, self.device.index) not in _SIDE_SET:
            # the library's weight-gradient side stream: one per device, picked so that it runs beside the caller's stream (streams.py)
            from . import streams
            _SIDE_SET[(self.device.type, self.device.index)] = sw = streams.pick(self.device, 3)[0]
            check(self.lib.rgqa_set_side_stream(self.device.index, C.c_void_p(sw.cuda_stream)))
        check(self.lib.rgqa_engine_bind(self.h, ptr(self._params), ptr(self._grads), ptr(self._params_lp), ptr(self._params_lp_t),
                                        ptr(self.workspace), self.workspace.numel(), B, T, O))
        if self._upd_stream is None and self.device.type == "cuda":
            # right behind the engine's own side stream (made by bind), before anything the caller makes: which hardware queue a stream lands
            # on - and with whom it shares it - follows the creation order (csrc/engine.hip make_side_stream)
            self._upd_stream = _update_stream(self.device)
        self.shape = (B, T, O)
        na, H = self.cfg.num_answers, self.cfg.hidden
        self._io = dict(logits=torch.empty(B, na, dtype=torch.float32, device=self.device),
                        pooled=torch.empty(B, H, dtype=torch.float32, device=self.device),
                        loss=torch.zeros(1, dtype=torch.float32, device=self.device))

    def sync_weights(self):
        """Refreshes the operand copies of the weights (direct + transposed; bf16 or split f32) from the f32 master arena."""
        if self.shape is None:
            raise RuntimeError("sync_weights before the first ensure_shape/bind")
        self.join_update()      # (reads the masters on the caller's stream: behind an optimizer pass that may still be running beside the forward)
        check(self.lib.rgqa_engine_sync_weights(self.h, _stream()))

    def sync_transposed(self):
        """only the transposed operand copies (after rgqa_bertadam_step wrote the direct copy itself)"""
        self.join_update()
        check(self.lib.rgqa_engine_sync_transposed(self.h, _stream()))

    # ------------------------------------------------------------------ compute
    def forward(self, feats, boxes, input_ids, input_mask, segment_ids=None, train=False, seed=0, lengths=None):
        """feats [B,O,F] f32, boxes [B,O,4] f32, ids/mask/(segment) [B,T] i64, all on the engine's device and
        contiguous. Returns (logits [B,NA] f32, pooled [B,H] f32) — engine-owned buffers, overwritten by the next call.
        lengths (host sequence of B ints = input_mask.sum(1) for the prefix masks the tokenizer builds): packs the
        language rows to the real tokens only (rgqa_engine_set_lengths) — same logits and gradients, B*T -> sum(lengths)
        rows of work; None computes every padded position as the reference does."""
        B, O, F = feats.shape
        T = input_ids.shape[1]
        if F != self.cfg.feat_dim or boxes.shape != (B, O, self.cfg.pos_dim) or input_mask.shape != (B, T):
            raise ValueError("bad input shapes: feats %s boxes %s ids %s mask %s" % (tuple(feats.shape), tuple(boxes.shape),
                                                                                     tuple(input_ids.shape), tuple(input_mask.shape)))
        for t, dt in ((feats, torch.float32), (boxes, torch.float32), (input_ids, torch.int64), (input_mask, torch.int64)):
            if t.dtype != dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError("inputs must be contiguous %s tensors on %s" % (dt, self.device))
        if segment_ids is not None and (segment_ids.dtype != torch.int64 or not segment_ids.is_contiguous()):
            raise ValueError("segment_ids must be a contiguous int64 tensor")
        self.ensure_shape(B, T, O)
        if lengths is not None:
            import numpy as np
            ln = np.ascontiguousarray(lengths, dtype=np.int32).reshape(-1)
            check(self.lib.rgqa_engine_set_lengths(self.h, C.c_void_p(ln.ctypes.data), int(ln.shape[0])))
            self._varlen = True
        elif getattr(self, "_varlen", False):
            check(self.lib.rgqa_engine_set_lengths(self.h, None, 0))
            self._varlen = False
        self._keep = (feats, boxes, input_ids, input_mask, segment_ids)   # backward reads them again
        lg, pl = self._io["logits"], self._io["pooled"]
        check(self.lib.rgqa_engine_forward(self.h, ptr(feats), ptr(boxes), ptr(input_ids), ptr(segment_ids), ptr(input_mask),
                                           ptr(pl), ptr(lg), lg.stride(0), 1 if train else 0, C.c_uint64(seed), _stream()))
        for k in [k for k in self._wev_refs if k != "bwd"]:      # the waits are enqueued: the one-shot weight events may go
            del self._wev_refs[k]
        self.join_update()      # an optimizer pass beside this forward (it finished long ago: 1.3 ms against 3.6): whatever follows on this stream sees its results
        return lg, pl

    def loss_backward(self, target, grad_scale=1.0, accumulate=False):
        """BCE x NA loss on the last forward's logits + backward into the gradient arena. Returns the loss (device scalar)."""
        if target.dtype != torch.float32 or target.stride(1) != 1:
            raise ValueError("target must be f32 with unit inner stride")
        if self._pending_clip is not None:
            self.flush_deferred_clip() if accumulate else self.drop_deferred_clip()
        check(self.lib.rgqa_engine_loss_backward(self.h, ptr(target), target.stride(0), ptr(self._io["loss"]), grad_scale,
                                                 1 if accumulate else 0, _stream()))
        self._wev_refs.pop("bwd", None)
        self._seg_sumsq_valid = self._seg_sumsq is not None
        return self._io["loss"]

    def backward(self, dlogits, accumulate=False):
        dlogits = dlogits.contiguous().float()
        check(self.lib.rgqa_engine_backward(self.h, ptr(dlogits), dlogits.stride(0), 1 if accumulate else 0, _stream()))
        self._wev_refs.pop("bwd", None)
        self._seg_sumsq_valid = self._seg_sumsq is not None

    def backward_pooled(self, dpooled, accumulate=False):
        dpooled = dpooled.contiguous().float()
        check(self.lib.rgqa_engine_backward_pooled(self.h, ptr(dpooled), dpooled.stride(0), 1 if accumulate else 0, _stream()))
        self._wev_refs.pop("bwd", None)

    def set_input_grads(self, dfeats=None, dboxes=None):
        """f32 device tensors [B*O, feat_dim] / [B*O, pos_dim] (or None) that the following backward calls fill with dL/dfeats and
        dL/dboxes (the reference's ODIN scorer, tasks/gqa_odin.py:97-121)."""
        for t in (dfeats, dboxes):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.device != self.device):
                raise ValueError("input-gradient buffers must be contiguous f32 tensors on %s" % self.device)
        self._in_grads = (dfeats, dboxes)
        check(self.lib.rgqa_engine_set_input_grads(self.h, ptr(dfeats), ptr(dboxes)))

    def activation(self, name, rows, cols=None):
        """f32 copy of a saved activation of the last forward pass; `rows` must be the row count of the recorded layout (packed
        language rows; B rows for the last language output, of which only the [CLS] rows are computed)."""
        out = torch.empty(rows, cols or self.cfg.hidden, dtype=torch.float32, device=self.device)
        check(self.lib.rgqa_engine_get_activation(self.h, name.encode(), ptr(out), out.numel(), _stream()))
        return out

    def cross_attention(self, layer, direction):
        """Attention probabilities of cross-modality layer `layer` from the last forward pass (lxrt_vis `output_attention`):
        direction 'l2v' -> [B, heads, T, O], 'v2l' -> [B, heads, O, T]; f32."""
        d = {"l2v": 0, "v2l": 1}[direction]
        B, T, O = self.shape
        shp = (B, self.cfg.heads, T, O) if d == 0 else (B, self.cfg.heads, O, T)
        out = torch.empty(shp, dtype=torch.float32, device=self.device)
        check(self.lib.rgqa_engine_get_cross_attention(self.h, int(layer), d, ptr(out), out.numel(), _stream()))
        return out

    # "gemm_nt": the forward launches of the NT GEMM family, "gemm_nt_dgrad": its dgrad launches (one category until round 5; under bf16x3_fwd the
    # forward launches are split-f32 kernels and the dgrad launches bf16 kernels)
    PROFILE_CATS = ("gemm_nt", "gemm_tn", "attn_fwd", "attn_bwd", "layernorm", "other", "gemm_nt_dgrad")

    def profile(self, enable):
        check(self.lib.rgqa_engine_profile(self.h, 1 if enable else 0))

    def profile_read(self):
        """{category: dict(ms, flops, bytes, launches)} accumulated since the last read (synchronises)."""
        n = len(self.PROFILE_CATS)
        ms, fl, by, ln = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
        torch.cuda.synchronize()
        check(self.lib.rgqa_engine_profile_read(self.h, ms, fl, by, ln, n))
        return {c: dict(ms=ms[i], flops=fl[i], bytes=by[i], launches=ln[i]) for i, c in enumerate(self.PROFILE_CATS)}

    PROFILE_BLOCKS = ("embeddings", "single_modality_layers", "cross_modality_layers", "pooler_head_loss")

    def profile_blocks(self):
        """The records of the last profile_read() split by model block: {block: dict(ms, flops)} (kernel-duration sums; GEMM +
        attention FLOPs). "cross_modality_layers" is the LXRTXLayer stack the north-star roofline target names."""
        n = len(self.PROFILE_BLOCKS)
        ms, fl = (C.c_double * n)(), (C.c_double * n)()
        check(self.lib.rgqa_engine_profile_blocks(self.h, ms, fl, n))
        return {b: dict(ms=ms[i], flops=fl[i]) for i, b in enumerate(self.PROFILE_BLOCKS)}

    def profile_operand_bytes(self):
        """{category: bytes of the GEMM operands alone (A + B + C)} of the last profile_read(): `bytes` without the fused epilogues' operands."""
        n = len(self.PROFILE_CATS)
        by = (C.c_double * n)()
        check(self.lib.rgqa_engine_profile_operand_bytes(self.h, by, n))
        return {c: by[i] for i, c in enumerate(self.PROFILE_CATS)}

    def grad_segments(self):
        """[(begin, end, event)] gradient-arena ranges in the order backward finalises them (dead range excluded)."""
        n = C.c_int()
        check(self.lib.rgqa_engine_num_grad_segments(self.h, C.byref(n)))
        out = []
        b, e, ev = C.c_size_t(), C.c_size_t(), C.c_int()
        for k in range(n.value):
            check(self.lib.rgqa_engine_grad_segment(self.h, k, C.byref(b), C.byref(e), C.byref(ev)))
            out.append((b.value, e.value, ev.value))
        return out

    def wait_grad_event(self, event, stream):
        """Makes torch stream `stream` wait until the gradient segment(s) tagged `event` of the last backward are final."""
        check(self.lib.rgqa_engine_wait_grad_event(self.h, event, C.c_void_p(stream.cuda_stream)))

    def set_weight_event(self, segment_event, event):
        """The next forward pass waits for torch event `event` (recorded once the weights of gradient segment `segment_event` are in place) before
        the first launch that reads those weights (rgqa_engine_set_weight_event; one-shot).  The engine handle keeps `event` alive until that
        pass has been enqueued (the C++ side holds the raw hipEvent_t only: ADVICE r5)."""
        if event is None:
            self._wev_refs.pop(int(segment_event), None)
        else:
            self._wev_refs[int(segment_event)] = event
        check(self.lib.rgqa_engine_set_weight_event(self.h, int(segment_event), C.c_void_p(event.cuda_event) if event is not None else None))

    def num_weight_segments(self):
        """segment events the engine's forward pass waits for (set_weight_event); 0: none - the weights must be in place before forward is called"""
        n = C.c_int(0)
        check(self.lib.rgqa_engine_num_weight_segments(self.h, C.byref(n)))
        return n.value

    def set_backward_event(self, event):
        """the next backward pass waits for `event` before its first launch (the transposed operand copies are refreshed behind it); one-shot"""
        if event is None:
            self._wev_refs.pop("bwd", None)
        else:
            self._wev_refs["bwd"] = event
        check(self.lib.rgqa_engine_set_backward_event(self.h, C.c_void_p(event.cuda_event) if event is not None else None))

    def enable_segment_sumsq(self, on=True):
        """Single-GPU training loops: let backward leave sum(g^2) of every gradient segment behind (rgqa_engine_set_grad_sumsq_slots),
        so adam_step(clip=True) adds ~20 numbers instead of re-reading the gradient arena. Anything that changes the gradients after
        backward (a data-parallel all-reduce, manual edits) must call invalidate_segment_sumsq()."""
        if on:
            n = C.c_int(0)
            check(self.lib.rgqa_engine_num_grad_segments(self.h, C.byref(n)))
            self._seg_sumsq = torch.zeros(n.value, dtype=torch.float32, device=self.device)
            check(self.lib.rgqa_engine_set_grad_sumsq_slots(self.h, ptr(self._seg_sumsq), n.value))
        else:
            check(self.lib.rgqa_engine_set_grad_sumsq_slots(self.h, None, 0))
            self._seg_sumsq = None
        self._seg_sumsq_valid = False

    def invalidate_segment_sumsq(self):
        self._seg_sumsq_valid = False

    # ------------------------------------------------------------------ optimizer (fused clip + BertAdam over the arena)
    def live_ranges(self):
        b, e = self.dead_range
        n = self.arena_elems
        return [(0, b), (e, n)] if e > b else [(0, n)]

    def adam_step(self, lr_t, max_norm=5.0, b1=0.9, b2=0.999, eps=1e-6, weight_decay=0.01, grad_prescale=1.0, clip=True, overlap=None):
        """clip_grad_norm_(params, max_norm) + BertAdam.step over every parameter that receives a gradient
        (gqa_conf.py:201-202), on the caller's stream; the same kernel re-writes the operand copy of the weights (bf16 / split f32)
        and the transposed copies follow."""
        if self._pending_clip is not None:
            self.flush_deferred_clip()
        if self._sharded_owner is not None:
            raise RuntimeError("adam_step: the optimizer state of this engine is sharded over the data-parallel ranks "
                               "(ShardedExchange); step through the exchange, or call its gather_master() / release() first")
        if self._adam_m is None:
            self._adam_m = torch.zeros_like(self._params)
            self._adam_v = torch.zeros_like(self._params)
        if getattr(self, "_sumsq", None) is None:
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._sq_ws = torch.zeros(2048, dtype=torch.float32, device=self.device)
        s = _stream()
        rngs = self.live_ranges()
        if clip and self._seg_sumsq is not None and self._seg_sumsq_valid:
            torch.sum(self._seg_sumsq, dim=0, keepdim=True, out=self._sumsq)      # the segments' shares, taken during backward
        elif clip:
            for i, (a, b) in enumerate(rngs):
                check(self.lib.rgqa_grad_sumsq(ptr(self._grads[a:b]), b - a, ptr(self._sq_ws), ptr(self._sumsq), 1 if i else 0, s))
        self._seg_sumsq_valid = False
        lp_split = 1 if self.precision in ("bf16x3", "bf16x3_fwd") else 0

        def update(a, b, stream):
            lp = ptr(self._params_lp[a:b]) if self._params_lp is not None else None
            check(self.lib.rgqa_bertadam_step(ptr(self._params[a:b]), ptr(self._grads[a:b]), ptr(self._adam_m[a:b]), ptr(self._adam_v[a:b]),
                                              lp, lp_split, b - a, lr_t, b1, b2, eps, weight_decay, ptr(self._sumsq) if clip else None,
                                              max_norm, grad_prescale, stream))

        if overlap is None:
            overlap = self.adam_overlap
        if overlap and self.device.type == "cuda" and self.num_weight_segments() > 0:
            return self._update_beside_forward(update)
        for a, b in rngs:
            update(a, b, s)
        if self._params_lp is not None:
            check(self.lib.rgqa_engine_sync_transposed(self.h, s))

    def _update_beside_forward(self, update):
        """The optimizer pass beside the NEXT forward pass (round 5): BertAdam is a pure HBM stream (30 bytes per parameter: 1.1 ms at B = 256), the
        forward pass that follows it is bound by its GEMMs.  The update runs on a stream of its own, gradient segment by gradient segment in
        FORWARD order (embeddings first, the answer head last), one event per segment; the engine's forward waits, per layer, for the event of the
        segment that holds the layer's weights (rgqa_engine_set_weight_event - the mechanism the sharded exchange's weight all-gather uses), the
        next backward for the transposed operand copies re-made behind the last segment (rgqa_engine_set_backward_event).  The caller's stream
        goes straight on to the next forward.  Whoever reads the parameters OUTSIDE the engine (state_dict, a checkpoint, a test) first calls
        join_update() - or synchronises the device."""
        cur = torch.cuda.current_stream(self.device)
        if self._upd_order is None:
            if self._upd_stream is None:
                self._upd_stream = _update_stream(self.device)      # (stream priorities make no difference: measured)
            by_ev = {}
            for a, b, ev in self.grad_segments():
                by_ev.setdefault(ev, []).append((a, b))
            self._upd_order = [(ev, by_ev[ev]) for ev in sorted(by_ev, reverse=True)]
            cov = sorted(r for _, rs in self._upd_order for r in rs)
            merged = []
            for a, b in cov:
                if merged and a <= merged[-1][1]:
                    merged[-1] = (merged[-1][0], max(merged[-1][1], b))
                else:
                    merged.append((a, b))
            if merged != [tuple(r) for r in self.live_ranges()]:
                raise RuntimeError("adam_step(overlap): the gradient segments do not cover the live parameter ranges")
        side = self._upd_stream
        ready = torch.cuda.Event()
        ready.record(cur)                              # the gradients and the norm are final on the caller's stream
        evs = []
        with torch.cuda.stream(side):
            side.wait_event(ready)
            ss = C.c_void_p(side.cuda_stream)
            for ev, ranges in self._upd_order:
                for a, b in ranges:
                    update(a, b, ss)
                t = torch.cuda.Event()
                t.record(side)
                evs.append(t)
                self.set_weight_event(ev, t)
            if self._params_lp is not None:
                check(self.lib.rgqa_engine_sync_transposed(self.h, ss))
            t = torch.cuda.Event()
            t.record(side)
            evs.append(t)
            self.set_backward_event(t)
        self._upd_events, self._upd_done = evs, t      # alive until the passes that wait for them have been enqueued (the next step replaces them)

    def join_update(self):
        """the caller's stream waits for an optimizer pass still running beside the forward (adam_step(overlap=True)); no-op otherwise"""
        t, self._upd_done = getattr(self, "_upd_done", None), None
        if t is not None:
            torch.cuda.current_stream(self.device).wait_event(t)

    def flush_deferred_clip(self):
        """Materialises a deferred clip_grads_(..., defer=True): the gradient arena is scaled in place now (one kernel; no traffic when the
        norm was below max_norm)."""
        mn, self._pending_clip = getattr(self, "_pending_clip", None), None
        if mn is not None:
            self.join_update()      # (writes the gradient arena)
            s = _stream()
            for a, b in self.live_ranges():
                check(self.lib.rgqa_clip_scale(ptr(self._grads[a:b]), b - a, ptr(self._sumsq), float(mn), s))

    def drop_deferred_clip(self):
        self._pending_clip = None

    def clip_grads_(self, max_norm, defer=False):
        """nn.utils.clip_grad_norm_(params, max_norm) (tasks/gqa_conf.py:201) on the gradient arena, in place: the norm from the per-segment
        sums backward left behind when they are valid (enable_segment_sumsq), else from one pass over the live ranges; the gradients are
        rescaled by one kernel only when the norm exceeds max_norm.  Returns the total norm (device scalar), as torch does.
        defer=True (the drop-in clip_grad_norm_): the norm is taken and returned, the rescale is left PENDING (`_pending_clip` = max_norm, the sum
        of squares stays in `_sumsq`): lxrt.optimization.BertAdam.step folds the coefficient into its update kernel - as adam_step does - and
        whoever reads a .grad first (ArenaParameter.grad) or accumulates onto the gradients gets it materialised by flush_deferred_clip()."""
        if getattr(self, "_pending_clip", None) is not None:
            self.flush_deferred_clip()              # two clips in a row: the second one measures the scaled gradients
            self._seg_sumsq_valid = False
        if getattr(self, "_sumsq", None) is None:
            self._sumsq = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._sq_ws = torch.zeros(2048, dtype=torch.float32, device=self.device)
        s = _stream()
        rngs = self.live_ranges()
        if self._seg_sumsq is not None and self._seg_sumsq_valid:
            torch.sum(self._seg_sumsq, dim=0, keepdim=True, out=self._sumsq)
        else:
            for i, (a, b) in enumerate(rngs):
                check(self.lib.rgqa_grad_sumsq(ptr(self._grads[a:b]), b - a, ptr(self._sq_ws), ptr(self._sumsq), 1 if i else 0, s))
        self._seg_sumsq_valid = False            # the gradients change below (or may have: the caller owns them from here on)
        if defer:
            self._pending_clip = float(max_norm)
            return self._sumsq.sqrt().reshape(())
        for a, b in rngs:
            check(self.lib.rgqa_clip_scale(ptr(self._grads[a:b]), b - a, ptr(self._sumsq), float(max_norm), s))
        return self._sumsq.sqrt().reshape(())

    def grad_norm(self):
        """Global L2 norm of the live gradient ranges (what clip_grad_norm_ measures, gqa_conf.py:201) as a device scalar."""
        tot = torch.zeros(1, dtype=torch.float32, device=self.device)
        ws = torch.zeros(2048, dtype=torch.float32, device=self.device)
        for i, (a, b) in enumerate(self.live_ranges()):
            check(self.lib.rgqa_grad_sumsq(ptr(self._grads[a:b]), b - a, ptr(ws), ptr(tot), 1 if i else 0, _stream()))
        return tot.sqrt()
