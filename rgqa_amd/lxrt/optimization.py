"""Mirror of the reference's lxrt/optimization.py (BertAdam + warmup schedules, :20-180) on the fused HIP kernel.

Same constructor, defaults, `get_lr()` and `step()` semantics: no bias correction, eps outside the sqrt, decoupled
weight decay added to the update of EVERY parameter, learning rate from the per-parameter step counter read before it
is incremented, parameters whose grad is None skipped.  Gradient clipping stays outside (the reference calls
`nn.utils.clip_grad_norm_` before `step()`, tasks/gqa_conf.py:201).  Parameters that are adjacent views of a flat
arena (rgqa_amd models) are updated by one kernel launch per contiguous run instead of one per tensor."""
import ctypes as C
import logging
import math
import os

import torch
from torch.optim import Optimizer
from torch.optim.optimizer import required

from .. import _lib
from .. import engine as _engine
from ..engine import raw_grad

logger = logging.getLogger(__name__)

_torch_clip_grad_norm_ = torch.nn.utils.clip_grad_norm_


def clip_grad_norm_(parameters, max_norm, norm_type=2.0, error_if_nonfinite=False, foreach=None):
    """`torch.nn.utils.clip_grad_norm_` with a fast path for the trainer's call (tasks/gqa_conf.py:201, unchanged): when every parameter
    that has a gradient is a view of ONE engine's arena with its gradient the matching view of the gradient arena (what rgqa_amd
    models hand out), the 2-norm comes from the sums backward already took segment by segment and the rescale is one kernel over
    the arena that moves no data unless the norm exceeds max_norm - instead of ~440 tensors' norms, a stack, and ~440 multiplies.
    Same result (in-place scaled .grad, the total norm returned as a tensor); anything else goes to torch's implementation."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = list(parameters)
    eng = None
    if float(norm_type) == 2.0 and not error_if_nonfinite and params:
        sig = tuple(id(p) for p in params)
        cached = _CLIP_CACHE.get(sig[0])
        if cached is not None and cached[0] == sig and cached[1]() is not None:
            eng = cached[1]()
            # revalidation on EVERY call: each gradient is still the arena view it was when the engine was recognised (a foreign
            # tensor bound to .grad - a clone, an unscaled copy, None for a frozen parameter - sends the call to torch's implementation),
            # and the arenas themselves have not moved (a re-materialised module re-packs).  One data_ptr() per parameter.
            if eng._params is None or eng._grads is None or eng._grads.data_ptr() != cached[3]:
                eng = None
            else:
                for p, want in zip(params, cached[2]):
                    g = raw_grad(p)
                    if (g is None) != (want is None) or (g is not None and g.data_ptr() != want):
                        eng = None
                        break
            if eng is None:
                _CLIP_CACHE.clear()
        if eng is None:
            eng = _arena_engine_of(params)
            if eng is not None:
                import weakref
                _CLIP_CACHE.clear()
                _CLIP_CACHE[sig[0]] = (sig, weakref.ref(eng), [None if raw_grad(p) is None else raw_grad(p).data_ptr() for p in params], eng._grads.data_ptr())
    if eng is None:
        return _torch_clip_grad_norm_(params, max_norm, norm_type=norm_type, error_if_nonfinite=error_if_nonfinite, foreach=foreach)
    ex = _sharded_exchange_of(eng)
    if ex is not None:
        # data parallel, sharded optimizer (RGQA_DP_MODE=sharded): backward() left this rank the reduced gradients of the ranges it owns and the owners'
        # shares of sum(g^2); the global norm is their sum over the ranks (one scalar all-reduce, reused by BertAdam.step); the rescale is always deferred -
        # the gradient views outside the owned ranges hold local values nobody reads
        ss = ex.global_sumsq()
        eng._sumsq, eng._pending_clip, eng._seg_sumsq_valid = ss, float(max_norm), False
        return ss.sqrt().reshape(())
    binding = getattr(eng, "_binding_ref", None)
    binding = binding() if binding is not None else None
    if binding is None or not binding.grads_untouched():
        eng.invalidate_segment_sumsq()           # someone wrote to a .grad since backward: take the norm from the arena itself
    # The in-place rescale is DEFERRED (default): the norm is taken and returned now, the coefficient stays with the engine; the drop-in
    # BertAdam.step folds it into its update kernel (as the engine-direct adam_step does: identical arithmetic, g * coef in f32 either way),
    # and anything that reads a .grad before that (ArenaParameter.grad) or accumulates onto the gradients materialises it first.  What differs
    # from torch: AFTER BertAdam.step the .grad views hold the unclipped gradients (the reference's loops call zero_grad() next).
    # RGQA_DEFER_CLIP=0: scale in place here, as torch does (0.3 ms per step at the full model while clipping is active).
    return eng.clip_grads_(max_norm, defer=os.environ.get("RGQA_DEFER_CLIP", "1") != "0")


def _sharded_exchange_of(eng):
    """the engine's sharded data-parallel exchange (lxrt.modeling._dp_exchange under RGQA_DP_MODE=sharded) while a process group of more than one rank
    is up and the module is training, else None"""
    ex = getattr(eng, "_dp_sharded", None)
    if ex is None:
        return None
    from .modeling import _dp_exchanging
    return ex if _dp_exchanging() else None


_CLIP_CACHE = {}
_CLIP_OWNERS = 0


def install_clip_routing(owner):
    """Routes `torch.nn.utils.clip_grad_norm_` through clip_grad_norm_ above for as long as `owner` - an engine-backed module - is alive: a scope the
    drop-in objects install and remove (a finaliser on the owner; the last one restores torch's own function), not a side effect of an import.
    Names bound with `from torch.nn.utils import clip_grad_norm_` before the owner existed keep torch's function.  RGQA_PATCH_CLIP=0: no routing."""
    global _CLIP_OWNERS
    if os.environ.get("RGQA_PATCH_CLIP", "1") == "0":
        return False
    import weakref
    if _CLIP_OWNERS == 0 and torch.nn.utils.clip_grad_norm_ is not clip_grad_norm_:
        torch.nn.utils.clip_grad_norm_ = clip_grad_norm_
        logger.info("rgqa_amd: torch.nn.utils.clip_grad_norm_ is routed through rgqa_amd.lxrt.optimization.clip_grad_norm_ while an engine-backed model is alive "
                    "(engine arenas take a fused path, everything else torch's own implementation; RGQA_PATCH_CLIP=0 disables)")
    _CLIP_OWNERS += 1
    weakref.finalize(owner, _release_clip_routing)
    return True


def _release_clip_routing():
    global _CLIP_OWNERS
    _CLIP_OWNERS = max(0, _CLIP_OWNERS - 1)
    if _CLIP_OWNERS == 0 and torch.nn.utils.clip_grad_norm_ is clip_grad_norm_:
        torch.nn.utils.clip_grad_norm_ = _torch_clip_grad_norm_
        _CLIP_CACHE.clear()


def engine_view_offset(eng, p):
    """element offset of parameter p in the engine's arena if p AND p.grad are the arena views, else None"""
    off = p.data_ptr() - eng._params.data_ptr()
    if off < 0 or off % 4 or off // 4 + p.numel() > eng.arena_elems or p.dtype != torch.float32:
        return None
    g = raw_grad(p)
    if g is not None and g.data_ptr() != eng._grads.data_ptr() + off:
        return None
    return off // 4


def _arena_engine_of(params):
    """the engine all of `params` belong to (every gradient-carrying one a full arena view, all live ranges covered), or None"""
    live = [p for p in params if raw_grad(p) is not None]
    if not live or not live[0].is_cuda:
        return None
    eng = _engine.engine_of(live[0].data_ptr())
    if eng is None or eng._grads is None:
        return None
    covered = 0
    for p in live:
        if engine_view_offset(eng, p) is None or not raw_grad(p).is_contiguous():
            return None
        covered += p.numel()
    # every tensor that backward writes must be among them: otherwise the arena norm would count gradients the caller did not pass
    want = sum(sp.numel for sp in eng.specs if not sp.dead)
    return eng if covered == want else None


def warmup_cosine(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 0.5 * (1.0 + math.cos(math.pi * x))


def warmup_constant(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return 1.0


def warmup_linear(x, warmup=0.002):
    if x < warmup:
        return x / warmup
    return max((x - 1.) / (warmup - 1.), 0)


SCHEDULES = {'warmup_cosine': warmup_cosine, 'warmup_constant': warmup_constant, 'warmup_linear': warmup_linear}


class BertAdam(Optimizer):
    def __init__(self, params, lr=required, warmup=-1, t_total=-1, schedule='warmup_linear', b1=0.9, b2=0.999, e=1e-6,
                 weight_decay=0.01, max_grad_norm=1.0):
        if lr is not required and lr < 0.0:
            raise ValueError("Invalid learning rate: {} - should be >= 0.0".format(lr))
        if schedule not in SCHEDULES:
            raise ValueError("Invalid schedule parameter: {}".format(schedule))
        if not 0.0 <= warmup < 1.0 and not warmup == -1:
            raise ValueError("Invalid warmup: {} - should be in [0.0, 1.0[ or -1".format(warmup))
        if not 0.0 <= b1 < 1.0:
            raise ValueError("Invalid b1 parameter: {} - should be in [0.0, 1.0[".format(b1))
        if not 0.0 <= b2 < 1.0:
            raise ValueError("Invalid b2 parameter: {} - should be in [0.0, 1.0[".format(b2))
        if not e >= 0.0:
            raise ValueError("Invalid epsilon value: {} - should be >= 0.0".format(e))
        defaults = dict(lr=lr, schedule=schedule, warmup=warmup, t_total=t_total, b1=b1, b2=b2, e=e,
                        weight_decay=weight_decay, max_grad_norm=max_grad_norm)
        super(BertAdam, self).__init__(params, defaults)
        self._runs = {}

    def get_lr(self):
        lr = []
        for group in self.param_groups:
            for p in group['params']:
                state = self.state[p]
                if len(state) == 0:
                    return [0]
                if group['t_total'] != -1:
                    lr_scheduled = group['lr'] * SCHEDULES[group['schedule']](state['step'] / group['t_total'], group['warmup'])
                else:
                    lr_scheduled = group['lr']
                lr.append(lr_scheduled)
        return lr

    # a run = maximal sequence of parameters whose data AND grads are equally spaced neighbours in two flat buffers.
    # Parameters are visited in ADDRESS order and only alignment padding (< 64 elements) may separate neighbours, so a
    # run never spans another tensor (registration order differs from arena order; parameters without gradients sit
    # between live ones).
    def _build_runs(self, gi, plist):
        runs, cur = [], None
        for p in sorted(plist, key=lambda q: q.data_ptr()):
            g = raw_grad(p)
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous() and g.dtype == torch.float32):
                raise RuntimeError("BertAdam (rgqa_amd): parameters and gradients must be contiguous f32 tensors on the MI355X; "
                                   "there is no CPU path")
            st = self.state[p]
            if cur is not None:
                gap_p = p.data_ptr() - cur["p_end"]
                gap_g = g.data_ptr() - cur["g_end"]
                if 0 <= gap_p == gap_g <= 252 and st.get("step", 0) == cur["step"] and p.device == cur["dev"] and ("next_m" in st) == cur["has_state"]:
                    cur["params"].append(p)
                    cur["p_end"] = p.data_ptr() + 4 * p.numel()
                    cur["g_end"] = g.data_ptr() + 4 * p.numel()
                    continue
            cur = dict(params=[p], p0=p.data_ptr(), g0=g.data_ptr(), p_end=p.data_ptr() + 4 * p.numel(),
                       g_end=g.data_ptr() + 4 * p.numel(), step=st.get("step", 0), dev=p.device, has_state="next_m" in st)
            runs.append(cur)
        import weakref
        for r in runs:
            n = (r["p_end"] - r["p0"]) // 4
            r["n"] = n
            r["numel"] = sum(p.numel() for p in r["params"])
            eng = _engine.engine_of(r["p0"])
            r["eng"] = weakref.ref(eng) if eng is not None else None
            if not r["has_state"]:
                if eng is not None and _sharded_exchange_of(eng) is not None and eng._grads is not None and r["g0"] - eng._grads.data_ptr() == r["p0"] - eng._params.data_ptr():
                    # sharded data-parallel optimizer: the moments live in the ENGINE's arenas (each rank only ever touches the ranges it owns)
                    if eng._adam_m is None:
                        eng._adam_m, eng._adam_v = torch.zeros_like(eng._params), torch.zeros_like(eng._params)
                    off = (r["p0"] - eng._params.data_ptr()) // 4
                    r["m"], r["v"] = eng._adam_m[off:off + n], eng._adam_v[off:off + n]
                else:
                    r["m"] = torch.zeros(n, dtype=torch.float32, device=r["dev"])
                    r["v"] = torch.zeros(n, dtype=torch.float32, device=r["dev"])
                for p in r["params"]:
                    off = (p.data_ptr() - r["p0"]) // 4
                    st = self.state[p]
                    st["step"] = 0
                    st["next_m"] = r["m"][off:off + p.numel()].view_as(p)
                    st["next_v"] = r["v"][off:off + p.numel()].view_as(p)
            else:   # state restored from a checkpoint or created earlier per tensor: one launch per tensor
                r["m"] = r["v"] = None
        return runs

    def zero_grad(self, set_to_none=True):
        if not set_to_none:      # zeroing in place writes the gradient arena: an update still running beside the forward pass reads it
            for eng in list(_engine.ARENAS.values()):
                eng.join_update()
        return super().zero_grad(set_to_none=set_to_none)

    def step(self, closure=None):
        loss = None
        if closure is not None:
            loss = closure()
        lib = _lib.load()
        stream = None
        warned_for_t_total = False
        # ---- pass 1: the runs of every group (cached by the (data, grad) addresses) and how much of each engine's arena they cover
        work, cover = [], {}
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group['params'] if raw_grad(p) is not None]
            if not plist:
                continue
            for p in plist:
                if raw_grad(p).is_sparse:
                    raise RuntimeError('Adam does not support sparse gradients, please consider SparseAdam instead')
            sig = tuple((p.data_ptr(), raw_grad(p).data_ptr()) for p in plist)
            cached = self._runs.get(gi)
            if cached is None or cached[0] != sig:
                cached = (sig, self._build_runs(gi, plist))
                self._runs[gi] = cached
            work.append((group, cached[1]))
            for r in cached[1]:
                eng = self._run_engine(r) if r["m"] is not None else None      # (per-tensor state, e.g. restored from a checkpoint: not the fused path)
                if eng is not None:
                    ent = cover.setdefault(id(eng), [eng, 0])
                    ent[1] += r["numel"]
        # ---- data parallel with a sharded optimizer (RGQA_DP_MODE=sharded, round 6): this rank owns 1/N of the arena - the exchange clips and updates the
        # owned ranges and gathers the weights beside the next forward pass; replaces the every-rank update below for that engine's runs
        sharded = self._step_sharded(work, cover)
        # ---- a clip_grad_norm_ that was deferred to this step (clip_grad_norm_ above): folded into the update kernel when this call updates
        # EVERY gradient-carrying parameter of the engine (then no gradient is left behind unscaled), materialised in place otherwise
        fold = {}
        for key, (eng, n_cov) in cover.items():
            if key in sharded or getattr(eng, "_pending_clip", None) is None:
                continue
            if n_cov == self._live_numel(eng):
                fold[key] = (_lib.ptr(eng._sumsq), float(eng._pending_clip))
            else:
                eng.flush_deferred_clip()
        # ---- an engine whose EVERY live parameter this call updates through the fused path, with its operand copies current, takes the update
        # beside its next forward pass (Engine._update_beside_forward: gradient segment by gradient segment in forward order on a stream of its
        # own, the forward waiting per layer) - what the engine-direct adam_step does; RGQA_ADAM_OVERLAP=0 keeps it on this stream
        beside = {}
        for key, (eng, n_cov) in cover.items():
            if key in sharded or n_cov != self._live_numel(eng) or not getattr(eng, "adam_overlap", False) or eng._params_lp is None or eng.num_weight_segments() <= 0:
                continue
            b = getattr(eng, "_binding_ref", None)
            b = b() if b is not None else None
            if b is not None and b.in_sync():
                beside[key] = (eng, b, [])
        # ---- pass 2: the update launches
        for group, runs in work:
            if stream is None:
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            touched = {}
            for r in runs:
                if r["m"] is not None and sharded and id(self._run_engine(r)) in sharded:
                    continue                                    # updated by the exchange above
                step = self.state[r["params"][0]]["step"]
                if group['t_total'] != -1:
                    progress = step / group['t_total']
                    lr_scheduled = group['lr'] * SCHEDULES[group['schedule']](progress, group['warmup'])
                    if group['schedule'] == "warmup_linear" and progress > 1. and not warned_for_t_total:
                        logger.warning("Training beyond specified 't_total' steps with schedule '{}'. Learning rate set to {}. "
                                       "Please set 't_total' of {} correctly.".format(group['schedule'], lr_scheduled, self.__class__.__name__))
                        warned_for_t_total = True
                else:
                    lr_scheduled = group['lr']
                if r["m"] is not None:
                    # a run inside an engine's arena: the same kernel also re-writes the engine's operand copy of these weights (bf16 / split
                    # f32), so the next forward need not re-cast the whole arena
                    lp, lp_split = None, 0
                    sumsq, max_norm = None, 0.0
                    eng = self._run_engine(r)
                    if eng is not None:
                        sumsq, max_norm = fold.get(id(eng), (None, 0.0))
                    if eng is not None and id(eng) in beside:
                        # launched below, cut at the gradient segments' boundaries: (arena offset, elements, this run's arguments)
                        off = (r["p0"] - eng._params.data_ptr()) // 4
                        beside[id(eng)][2].append((off, r["n"], r, (lr_scheduled, group['b1'], group['b2'], group['e'], group['weight_decay'], sumsq, max_norm)))
                    else:
                        if eng is not None and eng._params_lp is not None:
                            off = (r["p0"] - eng._params.data_ptr()) // 4
                            lp_split = 1 if eng.precision in ("bf16x3", "bf16x3_fwd") else 0
                            lp = C.c_void_p(eng._params_lp.data_ptr() + off * eng._params_lp.element_size())
                            ent = touched.setdefault(id(eng), [eng, 0, None])
                            ent[1] += r["numel"]
                            if ent[2] is None:
                                b = getattr(eng, "_binding_ref", None)
                                b = b() if b is not None else None
                                ent[2] = b if (b is not None and b.in_sync()) else False
                        _lib.check(lib.rgqa_bertadam_step(C.c_void_p(r["p0"]), C.c_void_p(r["g0"]), _lib.ptr(r["m"]), _lib.ptr(r["v"]), lp, lp_split,
                                                          r["n"], lr_scheduled, group['b1'], group['b2'], group['e'], group['weight_decay'],
                                                          sumsq, max_norm, 1.0, stream))
                else:
                    for p in r["params"]:
                        st = self.state[p]
                        _lib.check(lib.rgqa_bertadam_step(_lib.ptr(p.data), _lib.ptr(raw_grad(p)), _lib.ptr(st["next_m"]), _lib.ptr(st["next_v"]),
                                                          None, 0, p.numel(), lr_scheduled, group['b1'], group['b2'], group['e'],
                                                          group['weight_decay'], None, 0.0, 1.0, stream))
                for p in r["params"]:
                    self.state[p]["step"] += 1
                # the kernel wrote through raw pointers: bump one version counter per run so that owners of derived data
                # (the engine's bf16 weight copies) notice the change, as they would after any in-place torch op
                p0 = r["params"][0]
                torch._C._autograd._unsafe_set_version_counter((p0,), (p0._version + 1,))
            for eng, n_upd, binding in touched.values():
                # every live parameter of the engine went through the fused path and its copies were current before: finish them (the
                # transposed copies) and tell the binding, instead of a full re-cast at the next forward
                if binding and n_upd == self._live_numel(eng):
                    eng.sync_transposed()
                    binding.mark_synced()
        for eng, binding, pieces in beside.values():
            lp_split = 1 if eng.precision in ("bf16x3", "bf16x3_fwd") else 0
            lp_base, lp_es = eng._params_lp.data_ptr(), eng._params_lp.element_size()

            def update(a, b, st, pieces=pieces, lp_split=lp_split, lp_base=lp_base, lp_es=lp_es):
                # the part of every run that lies in arena elements [a, b)
                for off, n, r, (lr_s, b1, b2, e_, wd, sumsq, max_norm) in pieces:
                    lo, hi = max(a, off), min(b, off + n)
                    if hi > lo:
                        d = lo - off
                        _lib.check(lib.rgqa_bertadam_step(C.c_void_p(r["p0"] + 4 * d), C.c_void_p(r["g0"] + 4 * d), C.c_void_p(r["m"].data_ptr() + 4 * d),
                                                          C.c_void_p(r["v"].data_ptr() + 4 * d), C.c_void_p(lp_base + lo * lp_es), lp_split, hi - lo,
                                                          lr_s, b1, b2, e_, wd, sumsq, max_norm, 1.0, st))
            eng._update_beside_forward(update)          # ... and the transposed copies behind the last segment
            binding.mark_synced()
        for key in fold:
            cover[key][0].drop_deferred_clip()          # consumed: the update used g * coef; the .grad views keep the unclipped gradients
        return loss

    def _step_sharded(self, work, cover):
        """-> {id(engine)} of the engines whose update went through their sharded exchange"""
        done = set()
        for key, (eng, n_cov) in cover.items():
            ex = _sharded_exchange_of(eng)
            if ex is None:
                continue
            if n_cov != self._live_numel(eng):
                raise RuntimeError("BertAdam (rgqa_amd, RGQA_DP_MODE=sharded): the sharded data-parallel optimizer needs EVERY gradient-carrying parameter of the "
                                   "model in this optimizer with freshly created state (the ranks own ranges of the flat arena, not tensors); use RGQA_DP_MODE=allreduce")
            hyper = set()
            runs_e = []
            for group, runs in work:
                for r in runs:
                    if r["m"] is None or self._run_engine(r) is not eng:
                        continue
                    step = self.state[r["params"][0]]["step"]
                    lr_s = group['lr'] * SCHEDULES[group['schedule']](step / group['t_total'], group['warmup']) if group['t_total'] != -1 else group['lr']
                    hyper.add((lr_s, group['b1'], group['b2'], group['e'], group['weight_decay']))
                    runs_e.append(r)
            if len(hyper) != 1:
                raise RuntimeError("BertAdam (rgqa_amd, RGQA_DP_MODE=sharded): one learning rate / beta / weight-decay setting for all parameters is required "
                                   "(got %d); use RGQA_DP_MODE=allreduce" % len(hyper))
            lr_s, b1, b2, e_, wd = next(iter(hyper))
            pend = getattr(eng, "_pending_clip", None)
            # dL/dlogits was scaled by 1 / world in backward(): the exchanged sums are the mean gradient already
            ex.step(lr_s, max_norm=float(pend) if pend is not None else 0.0, b1=b1, b2=b2, eps=e_, weight_decay=wd, clip=pend is not None, grad_prescale=1.0)
            eng.drop_deferred_clip()
            for r in runs_e:
                for p in r["params"]:
                    self.state[p]["step"] += 1
                p0 = r["params"][0]
                torch._C._autograd._unsafe_set_version_counter((p0,), (p0._version + 1,))
            b = getattr(eng, "_binding_ref", None)
            b = b() if b is not None else None
            if b is not None:
                b.mark_synced()          # the exchange re-made (or gathers beside the next forward) every operand copy
            done.add(key)
        return done

    @staticmethod
    def _live_numel(eng):
        n = getattr(eng, "_live_numel_cache", None)
        if n is None:
            n = eng._live_numel_cache = sum(sp.numel for sp in eng.specs if not sp.dead)
        return n

    @staticmethod
    def _run_engine(r):
        """the engine whose arenas the run lies in - parameters in its parameter arena, gradients at the same offsets of its gradient arena - or None"""
        ref = r.get("eng")
        eng = ref() if ref is not None else None
        if eng is None or eng._params is None or eng._grads is None:
            return None
        if r["g0"] - eng._grads.data_ptr() != r["p0"] - eng._params.data_ptr():
            return None
        return eng
