"""Mirror of the reference's lxrt/modeling.py surface for the GQA path, backed by the HIP engine.

Same names and meanings: `BertConfig` (modeling.py:172-258), `VisualConfig` / `VISUAL_CONFIG` (141-169),
`BertLayerNorm` (261), `gelu` / `GeLU` (112-131), `BertPreTrainedModel.init_bert_weights` (688-699),
`LXRTFeatureExtraction(config, mode='x')` (1005-1030) with `state_dict()` keys identical to the reference
(`bert.embeddings.word_embeddings.weight`, `bert.encoder.x_layers.3.visual_attention.att.query.weight`, ...).

The module tree below only CARRIES parameters (real nn.Linear / nn.Embedding / nn.LayerNorm leaves, so `.apply(init)`,
`load_state_dict`, optimizers and checkpoint surgery work as in the reference); every tensor operation of forward and
backward runs in librgqa_hip.so on flat arenas that these parameters are views of.  No CPU path: calling forward
without an MI355X / the built library raises.
"""
import copy
import json
import logging
import math
import os

import torch
from torch import nn

from ..engine import Engine, raw_grad, _RAW_GRAD

logger = logging.getLogger(__name__)


def gelu(x):
    """Exact-erf GeLU as the reference's helper (utility for callers' own heads; the engine fuses its own)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


class GeLU(nn.Module):
    def forward(self, x):
        return gelu(x)


def swish(x):
    return x * torch.sigmoid(x)


ACT2FN = {"gelu": gelu, "relu": torch.nn.functional.relu, "swish": swish}


class VisualConfig(object):
    VISUAL_LOSSES = ['obj', 'attr', 'feat']

    def __init__(self, l_layers=12, x_layers=5, r_layers=0):
        self.l_layers, self.x_layers, self.r_layers = l_layers, x_layers, r_layers
        self.visual_feat_dim = 2048
        self.visual_pos_dim = 4
        self.obj_id_num = 1600
        self.attr_id_num = 400
        self.visual_losses = self.VISUAL_LOSSES

    def set_visual_dims(self, feat_dim, pos_dim):
        self.visual_feat_dim = feat_dim
        self.visual_pos_dim = pos_dim


VISUAL_CONFIG = VisualConfig()


class BertConfig(object):
    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                for key, value in json.loads(reader.read()).items():
                    self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.initializer_range = initializer_range
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            return cls.from_dict(json.loads(reader.read()))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"


BertLayerNorm = torch.nn.LayerNorm


def _leaf_for(name, shape, on_meta):
    """nn leaf module that owns `<name>.weight` (and `.bias`): its type drives init_bert_weights exactly as in the reference."""
    dev = "meta" if on_meta else None
    if name.endswith("_embeddings"):
        return nn.Embedding(shape[0], shape[1], padding_idx=0, device=dev)
    if "LayerNorm" in name or "layer_norm" in name:
        return BertLayerNorm(shape[0], eps=1e-12, device=dev)
    return nn.Linear(shape[1], shape[0], device=dev)


def _build_tree(root, specs, prefix):
    """Creates container modules / leaves under `root` for every spec whose name starts with `prefix`."""
    leaves = {}
    for sp in specs:
        if not sp.name.startswith(prefix):
            continue
        rel = sp.name[len(prefix):]
        path, pname = rel.rsplit(".", 1)
        if path not in leaves:
            if pname != "weight":
                continue
            parts = path.split(".")
            cur = root
            for part in parts[:-1]:
                if part not in cur._modules:
                    cur.add_module(part, nn.Module())
                cur = cur._modules[part]
            leaf = _leaf_for(parts[-1], sp.shape, on_meta=True).to_empty(device="cpu")   # storage without the default init pass
            cur.add_module(parts[-1], leaf)
            leaves[path] = leaf
    return leaves


_RAW_DATA = torch.Tensor.data          # the C-level .data descriptor


class ArenaParameter(nn.Parameter):
    """nn.Parameter whose data / .grad are views of an engine's arenas.  The ONLY behavioural difference to nn.Parameter: reading `.grad` first
    materialises a deferred `clip_grad_norm_` (lxrt.optimization.clip_grad_norm_ leaves the clip coefficient with the engine so that
    BertAdam.step can fold it into the update kernel instead of one read-modify-write of the 820-MB gradient arena; anybody who LOOKS at a
    gradient between the two calls - a logger, a test, a foreign optimizer - gets the scaled values, as after torch's in-place clip).
    Parameters become ArenaParameters by class assignment when a module tree is bound to an engine (ArenaBinding.bind): identity, registration,
    state_dict keys, isinstance(p, nn.Parameter) are untouched."""

    @property
    def grad(self):
        ref = self.__dict__.get("_rgqa_binding")
        if ref is not None:
            b = ref()
            e = b.engine if b is not None else None
            if e is not None:
                if e._pending_clip is not None:
                    e.flush_deferred_clip()
                # whoever takes the gradient view may write it in place (nn.Module.zero_grad(set_to_none=False), a foreign optimizer): the caller's
                # stream first joins an optimizer pass still reading the arena beside the forward pass (a stream-side wait; two attribute reads when none is:
                # the trainer's zero_grad() comes through here once per parameter)
                if e._upd_done is not None:
                    e.join_update()
        return _RAW_GRAD.__get__(self, type(self))

    @grad.setter
    def grad(self, value):
        _RAW_GRAD.__set__(self, value)

    @grad.deleter
    def grad(self):
        _RAW_GRAD.__delete__(self)

    # `.data` - how loggers, initialisers and checkpoint code usually reach a parameter's values - first puts the caller's stream behind an
    # optimizer pass that may still be running beside the forward pass (Engine._update_beside_forward; a stream-side wait, free when none is)
    @property
    def data(self):
        ref = self.__dict__.get("_rgqa_binding")
        if ref is not None:
            b = ref()
            if b is not None and b.engine is not None:
                b.engine.join_update()
        return _RAW_DATA.__get__(self, type(self))

    @data.setter
    def data(self, value):
        ref = self.__dict__.get("_rgqa_binding")
        if ref is not None:
            b = ref()
            if b is not None and b.engine is not None:
                b.engine.join_update()
        _RAW_DATA.__set__(self, value)

    def __reduce_ex__(self, proto):
        # pickled (torch.save(model), copy.deepcopy of a container) as a plain nn.Parameter holding the current values: the arena binding is a
        # property of the live engine, not of the tensor (and its weak reference is not picklable: ADVICE r5)
        from collections import OrderedDict
        return (torch._utils._rebuild_parameter, (self.data, self.requires_grad, OrderedDict()))


class ArenaBinding(object):
    """Keeps the nn.Parameters of a module tree as views of the engine's flat f32 arena (and their .grad as views of
    the gradient arena), re-packing when the module was moved / re-materialised, and re-casting the bf16 weight copies
    when any parameter was modified in place (load_state_dict, init, foreign optimizers)."""

    def __init__(self):
        self.engine = None
        self.params = []          # (spec, nn.Parameter)
        self._versions = None

    def bind(self, engine, named_params):
        import weakref
        self.engine = engine
        engine._binding_ref = weakref.ref(self)
        byname = dict(named_params)
        self.params = []
        for sp in engine.specs:
            if sp.name not in byname:
                raise KeyError("parameter %s missing from the module tree" % sp.name)
            p = byname[sp.name]
            if type(p) is nn.Parameter:          # (a caller's own Parameter subclass keeps its class: its reads simply never defer)
                p.__class__ = ArenaParameter
            if isinstance(p, ArenaParameter):
                p.__dict__["_rgqa_binding"] = weakref.ref(self)
            self.params.append((sp, p))

    def materialize(self, device, init_fn=None):
        """(Re)allocates the arenas on `device` and points every parameter at its slice, preserving current values."""
        e = self.engine
        old = [(sp, p.data) for sp, p in self.params]
        e.allocate(device)
        for (sp, p), (_, data) in zip(self.params, old):
            view = e.view(e.params, sp)
            view.copy_(data.to(device=e.device, dtype=torch.float32))
            p.data = view
            p.grad = None
        self._versions = None

    def packed(self):
        e = self.engine
        if e is None or e._params is None:          # (the raw field: an address check must not wait for an optimizer pass in flight)
            return False
        base = e._params.data_ptr()
        for sp, p in (self.params[0], self.params[-1], self.params[len(self.params) // 2]):
            if p.data_ptr() != base + 4 * sp.offset:
                return False
        return True

    def ensure(self, device):
        if not self.packed() or any(p.data_ptr() != self.engine._params.data_ptr() + 4 * sp.offset for sp, p in self.params):
            self.materialize(device)

    def versions(self):
        return sum(p._version for _, p in self.params)

    def sync_if_stale(self):
        v = self.versions()
        if v != self._versions:
            self.engine.sync_weights()
            self._versions = v

    def mark_synced(self):
        self._versions = self.versions()

    def in_sync(self):
        """the engine's operand copies reflect the parameters as they are now"""
        return self._versions is not None and self.versions() == self._versions

    def grads_untouched(self):
        """no .grad was written in place since attach_grads() (the per-segment sums of squares backward took are still those of .grad)"""
        gv = getattr(self, "_grad_versions", None)
        return gv is not None and gv == sum(g._version for g in (raw_grad(p) for sp, p in self.params if not sp.dead) if g is not None)

    def _views(self):
        """the gradient-arena view of every live parameter, made once per gradient arena: re-attaching them after the trainer's
        zero_grad() (which sets every .grad to None) is then 439 attribute stores instead of 439 x (slice + view) tensor constructions
        per step (~3 ms of host time at the full model: the drop-in step was host-bound on it)"""
        e = self.engine
        key = (e._grads.data_ptr(), len(self.params))
        if getattr(self, "_gv_key", None) != key:
            self._gv = [None if sp.dead else e.view(e.grads, sp) for sp, p in self.params]
            self._gv_key = key
        return self._gv

    def grads_state(self):
        """'none' if every live parameter has grad None, 'views' if they are the arena views, else 'foreign'."""
        e = self.engine
        base = e._grads.data_ptr()
        gv = self._views()
        none = views = 0
        live = 0
        for (sp, p), v in zip(self.params, gv):
            if v is None:
                continue
            live += 1
            g = raw_grad(p)
            if g is None:
                none += 1
            elif g is v or g.data_ptr() == base + 4 * sp.offset:
                views += 1
        if none == live:
            return "none"
        if views == live:
            return "views"
        return "foreign"

    def attach_grads(self):
        gv = self._views()
        ver = 0
        for (sp, p), v in zip(self.params, gv):
            if v is None:
                continue
            g = raw_grad(p)
            if g is None:
                p.grad = g = v
            ver += g._version
        self._grad_versions = ver


def _dp_rank_world():
    """(rank, world) of the default process group, (0, 1) when torch.distributed is not in use"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _dp_exchanging(world=None):
    """does a training backward exchange gradients?  More than one rank - or a process group of ONE rank under RGQA_DP_REHEARSAL=1 (bench.py's one-GPU
    rehearsal of the drop-in step on RCCL: every collective and the exchange's local arithmetic run at full size)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return False
    w = dist.get_world_size() if world is None else world
    return w > 1 or os.environ.get("RGQA_DP_REHEARSAL") == "1"


class _EngineFunction(torch.autograd.Function):
    """Autograd boundary: forward and backward of the whole encoder (+ optional fused answer head) are single engine
    calls; parameter gradients are written straight into the gradient arena that the parameters' .grad view."""

    @staticmethod
    def forward(ctx, anchor, owner, feats, boxes, ids, mask, seg, want_logits):
        lg, pl = owner._engine_forward(feats, boxes, ids, mask, seg, train=owner.training)
        # an output the loss does not depend on gets None for its gradient, not a tensor of zeros: backward() can tell which of the two paths
        # (logits / pooled output) carries a gradient without reading device memory (a `(dpooled != 0).any()` here was a host
        # synchronisation in every training step: the CPU could not enqueue the backward pass before the forward pass had finished)
        ctx.set_materialize_grads(False)
        ctx.owner = owner
        ctx.in_shapes = (tuple(feats.shape), tuple(boxes.shape))
        ctx.want_logits = want_logits
        ctx.gen = owner._fwd_counter
        out_l = lg.clone() if want_logits else lg.new_zeros(())
        return out_l, pl.clone()

    @staticmethod
    def backward(ctx, dlogits, dpooled):
        owner = ctx.owner
        if ctx.gen != owner._fwd_counter:
            raise RuntimeError("rgqa: backward() after a newer forward of the same module; the engine keeps the activations of "
                               "the last forward only")
        b = owner._binding
        state = b.grads_state()
        if state == "foreign":
            raise RuntimeError("rgqa: parameter .grad tensors were replaced by foreign tensors; use zero_grad() / set_to_none")
        acc = state == "views"
        e = b.engine
        if getattr(e, "_pending_clip", None) is not None:
            # a deferred clip_grad_norm_ nobody consumed: it scales the gradients this pass accumulates ONTO, or concerns gradients about to be overwritten
            if acc:
                e.flush_deferred_clip()
            else:
                e.drop_deferred_clip()
        # gradients w.r.t. the RoI features / boxes when the caller asked for them (ODIN: tasks/gqa_odin.py:97-121)
        want_f, want_b = ctx.needs_input_grad[2], ctx.needs_input_grad[3]
        dfeats = torch.empty(ctx.in_shapes[0][0] * ctx.in_shapes[0][1], ctx.in_shapes[0][2], dtype=torch.float32, device=e.device) if want_f else None
        dboxes = torch.empty(ctx.in_shapes[1][0] * ctx.in_shapes[1][1], ctx.in_shapes[1][2], dtype=torch.float32, device=e.device) if want_b else None
        if want_f or want_b:
            e.set_input_grads(dfeats, dboxes)
        # Data parallelism for the unchanged trainers (tasks/gqa_conf.py:111-112 `--multiGPU`, lxrt/entry.py:102-103): with
        # torch.distributed initialised (torchrun, one process per GPU) every backward averages the gradient arena over the ranks
        # before clip_grad_norm_ / BertAdam.step see it - the incoming gradient is scaled by 1/world, the arena SUM-all-reduced.
        rank, world = _dp_rank_world()
        # The exchange belongs to TRAINING steps only: a backward in eval() mode (the test-time scorers that differentiate w.r.t. the
        # inputs, tasks/gqa_odin.py:97-121, run under model.eval()) exchanges nothing - ranks may then run different numbers of passes
        # without deadlocking, and no 819-MB collective rides on a scoring pass.
        dp = _dp_exchanging(world) and owner.training
        if not dp:
            world = 1
        if dp:
            if not ctx.want_logits and not owner.__dict__.get("_dp_warned"):
                import warnings
                warnings.warn("rgqa: data parallelism exchanges the engine's gradient arena only. This backward came through the pooled "
                              "output (LXRTEncoder under a head that attach_head() did not fuse): the parameters of that head are NOT "
                              "averaged over the ranks - all-reduce their .grad yourself, or use a Linear-GeLU-LayerNorm-Linear head "
                              "(tasks/gqa_model.py:22-27), which is fused and exchanged.  See INTEGRATION.md §3.", RuntimeWarning)
                owner.__dict__["_dp_warned"] = True
            if acc:
                raise RuntimeError("rgqa: gradient accumulation over several backward() calls is not supported under data parallelism "
                                   "(the exchange would count the earlier contributions again): call zero_grad() before every backward")
            dlogits = None if dlogits is None else dlogits * (1.0 / world)
            dpooled = None if dpooled is None else dpooled * (1.0 / world)
        try:
            two = ctx.want_logits and dlogits is not None and dpooled is not None
            if two and (want_f or want_b):
                raise RuntimeError("rgqa: input gradients through both the logits and the pooled output of one forward are not supported")
            if ctx.want_logits and dlogits is not None:
                e.backward(dlogits, accumulate=acc)
                acc = True
                if two:
                    e.backward_pooled(dpooled, accumulate=True)
            elif dpooled is not None:
                e.backward_pooled(dpooled, accumulate=acc)
        finally:
            if want_f or want_b:
                e.set_input_grads(None, None)
        if dp:
            owner._dp_exchange().all_reduce()
        b.attach_grads()
        return (None, None, dfeats.view(ctx.in_shapes[0]) if want_f else None, dboxes.view(ctx.in_shapes[1]) if want_b else None, None, None, None, None)


class BertPreTrainedModel(nn.Module):
    def __init__(self, config, *inputs, **kwargs):
        super(BertPreTrainedModel, self).__init__()
        if not isinstance(config, BertConfig):
            raise ValueError(
                "Parameter config in `{}(config)` should be an instance of class `BertConfig`. ".format(self.__class__.__name__))
        self.config = config

    def init_bert_weights(self, module):
        """Same rule as the reference (modeling.py:688-699)."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()


class LXRTFeatureExtraction(BertPreTrainedModel):
    """Engine-backed LXMERT encoder. mode='x' (pooled cross-modal output) is the GQA path and the only one built."""

    HEAD_PLACEHOLDER = 64
    TREE_PREFIX = "lxrt_encoder.model."      # state_dict prefix of this module's parameters inside the task model (gqa_model.py:17-19)

    def __init__(self, config, mode='x', precision=None):
        super().__init__(config)
        if mode != 'x':
            raise NotImplementedError("rgqa_amd builds the GQA path: LXRTFeatureExtraction(mode='x'); got mode=%r" % (mode,))
        self.mode = mode
        self.precision = precision or os.environ.get("RGQA_PRECISION", "bf16")
        self.__dict__["_head"] = None          # not a registered submodule: the head belongs to GQAModel's state_dict
        self._binding = ArenaBinding()
        self._fwd_counter = 0
        self._seed_base = None
        self.__dict__["_anchor"] = None
        self._make_engine(self.HEAD_PLACEHOLDER)
        _build_tree(self, self._binding.engine.specs, self.TREE_PREFIX)
        self._rebind()
        self.apply(self.init_bert_weights)      # as LXRTModel / LXRTFeatureExtraction do (modeling.py:843, 1018)
        # an optimizer pass may still be running beside the forward pass on a stream of its own (Engine._update_beside_forward): whoever reads or
        # overwrites the parameters through nn.Module's own paths joins it first (a stream-side wait, no host synchronisation)
        self._register_state_dict_hook(lambda module, *_: module._before_state_read())
        self._register_load_state_dict_pre_hook(lambda *_: self._before_state_read())
        # the unchanged trainers' `nn.utils.clip_grad_norm_(model.parameters(), 5.)` takes the arena-aware implementation while this module is alive
        from . import optimization as _optimization
        _optimization.install_clip_routing(self)

    # -- engine / arena -------------------------------------------------------------------------------------------
    def _make_engine(self, num_answers):
        c = self.config
        self._binding.engine = Engine(vocab_size=c.vocab_size, hidden=c.hidden_size, heads=c.num_attention_heads,
                                      inter=c.intermediate_size, max_pos=c.max_position_embeddings, type_vocab=c.type_vocab_size,
                                      l_layers=VISUAL_CONFIG.l_layers, x_layers=VISUAL_CONFIG.x_layers, r_layers=VISUAL_CONFIG.r_layers,
                                      feat_dim=VISUAL_CONFIG.visual_feat_dim, pos_dim=VISUAL_CONFIG.visual_pos_dim,
                                      num_answers=num_answers, precision=self.precision,
                                      hidden_dropout=c.hidden_dropout_prob, attn_dropout=c.attention_probs_dropout_prob)

    def _named_for_binding(self):
        named = {self.TREE_PREFIX + k: v for k, v in self.named_parameters()}
        if self._head is not None:
            named.update({"logit_fc." + k: v for k, v in self._head.named_parameters()})
        else:
            # encoder-only use: the engine still owns an (unused) head; keep its slices as hidden zero parameters
            e = self._binding.engine
            for sp in e.specs:
                if sp.name.startswith("logit_fc."):
                    named[sp.name] = self.__dict__.setdefault("_hidden_head", {}).setdefault(
                        sp.name, nn.Parameter(torch.zeros(sp.shape), requires_grad=False))
        return named

    def _rebind(self):
        self._binding.bind(self._binding.engine, self._named_for_binding())

    def attach_head(self, logit_fc):
        """Fuses a GQA answer head (nn.Sequential(Linear, GeLU, LayerNorm, Linear), tasks/gqa_model.py:22-27) into the engine."""
        lin0, lin3 = logit_fc[0], logit_fc[3]
        H = self.config.hidden_size
        if not (isinstance(lin0, nn.Linear) and isinstance(lin3, nn.Linear) and isinstance(logit_fc[2], nn.LayerNorm)
                and lin0.in_features == H and lin0.out_features == 2 * H and lin3.in_features == 2 * H):
            raise ValueError("attach_head expects Linear(H,2H) -> GeLU -> LayerNorm(2H) -> Linear(2H,NA)")
        self.__dict__["_head"] = logit_fc
        self._make_engine(lin3.out_features)
        # the encoder tree's parameters keep their identity; only the binding (offsets) is rebuilt
        self._rebind()
        self.__dict__.pop("_hidden_head", None)

    def __getstate__(self):
        raise TypeError("rgqa_amd: an engine-backed module holds device arenas and a native handle and cannot be pickled whole; "
                        "save model.state_dict() (what the reference's trainers do, tasks/gqa.py save()) and load it into a new model")

    def _ready(self, device):
        # parameters moved by .cuda()/.to() (or freshly built on the host) are packed into the flat arenas here
        b = self._binding
        if b.engine._params is None or b.engine.device != device or not b.packed():
            b.materialize(device)
        else:
            b.ensure(device)

    def _dp_exchange(self):
        """The gradient exchange of this module's engine (rgqa_amd.parallel), made on first use under an initialised process group.
        RGQA_DP_MODE = `allreduce` (the drop-in default: f32 SUM all-reduce beside backward - what the reference's nn.DataParallel reduces - and the
        trainer's optimizer steps every parameter on every rank; `allreduce_bf16` opts in to the 410-MB payload) or `sharded` (round 6: the
        reduce-scatter / sharded BertAdam / weight all-gather of bench.py behind the UNCHANGED trainer: backward() leaves every rank the summed
        gradients of the 1/N of the arena it owns, `clip_grad_norm_` returns the global norm from the owners' shares, `BertAdam.step()` - this
        package's class - updates the owned 1/N and gathers the weights beside the next forward pass; `state_dict()` gathers the f32 masters and is
        then a COLLECTIVE call: every rank must make it)."""
        ex = self.__dict__.get("_dp_ex")
        if ex is None or ex.e is not self._binding.engine:
            import torch.distributed as dist
            from ..parallel import make_exchange, ShardedExchange
            mode = os.environ.get("RGQA_DP_MODE", "allreduce")
            ex = make_exchange(self._binding.engine, dist, mode=mode)
            self.__dict__["_dp_ex"] = ex
            self._binding.engine._dp_sharded = ex if isinstance(ex, ShardedExchange) else None
        return ex

    def _before_state_read(self):
        """state_dict() / load_state_dict(): behind an optimizer pass still running beside the forward pass; under the sharded exchange the f32 masters
        are current on their owner rank only and are gathered first (a collective: every rank calls state_dict())"""
        e = self._binding.engine
        e.join_update()
        ex = getattr(e, "_dp_sharded", None)
        if ex is not None and e._sharded_owner is ex:
            ex.gather_master()

    def _engine_forward(self, feats, boxes, ids, mask, seg, train):
        b = self._binding
        self._ready(feats.device)
        e = b.engine
        B, O = feats.shape[0], feats.shape[1]
        e.ensure_shape(B, ids.shape[1], O)
        if train and e._seg_sumsq is None and not _dp_exchanging():
            e.enable_segment_sumsq(True)        # clip_grad_norm_ (lxrt.optimization.clip_grad_norm_) then adds ~20 numbers instead of re-reading 0.8 GB
        b.sync_if_stale()
        if self._seed_base is None:
            # data parallel (one process per GPU): every rank needs its own dropout stream although all of them seed torch alike
            self._seed_base = (int(torch.initial_seed()) + 1000003 * _dp_rank_world()[0]) & 0x7FFFFFFFFFFF
        self._fwd_counter += 1
        lengths = self.__dict__.pop("_pending_lengths", None)
        return e.forward(feats, boxes, ids, mask, seg, train=train, seed=self._seed_base + 7919 * self._fwd_counter, lengths=lengths)

    def _run(self, feats, boxes, ids, mask, seg, want_logits, lengths=None):
        if lengths is not None:
            self.__dict__["_pending_lengths"] = lengths      # consumed by the next _engine_forward
        feats = feats.contiguous().float()
        boxes = boxes.contiguous().float()
        if feats.device.type != "cuda":
            raise RuntimeError("rgqa_amd: inputs must be on the MI355X (device 'cuda'); there is no CPU path")
        ids, mask = ids.contiguous().long(), mask.contiguous().long()
        seg = None if seg is None else seg.contiguous().long()
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != feats.device:
                self.__dict__["_anchor"] = torch.zeros(1, device=feats.device, requires_grad=True)
            return _EngineFunction.apply(self._anchor, self, feats, boxes, ids, mask, seg, want_logits)
        lg, pl = self._engine_forward(feats, boxes, ids, mask, seg, train=self.training)
        return (lg.clone() if want_logits else None), pl.clone()

    def forward(self, input_ids, token_type_ids=None, attention_mask=None, visual_feats=None, visual_attention_mask=None, token_lengths=None):
        """Same signature as the reference (modeling.py:1020-1030); returns pooled_output [B, hidden] for mode 'x'.
        token_lengths (optional, host ints; LXRTEncoder passes what its tokenizer produced): compute only the real tokens."""
        if visual_attention_mask is not None:
            raise NotImplementedError("visual_attention_mask is None on the GQA path (lxrt/entry.py:109,119)")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        feats, boxes = visual_feats
        _, pooled = self._run(feats, boxes, input_ids, attention_mask, token_type_ids, want_logits=False, lengths=token_lengths)
        return pooled

    def forward_with_head(self, input_ids, token_type_ids, attention_mask, visual_feats, token_lengths=None):
        """Encoder + fused answer head -> (logits, pooled)."""
        if self._head is None:
            raise RuntimeError("no head attached")
        feats, boxes = visual_feats
        return self._run(feats, boxes, input_ids, attention_mask, token_type_ids, want_logits=True, lengths=token_lengths)

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, state_dict=None, cache_dir=None, *inputs, **kwargs):
        """Offline counterpart of the reference loader (modeling.py:701-832): reads `bert_config.json` and
        `pytorch_model.bin` from a local directory ($RGQA_BERT_DIR or the given path) when present; otherwise builds
        bert-base defaults with random init (what `--from_scratch` does), with a warning."""
        d = pretrained_model_name_or_path if os.path.isdir(str(pretrained_model_name_or_path)) else os.environ.get("RGQA_BERT_DIR", "")
        cfg_path = os.path.join(d, "bert_config.json") if d else ""
        config = BertConfig.from_json_file(cfg_path) if cfg_path and os.path.isfile(cfg_path) else BertConfig(30522)
        model = cls(config, *inputs, **kwargs)
        wpath = os.path.join(d, "pytorch_model.bin") if d else ""
        if state_dict is None and wpath and os.path.isfile(wpath):
            state_dict = torch.load(wpath, map_location="cpu")
        model._pending_bert_state = state_dict
        if state_dict is None:
            logger.warning("no local BERT weights for '%s' (no network): LXRT encoder starts from random init",
                           pretrained_model_name_or_path)
        return model

    def load_pending_bert(self):
        sd = getattr(self, "_pending_bert_state", None)
        if sd is None:
            return
        mapped = {}
        for k, v in sd.items():
            k = k.replace("gamma", "weight").replace("beta", "bias")
            mapped[k if k.startswith("bert.") else "bert." + k] = v
        self.load_state_dict(mapped, strict=False)
        self._pending_bert_state = None
