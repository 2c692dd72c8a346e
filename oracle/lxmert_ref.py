"""ORACLE — test infrastructure only, never the product path.

CPU (PyTorch fp32) restatement of the reference's LXMERT-GQA hot path, written functionally over a
plain ``{state_dict key: tensor}`` mapping.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the shipped package ``rgqa_amd`` never
does and fails loudly without its HIP library.

Pinning: the reference holds no tests or golden vectors for this path, so this
restatement is pinned by fixtures generated in the build container by running the reference's own
Python on deterministic inputs (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``), checked by
``tests/test_oracle_golden.py``.

Each function cites the reference lines it follows (paths relative to /root/reference/src).
All arithmetic bottoms out in torch CPU ops (nn.functional.linear / softmax / layer_norm / erf),
exactly the ops the reference calls.
"""
import math
import unicodedata
from dataclasses import dataclass

import torch
import torch.nn.functional as F


@dataclass
class RefConfig:
    """BertConfig (lxrt/modeling.py:172-258) + VisualConfig (141-169) fields used on the GQA path."""
    vocab_size: int = 30522
    hidden: int = 768
    heads: int = 12
    inter: int = 3072
    max_pos: int = 512
    type_vocab: int = 2
    l_layers: int = 9
    x_layers: int = 5
    r_layers: int = 5
    feat_dim: int = 2048
    pos_dim: int = 4
    num_answers: int = 1842
    ln_eps: float = 1e-12


# --------------------------------------------------------------------------- parameter table
def param_shapes(cfg):
    """state_dict keys and shapes of GQAModel (tasks/gqa_model.py:14-28) -> ordered dict.

    Key names follow the module attribute names of lxrt/modeling.py.
    """
    H, I = cfg.hidden, cfg.inter
    out = {}
    pre = "lxrt_encoder.model.bert."

    def lin(name, o, i):
        out[name + ".weight"] = (o, i)
        out[name + ".bias"] = (o,)

    def ln(name, n=H):
        out[name + ".weight"] = (n,)
        out[name + ".bias"] = (n,)

    def att(name, ):
        lin(name + ".query", H, H)
        lin(name + ".key", H, H)
        lin(name + ".value", H, H)

    def att_out(name):
        lin(name + ".dense", H, H)
        ln(name + ".LayerNorm")

    def bert_layer(name):
        att(name + ".attention.self")
        att_out(name + ".attention.output")
        lin(name + ".intermediate.dense", I, H)
        lin(name + ".output.dense", H, I)
        ln(name + ".output.LayerNorm")

    out[pre + "embeddings.word_embeddings.weight"] = (cfg.vocab_size, H)
    out[pre + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    out[pre + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab, H)
    ln(pre + "embeddings.LayerNorm")
    enc = pre + "encoder."
    lin(enc + "visn_fc.visn_fc", H, cfg.feat_dim)
    ln(enc + "visn_fc.visn_layer_norm")
    lin(enc + "visn_fc.box_fc", H, cfg.pos_dim)
    ln(enc + "visn_fc.box_layer_norm")
    for i in range(cfg.l_layers):
        bert_layer(enc + "layer.%d" % i)
    for i in range(cfg.x_layers):
        x = enc + "x_layers.%d" % i
        att(x + ".visual_attention.att")
        att_out(x + ".visual_attention.output")
        for m in ("lang", "visn"):
            att(x + ".%s_self_att.self" % m)
            att_out(x + ".%s_self_att.output" % m)
        for m in ("lang", "visn"):
            lin(x + ".%s_inter.dense" % m, I, H)
            lin(x + ".%s_output.dense" % m, H, I)
            ln(x + ".%s_output.LayerNorm" % m)
    for i in range(cfg.r_layers):
        bert_layer(enc + "r_layers.%d" % i)
    lin(pre + "pooler.dense", H, H)
    lin("logit_fc.0", 2 * H, H)
    ln("logit_fc.2", 2 * H)
    lin("logit_fc.3", cfg.num_answers, 2 * H)
    return out


# --------------------------------------------------------------------------- elementary ops
def gelu(x):
    """Exact-erf GeLU (lxrt/modeling.py:112-118)."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def layer_norm(x, P, name, eps):
    """BertLayerNorm = torch.nn.LayerNorm, eps 1e-12 everywhere (lxrt/modeling.py:261,275,354,408)."""
    return F.layer_norm(x, (x.shape[-1],), P[name + ".weight"], P[name + ".bias"], eps)


def linear(x, P, name):
    return F.linear(x, P[name + ".weight"], P[name + ".bias"])


def attention(P, name, cfg, hidden, context, mask, probs_out=None):
    """BertAttention.forward (lxrt/modeling.py:320-347), dropout in eval mode."""
    B, Lq, H = hidden.shape
    Lk = context.shape[1]
    nh, dh = cfg.heads, cfg.hidden // cfg.heads
    q = linear(hidden, P, name + ".query").view(B, Lq, nh, dh).permute(0, 2, 1, 3)
    k = linear(context, P, name + ".key").view(B, Lk, nh, dh).permute(0, 2, 1, 3)
    v = linear(context, P, name + ".value").view(B, Lk, nh, dh).permute(0, 2, 1, 3)
    s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
    if mask is not None:
        s = s + mask
    p = torch.softmax(s, dim=-1)
    if probs_out is not None:
        probs_out.append(p)
    c = torch.matmul(p, v).permute(0, 2, 1, 3).contiguous().view(B, Lq, H)
    return c


def att_output(P, name, cfg, ctx, resid):
    """BertAttOutput.forward (lxrt/modeling.py:357-361)."""
    return layer_norm(linear(ctx, P, name + ".dense") + resid, P, name + ".LayerNorm", cfg.ln_eps)


def self_att(P, name, cfg, x, mask):
    """BertSelfattLayer.forward (lxrt/modeling.py:382-386); sub-modules `.self`, `.output`."""
    return att_output(P, name + ".output", cfg, attention(P, name + ".self", cfg, x, x, mask), x)


def cross_att(P, name, cfg, x, ctx, ctx_mask, probs_out=None):
    """BertCrossattLayer.forward (lxrt/modeling.py:370-373); sub-modules `.att`, `.output`. probs_out collects the
    attention probabilities the way lxrt_vis/modeling.py:373-376 hands them back with output_attention=True."""
    return att_output(P, name + ".output", cfg, attention(P, name + ".att", cfg, x, ctx, ctx_mask, probs_out), x)


def ffn(P, inter, output, cfg, x):
    """BertIntermediate (lxrt/modeling.py:398-401) then BertOutput (411-415)."""
    h = gelu(linear(x, P, inter + ".dense"))
    return layer_norm(linear(h, P, output + ".dense") + x, P, output + ".LayerNorm", cfg.ln_eps)


def bert_layer(P, name, cfg, x, mask):
    """BertLayer.forward (lxrt/modeling.py:425-429)."""
    a = self_att(P, name + ".attention", cfg, x, mask)
    return ffn(P, name + ".intermediate", name + ".output", cfg, a)


def x_layer(P, name, cfg, lang, lang_mask, visn, visn_mask, att_out=None):
    """LXRTXLayer.forward (lxrt/modeling.py:477-488): cross (shared weights, both from the OLD
    lang/visn, 455-459) -> per-modality self attention (461-465) -> per-modality FFN (467-475)."""
    pl, pv = ([], []) if att_out is not None else (None, None)
    l1 = cross_att(P, name + ".visual_attention", cfg, lang, visn, visn_mask, pl)
    v1 = cross_att(P, name + ".visual_attention", cfg, visn, lang, lang_mask, pv)
    if att_out is not None:      # lxrt_vis/modeling.py:458-462: (l2v_att, v2l_att) of this layer
        att_out.append((pl[0], pv[0]))
    l2 = self_att(P, name + ".lang_self_att", cfg, l1, lang_mask)
    v2 = self_att(P, name + ".visn_self_att", cfg, v1, visn_mask)
    l3 = ffn(P, name + ".lang_inter", name + ".lang_output", cfg, l2)
    v3 = ffn(P, name + ".visn_inter", name + ".visn_output", cfg, v2)
    return l3, v3


def embeddings(P, pre, cfg, input_ids, token_type_ids):
    """BertEmbeddings.forward (lxrt/modeling.py:278-292)."""
    T = input_ids.shape[1]
    pos = torch.arange(T, dtype=torch.long, device=input_ids.device).unsqueeze(0).expand_as(input_ids)
    # all three tables are built with padding_idx=0 (269-271): row 0 of each receives NO gradient —
    # i.e. the [PAD] word, position 0 (the [CLS] slot) and token type 0 (every token on this path).
    e = (F.embedding(input_ids, P[pre + "word_embeddings.weight"], padding_idx=0)
         + F.embedding(pos, P[pre + "position_embeddings.weight"], padding_idx=0)
         + F.embedding(token_type_ids, P[pre + "token_type_embeddings.weight"], padding_idx=0))
    return layer_norm(e, P, pre + "LayerNorm", cfg.ln_eps)


def visual_embed(P, pre, cfg, feats, boxes):
    """VisualFeatEncoder.forward (lxrt/modeling.py:507-517)."""
    x = layer_norm(linear(feats, P, pre + "visn_fc"), P, pre + "visn_layer_norm", cfg.ln_eps)
    y = layer_norm(linear(boxes, P, pre + "box_fc"), P, pre + "box_layer_norm", cfg.ln_eps)
    return (x + y) / 2


def encoder_forward(P, cfg, input_ids, token_type_ids, attention_mask, feats, boxes, trace=None):
    """LXRTModel.forward (lxrt/modeling.py:845-886) + LXRTEncoder.forward (546-566), mode 'x',
    visual_attention_mask=None (lxrt/entry.py:109-120).  Returns (lang, visn, pooled)."""
    pre = "lxrt_encoder.model.bert."
    ext = attention_mask.unsqueeze(1).unsqueeze(2).to(torch.float32)
    ext = (1.0 - ext) * -10000.0
    lang = embeddings(P, pre + "embeddings.", cfg, input_ids, token_type_ids)
    visn = visual_embed(P, pre + "encoder.visn_fc.", cfg, feats, boxes)
    if trace is not None:
        trace["embed_lang"], trace["embed_visn"] = lang, visn
    for i in range(cfg.l_layers):
        lang = bert_layer(P, pre + "encoder.layer.%d" % i, cfg, lang, ext)
        if trace is not None:
            trace["l%d" % i] = lang
    for i in range(cfg.r_layers):
        visn = bert_layer(P, pre + "encoder.r_layers.%d" % i, cfg, visn, None)
        if trace is not None:
            trace["r%d" % i] = visn
    for i in range(cfg.x_layers):
        att = [] if trace is not None else None
        lang, visn = x_layer(P, pre + "encoder.x_layers.%d" % i, cfg, lang, ext, visn, None, att)
        if trace is not None:
            trace["x%d_lang" % i], trace["x%d_visn" % i] = lang, visn
            trace["x%d_l2v" % i], trace["x%d_v2l" % i] = att[0]       # [B, heads, T, O], [B, heads, O, T] (lxrt_vis/modeling.py:564-572)
    pooled = torch.tanh(linear(lang[:, 0], P, pre + "pooler.dense"))  # BertPooler (575-581)
    return lang, visn, pooled


def head_forward(P, cfg, pooled):
    """GQAModel.logit_fc (tasks/gqa_model.py:22-27,41): Linear -> GeLU -> LN(eps 1e-12) -> Linear."""
    h = gelu(linear(pooled, P, "logit_fc.0"))
    h = layer_norm(h, P, "logit_fc.2", cfg.ln_eps)
    return linear(h, P, "logit_fc.3")


def gqa_forward(P, cfg, feats, boxes, input_ids, input_mask, segment_ids=None, trace=None):
    """GQAModel.forward (tasks/gqa_model.py:30-43) on pre-tokenised ids. Returns (logits, pooled)."""
    if segment_ids is None:
        segment_ids = torch.zeros_like(input_ids)
    _, _, pooled = encoder_forward(P, cfg, input_ids, segment_ids, input_mask, feats, boxes, trace)
    if trace is not None:
        trace["pooled"] = pooled
    return head_forward(P, cfg, pooled), pooled


# --------------------------------------------------------------------------- loss / batch logic
def bce_loss(logit, target):
    """tasks/gqa_conf.py:197-198: BCEWithLogitsLoss() (mean over B*NA) times logit.size(1)."""
    return F.binary_cross_entropy_with_logits(logit, target) * logit.size(1)


def drop_uq_column(target):
    """tasks/gqa_conf.py:153 — the appended 'UQ' answer column is dropped from the target."""
    return target[:, :-1]


def roi_mixup(feats, boxes, target, partner, prop, idx_lists, mode="mixup_v1"):
    """RoI-mixup batch construction (tasks/gqa_mixup_vis.py:134-181) given the RNG draws.

    partner[j]: index of the other-image sample; prop[j] ~ Beta(a, b) (147); idx_lists[j] the
    first int(prop*O) entries of a shuffled arange(O) (150-153)."""
    pf, pb, pt = [], [], []
    for j in range(feats.shape[0]):
        r = int(partner[j])
        idx = torch.as_tensor(idx_lists[j], dtype=torch.long)
        f = torch.zeros_like(feats[r]) if mode == "mixup_v3" else feats[r].clone()
        f[idx] = feats[j][idx]
        b = boxes[r].clone()
        b[idx] = boxes[j][idx]
        pf.append(f)
        pb.append(b)
        if mode in ("mixup_v1", "mixup_v3"):
            pt.append(target[j] * float(prop[j]))
        elif mode == "mixup_v2":
            pt.append(target[j] * 0)
        else:
            raise ValueError(mode)
    return (torch.cat([feats, torch.stack(pf, 0)], 0), torch.cat([boxes, torch.stack(pb, 0)], 0),
            torch.cat([target, torch.stack(pt, 0)], 0))


def perturb_batch(feats, boxes, target, perm):
    """'perturb' batch construction (tasks/gqa_mixup_vis.py:124-133): the second half repeats the features with the boxes of every
    sample permuted by ONE random permutation of the RoIs (`perm` = the torch.randperm draw) and all-zero targets."""
    perm = torch.as_tensor(perm, dtype=torch.long)
    return (feats.repeat(2, 1, 1), torch.cat([boxes, boxes[:, perm, :]], 0), torch.cat([target, torch.zeros_like(target)], 0))


def weighted_sum_batch(feats, boxes, target, partner, prop, mode="weighted_sum_v1"):
    """'weighted_sum_v1/v2' (tasks/gqa_mixup_vis.py:217-244): second half = feat_pos * prop + feat_neg * (1 - prop) with
    prop = random.random() (a Python float: each product is rounded to f32 before the sum), boxes repeated, targets scaled by
    prop (v1) or zero (v2)."""
    pf, pt = [], []
    for j in range(feats.shape[0]):
        p = float(prop[j])
        pf.append(feats[j] * p + feats[int(partner[j])] * (1 - p))
        if mode == "weighted_sum_v1":
            pt.append(target[j] * p)
        elif mode == "weighted_sum_v2":
            pt.append(target[j] * 0)
        else:
            raise ValueError(mode)
    return (torch.cat([feats, torch.stack(pf, 0)], 0), boxes.repeat(2, 1, 1), torch.cat([target, torch.stack(pt, 0)], 0))


def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_(params, 5.) semantics (tasks/gqa_conf.py:201): global L2 norm over
    grads that are not None, scale by max_norm/(norm+1e-6) when that is < 1. Returns the norm."""
    gs = [g for g in grads if g is not None]
    total = torch.sqrt(sum((g.detach().double() ** 2).sum() for g in gs)).float()
    coef = max_norm / (total + 1e-6)
    if coef < 1:
        for g in gs:
            g.mul_(coef)
    return total


def warmup_linear(x, warmup=0.002):
    """lxrt/optimization.py:38-43."""
    if x < warmup:
        return x / warmup
    return max((x - 1.0) / (warmup - 1.0), 0)


class BertAdamRef:
    """BertAdam.step (lxrt/optimization.py:101-180): no bias correction, eps outside sqrt, decoupled
    weight decay added to the update for EVERY param, per-param step counter read before increment
    (so the first update has lr 0 under warmup), params with grad None skipped."""

    def __init__(self, params, lr, warmup=-1, t_total=-1, b1=0.9, b2=0.999, e=1e-6, weight_decay=0.01):
        self.params = list(params)
        self.lr, self.warmup, self.t_total = lr, warmup, t_total
        self.b1, self.b2, self.e, self.wd = b1, b2, e, weight_decay
        self.state = {}

    def step(self, grads):
        for i, (p, g) in enumerate(zip(self.params, grads)):
            if g is None:
                continue
            st = self.state.setdefault(i, dict(step=0, m=torch.zeros_like(p), v=torch.zeros_like(p)))
            st["m"].mul_(self.b1).add_(g, alpha=1 - self.b1)
            st["v"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            upd = st["m"] / (st["v"].sqrt() + self.e)
            if self.wd > 0.0:
                upd = upd + self.wd * p
            if self.t_total != -1:
                lr_t = self.lr * warmup_linear(st["step"] / self.t_total, self.warmup)
            else:
                lr_t = self.lr
            p.add_(-lr_t * upd)
            st["step"] += 1


# --------------------------------------------------------------------------- tokenizer
def _is_ws(ch):
    return ch in " \t\n\r" or unicodedata.category(ch) == "Zs"


def _is_ctrl(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punct(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF
            or 0x2A700 <= cp <= 0x2B73F or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF
            or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


_NEVER_SPLIT = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")


def basic_tokenize(text):
    """BasicTokenizer.tokenize, do_lower_case=True (lxrt/tokenization.py:188-207, 209-239, 284-295)."""
    cleaned = []
    for ch in text:
        cp = ord(ch)
        if cp == 0 or cp == 0xFFFD or _is_ctrl(ch):
            continue
        if _is_ws(ch):
            cleaned.append(" ")
        elif _is_cjk(cp):
            cleaned.extend((" ", ch, " "))
        else:
            cleaned.append(ch)
    pieces = []
    for tok in "".join(cleaned).split():
        if tok not in _NEVER_SPLIT:
            tok = tok.lower()
            tok = "".join(c for c in unicodedata.normalize("NFD", tok) if unicodedata.category(c) != "Mn")
        if tok in _NEVER_SPLIT:
            pieces.append(tok)
            continue
        cur = ""
        for ch in tok:
            if _is_punct(ch):
                if cur:
                    pieces.append(cur)
                pieces.append(ch)
                cur = ""
            else:
                cur += ch
        if cur:
            pieces.append(cur)
    return " ".join(pieces).split()


def wordpiece(token, vocab, unk="[UNK]", max_chars=100):
    """WordpieceTokenizer.tokenize on one basic token (lxrt/tokenization.py:298-348)."""
    if len(token) > max_chars:
        return [unk]
    out, start = [], 0
    while start < len(token):
        end, found = len(token), None
        while start < end:
            sub = token[start:end]
            if start > 0:
                sub = "##" + sub
            if sub in vocab:
                found = sub
                break
            end -= 1
        if found is None:
            return [unk]
        out.append(found)
        start = end
    return out


def sents_to_features(sents, max_seq_length, vocab):
    """convert_sents_to_features (lxrt/entry.py:36-71) -> (ids, mask, segment) nested lists."""
    ids, masks, segs = [], [], []
    for s in sents:
        toks = [w for t in basic_tokenize(s.strip()) for w in wordpiece(t, vocab)]
        toks = ["[CLS]"] + toks[: max_seq_length - 2] + ["[SEP]"]
        row = [vocab[t] for t in toks]
        pad = max_seq_length - len(row)
        ids.append(row + [0] * pad)
        masks.append([1] * len(row) + [0] * pad)
        segs.append([0] * max_seq_length)
    return ids, masks, segs


# --------------------------------------------------------------------------- whole train step (CPU baseline)
def train_step(P, cfg, batch, opt, max_norm=5.0):
    """One reference train step in eval-free form (tasks/gqa_conf.py:174-202): forward, BCE x NA, backward,
    clip_grad_norm_(5.), BertAdam.step.  P: {key: leaf tensor requiring grad}; opt: BertAdamRef over list(P.values()).
    Dropout is not applied (the timing difference is negligible on CPU). Returns the loss value."""
    for p in P.values():
        p.grad = None
    logits, _ = gqa_forward(P, cfg, batch["feats"], batch["boxes"], batch["input_ids"], batch["input_mask"], batch.get("segment_ids"))
    loss = bce_loss(logits, batch["target"])
    loss.backward()
    grads = [p.grad for p in P.values()]
    clip_grad_norm(grads, max_norm)
    with torch.no_grad():
        opt.step(grads)
    return float(loss.detach())
