"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's UNITER-GQA forward
(uniter/modeling.py:560-655 + uniter/uniter.py:15-44), built on the BertLayer restatement of oracle/lxmert_ref.py (the two
packages share that module code, uniter/modeling.py:435-557 = lxrt/modeling.py:295-435).

Pinned by tests/golden/g11_uniter_*.npz, produced by oracle/gen_golden.py from the reference's own uniter/modeling.py."""
import torch
import torch.nn.functional as F

from . import lxmert_ref as R

PRE = "encoder.model.uniter."


def param_shapes(cfg):
    H, I = cfg.hidden, cfg.inter
    out = {}

    def lin(name, o, i):
        out[name + ".weight"] = (o, i); out[name + ".bias"] = (o,)

    def ln(name, n=H):
        out[name + ".weight"] = (n,); out[name + ".bias"] = (n,)

    out[PRE + "embeddings.word_embeddings.weight"] = (cfg.vocab_size, H)
    out[PRE + "embeddings.position_embeddings.weight"] = (cfg.max_pos, H)
    out[PRE + "embeddings.token_type_embeddings.weight"] = (cfg.type_vocab, H)
    ln(PRE + "embeddings.LayerNorm")
    lin(PRE + "img_embeddings.img_linear", H, cfg.feat_dim)
    ln(PRE + "img_embeddings.img_layer_norm")
    ln(PRE + "img_embeddings.pos_layer_norm")
    lin(PRE + "img_embeddings.pos_linear", H, cfg.pos_dim)
    ln(PRE + "img_embeddings.LayerNorm")
    for i in range(cfg.l_layers):
        n = PRE + "encoder.layer.%d" % i
        for q in ("query", "key", "value"):
            lin(n + ".attention.self." + q, H, H)
        lin(n + ".attention.output.dense", H, H); ln(n + ".attention.output.LayerNorm")
        lin(n + ".intermediate.dense", I, H)
        lin(n + ".output.dense", H, I); ln(n + ".output.LayerNorm")
    lin(PRE + "pooler.dense", H, H)
    lin("logit_fc.0", 2 * H, H); ln("logit_fc.2", 2 * H); lin("logit_fc.3", cfg.num_answers, 2 * H)
    return out


def text_embeddings(P, cfg, input_ids, token_type_ids):
    """UniterTextEmbeddings.forward (uniter/modeling.py:575-591): only the word table has padding_idx=0 (:563-568)."""
    T = input_ids.shape[1]
    pos = torch.arange(T, dtype=torch.long).unsqueeze(0).expand_as(input_ids)
    e = (F.embedding(input_ids, P[PRE + "embeddings.word_embeddings.weight"], padding_idx=0)
         + F.embedding(pos, P[PRE + "embeddings.position_embeddings.weight"])
         + F.embedding(token_type_ids, P[PRE + "embeddings.token_type_embeddings.weight"]))
    return R.layer_norm(e, P, PRE + "embeddings.LayerNorm", cfg.ln_eps)


def image_embeddings(P, cfg, feats, pos7, type_emb):
    """UniterImageEmbeddings.forward (uniter/modeling.py:606-612)."""
    im = R.layer_norm(R.linear(feats, P, PRE + "img_embeddings.img_linear"), P, PRE + "img_embeddings.img_layer_norm", cfg.ln_eps)
    ps = R.layer_norm(R.linear(pos7, P, PRE + "img_embeddings.pos_linear"), P, PRE + "img_embeddings.pos_layer_norm", cfg.ln_eps)
    return R.layer_norm(im + ps + type_emb, P, PRE + "img_embeddings.LayerNorm", cfg.ln_eps)


def gqa_forward(P, cfg, feats, pos7, input_ids, input_mask, segment_ids=None, trace=None):
    """GQAUNITER.forward (uniter/uniter.py:33-44) on pre-tokenised ids: UniterEncoder.forward (entry.py:85-101: regions get token type
    1 and an all-ones mask) -> UniterModel.forward (modeling.py:622-635) -> logit_fc. Returns (logits, pooled)."""
    B, O = feats.shape[0], feats.shape[1]
    if segment_ids is None:
        segment_ids = torch.zeros_like(input_ids)
    mask = torch.cat([input_mask, torch.ones(B, O, dtype=input_mask.dtype)], 1)
    ext = (1.0 - mask.unsqueeze(1).unsqueeze(2).to(torch.float32)) * -10000.0
    txt = text_embeddings(P, cfg, input_ids, segment_ids)
    type1 = F.embedding(torch.ones(B, O, dtype=torch.long), P[PRE + "embeddings.token_type_embeddings.weight"])
    img = image_embeddings(P, cfg, feats, pos7, type1)
    x = torch.cat([txt, img], 1)
    if trace is not None:
        trace["embed"] = x
    for i in range(cfg.l_layers):
        x = R.bert_layer(P, PRE + "encoder.layer.%d" % i, cfg, x, ext)
        if trace is not None:
            trace["l%d" % i] = x
    pooled = torch.tanh(R.linear(x[:, 0], P, PRE + "pooler.dense"))
    if trace is not None:
        trace["pooled"] = pooled
    return R.head_forward(P, cfg, pooled), pooled
