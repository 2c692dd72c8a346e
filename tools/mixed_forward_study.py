"""Which parts of the forward pass could run on plain bf16 operands (ONE MFMA product) while the logits stay inside the north star's 1e-3 bound?
(round 6.)  CPU emulation on the oracle, as tools/residual_precision_study.py: everything f32 (what the split-f32 kernels deliver to 1e-4) except ONE
component at a time, whose GEMM operands are rounded to bf16:
  attn_core   q, k, v as bf16 (the QKV projection's result stored as bf16 only), probabilities as bf16 for P V   (scores, softmax, context in f32)
  qkv / att_out / ffn1 / ffn2 / visn_fc   the projection's two operands rounded to bf16 (one product instead of three)
and the logits compared with the pure-f32 oracle on the G2 inputs (full 9/5/5 architecture, B = 4).  Second table: the same with the component's ACTIVATION
operand rounded to fp16 (11 significant bits) and exact weights - what a two-product scheme a16 * (w_hi + w_lo) would compute.
usage: python tools/mixed_forward_study.py   (CPU, ~4 min)"""
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lxmert_ref as R            # noqa: E402
from rgqa_amd.synth import FULL, full_batch  # noqa: E402
from rgqa_amd import synth                    # noqa: E402

r = lambda t: t.bfloat16().float()
h16 = lambda t: t.half().float()
R_layer_norm, R_gelu = R.layer_norm, R.gelu
ORIG = {k: getattr(R, k) for k in ("linear", "attention", "att_output", "ffn", "visual_embed")}


def install(comp, fp16_act=False):
    def lin(x, P, name, low):
        if low and fp16_act:
            return F.linear(h16(x), P[name + ".weight"], P[name + ".bias"])
        if low:
            return F.linear(r(x), r(P[name + ".weight"]), P[name + ".bias"])
        return F.linear(x, P[name + ".weight"], P[name + ".bias"])

    def attention(P, name, cfg, hidden, context, mask, probs_out=None):
        B, Lq, H = hidden.shape
        Lk = context.shape[1]
        nh, dh = cfg.heads, cfg.hidden // cfg.heads
        lowp = comp == "qkv"
        q = lin(hidden, P, name + ".query", lowp)
        k = lin(context, P, name + ".key", lowp)
        v = lin(context, P, name + ".value", lowp)
        core = comp == "attn_core"
        rc = h16 if fp16_act else r
        if core:
            q, k, v = rc(q), rc(k), rc(v)
        q = q.view(B, Lq, nh, dh).permute(0, 2, 1, 3)
        k = k.view(B, Lk, nh, dh).permute(0, 2, 1, 3)
        v = v.view(B, Lk, nh, dh).permute(0, 2, 1, 3)
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        if mask is not None:
            s = s + mask
        p = torch.softmax(s, dim=-1)
        if core:
            p = rc(p)
        return torch.matmul(p, v).permute(0, 2, 1, 3).contiguous().view(B, Lq, H)

    def att_output(P, name, cfg, ctx, resid):
        return R_layer_norm(lin(ctx, P, name + ".dense", comp == "att_out") + resid, P, name + ".LayerNorm", cfg.ln_eps)

    def ffn(P, inter, output, cfg, x):
        h = R_gelu(lin(x, P, inter + ".dense", comp == "ffn1"))
        return R_layer_norm(lin(h, P, output + ".dense", comp == "ffn2") + x, P, output + ".LayerNorm", cfg.ln_eps)

    def visual_embed(P, pre, cfg, feats, boxes):
        x = R_layer_norm(lin(feats, P, pre + "visn_fc", comp == "visn_fc"), P, pre + "visn_layer_norm", cfg.ln_eps)
        y = R_layer_norm(F.linear(boxes, P[pre + "box_fc.weight"], P[pre + "box_fc.bias"]), P, pre + "box_layer_norm", cfg.ln_eps)
        return (x + y) / 2

    R.attention, R.att_output, R.ffn, R.visual_embed = attention, att_output, ffn, visual_embed


if __name__ == "__main__":
    torch.set_num_threads(8)
    cfg = R.RefConfig(**FULL)
    P = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(R.param_shapes(cfg)).items()}
    b = {k: torch.from_numpy(v) for k, v in full_batch(20).items() if k != "lengths"}
    with torch.no_grad():
        ref, _ = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
        for comp in ("attn_core", "qkv", "att_out", "ffn1", "ffn2", "visn_fc"):
            install(comp)
            lg, _ = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
            for k, v in ORIG.items():
                setattr(R, k, v)
            d = (lg - ref).abs()
            print("bf16 operands in %-9s only: logits max err %.3e mean %.3e   (bound 1e-3; all-bf16 engine 3.9e-2 / 8.3e-3)" % (comp, float(d.max()), float(d.mean())))
        for comp in ("attn_core", "qkv", "att_out", "ffn1", "ffn2", "visn_fc"):
            install(comp, fp16_act=True)
            lg, _ = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
            for k, v in ORIG.items():
                setattr(R, k, v)
            d = (lg - ref).abs()
            print("fp16 activation operand (weights exact) in %-9s only: logits max err %.3e mean %.3e" % (comp, float(d.max()), float(d.mean())))
