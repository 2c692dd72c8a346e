"""What would an f32 residual stream buy?    CPU emulation on the oracle: the engine's bf16 rounding points are
applied to the oracle's forward pass (full 9/5/5 architecture, the G2 inputs) under two policies -
  A  'engine'      : what the HIP engine does: bf16 GEMM operands AND bf16 storage of every activation (qkv, probabilities, context,
                     pre-LayerNorm sums, LayerNorm outputs, GELU outputs)
  B  'f32 stream'  : bf16 GEMM operands only; the residual stream (pre-LayerNorm sums, LayerNorm outputs) stays f32
and the logits are compared with the pure-f32 oracle.  Policy A reproducing the error the real engine shows on the same inputs
(tests/test_gpu_engine.py::test_bf16_full_config_vs_golden: max 3.8e-2, mean 8.2e-3 at T=20) validates the emulation; B is the estimate.
usage: python tools/residual_precision_study.py     (CPU, ~1 min)"""
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lxmert_ref as R            # noqa: E402
from rgqa_amd.synth import FULL, full_batch  # noqa: E402
from rgqa_amd import synth                    # noqa: E402

r = lambda t: t.bfloat16().float()


def install(policy):
    keep = (lambda t: t) if policy == "B" else r          # storage rounding of the residual stream

    def linear(x, P, name):                                # bf16 operands, f32 accumulate (both policies)
        return F.linear(r(x), r(P[name + ".weight"]), P[name + ".bias"])

    def attention(P, name, cfg, hidden, context, mask, probs_out=None):
        B, Lq, H = hidden.shape
        Lk = context.shape[1]
        nh, dh = cfg.heads, cfg.hidden // cfg.heads
        q = r(linear(hidden, P, name + ".query")).view(B, Lq, nh, dh).permute(0, 2, 1, 3)
        k = r(linear(context, P, name + ".key")).view(B, Lk, nh, dh).permute(0, 2, 1, 3)
        v = r(linear(context, P, name + ".value")).view(B, Lk, nh, dh).permute(0, 2, 1, 3)
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(dh)
        if mask is not None:
            s = s + mask
        p = r(torch.softmax(s, dim=-1))                    # P feeds the PV MFMA as bf16
        return r(torch.matmul(p, v).permute(0, 2, 1, 3).contiguous().view(B, Lq, H))

    def att_output(P, name, cfg, ctx, resid):
        z = keep(linear(ctx, P, name + ".dense") + resid)
        return keep(R_layer_norm(z, P, name + ".LayerNorm", cfg.ln_eps))

    def ffn(P, inter, output, cfg, x):
        h = r(R_gelu(linear(x, P, inter + ".dense")))
        z = keep(linear(h, P, output + ".dense") + x)
        return keep(R_layer_norm(z, P, output + ".LayerNorm", cfg.ln_eps))

    def embeddings(P, pre, cfg, input_ids, token_type_ids):
        return keep(R_embeddings(P, pre, cfg, input_ids, token_type_ids))

    def visual_embed(P, pre, cfg, feats, boxes):
        x = R_layer_norm(linear(feats, P, pre + "visn_fc"), P, pre + "visn_layer_norm", cfg.ln_eps)
        y = R_layer_norm(F.linear(boxes, P[pre + "box_fc.weight"], P[pre + "box_fc.bias"]), P, pre + "box_layer_norm", cfg.ln_eps)   # K=4 projection in f32 registers
        return keep((x + y) / 2)

    def head_forward(P, cfg, pooled):
        h = r(R_gelu(linear(r(pooled), P, "logit_fc.0")))
        h = r(R_layer_norm(h, P, "logit_fc.2", cfg.ln_eps))
        return linear(h, P, "logit_fc.3")

    R.linear, R.attention, R.att_output, R.ffn, R.embeddings, R.visual_embed, R.head_forward = linear, attention, att_output, ffn, embeddings, visual_embed, head_forward


R_layer_norm, R_gelu, R_embeddings = R.layer_norm, R.gelu, R.embeddings
ORIG = {k: getattr(R, k) for k in ("linear", "attention", "att_output", "ffn", "embeddings", "visual_embed", "head_forward")}

if __name__ == "__main__":
    torch.set_num_threads(8)
    cfg = R.RefConfig(**FULL)
    P = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(R.param_shapes(cfg)).items()}
    for T in (20, 30):
        b = {k: torch.from_numpy(v) for k, v in full_batch(T).items() if k != "lengths"}
        with torch.no_grad():
            ref, _ = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
            out = {}
            for pol in ("A", "B"):
                install(pol)
                lg, _ = R.gqa_forward(P, cfg, b["feats"], b["boxes"], b["input_ids"], b["input_mask"], b["segment_ids"])
                out[pol] = (lg - ref).abs()
                for k, v in ORIG.items():
                    setattr(R, k, v)
        print("T=%d  logits |z| mean %.2f | A (engine policy, emulated): max %.3e mean %.3e | B (f32 residual stream): max %.3e mean %.3e" % (
            T, float(ref.abs().mean()), float(out["A"].max()), float(out["A"].mean()), float(out["B"].max()), float(out["B"].mean())))
