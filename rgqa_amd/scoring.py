"""Test-time scoring of answer logits on the device — the expressions the reference's RVQA test scripts
apply to `logit = model(feats, boxes, sent)`:

    score, label = torch.sigmoid(logit).max(1)                tasks/gqa_conf.py:344, gqa_energy.py:184, gqa_dropout.py:109
    outputs = torch.sigmoid(logit / args.temperature)         tasks/gqa_odin.py:130-131
    score = torch.log(1 + torch.exp(logit)).sum(1)            tasks/gqa_energy.py:185 (energy score; :135 in training)
    logit_k = logit.topk(k=2).values; log(1+exp(logit_k)).sum tasks/gqa_energy.py:205-206

in one fused HIP kernel (`rgqa_score_rows`, csrc/score.hip). There is no CPU fallback: without the HIP library this raises."""
import collections
import ctypes as C

import torch

from . import _lib

Scores = collections.namedtuple("Scores", "max_score label energy topk_values topk_indices topk_energy")


def score_rows(logits, temperature=1.0, k=0):
    """logits: f32 CUDA tensor [B, NA] (row-major, any row stride). Returns `Scores`; the top-k fields are None for k == 0."""
    if not (isinstance(logits, torch.Tensor) and logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 2):
        raise ValueError("score_rows: logits must be a 2-D float32 CUDA tensor")
    if logits.stride(1) != 1:
        logits = logits.contiguous()
    lib = _lib.load()
    B, NA = logits.shape
    dev = logits.device
    ms = torch.empty(B, dtype=torch.float32, device=dev)
    lab = torch.empty(B, dtype=torch.int64, device=dev)
    en = torch.empty(B, dtype=torch.float32, device=dev)
    tv = ti = te = None
    if k > 0:
        tv = torch.empty(B, k, dtype=torch.float32, device=dev)
        ti = torch.empty(B, k, dtype=torch.int64, device=dev)
        te = torch.empty(B, dtype=torch.float32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(lib.rgqa_score_rows(p(logits), logits.stride(0), B, NA, float(temperature), int(k), p(ms), p(lab), p(en), p(tv), p(ti), p(te),
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return Scores(ms, lab, en, tv, ti, te)


def sigmoid_max(logits, temperature=1.0):
    """(score, label) = torch.sigmoid(logit / temperature).max(1)"""
    s = score_rows(logits, temperature)
    return s.max_score, s.label


def energy_score(logits, k=0):
    """torch.log(1 + torch.exp(logit)).sum(1), or over the top-k logits when k > 0 (gqa_energy.py:185 / :205-206)"""
    s = score_rows(logits, 1.0, k)
    return s.topk_energy if k > 0 else s.energy
