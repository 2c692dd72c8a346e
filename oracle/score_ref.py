"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of the reference's test-time scoring of logits.

Pinned by tests/golden/g9_scores.npz, which oracle/gen_golden.py produced by evaluating the reference's own torch
expressions (cited per function) on the CPU.  numpy float32 throughout, like the reference's float32 tensors."""
import numpy as np


def sigmoid_max(logit, temperature=1.0):
    """score, label = torch.sigmoid(logit / temperature).max(1)   (tasks/gqa_conf.py:344; gqa_odin.py:130-131)
    torch's CPU max returns the first index among equal values."""
    x = (logit.astype(np.float32) / np.float32(temperature)).astype(np.float32)
    s = (np.float32(1) / (np.float32(1) + np.exp(-x, dtype=np.float32))).astype(np.float32)
    return s.max(1), s.argmax(1).astype(np.int64)


def energy(logit):
    """torch.log(1 + torch.exp(logit)).sum(1)   (tasks/gqa_energy.py:135,185) - the naive softplus: +inf above ~88.7"""
    with np.errstate(over="ignore"):
        sp = np.log(np.float32(1) + np.exp(logit.astype(np.float32), dtype=np.float32), dtype=np.float32)
    return sp.sum(1, dtype=np.float32)


def topk(logit, k):
    """logit.topk(k): values descending; equal values in ascending index order (the order the HIP kernel defines; torch
    leaves the order of ties unspecified)."""
    idx = np.lexsort((np.arange(logit.shape[1])[None, :].repeat(logit.shape[0], 0), -logit.astype(np.float64)), axis=1)[:, :k]
    return np.take_along_axis(logit, idx, 1).astype(np.float32), idx.astype(np.int64)


def topk_energy(logit, k):
    """logit_k = logit.topk(k).values; torch.log(1 + torch.exp(logit_k)).sum(1)   (tasks/gqa_energy.py:205-206)"""
    v, _ = topk(logit, k)
    return energy(v)
