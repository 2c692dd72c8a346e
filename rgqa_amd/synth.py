"""Deterministic, platform-independent synthetic weights and batches.

The reference ships no weights, data or vocabulary (SURVEY.md headline fact 4), so every
parity vector and every bench input is synthetic.  Nothing here uses a library RNG: values are
a counter-based integer hash of (crc32(tensor name), element index), so the golden generator
(which runs the reference in the build container), the oracle, the tests and the bench on the
GPU box all regenerate bit-identical tensors from names and shapes alone.
"""
import zlib

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    """murmur3 fmix32 on a uint64 array holding 32-bit values."""
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def hash_u32(name, n, salt=0):
    """n 32-bit hash words for tensor `name` (uint64 array, values < 2**32)."""
    seed = np.uint64((zlib.crc32(name.encode("utf-8")) ^ (salt * 0x9E3779B9)) & 0xFFFFFFFF)
    idx = np.arange(n, dtype=np.uint64)
    return _mix32(_mix32(idx * np.uint64(0x9E3779B1) + seed) ^ seed)


def uniform(name, shape, lo=-1.0, hi=1.0, salt=0):
    """float32 array, uniform in [lo, hi), exactly reproducible everywhere (24-bit mantissa draw)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = (hash_u32(name, n, salt) >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def fill_value(name, shape):
    """Deterministic value for the state_dict entry `name`.

    Linear / embedding weights: uniform with std 0.02 (the reference's initializer_range,
    modeling.py:186,688-699); biases: small non-zero; LayerNorm gamma ~ 1 +- 0.1, beta ~ +-0.1,
    so bias and LN-affine paths are exercised (unlike the reference's zero/one init).
    """
    leaf = name.rsplit(".", 1)[-1]
    parent = name.rsplit(".", 2)[-2] if name.count(".") >= 1 else ""
    is_ln = "LayerNorm" in parent or "layer_norm" in parent or name.startswith("logit_fc.2.")
    if is_ln:
        if leaf == "weight":
            return 1.0 + uniform(name, shape, -0.1, 0.1)
        return uniform(name, shape, -0.1, 0.1)
    if leaf == "bias":
        return uniform(name, shape, -0.05, 0.05)
    a = 0.02 * np.sqrt(3.0)
    return uniform(name, shape, -a, a)


def fill_state_dict(shapes):
    """shapes: {name: shape} -> {name: float32 ndarray}."""
    return {k: fill_value(k, tuple(s)) for k, s in shapes.items()}


def synth_batch(B, T, O=36, F=2048, NA=1842, vocab=30522, seed=1234, uq_frac=0.25, min_len=5):
    """Synthetic GQA batch of SURVEY.md §8 D1.

    feats: max(N(0,1),0)-like non-negative f32 [B,O,F] (pool5 features are post-ReLU);
    boxes: x1<x2, y1<y2 in [0,1] f32 [B,O,4]; input_ids [B,T] i64 with [CLS]=101 ... [SEP]=102 and
    zero padding, question length ~ U{min_len..T}; mask = ids != 0; segment = 0;
    target [B,NA]: one-hot score 1.0, `uq_frac` of rows all-zero (pseudo-UQ rows, gqa_conf.py:153).
    """
    tag = "synth%d" % seed
    u1 = uniform(tag + ".f1", (B, O, F), 1e-7, 1.0)
    u2 = uniform(tag + ".f2", (B, O, F), 0.0, 1.0)
    g = np.sqrt(-2.0 * np.log(u1.astype(np.float64))) * np.cos(2.0 * np.pi * u2.astype(np.float64))
    feats = np.maximum(g, 0.0).astype(np.float32)
    c = uniform(tag + ".box", (B, O, 4), 0.0, 1.0)
    x = np.sort(c[..., 0::2], axis=-1)
    y = np.sort(c[..., 1::2], axis=-1)
    boxes = np.stack([x[..., 0], y[..., 0], x[..., 1], y[..., 1]], axis=-1).astype(np.float32)
    lo = min(min_len, T)
    lens = lo + (hash_u32(tag + ".len", B) % np.uint64(T - lo + 1)).astype(np.int64)
    lo_v, hi_v = (1000, min(30000, vocab)) if vocab > 2000 else (5, vocab)
    body = lo_v + (hash_u32(tag + ".ids", B * T) % np.uint64(hi_v - lo_v)).astype(np.int64).reshape(B, T)
    cls_id, sep_id = (101, 102) if vocab > 200 else (2, 3)
    ids = np.zeros((B, T), dtype=np.int64)
    for b in range(B):
        L = int(lens[b])
        ids[b, :L] = body[b, :L]
        ids[b, 0] = cls_id
        ids[b, L - 1] = sep_id
    mask = (ids != 0).astype(np.int64)
    seg = np.zeros_like(ids)
    target = np.zeros((B, NA), dtype=np.float32)
    cls = (hash_u32(tag + ".cls", B) % np.uint64(NA)).astype(np.int64)
    uq = uniform(tag + ".uq", (B,), 0.0, 1.0) < uq_frac
    for b in range(B):
        if not uq[b]:
            target[b, cls[b]] = 1.0
    return dict(feats=feats, boxes=boxes, input_ids=ids, input_mask=mask, segment_ids=seg,
                target=target, lengths=lens)
