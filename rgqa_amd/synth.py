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


# ---------------------------------------------------------------------------------------------------------------
# Fixture configurations and batches shared by the golden generator (oracle/gen_golden.py), the tests and the tools: they
# describe DATA (shapes, seeds, which rows are edge cases), so they live with the filler, not with the script that imports the
# reference.
def sample_idx(name, numel, k=64):
    k = min(k, numel)
    return (hash_u32("gradsample." + name, k) % np.uint64(numel)).astype(np.int64)


SMALL = dict(vocab_size=64, hidden=64, heads=4, inter=128, max_pos=32, type_vocab=2, l_layers=2, x_layers=2,
             r_layers=2, feat_dim=32, pos_dim=4, num_answers=11)
FULL = dict(vocab_size=30522, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=9,
            x_layers=5, r_layers=5, feat_dim=2048, pos_dim=4, num_answers=1842)


def small_batch(T):
    b = synth_batch(3, T, O=6, F=32, NA=11, vocab=64, seed=77 + T, uq_frac=0.34, min_len=2)
    b["input_ids"][1, 1:] = 0          # a 1-token question: only [CLS] survives as a real token
    b["input_ids"][1, 0] = 2
    b["input_mask"] = (b["input_ids"] != 0).astype(np.int64)
    return b


def full_batch(T):
    b = synth_batch(4, T, seed=4242 + T)
    ids = b["input_ids"]
    ids[0, :] = 0
    ids[0, 0], ids[0, 1] = 101, 102   # shortest question: [CLS][SEP]
    full = 1000 + (hash_u32("fullrow%d" % T, T) % np.uint64(29000)).astype(np.int64)
    ids[3, :] = full
    ids[3, 0], ids[3, T - 1] = 101, 102  # max-length question, no padding
    b["input_mask"] = (ids != 0).astype(np.int64)
    return b


BUTD_WORDS = "what color is the dog 's a an on in to left right man woman cat table red blue who holding bottle".split()
BUTD_SENTS = ["What color is the man's dog?", "Is the cat on the table, to the left?", "who is holding the red bottle", "zebra", ""]


U_SMALL = dict(vocab_size=64, hidden=64, heads=4, inter=128, max_pos=32, type_vocab=2, l_layers=3, x_layers=0, r_layers=0,
               feat_dim=32, pos_dim=7, num_answers=11)
U_FULL = dict(vocab_size=28996, hidden=768, heads=12, inter=3072, max_pos=512, type_vocab=2, l_layers=12, x_layers=0, r_layers=0,
              feat_dim=2048, pos_dim=7, num_answers=1842)


def uniter_batch(cfgd, T, B, O, seed):
    """Deterministic UNITER batch: a synth LXMERT batch + 7-d position features (normalised box, w, h, area: entry of
    GQATorchDataset._uniterBoxes, tasks/gqa_data.py:240-250)."""
    b = synth_batch(B, T, O=O, F=cfgd["feat_dim"], NA=cfgd["num_answers"], vocab=cfgd["vocab_size"], seed=seed, min_len=2)
    if B >= 3:
        b["input_ids"][1, 1:] = 0          # a 1-token question
        b["input_ids"][1, 0] = 2 if cfgd["vocab_size"] < 1000 else 101
        b["input_ids"][2, :] = np.maximum(b["input_ids"][2, :], 3)        # a full-length question (no padding)
        b["input_mask"] = (b["input_ids"] != 0).astype(np.int64)
    bx = b["boxes"]
    w, h = bx[:, :, 2] - bx[:, :, 0], bx[:, :, 3] - bx[:, :, 1]
    b["pos7"] = np.stack([bx[:, :, 0], bx[:, :, 1], bx[:, :, 2], bx[:, :, 3], w, h, w * h], 2).astype(np.float32)
    return b
