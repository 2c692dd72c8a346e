"""Batch construction of the RoI-mixup trainer (reference tasks/gqa_mixup_vis.py:117-259) on the device.

The reference builds the second half of every batch in a Python loop over the samples on the host (clone / index-assign per
sample), then uploads 2B rows.  Here the loader's B rows are already on the MI355X; the host only makes the reference's random
draws - in the reference's order, from the same generators (`random.choice` until the partner has another image id;
`np.random.beta` + `np.random.shuffle` for mixup_v*; `random.random` for weighted_sum_*; `torch.randperm` for perturb) - and one
HIP kernel per batch writes rows [B, 2B) (rgqa_mixup_gather / rgqa_mixup_perturb / rgqa_mixup_weighted_sum + rgqa_scale_rows).

    mix = RoIMixup(mode="mixup_v1", alpha=1.0, beta=5.0)
    feats, boxes, target = mix(feats, boxes, target, img_ids)      # [2B, ...] device tensors; caller does `sent = sent + sent`
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr

MODES = ("perturb", "mixup_v1", "mixup_v2", "mixup_v3", "weighted_sum_v1", "weighted_sum_v2")


def draw_partner(img_ids, j, choice=random.choice):
    """gqa_mixup_vis.py:140-143 / 221-224: a batch index whose image differs from sample j's (rejection sampling over the batch,
    `random.choice` over a length-B sequence: the reference's draws)."""
    idx = range(len(img_ids))
    r = choice(idx)
    while img_ids[r] == img_ids[j]:
        r = choice(idx)
    return r


class RoIMixup:
    def __init__(self, mode="mixup_v1", alpha=1.0, beta=1.0):
        if mode not in MODES:
            raise ValueError(mode)       # as the reference's `raise ValueError(args.mixup_mode)`
        self.mode, self.alpha, self.beta = mode, alpha, beta
        self.lib = _lib.load()
        self._buf = {}

    def _out(self, name, like, rows):
        t = self._buf.get(name)
        shape = (rows,) + tuple(like.shape[1:])
        if t is None or t.shape != shape or t.device != like.device:
            t = torch.empty(shape, dtype=like.dtype, device=like.device)
            self._buf[name] = t
        return t

    def __call__(self, feats, boxes, target, img_ids, draws=None):
        """feats [B,O,F] f32, boxes [B,O,4] f32, target [B,NA] f32 on the device (the 'UQ' column already dropped, :122);
        img_ids: the B image ids (partner sampling).  draws: pre-recorded RNG draws (tests).  Returns 2B-row tensors owned by this
        object (overwritten by the next call)."""
        if not (feats.is_cuda and boxes.is_cuda and target.is_cuda):
            raise RuntimeError("rgqa_amd.mixup runs on the MI355X; there is no CPU path")
        B, O, F = feats.shape
        if draws is None and self.mode != "perturb" and len(set(img_ids)) < 2:
            raise ValueError("RoI-mixup needs two different images in a batch (the reference's partner loop would not terminate)")
        f2, b2, t2 = self._out("f", feats, 2 * B), self._out("b", boxes, 2 * B), self._out("t", target, 2 * B)
        f2[:B].copy_(feats)
        b2[:B].copy_(boxes)
        t2[:B].copy_(target)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        dev = feats.device
        mode = self.mode
        d = draws or {}
        if mode == "perturb":
            perm = d["perm"] if "perm" in d else torch.randperm(O).numpy()
            pg = torch.from_numpy(np.ascontiguousarray(perm, dtype=np.int32)).to(dev, non_blocking=True)
            check(self.lib.rgqa_mixup_perturb(ptr(f2), ptr(b2), ptr(pg), B, O, F, st))
            t2[B:].zero_()
            self._keep = (pg,)
            return f2, b2, t2
        if mode.startswith("mixup"):
            partner, prop, take = d.get("partner"), d.get("prop"), d.get("take")
            if partner is None:
                partner, prop, take = [], [], np.zeros((B, O), dtype=np.uint8)
                for j in range(B):        # the reference interleaves the draws per sample: choice(s), beta, shuffle
                    partner.append(draw_partner(img_ids, j))
                    p = np.random.beta(self.alpha, self.beta)
                    idx = np.arange(O)
                    np.random.shuffle(idx)
                    take[j, idx[:int(p * O)]] = 1
                    prop.append(p)
            pg = torch.from_numpy(np.ascontiguousarray(partner, dtype=np.int32)).to(dev, non_blocking=True)
            tg = torch.from_numpy(np.ascontiguousarray(take, dtype=np.uint8)).to(dev, non_blocking=True)
            check(self.lib.rgqa_mixup_gather(ptr(f2), ptr(b2), ptr(pg), ptr(tg), B, O, F, 1 if mode == "mixup_v3" else 0, st))
            self._keep = (pg, tg)
        else:
            partner, prop = d.get("partner"), d.get("prop")
            if partner is None:
                partner, prop = [], []
                for j in range(B):
                    partner.append(draw_partner(img_ids, j))
                    prop.append(random.random())
            pg = torch.from_numpy(np.ascontiguousarray(partner, dtype=np.int32)).to(dev, non_blocking=True)
            p64 = np.asarray(prop, dtype=np.float64)
            p32 = torch.from_numpy(p64.astype(np.float32)).to(dev, non_blocking=True)
            q32 = torch.from_numpy((1.0 - p64).astype(np.float32)).to(dev, non_blocking=True)     # (1 - prop) in double, then f32, as torch's scalar multiply
            check(self.lib.rgqa_mixup_weighted_sum(ptr(f2), ptr(b2), ptr(pg), ptr(p32), ptr(q32), B, O, F, st))
            self._keep = (pg, p32, q32)
        if mode in ("mixup_v1", "mixup_v3", "weighted_sum_v1"):
            sg = torch.from_numpy(np.asarray(prop, dtype=np.float64).astype(np.float32)).to(dev, non_blocking=True)
            check(self.lib.rgqa_scale_rows(ptr(t2), ptr(sg), B, t2.shape[1], t2.stride(0), B, st))
            self._keep += (sg,)
        else:
            t2[B:].zero_()
        return f2, b2, t2
