"""Data parallelism: one process per GPU, gradients of the flat arena all-reduced over RCCL/xGMI
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" in the CPU / single-GPU tests).

Replaces the reference's single-process nn.DataParallel (lxrt/entry.py:102-103): samples are independent through the
whole forward, so each rank runs its own shard and the only exchange is a SUM all-reduce of the
gradient arena per step, with the 1/world average folded into the optimizer kernel's grad_prescale.

The exchange overlaps with backward: the engine finalises the arena range by range (head first, embeddings last) and
records an event per range; buckets of adjacent ranges are all-reduced from a side stream that waits on those events,
so only the last bucket (embeddings + visual embedding, ~125 MB) is exposed. xGMI is point-to-point, so buckets are
kept large (>= ~64 MB). Parameters of the final x-layer's visn branch never receive gradients in mode 'x' and are in no
bucket (a DDP that waited for them would hang; exchanging them would waste 28 MB per step)."""
import os

import torch


def bucket_ranges(ranges, bucket_elems):
    """Split [(a,b), ...] element ranges into consecutive buckets of at most bucket_elems elements."""
    out = []
    for a, b in ranges:
        while a < b:
            n = min(bucket_elems, b - a)
            out.append((a, a + n))
            a += n
    return out


def merge_segments(segs, min_elems):
    """segs: [(begin, end, event)] in completion order. Grows buckets downwards in the arena (layers complete from the
    last to the first) until they hold >= min_elems elements. Returns [(begin, end, event_to_wait)] in flush order."""
    open_by_begin, done = {}, []
    for b, e, ev in segs:
        cur = open_by_begin.pop(e, None)        # an open bucket starting exactly where this segment ends
        if cur is not None:
            cur = (b, cur[1], max(ev, cur[2]))
        else:
            cur = (b, e, ev)
        if cur[1] - cur[0] >= min_elems:
            done.append(cur)
        else:
            open_by_begin[cur[0]] = cur
    done.extend(sorted(open_by_begin.values(), key=lambda c: c[2]))
    return sorted(done, key=lambda c: c[2])


class GradAllReduce:
    def __init__(self, engine, dist, bucket_mb=64, overlap=None):
        self.e, self.dist = engine, dist
        if overlap is None:
            overlap = os.environ.get("RGQA_DP_OVERLAP", "1") != "0"
        self.overlap = overlap and hasattr(engine, "grad_segments")
        if self.overlap:
            self.buckets = merge_segments(engine.grad_segments(), bucket_mb * (1 << 20) // 4)
            self.side = None
        else:
            self.buckets = [(a, b, -1) for a, b in bucket_ranges(engine.live_ranges(), 256 * (1 << 20) // 4)]

    def all_reduce(self, grads=None):
        """Call right after the engine's backward returned (its work is enqueued, not necessarily finished)."""
        g = self.e.grads if grads is None else grads
        dist = self.dist
        if hasattr(self.e, "invalidate_segment_sumsq"):
            self.e.invalidate_segment_sumsq()          # the norm must come from the REDUCED gradients
        if not self.overlap or not g.is_cuda:
            hs = [dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True) for a, b, _ in self.buckets]
            for h in hs:
                h.wait()
            return
        if self.side is None:
            self.side = torch.cuda.Stream(device=g.device)
        hs = []
        for a, b, ev in self.buckets:
            with torch.cuda.stream(self.side):
                self.e.wait_grad_event(ev, self.side)
                hs.append(dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, async_op=True))
        for h in hs:
            h.wait()        # the CURRENT stream waits for the collective; no host synchronisation
