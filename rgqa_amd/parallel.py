"""Data parallelism: one process per GPU, gradients of the flat arena all-reduced over RCCL/xGMI
(torch.distributed backend "nccl" == RCCL on ROCm; "gloo" on CPU for tests).

Replaces the reference's single-process nn.DataParallel (lxrt/entry.py:102-103): samples are independent through
the whole forward, so each rank runs its own shard and the only exchange is ONE sum all-reduce of
the gradient arena per step, issued as a few large buckets (xGMI is point-to-point: few, large collectives), with the
1/world average folded into the optimizer kernel's grad_prescale. Parameters of the final x-layer's visn branch never
receive gradients in mode 'x' and are excluded (a DDP that waits for them would hang or waste 28 MB per step)."""
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


class GradAllReduce:
    def __init__(self, engine, dist, bucket_mb=256):
        self.e, self.dist = engine, dist
        self.buckets = bucket_ranges(engine.live_ranges(), bucket_mb * (1 << 20) // 4)

    def all_reduce(self, grads=None):
        g = self.e.grads if grads is None else grads
        hs = [self.dist.all_reduce(g[a:b], op=self.dist.ReduceOp.SUM, async_op=True) for a, b in self.buckets]
        for h in hs:
            h.wait()
