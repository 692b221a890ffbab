"""Flat fp32 buffers for master weights, quantized shadows and gradients, and the data-parallel
gradient synchroniser.  Pure torch (device-agnostic), so the multi-process logic is testable on
CPU with gloo."""
import torch
import torch.distributed as dist

ALIGN = 64       # elements: every tensor starts on a 256-byte boundary (16-byte vector accesses)


class FlatLayout(object):
    """Offsets of a list of shapes inside one flat buffer."""

    def __init__(self, shapes):
        self.shapes = [tuple(s) for s in shapes]
        self.numels = [int(torch.Size(s).numel()) for s in self.shapes]
        self.offsets = []
        off = 0
        for n in self.numels:
            self.offsets.append(off)
            off += -(-n // ALIGN) * ALIGN
        self.total = off

    def views(self, flat):
        return [flat[o:o + n].view(s) for o, n, s in zip(self.offsets, self.numels, self.shapes)]


class GradSynchronizer(object):
    """Data-parallel gradient exchange: ONE all-reduce (sum) of the flat fp32 gradient buffer per
    step, then a scale by 1/world -- the MI355X-native stand-in for nn.DataParallel's
    reduce-to-GPU0 + broadcast.  Over RCCL/xGMI on GPUs ("nccl" backend),
    over gloo in the CPU tests.  `chunks` > 1 splits the buffer into that many contiguous
    all-reduces (large models: lets the first chunks overlap the rest of backward when issued
    from hooks; a single call otherwise)."""

    def __init__(self, flat_grad, group=None, chunks=1):
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        n = flat_grad.numel()
        chunks = max(1, min(chunks, n))
        step = -(-n // chunks)
        self.bounds = [(i, min(i + step, n)) for i in range(0, n, step)]

    def sync(self):
        if self.world == 1:
            return
        handles = [dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                   for a, b in self.bounds]
        for h in handles:
            h.wait()
        self.flat_grad.mul_(1.0 / self.world)


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) share of `total` independent units for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
