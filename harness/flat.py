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
    """Data-parallel gradient exchange: all-reduce (sum) of the flat fp32 gradient buffer, then a
    scale by 1/world -- the MI355X-native stand-in for nn.DataParallel's reduce-to-GPU0 +
    broadcast.  Over RCCL/xGMI on GPUs ("nccl" backend), over gloo in the CPU
    tests.

    chunks == 1: ONE all-reduce of the whole buffer after backward (small models).
    chunks  > 1: the buffer is cut at parameter boundaries into `chunks` contiguous pieces; with
    attach() every piece is all-reduced asynchronously AS SOON AS the gradients of all its
    parameters have been accumulated, i.e. overlapped with the rest of backward (autograd
    produces gradients roughly in reverse parameter order, so the last piece goes first).
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large pieces keep every link busy
    without paying the per-collective latency of hundreds of per-tensor reductions."""

    def __init__(self, flat_grad, group=None, chunks=1):
        self.flat_grad = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        n = flat_grad.numel()
        self.chunks = max(1, min(chunks, n))
        step = -(-n // self.chunks)
        self.bounds = [(i, min(i + step, n)) for i in range(0, n, step)]
        self._pending = None
        self._handles = []
        self._launched = []

    def attach(self, params, layout):
        """Overlap mode: register post-accumulate hooks on `params` (laid out by `layout`)."""
        total_params = len(params)
        per = -(-total_params // self.chunks)
        groups = [list(range(i, min(i + per, total_params))) for i in range(0, total_params, per)]
        self.bounds = []
        for g in groups:
            a = layout.offsets[g[0]]
            last = g[-1]
            b = layout.offsets[last + 1] if last + 1 < total_params else layout.total
            self.bounds.append((a, b))
        self._group_sizes = [len(g) for g in groups]
        self._pending = list(self._group_sizes)
        self._launched = [False] * len(groups)
        if self.world == 1:
            return
        for c, g in enumerate(groups):
            for i in g:
                params[i].register_post_accumulate_grad_hook(lambda _p, c=c: self._ready(c))

    def _launch(self, c):
        a, b = self.bounds[c]
        self._handles.append(dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))
        self._launched[c] = True

    def _ready(self, c):
        self._pending[c] -= 1
        if self._pending[c] == 0 and not self._launched[c]:
            self._launch(c)

    def sync(self):
        if self.world == 1:
            return
        if self._pending is None:                       # no hooks: everything now
            self._handles = [dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                             async_op=True) for a, b in self.bounds]
        else:
            for c in range(len(self.bounds)):           # pieces whose parameters got no gradient this step
                if not self._launched[c]:
                    self._launch(c)
            self._pending = list(self._group_sizes)
            self._launched = [False] * len(self.bounds)
        for h in self._handles:
            h.wait()
        self._handles = []
        self.flat_grad.mul_(1.0 / self.world)


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) share of `total` independent units for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
