"""Flat fp32 buffers for master weights, quantized shadows and gradients, and the data-parallel
gradient synchroniser.  Pure torch (device-agnostic), so the multi-process logic is testable on
CPU with gloo."""
import os

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

    def end(self, i):
        """One past the last element of slot i including its alignment padding."""
        return self.offsets[i + 1] if i + 1 < len(self.offsets) else self.total


def force_collectives():
    """QD_FORCE_DIST=1: issue the collectives even in a single-rank group, so that the RCCL call
    path (communicator, stream ordering, async handles) is executed on a one-GPU box."""
    return os.environ.get('QD_FORCE_DIST') == '1'


class GradSynchronizer(object):
    """Data-parallel gradient exchange: all-reduce (MEAN) of the flat fp32 gradient buffer -- the MI355X-native stand-in
    for nn.DataParallel's reduce-to-GPU0 + broadcast (ref: resnet34_doublefilters.py:69-70,81-82,
    translation_models/model.py:47-48).  Over RCCL/xGMI on GPUs ("nccl" backend): ReduceOp.AVG, the 1/world folded into
    the collective (a separate scaling pass over a 103-331 MB gradient buffer is a full extra read + write of HBM);
    over gloo in the CPU tests, which has no AVG: sum, then scale.

    chunks == 1: ONE all-reduce of the whole buffer after backward (small models).
    chunks  > 1 without attach(): the buffer is cut into `chunks` equal pieces, all launched
    asynchronously after backward.
    attach(): OVERLAP mode.  Every parameter gets a post-accumulate-grad hook.  The first
    backward only RECORDS the order in which gradients become ready (rank 0's order is
    broadcast so that every rank forms the same plan); the arrival sequence is then cut into
    `chunks` groups of about equal bytes, and from the second step on a group's ranges are
    all-reduced asynchronously as soon as its last gradient has been accumulated, i.e.
    overlapped with the rest of backward.  Forming the groups from the observed order matters:
    parameters()[0] of the reference's ConvolForwardNet is the OUTPUT layer (ref:
    conv_forward_model.py:124), whose gradient arrives first -- groups cut in parameter order
    would put it together with the first conv layer, whose gradient arrives last.
    xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large pieces keep every link busy
    without paying the per-collective latency of hundreds of per-tensor reductions."""

    def __init__(self, flat_grad, group=None, chunks=1, force=None):
        self.flat_grad = flat_grad
        self.group = group
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if ready else 1
        force = force_collectives() if force is None else force
        self.active = ready and (self.world > 1 or force)
        self.world_active = self.active      # `active` may be switched off and on between steps; this is what it returns to
        self._avg = bool(ready and dist.get_backend(group) == 'nccl')     # RCCL reduces to the mean itself
        n = flat_grad.numel()
        self.chunks = max(1, min(chunks, n))
        step = -(-n // self.chunks)
        self.bounds = [(i, min(i + step, n)) for i in range(0, n, step)]
        self.collectives_issued = 0
        # overlap mode state
        self._layout = None
        self._numels = None
        self._order = None            # arrival order being recorded during the first backward
        self._group_of = None         # param index -> group
        self._group_ranges = None     # group -> [(a, b), ...]
        self._pending = None
        self._group_sizes = None
        self._launched = None
        self._handles = []

    # ------------------------------------------------------------------ overlap mode
    def attach(self, params, layout):
        """Overlap mode: register post-accumulate hooks on `params` (laid out by `layout`)."""
        self._layout = layout
        self._numels = list(layout.numels)
        self._order = []
        if not self.active:
            return
        for i, p in enumerate(params):
            p.register_post_accumulate_grad_hook(lambda _p, i=i: self._ready(i))

    def _ready(self, i):
        if not self.active:                  # switched off between steps (bench.py's step-without-exchange legs)
            return
        if self._group_of is None:
            self._order.append(i)
            return
        c = self._group_of[i]
        self._pending[c] -= 1
        if self._pending[c] == 0 and not self._launched[c]:
            self._launch(c)

    def _plan(self):
        """Cut the recorded arrival order into groups of ~equal bytes; merge each group's
        parameters into contiguous ranges of the flat buffer."""
        nparams = len(self._numels)
        order = list(dict.fromkeys(self._order))                  # first arrival of each parameter
        order += [i for i in range(nparams) if i not in set(order)]     # parameters that got no gradient
        if self.world > 1:                                        # every rank must cut the same groups
            t = torch.tensor(order, dtype=torch.int64, device=self.flat_grad.device)
            dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0,
                           group=self.group)
            order = [int(v) for v in t.tolist()]
        total = float(sum(self._numels)) or 1.0
        groups, acc, cur = [], 0.0, []
        for i in order:
            cur.append(i)
            acc += self._numels[i]
            if acc >= total * (len(groups) + 1) / self.chunks and len(groups) < self.chunks - 1:
                groups.append(cur)
                cur = []
        if cur:
            groups.append(cur)
        self._group_of = {}
        self._group_ranges = []
        for c, g in enumerate(groups):
            ranges = []
            for i in sorted(g):
                self._group_of[i] = c
                a, b = self._layout.offsets[i], self._layout.end(i)
                if ranges and ranges[-1][1] == a:
                    ranges[-1] = (ranges[-1][0], b)
                else:
                    ranges.append((a, b))
            self._group_ranges.append(ranges)
        self._group_sizes = [len(g) for g in groups]
        self._pending = list(self._group_sizes)
        self._launched = [False] * len(groups)
        self.bounds = [r for ranges in self._group_ranges for r in ranges]

    def _all_reduce_async(self, a, b):
        self.collectives_issued += 1
        return dist.all_reduce(self.flat_grad[a:b], op=dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM,
                               group=self.group, async_op=True)

    def _launch(self, c):
        for a, b in self._group_ranges[c]:
            self._handles.append(self._all_reduce_async(a, b))
        self._launched[c] = True

    # ------------------------------------------------------------------ per step
    def sync(self):
        if not self.active:
            return
        if self._layout is None:                        # no hooks: everything now
            self._handles = [self._all_reduce_async(a, b) for a, b in self.bounds]
        elif self._group_of is None:                    # first step of overlap mode: plan, reduce un-overlapped
            self._plan()
            for c in range(len(self._group_ranges)):
                self._launch(c)
        else:
            for c in range(len(self._group_ranges)):    # groups whose parameters got no gradient this step
                if not self._launched[c]:
                    self._launch(c)
        for h in self._handles:
            h.wait()
        self._handles = []
        if self._group_of is not None:
            self._pending = list(self._group_sizes)
            self._launched = [False] * len(self._group_ranges)
        if self.world > 1 and not self._avg:
            self.flat_grad.mul_(1.0 / self.world)


def broadcast_from_rank0(*tensors, group=None):
    """Make every replica start from rank 0's values (weights, buffers, quantization points).
    nn.DataParallel re-replicates from GPU 0 every step (ref: resnet34_doublefilters.py:69-70);
    here the replicas are kept identical by construction (same gradient, same update), so one
    broadcast at setup is what it takes."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    if dist.get_world_size(group) == 1 and not force_collectives():
        return
    src = dist.get_global_rank(group, 0) if group is not None else 0
    for t in tensors:
        dist.broadcast(t, src=src, group=group)


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) share of `total` independent units for `rank`."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)
