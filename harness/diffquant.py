"""One step of differentiable quantization, MI355X-native.

Restates the reference's loop (ref: cnn_models/conv_forward_model.py:395-592): the weights are
frozen, the k quantization points of every tensor are trained by SGD; each step re-assigns every
weight to its nearest point (K5), runs student (quantized copy) and teacher (the original model)
forward, the KD loss, backward, and reduces the weight gradient to the k point gradients (K6).

    reference, per tensor per step                        here
    ----------------------------------------------------  ---------------------------------------
    points -> numpy; SearchSorted.query on the CPU;        one kernel: u (resident, fp32) -> q + uint8
    build int64 index + fp32 value arrays; H2D of both     index; nothing leaves the device
    k masked_select(...).sum() passes (+ k host syncs)     one two-stage segmented reduction

In data-parallel runs only the point gradients are exchanged (ntensors x k floats, ~1 KB for
WRN-16-22) instead of the 331 MB weight gradient.
"""
import copy

import torch
import torch.distributed as dist

import quantization
import quantization.help_functions as qhf

from . import models
from .flat import broadcast_from_rank0, force_collectives


class DiffQuantTrainer(object):
    def __init__(self, model, device, num_points=4, bucket_size=256, lr=1e-5, momentum=0.9, nesterov=True,
                 quantize_first_and_last_layer=True, mode='per_tensor', assign_bits_automatically=False,
                 estimate_batches=None):
        """num_points: one count for every tensor.  With assign_bits_automatically the counts are spread over
        the tensors by the 2-norm of their gradients under the plain cross-entropy loss on `estimate_batches`
        (the reference uses 5 mini-batches; ref: :424-448, help_functions.py:97-138), so tensors end up with
        different numbers of points."""
        self.device = device
        self.teacher = model.to(device).eval()                       # ref: :496 modelToQuantize.eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)
        self.student = copy.deepcopy(self.teacher).train()           # ref: :498,:516
        for p in self.student.parameters():
            p.requires_grad_(True)
        params = list(self.student.parameters())
        self.teacher_params = [p.data for p in self.teacher.parameters()]
        n = len(params)
        self.slots = [i for i in range(n) if quantize_first_and_last_layer or (i != 0 and i != n - 1)]
        self.params = params
        self.counts = [int(num_points)] * len(self.slots)
        if assign_bits_automatically:
            self.counts = self._assign_counts(estimate_batches, self.counts)
        self.k = max(self.counts)
        scaling = quantization.ScalingFunction('linear', False, False, bucket_size, False)     # ref: :421
        # all points live in ONE [ntensors, k] tensor: one optimizer state, one all-reduce.  Rows of tensors
        # with fewer than k points are padded with +inf: a point at +inf is never the nearest one (its
        # midpoint is +inf or NaN, never <= u), so it receives no weight and a zero gradient, SGD leaves it
        # where it is (inf - lr * 0) and the per-step sort keeps it at the end of the row.
        self.points = torch.full((len(self.slots), self.k), float('inf'), device=device)
        self.points_grad = torch.zeros_like(self.points)
        self.fns = []
        for row, i in enumerate(self.slots):
            w = params[i].data
            self.points[row, :self.counts[row]] = qhf.initialize_quantization_points(w, scaling, self.counts[row])   # ref: :460-462
            if mode != 'multi':
                # the reference's per-tensor objects (ref: :507-509); the multi-tensor path keeps its own resident u / alpha /
                # beta (MultiTensorDiffQuant below) -- building both would hold a second copy of u (+331 MB on WRN-16-22)
                # and run a second scale_down per tensor at setup
                self.fns.append(quantization.nonUniformQuantization_variable(
                    bucket_size=bucket_size, pre_process_tensors=True, tensor=w))
        self.points.grad = self.points_grad
        # identical replicas: the points (and the frozen weights they were initialised from) come from rank 0
        broadcast_from_rank0(self.points)
        self.mode = mode
        if mode == 'multi':
            # persistent, pointer-stable buffers: the student's weights and gradients become views of
            # two flat buffers, so the device table of the multi-tensor kernels never changes
            from harness.flat import FlatLayout
            from quantized_distillation_amd.multi_tensor import MultiTensorDiffQuant
            layout = FlatLayout([params[i].shape for i in self.slots])
            self.flat_q = torch.zeros(layout.total, device=device)
            self.flat_g = torch.zeros(layout.total, device=device)
            qs, gs = layout.views(self.flat_q), layout.views(self.flat_g)
            for row, i in enumerate(self.slots):
                params[i].data = qs[row]
                params[i].grad = gs[row]
            self.mt = MultiTensorDiffQuant([self.teacher_params[i] for i in self.slots], qs, gs, self.k, bucket_size)
        opts = dict(momentum=momentum, nesterov=nesterov) if momentum != 0 else {}
        self.opt = torch.optim.SGD([self.points], lr=lr, **opts)                                # ref: :482-484
        ready = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size() if ready else 1
        self.exchange = ready and (self.world > 1 or force_collectives())

    def _assign_counts(self, batches, counts):
        """Gradient 2-norms under the plain loss -> points per tensor (ref: conv_forward_model.py:424-448)."""
        if not batches:
            raise ValueError('assign_bits_automatically needs estimate_batches (the reference uses 5 mini-batches)')
        self.student.zero_grad()
        for images, labels in batches:                                                          # ref: :427-431
            torch.nn.functional.cross_entropy(self.student(images), labels).backward()
        norms = torch.stack([(self.params[i].grad / len(batches)).norm() for i in self.slots]).tolist()   # ref: :434-439
        self.student.zero_grad(set_to_none=True)                                                # ref: :442
        return [int(c) for c in qhf.assign_bits_automatically(norms, counts, input_is_point=True)]   # ref: :446-448

    def quantize(self):
        if self.mode == 'multi':
            self.mt.forward(self.points)                 # one launch: every tensor's weights re-assigned in place
            return
        for row, i in enumerate(self.slots):                                                    # ref: :524-532
            self.params[i].data = self.fns[row].forward(None, self.points[row, :self.counts[row]].contiguous())

    def forward_backward(self, images, labels):
        if self.mode == 'multi':
            self.flat_g.zero_()
            for i, p in enumerate(self.params):
                if p.grad is not None and i not in self.slots:
                    p.grad = None
        else:
            for p in self.params:
                p.grad = None
        out = self.student(images)
        with torch.no_grad():
            t_out = self.teacher(images)
        loss = models.kd_loss(out, t_out, labels)                                               # ref: :534-536
        loss.backward()
        return loss

    def point_gradients(self):
        if self.mode == 'multi':
            self.mt.backward(out=self.points_grad)       # one launch (+ one fold) for all tensors
            return
        for row, i in enumerate(self.slots):                                                    # ref: :538-545
            self.points_grad[row, :self.counts[row]] = self.fns[row].backward(self.params[i].grad)[1]

    def step(self, images, labels):
        self.quantize()
        loss = self.forward_backward(images, labels)
        self.point_gradients()
        if self.exchange:                                # exchange only ntensors*k floats
            dist.all_reduce(self.points_grad)
            if self.world > 1:
                self.points_grad.mul_(1.0 / self.world)
        self.opt.step()
        self.points.copy_(torch.sort(self.points, dim=1)[0])                                    # ref: :550-551
        return loss
