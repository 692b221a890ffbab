"""One-launch quantization of every parameter tensor of a model (multi-tensor K1).

The reference's training loops quantize parameter by parameter
(cnn_models/conv_forward_model.py:235-247, translation_models/model.py:247-258):

    for p in model.parameters():
        p.data = quantization.uniformQuantization(p.data, s, bucket_size=...)[0]

With 22-110 tensors per model, most of them tiny, that is launch/host bound.  This class builds
the table of {master pointer, shadow pointer, numel} once and quantizes the whole model with a
single kernel launch per step (qd_multi_uniform_f32, include/qd_hip.h); the results are
bit-identical to calling uniformQuantization on each tensor.
"""
import ctypes

import torch

from . import _lib


class MultiTensorQuantizer(object):
    def __init__(self, tensors, s, bucket_size, outputs=None):
        if bucket_size is not None and (not isinstance(bucket_size, int) or bucket_size <= 0):
            raise ValueError('bucket_size must be a positive integer or None')
        if int(s) != s or s < 2:
            raise ValueError('s must be an integer >= 2')
        self.s = int(s)
        self.bucket_size = bucket_size
        self.inputs = list(tensors)
        if not self.inputs:
            raise ValueError('no tensors')
        for t in self.inputs:
            _lib.require_device_f32(t)
            if not t.is_contiguous():
                raise ValueError('multi-tensor quantization needs contiguous tensors')
        self.device = self.inputs[0].device
        if any(t.device != self.device for t in self.inputs):
            raise ValueError('all tensors of a multi-tensor launch must live on one device')
        self.outputs = list(outputs) if outputs is not None else [torch.empty_like(t) for t in self.inputs]
        if len(self.outputs) != len(self.inputs):
            raise ValueError('need one output per input')
        for t, o in zip(self.inputs, self.outputs):
            _lib.require_device_f32(o, 'output')
            if o.numel() != t.numel() or not o.is_contiguous() or o.device != t.device:
                raise ValueError('outputs must match the inputs in size and device and be contiguous')
        self._table = None
        self._ptrs = None
        self._tiles = 0
        self._plan()

    def _plan(self):
        lib = _lib.load()
        n = len(self.inputs)
        host = (_lib.QdTensorDesc * n)()
        for i, (t, o) in enumerate(zip(self.inputs, self.outputs)):
            host[i].x = t.data_ptr()
            host[i].q = o.data_ptr()
            host[i].n = t.numel()
        if self.bucket_size is None:
            self._tiles = int(lib.qd_multi_global_plan(host, n))
            self.alpha_beta = torch.empty(n, 2, dtype=torch.float32, device=self.device)     # per-tensor (alpha, beta)
            self._scratch = torch.empty(max(4, 2 * self._tiles), dtype=torch.float32, device=self.device)
        else:
            self._tiles = int(lib.qd_multi_plan(host, n, self.bucket_size))
        if self._tiles < 0:
            raise RuntimeError('qd_multi_plan failed')
        raw = bytes(host)
        self._table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._ptrs = [(t.data_ptr(), o.data_ptr()) for t, o in zip(self.inputs, self.outputs)]

    def quantize(self, check_pointers=True):
        """Quantize all tensors (one launch).  Returns the list of output tensors."""
        if check_pointers:
            for (px, pq), t, o in zip(self._ptrs, self.inputs, self.outputs):
                if t.data_ptr() != px or o.data_ptr() != pq:
                    self._plan()       # storage moved (e.g. p.data was rebound): rebuild the table
                    break
        if self._tiles <= 0:
            return self.outputs
        if _lib.on_other_device(self._table):        # launch with the tensors' device current
            with torch.cuda.device(self.device):
                return self.quantize(check_pointers=False)
        if self.bucket_size is None:
            _lib.check(_lib.load().qd_multi_uniform_global_f32(
                self._table.data_ptr(), len(self.inputs), self._tiles, self.s, self.alpha_beta.data_ptr(),
                self._scratch.data_ptr(), self._scratch.numel() * 4, _lib.stream_ptr(self.device)))
        else:
            _lib.check(_lib.load().qd_multi_uniform_f32(self._table.data_ptr(), len(self.inputs), self._tiles,
                                                        self.bucket_size, self.s, _lib.stream_ptr(self.device)))
        _lib.mark_written(self.outputs)          # written through the device table: one native call bumps their version counters
        return self.outputs


class MultiTensorDiffQuant(object):
    """Both per-step sweeps of differentiable quantization over ALL tensors of a model in one
    launch each (qd_multi_nearest_f32 / qd_multi_point_grad_f32, include/qd_hip.h).

    The reference loops over the tensors calling nonUniformQuantization_variable.forward and
    .backward (cnn_models/conv_forward_model.py:524-545).  Here the scaled weights `u`, alpha and
    beta of every tensor are computed once (K2) and stay resident; `points` is ONE [ntensors, k]
    device tensor; every step
        forward():  points -> quantized weights written straight into `outputs[i]` (+ uint8 indices)
        backward(): gradients `grads[i]` -> grad of the points, [ntensors, k]
    Results are bit-identical (forward) / equal to rounding (backward) to the per-tensor calls.
    """

    def __init__(self, tensors, outputs, grads, num_points, bucket_size):
        from .quantization.quant_functions import ScalingFunction
        if not isinstance(bucket_size, int) or bucket_size <= 0 or bucket_size & (bucket_size - 1):
            raise ValueError('the multi-tensor diff-quant path needs a power-of-two bucket_size')
        if not 1 <= num_points <= 64:
            raise ValueError('the multi-tensor diff-quant path supports 1..64 points per tensor')
        self.k, self.bucket_size = int(num_points), bucket_size
        # OWNING references: the device table below holds raw pointers into these tensors, so they are kept
        # alive here for the lifetime of the object.  A caller that rebinds `p.grad` (zero_grad(set_to_none=True))
        # does not free them; backward() then reads the buffers given HERE, which is what check_pointers guards.
        self.outputs, self.grads = list(outputs), list(grads)
        self.device = tensors[0].device
        self.scalings, self.scaled, self.indices = [], [], []
        for t, o, g in zip(tensors, self.outputs, self.grads):
            _lib.require_device_f32(t)
            for other in (o, g):
                _lib.require_device_f32(other)
                if other.numel() != t.numel() or not other.is_contiguous():
                    raise ValueError('outputs / grads must be contiguous and match the tensors in size')
                if other.device != self.device:
                    raise ValueError('all tensors of a multi-tensor launch must live on one device')
            if t.device != self.device:
                raise ValueError('all tensors of a multi-tensor launch must live on one device')
            sf = ScalingFunction('linear', False, False, bucket_size)
            u = sf.scale_down(t).view(-1)[0:t.numel()].contiguous()
            self.scalings.append(sf)
            self.scaled.append(u)
            self.indices.append(torch.empty(t.numel(), dtype=torch.uint8, device=self.device))
        self._plan()

    def _plan(self):
        bucket_size = self.bucket_size
        n = len(self.scaled)
        lib = _lib.load()
        host = (_lib.QdDiffQuantDesc * n)()
        for i in range(n):
            host[i].u = self.scaled[i].data_ptr()
            host[i].q = self.outputs[i].data_ptr()
            host[i].idx = self.indices[i].data_ptr()
            host[i].alpha = self.scalings[i].alpha.data_ptr()
            host[i].beta = self.scalings[i].beta.data_ptr()
            host[i].grad = self.grads[i].data_ptr()
            host[i].n = self.scaled[i].numel()
        blocks = ctypes.c_int64(0)
        self._tiles = int(lib.qd_multi_dq_plan(host, n, bucket_size, ctypes.byref(blocks)))
        self._blocks = int(blocks.value)
        if self._tiles < 0:
            raise RuntimeError('qd_multi_dq_plan failed')
        self._table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.device)
        self._scratch = torch.empty(max(1, self._blocks * self.k), dtype=torch.float32, device=self.device)
        self.n_tensors = n
        self._ptrs = [(o.data_ptr(), g.data_ptr()) for o, g in zip(self.outputs, self.grads)]

    def rebind(self, outputs=None, grads=None):
        """Point the device table at new output / gradient buffers (e.g. after the caller re-allocated
        its gradients) -- one small H2D copy."""
        if outputs is not None:
            self.outputs = list(outputs)
        if grads is not None:
            self.grads = list(grads)
        for u, o, g in zip(self.scaled, self.outputs, self.grads):
            for other in (o, g):
                _lib.require_device_f32(other)
                if other.numel() != u.numel() or not other.is_contiguous() or other.device != self.device:
                    raise ValueError('outputs / grads must be contiguous, on the same device and match the tensors in size')
        self._plan()

    def _check(self):
        for (po, pg), o, g in zip(self._ptrs, self.outputs, self.grads):
            if o.data_ptr() != po or g.data_ptr() != pg:
                self._plan()           # a held tensor's storage was swapped (set_, resize_): rebuild the table
                break

    def forward(self, points):
        """points: [ntensors, k] fp32 device tensor, each row sorted.  Writes outputs[i] in place."""
        if points.shape != (self.n_tensors, self.k) or not points.is_contiguous():
            raise ValueError('points must be a contiguous [ntensors, k] tensor')
        if _lib.on_other_device(self._table):
            with torch.cuda.device(self.device):
                return self.forward(points)
        self._check()
        _lib.check(_lib.load().qd_multi_nearest_f32(self._table.data_ptr(), self.n_tensors, self._tiles, self.bucket_size,
                                                    points.data_ptr(), self.k, _lib.stream_ptr(self.device)))
        _lib.mark_written(self.outputs)
        return self.outputs

    def backward(self, out=None):
        """grad of the points from the gradient buffers given at construction: [ntensors, k]."""
        if out is None:
            out = torch.empty(self.n_tensors, self.k, dtype=torch.float32, device=self.device)
        if _lib.on_other_device(self._table):
            with torch.cuda.device(self.device):
                return self.backward(out)
        self._check()
        _lib.check(_lib.load().qd_multi_point_grad_f32(self._table.data_ptr(), self.n_tensors, self._blocks,
                                                       self.bucket_size, self.k, out.data_ptr(), self._scratch.data_ptr(),
                                                       self._scratch.numel() * 4, _lib.stream_ptr(self.device)))
        return out
