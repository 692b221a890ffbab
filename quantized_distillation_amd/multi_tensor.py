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
        if bucket_size is None or not isinstance(bucket_size, int) or bucket_size <= 0:
            raise ValueError('the multi-tensor path needs a positive integer bucket_size')
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
        if self._tiles > 0:
            _lib.check(_lib.load().qd_multi_uniform_f32(self._table.data_ptr(), len(self.inputs), self._tiles,
                                                        self.bucket_size, self.s, _lib.stream_ptr()))
        return self.outputs
