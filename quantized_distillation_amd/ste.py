"""Straight-through-estimator epilogues of the quantized training loops, as device kernels (CPU tensors: the same entry
points of libqd_host.so -- the tensor's device decides, _lib.lib_for).

The reference's loops do these inline with torch ops around the quantizer
(ref: cnn_models/conv_forward_model.py:240-241 `p.data.clamp_(-1, 1)`, :263-264
`p.grad.data[p.data.abs() > 1] = 0`, :266 `quantizeFunctions[idx].backward(p.grad.data)`;
translation_models/model.py:250-251,276-279 are the same lines for the seq2seq loop).  Each
function below is one launch through the C ABI (K8 / K7 of include/qd_hip.h) and works on any
contiguous fp32 tensor -- a single parameter or a flat buffer holding many.
"""
import torch

from . import _lib


def _flat(t, what):
    _lib.require_f32(t, what)
    if not t.is_contiguous():
        raise ValueError('%s must be contiguous (it is modified in place)' % what)
    return t


class _on(object):
    """`with _on(t):` -- t's HIP device current for the launch; nothing to do for a CPU tensor."""

    def __init__(self, t):
        self.ctx = torch.cuda.device(t.device) if t.is_cuda else None

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)


def clamp_(weights, limit=1.0):
    """weights.clamp_(-limit, limit) -- ref: conv_forward_model.py:240-241."""
    w = _flat(weights, 'weights')
    if w.numel():
        with _on(w):
            _lib.check(_lib.lib_for(w).qd_clamp_f32(w.data_ptr(), w.numel(), float(limit), _lib.stream_for(w)))
        _lib.mark_written(w)               # written through its raw pointer: bump the version counter as clamp_() would
    return weights


def truncated_ste_(grad, weights, limit=1.0):
    """grad[|weights| > limit] = 0 in place -- ref: conv_forward_model.py:263-264."""
    g, w = _flat(grad, 'grad'), _flat(weights, 'weights')
    if g.numel() != w.numel() or g.device != w.device:
        raise ValueError('grad and weights must have the same number of elements and live on one device')
    if g.numel():
        with _on(g):
            _lib.check(_lib.lib_for(g).qd_truncated_ste_f32(w.data_ptr(), g.data_ptr(), g.numel(), float(limit),
                                                            _lib.stream_for(g)))
        _lib.mark_written(g)
    return grad


def ste_bucket_backward(weights, grad, bucket_size, s, out=None, tie_mode='reference'):
    """The 'complicated' STE backward of uniformQuantization_variable (ref: quant_functions.py:319-406)
    without the forward's clone of the input: `weights` are the full-precision values the forward
    quantized (unchanged since).  Returns `out` (a new tensor unless given; may be `grad` itself)."""
    x, g = _flat(weights, 'weights'), _flat(grad, 'grad')
    if g.numel() != x.numel() or g.device != x.device:
        raise ValueError('grad and weights must have the same number of elements and live on one device')
    if bucket_size is None:                                                         # ref: :332-334
        raise NotImplementedError('Right now the code does not work with bucket_size None.'
                                  ' Not hard to modify though')
    given = out is not None
    if out is None:
        out = torch.empty_like(g)
    else:
        _flat(out, 'out')
        if out.numel() != g.numel():
            raise ValueError('out must have as many elements as grad')
    if x.numel():
        with _on(x):
            _lib.check(_lib.lib_for(x).qd_ste_bucket_backward_f32(
                x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel(), int(bucket_size), int(s),
                0 if tie_mode == 'reference' else 1, _lib.stream_for(x)))
        if given:
            _lib.mark_written(out)
    return out
