"""Host-side mirror of the reference's quantization/quant_functions.py on top of libqd_hip.so.

Same public names, argument meaning, return tuples and exceptions as the reference
(antspy/quantized_distillation, quantization/quant_functions.py -- cited as `ref:` below), so that
`train_model(..., quantizeWeights=True)` (cnn_models/conv_forward_model.py:165-393,
translation_models/model.py:161-317) and `optimize_quantization_points` (:395-592 / :319-442)
can call it unchanged.  What differs is *how* it runs: each call is one (at most three) HIP
kernel launches through the C ABI of include/qd_hip.h instead of ~12 unfused torch ops, there
is no host synchronisation and no device<->host round trip anywhere on the per-step path.

Tensors must be float32.  A tensor on a HIP device is computed by libqd_hip.so; a CPU tensor -- the reference's
functions accept those too (ref: :186,254,283-284) -- by libqd_host.so, the same entry points for host pointers
(csrc/host/qd_host.cpp), with results on the CPU.  The tensor's device decides, nothing else does: there is no fallback
from one to the other (_lib.lib_for).
"""
import numbers

import torch

from .. import _lib

_STOCHASTIC_CALLS = [0]


def next_stochastic_seed(peek=False, host=False):
    """Seed of the next stochastic-rounding launch.  The reference draws torch.rand on the host
    (ref: :185-186); here the draws come from an in-kernel counter-based generator keyed by this
    seed and the element index.  The seed is derived from the CURRENT DEVICE's default generator
    seed (so torch.manual_seed / torch.cuda.manual_seed control it) and a per-process call counter
    (so successive calls draw different numbers)."""
    calls = _STOCHASTIC_CALLS[0] + 1
    if not peek:
        _STOCHASTIC_CALLS[0] = calls
    base = torch.initial_seed() if host else torch.cuda.initial_seed()       # (CPU tensors: the CPU generator's seed)
    return (base * 0x9E3779B97F4A7C15 + calls) & 0xFFFFFFFFFFFFFFFF


def _bucket_arg(bucket_size):
    return 0 if bucket_size is None else int(bucket_size)


def _geometry(n, bucket_size):
    """(num_buckets, row_length) of the bucket view; ref: help_functions.py:67-94."""
    if bucket_size is None or n < bucket_size:
        return 1, n
    return -(-n // bucket_size), bucket_size


def _ptr(t):
    return None if t is None else t.data_ptr()


def _version_of(t):
    """Version counter of the tensor the lazy arg indices will be read from, or None for an inference tensor (those do
    not track versions -- `._version` raises -- and cannot be modified in place outside inference mode: no guard needed)."""
    return None if t.is_inference() else t._version


class ScalingFunction(object):
    """Scaling of a tensor to [0,1] and its inverse (ref: quant_functions.py:7-152).

    Holds what the inverse needs: `alpha`, `beta` (per bucket, shape (nb, 1); shape (1,) without
    buckets), `mean_tensor`, `original_tensor_size`, `original_tensor_length`,
    `expected_tensor_size`.  `idx_min_rows` / `idx_max_rows` (first-occurrence arg-min/max per
    bucket, int64) are computed ON DEMAND from the retained input the first time they are read:
    nothing on the per-step path reads them, and materialising them eagerly as the reference
    does (:85-90) would add 16 B/bucket of traffic plus index tracking to the hot kernel.  They
    are therefore only available while the tensor passed to `scale_down` has not been modified
    (checked through the tensor's version counter: reading them afterwards raises instead of
    returning stale indices; with `modify_in_place=True` they are computed eagerly, before the
    data is overwritten).
    """

    # Class-level defaults: an instance only stores what a call actually sets -- the per-step loops build one
    # ScalingFunction per parameter tensor per step (ref: conv_forward_model.py:235-247), so its construction
    # is on the host-side critical path of the drop-in.
    tol_diff_zero = 1e-10
    type_scaling = 'linear'        # what the native common-case entry point (glue.uniform_common) leaves unset: it builds the
    max_element = False            # instance without running __init__, for exactly this configuration
    subtract_mean = False
    modify_in_place = True         # as the reference's uniformQuantization sets it, :166-167
    mean_tensor = None
    _shape = None
    norm_scaling = None
    tensor_sign = None
    _n = None
    _ab = None                     # [2, nb, 1] / [2, 1]: alpha and beta in one allocation, split on first access
    _ab_slab = None                # ... or 2 * nb floats at offset _ab_off of a slab shared by many calls (glue.uniform_common)
    _ab_off = None
    _alpha = None
    _beta = None
    _idx_min_rows = None
    _idx_max_rows = None
    _arg_source = None             # tensor kept for the lazy arg-min/max
    _arg_version = None            # its ._version when it was scaled: a later in-place write invalidates the lazy indices
    _mean_buf = None

    def __init__(self, type_scaling, max_element, subtract_mean, bucket_size, modify_in_place=False):
        if type_scaling != 'linear':
            type_scaling = type_scaling.lower()
            if type_scaling not in ('linear', 'absmax', 'absnorm'):                 # ref: :22-25
                raise ValueError('Incorrect parameter: type of scaling must be "linear", "absMax" or "absNorm"')
        if bucket_size is not None and (type(bucket_size) is not int or bucket_size <= 0):
            # (np.integer and bool are rejected as the reference's isinstance(bucket_size, int) check does, :27-29)
            raise ValueError('Bucket size must be an integer and strictly positive. '
                             'Pass None if you want to avoid using buckets')
        if max_element is not False and (max_element is True or not isinstance(max_element, numbers.Number)):
            raise ValueError('maxElementAllowed must be a number')                  # ref: :31-33
        self.type_scaling = type_scaling
        self.max_element = max_element
        self.subtract_mean = subtract_mean
        self.bucket_size = bucket_size
        self.modify_in_place = modify_in_place

    # ------------------------------------------------------------------ lazily materialised fields
    def _split_ab(self):
        ab = self._ab
        if ab is None:
            if self._ab_slab is None:
                return
            nb = _geometry(self._n, self.bucket_size)[0]
            # a private copy of the 2 * nb floats, and the 1 MiB slab is let go: an object that is kept around (lists of
            # scaling functions, the Huffman accounting of help_functions) must not pin a slab shared with 65536 other calls
            ab = self._ab_slab[self._ab_off:self._ab_off + 2 * nb].clone()
            ab = ab.view(2, 1) if self.bucket_size is None else ab.view(2, nb, 1)
            self._ab = ab
            self._ab_slab = None
        self._alpha, self._beta = ab.unbind(0)

    @property
    def alpha(self):
        if self._alpha is None:
            self._split_ab()
        return self._alpha

    @alpha.setter
    def alpha(self, v):
        self._alpha = v

    @property
    def beta(self):
        if self._beta is None:
            self._split_ab()
        return self._beta

    @beta.setter
    def beta(self, v):
        self._beta = v

    @property
    def original_tensor_size(self):
        """Shape of the tensor that was scaled (ref: :76-77); taken from the retained source when not stored."""
        if self._shape is None and self._arg_source is not None:
            self._shape = self._arg_source.shape
        return self._shape

    @original_tensor_size.setter
    def original_tensor_size(self, v):
        self._shape = v

    @property
    def original_tensor_length(self):
        return self._n

    @original_tensor_length.setter
    def original_tensor_length(self, v):
        self._n = v

    @property
    def expected_tensor_size(self):
        """Shape of the bucket view (ref: :79-81, help_functions.py:67-94)."""
        if self._n is None:
            return None
        if self.bucket_size is None:
            return torch.Size([self._n])
        return torch.Size(_geometry(self._n, self.bucket_size))

    # ------------------------------------------------------------------ helpers
    def _abs_kind(self):
        """None for linear scaling; 0 ('absmax': max|x| per bucket) or 1 ('absnorm': L2 norm per bucket).
        The reference's code for the two abs types raises on every torch version
        (ref: :109-127 -- `tensor.max(p=2, ...)` is not a valid call, :126 stores a bound method), so these
        follow the intended math only (parity unpinned; see csrc/qd_abs.hip)."""
        return {'linear': None, 'absmax': 0, 'absnorm': 1}[self.type_scaling]

    def _clamp_args(self):
        if self.max_element is False:
            return 0, 0.0
        return 1, float(self.max_element)

    def _begin(self, tensor):
        """Record sizes, compute the mean if requested; returns (flat contiguous tensor, n, nb, row)."""
        _lib.require_f32(tensor)
        if not tensor.is_contiguous():
            if self.modify_in_place:
                raise ValueError('modify_in_place=True needs a contiguous tensor')
            tensor = tensor.contiguous()
        n = tensor.numel()
        self.original_tensor_size = tensor.size()
        self.original_tensor_length = n
        nb, row = _geometry(n, self.bucket_size)
        if self.subtract_mean:
            self._mean_buf = torch.empty(1, dtype=torch.float32, device=tensor.device)
            if n > 0:
                ws = _lib.workspace_for(tensor)
                _lib.check(_lib.lib_for(tensor).qd_mean_f32(tensor.data_ptr(), n, self._mean_buf.data_ptr(),
                                                            ws.data_ptr(), ws.numel(), _lib.stream_for(tensor)))
            self.mean_tensor = self._mean_buf.view(())                               # 0-dim, ref: :67
        else:
            self._mean_buf = None
            self.mean_tensor = 0                                                     # ref: :70
        return tensor, n, nb, row

    def _alloc_alpha_beta(self, nb, device):
        """One [2, nb] allocation; alpha/beta are its two rows, shaped (nb, 1) -- or (1,) without
        buckets -- as the reference's min/max(keepdim=True) shapes them (ref: :85-92)."""
        if self.bucket_size is None:
            ab = torch.empty(2, 1, dtype=torch.float32, device=device)
        else:
            ab = torch.empty(2, nb, 1, dtype=torch.float32, device=device)
        self._ab, self._alpha, self._beta, self._ab_slab = ab, None, None, None
        return ab

    def _note_arg_source(self, tensor, overwritten):
        self._idx_min_rows = None
        self._idx_max_rows = None
        self._arg_source = tensor
        self._arg_version = _version_of(tensor)
        if overwritten:                # the data is about to be replaced: materialise now
            self._compute_arg_indices()

    def _compute_arg_indices(self):
        t = self._arg_source
        if t is None:
            return
        if self._arg_version is not None and t._version != self._arg_version:
            # the reference computes the indices eagerly inside scale_down (ref: :85-90); here they are
            # taken from the tensor on first access, which is only the same thing while it is unchanged
            raise RuntimeError('idx_min_rows / idx_max_rows were requested after the tensor passed to scale_down / the '
                               'quantizer was modified in place: they are computed lazily from that tensor. Read them '
                               'before modifying it (or pass a copy).')
        n = t.numel()
        nb, _ = _geometry(n, self.bucket_size)
        out = torch.empty(2, nb, dtype=torch.int64, device=t.device)
        clamp, me = self._clamp_args()
        if n > 0:
            ws = _lib.workspace_for(t)
            _lib.check(_lib.lib_for(t).qd_bucket_argminmax_f32(
                t.data_ptr(), n, _bucket_arg(self.bucket_size), _ptr(self._mean_buf), clamp, me,
                out[0].data_ptr(), out[1].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_for(t)))
        shape = (1,) if self.bucket_size is None else (nb, 1)
        self._idx_min_rows = out[0].view(*shape)
        self._idx_max_rows = out[1].view(*shape)
        if self._shape is None:
            self._shape = t.shape
        self._arg_source = None

    @property
    def idx_min_rows(self):
        if self._idx_min_rows is None:
            self._compute_arg_indices()
        return self._idx_min_rows

    @idx_min_rows.setter
    def idx_min_rows(self, v):
        self._idx_min_rows = v

    @property
    def idx_max_rows(self):
        if self._idx_max_rows is None:
            self._compute_arg_indices()
        return self._idx_max_rows

    @idx_max_rows.setter
    def idx_max_rows(self, v):
        self._idx_max_rows = v

    # ------------------------------------------------------------------ API
    def scale_down(self, tensor):
        """u = (x - beta)/alpha per bucket, returned in the bucket layout (nb, bucket) -- padded
        with the scaled last element when the tensor is ragged -- or 1-D without buckets.
        ref: :56-129.  One kernel (K2)."""
        if isinstance(tensor, torch.Tensor) and _lib.on_other_device(tensor):
            with torch.cuda.device(tensor.device):
                return self.scale_down(tensor)
        tensor, n, nb, row = self._begin(tensor)
        padded = nb * row
        if self._abs_kind() is not None:
            return self._scale_down_abs(tensor, n, nb, padded)
        in_place = self.modify_in_place and padded == n
        self._note_arg_source(tensor, overwritten=in_place)
        out = tensor.view(-1) if in_place else (torch.empty(padded, dtype=torch.float32, device=tensor.device) if tensor.is_cuda
                                                else _lib.fresh_host_output(padded, torch.float32))
        ab = self._alloc_alpha_beta(nb, tensor.device)
        clamp, me = self._clamp_args()
        if n > 0:
            ws = _lib.workspace_for(tensor)
            _lib.check(_lib.lib_for(tensor).qd_scale_down_f32(
                tensor.data_ptr(), out.data_ptr(), n, _bucket_arg(self.bucket_size), ab[0].data_ptr(),
                ab[1].data_ptr(), _ptr(self._mean_buf), clamp, me, ws.data_ptr(), ws.numel(), _lib.stream_for(tensor)))
            if in_place:
                _lib.mark_written(tensor)          # the kernel wrote over the input through its raw pointer
        return out.view(self.expected_tensor_size)

    def _scale_down_abs(self, tensor, n, nb, padded):
        """sign + magnitude scaling (intended math of ref: :109-127): u = |x| / norm_b."""
        _abs_needs_device(tensor)
        dev = tensor.device
        u = torch.empty(padded, dtype=torch.float32, device=dev)
        sign = torch.empty(padded, dtype=torch.float32, device=dev)
        norm = torch.empty(nb, dtype=torch.float32, device=dev)
        clamp, me = self._clamp_args()
        if n > 0:
            ws = _lib.workspace(dev)
            _lib.check(_lib.load().qd_scale_down_abs_f32(
                tensor.data_ptr(), u.data_ptr(), sign.data_ptr(), n, _bucket_arg(self.bucket_size), self._abs_kind(),
                norm.data_ptr(), _ptr(self._mean_buf), clamp, me, ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
        self.norm_scaling = norm.view(1) if self.bucket_size is None else norm.view(nb, 1)
        self.tensor_sign = sign.view(self.expected_tensor_size)
        return u.view(self.expected_tensor_size)

    def inv_scale_down(self, tensor):
        """Inverse of scale_down (max_element truncation is not inverted).  ref: :131-152.  K3."""
        if isinstance(tensor, torch.Tensor) and _lib.on_other_device(tensor):
            with torch.cuda.device(tensor.device):
                return self.inv_scale_down(tensor)
        _lib.require_f32(tensor)
        if tensor.size() != self.expected_tensor_size:                               # ref: :138-139
            raise ValueError('The tensor passed has not the expected size.')
        if not tensor.is_contiguous():
            tensor = tensor.contiguous()
        n = self.original_tensor_length
        if self._abs_kind() is not None:                                             # ref: :144-146
            if self.tensor_sign is None:
                raise ValueError('inv_scale_down needs the signs recorded by scale_down')
            _abs_needs_device(tensor)
            out = torch.empty(n, dtype=torch.float32, device=tensor.device)
            if n > 0:
                _lib.check(_lib.load().qd_inv_scale_abs_f32(
                    tensor.data_ptr(), self.tensor_sign.data_ptr(), out.data_ptr(), n, _bucket_arg(self.bucket_size),
                    self.norm_scaling.data_ptr(), _ptr(self._mean_buf), _lib.stream_ptr(tensor.device)))
            return out.view(self.original_tensor_size)
        out = tensor.view(-1)[0:n] if self.modify_in_place else torch.empty(n, dtype=torch.float32,
                                                                          device=tensor.device)
        if n > 0:
            _lib.check(_lib.lib_for(tensor).qd_inv_scale_f32(
                tensor.data_ptr(), out.data_ptr(), n, _bucket_arg(self.bucket_size), self.alpha.data_ptr(),
                self.beta.data_ptr(), _ptr(self._mean_buf), _lib.stream_for(tensor)))
            if self.modify_in_place:
                _lib.mark_written(tensor)
        return out.view(self.original_tensor_size)


def _abs_needs_device(tensor):
    if not tensor.is_cuda:
        raise NotImplementedError("'absmax' / 'absnorm' scaling is implemented for tensors on a HIP device only (the reference's "
                                  "own code for the two raises on every torch version, ref: :109-127)")


def _uniform_host(tensor, s, type_of_scaling, stochastic_rounding, max_element, subtract_mean, bucket_size, modify_in_place):
    """uniformQuantization of a CPU tensor: the same single call into the C ABI (qd_uniform_f32: per-bucket min/max, alpha/beta,
    scale, round, rescale in one pass), served by libqd_host.so.  ref: :155-194."""
    sf = ScalingFunction(type_of_scaling, max_element, subtract_mean, bucket_size, True)    # validates as the reference, :22-33,166-167
    if int(s) != s or s < 2:
        raise ValueError('s must be an integer >= 2')
    if sf.type_scaling != 'linear':
        _abs_needs_device(tensor)
    sf.modify_in_place = modify_in_place                   # governs _begin's contiguity rule
    x, n, nb, row = sf._begin(tensor)                      # sizes, the mean if asked for
    sf.modify_in_place = True
    sf._note_arg_source(x, overwritten=bool(modify_in_place))      # the lazy arg indices: now, if x is about to be overwritten
    q = x if modify_in_place else _lib.fresh_host_output(n, torch.float32, like=x)
    ab = sf._alloc_alpha_beta(nb, x.device)
    clamp, me = sf._clamp_args()
    if n > 0:
        _lib.check(_lib.host().qd_uniform_f32(
            x.data_ptr(), q.data_ptr(), n, _bucket_arg(bucket_size), int(s), ab[0].data_ptr(), ab[1].data_ptr(), None,
            _ptr(sf._mean_buf), clamp, me, 1 if stochastic_rounding else 0,
            next_stochastic_seed(host=True) if stochastic_rounding else 0, None, 0, None))
        if modify_in_place:
            _lib.mark_written(x)
    return q.view(sf.original_tensor_size), sf


_glue_uniform = None
_glue_uniform_common = None
_new_object = object.__new__


def uniformQuantization(tensor, s, type_of_scaling='linear', stochastic_rounding=False,
                        max_element=False, subtract_mean=False, bucket_size=None, modify_in_place=False):
    """k-level uniform quantize-dequantize ("fake quantization") of `tensor` with `s` levels.
    Returns (quantized tensor of the same shape, ScalingFunction).  ref: :155-194.

    One fused kernel (K1) for bucketed tensors -- per-bucket min/max, alpha/beta, scale, round
    half to even, rescale; without buckets one kernel too when the tensor is small enough for the
    register-resident kernel, reduce + apply otherwise.  The input is left untouched unless
    modify_in_place=True.

    The call goes through the CPython/ATen binding of the C ABI (csrc/qd_torch_glue.cpp): output
    allocation, current stream and the launch happen in one native call, and the returned
    ScalingFunction materialises alpha / beta / sizes / the arg indices only when they are read.
    The common configuration of the training loops (linear scaling, no clamp, no mean, out of
    place; ref: conv_forward_model.py:216-221) takes a path with the argument checks inlined --
    this function is called once per parameter tensor per step."""
    global _glue_uniform, _glue_uniform_common
    if isinstance(tensor, torch.Tensor) and not tensor.is_cuda:      # a CPU tensor: libqd_host.so (the tensor's device decides)
        return _uniform_host(tensor, s, type_of_scaling, stochastic_rounding, max_element, subtract_mean, bucket_size,
                             modify_in_place)
    if _glue_uniform is None:
        g = _lib.glue()
        g.register(ScalingFunction)
        _glue_uniform, _glue_uniform_common = g.uniform, g.uniform_common
    if (type_of_scaling == 'linear' and max_element is False and not modify_in_place and not stochastic_rounding
            and not subtract_mean):
        # the configuration of the training loops (ref: conv_forward_model.py:216-221, 235-247): one native call that
        # returns (q, ScalingFunction) -- or None when an argument needs the checks / conversions of the general path
        done = _glue_uniform_common(tensor, s, bucket_size)
        if done is not None:
            return done
    if (type_of_scaling == 'linear' and max_element is False and not modify_in_place
            and (bucket_size is None or (type(bucket_size) is int and bucket_size > 0))
            and type(s) is int and s >= 2):
        q, ab, mean, n, x_read = _glue_uniform(tensor, s, bucket_size or 0, False, 0.0, stochastic_rounding,
                                               next_stochastic_seed() if stochastic_rounding else 0, subtract_mean, False)
        sf = _new_object(ScalingFunction)
        d = sf.__dict__
        d['type_scaling'] = 'linear'
        d['max_element'] = False
        d['subtract_mean'] = subtract_mean
        d['bucket_size'] = bucket_size
        d['modify_in_place'] = True                                                  # as the reference, :166-167
        d['_n'] = n
        d['_ab'] = ab
        d['_arg_source'] = x_read
        d['_arg_version'] = _version_of(x_read)          # of the tensor actually retained (a non-contiguous input was copied)
        if mean is not None:
            d['_mean_buf'] = mean
            d['mean_tensor'] = mean.view(())                                         # 0-dim, ref: :67
        else:
            d['mean_tensor'] = 0                                                     # ref: :70
        return q, sf
    sf = ScalingFunction(type_of_scaling, max_element, subtract_mean, bucket_size, True)    # as the reference, :166-167
    if int(s) != s or s < 2:
        raise ValueError('s must be an integer >= 2')
    if sf.type_scaling != 'linear':
        return _uniform_abs(tensor, s, sf, stochastic_rounding, modify_in_place)
    if modify_in_place:
        # the arg indices are taken lazily from the input, which is about to be overwritten: materialise them first
        _lib.require_device_f32(tensor)
        if not tensor.is_contiguous():
            raise ValueError('modify_in_place=True needs a contiguous tensor')
        if subtract_mean:
            sf.modify_in_place = True
            with torch.cuda.device(tensor.device):
                sf._begin(tensor)                  # the mean the indices are relative to
        sf._n = tensor.numel()
        with torch.cuda.device(tensor.device):
            sf._note_arg_source(tensor, overwritten=True)
    clamp = max_element is not False
    q, ab, mean, n, x_read = _glue_uniform(tensor, int(s), bucket_size or 0, clamp, float(max_element) if clamp else 0.0,
                                           stochastic_rounding, next_stochastic_seed() if stochastic_rounding else 0,
                                           subtract_mean, modify_in_place)
    sf.original_tensor_size = q.shape
    sf._n = n
    sf._ab = ab
    if mean is not None:
        sf._mean_buf = mean
        sf.mean_tensor = mean.view(())                                               # 0-dim, ref: :67
    else:
        sf.mean_tensor = 0                                                           # ref: :70
    if not modify_in_place:
        sf._arg_source = x_read
        sf._arg_version = _version_of(x_read)            # of the tensor actually retained (a non-contiguous input was copied)
    return q, sf


def _uniform_abs(tensor, s, scaling_function, stochastic_rounding, modify_in_place):
    """'absmax' / 'absnorm' scaling: intended math only, parity unpinned (see ScalingFunction._abs_kind)."""
    if isinstance(tensor, torch.Tensor) and _lib.on_other_device(tensor):
        with torch.cuda.device(tensor.device):
            return _uniform_abs(tensor, s, scaling_function, stochastic_rounding, modify_in_place)
    bucket_size = scaling_function.bucket_size
    scaling_function.modify_in_place = modify_in_place      # governs _begin's contiguity rule
    tensor, n, nb, row = scaling_function._begin(tensor)
    scaling_function.modify_in_place = True
    if stochastic_rounding:
        raise NotImplementedError('stochastic rounding is implemented for linear scaling only')
    out = tensor if modify_in_place else torch.empty_like(tensor)
    norm = torch.empty(nb, dtype=torch.float32, device=tensor.device)
    clamp, me = scaling_function._clamp_args()
    if n > 0:
        ws = _lib.workspace(tensor.device)
        _lib.check(_lib.load().qd_uniform_abs_f32(
            tensor.data_ptr(), out.data_ptr(), n, _bucket_arg(bucket_size), int(s), scaling_function._abs_kind(),
            norm.data_ptr(), _ptr(scaling_function._mean_buf), clamp, me, ws.data_ptr(), ws.numel(),
            _lib.stream_ptr(tensor.device)))
        if modify_in_place:
            _lib.mark_written(tensor)
    scaling_function.norm_scaling = norm.view(1) if bucket_size is None else norm.view(nb, 1)
    return out, scaling_function


class SearchSorted:
    """Device handle standing in for the reference's SearchSorted (ref: :509-573).

    The reference sorts the scaled tensor on the CPU (two argsorts, 4x the tensor's memory) so
    that each query is a searchsorted of the k-1 midpoints.  On the GPU none of that is needed:
    the index of an element is simply #{midpoints <= u}, computed per element by K5.  This object
    therefore only keeps `u` (fp32, device, unpadded order) resident between steps."""

    def __init__(self, tensor, use_k_optimization=True):
        _lib.require_f32(tensor, 'SearchSorted tensor')
        self.scaled_tensor = tensor.contiguous().view(-1)
        self.use_k_optimization = use_k_optimization

    def query(self, k):
        """Indices (int64, flat) of the nearest point of `k` (sorted 1-D points) for every
        element, by the midpoint rule of ref: :531-563."""
        pts = _points_on(k, self.scaled_tensor.device)
        n = self.scaled_tensor.numel()
        idx = torch.empty(n, dtype=torch.int64, device=self.scaled_tensor.device)
        one = torch.ones(1, dtype=torch.float32, device=pts.device)
        zero = torch.zeros(1, dtype=torch.float32, device=pts.device)
        ws = _lib.workspace_for(pts)
        if n > 0:
            # indices only (q = NULL): 4 B read + 8 B written per element; the kernel wants n >= 4 and a 16-byte aligned base,
            # the few-element / offset-view case writes its values into a throw-away buffer instead
            only = n >= 4 and self.scaled_tensor.data_ptr() % 16 == 0
            scratch = None if only else torch.empty(n, dtype=torch.float32, device=self.scaled_tensor.device)
            _lib.check(_lib.lib_for(self.scaled_tensor).qd_nearest_point_f32(
                self.scaled_tensor.data_ptr(), 1, pts.data_ptr(), pts.numel(), 1, None if only else scratch.data_ptr(),
                idx.data_ptr(), 8, n, 0, one.data_ptr(), zero.data_ptr(), None, 0, 0.0,
                ws.data_ptr(), ws.numel(), _lib.stream_for(self.scaled_tensor)))
        return idx


def _points_on(points, device):
    if isinstance(points, (list, tuple)):
        points = torch.tensor(points, dtype=torch.float32)                            # ref: :238-239
    if not isinstance(points, torch.Tensor):
        points = torch.as_tensor(points, dtype=torch.float32)
    if points.requires_grad:
        points = points.detach()
    if points.dtype != torch.float32 or points.device != device or not points.is_contiguous():
        points = points.to(device=device, dtype=torch.float32).contiguous()
    return points


def _nearest(x, prescaled, points, assign_mode, n, bucket_size, alpha, beta, mean_buf, clamp, me, idx_bytes, in_place=False):
    """(q [n], idx [n]) -- one K4/K5 launch through the native binding (allocation + stream + launch).  in_place: q is
    written over x (every kernel of the family loads a bucket before it stores it)."""
    if not x.is_cuda:                                   # a CPU tensor: the same entry point of libqd_host.so
        q = x.view(-1)[0:n] if in_place else _lib.fresh_host_output(n, torch.float32)
        idx = _lib.fresh_host_output(n, torch.int64 if idx_bytes == 8 else torch.uint8)
        if n > 0:
            _lib.check(_lib.host().qd_nearest_point_f32(
                x.data_ptr(), 1 if prescaled else 0, points.data_ptr(), points.numel(), assign_mode, q.data_ptr(), idx.data_ptr(),
                idx_bytes, n, bucket_size or 0, alpha.data_ptr(), beta.data_ptr(), _ptr(mean_buf), clamp, me, None, 0, None))
            if in_place:
                _lib.mark_written(x)
        return q, idx
    return _lib.glue().nearest(x, prescaled, points, assign_mode, n, bucket_size or 0, alpha, beta, mean_buf,
                               clamp, me, idx_bytes, in_place)


def nonUniformQuantization(tensor, listQuantizationPoints, max_element=False,
                           subtract_mean=False, modify_in_place=False, bucket_size=None,
                           pre_processed_values=False, search_sorted_obj=None, scaling_function=None,
                           tensors_info=None, index_dtype=torch.int64):
    """Quantize every element to the nearest of the (sorted, in [0,1]) quantization points after
    per-bucket linear scaling.  Returns (quantized, indices int64 of the same shape,
    ScalingFunction).  ref: :196-290.

    index_dtype (an extension; the default is the reference's LongTensor, :288-289): torch.uint8 returns the same indices
    one byte each -- 9 instead of 16 bytes of HBM traffic per element, 12 of the 16 being the int64 store -- for at most 256
    points; `indices.long()` is then exactly what the default returns.

    Plain path: one fused kernel (K4: scale, nearest point by the distance rule of :267-273,
    gather, rescale).  Pre-processed path (`pre_processed_values=True`, the per-step call of
    differentiable quantization): the scaled tensor stays resident on the device inside
    `search_sorted_obj`, and one kernel (K5) assigns by the midpoint rule of :531-563."""
    _where = tensor if isinstance(tensor, torch.Tensor) else (
        search_sorted_obj.scaled_tensor if isinstance(search_sorted_obj, SearchSorted) else None)
    if _where is not None and _lib.on_other_device(_where):
        with torch.cuda.device(_where.device):
            return nonUniformQuantization(tensor, listQuantizationPoints, max_element, subtract_mean, modify_in_place,
                                          bucket_size, pre_processed_values, search_sorted_obj, scaling_function,
                                          tensors_info, index_dtype)
    if index_dtype not in (torch.int64, torch.uint8):
        raise ValueError('index_dtype must be torch.int64 (the reference\'s) or torch.uint8')
    if index_dtype is torch.uint8 and len(listQuantizationPoints) > 256:
        raise ValueError('uint8 indices address at most 256 quantization points')
    idx_bytes = 8 if index_dtype is torch.int64 else 1
    if pre_processed_values is True and (search_sorted_obj is None or scaling_function is None
                                         or tensors_info is None):                  # ref: :230-231
        raise ValueError('If values are preprocessed, all pre processed arguments need to be passed')
    if pre_processed_values is False and not (search_sorted_obj is None and scaling_function is None
                                              and tensors_info is None):              # ref: :233-236
        raise ValueError('pre processing is False but you are passing some pre processing values. '
                         'This is probably not what you wanted to do, so to avoid bugs an error is raised')

    if not pre_processed_values:
        sf = ScalingFunction(type_scaling='linear', max_element=max_element, subtract_mean=subtract_mean,
                             bucket_size=bucket_size, modify_in_place=True)           # ref: :248-250
        sf.modify_in_place = modify_in_place
        tensor, n, nb, row = sf._begin(tensor)
        sf.modify_in_place = True
        sf._note_arg_source(tensor, overwritten=modify_in_place)
        points = _points_on(listQuantizationPoints, tensor.device)
        sf._alloc_alpha_beta(nb, tensor.device)
        clamp, me = sf._clamp_args()
        q, idx = _nearest(tensor, False, points, 0, n, bucket_size, sf.alpha, sf.beta, sf._mean_buf,
                          clamp, me, idx_bytes, in_place=bool(modify_in_place))  # in place: the kernel writes over the input
        return q.view(sf.original_tensor_size), idx.view(sf.original_tensor_size), sf

    sf = scaling_function
    u = search_sorted_obj.scaled_tensor
    n = sf.original_tensor_length
    points = _points_on(listQuantizationPoints, u.device)
    q, idx = _nearest(u, True, points, 1, n, sf.bucket_size, sf.alpha, sf.beta, sf._mean_buf, 0, 0.0, idx_bytes)
    return q.view(sf.original_tensor_size), idx.view(sf.original_tensor_size), sf


class uniformQuantization_variable(object):
    """Forward/backward pair around uniformQuantization, called as plain methods by the training
    loops (cnn_models/conv_forward_model.py:245,266), not through autograd.  ref: :293-406.

    `backward` is the 'complicated' straight-through estimator: one kernel (K7), one wave per
    bucket, instead of the reference's N x N sparse matrices (which, as shipped, raise for more
    than one bucket -- this implements the intended math with the reference's tie rule)."""

    def __init__(self, s, type_of_scaling='linear', stochastic_rounding=False, max_element=False,
                 subtract_mean=False, modify_in_place=False, bucket_size=None):
        self.s = s
        self.typeOfScaling = type_of_scaling
        self.stochasticRounding = stochastic_rounding
        self.maxElementAllowed = max_element
        self.subtractMean = subtract_mean
        self.modifyInPlace = modify_in_place
        self.bucket_size = bucket_size
        self.saved_for_backward = None

    def forward(self, input):
        self.saved_for_backward = {'input': input.clone()}                           # ref: :308-309
        return uniformQuantization(input, s=self.s, type_of_scaling=self.typeOfScaling,
                                   stochastic_rounding=self.stochasticRounding,
                                   max_element=self.maxElementAllowed, subtract_mean=self.subtractMean,
                                   modify_in_place=self.modifyInPlace, bucket_size=self.bucket_size)[0]

    __call__ = forward

    def backward(self, grad_output, tie_mode='reference'):
        if self.typeOfScaling != 'linear':                                           # ref: :326-327
            raise ValueError('Linear scaling is necessary to backpropagate')
        if self.subtractMean is True:                                                # ref: :329-330
            raise NotImplementedError('The backprop function assumes subtractMean to be False for now')
        if self.bucket_size is None:                                                 # ref: :332-334
            raise NotImplementedError('Right now the code does not work with bucket_size None.'
                                      ' Not hard to modify though')
        if self.saved_for_backward is None:                                          # ref: :336-337
            raise ValueError('Need to have called .forward() to be able to call .backward()')
        x = self.saved_for_backward['input']
        _lib.require_f32(grad_output, 'grad_output')
        if grad_output.device != x.device:
            raise ValueError('grad_output must live on the device of the input of forward()')
        if _lib.on_other_device(grad_output):
            with torch.cuda.device(grad_output.device):
                return self.backward(grad_output, tie_mode)
        g = grad_output.contiguous()
        if g.numel() != x.numel():
            raise ValueError('grad_output must have as many elements as the input of forward()')
        out = torch.empty_like(g)
        if x.numel() > 0:
            x = x.contiguous()
            _lib.check(_lib.lib_for(x).qd_ste_bucket_backward_f32(
                x.data_ptr(), g.data_ptr(), out.data_ptr(), x.numel(), int(self.bucket_size), int(self.s),
                0 if tie_mode == 'reference' else 1, _lib.stream_for(x)))
        self.saved_for_backward = None                                               # ref: :404-405
        return out.view(x.size())


class _Saved(dict):
    """savedForBackward of nonUniformQuantization_variable.  The kernels keep the assignment as
    uint8 when k <= 256 (1 B/element instead of 8); callers that read ['indices'] still get the
    LongTensor the reference stores (converted on access)."""

    def __getitem__(self, key):
        if key == 'indices':
            raw = dict.__getitem__(self, 'indices')
            return raw if raw.dtype == torch.int64 else raw.long()
        return dict.__getitem__(self, key)

    def raw_indices(self):
        return dict.__getitem__(self, 'indices')


class nonUniformQuantization_variable(object):
    """Forward/backward pair of non-uniform quantization used by the differentiable-quantization
    loop (cnn_models/conv_forward_model.py:507-545).  ref: :408-506.

    With `pre_process_tensors=True` the tensor is scaled ONCE (K2) and `u`, alpha, beta stay on
    the device; every `forward(None, points)` is then a single kernel (K5) that reads u (4 B),
    writes q (4 B) and a uint8 index (1 B) per element, and `backward` is a deterministic
    two-stage segmented reduction (K6).  No sort, no host round trip (the reference moves N
    int64 + N fp32 across PCIe per tensor per step, :279-284)."""

    def __init__(self, max_element=False, subtract_mean=False, modify_in_place=False, bucket_size=None,
                 pre_process_tensors=False, tensor=None):
        if pre_process_tensors is True and (tensor is None):                          # ref: :413-414
            raise ValueError('To pre-process tensors you need to pass the tensor and the scaling function options')
        self.maxElementAllowed = max_element
        self.subtractMean = subtract_mean
        self.modifyInPlace = modify_in_place
        self.bucket_size = bucket_size
        self.savedForBackward = None
        self.pre_process_tensors = pre_process_tensors
        self.search_sorted_obj = None
        self.tensors_info = None
        self.scaling_function = None
        if self.pre_process_tensors:
            self.preprocess(tensor)

    def preprocess(self, tensor):
        sf = ScalingFunction(type_scaling='linear', max_element=self.maxElementAllowed,
                             subtract_mean=self.subtractMean, bucket_size=self.bucket_size,
                             modify_in_place=self.modifyInPlace)
        u = sf.scale_down(tensor)                                                     # ref: :437
        sf.modify_in_place = True
        self.search_sorted_obj = SearchSorted(u.view(-1)[0:sf.original_tensor_length])
        self.tensors_info = (tensor.type(), tensor.is_cuda)
        self.scaling_function = sf

    def forward(self, inputTensor, listQuantizationPoints):
        if listQuantizationPoints.dim() != 1:                                         # ref: :451-452
            raise ValueError('listPoints must be a 1-D tensor')
        numPoints = listQuantizationPoints.size()[0]
        if self.pre_process_tensors and _lib.on_other_device(self.search_sorted_obj.scaled_tensor):
            with torch.cuda.device(self.search_sorted_obj.scaled_tensor.device):
                return self.forward(inputTensor, listQuantizationPoints)
        if self.pre_process_tensors:
            sf = self.scaling_function
            u = self.search_sorted_obj.scaled_tensor
            points = _points_on(listQuantizationPoints, u.device)
            n = sf.original_tensor_length
            q, idx = _nearest(u, True, points, 1, n, sf.bucket_size, sf.alpha, sf.beta, sf._mean_buf,
                              0, 0.0, 1 if numPoints <= 256 else 8)
            shape = sf.original_tensor_size
            if len(shape) != 1:
                q = q.view(shape)
                idx = idx.view(shape)
        else:
            q, idx, sf = nonUniformQuantization(
                inputTensor, listQuantizationPoints, modify_in_place=self.modifyInPlace,
                max_element=self.maxElementAllowed, subtract_mean=self.subtractMean, bucket_size=self.bucket_size)
        self.savedForBackward = _Saved(indices=idx, numPoints=numPoints, scalingFactor=sf.alpha)   # ref: :467-468
        return q

    __call__ = forward

    def backward(self, grad_output):
        """Returns (grad wrt the input = grad_output unchanged, grad wrt the points).  ref: :471-506."""
        if self.savedForBackward is None:                                             # ref: :478-479
            raise ValueError('Need savedIndices to be able to call backward()')
        _lib.require_f32(grad_output, 'grad_output')
        if _lib.on_other_device(grad_output):
            with torch.cuda.device(grad_output.device):
                return self.backward(grad_output)
        idx = self.savedForBackward.raw_indices()
        k = self.savedForBackward['numPoints']
        alpha = self.savedForBackward['scalingFactor']
        if not grad_output.is_cuda:                         # CPU tensors: the same entry point of libqd_host.so
            g = grad_output.contiguous()
            if idx.device != g.device or idx.numel() != g.numel():
                raise ValueError('grad_output must match the quantized tensor in size and device')
            grad_points = torch.empty(int(k), dtype=torch.float32)
            _lib.check(_lib.host().qd_point_grad_f32(
                g.data_ptr(), idx.data_ptr(), 8 if idx.dtype == torch.int64 else 1, alpha.data_ptr(), g.numel(),
                self.bucket_size or 0, int(k), grad_points.data_ptr(), None, 0, None))
            self.savedIndices = None                                                  # ref: :505
            return grad_output, grad_points
        grad_points = _lib.glue().point_grad(grad_output, idx, alpha, self.bucket_size or 0, int(k))
        self.savedIndices = None                                                      # ref: :505
        return grad_output, grad_points
