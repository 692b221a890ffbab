"""Packed representation of uniformly quantized tensors and device-side size accounting.

The reference only *accounts* for the compressed size of a quantized model
(ref: helpers/functions.py:216-262 -- `bits*N/8` bytes of level indices + 8 bytes per bucket --
and quantization/help_functions.py:175-232 for the Huffman mean code length); it never builds
the compressed form.  This module builds it on the device (qd_pack_uniform_f32), decodes it back
to exactly the tensor uniformQuantization returns (qd_unpack_uniform_f32), and computes the level
histogram / Huffman length without copying tensors to the host (qd_histogram_u8).
"""
import torch

from . import _lib
from .quantization.help_functions import huffman_encode


def bits_for_levels(s):
    for b in (1, 2, 4, 8):
        if s <= (1 << b):
            return b
    raise ValueError('the packed format holds at most 256 levels')


class PackedUniform(object):
    """`bits`-per-element level indices + per-bucket (alpha, beta) of one tensor."""

    def __init__(self, packed, alpha, beta, shape, s, bucket_size, bits):
        self.packed, self.alpha, self.beta = packed, alpha, beta
        self.shape, self.s, self.bucket_size, self.bits = torch.Size(shape), s, bucket_size, bits

    @property
    def nbytes(self):
        """Bytes of the compressed form: what helpers/functions.py:255-259 charges for it."""
        return self.packed.numel() + 4 * (self.alpha.numel() + self.beta.numel())

    def unpack(self):
        """The fake-quantized fp32 tensor, bit-identical to uniformQuantization(x, s, bucket)[0]."""
        n = self.shape.numel()
        y = torch.empty(n, dtype=torch.float32, device=self.packed.device)
        if n > 0:
            _lib.check(_lib.load().qd_unpack_uniform_f32(self.packed.data_ptr(), n, self.bucket_size or 0, self.s, self.bits,
                                                         self.alpha.data_ptr(), self.beta.data_ptr(), y.data_ptr(),
                                                         _lib.stream_ptr()))
        return y.view(self.shape)


_FUSED_BUCKETS = (64, 128, 256, 512, 1024, 2048)
QD_ERR_UNSUPPORTED = -3                 # include/qd_hip.h


def pack_uniform(tensor, s, bucket_size=256, bits=None):
    """Quantize `tensor` with `s` levels per bucket and keep only the packed level indices (+ alpha / beta per bucket).
    Bucket sizes 64 ... 2048 (powers of two) take one fused kernel (4 B read + bits/8 B written per element); any other
    bucket size -- and bucket_size=None -- quantizes with the level-index side output of the quantize kernel and packs
    those (qd_uniform_f32 + qd_pack_levels_u8): the same bytes, whatever the bucket geometry."""
    _lib.require_device_f32(tensor)
    if bucket_size is not None and (not isinstance(bucket_size, int) or isinstance(bucket_size, bool) or bucket_size <= 0):
        raise ValueError('Bucket size must be an integer and strictly positive. Pass None if you want to avoid using buckets')
    if int(s) != s or s < 2:
        raise ValueError('s must be an integer >= 2')
    bits = bits_for_levels(s) if bits is None else bits
    if bits not in (1, 2, 4, 8) or s > (1 << bits):
        raise ValueError('bits must be 1, 2, 4 or 8 and hold s levels')
    if _lib.on_other_device(tensor):
        with torch.cuda.device(tensor.device):
            return pack_uniform(tensor, s, bucket_size, bits)
    x = tensor.contiguous()
    n = x.numel()
    lib = _lib.load()
    nb = 1 if (bucket_size is None or n < bucket_size) else -(-n // bucket_size)
    nbytes = int(lib.qd_packed_bytes(n, bits))
    packed = torch.empty(nbytes + 8, dtype=torch.uint8, device=x.device)[:nbytes]
    ab = torch.empty(2, nb, dtype=torch.float32, device=x.device)
    if n > 0 and bucket_size in _FUSED_BUCKETS and x.data_ptr() % 16 == 0:
        _lib.check(lib.qd_pack_uniform_f32(x.data_ptr(), n, bucket_size, int(s), bits, packed.data_ptr(),
                                           ab[0].data_ptr(), ab[1].data_ptr(), _lib.stream_ptr()))
    elif n > 0:
        lev = torch.empty(n, dtype=torch.uint8, device=x.device)
        q = torch.empty_like(x)
        ws = _lib.workspace(x.device)
        _lib.check(lib.qd_uniform_f32(x.data_ptr(), q.data_ptr(), n, 0 if bucket_size is None else bucket_size, int(s),
                                      ab[0].data_ptr(), ab[1].data_ptr(), lev.data_ptr(), None, 0, 0.0, 0, 0,
                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        _lib.check(lib.qd_pack_levels_u8(lev.data_ptr(), n, bits, packed.data_ptr(), _lib.stream_ptr()))
    return PackedUniform(packed, ab[0], ab[1], tensor.shape, int(s), bucket_size, bits)


def histogram_u8(idx, k):
    """Counts of the symbols 0..k-1 in a uint8 device tensor, as an int64 device tensor [k]."""
    if idx.dtype != torch.uint8 or not idx.is_cuda:
        raise TypeError('histogram_u8 needs a uint8 tensor on a HIP device')
    idx = idx.contiguous()
    if _lib.on_other_device(idx):
        with torch.cuda.device(idx.device):
            return histogram_u8(idx, k)
    hist = torch.empty(k, dtype=torch.int64, device=idx.device)
    ws = _lib.workspace(idx.device)
    _lib.check(_lib.load().qd_histogram_u8_ws(idx.data_ptr(), idx.numel(), int(k), hist.data_ptr(), ws.data_ptr(), ws.numel(),
                                              _lib.stream_ptr()))
    return hist


def level_histogram(tensor, s, bucket_size=None):
    """Histogram of the quantization levels of `tensor` (s <= 256), computed on the device.  At the vector bucket sizes
    (64 ... 2048) ONE pass over the tensor counts the levels in the kernel that computes them (qd_level_histogram_f32: 4 B read
    per element, nothing written but s counters).  Other bucket geometries (bucket_size None, any other size, a misaligned
    view) write the uint8 levels with the quantize kernel and count those."""
    _lib.require_device_f32(tensor)
    x = tensor.contiguous().view(-1)
    if _lib.on_other_device(x):
        with torch.cuda.device(x.device):
            return level_histogram(tensor, s, bucket_size)
    n = x.numel()
    lib = _lib.load()
    ws = _lib.workspace(x.device)
    bucket = 0 if bucket_size is None else bucket_size
    hist = torch.empty(int(s), dtype=torch.int64, device=x.device)
    rc = lib.qd_level_histogram_f32(x.data_ptr(), n, bucket, int(s), hist.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
    if rc == 0:
        return hist
    if rc != QD_ERR_UNSUPPORTED:
        _lib.check(rc)
    lev = torch.empty(n, dtype=torch.uint8, device=x.device)
    if n > 0:
        q = torch.empty_like(x)
        _lib.check(lib.qd_uniform_f32(x.data_ptr(), q.data_ptr(), n, bucket, int(s), None, None, lev.data_ptr(), None, 0, 0.0, 0, 0,
                                      ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    return histogram_u8(lev, s)


def huffman_mean_bit_length_uniform(params, s, bucket_size=None):
    """Mean Huffman code length (bits/weight) of the level indices of a model, as
    get_huffman_encoding_mean_bit_length(..., 'uniform', s) computes it
    (ref: help_functions.py:175-232), with the histogram built on the device: only s counters per
    model cross PCIe instead of every quantized tensor."""
    total = None
    for p in params:
        t = p.data if hasattr(p, 'data') else p
        h = level_histogram(t, s, bucket_size)
        total = h if total is None else total + h
    counts = total.cpu().tolist()
    n = sum(counts)
    freq = {j: c / n for j, c in enumerate(counts) if c > 0}
    return sum(freq[sym] * len(code) for sym, code in huffman_encode(freq))
