"""ctypes binding of libqd_hip.so (C ABI: include/qd_hip.h), and of libqd_host.so for CPU tensors.

PyTorch is used only for device memory and the current HIP stream; every kernel is launched
through the C ABI with raw device pointers.  Loading fails loudly when the library has not been
built -- there is no fallback implementation.

One library per device: a tensor on a HIP device goes to libqd_hip.so and nowhere else (a missing
libqd_hip.so is an error, whatever else is installed); a CPU tensor goes to libqd_host.so
(csrc/host/qd_host.cpp: the per-call entry points of the same header for host pointers -- the
reference's functions accept CPU tensors, quantization/quant_functions.py:186,254,283-284).  The
choice is made from the tensor's device by lib_for(); nothing is ever computed on another
device than the one the caller's tensor lives on.
"""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libqd_hip.so')
HOST_LIB_PATH = os.path.join(_HERE, 'libqd_host.so')      # the same entry points for CPU tensors (csrc/host/qd_host.cpp)
CSRC = os.path.join(_HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(_HERE), 'include')

QD_ERR_UNSUPPORTED = -3            # include/qd_hip.h
ABI_VERSION = 3                # QD_ABI_VERSION of include/qd_hip.h this file mirrors (tests/test_abi.py compares the two)

_lib = None
_lock = threading.Lock()

c_f = ctypes.c_void_p          # device float*
c_p = ctypes.c_void_p
i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_size = ctypes.c_size_t
u64 = ctypes.c_uint64


class QdTensorDesc(ctypes.Structure):
    """Mirror of `struct QdTensorDesc` in include/qd_hip.h."""
    _fields_ = [('x', ctypes.c_void_p), ('q', ctypes.c_void_p), ('n', ctypes.c_int64),
                ('first_tile', ctypes.c_int64)]


class QdDiffQuantDesc(ctypes.Structure):
    """Mirror of `struct QdDiffQuantDesc` in include/qd_hip.h."""
    _fields_ = [('u', ctypes.c_void_p), ('q', ctypes.c_void_p), ('idx', ctypes.c_void_p), ('alpha', ctypes.c_void_p),
                ('beta', ctypes.c_void_p), ('grad', ctypes.c_void_p), ('n', ctypes.c_int64),
                ('first_tile', ctypes.c_int64), ('first_block', ctypes.c_int64), ('first_row', ctypes.c_int64)]


# symbol -> (restype, argtypes); every symbol declared in include/qd_hip.h must be listed here
# (tests/test_abi.py cross-checks this table against the header).
SIGNATURES = {
    'qd_abi_version': (c_int, []),
    'qd_target_arch': (ctypes.c_char_p, []),
    'qd_error_string': (ctypes.c_char_p, [c_int]),
    'qd_workspace_bytes': (c_size, []),
    'qd_set_single_fused_mode': (c_int, [c_int]),
    'qd_num_buckets': (i64, [i64, i64]),
    'qd_padded_length': (i64, [i64, i64]),
    'qd_mean_f32': (c_int, [c_f, i64, c_f, c_p, c_size, c_p]),
    'qd_uniform_f32': (c_int, [c_f, c_f, i64, i64, c_int, c_f, c_f, c_p, c_f, c_int, c_float, c_int, u64,
                               c_p, c_size, c_p]),
    'qd_scale_down_f32': (c_int, [c_f, c_f, i64, i64, c_f, c_f, c_f, c_int, c_float, c_p, c_size, c_p]),
    'qd_inv_scale_f32': (c_int, [c_f, c_f, i64, i64, c_f, c_f, c_f, c_p]),
    'qd_bucket_argminmax_f32': (c_int, [c_f, i64, i64, c_f, c_int, c_float, c_p, c_p, c_p, c_size, c_p]),
    'qd_nearest_point_f32': (c_int, [c_f, c_int, c_f, c_int, c_int, c_f, c_p, c_int, i64, i64, c_f, c_f, c_f,
                                     c_int, c_float, c_p, c_size, c_p]),
    'qd_point_grad_f32': (c_int, [c_f, c_p, c_int, c_f, i64, i64, c_int, c_f, c_p, c_size, c_p]),
    'qd_ste_bucket_backward_f32': (c_int, [c_f, c_f, c_f, i64, i64, c_int, c_int, c_p]),
    'qd_clamp_f32': (c_int, [c_f, i64, c_float, c_p]),
    'qd_truncated_ste_f32': (c_int, [c_f, c_f, i64, c_float, c_p]),
    'qd_multi_plan': (i64, [ctypes.POINTER(QdTensorDesc), c_int, i64]),
    'qd_multi_uniform_f32': (c_int, [c_p, c_int, i64, i64, c_int, c_p]),
    'qd_multi_global_plan': (i64, [ctypes.POINTER(QdTensorDesc), c_int]),
    'qd_multi_uniform_global_f32': (c_int, [c_p, c_int, i64, c_int, c_f, c_p, c_size, c_p]),
    'qd_uniform_abs_f32': (c_int, [c_f, c_f, i64, i64, c_int, c_int, c_f, c_f, c_int, c_float, c_p, c_size, c_p]),
    'qd_scale_down_abs_f32': (c_int, [c_f, c_f, c_f, i64, i64, c_int, c_f, c_f, c_int, c_float, c_p, c_size, c_p]),
    'qd_inv_scale_abs_f32': (c_int, [c_f, c_f, c_f, i64, i64, c_f, c_f, c_p]),
    'qd_multi_dq_plan': (i64, [ctypes.POINTER(QdDiffQuantDesc), c_int, i64, ctypes.POINTER(ctypes.c_int64)]),
    'qd_multi_nearest_f32': (c_int, [c_p, c_int, i64, i64, c_f, c_int, c_p]),
    'qd_multi_point_grad_f32': (c_int, [c_p, c_int, i64, i64, c_int, c_f, c_p, c_size, c_p]),
    'qd_packed_bytes': (i64, [i64, c_int]),
    'qd_pack_uniform_f32': (c_int, [c_f, i64, i64, c_int, c_int, c_p, c_f, c_f, c_p]),
    'qd_pack_levels_u8': (c_int, [c_p, i64, c_int, c_p, c_p]),
    'qd_unpack_uniform_f32': (c_int, [c_p, i64, i64, c_int, c_int, c_f, c_f, c_f, c_p]),
    'qd_histogram_u8': (c_int, [c_p, i64, c_int, c_p, c_p]),
    'qd_histogram_u8_ws': (c_int, [c_p, i64, c_int, c_p, c_p, c_size, c_p]),
    'qd_level_histogram_f32': (c_int, [c_f, i64, i64, c_int, c_p, c_p, c_size, c_p]),
    'qd_digitize_histogram_f32': (c_int, [c_f, i64, c_p, c_int, c_p, c_p, c_size, c_p]),
    'qd_histogram_i64': (c_int, [c_p, i64, c_int, c_p, c_p, c_size, c_p]),
    'qd_scale_digitize_histogram_f32': (c_int, [c_f, i64, i64, c_p, c_int, c_p, c_p, c_size, c_p]),
    'qd_order_stats_workspace_bytes': (ctypes.c_size_t, [c_int]),
    'qd_order_stats_f32': (c_int, [c_p, i64, c_p, c_int, c_p, c_p, ctypes.c_size_t, c_p]),
    'qd_selftest_div_invariant': (c_int, [u64, i64, c_int, c_p, c_p]),
}


class QdLibraryMissing(RuntimeError):
    pass


def load():
    """Load libqd_hip.so (once).  Raises QdLibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise QdLibraryMissing(
                'quantized_distillation_amd: %s is missing. Build the HIP extension first: '
                'python -c "import __graft_entry__ as g; g.build()" (needs hipcc, gfx950). '
                'There is no CPU fallback.' % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError = ABI mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.qd_abi_version() != ABI_VERSION:
            raise RuntimeError('libqd_hip.so has ABI version %d, this binding is written for %d: rebuild it '
                               '(python -c "import __graft_entry__ as g; g.build()")' % (lib.qd_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


# the entry points libqd_host.so implements (the per-call functions; multi-tensor, codec, order statistics and the
# 'absmax' / 'absnorm' scalings exist for device tensors only)
HOST_SYMBOLS = ('qd_abi_version', 'qd_target_arch', 'qd_error_string', 'qd_workspace_bytes', 'qd_num_buckets', 'qd_padded_length',
                'qd_mean_f32', 'qd_uniform_f32', 'qd_scale_down_f32', 'qd_inv_scale_f32', 'qd_bucket_argminmax_f32',
                'qd_nearest_point_f32', 'qd_point_grad_f32', 'qd_ste_bucket_backward_f32', 'qd_clamp_f32', 'qd_truncated_ste_f32')
_host = None


def host():
    """Load libqd_host.so (once): the library CPU tensors are computed by.  Raises QdLibraryMissing if it has not been built."""
    global _host
    if _host is not None:
        return _host
    with _lock:
        if _host is not None:
            return _host
        if not os.path.exists(HOST_LIB_PATH):
            raise QdLibraryMissing(
                'quantized_distillation_amd: %s is missing. Build it first: '
                'python -c "import __graft_entry__ as g; g.build()" (needs g++ with OpenMP).' % HOST_LIB_PATH)
        lib = ctypes.CDLL(HOST_LIB_PATH)
        for name in HOST_SYMBOLS:
            res, args = SIGNATURES[name]
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        lib.qd_host_max_threads.restype = c_int
        lib.qd_host_max_threads.argtypes = []
        lib.qd_host_set_threads.restype = None
        lib.qd_host_set_threads.argtypes = [c_int]
        if lib.qd_abi_version() != ABI_VERSION:
            raise RuntimeError('libqd_host.so has ABI version %d, this binding is written for %d: rebuild it '
                               '(python -c "import __graft_entry__ as g; g.build()")' % (lib.qd_abi_version(), ABI_VERSION))
        _host = lib
    return _host


def lib_for(t):
    """The library that computes on `t`'s device: libqd_hip.so for a HIP tensor, libqd_host.so for a CPU tensor."""
    return load() if t.is_cuda else host()


_glue = None
GLUE_PATH = os.path.join(_HERE, '_qd_glue.so')


def glue():
    """The CPython/ATen binding of the per-call entry points (csrc/qd_torch_glue.cpp -> _qd_glue.so): one native
    call does output allocation + current stream + the C-ABI launch.  Fails loudly when it has not been built."""
    global _glue
    if _glue is None:
        load()
        if not os.path.exists(GLUE_PATH):
            raise QdLibraryMissing(
                'quantized_distillation_amd: %s is missing. Build it first: '
                'python -c "import __graft_entry__ as g; g.build()". There is no fallback binding.' % GLUE_PATH)
        import torch  # noqa: F401  (libtorch must be loaded before the extension)
        from . import _qd_glue
        if _qd_glue.abi_version() != ABI_VERSION:
            raise RuntimeError('_qd_glue.so was compiled for ABI version %d, this binding and libqd_hip.so are at %d: rebuild it '
                               '(python -c "import __graft_entry__ as g; g.build()")' % (_qd_glue.abi_version(), ABI_VERSION))
        _glue = _qd_glue
    return _glue


def mark_written(tensors):
    """A kernel launched through ctypes wrote over `tensors` (one tensor or a sequence): bump their version counters, as an
    in-place torch op would have -- ScalingFunction's lazy arg indices and autograd's saved-tensor check rely on it."""
    import torch
    first = tensors if isinstance(tensors, torch.Tensor) else (tensors[0] if len(tensors) else None)
    if first is not None and not first.is_cuda:
        torch.autograd.graph.increment_version(tensors if isinstance(tensors, torch.Tensor) else tuple(tensors))
        return
    glue().mark_written(tensors)


def check(code):
    if code != 0:
        msg = load().qd_error_string(code)
        raise RuntimeError('qd_hip: %s (code %d)' % (msg.decode() if msg else '?', code))


# ---------------------------------------------------------------- torch plumbing (memory, stream)
_workspaces = {}


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream on `device` (an int pointer).  Goes straight to
    the C++ binding: torch.cuda.current_stream() builds a Stream object per call (~3 us), which is
    a third of the host cost of a small-tensor launch."""
    import torch
    idx = device.index if (device is not None and device.index is not None) else torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


def workspace(device):
    """One scratch buffer per (device, stream); kernels on a stream are ordered, so sharing it
    between consecutive calls on that stream is safe."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    ws = _workspaces.get(key)
    if ws is None:
        nbytes = int(load().qd_workspace_bytes())
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def on_other_device(t):
    """True when `t` lives on a HIP device that is not the current one: kernels must be launched
    with that device current (callers re-enter themselves under torch.cuda.device(t.device))."""
    import torch
    return t.is_cuda and t.device.index is not None and t.device.index != torch.cuda.current_device()


_host_scratch = None


def workspace_for(t):
    """Scratch buffer for a call on `t`'s device: the per-stream device workspace, or (CPU) a token buffer -- libqd_host.so
    needs none."""
    global _host_scratch
    if t.is_cuda:
        return workspace(t.device)
    if _host_scratch is None:
        import torch
        _host_scratch = torch.empty(16, dtype=torch.uint8)
    return _host_scratch


_libc_madvise = None


def fresh_host_output(n, dtype, like=None):
    """A new, untouched CPU tensor for a kernel of libqd_host.so to fill.  From 4 MiB up its pages are advised to be huge ones
    (madvise(MADV_HUGEPAGE) on the 2 MiB-aligned inside of the allocation, as numpy does for its own large arrays): the first
    touch of a fresh 256 MiB result is 65536 page faults with 4 KiB pages, 128 with 2 MiB ones -- on a 64-thread host that is the
    difference between 13 and 25+ GB/s for the whole call.  Advice only: wherever the kernel declines it, nothing changes."""
    global _libc_madvise
    import torch
    t = torch.empty(n, dtype=dtype) if like is None else torch.empty_like(like)
    nbytes = t.numel() * t.element_size()
    if nbytes >= (4 << 20):
        try:
            if _libc_madvise is False:
                return t
            if _libc_madvise is None:
                libc = ctypes.CDLL(None, use_errno=True)
                libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                libc.madvise.restype = ctypes.c_int
                _libc_madvise = libc.madvise
            huge = 2 << 20
            lo = (t.data_ptr() + huge - 1) & ~(huge - 1)
            hi = (t.data_ptr() + nbytes) & ~(huge - 1)
            if hi > lo:
                _libc_madvise(lo, hi - lo, 14)              # MADV_HUGEPAGE
        except (OSError, AttributeError):
            _libc_madvise = False if _libc_madvise is None else _libc_madvise
    return t


def stream_for(t):
    """hipStream_t of the current stream on `t`'s device, None for a CPU tensor (the host call is complete when it returns)."""
    return stream_ptr(t.device) if t.is_cuda else None


def require_f32(t, what='tensor'):
    """A float32 tensor on a HIP device or on the CPU (the entry points that exist in both libraries)."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %r' % (what, type(t)))
    if not (t.is_cuda or t.device.type == 'cpu'):
        raise RuntimeError('quantized_distillation_amd: %s must live on a HIP device or on the CPU (got %s)' % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32 (the reference path is fp32-only), got %s' % (what, t.dtype))


def require_device_f32(t, what='tensor'):
    """A float32 tensor on a HIP device: the entry points that exist for device tensors only (multi-tensor launches, the
    packed codec, device histograms and order statistics)."""
    import torch
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor, got %r' % (what, type(t)))
    if not t.is_cuda:
        raise RuntimeError('quantized_distillation_amd: %s must live on a HIP device (got %s); '
                           'this entry point has no CPU path' % (what, t.device))
    if t.dtype != torch.float32:
        raise TypeError('%s must be float32 (the reference path is fp32-only), got %s' % (what, t.dtype))
