"""ctypes binding of oracle/qd_oracle.c -- TEST INFRASTRUCTURE ONLY (see that file's header).

Used where the numpy oracle would be too slow (full 64 Mi-element parity checks) and as the
`cpu_baseline` leg of bench.py.  Never imported by the product.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libqd_oracle.so')
_lib = None

_f = ctypes.POINTER(ctypes.c_float)
_i64 = ctypes.POINTER(ctypes.c_int64)
_i32 = ctypes.POINTER(ctypes.c_int32)
_d = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    src = os.path.join(_HERE, 'qd_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'all'])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.qdo_max_threads.restype = ctypes.c_int
        L.qdo_set_threads.argtypes = [ctypes.c_int]
        L.qdo_num_buckets.restype = ctypes.c_int64
        L.qdo_num_buckets.argtypes = [ctypes.c_int64, ctypes.c_int64]
        L.qdo_mean_f32.restype = ctypes.c_float
        L.qdo_mean_f32.argtypes = [_f, ctypes.c_int64]
        L.qdo_uniform_f32.restype = None
        L.qdo_uniform_f32.argtypes = [_f, _f, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _f, _f, _i64, _i64, _i32,
                                      ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float]
        L.qdo_scale_down_f32.restype = None
        L.qdo_scale_down_f32.argtypes = [_f, _f, ctypes.c_int64, ctypes.c_int64, _f, _f, _i64, _i64,
                                         ctypes.c_int, ctypes.c_float, ctypes.c_int, ctypes.c_float]
        L.qdo_nonuniform_f32.restype = None
        L.qdo_nonuniform_f32.argtypes = [_f, _f, ctypes.c_int, ctypes.c_int, _f, _i64, ctypes.c_int64,
                                         ctypes.c_int64, _f, _f]
        L.qdo_point_grad_f32.restype = None
        L.qdo_point_grad_f32.argtypes = [_f, _i64, _f, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, _d, _d]
        L.qdo_ste_backward_f32.restype = None
        L.qdo_ste_backward_f32.argtypes = [_f, _f, _f, ctypes.c_int64, ctypes.c_int64, ctypes.c_int]
        L.qdo_checksum_f32.restype = None
        L.qdo_checksum_f32.argtypes = [_f, ctypes.c_int64, _d, _d]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t) if a is not None else None


def _c(x, dt=np.float32):
    return np.ascontiguousarray(x, dtype=dt)


def max_threads():
    return int(lib().qdo_max_threads())


def set_threads(n):
    lib().qdo_set_threads(int(n))


def num_buckets(n, bucket):
    return int(lib().qdo_num_buckets(n, bucket or 0))


def uniform_quantize(x, s, bucket=None, max_element=False, subtract_mean=False, mean=None,
                     want_idx=True, want_lev=True):
    x = _c(x)
    flat = x.reshape(-1)
    n = flat.size
    nb = num_buckets(n, bucket)
    q = np.empty(n, np.float32)
    alpha, beta = np.empty(nb, np.float32), np.empty(nb, np.float32)
    imin = np.empty(nb, np.int64) if want_idx else None
    imax = np.empty(nb, np.int64) if want_idx else None
    lev = np.empty(n, np.int32) if want_lev else None
    if subtract_mean and mean is None:
        mean = lib().qdo_mean_f32(_p(flat, _f), n)
    lib().qdo_uniform_f32(_p(flat, _f), _p(q, _f), n, bucket or 0, s, _p(alpha, _f), _p(beta, _f), _p(imin, _i64),
                          _p(imax, _i64), _p(lev, _i32), int(bool(subtract_mean)), float(mean or 0.0),
                          int(max_element is not False), float(max_element or 0.0))
    return dict(q=q.reshape(x.shape), alpha=alpha, beta=beta, imin=imin, imax=imax, lev=lev,
                mean=np.float32(mean or 0.0))


def scale_down(x, bucket=None, max_element=False, subtract_mean=False, mean=None):
    x = _c(x)
    flat = x.reshape(-1)
    n = flat.size
    nb = num_buckets(n, bucket)
    u = np.empty(n, np.float32)
    alpha, beta = np.empty(nb, np.float32), np.empty(nb, np.float32)
    imin, imax = np.empty(nb, np.int64), np.empty(nb, np.int64)
    if subtract_mean and mean is None:
        mean = lib().qdo_mean_f32(_p(flat, _f), n)
    lib().qdo_scale_down_f32(_p(flat, _f), _p(u, _f), n, bucket or 0, _p(alpha, _f), _p(beta, _f), _p(imin, _i64),
                             _p(imax, _i64), int(bool(subtract_mean)), float(mean or 0.0),
                             int(max_element is not False), float(max_element or 0.0))
    return dict(u=u, alpha=alpha, beta=beta, imin=imin, imax=imax, mean=np.float32(mean or 0.0))


def nonuniform_quantize(x, pts, bucket=None, mode='distance'):
    x = _c(x)
    pts = _c(pts)
    flat = x.reshape(-1)
    n = flat.size
    nb = num_buckets(n, bucket)
    q = np.empty(n, np.float32)
    idx = np.empty(n, np.int64)
    alpha, beta = np.empty(nb, np.float32), np.empty(nb, np.float32)
    lib().qdo_nonuniform_f32(_p(flat, _f), _p(pts, _f), pts.size, 0 if mode == 'distance' else 1, _p(q, _f),
                             _p(idx, _i64), n, bucket or 0, _p(alpha, _f), _p(beta, _f))
    return dict(q=q.reshape(x.shape), idx=idx.reshape(x.shape), alpha=alpha, beta=beta)


def point_grad(g, idx, alpha, bucket, k):
    g = _c(g).reshape(-1)
    idx = _c(idx, np.int64).reshape(-1)
    alpha = _c(alpha).reshape(-1)
    out, ab = np.empty(k, np.float64), np.empty(k, np.float64)
    lib().qdo_point_grad_f32(_p(g, _f), _p(idx, _i64), _p(alpha, _f), g.size, bucket or 0, k, _p(out, _d), _p(ab, _d))
    return out, ab


def ste_complicated_backward(x, g, s, bucket):
    x, g = _c(x), _c(g)
    out = np.empty(x.size, np.float32)
    lib().qdo_ste_backward_f32(_p(x.reshape(-1), _f), _p(g.reshape(-1), _f), _p(out, _f), x.size, bucket or 0, s)
    return out.reshape(x.shape)


def checksum(x):
    x = _c(x).reshape(-1)
    a, b = ctypes.c_double(), ctypes.c_double()
    lib().qdo_checksum_f32(_p(x, _f), x.size, ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value
