// _qd_glue -- the per-call binding of the hot entry points of libqd_hip.so for Python/PyTorch callers.
//
// The drop-in boundary is the C ABI of include/qd_hip.h (raw device pointers, no torch types).  The
// reference's training loops call the quantizer once per parameter tensor per step
// (cnn_models/conv_forward_model.py:235-247: 22-110 calls per step, most of them on tiny tensors), so
// for those loops the cost of a call is the HOST cost of getting from a torch.Tensor to that ABI: with
// ctypes that was 11-13 us per call (two torch allocations, ~16 argument conversions, a Python-side
// ScalingFunction).  This file is the same binding written against the CPython C API and ATen directly:
// unpack the tensor, allocate the outputs from torch's caching allocator, take torch's current HIP
// stream, call the C ABI -- nothing else.  It contains no kernels and no arithmetic; every entry point
// forwards to exactly one qd_* function (two when subtract_mean adds qd_mean_f32).
//
// PyTorch is used for device memory and the current stream only.
#include <Python.h>

#include <ATen/ATen.h>
#include <ATen/hip/EmptyTensor.h>
#include <c10/hip/HIPFunctions.h>
#include <c10/hip/HIPGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/Exceptions.h>
#include <torch/csrc/autograd/python_variable.h>

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdint>
#include <string>
#include <vector>

#include "qd_hip.h"

namespace {

struct Workspace {
    int device;
    void* stream;
    at::Tensor buf;
};
std::vector<Workspace> g_workspaces;   // one scratch buffer per (device, stream): kernels on a stream are ordered
size_t g_workspace_bytes = 0;

inline at::Tensor empty_f32(c10::IntArrayRef sizes, const c10::Device& dev) {
    return at::Tensor(at::detail::empty_cuda(sizes, at::kFloat, dev, c10::nullopt));
}
inline at::Tensor empty_of(c10::IntArrayRef sizes, at::ScalarType t, const c10::Device& dev) {
    return at::Tensor(at::detail::empty_cuda(sizes, t, dev, c10::nullopt));
}

void* workspace_for(const c10::Device& dev, void* stream, size_t* bytes) {
    for (auto& w : g_workspaces)
        if (w.device == dev.index() && w.stream == stream) {
            *bytes = g_workspace_bytes;
            return w.buf.data_ptr();
        }
    if (g_workspace_bytes == 0) g_workspace_bytes = qd_workspace_bytes();
    at::Tensor buf = empty_of({static_cast<int64_t>(g_workspace_bytes)}, at::kByte, dev);
    g_workspaces.push_back({static_cast<int>(dev.index()), stream, buf});
    *bytes = g_workspace_bytes;
    return buf.data_ptr();
}

void check_rc(int rc) {
    if (rc != 0) {
        const char* msg = qd_error_string(rc);
        throw std::runtime_error(std::string("qd_hip: ") + (msg ? msg : "?") + " (code " + std::to_string(rc) + ")");
    }
}

const at::Tensor& tensor_arg(PyObject* o, const char* what) {
    if (!THPVariable_Check(o)) {
        PyErr_Format(PyExc_TypeError, "%s must be a torch.Tensor, got %s", what, Py_TYPE(o)->tp_name);
        throw python_error();
    }
    return THPVariable_Unpack(o);
}

// A kernel has written over `t` through its raw pointer: tell torch.  The version counter is what
// ScalingFunction's lazily computed arg-min/max indices check before they trust a retained tensor (quant_functions.py:
// _compute_arg_indices) and what autograd checks on tensors saved for backward; views share the counter of their base.
// (Inference tensors carry no counter and cannot be written in place outside inference mode.)
inline void mark_written(const at::Tensor& t) {
    if (t.defined() && !t.is_inference()) t.unsafeGetTensorImpl()->bump_version();
}

void require_device_f32(const at::Tensor& t, const char* what) {
    if (!t.is_cuda())
        throw std::runtime_error(std::string("quantized_distillation_amd: ") + what +
                                 " must live on a HIP device (got " + t.device().str() + "); this binding has no CPU path (CPU tensors are served by libqd_host.so through the Python API)");
    if (t.scalar_type() != at::kFloat) {
        PyErr_Format(PyExc_TypeError, "%s must be float32 (the reference path is fp32-only), got %s", what,
                     c10::toString(t.scalar_type()));
        throw python_error();
    }
}

inline int64_t num_buckets(int64_t n, int64_t bucket) { return (bucket <= 0 || n < bucket) ? 1 : (n + bucket - 1) / bucket; }

// uniform(x, levels, bucket, clamp, max_element, stochastic, seed, subtract_mean, in_place)
//   -> (q, ab, mean | None, n, x_read)
//   ab: [2, nb, 1] (bucket > 0) or [2, 1] (bucket == 0): row 0 = alpha, row 1 = beta, shaped as the reference's
//   min/max(keepdim=True) results (quant_functions.py:85-92).  One qd_uniform_f32 launch.
PyObject* glue_uniform(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    HANDLE_TH_ERRORS
    if (nargs != 9) {
        PyErr_SetString(PyExc_TypeError, "uniform() takes 9 positional arguments");
        return nullptr;
    }
    const at::Tensor& x0 = tensor_arg(args[0], "tensor");
    require_device_f32(x0, "tensor");
    const long levels = PyLong_AsLong(args[1]);
    const long long bucket = PyLong_AsLongLong(args[2]);
    const int clamp = PyObject_IsTrue(args[3]);
    const double max_element = PyFloat_AsDouble(args[4]);
    const int stochastic = PyObject_IsTrue(args[5]);
    const unsigned long long seed = PyLong_AsUnsignedLongLongMask(args[6]);
    const int subtract_mean = PyObject_IsTrue(args[7]);
    const int in_place = PyObject_IsTrue(args[8]);
    if (PyErr_Occurred()) return nullptr;

    at::Tensor x = x0;
    if (!x.is_contiguous()) {
        if (in_place) {
            PyErr_SetString(PyExc_ValueError, "modify_in_place=True needs a contiguous tensor");
            return nullptr;
        }
        x = x.contiguous();
    }
    const c10::Device dev = x.device();
    c10::hip::OptionalHIPGuard guard;
    if (dev.index() != c10::hip::current_device()) guard.set_index(dev.index());
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();

    const int64_t n = x.numel();
    const int64_t nb = num_buckets(n, bucket);
    at::Tensor q = in_place ? x : at::Tensor(at::detail::empty_cuda(x.sizes(), at::kFloat, dev, c10::nullopt));
    at::Tensor ab = bucket > 0 ? empty_f32({2, nb, 1}, dev) : empty_f32({2, 1}, dev);
    at::Tensor mean;
    float* mean_ptr = nullptr;
    if (n > 0) {
        void* ws = nullptr;
        size_t ws_bytes = 0;
        if (nb == 1 || subtract_mean) ws = workspace_for(dev, stream, &ws_bytes);
        if (subtract_mean) {
            mean = empty_f32({1}, dev);
            mean_ptr = mean.data_ptr<float>();
            check_rc(qd_mean_f32(x.data_ptr<float>(), n, mean_ptr, ws, ws_bytes, stream));
        }
        float* abp = ab.data_ptr<float>();
        check_rc(qd_uniform_f32(x.data_ptr<float>(), q.data_ptr<float>(), n, bucket, static_cast<int>(levels), abp, abp + nb,
                                nullptr, mean_ptr, clamp, static_cast<float>(max_element), stochastic, seed,
                                nb == 1 ? ws : nullptr, nb == 1 ? ws_bytes : 0, stream));
        if (in_place) mark_written(x);
    } else {                                               // nothing to scale: defined values, not uninitialised memory
        ab.zero_();
        if (subtract_mean) {
            mean = empty_f32({1}, dev);
            mean.zero_();
        }
    }
    PyObject* out = PyTuple_New(5);
    PyTuple_SET_ITEM(out, 0, THPVariable_Wrap(q));
    PyTuple_SET_ITEM(out, 1, THPVariable_Wrap(ab));
    if (mean.defined()) {
        PyTuple_SET_ITEM(out, 2, THPVariable_Wrap(mean));
    } else {
        Py_INCREF(Py_None);
        PyTuple_SET_ITEM(out, 2, Py_None);
    }
    PyTuple_SET_ITEM(out, 3, PyLong_FromLongLong(n));
    if (x.is_same(x0)) {                                   // the tensor that was read (the input itself when contiguous)
        Py_INCREF(args[0]);
        PyTuple_SET_ITEM(out, 4, args[0]);
    } else {
        PyTuple_SET_ITEM(out, 4, THPVariable_Wrap(x));
    }
    return out;
    END_HANDLE_TH_ERRORS
}

// ---- the common case of uniformQuantization as ONE native call ---------------------------------------------------
// uniform_common(tensor, s, bucket_size | None) -> (q, ScalingFunction) | None
//
// The reference's training loops call uniformQuantization(p.data, s, bucket_size=B) once per parameter tensor per step
// (cnn_models/conv_forward_model.py:235-247) and drop the ScalingFunction.  For exactly that configuration -- linear
// scaling, deterministic rounding, no clamp, no mean, out of place, contiguous fp32 device tensor -- this entry point
// does everything the Python function would do: allocate q, take alpha/beta from a per-(device, stream) SLAB (a bump
// pointer instead of a second trip through the caching allocator: only the slab and an offset are recorded, the views
// are made when sf.alpha / sf.beta are read), launch qd_uniform_f32, and build the ScalingFunction instance directly
// (its class-level defaults cover every field the call does not set).  Anything else returns None and the caller takes
// the general path, which also raises the reference's exceptions for bad arguments.
PyTypeObject* g_sf_type = nullptr;
PyObject *s_bucket_size, *s_n, *s_ab, *s_ab_slab, *s_ab_off, *s_arg_source, *s_arg_version, *s_mean_tensor, *s_zero;

struct Slab {
    int device;
    void* stream;
    at::Tensor t;      // [kSlabFloats] fp32
    PyObject* py;      // its Python wrapper (one reference held here; every ScalingFunction carved from it holds one too)
    int64_t used;
};
std::vector<Slab> g_slabs;
constexpr int64_t kSlabFloats = 1 << 18;          // 1 MiB; requests above a quarter of it get their own allocation

PyObject* glue_register(PyObject*, PyObject* type) {
    if (!PyType_Check(type)) {
        PyErr_SetString(PyExc_TypeError, "register() takes the ScalingFunction class");
        return nullptr;
    }
    Py_INCREF(type);
    Py_XDECREF(reinterpret_cast<PyObject*>(g_sf_type));
    g_sf_type = reinterpret_cast<PyTypeObject*>(type);
    if (!s_n) {
        s_bucket_size = PyUnicode_InternFromString("bucket_size");
        s_n = PyUnicode_InternFromString("_n");
        s_ab = PyUnicode_InternFromString("_ab");
        s_ab_slab = PyUnicode_InternFromString("_ab_slab");
        s_ab_off = PyUnicode_InternFromString("_ab_off");
        s_arg_source = PyUnicode_InternFromString("_arg_source");
        s_arg_version = PyUnicode_InternFromString("_arg_version");
        s_mean_tensor = PyUnicode_InternFromString("mean_tensor");
        s_zero = PyLong_FromLong(0);
    }
    Py_RETURN_NONE;
}

PyObject* glue_uniform_common(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    HANDLE_TH_ERRORS
    if (nargs != 3 || !g_sf_type || !THPVariable_Check(args[0]) || !PyLong_CheckExact(args[1])) Py_RETURN_NONE;
    const at::Tensor& x = THPVariable_Unpack(args[0]);
    if (!x.is_cuda() || x.scalar_type() != at::kFloat || !x.is_contiguous() || x.is_inference()) Py_RETURN_NONE;
    const long levels = PyLong_AsLong(args[1]);
    long long bucket = 0;
    if (args[2] != Py_None) {
        if (!PyLong_CheckExact(args[2])) Py_RETURN_NONE;
        bucket = PyLong_AsLongLong(args[2]);
        if (bucket <= 0) bucket = -1;
    }
    if (PyErr_Occurred()) {                               // out-of-range integers: the general path reports them
        PyErr_Clear();
        Py_RETURN_NONE;
    }
    const int64_t n = x.numel();
    if (levels < 2 || levels > 0x7fffffffL || bucket < 0 || n == 0) Py_RETURN_NONE;

    const c10::Device dev = x.device();
    c10::hip::OptionalHIPGuard guard;
    if (dev.index() != c10::hip::current_device()) guard.set_index(dev.index());
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    const int64_t nb = num_buckets(n, bucket);
    at::Tensor q(at::detail::empty_cuda(x.sizes(), at::kFloat, dev, c10::nullopt));

    // alpha / beta: 2 * nb floats from the slab of this (device, stream), 16-byte granules
    const int64_t need = (2 * nb + 3) & ~int64_t(3);
    float* abp = nullptr;
    PyObject* ab_owner = nullptr;                          // new reference: the slab's wrapper or a dedicated [2, nb, 1] tensor
    int64_t ab_off = -1;
    if (need <= kSlabFloats / 4) {
        Slab* sl = nullptr;
        for (auto& c : g_slabs)
            if (c.device == dev.index() && c.stream == stream) { sl = &c; break; }
        if (!sl) {
            if (g_slabs.size() >= 64) {                   // programs that keep creating streams: forget the oldest entry (its
                Py_XDECREF(g_slabs.front().py);           // slab lives on through whatever was carved from it)
                g_slabs.erase(g_slabs.begin());
            }
            g_slabs.push_back({static_cast<int>(dev.index()), stream, at::Tensor(), nullptr, kSlabFloats});
            sl = &g_slabs.back();
        }
        if (sl->used + need > kSlabFloats) {               // a fresh slab; the old one lives as long as something carved from it
            at::Tensor t = empty_f32({kSlabFloats}, dev);
            PyObject* py = THPVariable_Wrap(t);
            if (!py) return nullptr;
            Py_XDECREF(sl->py);
            sl->t = std::move(t);
            sl->py = py;
            sl->used = 0;
        }
        ab_off = sl->used;
        sl->used += need;
        abp = sl->t.data_ptr<float>() + ab_off;
        ab_owner = sl->py;
        Py_INCREF(ab_owner);
    } else {
        at::Tensor ab = bucket > 0 ? empty_f32({2, nb, 1}, dev) : empty_f32({2, 1}, dev);
        abp = ab.data_ptr<float>();
        ab_owner = THPVariable_Wrap(ab);
        if (!ab_owner) return nullptr;
    }
    struct Drop { PyObject* o; ~Drop() { Py_XDECREF(o); } } drop_owner{ab_owner};

    void* ws = nullptr;
    size_t ws_bytes = 0;
    if (nb == 1) ws = workspace_for(dev, stream, &ws_bytes);
    check_rc(qd_uniform_f32(x.data_ptr<float>(), q.data_ptr<float>(), n, bucket, static_cast<int>(levels), abp, abp + nb, nullptr,
                            nullptr, 0, 0.0f, 0, 0, ws, ws_bytes, stream));

    // ScalingFunction instance without running __init__ (the checks it makes are the ones made above)
    PyObject* sf = g_sf_type->tp_alloc(g_sf_type, 0);
    if (!sf) return nullptr;
    Drop drop_sf{sf};
    PyObject** dp = _PyObject_GetDictPtr(sf);
    if (!dp) {
        PyErr_SetString(PyExc_TypeError, "ScalingFunction instances must have a __dict__");
        return nullptr;
    }
    PyObject* d = *dp;
    if (!d) {
        d = PyDict_New();
        if (!d) return nullptr;
        *dp = d;
    }
    PyObject* n_obj = PyLong_FromLongLong(n);
    PyObject* ver_obj = PyLong_FromLongLong(static_cast<long long>(x._version()));
    Drop drop_n{n_obj}, drop_ver{ver_obj};
    if (!n_obj || !ver_obj) return nullptr;
    int rc = PyDict_SetItem(d, s_bucket_size, args[2]) | PyDict_SetItem(d, s_n, n_obj) |
             PyDict_SetItem(d, s_arg_source, args[0]) | PyDict_SetItem(d, s_arg_version, ver_obj) |
             PyDict_SetItem(d, s_mean_tensor, s_zero);                               // ref: :70
    if (ab_off >= 0) {
        PyObject* off_obj = PyLong_FromLongLong(ab_off);
        Drop drop_off{off_obj};
        if (!off_obj) return nullptr;
        rc |= PyDict_SetItem(d, s_ab_slab, ab_owner) | PyDict_SetItem(d, s_ab_off, off_obj);
    } else {
        rc |= PyDict_SetItem(d, s_ab, ab_owner);
    }
    if (rc) return nullptr;
    PyObject* qo = THPVariable_Wrap(q);
    if (!qo) return nullptr;
    PyObject* out = PyTuple_New(2);
    if (!out) { Py_DECREF(qo); return nullptr; }
    PyTuple_SET_ITEM(out, 0, qo);
    drop_sf.o = nullptr;                                    // the tuple owns it now
    PyTuple_SET_ITEM(out, 1, sf);
    return out;
    END_HANDLE_TH_ERRORS
}

// nearest(x, prescaled, points, assign_mode, n, bucket, alpha, beta, mean | None, clamp, max_element, idx_bytes)
//   -> (q [n], idx [n] int64 | uint8).  alpha/beta are outputs when prescaled == 0, inputs otherwise.
//   One qd_nearest_point_f32 launch.
PyObject* glue_nearest(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    HANDLE_TH_ERRORS
    if (nargs != 12 && nargs != 13) {
        PyErr_SetString(PyExc_TypeError, "nearest() takes 12 or 13 positional arguments");
        return nullptr;
    }
    const at::Tensor& x = tensor_arg(args[0], "tensor");
    const int prescaled = PyObject_IsTrue(args[1]);
    const at::Tensor& points = tensor_arg(args[2], "points");
    const long assign_mode = PyLong_AsLong(args[3]);
    const long long n = PyLong_AsLongLong(args[4]);
    const long long bucket = PyLong_AsLongLong(args[5]);
    const at::Tensor& alpha = tensor_arg(args[6], "alpha");
    const at::Tensor& beta = tensor_arg(args[7], "beta");
    const float* mean_ptr = args[8] == Py_None ? nullptr : tensor_arg(args[8], "mean").data_ptr<float>();
    const int clamp = PyObject_IsTrue(args[9]);
    const double max_element = PyFloat_AsDouble(args[10]);
    const long idx_bytes = PyLong_AsLong(args[11]);
    const int in_place = nargs == 13 ? PyObject_IsTrue(args[12]) : 0;      // q written over x (modify_in_place=True): no copy pass
    if (PyErr_Occurred()) return nullptr;
    require_device_f32(x, "tensor");
    require_device_f32(points, "points");
    if (!x.is_contiguous() || !points.is_contiguous() || x.numel() < n) {
        PyErr_SetString(PyExc_ValueError, "nearest(): tensor and points must be contiguous and hold n elements");
        return nullptr;
    }
    const c10::Device dev = x.device();
    c10::hip::OptionalHIPGuard guard;
    if (dev.index() != c10::hip::current_device()) guard.set_index(dev.index());
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    at::Tensor q = in_place ? x.view({x.numel()}).narrow(0, 0, static_cast<int64_t>(n)) : empty_f32({static_cast<int64_t>(n)}, dev);
    at::Tensor idx = empty_of({static_cast<int64_t>(n)}, idx_bytes == 8 ? at::kLong : at::kByte, dev);
    if (n > 0) {
        void* ws = nullptr;
        size_t ws_bytes = 0;
        if (num_buckets(n, bucket) == 1) ws = workspace_for(dev, stream, &ws_bytes);
        check_rc(qd_nearest_point_f32(x.data_ptr<float>(), prescaled, points.data_ptr<float>(), static_cast<int>(points.numel()),
                                      static_cast<int>(assign_mode), q.data_ptr<float>(), idx.data_ptr(),
                                      static_cast<int>(idx_bytes), n, bucket, alpha.data_ptr<float>(), beta.data_ptr<float>(),
                                      mean_ptr, clamp, static_cast<float>(max_element), ws, ws_bytes, stream));
        if (in_place) mark_written(x);
    }
    PyObject* out = PyTuple_New(2);
    PyTuple_SET_ITEM(out, 0, THPVariable_Wrap(q));
    PyTuple_SET_ITEM(out, 1, THPVariable_Wrap(idx));
    return out;
    END_HANDLE_TH_ERRORS
}

// point_grad(g, idx, alpha, bucket, k) -> grad_points [k].  One qd_point_grad_f32 call (two launches inside).
PyObject* glue_point_grad(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    HANDLE_TH_ERRORS
    if (nargs != 5) {
        PyErr_SetString(PyExc_TypeError, "point_grad() takes 5 positional arguments");
        return nullptr;
    }
    const at::Tensor& g0 = tensor_arg(args[0], "grad_output");
    const at::Tensor& idx = tensor_arg(args[1], "indices");
    const at::Tensor& alpha = tensor_arg(args[2], "alpha");
    const long long bucket = PyLong_AsLongLong(args[3]);
    const long k = PyLong_AsLong(args[4]);
    if (PyErr_Occurred()) return nullptr;
    require_device_f32(g0, "grad_output");
    at::Tensor g = g0.is_contiguous() ? g0 : g0.contiguous();
    const int64_t n = g.numel();
    if (idx.numel() != n) {
        PyErr_SetString(PyExc_ValueError, "grad_output must have as many elements as the quantized tensor");
        return nullptr;
    }
    const int idx_bytes = idx.scalar_type() == at::kLong ? 8 : 1;
    const c10::Device dev = g.device();
    c10::hip::OptionalHIPGuard guard;
    if (dev.index() != c10::hip::current_device()) guard.set_index(dev.index());
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    at::Tensor gp = empty_f32({static_cast<int64_t>(k)}, dev);
    size_t ws_bytes = 0;
    void* ws = workspace_for(dev, stream, &ws_bytes);
    check_rc(qd_point_grad_f32(g.data_ptr<float>(), idx.data_ptr(), idx_bytes, alpha.data_ptr<float>(), n, bucket,
                               static_cast<int>(k), gp.data_ptr<float>(), ws, ws_bytes, stream));
    return THPVariable_Wrap(gp);
    END_HANDLE_TH_ERRORS
}

// the version of include/qd_hip.h THIS module was compiled against (a stale _qd_glue.so next to a newer libqd_hip.so, or the
// other way round, must not pass: _lib.glue() compares it with the library's qd_abi_version())
PyObject* glue_abi_version(PyObject*, PyObject*) { return PyLong_FromLong(QD_ABI_VERSION); }

// host_cost_probe(x, levels, bucket, iters) -> (us per bare C-ABI launch, us per output allocation pair, us per launch with
// allocation).  Measurement aid for docs/history/tools/profile_api_overhead.py: where the per-call host time of uniform() goes.
PyObject* glue_host_cost_probe(PyObject*, PyObject* const* args, Py_ssize_t nargs) {
    HANDLE_TH_ERRORS
    if (nargs != 4) {
        PyErr_SetString(PyExc_TypeError, "host_cost_probe() takes 4 positional arguments");
        return nullptr;
    }
    const at::Tensor& x = tensor_arg(args[0], "tensor");
    require_device_f32(x, "tensor");
    const long levels = PyLong_AsLong(args[1]);
    const long long bucket = PyLong_AsLongLong(args[2]);
    const long iters = PyLong_AsLong(args[3]);
    if (PyErr_Occurred()) return nullptr;
    const c10::Device dev = x.device();
    void* stream = c10::hip::getCurrentHIPStream(dev.index()).stream();
    const int64_t n = x.numel(), nb = num_buckets(n, bucket);
    at::Tensor q = empty_f32({n}, dev), ab = empty_f32({2, nb, 1}, dev);
    size_t ws_bytes = 0;
    void* ws = workspace_for(dev, stream, &ws_bytes);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::micro>(b - a).count();
    };
    double t_launch = 0, t_alloc = 0, t_both = 0;
    for (long done = 0; done < iters; done += 64) {      // keep the device queue shallow: sync outside the timed parts
        auto a = now();
        for (int i = 0; i < 64; ++i)
            check_rc(qd_uniform_f32(x.data_ptr<float>(), q.data_ptr<float>(), n, bucket, (int)levels, ab.data_ptr<float>(),
                                    ab.data_ptr<float>() + nb, nullptr, nullptr, 0, 0.f, 0, 0, ws, ws_bytes, stream));
        auto b = now();
        t_launch += us(a, b);
        (void)hipStreamSynchronize((hipStream_t)stream);
        a = now();
        for (int i = 0; i < 64; ++i) {
            at::Tensor q2 = at::Tensor(at::detail::empty_cuda(x.sizes(), at::kFloat, dev, c10::nullopt));
            at::Tensor ab2 = empty_f32({2, nb, 1}, dev);
        }
        b = now();
        t_alloc += us(a, b);
        a = now();
        for (int i = 0; i < 64; ++i) {
            at::Tensor q2 = at::Tensor(at::detail::empty_cuda(x.sizes(), at::kFloat, dev, c10::nullopt));
            at::Tensor ab2 = empty_f32({2, nb, 1}, dev);
            check_rc(qd_uniform_f32(x.data_ptr<float>(), q2.data_ptr<float>(), n, bucket, (int)levels, ab2.data_ptr<float>(),
                                    ab2.data_ptr<float>() + nb, nullptr, nullptr, 0, 0.f, 0, 0, ws, ws_bytes, stream));
        }
        b = now();
        t_both += us(a, b);
        (void)hipStreamSynchronize((hipStream_t)stream);
    }
    const double calls = (double)((iters + 63) / 64 * 64);
    return Py_BuildValue("(ddd)", t_launch / calls, t_alloc / calls, t_both / calls);
    END_HANDLE_TH_ERRORS
}

// mark_written(t | sequence of tensors): for the entry points bound with ctypes (in-place scale_down / inv_scale_down, the
// STE kernels, the multi-tensor launches), whose outputs torch has not seen being written.
PyObject* glue_mark_written(PyObject*, PyObject* arg) {
    HANDLE_TH_ERRORS
    if (THPVariable_Check(arg)) {
        mark_written(THPVariable_Unpack(arg));
    } else {
        PyObject* seq = PySequence_Fast(arg, "mark_written() takes a tensor or a sequence of tensors");
        if (!seq) return nullptr;
        const Py_ssize_t m = PySequence_Fast_GET_SIZE(seq);
        for (Py_ssize_t i = 0; i < m; ++i) {
            PyObject* o = PySequence_Fast_GET_ITEM(seq, i);
            if (!THPVariable_Check(o)) {
                Py_DECREF(seq);
                PyErr_SetString(PyExc_TypeError, "mark_written() takes a tensor or a sequence of tensors");
                return nullptr;
            }
            mark_written(THPVariable_Unpack(o));
        }
        Py_DECREF(seq);
    }
    Py_RETURN_NONE;
    END_HANDLE_TH_ERRORS
}

PyMethodDef methods[] = {
    {"mark_written", glue_mark_written, METH_O, "mark_written(tensor | [tensors]): bump the version counter of tensors a kernel wrote through raw pointers"},
    {"uniform", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(glue_uniform)), METH_FASTCALL,
     "uniform(x, levels, bucket, clamp, max_element, stochastic, seed, subtract_mean, in_place) -> (q, ab, mean, n, x_read)"},
    {"uniform_common", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(glue_uniform_common)), METH_FASTCALL,
     "uniform_common(tensor, s, bucket_size | None) -> (q, ScalingFunction) for the training loops' configuration, None otherwise"},
    {"register", glue_register, METH_O, "register(ScalingFunction): the class uniform_common instantiates"},
    {"nearest", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(glue_nearest)), METH_FASTCALL,
     "nearest(x, prescaled, points, assign_mode, n, bucket, alpha, beta, mean, clamp, max_element, idx_bytes) -> (q, idx)"},
    {"point_grad", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(glue_point_grad)), METH_FASTCALL,
     "point_grad(g, idx, alpha, bucket, k) -> grad_points"},
    {"host_cost_probe", reinterpret_cast<PyCFunction>(reinterpret_cast<void (*)()>(glue_host_cost_probe)), METH_FASTCALL,
     "host_cost_probe(x, levels, bucket, iters) -> (launch us, allocation us, both us): host time per call"},
    {"abi_version", glue_abi_version, METH_NOARGS, "QD_ABI_VERSION of the include/qd_hip.h this module was compiled against"},
    {nullptr, nullptr, 0, nullptr}};

PyModuleDef module = {PyModuleDef_HEAD_INIT, "_qd_glue",
                      "CPython/ATen binding of the per-call entry points of libqd_hip.so (include/qd_hip.h)", -1, methods,
                      nullptr, nullptr, nullptr, nullptr};

}  // namespace

PyMODINIT_FUNC PyInit__qd_glue(void) { return PyModule_Create(&module); }
