// qd_host.cpp -- libqd_host.so: the per-call entry points of include/qd_hip.h for HOST (CPU) fp32 tensors.
//
// The reference's quantization.* functions accept CPU tensors as well as CUDA ones (quantization/__init__.py:3 and the
// `is_cuda` branches of quant_functions.py:186,254,283-284,367-368,395-396,439); SURVEY.md 8b makes that part of the drop-in
// boundary.  This library is the product's path for them: same C-ABI names, argument order and meaning as libqd_hip.so
// (so the Python package binds either with one table of signatures), every pointer a HOST pointer, `stream` and
// `workspace` ignored (the call has completed when it returns), OpenMP over buckets / element blocks.
//
// It is NOT a fallback for device tensors: the Python package picks the library by the tensor's device, device tensors
// never come here, and a missing libqd_hip.so fails loudly whatever this file can do (quantized_distillation_amd/_lib.py).
// It is not the test oracle either (oracle/ is not linked, loaded or imported by anything in the package).
//
// Arithmetic: one separately rounded IEEE fp32 operation per reference tensor op, in the reference's order -- the same
// sequence the HIP kernels execute (csrc/qd_common.h: alpha_beta, qdq, qdq_stochastic) -- so results are equal to the device
// path on the same input, and bit-identical to the reference's CPU path for values and indices with ONE exception:
// subtract_mean=True.  qd_mean_f32 sums in float64 and rounds once (as the device path and the oracle do), the reference takes
// torch's fp32 tensor.mean(), whose cascade summation differs from it in the last bits on about half of all tensors; beta and
// alpha then differ too and a level can flip.  No driver of the reference sets subtract_mean.  Built with
// -ffp-contract=off -fno-fast-math (quantized_distillation_amd/build.py): no fused multiply-add, no reassociation.
//
// Entry points present: qd_mean_f32, qd_uniform_f32, qd_scale_down_f32, qd_inv_scale_f32, qd_bucket_argminmax_f32,
// qd_nearest_point_f32, qd_point_grad_f32, qd_ste_bucket_backward_f32, qd_clamp_f32, qd_truncated_ste_f32 and the host
// helpers.  The multi-tensor, codec, order-statistics and 'absmax' / 'absnorm' entry points exist for device tensors only.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../../include/qd_hip.h"

namespace {

constexpr float kTolDiffZero = 1e-10f;               // quant_functions.py:40, compared in fp32 as torch compares a float tensor with it
constexpr int64_t kChunk = 1 << 16;                   // elements per work item of the flat (bucket_size=None) loops and reductions
constexpr int64_t kParallelMin = 1 << 15;             // below this many elements a call runs on the calling thread

// threads worth waking for n elements: one per 64 Ki elements, at most what OpenMP would use (a 64-thread host quantizing an
// 800 K-element layer needs a dozen of them, not the whole team)
inline int threads_for(int64_t n) {
#ifdef _OPENMP
    const int64_t want = n / kChunk + 1;
    const int have = omp_get_max_threads();
    return (int)(want < have ? want : have);
#else
    (void)n;
    return 1;
#endif
}

inline void geometry(int64_t n, int64_t bucket, int64_t& nb, int64_t& row) {       // help_functions.py:67-94
    if (bucket <= 0 || n < bucket) { nb = 1; row = n; return; }
    row = bucket;
    nb = (n + bucket - 1) / bucket;
}

struct Prep {                                         // quant_functions.py:66-74: subtract the mean, then clamp
    float mean;                                       // 0 when subtract_mean is off: x - 0 is exact
    float me;                                         // +inf when max_element is off
    inline float operator()(float x) const {
        x = x - mean;
        x = x > me ? me : x;
        x = x < -me ? -me : x;
        return x;
    }
};
inline Prep make_prep(const float* mean, int clamp, float max_element) {
    Prep p;
    p.mean = mean ? *mean : 0.0f;
    p.me = clamp ? max_element : std::numeric_limits<float>::infinity();
    return p;
}

// min / max of prep(x[lo..hi)) with torch's NaN propagation (one NaN makes both NaN) -> alpha, beta (:85-99)
static void span_minmax(const float* x, int64_t lo, int64_t hi, float mean, float me, float* mn_out, float* mx_out, int* nan_out);
inline void range_minmax(const float* x, int64_t lo, int64_t hi, const Prep& pp, float& mn, float& mx, bool& nan) {
    int nans = 0;
    span_minmax(x, lo, hi, pp.mean, pp.me, &mn, &mx, &nans);
    nan = nans != 0;
}
inline void alpha_beta(float mn, float mx, bool nan, float& a, float& b) {
    if (nan) { mn = std::numeric_limits<float>::quiet_NaN(); mx = mn; }
    a = mx - mn;
    a = a < kTolDiffZero ? 1.0f : a;                  // (false for NaN: alpha stays NaN, as in the reference)
    b = mn;
}
// the same for one bucket that is the whole tensor: chunked, folded in a fixed order
void whole_minmax(const float* x, int64_t n, const Prep& pp, float& a, float& b) {
    const int64_t chunks = (n + kChunk - 1) / kChunk;
    std::vector<float> mns((size_t)chunks), mxs((size_t)chunks);
    std::vector<char> nans((size_t)chunks);
#pragma omp parallel for schedule(static) if (chunks > 1)
    for (int64_t c = 0; c < chunks; ++c) {
        const int64_t lo = c * kChunk, hi = lo + kChunk < n ? lo + kChunk : n;
        float mn, mx;
        bool nan;
        range_minmax(x, lo, hi, pp, mn, mx, nan);
        mns[(size_t)c] = mn; mxs[(size_t)c] = mx; nans[(size_t)c] = nan;
    }
    float mn = std::numeric_limits<float>::infinity(), mx = -mn;
    bool nan = false;
    for (int64_t c = 0; c < chunks; ++c) {
        mn = mns[(size_t)c] < mn ? mns[(size_t)c] : mn;
        mx = mxs[(size_t)c] > mx ? mxs[(size_t)c] : mx;
        nan |= nans[(size_t)c] != 0;
    }
    alpha_beta(mn, mx, nan, a, b);
}

// ---- Philox4x32-7, the generator of the device's stochastic-rounding branch (csrc/qd_common.h): element e draws component
// e & 3 of the block e >> 2, so a CPU tensor and a device tensor quantized with the same seed round the same way
inline void philox_uniform4(uint64_t seed, uint64_t block, float (&out)[4]) {
    uint32_t c[4] = {(uint32_t)block, (uint32_t)(block >> 32), 0x51ed270bu, 0x2545f491u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 7; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    for (int i = 0; i < 4; ++i) out[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}
inline float philox_uniform(uint64_t seed, int64_t e) {
    float r4[4];
    philox_uniform4(seed, (uint64_t)e >> 2, r4);
    return r4[e & 3];
}

// rintf() without the libm call (the baseline x86-64 ISA has no rounding instruction, so std::nearbyintf is a function call
// per element and the loop does not vectorise): below 2^23 adding and subtracting 2^23 rounds to an integer in the current
// (default: nearest-even) mode, exactly; from 2^23 on every float is an integer.  The sign is kept (-0.4 -> -0.0, as rintf).
// NaN: the comparison is false, t itself is returned.  (No -ffast-math: the compiler may not fold (a + c) - c.)
inline float rint_even(float t) {
    const float a = std::fabs(t);
    const float r = (a + 8388608.0f) - 8388608.0f;
    return a < 8388608.0f ? std::copysign(r, t) : t;
}

// the k-level quantize-dequantize of one element: seven separately rounded fp32 ops (:106-107,189-191,142-148)
inline float qdq(float v, float a, float b, float sm1, float mean, float& level) {
    float u = v - b;
    u = u / a;
    float t = u * sm1;
    float r = rint_even(t);                           // round half to even, torch.round
    level = r;
    float w = r / sm1;
    float y = w * a;
    y = y + b;
    y = y + mean;
    return y;
}
// ---- the four streaming loops of the per-step calls as free functions, compiled three times (function multi-versioning: the
// dynamic linker picks the AVX-512 / AVX2 / baseline SSE2 clone for the CPU it runs on).  Same C source, same IEEE operations in
// every clone -- vector width is the only difference (-ffp-contract=off: no clone fuses a multiply-add) -- so the results are the
// same bits on every machine.
#if defined(__x86_64__) && defined(__GLIBC__) && defined(__GNUC__) && !defined(__clang__)
#define QD_CLONES __attribute__((target_clones("avx512f", "avx2", "default"), noinline))
#else
#define QD_CLONES                                     /* no ifunc multi-versioning here: one baseline build of each loop */
#endif

QD_CLONES static void span_minmax(const float* x, int64_t lo, int64_t hi, float mean, float me, float* mn_out, float* mx_out, int* nan_out) {
    float mn = std::numeric_limits<float>::infinity(), mx = -mn;
    int nans = 0;
#pragma omp simd reduction(min : mn) reduction(max : mx) reduction(| : nans)
    for (int64_t i = lo; i < hi; ++i) {
        float v = x[i] - mean;
        v = v > me ? me : v;
        v = v < -me ? -me : v;
        nans |= (v != v) ? 1 : 0;
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    *mn_out = mn; *mx_out = mx; *nan_out = nans;
}
// q may alias x exactly: element i is read before it is written, no dependence between iterations
QD_CLONES static void span_qdq(const float* x, float* q, int64_t lo, int64_t hi, float a, float b, float sm1, float mean, float me) {
#pragma omp simd
    for (int64_t i = lo; i < hi; ++i) {
        float v = x[i] - mean;
        v = v > me ? me : v;
        v = v < -me ? -me : v;
        float lev;
        q[i] = qdq(v, a, b, sm1, mean, lev);
    }
}
QD_CLONES static void span_scale(const float* x, float* u, int64_t lo, int64_t hi, float a, float b, float mean, float me) {
#pragma omp simd
    for (int64_t i = lo; i < hi; ++i) {
        float v = x[i] - mean;
        v = v > me ? me : v;
        v = v < -me ? -me : v;
        v = v - b;                                                                  // :106-107
        u[i] = v / a;
    }
}
QD_CLONES static void span_inv(const float* u, float* y, int64_t lo, int64_t hi, float a, float b, float m) {
#pragma omp simd
    for (int64_t i = lo; i < hi; ++i) {
        float r = u[i] * a;                                                         // :142-143, two ops
        r = r + b;
        r = r + m;                                                                  // :148
        y[i] = r;
    }
}

// stochastic variant (:174-187): floor + Bernoulli(frac)
inline float qdq_stochastic(float v, float a, float b, float sm1, float mean, float rnd, float& level) {
    float u = v - b;
    u = u / a;
    float t = u * sm1;
    float l = std::floor(t);
    float p = t - l;
    float w = l / sm1;
    float inc = (rnd <= p) ? (1.0f / sm1) : 0.0f;
    level = l + ((rnd <= p) ? 1.0f : 0.0f);
    w = w + inc;
    float y = w * a;
    y = y + b;
    y = y + mean;
    return y;
}

// nearest point, distance rule (:267-273): searchsorted-left, clip, step down when STRICTLY closer to the lower point
// (a NaN sorts after every point, as in numpy: the last point; its value is NaN whatever the index is)
inline int assign_distance(float u, const float* p, int k) {
    if (u != u) return k - 1;
    int lo = 0, hi = k;                               // first i with p[i] >= u
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (p[mid] < u) lo = mid + 1; else hi = mid; }
    int i = lo > k - 1 ? k - 1 : lo;
    if (i > 0 && std::fabs(u - p[i - 1]) < std::fabs(u - p[i])) i -= 1;
    return i;
}
// midpoint rule (SearchSorted.query, :531-563): #{m_j <= u}
inline int assign_midpoint(float u, const float* m, int km1) {
    if (u != u) return km1;
    int lo = 0, hi = km1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (m[mid] <= u) lo = mid + 1; else hi = mid; }
    return lo;
}

// run body(bucket, lo, hi, alpha, beta) over the buckets: in parallel over buckets when there are several, over element
// chunks of the one bucket otherwise (its alpha / beta are computed first, by whole_minmax or taken from `given`)
template <typename Stats, typename Body>
void for_buckets(int64_t n, int64_t nb, int64_t row, Stats stats, Body body) {
    if (nb == 1) {
        float a, b;
        stats(0, 0, n, a, b, true);
        const int64_t chunks = (n + kChunk - 1) / kChunk;
#pragma omp parallel for schedule(static) if (chunks > 1) num_threads(threads_for(n))
        for (int64_t c = 0; c < chunks; ++c) {
            const int64_t lo = c * kChunk, hi = lo + kChunk < n ? lo + kChunk : n;
            body(0, lo, hi, a, b);
        }
        return;
    }
    // (a parameter tensor of a few buckets -- biases, batch-norm vectors -- is not worth waking a thread team for)
#pragma omp parallel for schedule(static) if (n >= kParallelMin) num_threads(threads_for(n))
    for (int64_t bk = 0; bk < nb; ++bk) {
        const int64_t lo = bk * row, hi = lo + row < n ? lo + row : n;
        float a, b;
        stats(bk, lo, hi, a, b, false);
        body(bk, lo, hi, a, b);
    }
}

}  // namespace

#pragma GCC visibility push(default)          // (the file is compiled with -fvisibility=hidden: only the entry points leave the library)
extern "C" {

int qd_abi_version(void) { return QD_ABI_VERSION; }
const char* qd_target_arch(void) { return "host"; }
const char* qd_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case QD_ERR_INVALID_ARGUMENT: return "invalid argument";
        case QD_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
        case QD_ERR_UNSUPPORTED: return "not supported for host tensors";
        default: return "unknown error";
    }
}
size_t qd_workspace_bytes(void) { return 16; }        // nothing on the host needs caller-provided scratch

int64_t qd_num_buckets(int64_t n, int64_t bucket) { int64_t nb, row; geometry(n, bucket, nb, row); return nb; }
int64_t qd_padded_length(int64_t n, int64_t bucket) { int64_t nb, row; geometry(n, bucket, nb, row); return nb * row; }

// (the two symbols below exist in this library only: OpenMP threads of the host loops)
int qd_host_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void qd_host_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// mean(x): float64 accumulation in fixed chunks, one rounding (replaces tensor.mean(), :67)
int qd_mean_f32(const float* x, int64_t n, float* mean_out, void*, size_t, void*) {
    if (n <= 0 || !x || !mean_out) return QD_ERR_INVALID_ARGUMENT;
    const int64_t chunks = (n + kChunk - 1) / kChunk;
    std::vector<double> part((size_t)chunks);
#pragma omp parallel for schedule(static) if (chunks > 1)
    for (int64_t c = 0; c < chunks; ++c) {
        const int64_t lo = c * kChunk, hi = lo + kChunk < n ? lo + kChunk : n;
        double acc = 0.0;
        for (int64_t i = lo; i < hi; ++i) acc += (double)x[i];
        part[(size_t)c] = acc;
    }
    double acc = 0.0;
    for (int64_t c = 0; c < chunks; ++c) acc += part[(size_t)c];
    *mean_out = (float)(acc / (double)n);
    return 0;
}

// K1 / K1g: uniformQuantization, linear scaling (:155-194)
int qd_uniform_f32(const float* x, float* q, int64_t n, int64_t bucket, int levels, float* alpha, float* beta,
                   uint8_t* level_idx, const float* mean, int clamp, float max_element, int stochastic, uint64_t seed,
                   void*, size_t, void*) {
    if (n < 0 || bucket < 0 || levels < 2 || (n > 0 && !x)) return QD_ERR_INVALID_ARGUMENT;
    if (level_idx && levels > 256) return QD_ERR_INVALID_ARGUMENT;
    if (!q) return level_idx ? QD_ERR_UNSUPPORTED : QD_ERR_INVALID_ARGUMENT;       // the levels-only form is the device codec's
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const Prep pp = make_prep(mean, clamp, max_element);
    const float sm1 = (float)(levels - 1);                                          // :172
    auto stats = [&](int64_t bk, int64_t lo, int64_t hi, float& a, float& b, bool whole) {
        if (whole) {
            whole_minmax(x, n, pp, a, b);
        } else {
            float mn, mx;
            bool nan;
            range_minmax(x, lo, hi, pp, mn, mx, nan);
            alpha_beta(mn, mx, nan, a, b);
        }
        if (alpha) alpha[bk] = a;
        if (beta) beta[bk] = b;
    };
    auto body = [&](int64_t, int64_t lo, int64_t hi, float a, float b) {
        if (!stochastic && !level_idx) {
            // the per-step case: a branch-free vector loop (IEEE vector division)
            span_qdq(x, q, lo, hi, a, b, sm1, pp.mean, pp.me);
            return;
        }
        for (int64_t i = lo; i < hi; ++i) {
            float lev;
            const float v = pp(x[i]);
            q[i] = stochastic ? qdq_stochastic(v, a, b, sm1, pp.mean, philox_uniform(seed, i), lev) : qdq(v, a, b, sm1, pp.mean, lev);
            if (level_idx) level_idx[i] = (uint8_t)(int)lev;                        // (NaN -> 0, as the device's conversion)
        }
    };
    // a bucket's statistics need the whole bucket before any element of it is overwritten (in place): stats runs first
    for_buckets(n, nb, row, stats, body);
    return 0;
}

// K2: ScalingFunction.scale_down, linear (:56-107); u in the PADDED bucket layout
int qd_scale_down_f32(const float* x, float* u, int64_t n, int64_t bucket, float* alpha, float* beta, const float* mean,
                      int clamp, float max_element, void*, size_t, void*) {
    if (n < 0 || bucket < 0 || !alpha || !beta || (n > 0 && (!x || !u))) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    if (u == x && nb * row != n) return QD_ERR_INVALID_ARGUMENT;                    // in place only when no padding is needed
    const Prep pp = make_prep(mean, clamp, max_element);
    auto stats = [&](int64_t bk, int64_t lo, int64_t hi, float& a, float& b, bool whole) {
        if (whole) {
            whole_minmax(x, n, pp, a, b);
        } else {
            float mn, mx;
            bool nan;
            range_minmax(x, lo, hi, pp, mn, mx, nan);
            alpha_beta(mn, mx, nan, a, b);
        }
        alpha[bk] = a;
        beta[bk] = b;
    };
    auto body = [&](int64_t, int64_t lo, int64_t hi, float a, float b) {
        span_scale(x, u, lo, hi, a, b, pp.mean, pp.me);
        if (hi == n && nb > 1) {                                                    // padding: the scaled last element (help_functions.py:76-86)
            const float last = u[n - 1];
            for (int64_t i = n; i < nb * row; ++i) u[i] = last;
        }
    };
    for_buckets(n, nb, row, stats, body);
    return 0;
}

// K3: ScalingFunction.inv_scale_down, linear (:131-152): y = u alpha + beta (+ mean), padding dropped
int qd_inv_scale_f32(const float* u, float* y, int64_t n, int64_t bucket, const float* alpha, const float* beta,
                     const float* mean, void*) {
    if (n < 0 || bucket < 0 || (n > 0 && (!u || !y || !alpha || !beta))) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const float m = mean ? *mean : 0.0f;
    auto stats = [&](int64_t bk, int64_t, int64_t, float& a, float& b, bool) { a = alpha[bk]; b = beta[bk]; };
    auto body = [&](int64_t, int64_t lo, int64_t hi, float a, float b) {
        span_inv(u, y, lo, hi, a, b, m);
    };
    for_buckets(n, nb, row, stats, body);
    return 0;
}

// first-occurrence arg-min / arg-max of each bucket, relative to the bucket start (:85-90,103-104); a bucket that holds a NaN
// reports the position of its first NaN for both, as torch.min / max(dim) do
int qd_bucket_argminmax_f32(const float* x, int64_t n, int64_t bucket, const float* mean, int clamp, float max_element,
                            int64_t* argmin, int64_t* argmax, void*, size_t, void*) {
    if (n <= 0 || bucket < 0 || !x || !argmin || !argmax) return QD_ERR_INVALID_ARGUMENT;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const Prep pp = make_prep(mean, clamp, max_element);
#pragma omp parallel for schedule(static) if (nb > 1 && n >= kParallelMin)
    for (int64_t bk = 0; bk < nb; ++bk) {
        const int64_t lo = bk * row, hi = lo + row < n ? lo + row : n;
        float mn = pp(x[lo]), mx = mn;
        int64_t jmn = 0, jmx = 0, jnan = mn != mn ? 0 : -1;
        for (int64_t i = lo + 1; i < hi; ++i) {
            const float v = pp(x[i]);
            if (v != v && jnan < 0) jnan = i - lo;
            if (v < mn) { mn = v; jmn = i - lo; }                                   // strict: the first occurrence wins
            if (v > mx) { mx = v; jmx = i - lo; }
        }
        argmin[bk] = jnan >= 0 ? jnan : jmn;
        argmax[bk] = jnan >= 0 ? jnan : jmx;
    }
    return 0;
}

// K4 / K5: nearest-point (non-uniform) quantization (:196-290)
int qd_nearest_point_f32(const float* x, int prescaled, const float* points, int k, int assign_mode, float* q, void* idx,
                         int idx_bytes, int64_t n, int64_t bucket, float* alpha, float* beta, const float* mean, int clamp,
                         float max_element, void*, size_t, void*) {
    if (n < 0 || bucket < 0 || k < 1 || !points || (n > 0 && (!x || !alpha || !beta))) return QD_ERR_INVALID_ARGUMENT;
    if (assign_mode != QD_ASSIGN_DISTANCE && assign_mode != QD_ASSIGN_MIDPOINT) return QD_ERR_INVALID_ARGUMENT;
    if (idx && idx_bytes != 8 && idx_bytes != 1) return QD_ERR_INVALID_ARGUMENT;
    if (idx && idx_bytes == 1 && k > 256) return QD_ERR_INVALID_ARGUMENT;
    if (!q && (!idx || !prescaled)) return QD_ERR_INVALID_ARGUMENT;                 // indices only: the pre-scaled form
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const Prep pp = make_prep(mean, clamp, max_element);
    std::vector<float> mids((size_t)(k > 1 ? k - 1 : 1));
    for (int j = 0; j + 1 < k; ++j) {                                               // fp32 midpoints, :533
        float d = points[j + 1] - points[j];
        d = d / 2.0f;
        mids[(size_t)j] = points[j] + d;
    }
    auto stats = [&](int64_t bk, int64_t lo, int64_t hi, float& a, float& b, bool whole) {
        if (prescaled) { a = alpha[bk]; b = beta[bk]; return; }
        if (whole) {
            whole_minmax(x, n, pp, a, b);
        } else {
            float mn, mx;
            bool nan;
            range_minmax(x, lo, hi, pp, mn, mx, nan);
            alpha_beta(mn, mx, nan, a, b);
        }
        alpha[bk] = a;
        beta[bk] = b;
    };
    auto body = [&](int64_t, int64_t lo, int64_t hi, float a, float b) {
        for (int64_t i = lo; i < hi; ++i) {
            float u;
            if (prescaled) {
                u = x[i];
            } else {
                u = pp(x[i]) - b;
                u = u / a;
            }
            const int j = assign_mode == QD_ASSIGN_DISTANCE ? assign_distance(u, points, k) : assign_midpoint(u, mids.data(), k - 1);
            if (q) {
                float y = points[j] * a;                                            // gather + inv_scale_down, :278,286-287
                y = y + b;
                y = y + pp.mean;
                q[i] = y;
            }
            if (idx) {
                if (idx_bytes == 8) ((int64_t*)idx)[i] = j; else ((uint8_t*)idx)[i] = (uint8_t)j;
            }
        }
    };
    for_buckets(n, nb, row, stats, body);
    return 0;
}

// K6: grad_points[j] = sum_{i : idx_i == j} g_i alpha_bucket(i) (:471-506): the fp32 product of the reference (:495), summed in
// float64 per fixed chunk and then over the chunks in order -- the same result whatever the number of threads
int qd_point_grad_f32(const float* g, const void* idx, int idx_bytes, const float* alpha, int64_t n, int64_t bucket, int k,
                      float* grad_points, void*, size_t, void*) {
    if (n < 0 || bucket < 0 || k < 1 || !grad_points || (n > 0 && (!g || !idx || !alpha))) return QD_ERR_INVALID_ARGUMENT;
    if (idx_bytes != 8 && idx_bytes != 1) return QD_ERR_INVALID_ARGUMENT;
    int64_t nb, row;
    geometry(n > 0 ? n : 1, bucket, nb, row);
    const int64_t chunks = (n + kChunk - 1) / kChunk;
    std::vector<double> part((size_t)(chunks > 0 ? chunks : 1) * (size_t)k, 0.0);
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad) if (chunks > 1) num_threads(threads_for(n))
    for (int64_t c = 0; c < chunks; ++c) {
        const int64_t lo = c * kChunk, hi = lo + kChunk < n ? lo + kChunk : n;
        double* acc = part.data() + (size_t)c * (size_t)k;
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t j = idx_bytes == 8 ? ((const int64_t*)idx)[i] : (int64_t)((const uint8_t*)idx)[i];
            if (j < 0 || j >= k) { bad = 1; continue; }
            const float m = g[i] * alpha[nb == 1 ? 0 : i / row];
            acc[j] += (double)m;
        }
    }
    if (bad) return QD_ERR_INVALID_ARGUMENT;
    for (int j = 0; j < k; ++j) {
        double acc = 0.0;
        for (int64_t c = 0; c < chunks; ++c) acc += part[(size_t)c * (size_t)k + (size_t)j];
        grad_points[j] = (float)acc;
    }
    return 0;
}

// K7: 'complicated' straight-through backward (:319-406, intended math): per bucket S = sum g (qs - u) with qs, u scaled by the
// alpha / beta of the QUANTIZED bucket (:350); out = g, out[jmax] += S, out[jmin] -= S.  Every term in the reference's fp32
// operations, summed in float64, S rounded once.
int qd_ste_bucket_backward_f32(const float* x, const float* g, float* out, int64_t n, int64_t bucket, int levels, int tie_mode,
                               void*) {
    if (n < 0 || bucket <= 0 || levels < 2 || (n > 0 && (!x || !g || !out))) return QD_ERR_INVALID_ARGUMENT;
    if (tie_mode != QD_STE_TIE_REFERENCE && tie_mode != QD_STE_TIE_TRUE_ARG) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const float sm1 = (float)(levels - 1);
    const Prep none = make_prep(nullptr, 0, 0.0f);
#pragma omp parallel if (n >= kParallelMin) num_threads(threads_for(n))
    {
        std::vector<float> qb((size_t)row);
#pragma omp for schedule(static)
        for (int64_t bk = 0; bk < nb; ++bk) {
            const int64_t lo = bk * row, hi = lo + row < n ? lo + row : n;
            float mn, mx, a, b;
            bool nan;
            range_minmax(x, lo, hi, none, mn, mx, nan);
            alpha_beta(mn, mx, nan, a, b);
            if (nan) mn = mx = std::numeric_limits<float>::quiet_NaN();            // torch's min / max of such a bucket
            float qmn = std::numeric_limits<float>::infinity(), qmx = -qmn;
            bool qnan = false;
            for (int64_t i = lo; i < hi; ++i) {
                float lev;
                const float qv = qdq(x[i], a, b, sm1, 0.0f, lev);
                qb[(size_t)(i - lo)] = qv;
                qnan |= (qv != qv);
                qmn = qv < qmn ? qv : qmn;
                qmx = qv > qmx ? qv : qmx;
            }
            float aq, bq;
            alpha_beta(qmn, qmx, qnan, aq, bq);                                     // scale_down of the QUANTIZED bucket, :350
            // min / max of the searched tensor are NaN when it holds one, and torch reports both at its FIRST NaN: the
            // reference then adds and subtracts S = NaN there (:383-400), as the device kernels do
            const bool bad = tie_mode == QD_STE_TIE_REFERENCE ? qnan : nan;
            int64_t jmax = -1, jmin = -1;                                           // FIRST element at the top / bottom (level, or true arg)
            double sb = 0.0;
            for (int64_t i = lo; i < hi; ++i) {
                const float qv = qb[(size_t)(i - lo)];
                const float sv = tie_mode == QD_STE_TIE_REFERENCE ? qv : x[i];
                const bool top = bad ? (sv != sv) : (tie_mode == QD_STE_TIE_REFERENCE ? (qv == qmx) : (x[i] == mx));
                const bool bot = bad ? (sv != sv) : (tie_mode == QD_STE_TIE_REFERENCE ? (qv == qmn) : (x[i] == mn));
                if (top && jmax < 0) jmax = i;
                if (bot && jmin < 0) jmin = i;
                float qs = qv - bq;  qs = qs / aq;
                float u = x[i] - bq;  u = u / aq;                                    // :400
                const float d = qs - u;
                const float t = g[i] * d;
                sb += (double)t;
            }
            const float s = (float)sb;
            const float gmax = jmax >= 0 ? g[jmax] : 0.0f, gmin = jmin >= 0 ? g[jmin] : 0.0f;      // (out may alias g)
            if (out != g) for (int64_t i = lo; i < hi; ++i) out[i] = g[i];
            if (jmax >= 0 && jmin >= 0 && jmax != jmin) {                            // a constant bucket: +S and -S cancel
                out[jmax] = gmax + s;
                out[jmin] = gmin - s;
            } else if (bad && jmax >= 0) {
                out[jmax] = (gmax + s) - s;                                         // NaN
            }
        }
    }
    return 0;
}

// K8: 'truncated' STE (cnn_models/conv_forward_model.py:240-241,263-264)
int qd_clamp_f32(float* w, int64_t n, float limit, void*) {
    if (n < 0 || (n > 0 && !w)) return QD_ERR_INVALID_ARGUMENT;
#pragma omp parallel for schedule(static) if (n >= kParallelMin)
    for (int64_t i = 0; i < n; ++i) {
        float v = w[i];
        v = v > limit ? limit : v;                                                  // (NaN stays NaN, as torch.clamp)
        v = v < -limit ? -limit : v;
        w[i] = v;
    }
    return 0;
}
int qd_truncated_ste_f32(const float* w, float* grad, int64_t n, float limit, void*) {
    if (n < 0 || (n > 0 && (!w || !grad))) return QD_ERR_INVALID_ARGUMENT;
#pragma omp parallel for schedule(static) if (n >= kParallelMin)
    for (int64_t i = 0; i < n; ++i)
        if (std::fabs(w[i]) > limit) grad[i] = 0.0f;
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop
