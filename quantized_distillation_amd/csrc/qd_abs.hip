// qd_abs.hip -- 'absmax' / 'absnorm' scaling (sign + magnitude scaled by the bucket's max|x| or
// L2 norm), the two non-linear ScalingFunction types of the reference.
//
// PARITY UNPINNED: the reference's implementation (quantization/quant_functions.py:109-127,
// 144-146) raises on every torch version (`tensor.max(p=2, ...)` is not a valid call; `:126` stores
// a bound method), so there is no reference output to compare with.  These kernels implement the
// math those lines evidently intend, and are checked against oracle/oracle_np.py's restatement
// of the same intent only:
//     sign = sign(x); m = |x|; norm_b = max_b(m) or sqrt(sum_b m^2); norm < 1e-10 -> 1
//     u = m / norm_b                                   (scale_down)
//     x' = u * norm_b * sign (+ mean)                  (inv_scale_down)
//     q = rint(u*(s-1))/(s-1) * norm_b * sign (+mean)  (uniformQuantization)
// Not a hot path (no driver reaches it): one wave per bucket, two passes, the second from L1/L2;
// a whole-tensor bucket uses a two-stage reduction.
#include "qd_common.h"
#include "../../include/qd_hip.h"

using namespace qd;

namespace {

constexpr int kParts = 1024;

__device__ __forceinline__ float signf(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// reduce |x| over [lo, hi) with a wave: max or sum of squares
template <int NORM>
__device__ __forceinline__ float wave_norm(const float* x, int64_t lo, int64_t hi, int lane, const Prep& pp,
                                           float extra_sq) {
    float acc = 0.0f;
    for (int64_t i = lo + lane; i < hi; i += 64) {
        const float m = fabsf(prep(x[i], pp));
        acc = NORM == 0 ? fmaxf(acc, m) : acc + m * m;
    }
    acc = NORM == 0 ? wave_max(acc) : wave_sum(acc) + extra_sq;
    float nrm = NORM == 0 ? acc : sqrtf(acc);
    return nrm < QD_TOL_DIFF_ZERO ? 1.0f : nrm;
}

// OP 0: quantize-dequantize; 1: scale_down (writes u and sign, padded layout); 
template <int NORM, int OP>
__global__ __launch_bounds__(256) void k_abs_buckets(const float* x, float* out, float* sign_out, float* norm_out,
                                                     const float* norm_in, int norm_stride, int pad, int64_t n,
                                                     int64_t row, int64_t nb, const float* mean, float me, float sm1) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Prep pp;
    pp.mean = mean ? *mean : 0.0f;
    pp.me = me;
    for (int64_t bkt = wave; bkt < nb; bkt += nwaves) {
        const int64_t lo = bkt * row;
        const int64_t hi = lo + row < n ? lo + row : n;
        // the reference pads the ragged last bucket with copies of the last element BEFORE taking the
        // norm (help_functions.py:76-86): copies cannot change a max, but they do enter an L2 norm
        float extra_sq = 0.0f;
        if (NORM == 1 && pad && hi < lo + row && hi == n && nb > 1) {
            const float ml = fabsf(prep(x[n - 1], pp));
            extra_sq = (float)(lo + row - hi) * (ml * ml);
        }
        const float nrm = norm_in ? norm_in[norm_stride ? bkt : 0] : wave_norm<NORM>(x, lo, hi, lane, pp, extra_sq);
        if (lane == 0 && norm_out && !norm_in) norm_out[bkt] = nrm;
        float u_last = 0.0f, s_last = 0.0f;
        for (int64_t i = lo + lane; i < hi; i += 64) {
            const float v = prep(x[i], pp);
            const float sg = signf(v);
            float u = fabsf(v) / nrm;
            if (OP == 0) {
                float t = u * sm1;
                float r = rintf(t);
                float w = r / sm1;
                float y = w * nrm;
                y = y * sg;
                out[i] = y + pp.mean;
            } else {
                out[i] = u;
                sign_out[i] = sg;
            }
        }
        if (OP == 1 && pad && hi < lo + row && hi == n && nb > 1) {           // padding = copies of the last element
            const float v = prep(x[n - 1], pp);
            u_last = fabsf(v) / nrm;
            s_last = signf(v);
            for (int64_t i = hi + lane; i < lo + row; i += 64) { out[i] = u_last; sign_out[i] = s_last; }
        }
    }
}

template <int NORM>
__global__ __launch_bounds__(256) void k_abs_partial(const float* x, int64_t n, const float* mean, float me, float* part) {
    __shared__ float red[4];
    Prep pp;
    pp.mean = mean ? *mean : 0.0f;
    pp.me = me;
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = fabsf(prep(x[i], pp));
        acc = NORM == 0 ? fmaxf(acc, m) : acc + m * m;
    }
    acc = NORM == 0 ? wave_max(acc) : wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        part[blockIdx.x] = NORM == 0 ? fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
}
template <int NORM>
__global__ __launch_bounds__(256) void k_abs_final(const float* part, int nparts, float* norm_out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc = NORM == 0 ? fmax(acc, (double)part[i]) : acc + (double)part[i];
    if (NORM == 0) { for (int s = 1; s < 64; s <<= 1) acc = fmax(acc, __shfl_xor(acc, s)); }
    else acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double t = NORM == 0 ? fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) : (red[0] + red[1]) + (red[2] + red[3]);
        float nrm = NORM == 0 ? (float)t : sqrtf((float)t);
        norm_out[0] = nrm < QD_TOL_DIFF_ZERO ? 1.0f : nrm;
    }
}

__global__ __launch_bounds__(256) void k_abs_inverse(const float* u, const float* sign, float* y, int64_t n, int64_t row,
                                                     int64_t nb, const float* norm, const float* mean) {
    const float m = mean ? *mean : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = u[i] * norm[nb == 1 ? 0 : i / row];
        v = v * sign[i];
        y[i] = v + m;
    }
}

inline void geometry(int64_t n, int64_t bucket, int64_t& nb, int64_t& row) {
    if (bucket <= 0 || n < bucket) { nb = 1; row = n; return; }
    row = bucket;
    nb = (n + bucket - 1) / bucket;
}
inline int nblocks(int64_t items, int per) {
    int64_t b = (items + per - 1) / per;
    return (int)(b < 1 ? 1 : (b > 65536 ? 65536 : b));
}

template <int OP>
int run_abs(const float* x, float* out, float* sign_out, float* norm_out, int64_t n, int64_t bucket, int norm_kind,
            const float* mean, int clamp, float max_element, float sm1, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const float me = clamp ? max_element : INFINITY;
    const float* norm_in = nullptr;
    if (nb == 1 && n > 65536) {                        // whole-tensor bucket: two-stage norm first
        if (!ws || ws_bytes < kParts * sizeof(float)) return QD_ERR_WORKSPACE_TOO_SMALL;
        float* part = (float*)ws;
        const int pb = nblocks(n, 256 * 32) > kParts ? kParts : nblocks(n, 256 * 32);
        if (norm_kind == 0) {
            hipLaunchKernelGGL(k_abs_partial<0>, dim3(pb), dim3(256), 0, st, x, n, mean, me, part);
            hipLaunchKernelGGL(k_abs_final<0>, dim3(1), dim3(256), 0, st, part, pb, norm_out);
        } else {
            hipLaunchKernelGGL(k_abs_partial<1>, dim3(pb), dim3(256), 0, st, x, n, mean, me, part);
            hipLaunchKernelGGL(k_abs_final<1>, dim3(1), dim3(256), 0, st, part, pb, norm_out);
        }
        norm_in = norm_out;
        // apply with many waves: treat the tensor as chunks that all use norm_in[0]
        const int64_t chunk = 4096;
        const int64_t nchunks = (n + chunk - 1) / chunk;
        if (norm_kind == 0)
            hipLaunchKernelGGL((k_abs_buckets<0, OP>), dim3(nblocks(nchunks, 4)), dim3(256), 0, st, x, out, sign_out, nullptr,
                               norm_in, 0, 0, n, chunk, nchunks, mean, me, sm1);
        else
            hipLaunchKernelGGL((k_abs_buckets<1, OP>), dim3(nblocks(nchunks, 4)), dim3(256), 0, st, x, out, sign_out, nullptr,
                               norm_in, 0, 0, n, chunk, nchunks, mean, me, sm1);
        return (int)hipGetLastError();
    }
    if (norm_kind == 0)
        hipLaunchKernelGGL((k_abs_buckets<0, OP>), dim3(nblocks(nb, 4)), dim3(256), 0, st, x, out, sign_out, norm_out, nullptr,
                           1, 1, n, row, nb, mean, me, sm1);
    else
        hipLaunchKernelGGL((k_abs_buckets<1, OP>), dim3(nblocks(nb, 4)), dim3(256), 0, st, x, out, sign_out, norm_out, nullptr,
                           1, 1, n, row, nb, mean, me, sm1);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" {

int qd_uniform_abs_f32(const float* x, float* q, int64_t n, int64_t bucket, int levels, int norm_kind, float* norm_out,
                       const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (n < 0 || bucket < 0 || levels < 2 || (norm_kind != 0 && norm_kind != 1) || (n > 0 && (!x || !q || !norm_out)))
        return QD_ERR_INVALID_ARGUMENT;
    return run_abs<0>(x, q, nullptr, norm_out, n, bucket, norm_kind, mean, clamp, max_element, (float)(levels - 1),
                      workspace, workspace_bytes, (hipStream_t)stream);
}

int qd_scale_down_abs_f32(const float* x, float* u, float* sign, int64_t n, int64_t bucket, int norm_kind,
                          float* norm_out, const float* mean, int clamp, float max_element, void* workspace,
                          size_t workspace_bytes, void* stream) {
    if (n < 0 || bucket < 0 || (norm_kind != 0 && norm_kind != 1) || (n > 0 && (!x || !u || !sign || !norm_out)))
        return QD_ERR_INVALID_ARGUMENT;
    return run_abs<1>(x, u, sign, norm_out, n, bucket, norm_kind, mean, clamp, max_element, 1.0f, workspace,
                      workspace_bytes, (hipStream_t)stream);
}

int qd_inv_scale_abs_f32(const float* u, const float* sign, float* y, int64_t n, int64_t bucket, const float* norm,
                         const float* mean, void* stream) {
    if (n < 0 || bucket < 0 || (n > 0 && (!u || !sign || !y || !norm))) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    hipLaunchKernelGGL(k_abs_inverse, dim3(nblocks(n, 256 * 8)), dim3(256), 0, (hipStream_t)stream, u, sign, y, n, row, nb,
                       norm, mean);
    return (int)hipGetLastError();
}

}  // extern "C"
