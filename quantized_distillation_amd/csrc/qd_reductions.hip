// qd_reductions.hip -- the kernels of the path that are not instantiations of the bucket-transform template (qd_transform.h):
//   qd_mean_f32                  tensor.mean() of subtract_mean=True          quantization/quant_functions.py:67
//   K3  qd_inv_scale_f32         ScalingFunction.inv_scale_down               :131-152
//       qd_bucket_argminmax_f32  idx_min_rows / idx_max_rows                  :85-90,103-104
//   K6  qd_point_grad_f32        nonUniformQuantization_variable.backward     :471-506
//   K7  qd_ste_bucket_backward_f32  uniformQuantization_variable.backward     :319-406
//   K8  qd_clamp_f32, qd_truncated_ste_f32   the callers' 'truncated' STE     cnn_models/conv_forward_model.py:240-241,263-264
//   K9  qd_multi_plan, qd_multi_uniform_f32  the per-parameter loop as one launch   conv_forward_model.py:235-247
// Its own translation unit so that hipcc builds it next to qd_kernels.hip / qd_nearest.hip / qd_scale.hip (parallel build).
#include "qd_transform.h"      // shared helpers: workspace carving, launch geometry, the per-bucket device routines K9 reuses

namespace {

// ---- mean (float64 accumulation, fixed order) -------------------------------------------------
__global__ __launch_bounds__(256) void k_sum_partial(const float* x, int64_t n, double* part) {
    __shared__ double red[4];
    double acc = 0.0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)x) & kDataAlign) == 0) {
        const int64_t n4 = n >> 2;
        const f4* x4 = (const f4*)x;
        for (int64_t i = tid; i < n4; i += nth) {
            const f4 v = x4[i];
            acc += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth) acc += (double)x[i];
    } else {
        for (int64_t i = tid; i < n; i += nth) acc += (double)x[i];
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ __launch_bounds__(256) void k_mean_final(const double* part, int nparts, int64_t n, float* mean_out) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) mean_out[0] = (float)(((red[0] + red[1]) + (red[2] + red[3])) / (double)n);
}

// ---- inverse scaling (quant_functions.py:131-152) --------------------------------------------
// y = u * alpha_b + beta_b (+ mean), three separately rounded operations, a plain stream at ANY bucket size from 4 elements
// up: every lane takes float4s in memory order (kInvU in flight); the float4's first element lies in bucket b0 = e / row --
// one division per lane, then advanced incrementally -- and the others in b0 or b0 + 1, so two (alpha, beta) pairs per
// float4, served by L1 / L2 as neighbouring lanes ask for the same buckets.  (Round 2 divided per float4 and sent every
// bucket size that is not a multiple of 4 down the scalar tail loop.)
// TWO = false: the bucket size is a multiple of 4 (or there is one bucket), a float4 never straddles two buckets: one pair.
constexpr int kInvU = 4;
template <bool TWO>
__global__ __launch_bounds__(256) void k_inv_scale(const float* u, float* y, int64_t n, int64_t row, int64_t nb,
                                                   const float* alpha, const float* beta, const float* mean) {
    const float m = mean ? *mean : 0.0f;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const bool vec = (((((uintptr_t)u) | ((uintptr_t)y)) & kDataAlign) == 0) && (nb == 1 || row >= 4);
    int64_t done = 0;
    if (vec) {
        // one CONTIGUOUS 4 KiB tile per wave and step (lane l holds float4 tile * 256 + 64 q + l), as the vector kernels
        // stream: four far-apart streams per lane (float4 i, i + nth, ...) measured 6 % slower -- 90.6 us against 85.5 us
        const int64_t n4 = n >> 2, last_b = nb - 1;
        const int lane = threadIdx.x & 63;
        const int64_t wave = uniform_wave_index(), nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
        const int64_t ntiles = (n4 + 255) >> 8;
        const int64_t span = nb == 1 ? ((int64_t)1 << 62) : row;
        const int64_t dq = nb == 1 ? 0 : 256 / row, dr = nb == 1 ? 0 : 256 % row;      // 64 float4 = 256 elements further
        const bool small = n < ((int64_t)1 << 31);
        for (int64_t t = wave; t < ntiles; t += nwaves) {
            const int64_t base = (t << 8) + lane;                       // this lane's first float4 of the tile
            int64_t bkt = 0, rem = 0;
            if (nb > 1) {
                if (small) { const uint32_t e = (uint32_t)(base << 2), rw = (uint32_t)row; bkt = e / rw; rem = e - (uint32_t)bkt * rw; }
                else { bkt = (base << 2) / row; rem = (base << 2) - bkt * row; }
            }
            f4 v[kInvU];
            float a0[kInvU], a1[kInvU], b0[kInvU], b1[kInvU];
            int split[kInvU];
#pragma unroll
            for (int q = 0; q < kInvU; ++q) {
                const int64_t ii = base + 64 * q;
                v[q] = __builtin_nontemporal_load((const f4*)u + (ii < n4 ? ii : n4 - 1));       // always issued, clamped
                const int64_t bb = bkt < last_b ? bkt : last_b, bn = bkt + 1 < last_b ? bkt + 1 : last_b;
                a0[q] = alpha[bb]; b0[q] = beta[bb];
                a1[q] = TWO ? alpha[bn] : a0[q]; b1[q] = TWO ? beta[bn] : b0[q];
                const int64_t left = span - rem;
                split[q] = (!TWO || left >= 4) ? 4 : (int)left;
                bkt += dq; rem += dr;
                if (rem >= span) { rem -= span; ++bkt; }
            }
#pragma unroll
            for (int q = 0; q < kInvU; ++q) {
                const int64_t ii = base + 64 * q;
                if (ii < n4) {
                    const float x[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool first = c < split[q];
                        float r = x[c] * (first ? a0[q] : a1[q]);
                        r = r + (first ? b0[q] : b1[q]);
                        r = r + m;
                        o[c] = r;
                    }
                    const f4 rr = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(rr, (f4*)y + ii);
                }
            }
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += nth) {
        const int64_t bkt = nb == 1 ? 0 : i / row;
        float r = u[i] * alpha[bkt];
        r = r + beta[bkt];
        r = r + m;
        y[i] = r;
    }
}

// ---- first-occurrence arg-min/arg-max per bucket (quant_functions.py:85-90) -------------------
// key = (value, index) compared lexicographically; block per bucket; any size (single bucket of
// a huge tensor is handled by a grid-stride over `chunks` partial blocks + a final fold).
struct ArgPair { float v; int64_t i; };
__device__ __forceinline__ void arg_fold_min(float& v, int64_t& i, float ov, int64_t oi) {
    if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void arg_fold_max(float& v, int64_t& i, float ov, int64_t oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__device__ __forceinline__ void block_argminmax(float& mnv, int64_t& mni, float& mxv, int64_t& mxi, float* sv,
                                                int64_t* si) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        arg_fold_min(mnv, mni, __shfl_xor(mnv, s), __shfl_xor((long long)mni, s));
        arg_fold_max(mxv, mxi, __shfl_xor(mxv, s), __shfl_xor((long long)mxi, s));
    }
    const int nw = blockDim.x >> 6, w = threadIdx.x >> 6;
    if (nw > 1) {
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { sv[w] = mnv; si[w] = mni; sv[16 + w] = mxv; si[16 + w] = mxi; }
        __syncthreads();
        mnv = sv[0]; mni = si[0]; mxv = sv[16]; mxi = si[16];
        for (int j = 1; j < nw; ++j) {
            arg_fold_min(mnv, mni, sv[j], si[j]);
            arg_fold_max(mxv, mxi, sv[16 + j], si[16 + j]);
        }
    }
}
// grid = nb * chunks blocks; chunk c of bucket b scans its slice; chunks == 1 writes the result
// directly, otherwise partial pairs go to `pv/pi` ([2*chunks] each) for k_arg_final.
__global__ __launch_bounds__(256) void k_argminmax(const float* x, int64_t n, int64_t row, int64_t nb, int chunks,
                                                   const float* mean, float me, int64_t* argmin, int64_t* argmax,
                                                   float* pv, int64_t* pi) {
    __shared__ float sv[32];
    __shared__ int64_t si[32];
    Prep pp;
    pp.mean = mean ? *mean : 0.0f;
    pp.me = me;
    for (int64_t w = blockIdx.x; w < nb * chunks; w += gridDim.x) {
        const int64_t bkt = w / chunks;
        const int c = (int)(w % chunks);
        const int64_t lo = bkt * row;
        const int64_t hi = lo + row < n ? lo + row : n;
        const int64_t len = hi - lo;
        const int64_t per = (len + chunks - 1) / chunks;
        const int64_t clo = lo + c * per;
        const int64_t chi = clo + per < hi ? clo + per : hi;
        float mnv = INFINITY, mxv = -INFINITY;
        int64_t mni = INT64_MAX, mxi = INT64_MAX;
        for (int64_t i = clo + threadIdx.x; i < chi; i += blockDim.x) {
            const float v = prep(x[i], pp);
            if (mni == INT64_MAX || v < mnv) { mnv = v; mni = i - lo; }   // strict: first occurrence wins
            if (mxi == INT64_MAX || v > mxv) { mxv = v; mxi = i - lo; }
        }
        block_argminmax(mnv, mni, mxv, mxi, sv, si);
        if (threadIdx.x == 0) {
            if (chunks == 1) { argmin[bkt] = mni; argmax[bkt] = mxi; }
            else { pv[c] = mnv; pi[c] = mni; pv[chunks + c] = mxv; pi[chunks + c] = mxi; }
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_arg_final(const float* pv, const int64_t* pi, int chunks, int64_t* argmin,
                                                   int64_t* argmax) {
    __shared__ float sv[32];
    __shared__ int64_t si[32];
    float mnv = INFINITY, mxv = -INFINITY;
    int64_t mni = INT64_MAX, mxi = INT64_MAX;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
        if (pi[c] != INT64_MAX) arg_fold_min(mnv, mni, pv[c], pi[c]);
        if (pi[chunks + c] != INT64_MAX) arg_fold_max(mxv, mxi, pv[chunks + c], pi[chunks + c]);
    }
    block_argminmax(mnv, mni, mxv, mxi, sv, si);
    if (threadIdx.x == 0) { argmin[0] = mni; argmax[0] = mxi; }
}

// ---- K6: point gradient (quant_functions.py:493-503) ------------------------------------------
//     grad_points[j] = sum_{i: idx_i == j} g_i * alpha_bucket(i)
// stage 1 (fast path): 16-byte g loads + packed 4-byte (uint8 x4) or 2 x 16-byte (int64 x4) index
// loads, two float4 in flight per lane, bucket = element >> shift (power-of-two buckets) or 0.
//   KR > 0 : k <= KR bins in registers, unrolled select-accumulate (k <= 4);
//   KR == 0: bins[k][256] in LDS, every lane owns a private column and does a plain
//            read-add-write (no atomics: LDS float atomics retire ~1 lane per 3 cycles on gfx950,
//            measured 390 us for the 64 Mi-element tensor); k <= 64.
// Both are deterministic: fixed per-lane order, fixed fold order.
// s[j] = the in-order sum of the elements i <= j of a float4 that share element j's index.  The four bins are read before
// any is written and the LDS writes of a wave complete in program order, so the LAST element of a group of duplicates
// -- whose s is the group's whole sum, added in element order -- is the one that stays: 6 compares, 6 selects, 6 adds
// (giving every duplicate the full sum, as round 2 did, costs 12 + 12).
__device__ __forceinline__ void merged_prefix_sums(const int (&id)[4], const float (&m)[4], float (&s)[4]) {
    const bool e01 = id[0] == id[1], e02 = id[0] == id[2], e03 = id[0] == id[3];
    const bool e12 = id[1] == id[2], e13 = id[1] == id[3], e23 = id[2] == id[3];
    const float z = 0.0f;
    s[0] = m[0];
    s[1] = (e01 ? m[0] : z) + m[1];
    s[2] = ((e02 ? m[0] : z) + (e12 ? m[1] : z)) + m[2];
    s[3] = (((e03 ? m[0] : z) + (e13 ? m[1] : z)) + (e23 ? m[2] : z)) + m[3];
}

// HALF: the two half-waves share 32 columns (lane l and lane l + 32 of a wave: column l) and update them one after the other --
// LDS operations of one wave complete in order, the wave barriers only pin the compiler -- so a [k][128] table serves a
// 256-lane block: 64 KiB at k = 128, two blocks per CU, every wave on its own columns (no token to pass).
template <int KR, bool HALF = false>
struct PgBins {
    float acc[KR > 0 ? KR : 1];
    float* col;
    int stride;
    int hi;                                                 // HALF: which half-wave this lane is in
    __device__ __forceinline__ void init(float* lds_col, int stride_, int hi_ = 0) {
        col = lds_col;
        stride = stride_;
        hi = hi_;
#pragma unroll
        for (int j = 0; j < (KR > 0 ? KR : 1); ++j) acc[j] = 0.0f;
    }
    __device__ __forceinline__ void add(int id, float m) {
        if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) acc[j] += (id == j) ? m : 0.0f;
        } else if (!HALF) {
            col[id * stride] += m;                          // private column: plain LDS read-add-write
        } else {
            if (hi == 0) col[id * stride] += m;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (hi == 1) col[id * stride] += m;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // four elements at once, for the large tables that leave one or two waves per CU: the four reads are
    // independent (one LDS round trip instead of four dependent ones); a lane's duplicates are resolved in
    // registers (merged_prefix_sums below) and the four writes go out in element order.
    __device__ __forceinline__ void add4_merged(const int (&id)[4], const float (&m)[4]) {
        float* a0 = col + id[0] * stride; float* a1 = col + id[1] * stride;
        float* a2 = col + id[2] * stride; float* a3 = col + id[3] * stride;
        float s[4];
        merged_prefix_sums(id, m, s);
        if (!HALF) {
            const float c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
            *a0 = c0 + s[0]; *a1 = c1 + s[1]; *a2 = c2 + s[2]; *a3 = c3 + s[3];
        } else {
            if (hi == 0) {
                const float c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
                *a0 = c0 + s[0]; *a1 = c1 + s[1]; *a2 = c2 + s[2]; *a3 = c3 + s[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (hi == 1) {
                const float c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
                *a0 = c0 + s[0]; *a1 = c1 + s[1]; *a2 = c2 + s[2]; *a3 = c3 + s[3];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
};

// The buckets of the float4s a lane visits -- tid, tid + nth, tid + 2 nth, ... in that order -- at ANY bucket size from 4
// elements up: one division per lane at the start, then (bucket, offset) advanced by 4 nth elements per step.  A float4
// touches at most two buckets: the alphas of both and the number of its elements that lie in the first one.
// 32-bit state: the host takes this path only with fewer than 2^30 buckets of fewer than 2^30 elements.
struct BucketWalk {
    int bkt, rem, dq, dr, row, last;
    __device__ __forceinline__ void init(int64_t tid, int64_t nth, int64_t row_, int64_t nb) {
        row = (int)row_; last = (int)(nb - 1);
        bkt = (int)((tid << 2) / row_); rem = (int)((tid << 2) % row_);
        dq = (int)((nth << 2) / row_); dr = (int)((nth << 2) % row_);
    }
    __device__ __forceinline__ void next(const float* alpha, float& a0, float& a1, int& split) {
        const int bb = bkt < last ? bkt : last, bn = bkt + 1 < last ? bkt + 1 : last;     // (clamped loads past the end are dead)
        a0 = alpha[bb]; a1 = alpha[bn];
        const int left = row - rem;
        split = left < 4 ? left : 4;
        bkt += dq; rem += dr;
        if (rem >= row) { rem -= row; ++bkt; }
    }
};

// BK: 0 = one alpha for the tensor, 1 = bucket = element >> row_shift (power-of-two buckets), 2 = any bucket size >= 4
// (BucketWalk: non-power-of-two sizes ran on the scalar kernel below before: 82-126 us against 54-58 us)
template <int KR, int IDXB, int BK, int U, bool MERGE, bool HALF = false>
__global__ __launch_bounds__(256) void k_point_grad_fast(const float* g, const void* idx, const float* alpha, int64_t n,
                                                         int row_shift, int64_t row, int64_t nb, int k, float* part /* [grid][k] */) {
    // KR == 0: bins[k][BS] with BS = blockDim.x (256 for k <= 128, 128 for k <= 256, 64 for k <= 512:
    // the table is at most 128 KiB of the CU's 160 KiB LDS); KR > 0: [4][KR]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int BS = blockDim.x;                              // 64 / 128 / 256
    const unsigned tx = threadIdx.x;
    const int64_t bbase = (int64_t)blockIdx.x * BS;         // block-uniform: lane 0's first float4
    const int64_t tid = bbase + tx;
    const int64_t nth = (int64_t)gridDim.x * BS;
    const int C = HALF ? BS >> 1 : BS;                      // columns of the table
    if (KR == 0) {
        for (int j = threadIdx.x; j < k * C; j += BS) lds[j] = 0.0f;
        __syncthreads();
    }
    PgBins<KR, HALF> B;
    if (HALF) B.init(lds + ((tx >> 6) << 5) + (tx & 31), C, (tx >> 5) & 1);
    else B.init(lds + threadIdx.x, BS);
    const float a_single = BK != 0 ? 0.0f : alpha[0];
    const int64_t n4 = n >> 2;
    BucketWalk walk;
    if (BK == 2) walk.init(tid, nth, row, nb);
    struct A2 { float a0, a1; int split; };
    auto load4 = [&](int64_t i, f4& gv, int (&id)[4], A2& a) {
        gv = __builtin_nontemporal_load((const f4*)g + i);
        if (IDXB == 8) {
            const l2 p0 = __builtin_nontemporal_load((const l2*)idx + 2 * i);
            const l2 p1 = __builtin_nontemporal_load((const l2*)idx + 2 * i + 1);
            id[0] = (int)p0.x; id[1] = (int)p0.y; id[2] = (int)p1.x; id[3] = (int)p1.y;
        } else {
            const uint32_t pk = __builtin_nontemporal_load((const uint32_t*)idx + i);
            id[0] = pk & 255; id[1] = (pk >> 8) & 255; id[2] = (pk >> 16) & 255; id[3] = pk >> 24;
        }
        if (BK == 2) {
            walk.next(alpha, a.a0, a.a1, a.split);          // (load4 is called for i = tid, tid + nth, ... in this order)
        } else {
            a.a0 = BK == 1 ? alpha[(i << 2) >> row_shift] : a_single;
            a.a1 = a.a0; a.split = 4;
        }
    };
    // the same for the float4 of lane tx of a block whose lane 0 is at the block-uniform index ub (a multiple of BS): scalar
    // base addresses + one loop-invariant lane offset each, no 64-bit vector arithmetic.  Power-of-two buckets: BS divides
    // ub and tx < BS, so bucket(ub + tx) = (ub >> sh) + (tx >> sh) for every shift.
    const int sh = BK == 1 ? row_shift - 2 : 0;
    const unsigned txs = sh < 31 ? tx >> sh : 0u;
    auto load4u = [&](int64_t ub, f4& gv, int (&id)[4], A2& a) {
        gv = __builtin_nontemporal_load((const f4*)g + ub + tx);
        if (IDXB == 8) {
            const l2* ip = (const l2*)idx + 2 * ub;
            const l2 p0 = __builtin_nontemporal_load(ip + 2 * tx);
            const l2 p1 = __builtin_nontemporal_load(ip + 2 * tx + 1);
            id[0] = (int)p0.x; id[1] = (int)p0.y; id[2] = (int)p1.x; id[3] = (int)p1.y;
        } else {
            const uint32_t pk = __builtin_nontemporal_load((const uint32_t*)idx + ub + tx);
            id[0] = pk & 255; id[1] = (pk >> 8) & 255; id[2] = (pk >> 16) & 255; id[3] = pk >> 24;
        }
        if (BK == 2) {
            walk.next(alpha, a.a0, a.a1, a.split);
        } else {
            a.a0 = BK == 1 ? (alpha + (ub >> sh))[txs] : a_single;
            a.a1 = a.a0; a.split = 4;
        }
    };
    auto accumulate = [&](const f4& gv, const int (&id)[4], const A2& a) {
        const float m[4] = {gv.x * (BK == 2 && a.split < 1 ? a.a1 : a.a0), gv.y * (BK == 2 && a.split < 2 ? a.a1 : a.a0),
                            gv.z * (BK == 2 && a.split < 3 ? a.a1 : a.a0), gv.w * (BK == 2 && a.split < 4 ? a.a1 : a.a0)};   // one fp32 multiply each, :495
        if (KR == 0 && MERGE) {
            B.add4_merged(id, m);
        } else {
            B.add(id[0], m[0]);
            B.add(id[1], m[1]);
            B.add(id[2], m[2]);
            B.add(id[3], m[3]);
        }
    };
    int64_t ub = bbase;
    for (; ub + (int64_t)(U - 1) * nth + BS <= n4; ub += (int64_t)U * nth) {   // U independent float4 in flight per lane
        f4 gv[U]; int id[U][4]; A2 a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) load4u(ub + (int64_t)u * nth, gv[u], id[u], a[u]);
        // without this the scheduler sinks every load to its first use (to save registers) and the loop pays
        // one full memory round trip per float4
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) accumulate(gv[u], id[u], a[u]);
    }
    for (int64_t i = ub + tx; i < n4; i += nth) {           // the last (up to U) float4 of a lane: per-lane addresses
        f4 ga; int ia[4]; A2 sa;
        load4(i, ga, ia, sa);
        accumulate(ga, ia, sa);
    }
    for (int64_t e = (n4 << 2) + tid; e < n; e += nth) {     // n % 4 leftover elements
        const int id = IDXB == 8 ? (int)((const int64_t*)idx)[e] : (int)((const uint8_t*)idx)[e];
        B.add(id, g[e] * (BK == 2 ? alpha[e / row] : (BK == 1 ? alpha[e >> row_shift] : a_single)));
    }
    if (KR > 0) {
        const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int j = 0; j < (KR > 0 ? KR : 1); ++j) {
            const float sum = wave_sum(B.acc[j]);
            if (lane == 0) lds[w * KR + j] = sum;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < k; j += BS)
            part[(int64_t)blockIdx.x * k + j] = (lds[j] + lds[KR + j]) + (lds[2 * KR + j] + lds[3 * KR + j]);
    } else {
        __syncthreads();
        // 4 threads per bin, C/4 columns each, rotated start (bank-conflict free), then a fixed fold
        const int quarter = C >> 2;
        for (int t = threadIdx.x; t < ((k * 4 + 3) & ~3); t += BS) {
            const int j = t >> 2, q = t & 3;
            float acc = 0.0f;
            if (j < k)
                for (int c = 0; c < quarter; ++c) acc += lds[j * C + q * quarter + ((c + j) & (quarter - 1))];
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            if (q == 0 && j < k) part[(int64_t)blockIdx.x * k + j] = acc;
        }
    }
}

// stage 1 for k > 128 points (64 < k <= 128: the half-wave columns of PgBins): G waves take TURNS on one set of 64 lane-private columns.
// With one lane per column a [k][256 / 128 / 64] table (k <= 128 / 256 / 512) fills the LDS and leaves 4 / 2 / 1 waves per
// CU, each paying the LDS round trip of every float4 (read 4 bins, add, write) AND its ~40 VALU instructions back to
// back: 66 / 121 / 305 us for 64 Mi elements.  (An intermediate form -- two thread groups sharing the columns and
// alternating table and preparation phases between block barriers -- reached 104 / 234 us at k = 256 / 512: the phases
// are as long as the HBM latency of a batch.)
// What serialises is only the table phase of a batch (U round trips of read-4-bins / add / write, ~1 k cycles); the rest --
// HBM latency of the batch's loads (~5 k cycles), products and duplicate merging (~1.3 k) -- is private to a wave.  So G
// single-wave groups share a [k][64] table (64 KiB at k = 256: two blocks per CU, eight waves instead of two) and pass a
// (ds_add_f32 on the lane's column instead of read / add / write -- in order within a wave, so still deterministic under the
// token, and no duplicate merging -- measured 345 us at k = 128 / 256 against 62 us: LDS float atomics are slow.)
// turn token round robin through LDS: a wave issues the loads of batch i+2, waits for its turn, updates the table with
// batch i, hands the token on, and prepares batch i+1 while the other waves take their turns.  No block barriers in the
// loop, no atomics on the table, and deterministic: the update order is fixed (wave 0, 1, ..., G-1, batch by batch).
template <int IDXB, int BK, int U, int G>
__global__ __launch_bounds__(64 * G) void k_point_grad_turns(const float* g, const void* idx, const float* alpha, int64_t n,
                                                            int row_shift, int64_t row, int64_t nb, int k, float* part /* [grid][k] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];       // [k][64]
    __shared__ int turn;
    constexpr int C = 64;
    constexpr int BS = 64 * G;
    const unsigned tx = threadIdx.x;
    const int grp = tx >> 6;
    float* col = lds + (tx & 63);
    for (int j = tx; j < k * C; j += BS) lds[j] = 0.0f;
    if (tx == 0) turn = 0;
    const float a_single = BK != 0 ? 0.0f : alpha[0];
    const int64_t n4 = n >> 2;
    const int64_t bbase = (int64_t)blockIdx.x * BS;         // block-uniform
    const int64_t tid = bbase + tx;
    const int64_t nth = (int64_t)gridDim.x * BS;
    const int64_t iters = (n4 + (int64_t)U * nth - 1) / ((int64_t)U * nth);   // the same for every wave: the token must circulate
    // (Block-uniform base addresses and no clamps / dead-lane selects for the batches that lie inside the tensor -- 40 instead of
    // 79 vector instructions per float4 -- measured 64.2 / 66.0 us at k = 128 / 256 against 62.2 / 64.9 with the general form
    // below on the same box; two rings of two waves on [k][128] for k <= 128: 64.2 against 62.0.  Neither the instruction
    // count nor the length of the token ring is what the rounds wait for; what does show is the hand-off: s_sleep 16 in the
    // polling loop costs 10 us, s_sleep 4 gains 1.2 us over s_sleep 1 -- three waiting waves poll the LDS less often.)
    BucketWalk walk;
    if (BK == 2) walk.init(tid, nth, row, nb);
    int id[U][4];
    float sm[U][4];
    // (TWO raw batches in flight per wave instead of one -- 20 KiB under way per wave, 229 VGPRs -- measured 66.9 / 69.9 us at
    // k = 128 / 256 against 61.5 / 63.4: more bytes under way do not shorten the rounds.)
    struct Raw { f4 gv[U]; uint32_t pk[U]; l2 p0[U], p1[U]; float al[U], al1[U]; int spl[U]; };
    Raw r;                                                  // the raw batch in flight
    auto issue = [&](int64_t it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i_raw = tid + ((int64_t)it * U + u) * nth;
            const int64_t i = i_raw < n4 ? i_raw : (n4 > 0 ? n4 - 1 : 0);  // always-issued loads, clamped address
            r.gv[u] = __builtin_nontemporal_load((const f4*)g + i);
            if (IDXB == 8) {
                r.p0[u] = __builtin_nontemporal_load((const l2*)idx + 2 * i);
                r.p1[u] = __builtin_nontemporal_load((const l2*)idx + 2 * i + 1);
            } else {
                r.pk[u] = __builtin_nontemporal_load((const uint32_t*)idx + i);
            }
            if (BK == 2) {
                walk.next(alpha, r.al[u], r.al1[u], r.spl[u]);
            } else {
                r.al[u] = BK == 1 ? alpha[(i << 2) >> row_shift] : a_single;
                r.al1[u] = r.al[u]; r.spl[u] = 4;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                  // issued here, not sunk to the first use
    };
    auto compute = [&](int64_t it) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t i_raw = tid + ((int64_t)it * U + u) * nth;
            const bool live = i_raw < n4;
            if (IDXB == 8) {
                id[u][0] = (int)r.p0[u].x; id[u][1] = (int)r.p0[u].y; id[u][2] = (int)r.p1[u].x; id[u][3] = (int)r.p1[u].y;
            } else {
                id[u][0] = r.pk[u] & 255; id[u][1] = (r.pk[u] >> 8) & 255; id[u][2] = (r.pk[u] >> 16) & 255; id[u][3] = r.pk[u] >> 24;
            }
            const float z = 0.0f;
            const float m0 = live ? r.gv[u].x * (BK == 2 && r.spl[u] < 1 ? r.al1[u] : r.al[u]) : z;   // one fp32 multiply each, :495
            const float m1 = live ? r.gv[u].y * (BK == 2 && r.spl[u] < 2 ? r.al1[u] : r.al[u]) : z;
            const float m2 = live ? r.gv[u].z * (BK == 2 && r.spl[u] < 3 ? r.al1[u] : r.al[u]) : z;
            const float m3 = live ? r.gv[u].w * (BK == 2 && r.spl[u] < 4 ? r.al1[u] : r.al[u]) : z;
            const float m[4] = {m0, m1, m2, m3};
            merged_prefix_sums(id[u], m, sm[u]);           // the last of a lane's duplicates carries their whole sum
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto update = [&]() {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float* a0 = col + id[u][0] * C; float* a1 = col + id[u][1] * C;
            float* a2 = col + id[u][2] * C; float* a3 = col + id[u][3] * C;
            const float c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
            *a0 = c0 + sm[u][0]; *a1 = c1 + sm[u][1]; *a2 = c2 + sm[u][2]; *a3 = c3 + sm[u][3];   // in element order (see above)
        }
    };
    if (iters > 0) {
        issue(0);
        compute(0);
        if (iters > 1) issue(1);
    }
    __syncthreads();                                        // table zeroed, token at wave 0
    for (int64_t it = 0; it < iters; ++it) {
        while (__hip_atomic_load(&turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != grp) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        update();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // this wave's table writes are done before the token moves
        if ((tx & 63) == 0)
            __hip_atomic_store(&turn, grp + 1 == G ? 0 : grp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (it + 1 < iters) {
            compute(it + 1);
            if (it + 2 < iters) issue(it + 2);
        }
    }
    __syncthreads();
    // n % 4 leftover elements: wave 0's first lanes of block 0, after the last table phase
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) {
        const int64_t e = (n4 << 2) + threadIdx.x;
        const int ide = IDXB == 8 ? (int)((const int64_t*)idx)[e] : (int)((const uint8_t*)idx)[e];
        col[ide * C] += g[e] * (BK == 2 ? alpha[e / row] : (BK == 1 ? alpha[e >> row_shift] : a_single));
    }
    __syncthreads();
    // 4 threads per bin, C / 4 columns each, rotated start (bank-conflict free), then a fixed fold
    constexpr int quarter = C >> 2;
    for (int t = threadIdx.x; t < ((k * 4 + 3) & ~3); t += blockDim.x) {
        const int j = t >> 2, q = t & 3;
        float acc = 0.0f;
        if (j < k)
            for (int c = 0; c < quarter; ++c) acc += lds[j * C + q * quarter + ((c + j) & (quarter - 1))];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (q == 0 && j < k) part[(int64_t)blockIdx.x * k + j] = acc;
    }
}

// stage 1 (any alignment, any bucket size, k <= 1024): scalar accesses, the bucket of an element advanced
// incrementally (no division in the loop), and the same lane-private columns as the fast path -- bins[k][C] in LDS,
// plain read-add-write, NO atomics (round 2 used ds_add_f32 here, whose order is not fixed: the result changed from run
// to run).  C = 256 columns for 256-lane blocks while the table fits 64 KiB (k <= 64); above that one WAVE per block
// with 64 columns (k <= 512, up to 128 KiB) or 32 columns that the two halves of the wave update one after the other
// (k <= 1024: LDS operations of one wave complete in order, the wave barrier only pins the compiler).  Fixed per-lane
// order, fixed fold order: deterministic.
template <int IDXB>
__global__ void k_point_grad_any(const float* g, const void* idx, const float* alpha, int64_t n, int64_t row, int64_t nb,
                                 int k, int C, float* part /* [grid][k] */) {
    extern __shared__ __attribute__((aligned(16))) float bins[];        // [k][C]
    const int T = blockDim.x;                                           // 256 (C == 256) or 64 (C == 64 / 32)
    for (int j = threadIdx.x; j < k * C; j += T) bins[j] = 0.0f;
    __syncthreads();
    float* col = bins + (threadIdx.x % C);
    const int half = threadIdx.x / C;                                   // 0, or 1 for the upper half-wave when C == 32
    const bool two_phases = C < T;
    const int64_t tid = (int64_t)blockIdx.x * T + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * T;
    // bucket of element i = tid + j * nth, advanced by (nth / row, nth % row) per step
    int64_t bkt = nb == 1 ? 0 : tid / row;
    int64_t rem = nb == 1 ? 0 : tid % row;
    const int64_t dq = nb == 1 ? 0 : nth / row, dr = nb == 1 ? 0 : nth % row;
    constexpr int U = 4;                                                 // loads in flight per lane
    const int64_t rounds = (n + nth * U - 1) / (nth * U);               // uniform over the grid: the phases below need whole waves
    int64_t i = tid;
    for (int64_t r = 0; r < rounds; ++r) {
        float m[U];
        int id[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = i < n;
            const int64_t ic = live ? i : 0;
            const float gv = g[ic];
            const float av = alpha[live ? bkt : 0];
            const int ii = IDXB == 8 ? (int)((const int64_t*)idx)[ic] : (int)((const uint8_t*)idx)[ic];
            m[u] = live ? gv * av : 0.0f;                               // one fp32 multiply, :495; a dead lane adds +0 to bin 0
            id[u] = live ? ii : 0;
            i += nth;
            bkt += dq; rem += dr;
            if (rem >= row) { rem -= row; ++bkt; }
        }
        __builtin_amdgcn_sched_barrier(0);                              // all loads issued before the first table update
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!two_phases) {
                col[id[u] * C] += m[u];
            } else {
                if (half == 0) col[id[u] * C] += m[u];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (half == 1) col[id[u] * C] += m[u];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += T) {                          // fixed order over the columns, rotated start
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc += bins[j * C + ((c + j) & (C - 1))];
        part[(int64_t)blockIdx.x * k + j] = acc;
    }
}
// stage 2: one block per bin folds the per-block partials in a fixed order (thread t sums rows
// t, t+256, ... in float64, then a fixed shuffle/LDS tree): reproducible run to run
__global__ __launch_bounds__(256) void k_point_grad_final(const float* part, int nblocks, int k, float* out) {
    __shared__ double red[4];
    const int j = blockIdx.x;
    double acc = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) acc += (double)part[(int64_t)b * k + j];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[j] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

// ---- K7: 'complicated' STE backward (quant_functions.py:319-406) --------------------------------
// Per bucket S = sum_i g_i (qs_i - u_i), out = g, out[jmax] += S, out[jmin] -= S.  Every TERM follows the reference's own
// fp32 operations -- qs = (q - beta_q) / alpha_q (:350, scale_down of the quantized tensor), u = (x - beta_q) / alpha_q,
// d = qs - u, t = g d (:400), each rounded -- so that only the order of the fp32 summation differs from the reference
// (torch.mm over a sparse +-1 matrix): the bucket sum agrees with a float64 sum of the same terms to ~1e-7 of sum|t|.
// (Round 2 hoisted the division out of the sum, S = (sum g (q - x)) / alpha_q: closer to exact arithmetic than the
// reference, but 3e-7 / 6e-6 of sum|t| away from THE REFERENCE at 16 / 256 levels, because the reference's two quotients
// are rounded before they are subtracted.)  The two quotients per element use the bucket-invariant division (three VALU
// operations each, qd_common.h); here the quotient itself is consumed, so the form is only taken when it is exact for
// every numerator of the bucket: alpha_q in [2^-60, 2^100] and no numerator in (0, 2^-100) -- a wave-uniform choice,
// otherwise the IEEE division.  The level of each element (the only thing q depends on) comes from the same shortcuts as
// K1: bucket-invariant division of (x - beta) / alpha, and level / (s-1) from the per-row table for <= 16 levels.
template <bool FAST>
__device__ __forceinline__ float ste_term(float g, float q, float x, float aq, float bq, float yq) {
    float qs = q - bq;  qs = div_alpha<FAST>(qs, aq, yq);
    float u = x - bq;   u = div_alpha<FAST>(u, aq, yq);
    const float d = qs - u;
    return g * d;
}
__device__ __forceinline__ bool ste_tiny_numerator(float x, float bq) {     // 0 < |x - beta_q| < 2^-100: outside the proven range
    const float n = fabsf(x - bq);
    return n < 0x1p-100f && n != 0.0f;
}

// vector path: LPB lanes per bucket, V float4 per lane, the whole bucket (x, g, q) in registers, one pass.  Index ties:
// the FIRST element (in memory order) at the top / bottom level of the quantized bucket (or the true arg of x).
template <int LPB, int V>
__global__ __launch_bounds__(256) void k_ste_backward_vec(const float* x, const float* g, float* out, int64_t nvec,
                                                          float sm1, int tie_mode) {
    constexpr int BPW = 64 / LPB;
    constexpr int ROW = LPB * V * 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPB, l = lane % LPB;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (nvec + BPW - 1) / BPW;
    const bool use_tab = sm1 <= 15.0f;                       // DPP rows are active or inactive as a whole here (qdq_tab)
    const float tab = (float)(lane & 15) / sm1;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t bkt = t * BPW + sub;
        if (bkt >= nvec) continue;
        const int64_t e0 = bkt * ROW + (int64_t)l * 4;
        f4 xv[V], gv[V], qv[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            xv[j] = __builtin_nontemporal_load((const f4*)(x + e0) + j * LPB);
            gv[j] = __builtin_nontemporal_load((const f4*)(g + e0) + j * LPB);
        }
        // NaN-propagating, as torch's min / max and as the forward kernel: a bucket that holds a NaN (or an infinity: alpha =
        // inf) quantizes to NaN as a whole, its sum S is NaN, and the reference adds and subtracts S at the FIRST NaN -- the
        // position torch's min and max both report (quant_functions.py:350,383-400)
        float mn = pmin4(xv[0]), mx = pmax4(xv[0]);
#pragma unroll
        for (int j = 1; j < V; ++j) { mn = pmin(mn, pmin4(xv[j])); mx = pmax(mx, pmax4(xv[j])); }
        mn = group_min<LPB>(mn); mx = group_max<LPB>(mx);
        float a, b;
        alpha_beta(mn, mx, a, b);
        auto quantize = [&](auto fast_c, auto tab_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            constexpr bool TAB = decltype(tab_c)::value;
            const float y = FAST ? 1.0f / a : 0.0f;
            float lev;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (TAB) {
                    qv[j].x = qdq_tab<FAST>(xv[j].x, a, b, sm1, 0.0f, lev, tab, y); qv[j].y = qdq_tab<FAST>(xv[j].y, a, b, sm1, 0.0f, lev, tab, y);
                    qv[j].z = qdq_tab<FAST>(xv[j].z, a, b, sm1, 0.0f, lev, tab, y); qv[j].w = qdq_tab<FAST>(xv[j].w, a, b, sm1, 0.0f, lev, tab, y);
                } else {
                    qv[j].x = qdq<FAST>(xv[j].x, a, b, sm1, 0.0f, lev, y); qv[j].y = qdq<FAST>(xv[j].y, a, b, sm1, 0.0f, lev, y);
                    qv[j].z = qdq<FAST>(xv[j].z, a, b, sm1, 0.0f, lev, y); qv[j].w = qdq<FAST>(xv[j].w, a, b, sm1, 0.0f, lev, y);
                }
            }
        };
        const bool fa = !__any(!fastdiv_ok(a));              // wave-uniform
        if (fa) { if (use_tab) quantize(std::true_type{}, std::true_type{}); else quantize(std::true_type{}, std::false_type{}); }
        else { if (use_tab) quantize(std::false_type{}, std::true_type{}); else quantize(std::false_type{}, std::false_type{}); }
        float qmn = pmin4(qv[0]), qmx = pmax4(qv[0]);
#pragma unroll
        for (int j = 1; j < V; ++j) { qmn = pmin(qmn, pmin4(qv[j])); qmx = pmax(qmx, pmax4(qv[j])); }
        qmn = group_min<LPB>(qmn); qmx = group_max<LPB>(qmx);
        float aq, bq;
        alpha_beta(qmn, qmx, aq, bq);                       // scale_down of the QUANTIZED bucket, :350
        int jmax = 0x7fffffff, jmin = 0x7fffffff;           // index inside the bucket
        const bool ref_tie = tie_mode == QD_STE_TIE_REFERENCE;
        const bool bad = ref_tie ? (qmn != qmn) : (mn != mn);   // the searched tensor's min / max are NaN: both sit at its first NaN
        bool tiny = false;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int base = (j * LPB + l) * 4;
#define QD_STE_ELEM(c, off)                                                          \
            {                                                                        \
                tiny |= ste_tiny_numerator(xv[j].c, bq);                             \
                const float sv = ref_tie ? qv[j].c : xv[j].c;                        \
                const bool top = bad ? (sv != sv) : (sv == (ref_tie ? qmx : mx));    \
                const bool bot = bad ? (sv != sv) : (sv == (ref_tie ? qmn : mn));    \
                jmax = (top && base + off < jmax) ? base + off : jmax;               \
                jmin = (bot && base + off < jmin) ? base + off : jmin;               \
            }
            QD_STE_ELEM(x, 0) QD_STE_ELEM(y, 1) QD_STE_ELEM(z, 2) QD_STE_ELEM(w, 3)
#undef QD_STE_ELEM
        }
        float sum = 0.0f;
        auto bucket_sum = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            const float yq = FAST ? 1.0f / aq : 0.0f;
#pragma unroll
            for (int j = 0; j < V; ++j) {                    // per lane in memory order, then the DPP tree
                sum += ste_term<FAST>(gv[j].x, qv[j].x, xv[j].x, aq, bq, yq);
                sum += ste_term<FAST>(gv[j].y, qv[j].y, xv[j].y, aq, bq, yq);
                sum += ste_term<FAST>(gv[j].z, qv[j].z, xv[j].z, aq, bq, yq);
                sum += ste_term<FAST>(gv[j].w, qv[j].w, xv[j].w, aq, bq, yq);
            }
        };
        if (!__any(!fastdiv_ok(aq) || tiny)) bucket_sum(std::true_type{}); else bucket_sum(std::false_type{});
        sum = group_sum<LPB>(sum); jmax = group_imin<LPB>(jmax); jmin = group_imin<LPB>(jmin);
        const bool touch = jmax != jmin || bad;             // constant bucket: +S and -S cancel; a NaN bucket: (g + S) - S = NaN there
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int base = (j * LPB + l) * 4;
            f4 o = gv[j];
            if (touch) {
                if (base + 0 == jmax) o.x = o.x + sum;  if (base + 0 == jmin) o.x = o.x - sum;
                if (base + 1 == jmax) o.y = o.y + sum;  if (base + 1 == jmin) o.y = o.y - sum;
                if (base + 2 == jmax) o.z = o.z + sum;  if (base + 2 == jmin) o.z = o.z - sum;
                if (base + 3 == jmax) o.w = o.w + sum;  if (base + 3 == jmin) o.w = o.w - sum;
            }
            if (V == 1) store_nt_pinned((QD_AS_GLOBAL f4*)(out + e0) + j * LPB, o);      // (the V = 1 instance loses the hint otherwise)
            else __builtin_nontemporal_store(o, (f4*)(out + e0) + j * LPB);
        }
    }
}

// any bucket size / alignment: one wave per bucket, four passes over the (L1/L2-resident) bucket
__global__ __launch_bounds__(256) void k_ste_backward(const float* x, const float* g, float* out, int64_t n,
                                                      int64_t row, int64_t first, int64_t nb, float sm1, int tie_mode) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t bkt = first + wave; bkt < nb; bkt += nwaves) {
        const int64_t lo = bkt * row;
        const int64_t hi = lo + row < n ? lo + row : n;
        // pass 1: alpha/beta of x
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int64_t i = lo + lane; i < hi; i += 64) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); nan |= (v != v); }
        mn = wave_min(mn); mx = wave_max(mx);
        if (group_any<64>(nan)) { mn = NAN; mx = NAN; }      // torch's min / max propagate a NaN (as the register-resident path does)
        float a, b;
        alpha_beta(mn, mx, a, b);
        // pass 2: min/max of the QUANTIZED bucket (the reference re-runs scale_down on q, :350)
        float qmn = INFINITY, qmx = -INFINITY;
        bool tiny = false, qnan = false;
        for (int64_t i = lo + lane; i < hi; i += 64) {
            float lev;
            const float q = qdq(x[i], a, b, sm1, 0.0f, lev);
            qmn = fminf(qmn, q); qmx = fmaxf(qmx, q); qnan |= (q != q);
        }
        qmn = wave_min(qmn); qmx = wave_max(qmx);
        if (group_any<64>(qnan)) { qmn = NAN; qmx = NAN; }
        float aq, bq;
        alpha_beta(qmn, qmx, aq, bq);
        const bool bad = tie_mode == QD_STE_TIE_REFERENCE ? (qmn != qmn) : (mn != mn);   // min / max of the searched tensor are NaN: both at its first NaN
        for (int64_t i = lo + lane; i < hi; i += 64) tiny |= ste_tiny_numerator(x[i], bq);
        // pass 3: S_b (the reference's own per-element operations, :400) and the first index at the top / bottom level
        // (or the true arg of x)
        float s = 0.0f;
        long long jmax = INT64_MAX, jmin = INT64_MAX;
        auto pass3 = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            const float yq = FAST ? 1.0f / aq : 0.0f;
            for (int64_t i = lo + lane; i < hi; i += 64) {
                float lev;
                const float xv = x[i];
                const float q = qdq(xv, a, b, sm1, 0.0f, lev);
                s += ste_term<FAST>(g[i], q, xv, aq, bq, yq);
                const float sv = tie_mode == QD_STE_TIE_REFERENCE ? q : xv;
                const bool top = bad ? (sv != sv) : (sv == (tie_mode == QD_STE_TIE_REFERENCE ? qmx : mx));
                const bool bot = bad ? (sv != sv) : (sv == (tie_mode == QD_STE_TIE_REFERENCE ? qmn : mn));
                if (top && (long long)i < jmax) jmax = i;
                if (bot && (long long)i < jmin) jmin = i;
            }
        };
        if (!__any(!fastdiv_ok(aq) || tiny)) pass3(std::true_type{}); else pass3(std::false_type{});
        s = wave_sum(s);
        jmax = wave_min_ll(jmax);
        jmin = wave_min_ll(jmin);
        // pass 4: out = g, +S at jmax, -S at jmin (they cancel when the bucket is constant)
        for (int64_t i = lo + lane; i < hi; i += 64) {
            float o = g[i];
            if (jmax != jmin || bad) {                     // (a NaN bucket: (g + S) - S = NaN at the first NaN, as the reference)
                if (i == jmax) o = o + s;
                if (i == jmin) o = o - s;
            }
            out[i] = o;
        }
    }
}

// ---- K8: 'truncated' STE ----------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_clamp(float* w, int64_t n, float limit) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if ((((uintptr_t)w) & kDataAlign) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            f4 v = ((f4*)w)[i];
            f4 r;
            r.x = v.x > limit ? limit : (v.x < -limit ? -limit : v.x);
            r.y = v.y > limit ? limit : (v.y < -limit ? -limit : v.y);
            r.z = v.z > limit ? limit : (v.z < -limit ? -limit : v.z);
            r.w = v.w > limit ? limit : (v.w < -limit ? -limit : v.w);
            if (r.x != v.x || r.y != v.y || r.z != v.z || r.w != v.w) ((f4*)w)[i] = r;   // write only what changes
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += nth) {
        float v = w[i];
        v = v > limit ? limit : v;
        v = v < -limit ? -limit : v;
        w[i] = v;
    }
}
// grad[|w| > limit] = 0: reads w (4 B) and touches grad only where the mask hits
__global__ __launch_bounds__(256) void k_truncated_ste(const float* w, float* grad, int64_t n, float limit) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    int64_t done = 0;
    if ((((uintptr_t)w | (uintptr_t)grad) & kDataAlign) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = tid; i < n4; i += nth) {
            const f4 v = __builtin_nontemporal_load((const f4*)w + i);
            const bool m0 = fabsf(v.x) > limit, m1 = fabsf(v.y) > limit, m2 = fabsf(v.z) > limit, m3 = fabsf(v.w) > limit;
            if (m0 | m1 | m2 | m3) {                      // one 16-byte read-modify-write instead of 4-byte pokes
                f4 gv = ((const f4*)grad)[i];
                gv.x = m0 ? 0.0f : gv.x; gv.y = m1 ? 0.0f : gv.y; gv.z = m2 ? 0.0f : gv.z; gv.w = m3 ? 0.0f : gv.w;
                ((f4*)grad)[i] = gv;
            }
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < n; i += nth)
        if (fabsf(w[i]) > limit) grad[i] = 0.0f;
}

// ---- multi-tensor K1: one launch for every parameter of a model -------------------------------
// A tile = 4 buckets of one tensor = one wave iteration; a DPP row owns a bucket.  Full, 16-byte
// aligned 256-element buckets take the register path, everything else the row16 scalar path.
template <int ROW>
__global__ __launch_bounds__(256) void k_multi_uniform(const QdTensorDesc* __restrict__ table, int ntensors, int64_t total_tiles,
                                                       int64_t bucket, float sm1) {
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4, l = lane & 15;
    const int64_t wave = uniform_wave_index();      // scalar: the table search below runs on s_load
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    Prep pp;
    pp.mean = 0.0f;
    pp.me = INFINITY;
    const bool use_tab = sm1 <= 15.0f;
    const float tab = (float)(lane & 15) / sm1;
    for (int64_t t = wave; t < total_tiles; t += nwaves) {
        int lo_t = 0, hi_t = ntensors - 1;               // last tensor with first_tile <= t
        while (lo_t < hi_t) {
            const int mid = (lo_t + hi_t + 1) >> 1;
            if (table[mid].first_tile <= t) lo_t = mid; else hi_t = mid - 1;
        }
        const QdTensorDesc d = table[lo_t];
        KParams p;
        p.x = d.x; p.out = d.q; p.n = d.n;
        p.row = d.n < bucket ? d.n : bucket;
        p.nb = (d.n + p.row - 1) / p.row;
        p.alpha = nullptr; p.beta = nullptr; p.mean = nullptr; p.me = INFINITY; p.sm1 = sm1; p.lev8 = nullptr;
        p.idx = nullptr; p.idx_bytes = 0; p.pts = nullptr; p.k = 0; p.assign_mode = 0; p.prescaled = 0;
        p.stochastic = 0; p.seed = 0; p.nvec = 0;
        const int64_t bkt = (t - d.first_tile) * 4 + sub;
        if (bkt >= p.nb) continue;
        const int64_t lo = bkt * p.row;
        const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
        const bool fast = ROW > 0 && (hi - lo) == ROW && p.row == ROW &&
                          (((((uintptr_t)d.x) | ((uintptr_t)d.q)) & 15) == 0);
        if (fast) {
            constexpr int V = ROW > 0 ? ROW / 64 : 1;
            const f4* src = (const f4*)(p.x + lo) + l;
            f4 v[V];
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = ldg_nt(src + j * 16);   // masters: read once
            float mn = pmin4(v[0]), mx = pmax4(v[0]);      // NaN-propagating
#pragma unroll
            for (int j = 1; j < V; ++j) { mn = pmin(mn, pmin4(v[j])); mx = pmax(mx, pmax4(v[j])); }
            mn = row16_min(mn); mx = row16_max(mx);
            float a, b, lev;
            alpha_beta(mn, mx, a, b);
            f4* dst = (f4*)(p.out + lo) + l;
            // rows of the wave that took this branch: all in the proven range -> bucket-invariant division (qd_common.h)
            const bool fdiv = !__any(!fastdiv_ok(a));
            auto body = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                const float y = FAST ? 1.0f / a : 0.0f;
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    f4 r;
                    if (use_tab) {                         // <= 16 levels: see k_bucket_vec (a DPP row is active as a whole here)
                        r.x = qdq_tab<FAST>(v[j].x, a, b, sm1, 0.0f, lev, tab, y);
                        r.y = qdq_tab<FAST>(v[j].y, a, b, sm1, 0.0f, lev, tab, y);
                        r.z = qdq_tab<FAST>(v[j].z, a, b, sm1, 0.0f, lev, tab, y);
                        r.w = qdq_tab<FAST>(v[j].w, a, b, sm1, 0.0f, lev, tab, y);
                    } else {
                        r.x = qdq<FAST>(v[j].x, a, b, sm1, 0.0f, lev, y);
                        r.y = qdq<FAST>(v[j].y, a, b, sm1, 0.0f, lev, y);
                        r.z = qdq<FAST>(v[j].z, a, b, sm1, 0.0f, lev, y);
                        r.w = qdq<FAST>(v[j].w, a, b, sm1, 0.0f, lev, y);
                    }
                    stg_nt(r, dst + j * 16);
                }
            };
            if (fdiv) body(std::true_type{}); else body(std::false_type{});
        } else {
            bucket_row16<MODE_QDQ>(p, nullptr, bkt, lo, hi, l, pp);
        }
    }
}

}  // namespace

extern "C" {

int qd_mean_f32(const float* x, int64_t n, float* mean_out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !mean_out || n <= 0) return QD_ERR_INVALID_ARGUMENT;
    Workspace w;
    if (!carve(workspace, workspace_bytes, w)) return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    int pb = blocks_for(n, 256 * 4 * 8);
    if (pb > kPartialBlocks) pb = kPartialBlocks;
    hipLaunchKernelGGL(k_sum_partial, dim3(pb), dim3(256), 0, st, x, n, w.sum_part);
    hipLaunchKernelGGL(k_mean_final, dim3(1), dim3(256), 0, st, w.sum_part, pb, n, mean_out);
    return check_launch();
}

int qd_inv_scale_f32(const float* u, float* y, int64_t n, int64_t bucket, const float* alpha, const float* beta,
                     const float* mean, void* stream) {
    if (n < 0 || bucket < 0 || (n > 0 && (!u || !y || !alpha || !beta))) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    const int blocks = blocks_for(n, 256 * 4 * 4);          // one 4 KiB tile per wave
    if (nb == 1 || (row & 3) == 0)
        hipLaunchKernelGGL(k_inv_scale<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, u, y, n, row, nb, alpha, beta, mean);
    else
        hipLaunchKernelGGL(k_inv_scale<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, u, y, n, row, nb, alpha, beta,
                       mean);
    return check_launch();
}

int qd_bucket_argminmax_f32(const float* x, int64_t n, int64_t bucket, const float* mean, int clamp,
                            float max_element, int64_t* argmin, int64_t* argmax, void* workspace,
                            size_t workspace_bytes, void* stream) {
    if (n <= 0 || bucket < 0 || !x || !argmin || !argmax) return QD_ERR_INVALID_ARGUMENT;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    hipStream_t st = (hipStream_t)stream;
    const float me = clamp ? max_element : INFINITY;
    int chunks = 1;
    if (nb == 1 && n > 65536) {
        chunks = (int)((n + 65535) / 65536);
        if (chunks > kPartialBlocks) chunks = kPartialBlocks;
    }
    Workspace w = {};
    if (chunks > 1 && !carve(workspace, workspace_bytes, w)) return QD_ERR_WORKSPACE_TOO_SMALL;
    const int threads = row <= 64 ? 64 : 256;
    const int blocks = blocks_for(nb * chunks, 1);
    hipLaunchKernelGGL(k_argminmax, dim3(blocks), dim3(threads), 0, st, x, n, row, nb, chunks, mean, me, argmin,
                       argmax, w.arg_pv, w.arg_pi);
    if (chunks > 1) hipLaunchKernelGGL(k_arg_final, dim3(1), dim3(256), 0, st, w.arg_pv, w.arg_pi, chunks, argmin, argmax);
    return check_launch();
}

int qd_point_grad_f32(const float* g, const void* idx, int idx_bytes, const float* alpha, int64_t n, int64_t bucket,
                      int k, float* grad_points, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || bucket < 0 || k < 1 || k > kMaxPoints || !grad_points || (n > 0 && (!g || !idx || !alpha)))
        return QD_ERR_INVALID_ARGUMENT;
    if (idx_bytes != 8 && idx_bytes != 1) return QD_ERR_INVALID_ARGUMENT;
    Workspace w;
    if (!carve(workspace, workspace_bytes, w)) return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    int64_t nb, row;
    geometry(n > 0 ? n : 1, bucket, nb, row);
    // partial rows of k floats each must fit the workspace's [kPartialBlocks * kMaxPoints] floats
    int blocks = blocks_for(n, 256 * 4 * 2);
    const int64_t max_rows = (int64_t)kPartialBlocks * kMaxPoints / k;
    // Grid: measured at 64 Mi elements, k = 4 / 16 (blocks x float4 in flight per lane): 2048 x 2 -> 65.5 / 67.0 us,
    // 1024 x 2 -> 57 / 56, 1024 x 4 -> 62 / 65, 512 x 4 -> 54.4 / 54.1, 512 x 8 -> 59 / 59, 384 x 4 -> 57 / 59,
    // 256 x 8 -> 70 / 70: about 2048 waves x 4 independent 1 KiB streams is what the HBM controllers like; more
    // concurrent streams cost more than they hide.  (It also leaves the fold kernel 512 rows instead of 2048.)
    const int64_t cap = 512;
    if (blocks > cap) blocks = (int)cap;
    if (blocks > max_rows) blocks = (int)max_rows;
    int row_shift = 0;
    const bool pow2 = nb == 1 || (row & (row - 1)) == 0;
    if (nb > 1 && pow2) while (((int64_t)1 << row_shift) < row) ++row_shift;
    const bool idx_ok = idx_bytes == 8 ? ((((uintptr_t)idx) & 15) == 0) : ((((uintptr_t)idx) & 3) == 0);
    // (any bucket size from 4 elements up: BK = 2 walks the buckets; only shorter buckets and misaligned tensors are left
    // to the scalar kernel)
    const bool walk_ok = pow2 || (nb < ((int64_t)1 << 30) && row < ((int64_t)1 << 30));
    const bool fast = k <= 512 && idx_ok && ((((uintptr_t)g) & kDataAlign) == 0) && (nb == 1 || (row >= 4 && walk_ok));
    if (fast) {
        // k <= 4: register bins; otherwise an LDS table [k][threads] of lane-private columns
        const int threads = k <= 128 ? 256 : (k <= 256 ? 128 : 64);
        // 64 < k <= 128: the two half-waves of a wave share 32 columns -- a [k][128] table for the 256 lanes (<= 64 KiB, two
        // blocks per CU) instead of four waves passing a token around [k][64]
        const bool halves = k > 64 && k <= 128;
        const size_t lds_bytes = (size_t)(k <= 4 ? 4 * 4 : k * (halves ? threads / 2 : threads)) * sizeof(float);
        // tables above 16 KiB leave few waves per CU: a resident grid (as many blocks as fit the CUs' LDS at
        // once, so the table is zeroed and folded once per CU), more loads in flight per lane, and the merged
        // four-element update that needs one LDS round trip per float4 instead of four
        const bool big = lds_bytes > 16 * 1024;
        if (big) {
            const int per_cu = (int)((160 * 1024) / lds_bytes) > 0 ? (int)((160 * 1024) / lds_bytes) : 1;
            const int resident = num_cus() * per_cu;
            if (blocks > resident) blocks = resident;
        }
#define QD_PG(KR, IDXB, BK, U, MG, ...)                                                                             \
        {                                                                                                           \
            auto kern = k_point_grad_fast<KR, IDXB, BK, U, MG, ##__VA_ARGS__>;                                      \
            if (lds_bytes > 64 * 1024)                                                                              \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(KR > 0 ? 256 : threads), lds_bytes, st, g, idx, alpha, n,   \
                               row_shift, row, nb, k, w.pg_part);                                                   \
        }
#define QD_PG_K(IDXB, BK)                                                                                           \
        {                                                                                                           \
            if (k <= 4) QD_PG(4, IDXB, BK, 4, false)                                                                \
            else if (!big) QD_PG(0, IDXB, BK, 4, false)                                                             \
            else if (halves) QD_PG(0, IDXB, BK, 4, true, true)                                                      \
            else if (lds_bytes <= 64 * 1024) QD_PG(0, IDXB, BK, 4, true)                                            \
            else {                                           /* k > 128: four waves take turns on 64 columns */      \
                const size_t tl = (size_t)k * 64 * sizeof(float);                                                   \
                int per_cu_t = (int)((160 * 1024) / (tl + 64));   /* LDS; the 142-163 VGPRs allow 3 blocks per CU */  \
                if (per_cu_t > kTurnsBlocksPerCu) per_cu_t = kTurnsBlocksPerCu;                                      \
                int tb = num_cus() * per_cu_t;                                                                       \
                if (tb > blocks_all) tb = blocks_all;                                                                \
                if (tb > (int)max_rows) tb = (int)max_rows;                                                          \
                if (tb < 1) tb = 1;                                                                                  \
                blocks = tb;                                                                                         \
                auto kern = k_point_grad_turns<IDXB, BK, IDXB == 8 ? 4 : 8, 4>;                                      \
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl);   \
                hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), tl, st, g, idx, alpha, n, row_shift, row, nb, k, w.pg_part); \
            }                                                                                                       \
        }
        const int blocks_all = blocks_for(n, 256 * 4 * 2);
        // resident blocks per CU of the turn-token kernel; measured at k = 128: 63.4 us (2) / 65.9 us (3).  (One lane per
        // column under a table above 64 KiB -- one block per CU -- measured 66.6 / 110-121 us at k = 128 / 256 with 8 .. 32
        // float4 in flight per lane: those waves are bound by their own VALU + LDS round trips.)
        constexpr int kTurnsBlocksPerCu = 2;     /* two rings while two blocks of [k][128] fit a CU */
        if (idx_bytes == 8) { if (nb == 1) QD_PG_K(8, 0) else if (pow2) QD_PG_K(8, 1) else QD_PG_K(8, 2) }
        else { if (nb == 1) QD_PG_K(1, 0) else if (pow2) QD_PG_K(1, 1) else QD_PG_K(1, 2) }
#undef QD_PG_K
#undef QD_PG
    } else {
        // lane-private columns at any bucket size / alignment (deterministic): 256 lanes x 256 columns while the table fits
        // 64 KiB, else one wave per block on 64 (k <= 512) or 32 columns
        const int C = k <= 64 ? 256 : (k <= 512 ? 64 : 32);
        const int threads = k <= 64 ? 256 : 64;
        const size_t lds_bytes = (size_t)k * C * sizeof(float);
        int want = blocks_for(n, threads * 4 * 4);
        const int per_cu = (int)((160 * 1024) / (lds_bytes + 256));
        const int resident = num_cus() * (per_cu < 1 ? 1 : (per_cu > 8 ? 8 : per_cu));
        if (want > resident) want = resident;
        if (want > max_rows) want = (int)max_rows;
        blocks = want < 1 ? 1 : want;
        if (idx_bytes == 8) {
            auto kern = k_point_grad_any<8>;
            if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, st, g, idx, alpha, n, row, nb, k, C, w.pg_part);
        } else {
            auto kern = k_point_grad_any<1>;
            if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds_bytes, st, g, idx, alpha, n, row, nb, k, C, w.pg_part);
        }
    }
    hipLaunchKernelGGL(k_point_grad_final, dim3(k), dim3(256), 0, st, w.pg_part, blocks, k, grad_points);
    return check_launch();
}

int qd_ste_bucket_backward_f32(const float* x, const float* g, float* out, int64_t n, int64_t bucket, int levels,
                               int tie_mode, void* stream) {
    if (n < 0 || bucket <= 0 || levels < 2 || (n > 0 && (!x || !g || !out))) return QD_ERR_INVALID_ARGUMENT;
    if (tie_mode != QD_STE_TIE_REFERENCE && tie_mode != QD_STE_TIE_TRUE_ARG) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    hipStream_t st = (hipStream_t)stream;
    const float sm1 = (float)(levels - 1);
    int64_t first = 0;                                   // buckets [0, first) take the register path
    const bool aligned = (((((uintptr_t)x) | ((uintptr_t)g) | ((uintptr_t)out)) & kDataAlign) == 0) && nb > 1;
    const int64_t nfull = n / row;
#define QD_STE(LPB, V)                                                                                   \
    {                                                                                                    \
        first = nfull;                                                                                   \
        const int64_t tiles = (nfull + (64 / LPB) - 1) / (64 / LPB);                                     \
        hipLaunchKernelGGL((k_ste_backward_vec<LPB, V>), dim3(blocks_for(tiles, 4)), dim3(256), 0, st, x, g, \
                           out, nfull, sm1, tie_mode);                                                   \
    }
    if (aligned && nfull > 0) {
        if (row == 64) QD_STE(16, 1)
        else if (row == 128) QD_STE(16, 2)
        else if (row == 256) QD_STE(16, 4)      // (32,2) and (64,1) lane groupings measured slower: 169-195 / 182-188 vs 166 us
        else if (row == 512) QD_STE(64, 2)
        else if (row == 1024) QD_STE(64, 4)
    }
#undef QD_STE
    if (first < nb)
        hipLaunchKernelGGL(k_ste_backward, dim3(blocks_for(nb - first, 4)), dim3(256), 0, st, x, g, out, n, row, first,
                           nb, sm1, tie_mode);
    return check_launch();
}

int qd_clamp_f32(float* w, int64_t n, float limit, void* stream) {
    if (n < 0 || (n > 0 && !w)) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_clamp, dim3(blocks_for(n, 256 * 4 * 4)), dim3(256), 0, (hipStream_t)stream, w, n, limit);
    return check_launch();
}

int qd_truncated_ste_f32(const float* w, float* grad, int64_t n, float limit, void* stream) {
    if (n < 0 || (n > 0 && (!w || !grad))) return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_truncated_ste, dim3(blocks_for(n, 256 * 4 * 4)), dim3(256), 0, (hipStream_t)stream, w, grad, n,
                       limit);
    return check_launch();
}

int64_t qd_multi_plan(QdTensorDesc* host_table, int ntensors, int64_t bucket) {
    if (!host_table || ntensors < 0 || bucket <= 0) return -1;
    int64_t tiles = 0;
    for (int i = 0; i < ntensors; ++i) {
        int64_t nb, row;
        geometry(host_table[i].n > 0 ? host_table[i].n : 1, bucket, nb, row);
        host_table[i].first_tile = tiles;
        tiles += host_table[i].n > 0 ? (nb + 3) / 4 : 0;
    }
    return tiles;
}

int qd_multi_uniform_f32(const QdTensorDesc* table, int ntensors, int64_t total_tiles, int64_t bucket, int levels,
                         void* stream) {
    if (!table || ntensors <= 0 || total_tiles < 0 || bucket <= 0 || levels < 2) return QD_ERR_INVALID_ARGUMENT;
    if (total_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = blocks_for(total_tiles, 4);
    const float sm1 = (float)(levels - 1);
    if (bucket == 256)
        hipLaunchKernelGGL((k_multi_uniform<256>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, sm1);
    else if (bucket == 128)
        hipLaunchKernelGGL((k_multi_uniform<128>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, sm1);
    else if (bucket == 64)
        hipLaunchKernelGGL((k_multi_uniform<64>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, sm1);
    else
        hipLaunchKernelGGL((k_multi_uniform<0>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, sm1);
    return check_launch();
}

}  // extern "C"
