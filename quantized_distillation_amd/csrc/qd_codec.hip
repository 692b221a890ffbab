// qd_codec.hip -- packed-index codec for uniformly quantized tensors + level histograms (gfx950).
//
// The reference never materialises the compressed form; it only accounts for its size:
// helpers/functions.py:226-262 charges  bits*N/8  bytes for the level indices plus 8 bytes
// (alpha, beta as fp32) per bucket, and quantization/help_functions.py:175-232 computes the Huffman
// mean code length from the histogram of the level indices.  This file produces exactly that
// representation on the device -- `bits`-per-element packed level indices + per-bucket alpha/beta --
// decodes it back to the fake-quantized fp32 tensor (bit-identical to qd_uniform_f32's output),
// and builds the level histogram without copying the tensor to the host.
//
// pack   : read 4 B/elem, write bits/8 B/elem           (HBM-bound on the read)
// unpack : read bits/8 B/elem, write 4 B/elem            (HBM-bound on the write)
// hist   : read 1 B/elem (uint8 indices)
#include "qd_common.h"

#include <atomic>
#include <type_traits>
#include "../../include/qd_hip.h"

using namespace qd;

namespace {

template <int BITS>
__device__ __forceinline__ uint32_t pack4(const float (&lev)[4]) {
    return (uint32_t)(int)lev[0] | ((uint32_t)(int)lev[1] << BITS) | ((uint32_t)(int)lev[2] << (2 * BITS)) |
           ((uint32_t)(int)lev[3] << (3 * BITS));
}

// LPB lanes per bucket, V float4 per lane (same geometry as k_bucket_vec).  Element e of the
// tensor occupies bits [e*BITS, (e+1)*BITS) of the packed stream (little endian inside a byte).
template <int LPB, int V, int BITS>
__global__ __launch_bounds__(256) void k_pack_vec(const float* x, uint8_t* packed, float* alpha, float* beta,
                                                  int64_t nvec, float sm1) {
    constexpr int BPW = 64 / LPB;
    constexpr int ROW = LPB * V * 4;
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPB, l = lane % LPB;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (nvec + BPW - 1) / BPW;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t bkt = t * BPW + sub;
        if (bkt >= nvec) continue;
        const int64_t e0 = bkt * ROW + (int64_t)l * 4;
        f4 v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = __builtin_nontemporal_load((const f4*)(x + e0) + j * LPB);
        float mn = INFINITY, mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            mn = fminf(mn, fminf(fminf(v[j].x, v[j].y), fminf(v[j].z, v[j].w)));
            mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
        }
        bool nan = false;
#pragma unroll
        for (int j = 0; j < V; ++j) nan |= has_nan4(v[j]);
        if (LPB == 16) { mn = row16_min(mn); mx = row16_max(mx); } else { mn = wave_min(mn); mx = wave_max(mx); }
        if (group_any<LPB>(nan)) { mn = NAN; mx = NAN; }   // a NaN bucket has NaN alpha/beta (its indices are meaningless)
        float a, b;
        alpha_beta(mn, mx, a, b);
        if (l == 0) { if (alpha) alpha[bkt] = a; if (beta) beta[bkt] = b; }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int64_t e = e0 + (int64_t)j * LPB * 4;
            float lev[4];
            qdq(v[j].x, a, b, sm1, 0.0f, lev[0]);
            qdq(v[j].y, a, b, sm1, 0.0f, lev[1]);
            qdq(v[j].z, a, b, sm1, 0.0f, lev[2]);
            qdq(v[j].w, a, b, sm1, 0.0f, lev[3]);
            const uint32_t pk = pack4<BITS>(lev);
            if (BITS == 8) *(uint32_t*)(packed + e) = pk;
            else if (BITS == 4) *(uint16_t*)(packed + (e >> 1)) = (uint16_t)pk;
            else if (BITS == 2) packed[e >> 2] = (uint8_t)pk;
            else {                                   // 1 bit: two lanes share a byte; pair them with a quad swap
                const uint32_t other = (uint32_t)dpp_i<0xB1>((int)pk);      // quad_perm [1,0,3,2]
                if ((l & 1) == 0) packed[e >> 3] = (uint8_t)(pk | (other << 4));
            }
        }
    }
}

// ragged last bucket [lo, n): one wave; every lane produces whole output bytes
template <int BITS>
__global__ __launch_bounds__(64) void k_pack_tail(const float* x, uint8_t* packed, float* alpha, float* beta,
                                                  int64_t lo, int64_t n, int64_t bkt, float sm1) {
    const int lane = threadIdx.x;
    float mn = INFINITY, mx = -INFINITY;
    bool nan = false;
    for (int64_t i = lo + lane; i < n; i += 64) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); nan |= (v != v); }
    mn = wave_min(mn); mx = wave_max(mx);
    if (group_any<64>(nan)) { mn = NAN; mx = NAN; }
    float a, b;
    alpha_beta(mn, mx, a, b);
    if (lane == 0) { if (alpha) alpha[bkt] = a; if (beta) beta[bkt] = b; }
    constexpr int EPB = 8 / BITS;                    // elements per byte
    const int64_t first_byte = (lo * BITS) >> 3;     // lo is a multiple of the bucket (>= 64)
    const int64_t nbytes = (((n - lo) * BITS) + 7) >> 3;
    for (int64_t t = lane; t < nbytes; t += 64) {
        uint32_t byte = 0;
#pragma unroll
        for (int c = 0; c < EPB; ++c) {
            const int64_t e = lo + t * EPB + c;
            if (e < n) {
                float lev;
                qdq(x[e], a, b, sm1, 0.0f, lev);
                byte |= (uint32_t)(int)lev << (c * BITS);
            }
        }
        packed[first_byte + t] = (uint8_t)byte;
    }
}

// decode: each thread expands 4 consecutive elements into ONE float4 (lanes contiguous: a wave
// writes 1 KiB per store instruction); the packed read is 4*BITS bits per lane
template <int BITS>
__global__ __launch_bounds__(256) void k_unpack(const uint8_t* packed, float* y, const float* alpha, const float* beta,
                                                int64_t n, int row_shift, float sm1) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t ngroups = (n + 3) >> 2;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const bool y_vec = (((uintptr_t)y) & 15) == 0;
    for (int64_t gI = tid; gI < ngroups; gI += nth) {
        const int64_t e0 = gI << 2;
        const int64_t bkt = e0 >> row_shift;
        const float a = alpha[bkt], b = beta[bkt];
        uint32_t bits;
        const bool full = e0 + 4 <= n;
        if (BITS == 8) {
            if (full) bits = *(const uint32_t*)(packed + e0);
            else { bits = 0; for (int64_t c = 0; e0 + c < n; ++c) bits |= (uint32_t)packed[e0 + c] << (8 * c); }
        } else if (BITS == 4) {
            if (full) bits = *(const uint16_t*)(packed + (e0 >> 1));
            else { bits = packed[e0 >> 1]; if (e0 + 2 < n) bits |= (uint32_t)packed[(e0 >> 1) + 1] << 8; }
        } else if (BITS == 2) {
            bits = packed[e0 >> 2];
        } else {
            bits = (uint32_t)packed[e0 >> 3] >> (e0 & 4);      // low or high nibble of the shared byte
        }
        float out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float r = (float)((bits >> (c * BITS)) & MASK);
            float w = r / sm1;                         // same three ops as the tail of qdq()
            float v = w * a;
            v = v + b;
            out[c] = v + 0.0f;
        }
        if (full && y_vec) {
            f4 o = {out[0], out[1], out[2], out[3]};
            __builtin_nontemporal_store(o, (f4*)(y + e0));
        } else {
            for (int c = 0; c < 4 && e0 + c < n; ++c) y[e0 + c] = out[c];
        }
    }
}

// (Round 4 measured a variant that loads the packed stream 16 bytes per lane -- a wave parks 1 KiB of it in LDS and decodes
// 8 float4 stores from there: HBM traffic 1.009 x the algorithmic bytes against 1.064 x for this kernel by the PMC counters,
// but 56-60 us against 47.9 us at 64 Mi elements, with or without a resident grid, the next chunk's load in flight, or a level
// table in LDS.  Here the whole grid sweeps the output front to back, 1 KiB per wave and store; there every wave streams
// into its own 8 KiB region, thousands of concurrent write streams.  docs/history/profiles/r04_ab_codec.txt.)
// uint8 level indices -> packed bits (any bucket geometry: the levels come from qd_uniform_f32's level_idx output).
// Every thread produces one 32-bit word = 32 / BITS levels; the last, partial word byte by byte.
template <int BITS>
__global__ __launch_bounds__(256) void k_pack_levels(const uint8_t* lev, int64_t n, uint8_t* packed) {
    constexpr int L = 32 / BITS;                     // levels per 32-bit word
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t nwords = n / L;
    const bool vec = ((((uintptr_t)lev) & (L < 16 ? L - 1 : 15)) == 0) && ((((uintptr_t)packed) & 3) == 0);
    for (int64_t w = tid; w < nwords; w += nth) {
        const uint8_t* src = lev + w * L;
        uint32_t word = 0;
        if (vec) {
            uint32_t in[L / 4];
            if (L == 4) in[0] = *(const uint32_t*)src;
            else if (L == 8) { const uint2 t = *(const uint2*)src; in[0] = t.x; in[1] = t.y; }
            else {
#pragma unroll
                for (int q = 0; q < L / 16; ++q) {
                    const uint4 t = *(const uint4*)(src + 16 * q);
                    in[4 * q] = t.x; in[4 * q + 1] = t.y; in[4 * q + 2] = t.z; in[4 * q + 3] = t.w;
                }
            }
#pragma unroll
            for (int c = 0; c < L; ++c) word |= ((in[c >> 2] >> (8 * (c & 3))) & MASK) << (c * BITS);
            *(uint32_t*)(packed + 4 * w) = word;
        } else {
#pragma unroll
            for (int c = 0; c < L; ++c) word |= ((uint32_t)src[c] & MASK) << (c * BITS);
            packed[4 * w] = (uint8_t)word; packed[4 * w + 1] = (uint8_t)(word >> 8);
            packed[4 * w + 2] = (uint8_t)(word >> 16); packed[4 * w + 3] = (uint8_t)(word >> 24);
        }
    }
    if (tid == 0) {                                  // the levels after the last whole word
        constexpr int EPB = 8 / BITS;
        const int64_t e0 = nwords * L;
        for (int64_t e = e0; e < n; e += EPB) {
            uint32_t byte = 0;
            for (int c = 0; c < EPB && e + c < n; ++c) byte |= ((uint32_t)lev[e + c] & MASK) << (c * BITS);
            packed[(e * BITS) >> 3] = (uint8_t)byte;
        }
    }
}

// decode for ANY bucket size (bucket == 0: one alpha / beta for the tensor): as k_unpack, the bucket of every element from
// one 32- or 64-bit division per group of four
template <int BITS>
__global__ __launch_bounds__(256) void k_unpack_any(const uint8_t* packed, float* y, const float* alpha, const float* beta,
                                                    int64_t n, int64_t bucket, float sm1) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t ngroups = (n + 3) >> 2;
    constexpr uint32_t MASK = (1u << BITS) - 1u;
    const bool y_vec = (((uintptr_t)y) & 15) == 0;
    const bool small = n < ((int64_t)1 << 32);
    for (int64_t gI = tid; gI < ngroups; gI += nth) {
        const int64_t e0 = gI << 2;
        int64_t bkt = 0, rem = e0;
        if (bucket > 0) {
            if (small) { bkt = (uint32_t)e0 / (uint32_t)bucket; rem = e0 - bkt * bucket; }
            else { bkt = e0 / bucket; rem = e0 - bkt * bucket; }
        }
        uint32_t bits;
        const bool full = e0 + 4 <= n;
        if (BITS == 8) {
            if (full) bits = *(const uint32_t*)(packed + e0);
            else { bits = 0; for (int64_t c = 0; e0 + c < n; ++c) bits |= (uint32_t)packed[e0 + c] << (8 * c); }
        } else if (BITS == 4) {
            if (full) bits = *(const uint16_t*)(packed + (e0 >> 1));
            else { bits = packed[e0 >> 1]; if (e0 + 2 < n) bits |= (uint32_t)packed[(e0 >> 1) + 1] << 8; }
        } else if (BITS == 2) {
            bits = packed[e0 >> 2];
        } else {
            bits = (uint32_t)packed[e0 >> 3] >> (e0 & 4);
        }
        float out[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            int64_t bc = bkt;
            if (bucket > 0 && rem + c >= bucket) bc = bkt + (rem + c) / bucket;   // a group may straddle buckets (bucket < 4: several)
            const float a = alpha[bc], b = beta[bc];
            const float r = (float)((bits >> (c * BITS)) & MASK);
            float w = r / sm1;                         // same three ops as the tail of qdq()
            float v = w * a;
            v = v + b;
            out[c] = v + 0.0f;
        }
        if (full && y_vec) {
            f4 o = {out[0], out[1], out[2], out[3]};
            __builtin_nontemporal_store(o, (f4*)(y + e0));
        } else {
            for (int c = 0; c < 4 && e0 + c < n; ++c) y[e0 + c] = out[c];
        }
    }
}

// histogram of uint8 symbols, any k <= 256, with INTEGER LDS atomics (ds_add_u32 without return) on a [k + 1][32]
// table shared by the block: column = lane mod 32, so the 32 lanes the LDS serves per cycle hit 32 different banks
// whatever their symbols are, and two lanes (or waves) that meet on one counter are resolved by the LDS itself -- no
// read-modify-write in registers, no duplicate merging, 3 VALU operations per symbol, 4.1-33 KiB of LDS per block.
template <int U>
__global__ __launch_bounds__(256) void k_hist_atomic(const uint8_t* idx, int64_t n, int k, unsigned long long* hist,
                                                     unsigned long long* partial /* [k][gridDim.x] or null */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_lds[];
    uint32_t* cnt = (uint32_t*)hist_lds;                                   // [k + 1][32], row k = dummy (symbols >= k)
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * 256;
    for (int j = threadIdx.x; j < (k + 1) * 32; j += 256) cnt[j] = 0;
    __syncthreads();
    uint32_t* col = cnt + (threadIdx.x & 31);
    const uint32_t kk = (uint32_t)k;
    auto bump = [&](uint32_t sy) { __hip_atomic_fetch_add(col + (sy < kk ? sy : kk) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto bump_word = [&](uint32_t v) { bump(v & 255u); bump((v >> 8) & 255u); bump((v >> 16) & 255u); bump(v >> 24); };
    // bytes before the first 16-byte boundary one by one, then 16-byte loads, then the last < 16 bytes
    int64_t head = (int64_t)((16 - (((uintptr_t)idx) & 15)) & 15);
    if (head > n) head = n;
    for (int64_t i = tid; i < head; i += nth) bump(idx[i]);
    int64_t done = head;
    {
        const uint8_t* body = idx + head;
        const int64_t n16 = (n - head) >> 4;
        const int64_t per_round = (int64_t)U * nth;
        const int64_t rounds = n16 / per_round;
        const u4* base = (const u4*)body + tid;
        auto fetch = [&](u4 (&w)[U], int64_t r) {
            const int64_t rr = r < rounds ? r : rounds - 1;
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(base + rr * per_round + (int64_t)u * nth);
        };
        auto count = [&](const u4 (&w)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) { bump_word(w[u].x); bump_word(w[u].y); bump_word(w[u].z); bump_word(w[u].w); }
        };
        // whole rounds (every lane U valid loads; the count is grid-uniform) run double-buffered: the loads of round r + 1 are
        // in flight while round r is counted; always issued (round index clamped: the last round is fetched twice, from L2) so
        // that no load sits behind a branch
        if (rounds > 0) {
            u4 a[U], b[U];
            fetch(a, 0);
            for (int64_t r = 0; r < rounds; r += 2) {
                fetch(b, r + 1);
                __builtin_amdgcn_sched_barrier(0);
                count(a);
                __builtin_amdgcn_sched_barrier(0);
                fetch(a, r + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (r + 1 < rounds) count(b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int64_t i = rounds * per_round + tid; i < n16; i += nth) {
            const u4 w = __builtin_nontemporal_load((const u4*)body + i);
            bump_word(w.x); bump_word(w.y); bump_word(w.z); bump_word(w.w);
        }
        done = head + (n16 << 4);
    }
    for (int64_t i = done + tid; i < n; i += nth) bump(idx[i]);
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        unsigned long long total = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) total += cnt[j * 32 + ((c + j) & 31)];
        if (partial) partial[(size_t)j * gridDim.x + blockIdx.x] = total;   // plain store: k_hist_fold sums the row
        else if (total) atomicAdd(&hist[j], total);
    }
}

// hist[j] (+)= sum of the per-block totals of bin j: one block per bin, fixed order (deterministic, no atomics, and hist
// needs no zeroing launch)
__global__ __launch_bounds__(256) void k_hist_fold(const unsigned long long* partial, int blocks, unsigned long long* hist,
                                                   int accumulate) {
    __shared__ unsigned long long wsum[4];
    const int j = blockIdx.x;
    unsigned long long t = 0;
    for (int b = threadIdx.x; b < blocks; b += 256) t += partial[(size_t)j * blocks + b];
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) t += __shfl_xor(t, sft);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) hist[j] = (accumulate ? hist[j] : 0ull) + wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// Histograms of symbols that are COMPUTED from wider elements, for the Huffman accounting of quantization/help_functions.py:
// 175-232 run on the device:
//   SRC_DIGITIZE  fp32 values v -> c = #{ j < m : edges[j] <= v } in 0..m, NaN -> m: np.digitize(v, edges) for increasing
//                 float64 edges, compared in float64 as numpy does after promoting the float32 data (:216-218);
//   SRC_I64       int64 symbols (the indices nonUniformQuantization returns, :220-221); anything outside [0, nrows - 1) is
//                 counted in the LAST row, so the caller can tell that the table was too small.
// Same counting scheme as k_hist_atomic ([nrows][32] uint32 in LDS, integer atomics, column = lane mod 32) without its
// software pipeline: 4 / 8 bytes per symbol instead of 1, so the HBM stream, not the LDS atomics, is what bounds it.
// Per-block totals go to partial[row][block]; k_hist_fold sums them.
enum { SRC_DIGITIZE = 0, SRC_I64 = 1 };

// np.digitize compares the float32 data, promoted to float64, with float64 edges.  Promotion is exact and monotonic, so
//     edges[j] <= (double)v   <=>   v >= thr[j],   thr[j] = the smallest float32 whose promotion is >= edges[j]
// and the whole search runs in float32 against a table every block derives from the edges once: no float64 arithmetic per
// element (the float64 form bounded both digitizing kernels by VALU, not by HBM).  The table holds PAIRS,
//     P[i] = (T[i], T[i + 1]),  T[0] = -inf, T[i] = thr[i - 1], T[m + 1] = +inf,   i = 0 .. m
// so that one 8-byte LDS read decides whether a candidate count c is the answer: T[c] <= v < T[c + 1].
__device__ __forceinline__ float digitize_threshold(double e) {
    float f = (float)e;                                     // round to nearest; then up if that fell below e
    if ((double)f < e) f = __uint_as_float(f == 0.0f ? 1u : (f > 0.0f ? __float_as_uint(f) + 1u : __float_as_uint(f) - 1u));
    return f;                                               // (f = -inf < e gives -FLT_MAX; e above FLT_MAX gives +inf)
}
__device__ __forceinline__ void digitize_table(float2* P, const double* edges_g, int m) {   // all 256 threads; caller syncs
    for (int i = threadIdx.x; i <= m; i += 256) {
        const float lo = i == 0 ? -INFINITY : digitize_threshold(edges_g[i - 1]);
        const float hi = i == m ? INFINITY : digitize_threshold(edges_g[i]);
        P[i] = make_float2(lo, hi);
    }
}
struct DigitizeGuess { float inv, off, top; };
__device__ __forceinline__ DigitizeGuess digitize_guess(const float2* P, int m) {           // after the table is complete
    DigitizeGuess G;
    const float t0 = P[0].y;                                // thr[0]
    const float span = P[m].x - t0;                         // thr[m - 1] - thr[0]
    G.inv = (m > 1 && span > 0.0f && span < INFINITY) ? (float)(m - 1) / span : 0.0f;
    G.off = 1.0f - t0 * G.inv;                              // candidate count = trunc((v - thr[0]) inv + 1), clamped to [0, m]
    G.top = (float)m;
    return G;
}
// Per element on the common path: one FMA, one median, one conversion, one 8-byte LDS read, two compares.  (The digitizing
// kernels are bound by VALU issue and LDS latency, not by HBM, at 4 B per element.)
__device__ __forceinline__ int digitize_candidate(const DigitizeGuess& G, float v) {
    return (int)__builtin_amdgcn_fmed3f(__builtin_fmaf(v, G.inv, G.off), 0.0f, G.top);      // in [0, m] for every v (NaN -> 0)
}
__device__ __noinline__ uint32_t digitize_walk(const float2* P, int m, float v, int c) {    // the rare path: kept out of line
    if (v != v) return (uint32_t)m;                        // NaN sorts last (numpy's searchsorted)
    while (c < m && P[c].y <= v) ++c;
    while (c > 0 && P[c].x > v) --c;
    return (uint32_t)c;
}
__device__ __forceinline__ uint32_t digitize_count(const float2* P, int m, const DigitizeGuess& G, float v) {
    // candidate from the mean spacing -- the answer right away for evenly spaced edges, which is what the caller has --
    // checked with one LDS read, walked to the exact place when it is not.  A NaN fails the check whatever the candidate is.
    const int c = digitize_candidate(G, v);
    const float2 p = P[c];
    if (__builtin_expect(!(p.x <= v && v < p.y), 0)) return digitize_walk(P, m, v, c);
    return (uint32_t)c;
}
// N elements at once: all candidates, all table reads, ONE branch for the lot (element by element, every check waits for
// its own LDS read and costs a branch: 60 instead of 4x us per 64 Mi elements).
template <int N>
__device__ __forceinline__ void digitize_many(const float2* P, int m, const DigitizeGuess& G, const float (&v)[N], uint32_t (&c)[N]) {
    float2 p[N];
#pragma unroll
    for (int k = 0; k < N; ++k) c[k] = (uint32_t)digitize_candidate(G, v[k]);
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = P[c[k]];
    bool bad = false;
#pragma unroll
    for (int k = 0; k < N; ++k) bad |= !(p[k].x <= v[k] && v[k] < p[k].y);
    if (__builtin_expect(bad, 0)) {
#pragma unroll 1
        for (int k = 0; k < N; ++k) c[k] = digitize_walk(P, m, v[k], (int)c[k]);
    }
}

template <int SRC>
__global__ __launch_bounds__(256) void k_hist_sym(const void* data, int64_t n, int nrows, const double* edges_g, int m,
                                                  unsigned long long* partial /* [nrows][gridDim.x] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_lds[];
    float2* P = (float2*)hist_lds;                                            // [m + 1] (SRC_DIGITIZE): digitize_table
    uint32_t* cnt = (uint32_t*)(hist_lds + (SRC == SRC_DIGITIZE ? (size_t)(m + 1) * sizeof(float2) : 0));   // [nrows][32]
    typedef typename std::conditional<SRC == SRC_DIGITIZE, float, long long>::type T;
    constexpr int E = 16 / (int)sizeof(T);                                    // symbols per 16-byte load
    const T* src = (const T*)data;
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * 256;
    for (int j = threadIdx.x; j < nrows * 32; j += 256) cnt[j] = 0;
    if (SRC == SRC_DIGITIZE) digitize_table(P, edges_g, m);
    __syncthreads();
    DigitizeGuess G = {0.0f, 0.0f, 0.0f};
    if (SRC == SRC_DIGITIZE) G = digitize_guess(P, m);
    uint32_t* col = cnt + (threadIdx.x & 31);
    const uint32_t last = (uint32_t)(nrows - 1);
    auto bump = [&](T v) {
        uint32_t sy;
        if constexpr (SRC == SRC_DIGITIZE) sy = digitize_count(P, m, G, v);
        else sy = (v < 0 || v >= (long long)last) ? last : (uint32_t)v;
        __hip_atomic_fetch_add(col + (sy < last ? sy : last) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    int64_t head = (int64_t)(((16 - (((uintptr_t)src) & 15)) & 15) / sizeof(T));   // elements before the first 16-byte boundary
    if (head > n) head = n;
    for (int64_t i = tid; i < head; i += nth) bump(src[i]);
    const int64_t nvec = (n - head) / E;
    typedef T vecT __attribute__((ext_vector_type(E)));
    const vecT* body = (const vecT*)(src + head);
    constexpr int U = 4;                                                      // loads in flight per lane
    int64_t i = tid;
    for (; i + (int64_t)(U - 1) * nth < nvec; i += (int64_t)U * nth) {
        vecT w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(body + i + (int64_t)u * nth);
        if constexpr (SRC == SRC_DIGITIZE) {
            float v[U * E];
            uint32_t sy[U * E];
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < E; ++c) v[u * E + c] = w[u][c];
            digitize_many<U * E>(P, m, G, v, sy);
#pragma unroll
            for (int k = 0; k < U * E; ++k)
                __hip_atomic_fetch_add(col + sy[k] * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int c = 0; c < E; ++c) bump(w[u][c]);
        }
    }
    for (; i < nvec; i += nth) {
        const vecT w = __builtin_nontemporal_load(body + i);
#pragma unroll
        for (int c = 0; c < E; ++c) bump(w[c]);
    }
    for (int64_t t = head + nvec * E + tid; t < n; t += nth) bump(src[t]);
    __syncthreads();
    for (int j = threadIdx.x; j < nrows; j += 256) {
        unsigned long long total = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) total += cnt[j * 32 + ((c + j) & 31)];
        partial[(size_t)j * gridDim.x + blockIdx.x] = total;
    }
}

// Level histogram of x in ONE pass (4 B read per element, nothing written but the counters): the quantize half of
// k_pack_vec -- a bucket in registers, min / max by DPP or wave reduction, level = rint((x - beta) / alpha * (s - 1)) --
// feeding a [levels][32] LDS counter table (the layout of k_hist_atomic's, without its out-of-range row) instead of a store.  What codec.level_histogram (the
// Huffman accounting of a model in its own packed format) needs: the level indices themselves never reach memory.
// Persistent grid: the table is zeroed and flushed once per block.  The short last bucket is done by block 0's first wave.
template <int LPB, int V>
__global__ __launch_bounds__(256) void k_level_hist_vec(const float* x, int64_t nvec, int64_t n, float sm1, int levels,
                                                        unsigned long long* partial /* [levels][gridDim.x] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_lds[];
    uint32_t* cnt = (uint32_t*)hist_lds;                                   // [levels][32]: 32 column copies of every counter (bank spread)
    constexpr int BPW = 64 / LPB;
    constexpr int ROW = LPB * V * 4;
    for (int j = threadIdx.x; j < levels * 32; j += 256) cnt[j] = 0;
    __syncthreads();
    uint32_t* col = cnt + (threadIdx.x & 31);
    const float top = (float)levels;
    // lev: an integer in [0, levels - 1], or NaN -- every element of a bucket that holds a NaN, and the +-inf elements (and, for
    // a -inf minimum, all elements) of a bucket that holds an infinity (alpha = inf: (x - beta) / alpha is 0 or NaN).  NaN counts
    // as level 0: what the uint8 level output of the quantize kernel stores for it ((uint8)(int)NaN = 0), so this one-pass form
    // and the write-levels-then-count form agree on such tensors too (tests/test_hip_huffman.py, the non-finite case).
    auto bump = [&](float lev) {
        const uint32_t li = (lev >= 0.0f && lev < top) ? (uint32_t)(int)lev : 0u;
        __hip_atomic_fetch_add(col + li * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPB, l = lane % LPB;
    const int64_t wave = uniform_wave_index();
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (nvec + BPW - 1) / BPW;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t bkt = t * BPW + sub;
        if (bkt >= nvec) continue;
        const int64_t e0 = bkt * ROW + (int64_t)l * 4;
        f4 v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = __builtin_nontemporal_load((const f4*)(x + e0) + j * LPB);
        float mn = pmin4(v[0]), mx = pmax4(v[0]);                          // NaN-propagating, as torch's min / max
#pragma unroll
        for (int j = 1; j < V; ++j) { mn = pmin(mn, pmin4(v[j])); mx = pmax(mx, pmax4(v[j])); }
        if (LPB == 16) { mn = row16_min(mn); mx = row16_max(mx); } else { mn = wave_min(mn); mx = wave_max(mx); }
        float a, b;
        alpha_beta(mn, mx, a, b);
        // only the LEVEL of u is consumed: the bucket-invariant division (qd_common.h) is exact here, taken when every bucket
        // of the wave is in its proven range (wave-uniform choice, as in the quantize kernels)
        const bool fast = !__any(!fastdiv_ok(a));
        const float ry = 1.0f / a;
        if (fast) {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float lev[4];
                qdq<true>(v[j].x, a, b, sm1, 0.0f, lev[0], ry);
                qdq<true>(v[j].y, a, b, sm1, 0.0f, lev[1], ry);
                qdq<true>(v[j].z, a, b, sm1, 0.0f, lev[2], ry);
                qdq<true>(v[j].w, a, b, sm1, 0.0f, lev[3], ry);
                bump(lev[0]); bump(lev[1]); bump(lev[2]); bump(lev[3]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float lev[4];
                qdq(v[j].x, a, b, sm1, 0.0f, lev[0]);
                qdq(v[j].y, a, b, sm1, 0.0f, lev[1]);
                qdq(v[j].z, a, b, sm1, 0.0f, lev[2]);
                qdq(v[j].w, a, b, sm1, 0.0f, lev[3]);
                bump(lev[0]); bump(lev[1]); bump(lev[2]); bump(lev[3]);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64 && nvec * ROW < n) {           // the short last bucket [nvec * ROW, n)
        const int64_t lo = nvec * ROW;
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int64_t i = lo + lane; i < n; i += 64) { const float v = x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); nan |= (v != v); }
        mn = wave_min(mn); mx = wave_max(mx);
        if (group_any<64>(nan)) { mn = NAN; mx = NAN; }
        float a, b;
        alpha_beta(mn, mx, a, b);
        for (int64_t i = lo + lane; i < n; i += 64) { float lev; qdq(x[i], a, b, sm1, 0.0f, lev); bump(lev); }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < levels; j += 256) {
        unsigned long long total = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) total += cnt[j * 32 + ((c + j) & 31)];
        partial[(size_t)j * gridDim.x + blockIdx.x] = total;
    }
}

// The Huffman accounting's re-scale + digitize + count in ONE pass (ref: quantization/help_functions.py:215-223): for every
// bucket of the QUANTIZED tensor q, u = (q - min) / (max - min) exactly as scale_down computes it (alpha_beta + the IEEE
// quotient; the bucket-invariant form where scale_fast_ok proves it equal), c = #{ j < m : edges[j] <= (double)u } as
// np.digitize does (digitize_count: in float32 against the promoted thresholds, exactly), counted in a [m + 1][32] LDS table.  4 B read per element, nothing written but the counters -- the
// two-kernel form (qd_scale_down_f32, then qd_digitize_histogram_f32 over its output) moves 12.  Same loop shape as
// k_level_hist_vec; the short last bucket is done by block 0's first wave (scale_down pads it with its last element, which
// changes neither the minimum nor the maximum, and the padding is not counted: :216 cuts it off).
template <int LPB, int V>
__global__ __launch_bounds__(256) void k_scale_digitize_hist_vec(const float* q, int64_t nvec, int64_t n, const double* edges_g, int m,
                                                                 unsigned long long* partial /* [m + 1][gridDim.x] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_lds[];
    float2* P = (float2*)hist_lds;                                            // [m + 1]: digitize_table
    uint32_t* cnt = (uint32_t*)(hist_lds + (size_t)(m + 1) * sizeof(float2));  // [m + 1][32]
    constexpr int BPW = 64 / LPB;
    constexpr int ROW = LPB * V * 4;
    const int nrows = m + 1;
    for (int j = threadIdx.x; j < nrows * 32; j += 256) cnt[j] = 0;
    digitize_table(P, edges_g, m);
    __syncthreads();
    const DigitizeGuess G = digitize_guess(P, m);
    uint32_t* col = cnt + (threadIdx.x & 31);
    auto bump = [&](float u) {
        const uint32_t c = digitize_count(P, m, G, u);
        __hip_atomic_fetch_add(col + c * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPB, l = lane % LPB;
    const int64_t wave = uniform_wave_index();
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (nvec + BPW - 1) / BPW;
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t bkt = t * BPW + sub;
        if (bkt >= nvec) continue;
        const int64_t el0 = bkt * ROW + (int64_t)l * 4;
        f4 v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) v[j] = __builtin_nontemporal_load((const f4*)(q + el0) + j * LPB);
        float mn = pmin4(v[0]), mx = pmax4(v[0]);                          // NaN-propagating, as torch's min / max
#pragma unroll
        for (int j = 1; j < V; ++j) { mn = pmin(mn, pmin4(v[j])); mx = pmax(mx, pmax4(v[j])); }
        if (LPB == 16) { mn = row16_min(mn); mx = row16_max(mx); } else { mn = wave_min(mn); mx = wave_max(mx); }
        float a, b;
        alpha_beta(mn, mx, a, b);
        unsigned key = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            key = min(key, min(scale_numerator_key(v[j].x, b), scale_numerator_key(v[j].y, b)));
            key = min(key, min(scale_numerator_key(v[j].z, b), scale_numerator_key(v[j].w, b)));
        }
        float u[V * 4];
        if (!__any(!scale_fast_ok(a, key))) {                               // wave-uniform choice, as in the scale_down kernel
            const float ry = 1.0f / a;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                u[4 * j + 0] = div_alpha<true>(v[j].x - b, a, ry); u[4 * j + 1] = div_alpha<true>(v[j].y - b, a, ry);
                u[4 * j + 2] = div_alpha<true>(v[j].z - b, a, ry); u[4 * j + 3] = div_alpha<true>(v[j].w - b, a, ry);
            }
        } else {
#pragma unroll
            for (int j = 0; j < V; ++j) {
                u[4 * j + 0] = (v[j].x - b) / a; u[4 * j + 1] = (v[j].y - b) / a;
                u[4 * j + 2] = (v[j].z - b) / a; u[4 * j + 3] = (v[j].w - b) / a;
            }
        }
        uint32_t c[V * 4];
        digitize_many<V * 4>(P, m, G, u, c);
#pragma unroll
        for (int k = 0; k < V * 4; ++k) __hip_atomic_fetch_add(col + c[k] * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if (blockIdx.x == 0 && threadIdx.x < 64 && nvec * ROW < n) {           // the short last bucket [nvec * ROW, n)
        const int64_t lo = nvec * ROW;
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int64_t i = lo + lane; i < n; i += 64) { const float x = q[i]; mn = fminf(mn, x); mx = fmaxf(mx, x); nan |= (x != x); }
        mn = wave_min(mn); mx = wave_max(mx);
        if (group_any<64>(nan)) { mn = NAN; mx = NAN; }
        float a, b;
        alpha_beta(mn, mx, a, b);
        for (int64_t i = lo + lane; i < n; i += 64) bump((q[i] - b) / a);
    }
    __syncthreads();
    for (int j = threadIdx.x; j < nrows; j += 256) {
        unsigned long long total = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) total += cnt[j * 32 + ((c + j) & 31)];
        partial[(size_t)j * gridDim.x + blockIdx.x] = total;
    }
}

__global__ __launch_bounds__(256) void k_zero_u64(unsigned long long* p, int n) {
    for (int i = threadIdx.x; i < n; i += 256) p[i] = 0ull;
}

inline int blocks_for(int64_t items, int per_block, int cap) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    return (int)(b < cap ? b : cap);
}

}  // namespace

extern "C" {

int64_t qd_packed_bytes(int64_t n, int bits) { return n < 0 || bits < 1 ? -1 : (n * bits + 7) / 8; }

int qd_pack_uniform_f32(const float* x, int64_t n, int64_t bucket, int levels, int bits, uint8_t* packed, float* alpha,
                        float* beta, void* stream) {
    if (n < 0 || levels < 2 || (bits != 1 && bits != 2 && bits != 4 && bits != 8) || levels > (1 << bits))
        return QD_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!x || !packed)) return QD_ERR_INVALID_ARGUMENT;      // alpha / beta: optional outputs
    if (bucket != 64 && bucket != 128 && bucket != 256 && bucket != 512 && bucket != 1024 && bucket != 2048)
        return QD_ERR_UNSUPPORTED;                       // the codec is defined for the vector bucket sizes
    if ((((uintptr_t)x) & 15) || (((uintptr_t)packed) & 3)) return QD_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const float sm1 = (float)(levels - 1);
    const int64_t nfull = n / bucket;
#define QD_PACK(LPB, V)                                                                                              \
    {                                                                                                                \
        const int64_t tiles = (nfull + (64 / LPB) - 1) / (64 / LPB);                                                 \
        const int blocks = blocks_for(tiles, 4, 1 << 20);                                                            \
        if (bits == 8) hipLaunchKernelGGL((k_pack_vec<LPB, V, 8>), dim3(blocks), dim3(256), 0, st, x, packed, alpha, beta, nfull, sm1); \
        else if (bits == 4) hipLaunchKernelGGL((k_pack_vec<LPB, V, 4>), dim3(blocks), dim3(256), 0, st, x, packed, alpha, beta, nfull, sm1); \
        else if (bits == 2) hipLaunchKernelGGL((k_pack_vec<LPB, V, 2>), dim3(blocks), dim3(256), 0, st, x, packed, alpha, beta, nfull, sm1); \
        else hipLaunchKernelGGL((k_pack_vec<LPB, V, 1>), dim3(blocks), dim3(256), 0, st, x, packed, alpha, beta, nfull, sm1); \
    }
    if (nfull > 0) {
        switch (bucket) {
            case 64: QD_PACK(16, 1) break;
            case 128: QD_PACK(16, 2) break;
            case 256: QD_PACK(16, 4) break;
            case 512: QD_PACK(64, 2) break;
            case 1024: QD_PACK(64, 4) break;
            default: QD_PACK(64, 8) break;
        }
    }
#undef QD_PACK
    if (nfull * bucket < n) {
        const int64_t lo = nfull * bucket;
        if (bits == 8) hipLaunchKernelGGL(k_pack_tail<8>, dim3(1), dim3(64), 0, st, x, packed, alpha, beta, lo, n, nfull, sm1);
        else if (bits == 4) hipLaunchKernelGGL(k_pack_tail<4>, dim3(1), dim3(64), 0, st, x, packed, alpha, beta, lo, n, nfull, sm1);
        else if (bits == 2) hipLaunchKernelGGL(k_pack_tail<2>, dim3(1), dim3(64), 0, st, x, packed, alpha, beta, lo, n, nfull, sm1);
        else hipLaunchKernelGGL(k_pack_tail<1>, dim3(1), dim3(64), 0, st, x, packed, alpha, beta, lo, n, nfull, sm1);
    }
    return (int)hipGetLastError();
}

int qd_pack_levels_u8(const uint8_t* levels_idx, int64_t n, int bits, uint8_t* packed, void* stream) {
    if (n < 0 || (bits != 1 && bits != 2 && bits != 4 && bits != 8) || (n > 0 && (!levels_idx || !packed)))
        return QD_ERR_INVALID_ARGUMENT;
    if (n == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    const int blocks = blocks_for(n / (32 / bits) + 1, 256 * 4, 1 << 20);
    if (bits == 8) hipLaunchKernelGGL(k_pack_levels<8>, dim3(blocks), dim3(256), 0, st, levels_idx, n, packed);
    else if (bits == 4) hipLaunchKernelGGL(k_pack_levels<4>, dim3(blocks), dim3(256), 0, st, levels_idx, n, packed);
    else if (bits == 2) hipLaunchKernelGGL(k_pack_levels<2>, dim3(blocks), dim3(256), 0, st, levels_idx, n, packed);
    else hipLaunchKernelGGL(k_pack_levels<1>, dim3(blocks), dim3(256), 0, st, levels_idx, n, packed);
    return (int)hipGetLastError();
}

int qd_unpack_uniform_f32(const uint8_t* packed, int64_t n, int64_t bucket, int levels, int bits, const float* alpha,
                          const float* beta, float* y, void* stream) {
    if (n < 0 || levels < 2 || (bits != 1 && bits != 2 && bits != 4 && bits != 8) || levels > (1 << bits))
        return QD_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!packed || !y || !alpha || !beta)) return QD_ERR_INVALID_ARGUMENT;
    if (bucket < 0) return QD_ERR_INVALID_ARGUMENT;
    if ((((uintptr_t)packed) & 3)) return QD_ERR_UNSUPPORTED;
    if (n == 0) return 0;
    if (bucket < 8 || (bucket & (bucket - 1)) || n < bucket) {
        // any other bucket size; 0 (or a tensor shorter than one bucket) = one alpha / beta for the whole tensor
        hipStream_t st2 = (hipStream_t)stream;
        const float sm1b = (float)(levels - 1);
        const int64_t bk = (bucket == 0 || n < bucket) ? 0 : bucket;
        const int blocks2 = blocks_for((n + 3) / 4, 256 * 4, 1 << 20);
        if (bits == 8) hipLaunchKernelGGL(k_unpack_any<8>, dim3(blocks2), dim3(256), 0, st2, packed, y, alpha, beta, n, bk, sm1b);
        else if (bits == 4) hipLaunchKernelGGL(k_unpack_any<4>, dim3(blocks2), dim3(256), 0, st2, packed, y, alpha, beta, n, bk, sm1b);
        else if (bits == 2) hipLaunchKernelGGL(k_unpack_any<2>, dim3(blocks2), dim3(256), 0, st2, packed, y, alpha, beta, n, bk, sm1b);
        else hipLaunchKernelGGL(k_unpack_any<1>, dim3(blocks2), dim3(256), 0, st2, packed, y, alpha, beta, n, bk, sm1b);
        return (int)hipGetLastError();
    }
    int row_shift = 0;
    while (((int64_t)1 << row_shift) < bucket) ++row_shift;
    hipStream_t st = (hipStream_t)stream;
    const float sm1 = (float)(levels - 1);
    const int blocks = blocks_for((n + 3) / 4, 256 * 4, 1 << 20);
    if (bits == 8) hipLaunchKernelGGL(k_unpack<8>, dim3(blocks), dim3(256), 0, st, packed, y, alpha, beta, n, row_shift, sm1);
    else if (bits == 4) hipLaunchKernelGGL(k_unpack<4>, dim3(blocks), dim3(256), 0, st, packed, y, alpha, beta, n, row_shift, sm1);
    else if (bits == 2) hipLaunchKernelGGL(k_unpack<2>, dim3(blocks), dim3(256), 0, st, packed, y, alpha, beta, n, row_shift, sm1);
    else hipLaunchKernelGGL(k_unpack<1>, dim3(blocks), dim3(256), 0, st, packed, y, alpha, beta, n, row_shift, sm1);
    return (int)hipGetLastError();
}

int qd_histogram_u8_ws(const uint8_t* idx, int64_t n, int k, uint64_t* hist, void* workspace, size_t workspace_bytes,
                       void* stream) {
    if (n < 0 || k < 1 || k > 256 || !hist || (n > 0 && !idx)) return QD_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const int cus = device_cus();
    const size_t lds = (size_t)(k + 1) * 32 * sizeof(uint32_t);
    // Measured at 64 Mi symbols (tools/tune_r2.py, docs/history/profiles/r02_tune_kernels.txt).  Private-column tables with the four
    // symbols of a word merged in registers (round 1 / early round 2): 35 us (k <= 64), 49 us (k = 256); register counters
    // for k <= 16: 21.5-23.5 us, unchanged by software-pipelining the loads.  This kernel with global atomics at the end of
    // each block, 1 / 2 / 4 blocks per CU: k = 16 19.4 / 18.9 / 24.8, k = 64 22.0 / 23.9 / 33.8, k = 256 22.7 / 24.7 / 34.9 us
    // (256-1024 same-address atomics per bin at the end).  With per-block totals + fold launch, 1 / 2 / 3 / 4 blocks per CU:
    // k = 16 17.9 / 16.2 / 18.0 / 19.4, k = 64 18.1 / 16.4 / 18.6 / 20.2, k = 256 19.3 / 18.8 / 21.8 / 23.7 us; 2 / 4 / 8
    // loads in flight per lane (2 blocks per CU): k = 16 15.5 / 16.2 / 18.9, k = 256 18.1 / 18.8 / 21.1 us.
    if (workspace && n > 0 && (((uintptr_t)workspace) & 7) == 0) {
        // per-block totals in the workspace ([k][blocks] uint64), summed per bin by a second launch: no zeroing launch, no
        // same-address global atomics at the end of every block
        int blocks = blocks_for(n, 256 * 16 * 2, cus * 2);
        const size_t room = workspace_bytes / ((size_t)k * sizeof(unsigned long long));
        if ((size_t)blocks > room) blocks = (int)room;
        if (blocks >= 1) {
            const int64_t slice = (int64_t)blocks << 31;           // a uint32 counter holds what ONE block counts in one launch
            for (int64_t off = 0; off < n; off += slice) {
                const int64_t len = n - off < slice ? n - off : slice;
                hipLaunchKernelGGL((k_hist_atomic<2>), dim3(blocks), dim3(256), lds, st, idx + off, len, k,
                                   (unsigned long long*)hist, (unsigned long long*)workspace);
                hipLaunchKernelGGL(k_hist_fold, dim3(k), dim3(256), 0, st, (const unsigned long long*)workspace, blocks,
                                   (unsigned long long*)hist, off > 0 ? 1 : 0);
            }
            return (int)hipGetLastError();
        }
    }
    // no workspace: zero the histogram, every block adds its totals with one global atomic per bin (one block per CU:
    // few, long-lived blocks keep the same-address atomics at the end short)
    hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(256), 0, st, (unsigned long long*)hist, k);   // (a memset node costs more)
    if (n == 0) return (int)hipGetLastError();
    const int blocks = blocks_for(n, 256 * 16 * 4, cus);
    const int64_t slice = (int64_t)blocks << 31;
    for (int64_t off = 0; off < n; off += slice) {
        const int64_t len = n - off < slice ? n - off : slice;
        hipLaunchKernelGGL((k_hist_atomic<4>), dim3(blocks), dim3(256), lds, st, idx + off, len, k, (unsigned long long*)hist,
                           (unsigned long long*)nullptr);
    }
    return (int)hipGetLastError();
}

}  // extern "C"

namespace {
template <int SRC>
int launch_hist_sym(const void* data, int64_t n, int nrows, const double* edges, int m, uint64_t* hist, void* workspace,
                    size_t workspace_bytes, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(256), 0, st, (unsigned long long*)hist, nrows);
        return (int)hipGetLastError();
    }
    if (!workspace || (((uintptr_t)workspace) & 7)) return QD_ERR_WORKSPACE_TOO_SMALL;
    const int cus = device_cus();
    int blocks = blocks_for(n, 256 * 16, cus * 4);
    const size_t room = workspace_bytes / ((size_t)nrows * sizeof(unsigned long long));
    if ((size_t)blocks > room) blocks = (int)room;
    if (blocks < 1) return QD_ERR_WORKSPACE_TOO_SMALL;
    const size_t lds = (size_t)nrows * 32 * sizeof(uint32_t) + (SRC == SRC_DIGITIZE ? (size_t)(m + 1) * sizeof(float2) : 0);
    const int64_t slice = (int64_t)blocks << 31;               // a uint32 counter holds what ONE block counts in one launch
    const size_t esz = SRC == SRC_DIGITIZE ? sizeof(float) : sizeof(long long);
    for (int64_t off = 0; off < n; off += slice) {
        const int64_t len = n - off < slice ? n - off : slice;
        hipLaunchKernelGGL((k_hist_sym<SRC>), dim3(blocks), dim3(256), lds, st, (const void*)((const char*)data + off * esz), len, nrows,
                           edges, m, (unsigned long long*)workspace);
        hipLaunchKernelGGL(k_hist_fold, dim3(nrows), dim3(256), 0, st, (const unsigned long long*)workspace, blocks,
                           (unsigned long long*)hist, off > 0 ? 1 : 0);
    }
    return (int)hipGetLastError();
}
}  // namespace

extern "C" {

int qd_level_histogram_f32(const float* x, int64_t n, int64_t bucket, int levels, uint64_t* hist, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (n < 0 || levels < 2 || levels > 256 || !hist || (n > 0 && !x)) return QD_ERR_INVALID_ARGUMENT;
    if (bucket != 64 && bucket != 128 && bucket != 256 && bucket != 512 && bucket != 1024 && bucket != 2048) return QD_ERR_UNSUPPORTED;
    if (((uintptr_t)x) & 15) return QD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(256), 0, st, (unsigned long long*)hist, levels);
        return (int)hipGetLastError();
    }
    if (!workspace || (((uintptr_t)workspace) & 7)) return QD_ERR_WORKSPACE_TOO_SMALL;
    const int64_t nfull = n / bucket;
    const size_t lds = (size_t)levels * 32 * sizeof(uint32_t);
    const float sm1 = (float)(levels - 1);
#define QD_LH(LPB, V)                                                                                                 \
    {                                                                                                                 \
        const int64_t tiles = (nfull + (64 / LPB) - 1) / (64 / LPB);                                                  \
        int blocks = blocks_for(tiles > 0 ? tiles : 1, 4, device_cus() * 8);                                          \
        const size_t room = workspace_bytes / ((size_t)levels * sizeof(unsigned long long));                          \
        if ((size_t)blocks > room) blocks = (int)room;                                                                \
        if (blocks < 1) return QD_ERR_WORKSPACE_TOO_SMALL;                                                            \
        if (n > ((int64_t)blocks << 31)) return QD_ERR_UNSUPPORTED;        /* a uint32 counter per block and level */  \
        hipLaunchKernelGGL((k_level_hist_vec<LPB, V>), dim3(blocks), dim3(256), lds, st, x, nfull, n, sm1, levels,    \
                           (unsigned long long*)workspace);                                                           \
        hipLaunchKernelGGL(k_hist_fold, dim3(levels), dim3(256), 0, st, (const unsigned long long*)workspace, blocks, \
                           (unsigned long long*)hist, 0);                                                             \
    }
    switch (bucket) {
        case 64: QD_LH(16, 1) break;
        case 128: QD_LH(16, 2) break;
        case 256: QD_LH(16, 4) break;
        case 512: QD_LH(64, 2) break;
        case 1024: QD_LH(64, 4) break;
        default: QD_LH(64, 8) break;
    }
#undef QD_LH
    return (int)hipGetLastError();
}

int qd_scale_digitize_histogram_f32(const float* q, int64_t n, int64_t bucket, const double* edges, int m, uint64_t* hist,
                                    void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || m < 1 || m > 256 || !edges || !hist || (n > 0 && !q) || (((uintptr_t)edges) & 7)) return QD_ERR_INVALID_ARGUMENT;
    if (bucket != 64 && bucket != 128 && bucket != 256 && bucket != 512 && bucket != 1024 && bucket != 2048) return QD_ERR_UNSUPPORTED;
    if (((uintptr_t)q) & 15) return QD_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const int nrows = m + 1;
    if (n == 0) {
        hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(256), 0, st, (unsigned long long*)hist, nrows);
        return (int)hipGetLastError();
    }
    if (!workspace || (((uintptr_t)workspace) & 7)) return QD_ERR_WORKSPACE_TOO_SMALL;
    const int64_t nfull = n / bucket;
    const size_t lds = (size_t)nrows * 32 * sizeof(uint32_t) + (size_t)(m + 1) * sizeof(float2);
#define QD_SDH(LPB, V)                                                                                                  \
    {                                                                                                                   \
        const int64_t tiles = (nfull + (64 / LPB) - 1) / (64 / LPB);                                                    \
        int blocks = blocks_for(tiles > 0 ? tiles : 1, 4, device_cus() * 8);                                            \
        const size_t room = workspace_bytes / ((size_t)nrows * sizeof(unsigned long long));                             \
        if ((size_t)blocks > room) blocks = (int)room;                                                                  \
        if (blocks < 1) return QD_ERR_WORKSPACE_TOO_SMALL;                                                              \
        if (n > ((int64_t)blocks << 31)) return QD_ERR_UNSUPPORTED;        /* a uint32 counter per block and row */      \
        hipLaunchKernelGGL((k_scale_digitize_hist_vec<LPB, V>), dim3(blocks), dim3(256), lds, st, q, nfull, n, edges, m, \
                           (unsigned long long*)workspace);                                                             \
        hipLaunchKernelGGL(k_hist_fold, dim3(nrows), dim3(256), 0, st, (const unsigned long long*)workspace, blocks,    \
                           (unsigned long long*)hist, 0);                                                               \
    }
    switch (bucket) {
        case 64: QD_SDH(16, 1) break;
        case 128: QD_SDH(16, 2) break;
        case 256: QD_SDH(16, 4) break;
        case 512: QD_SDH(64, 2) break;
        case 1024: QD_SDH(64, 4) break;
        default: QD_SDH(64, 8) break;
    }
#undef QD_SDH
    return (int)hipGetLastError();
}

int qd_digitize_histogram_f32(const float* v, int64_t n, const double* edges, int m, uint64_t* hist, void* workspace,
                              size_t workspace_bytes, void* stream) {
    if (n < 0 || m < 1 || m > 256 || !edges || !hist || (n > 0 && !v)) return QD_ERR_INVALID_ARGUMENT;
    if ((((uintptr_t)v) & 3) || (((uintptr_t)edges) & 7)) return QD_ERR_INVALID_ARGUMENT;
    return launch_hist_sym<SRC_DIGITIZE>(v, n, m + 1, edges, m, hist, workspace, workspace_bytes, stream);
}

int qd_histogram_i64(const int64_t* idx, int64_t n, int k, uint64_t* hist, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || k < 1 || k > 256 || !hist || (n > 0 && !idx)) return QD_ERR_INVALID_ARGUMENT;
    if (((uintptr_t)idx) & 7) return QD_ERR_INVALID_ARGUMENT;
    return launch_hist_sym<SRC_I64>(idx, n, k + 1, nullptr, 0, hist, workspace, workspace_bytes, stream);
}

int qd_histogram_u8(const uint8_t* idx, int64_t n, int k, uint64_t* hist, void* stream) {
    return qd_histogram_u8_ws(idx, n, k, hist, nullptr, 0, stream);
}

}  // extern "C"
