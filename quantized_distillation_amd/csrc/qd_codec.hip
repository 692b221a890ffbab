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
        if (l == 0) { alpha[bkt] = a; beta[bkt] = b; }
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
    if (lane == 0) { alpha[bkt] = a; beta[bkt] = b; }
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

// histogram of uint8 symbols: every lane counts into a private LDS column (k <= 64) -- plain
// read-increment-write, no atomics -- or into per-block LDS atomics (k <= 256); per-block totals
// go to the global uint64 histogram with one atomic per bin per block.
template <bool PRIVATE>
__global__ __launch_bounds__(256) void k_hist_u8(const uint8_t* idx, int64_t n, int k, unsigned long long* hist) {
    extern __shared__ __attribute__((aligned(16))) uint32_t cnt[];       // PRIVATE: [k][256], else [k]
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int cells = PRIVATE ? k * 256 : k;
    for (int j = threadIdx.x; j < cells; j += 256) cnt[j] = 0;
    __syncthreads();
    auto bump = [&](uint32_t s) {
        if (s < (uint32_t)k) {
            if (PRIVATE) cnt[s * 256 + threadIdx.x] += 1;
            else atomicAdd(&cnt[s], 1u);
        }
    };
    int64_t done = 0;
    if ((((uintptr_t)idx) & 15) == 0) {
        const int64_t n16 = n >> 4;
        typedef uint32_t u4 __attribute__((ext_vector_type(4)));
        for (int64_t i = tid; i < n16; i += nth) {
            const u4 w = __builtin_nontemporal_load((const u4*)idx + i);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const uint32_t v = c == 0 ? w.x : c == 1 ? w.y : c == 2 ? w.z : w.w;
                bump(v & 255); bump((v >> 8) & 255); bump((v >> 16) & 255); bump(v >> 24);
            }
        }
        done = n16 << 4;
    }
    for (int64_t i = done + tid; i < n; i += nth) bump(idx[i]);
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        unsigned long long total = 0;
        if (PRIVATE) { for (int c = 0; c < 256; ++c) total += cnt[j * 256 + ((c + j) & 255)]; }
        else total = cnt[j];
        if (total) atomicAdd(&hist[j], total);
    }
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
    if (n > 0 && (!x || !packed || !alpha || !beta)) return QD_ERR_INVALID_ARGUMENT;
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

int qd_unpack_uniform_f32(const uint8_t* packed, int64_t n, int64_t bucket, int levels, int bits, const float* alpha,
                          const float* beta, float* y, void* stream) {
    if (n < 0 || levels < 2 || (bits != 1 && bits != 2 && bits != 4 && bits != 8) || levels > (1 << bits))
        return QD_ERR_INVALID_ARGUMENT;
    if (n > 0 && (!packed || !y || !alpha || !beta)) return QD_ERR_INVALID_ARGUMENT;
    if (bucket < 8 || (bucket & (bucket - 1))) return QD_ERR_UNSUPPORTED;
    if ((((uintptr_t)packed) & 3)) return QD_ERR_UNSUPPORTED;
    if (n == 0) return 0;
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

int qd_histogram_u8(const uint8_t* idx, int64_t n, int k, uint64_t* hist, void* stream) {
    if (n < 0 || k < 1 || k > 256 || !hist || (n > 0 && !idx)) return QD_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(hist, 0, sizeof(uint64_t) * (size_t)k, st);
    if (n == 0) return (int)hipGetLastError();
    const int blocks = blocks_for(n, 256 * 16 * 4, 4096);
    if (k <= 64)
        hipLaunchKernelGGL(k_hist_u8<true>, dim3(blocks), dim3(256), (size_t)k * 256 * sizeof(uint32_t), st, idx, n, k,
                           (unsigned long long*)hist);
    else
        hipLaunchKernelGGL(k_hist_u8<false>, dim3(blocks), dim3(256), (size_t)k * sizeof(uint32_t), st, idx, n, k,
                           (unsigned long long*)hist);
    return (int)hipGetLastError();
}

}  // extern "C"
