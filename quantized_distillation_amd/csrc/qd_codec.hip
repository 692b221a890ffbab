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

// histogram of uint8 symbols.  Every lane counts into a private LDS column cnt[k][256] -- plain
// read-increment-write, no atomics -- of uint32 (k <= 64, 64 KiB) or uint16 (k <= 256, 128 KiB:
// a lane flushes before it can have seen 65535 symbols).  The four symbols of a 32-bit word are
// counted together: four independent LDS reads, duplicates resolved in registers (every symbol
// gets old + multiplicity, so equal addresses are written with equal values), four writes -- one
// LDS round trip per word instead of four dependent ones, which is what bounds the large tables
// that leave four waves per CU.  Per-block totals go to the global uint64 histogram with one
// atomic per bin per block; the grid is resident (as many blocks as fit the CUs at once).
template <typename CT, int U>
__global__ __launch_bounds__(256) void k_hist_u8(const uint8_t* idx, int64_t n, int k, unsigned long long* hist) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hist_lds[];
    CT* cnt = (CT*)hist_lds;                                               // [k + 1][256], row k = dummy
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * 256;
    CT* col = cnt + threadIdx.x;
    auto clear = [&]() {
        uint32_t* z = (uint32_t*)hist_lds;
        const int words = (k + 1) * 256 * (int)sizeof(CT) / 4;
        for (int j = threadIdx.x; j < words; j += 256) z[j] = 0;
        __syncthreads();
    };
    auto flush = [&]() {                     // 256 / kk threads per bin, rotated start, fixed shuffle fold
        __syncthreads();
        int per = 64;                        // threads per bin: a power of two within one wave, per * k <= 256
        while (per > 1 && per * k > 256) per >>= 1;
        const int span = 256 / per;          // columns each of them sums
        for (int t = threadIdx.x; t < k * per; t += 256) {
            const int j = t / per, q = t % per;
            unsigned long long total = 0;
            for (int c = 0; c < span; ++c) total += cnt[j * 256 + q * span + ((c + j) & (span - 1))];
            for (int sft = 1; sft < per; sft <<= 1) total += __shfl_xor(total, sft);
            if (q == 0 && total) atomicAdd(&hist[j], total);
        }
        __syncthreads();
    };
    // symbols >= k (never produced by the quantizer) land in a dummy row k that the fold ignores: no
    // predicated stores in the hot loop
    const uint32_t kk = (uint32_t)k;
    auto bump = [&](uint32_t s) { col[(s < kk ? s : kk) * 256] += 1; };
    auto bump_word = [&](uint32_t v) {
        uint32_t s0 = v & 255, s1 = (v >> 8) & 255, s2 = (v >> 16) & 255, s3 = v >> 24;
        s0 = s0 < kk ? s0 : kk; s1 = s1 < kk ? s1 : kk; s2 = s2 < kk ? s2 : kk; s3 = s3 < kk ? s3 : kk;
        CT* a0 = col + s0 * 256; CT* a1 = col + s1 * 256; CT* a2 = col + s2 * 256; CT* a3 = col + s3 * 256;
        const uint32_t c0 = *a0, c1 = *a1, c2 = *a2, c3 = *a3;
        const uint32_t e01 = (s0 == s1), e02 = (s0 == s2), e03 = (s0 == s3), e12 = (s1 == s2), e13 = (s1 == s3),
                       e23 = (s2 == s3);
        *a0 = (CT)(c0 + 1 + e01 + e02 + e03);
        *a1 = (CT)(c1 + 1 + e01 + e12 + e13);
        *a2 = (CT)(c2 + 1 + e02 + e12 + e23);
        *a3 = (CT)(c3 + 1 + e03 + e13 + e23);
    };
    clear();
    int64_t done = 0;
    if ((((uintptr_t)idx) & 15) == 0) {
        const int64_t n16 = n >> 4;
        // a uint16 column overflows after 65535 symbols: at most 4000 16-byte loads per lane between flushes
        const int64_t epoch = sizeof(CT) == 2 ? (int64_t)4000 * nth : n16 + nth;
        for (int64_t base = 0; base < n16; base += epoch) {
            const int64_t end = base + epoch < n16 ? base + epoch : n16;
            int64_t i = base + tid;
            for (; i + (int64_t)(U - 1) * nth < end; i += (int64_t)U * nth) {
                u4 w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load((const u4*)idx + i + (int64_t)u * nth);
                __builtin_amdgcn_sched_barrier(0);      // keep the U loads in flight together (not sunk to their uses)
#pragma unroll
                for (int u = 0; u < U; ++u) { bump_word(w[u].x); bump_word(w[u].y); bump_word(w[u].z); bump_word(w[u].w); }
            }
            for (; i < end; i += nth) {
                const u4 w = __builtin_nontemporal_load((const u4*)idx + i);
                bump_word(w.x); bump_word(w.y); bump_word(w.z); bump_word(w.w);
            }
            if (end < n16) { flush(); clear(); }
        }
        done = n16 << 4;
    }
    int tail = 0;
    for (int64_t b0 = done + (int64_t)blockIdx.x * 256; b0 < n; b0 += nth) {       // block-uniform trip count
        const int64_t i = b0 + threadIdx.x;
        if (i < n) bump(idx[i]);
        if (sizeof(CT) == 2 && ++tail == 60000) { tail = 0; flush(); clear(); }   // unaligned input only
    }
    flush();
}

// histogram of uint8 symbols with k <= 16 (4-bit and narrower quantization -- the BASELINE configurations): NO table
// at all.  Every lane counts into registers: a 64-bit accumulator of sixteen 4-bit fields takes `1 << 4 s` per symbol
// -- looked up per symbol PAIR in a 2 KiB LDS table -- (two words = 8 symbols at most per accumulator, so a field cannot overflow), the two accumulators of a 16-byte load are
// spread into two 64-bit accumulators of eight 8-bit fields each (even / odd symbols; 16 per field per load at most), and
// after U <= 15 loads those are added to sixteen 32-bit counters.  About 6.5 VALU operations per symbol and no LDS
// traffic, against ~20 operations and two LDS accesses per symbol of the private-column table (35 us for 64 Mi symbols).
// Symbols >= 16 cannot be represented (the shift would wrap): a 16-byte group that holds one (never produced by the
// quantizer) is counted symbol by symbol instead.  Symbols in [k, 16) are counted and dropped at the end.
// Blocks are 1024 lanes: every block ends with k global atomics on one cache line, so the grid is ONE block per CU -- 16
// waves, four per SIMD -- instead of many small ones.
template <int U, bool PF>
__global__ __launch_bounds__(1024) void k_hist_reg16(const uint8_t* idx, int64_t n, int k, unsigned long long* hist) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __shared__ unsigned long long wsum[16][16];
    const int64_t tid = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * 1024;
    uint32_t cnt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) cnt[j] = 0;
    const uint64_t M = 0x0F0F0F0F0F0F0F0Full;
    // (1 << 4 s0) + (1 << 4 s1) for the symbol PAIR in a byte s0 | s1 << 4: one 8-byte LDS read + one 64-bit add count two
    // symbols (variable 64-bit shifts are quarter rate on this chip: computing the increment cost as much as everything else)
    __shared__ unsigned long long pair_inc[256];
    if (threadIdx.x < 256) pair_inc[threadIdx.x] = (1ull << (4 * (threadIdx.x & 15))) + (1ull << (4 * (threadIdx.x >> 4)));
    __syncthreads();
    const char* lut = (const char*)pair_inc;
    auto nib2 = [&](uint32_t a, uint32_t b) -> uint64_t {          // eight symbols -> sixteen 4-bit fields (each <= 8)
        const uint32_t ta = a | (a >> 4), tb = b | (b >> 4);       // bytes 0 and 2: s0 | s1 << 4, s2 | s3 << 4
        uint64_t acc = *(const unsigned long long*)(lut + ((ta << 3) & 0x7F8u));
        acc += *(const unsigned long long*)(lut + ((ta >> 13) & 0x7F8u));
        acc += *(const unsigned long long*)(lut + ((tb << 3) & 0x7F8u));
        acc += *(const unsigned long long*)(lut + ((tb >> 13) & 0x7F8u));
        return acc;
    };
    auto one = [&](uint32_t sy) {
#pragma unroll
        for (int j = 0; j < 16; ++j) cnt[j] += (sy == (uint32_t)j) ? 1u : 0u;
    };
    auto spill8 = [&](uint64_t lo8, uint64_t hi8) {                 // 8-bit fields -> the 32-bit counters
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            cnt[2 * j] += (uint32_t)(lo8 >> (8 * j)) & 255u;
            cnt[2 * j + 1] += (uint32_t)(hi8 >> (8 * j)) & 255u;
        }
    };
    auto count_loads = [&](const u4 (&w)[U], int nvalid) {
        uint32_t bad = 0;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (u < nvalid) bad |= (w[u].x | w[u].y | w[u].z | w[u].w) & 0xF0F0F0F0u;
        if (__builtin_expect(__any(bad != 0), 0)) {                 // wave-uniform, never taken on quantizer output
#pragma nounroll
            for (int u = 0; u < nvalid; ++u) {
                const uint32_t ww[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma nounroll
                for (int c = 0; c < 16; ++c) one((ww[c >> 2] >> (8 * (c & 3))) & 255u);
            }
            return;
        }
        uint64_t lo8 = 0, hi8 = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (u < nvalid) {
                const uint64_t a = nib2(w[u].x, w[u].y), b = nib2(w[u].z, w[u].w);
                lo8 += (a & M) + (b & M);
                hi8 += ((a >> 4) & M) + ((b >> 4) & M);
            }
        }
        spill8(lo8, hi8);
    };
    int64_t done = 0;
    if ((((uintptr_t)idx) & 15) == 0) {
        const int64_t n16 = n >> 4;
        if (PF) {
            // Software pipeline.  All waves of the resident grid start together, so without it the whole chip alternates
            // between a load phase (VALU idle) and a counting phase (HBM idle): time = sum of the two, not their maximum.
            // The grid-uniform number of whole rounds (every lane U valid loads) runs double-buffered -- the loads of round
            // r + 1 are in flight while round r is counted -- with always-issued loads (the round index clamped: the last
            // round is fetched twice, from L2) so that no load sits behind a branch; the remainder is ONE more batch with
            // per-lane validity.
            const int64_t per_round = (int64_t)U * nth;
            const int64_t rounds = n16 / per_round;
            const u4* base = (const u4*)idx + tid;
            auto fetch = [&](u4 (&w)[U], int64_t r) {
                const int64_t rr = r < rounds ? r : rounds - 1;
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(base + rr * per_round + (int64_t)u * nth);
            };
            if (rounds > 0) {
                u4 a[U], b[U];
                fetch(a, 0);
                for (int64_t r = 0; r < rounds; r += 2) {
                    fetch(b, r + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    count_loads(a, U);
                    __builtin_amdgcn_sched_barrier(0);
                    fetch(a, r + 2);
                    __builtin_amdgcn_sched_barrier(0);
                    if (r + 1 < rounds) count_loads(b, U);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            {
                const int64_t i0 = rounds * per_round + tid;
                const int64_t left = n16 - i0;                     // may be <= 0
                const int nvalid = left <= 0 ? 0 : (int)((left + nth - 1) / nth < U ? (left + nth - 1) / nth : U);
                if (__any(nvalid > 0)) {
                    u4 w[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int64_t j = i0 + (int64_t)u * nth;
                        w[u] = __builtin_nontemporal_load((const u4*)idx + (j < n16 ? j : n16 - 1));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    count_loads(w, nvalid);
                }
            }
        } else {
            int64_t i = tid;
            for (; i + (int64_t)(U - 1) * nth < n16; i += (int64_t)U * nth) {
                u4 w[U];
#pragma unroll
                for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load((const u4*)idx + i + (int64_t)u * nth);
                __builtin_amdgcn_sched_barrier(0);                      // keep the U loads in flight together
                count_loads(w, U);
            }
            for (; i < n16; i += nth) {
                u4 w[U];
                w[0] = __builtin_nontemporal_load((const u4*)idx + i);
                count_loads(w, 1);
            }
        }
        done = n16 << 4;
    }
    for (int64_t i = done + tid; i < n; i += nth) one(idx[i]);
    // fixed fold: lanes -> wave (shuffles), waves -> block (LDS), one global atomic per bin per block
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        unsigned long long t = cnt[j];
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) t += __shfl_xor(t, sft);
        if (lane == 0) wsum[wv][j] = t;
    }
    __syncthreads();
    if (threadIdx.x < k) {
        unsigned long long t = 0;
        for (int w2 = 0; w2 < (int)(blockDim.x >> 6); ++w2) t += wsum[w2][threadIdx.x];
        if (t) atomicAdd(&hist[threadIdx.x], t);
    }
}

// histogram of uint8 symbols, any k <= 256, with INTEGER LDS atomics (ds_add_u32 without return) on a [k + 1][32]
// table shared by the block: column = lane mod 32, so the 32 lanes the LDS serves per cycle hit 32 different banks
// whatever their symbols are, and two lanes (or waves) that meet on one counter are resolved by the LDS itself -- no
// read-modify-write in registers, no duplicate merging, 3 VALU operations per symbol, 4.1-33 KiB of LDS per block.
template <int U>
__global__ __launch_bounds__(256) void k_hist_atomic(const uint8_t* idx, int64_t n, int k, unsigned long long* hist) {
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
    int64_t done = 0;
    if ((((uintptr_t)idx) & 15) == 0) {
        const int64_t n16 = n >> 4;
        const int64_t per_round = (int64_t)U * nth;
        const int64_t rounds = n16 / per_round;
        const u4* base = (const u4*)idx + tid;
        auto fetch = [&](u4 (&w)[U], int64_t r) {
            const int64_t rr = r < rounds ? r : rounds - 1;
#pragma unroll
            for (int u = 0; u < U; ++u) w[u] = __builtin_nontemporal_load(base + rr * per_round + (int64_t)u * nth);
        };
        auto count = [&](const u4 (&w)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) { bump_word(w[u].x); bump_word(w[u].y); bump_word(w[u].z); bump_word(w[u].w); }
        };
        if (rounds > 0) {                                    // double-buffered whole rounds, as k_hist_reg16
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
            const u4 w = __builtin_nontemporal_load((const u4*)idx + i);
            bump_word(w.x); bump_word(w.y); bump_word(w.z); bump_word(w.w);
        }
        done = n16 << 4;
    }
    for (int64_t i = done + tid; i < n; i += nth) bump(idx[i]);
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) {
        unsigned long long total = 0;
#pragma unroll 8
        for (int c = 0; c < 32; ++c) total += cnt[j * 32 + ((c + j) & 31)];
        if (total) atomicAdd(&hist[j], total);
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
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    static int use_reg = -1;
    if (use_reg < 0) { const char* e = getenv("QD_HIST_REG"); use_reg = e ? atoi(e) : 1; }   // QD_HIST_REG=0: the LDS-table kernel (A/B)
    hipLaunchKernelGGL(k_zero_u64, dim3(1), dim3(256), 0, st, (unsigned long long*)hist, k);   // (a memset node costs more)
    if (n == 0) return (int)hipGetLastError();
    if (k <= 16 && use_reg) {
        // register counters, no table: ONE 1024-lane block per CU (16 waves), U loads of 16 B in flight per lane.
        // (Variants measured at 64 Mi symbols: 2048 blocks of 256 lanes 39 us, 512 blocks 21.5 us, 256 blocks of 1024 lanes
        // 21.8 us; variable 64-bit shifts instead of the pair table 21.8 us; per-block totals + ticket + last-block fold
        // instead of the zeroing launch and the atomics 25 us -- and 83 us with an agent-scope fence in every lane.)
        static int u_sel = 0;
        if (u_sel == 0) { const char* e = getenv("QD_HIST_U"); u_sel = (e && atoi(e) > 0) ? atoi(e) : 8; }
        const int cap = cus < 256 ? cus : 256;
        static int pf_sel = -1;
        if (pf_sel < 0) { const char* e = getenv("QD_HIST_PF"); pf_sel = e ? atoi(e) : 0; }      // software-pipelined rounds (A/B)
        if (pf_sel) {
            if (u_sel == 2) hipLaunchKernelGGL((k_hist_reg16<2, true>), dim3(blocks_for(n, 1024 * 16 * 2, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
            else if (u_sel == 4) hipLaunchKernelGGL((k_hist_reg16<4, true>), dim3(blocks_for(n, 1024 * 16 * 4, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
            else hipLaunchKernelGGL((k_hist_reg16<8, true>), dim3(blocks_for(n, 1024 * 16 * 8, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
            return (int)hipGetLastError();
        }
        if (u_sel == 16) hipLaunchKernelGGL((k_hist_reg16<16, false>), dim3(blocks_for(n, 1024 * 16 * 16, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
        else if (u_sel == 4) hipLaunchKernelGGL((k_hist_reg16<4, false>), dim3(blocks_for(n, 1024 * 16 * 4, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
        else hipLaunchKernelGGL((k_hist_reg16<8, false>), dim3(blocks_for(n, 1024 * 16 * 8, cap)), dim3(1024), 0, st, idx, n, k, (unsigned long long*)hist);
        return (int)hipGetLastError();
    }
    static int at_sel = -1;
    if (at_sel < 0) { const char* e = getenv("QD_HIST_ATOMIC"); at_sel = e ? atoi(e) : 0; }   // LDS-atomic table, blocks per CU (A/B)
    if (at_sel > 0) {
        // a uint32 counter holds what ONE block counts in one launch: slices of at most blocks * 2^31 symbols
        const size_t lds = (size_t)(k + 1) * 32 * sizeof(uint32_t);
        const int blocks = blocks_for(n, 256 * 16 * 4, cus * at_sel);
        const int64_t slice = (int64_t)blocks << 31;
        for (int64_t off = 0; off < n; off += slice) {
            const int64_t len = n - off < slice ? n - off : slice;
            hipLaunchKernelGGL((k_hist_atomic<4>), dim3(blocks), dim3(256), lds, st, idx + off, len, k, (unsigned long long*)hist);
        }
        return (int)hipGetLastError();
    }
    const size_t lds_bytes = (size_t)(k + 1) * 256 * (k <= 64 ? sizeof(uint32_t) : sizeof(uint16_t));
    int per_cu = (int)((160 * 1024) / lds_bytes);
    if (per_cu < 1) per_cu = 1;
    // measured at 64 Mi symbols, k = 16: 43 / 36 / 41 / 61 us at 1 / 2 / 4 / 8 resident blocks per CU (the counting
    // is VALU/LDS-issue bound, and every block ends with k same-line global atomics)
    if (per_cu > 2) per_cu = 2;
    const int blocks = blocks_for(n, 256 * 16 * 4, cus * per_cu);
    if (k <= 64) {
        auto kern = k_hist_u8<uint32_t, 4>;
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, st, idx, n, k, (unsigned long long*)hist);
    } else {
        auto kern = k_hist_u8<uint16_t, 4>;
        if (lds_bytes > 64 * 1024)
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_bytes, st, idx, n, k, (unsigned long long*)hist);
    }
    return (int)hipGetLastError();
}

}  // extern "C"
