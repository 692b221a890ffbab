// qd_common.h -- device helpers shared by the gfx950 fake-quantization kernels.
//
// Everything here is written for CDNA4 wave64: DPP row rotations for the 16-lane reductions,
// ds_bpermute (via __shfl_xor) only for the two cross-row steps, 16-byte non-temporal global
// accesses.  Compile with -ffp-contract=off: the reference rounds every fp32 op separately.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef int64_t l2 __attribute__((ext_vector_type(2)));

#define QD_TOL_DIFF_ZERO 1e-10f  // reference: quantization/quant_functions.py:40

namespace qd {

// ---- explicitly global (address space 1) accesses ----
// Pointers that a kernel reads out of a descriptor table in memory are generic to the compiler and
// become flat_load / flat_store (which also occupy the LDS counter); these casts make them global_*.
#define QD_AS_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T ldg_nt(const T* p) {
    return __builtin_nontemporal_load((const QD_AS_GLOBAL T*)p);
}
template <typename T> __device__ __forceinline__ void stg_nt(const T& v, T* p) {
    __builtin_nontemporal_store(v, (QD_AS_GLOBAL T*)p);
}

// A non-temporal 16-byte store the optimiser cannot strip of its hint.  When the compiler merges two copies of a loop (the
// whole-tile and the partial-tile branch of a kernel) the merged store can lose !nontemporal -- the ISA then shows a plain
// global_store_dwordx4 and a write-once output starts competing for the caches (scale_down at bucket 256: 92 us against
// 85 us).  tests/test_abi.py checks the shipped code objects for that.  The s_nop is the wait state the hardware needs
// between a store of more than 8 bytes and a VALU write of its data registers: the hazard recogniser does not look inside
// an asm block (without it a golden case stored a register the next instruction had already overwritten).
// The instruction is a GLOBAL store, so the pointer type says so (address space 1: a generic pointer into LDS or scratch
// does not convert implicitly), and the wait-state count is gfx950's: this header is for that target only.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "qd_common.h: hand-written gfx950 ISA (store_nt_pinned's wait states, DPP controls): compile with --offload-arch=gfx950"
#endif
__device__ __forceinline__ void store_nt_pinned(QD_AS_GLOBAL f4* p, const f4& v) {
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
template <typename T> __device__ __forceinline__ T ldg(const T* p) { return *(const QD_AS_GLOBAL T*)p; }
template <typename T> __device__ __forceinline__ void stg(const T& v, T* p) { *(QD_AS_GLOBAL T*)p = v; }
// wave index within the grid as a scalar (the compiler cannot see that threadIdx.x >> 6 is wave-uniform)
__device__ __forceinline__ int64_t uniform_wave_index() {
    return (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// ---- compute units of the CURRENT device (256 on MI355X), cached per device ----
// (one process may drive several devices: the API takes tensors of any device and launches with that device current)
inline int device_cus() {
    static int cache[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 256; }
    const bool cached = dev >= 0 && dev < 64;                  // a device index beyond the table is asked every time, never aliased
    int v = cached ? __atomic_load_n(&cache[dev], __ATOMIC_RELAXED) : 0;
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 256; }
        if (cached) __atomic_store_n(&cache[dev], v, __ATOMIC_RELAXED);
    }
    return v;
}

// ---- NaN-propagating min / max ----
// torch's min/max propagate NaN (one NaN makes alpha, beta and every output of its bucket NaN in the reference), whereas
// v_min_f32 / v_max_f32 (fminf / fmaxf) return the other operand.  gfx950 has the IEEE-754-2019 `minimum` / `maximum` as
// three-operand instructions (v_minimum3_f32 / v_maximum3_f32): NaN if any operand is NaN, and two nested calls fuse into
// one instruction.  Every min/max reduction of DATA below uses these, so no separate NaN flag has to be carried.
__device__ __forceinline__ float pmin(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float pmax(float a, float b) { return __builtin_elementwise_maximum(a, b); }
__device__ __forceinline__ float pmin4(const f4& v) { return pmin(pmin(v.x, v.y), pmin(v.z, v.w)); }
__device__ __forceinline__ float pmax4(const f4& v) { return pmax(pmax(v.x, v.y), pmax(v.z, v.w)); }

// ---- DPP row rotations: lane i of each 16-lane row reads lane (i + s) mod 16 of its row ----
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false);
}
#define QD_ROR8 0x128
#define QD_ROR4 0x124
#define QD_ROR2 0x122
#define QD_ROR1 0x121

// all-reduce over the 16 lanes of a DPP row (min / max are idempotent; sum uses the same tree)
__device__ __forceinline__ float row16_min(float v) {
    v = pmin(v, dpp_f<QD_ROR8>(v));
    v = pmin(v, dpp_f<QD_ROR4>(v));
    v = pmin(v, dpp_f<QD_ROR2>(v));
    v = pmin(v, dpp_f<QD_ROR1>(v));
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = pmax(v, dpp_f<QD_ROR8>(v));
    v = pmax(v, dpp_f<QD_ROR4>(v));
    v = pmax(v, dpp_f<QD_ROR2>(v));
    v = pmax(v, dpp_f<QD_ROR1>(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    v = v + dpp_f<QD_ROR8>(v);
    v = v + dpp_f<QD_ROR4>(v);
    v = v + dpp_f<QD_ROR2>(v);
    v = v + dpp_f<QD_ROR1>(v);
    return v;
}
__device__ __forceinline__ int row16_imin(int v) {
    v = min(v, dpp_i<QD_ROR8>(v));
    v = min(v, dpp_i<QD_ROR4>(v));
    v = min(v, dpp_i<QD_ROR2>(v));
    v = min(v, dpp_i<QD_ROR1>(v));
    return v;
}

// all-reduce over the 64 lanes of a wave: row step by DPP, the two cross-row steps by bpermute
__device__ __forceinline__ float wave_min(float v) {
    v = row16_min(v);
    v = pmin(v, __shfl_xor(v, 16));
    v = pmin(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = pmax(v, __shfl_xor(v, 16));
    v = pmax(v, __shfl_xor(v, 32));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v = v + __shfl_xor(v, 16);
    v = v + __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) v += __shfl_xor(v, s);
    return v;
}
__device__ __forceinline__ long long wave_min_ll(long long v) {
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        long long o = __shfl_xor(v, s);
        v = o < v ? o : v;
    }
    return v;
}

// all-reduce over a group of LANES lanes (16 = DPP row, 32 = half wave, 64 = wave)
template <int LANES> __device__ __forceinline__ float group_min(float v) {
    v = row16_min(v);
    if (LANES >= 32) v = pmin(v, __shfl_xor(v, 16));
    if (LANES >= 64) v = pmin(v, __shfl_xor(v, 32));
    return v;
}
template <int LANES> __device__ __forceinline__ float group_max(float v) {
    v = row16_max(v);
    if (LANES >= 32) v = pmax(v, __shfl_xor(v, 16));
    if (LANES >= 64) v = pmax(v, __shfl_xor(v, 32));
    return v;
}
template <int LANES> __device__ __forceinline__ float group_sum(float v) {
    v = row16_sum(v);
    if (LANES >= 32) v = v + __shfl_xor(v, 16);
    if (LANES >= 64) v = v + __shfl_xor(v, 32);
    return v;
}
template <int LANES> __device__ __forceinline__ int group_imin(int v) {
    v = row16_imin(v);
    if (LANES >= 32) v = min(v, __shfl_xor(v, 16));
    if (LANES >= 64) v = min(v, __shfl_xor(v, 32));
    return v;
}

// block-wide all-reduce helpers (blockDim.x a multiple of 64, <= 1024); `red` is LDS scratch of
// >= 32 floats; results are broadcast to every thread.
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red) {
    mn = wave_min(mn);
    mx = wave_max(mx);
    const int nw = blockDim.x >> 6;
    if (nw == 1) return;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();  // protect `red` from a previous use
    if (lane == 0) { red[w] = mn; red[16 + w] = mx; }
    __syncthreads();
    float a = red[0], b = red[16];
    for (int i = 1; i < nw; ++i) { a = pmin(a, red[i]); b = pmax(b, red[16 + i]); }
    mn = a; mx = b;
}

// ---- per-element preparation: mean subtraction then clamp (quant_functions.py:66-74) ----
struct Prep {
    float mean;  // 0 when subtract_mean is off: x - 0 is exact
    float me;    // +inf when max_element is off
};
__device__ __forceinline__ float prep(float x, const Prep& p) {
    x = x - p.mean;
    x = x > p.me ? p.me : x;
    x = x < -p.me ? -p.me : x;
    return x;
}
__device__ __forceinline__ f4 prep4(f4 v, const Prep& p) {
    v.x = prep(v.x, p); v.y = prep(v.y, p); v.z = prep(v.z, p); v.w = prep(v.w, p);
    return v;
}

// NaN poisoning for the kernels that still reduce with a separate flag (the slow generic paths): torch's min/max
// propagate NaN, so one NaN makes alpha, beta and every output of its bucket NaN in the reference.  The hot kernels use
// pmin / pmax above instead.  (Infinities need nothing: alpha = inf or beta = -inf turn the bucket into NaN through the
// same arithmetic.)
__device__ __forceinline__ bool has_nan4(const f4& v) { return (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w); }
// true for every lane of a LANES-wide group (16 = DPP row, 64 = wave) if any lane of it raised `flag`
template <int LANES>
__device__ __forceinline__ bool group_any(bool flag) {
    const unsigned long long m = __ballot(flag);
    if (LANES == 64) return m != 0ull;
    const int lane = threadIdx.x & 63;
    return ((m >> (lane & 48)) & 0xFFFFull) != 0ull;
}

// alpha/beta of a bucket from its min/max (quant_functions.py:91-99)
__device__ __forceinline__ void alpha_beta(float mn, float mx, float& a, float& b) {
    a = mx - mn;
    a = a < QD_TOL_DIFF_ZERO ? 1.0f : a;
    b = mn;
}

// ---- division by a bucket-invariant alpha ----
// u = (x - beta) / alpha is the reference's IEEE division (quant_functions.py:106-107) and the level index
// rint(u (s-1)) must match bit for bit, so no reciprocal shortcut -- but the divisor is the same for a whole bucket.
// With y = RN(1 / alpha) computed ONCE per bucket by a true division (correctly rounded),
//     q = RN(n y);  r = n - alpha q  (exact in one FMA);  u = RN(q + r y)
// is the correctly rounded quotient RN(n / alpha) (Markstein 1990; the classic "division by a loop invariant"): three
// VALU operations per element instead of the ~10 of the IEEE division macro.  Preconditions are no underflow in r and a
// normal y; both hold for alpha in [2^-60, 2^100] and n = 0 or n >= 2^-100 (checked against true division on 5 * 10^9
// adversarial pairs incl. all-ones significands: 0 mismatches; outside those ranges mismatches do occur).  A numerator
// below 2^-100 with alpha >= 2^-60 gives u < 2^-40, whose level is 0 whatever its last bit is -- so the form is used
// where only the LEVEL of u is consumed (quantize-dequantize, deterministic or stochastic) without looking at the
// numerators; scale_down, which returns u itself, uses it in the buckets whose nonzero numerators are all at least
// max(2^-100, alpha 2^-120) -- a NORMAL quotient; a denormal one can differ in its last bit (scale_fast_ok below) --
// and never where u is compared with points.  Buckets outside the range (incl. inf / NaN alpha) take the IEEE path.
__device__ __forceinline__ bool fastdiv_ok(float a) { return a >= 0x1p-60f && a <= 0x1p100f; }   // false for NaN
template <bool FAST>
__device__ __forceinline__ float div_alpha(float n, float a, float y) {
    if (FAST) {
        const float q = n * y;
        const float r = __builtin_fmaf(-a, q, n);
        return __builtin_fmaf(r, y, q);
    }
    return n / a;
}

// scale_down RETURNS u, so the bucket-invariant division form (qd_common.h) may replace the IEEE division only in a bucket
// whose alpha is in the proven range AND whose numerators v - beta are all 0 or at least max(2^-100, alpha 2^-120): below
// 2^-100 the remainder underflows, and a DENORMAL quotient can differ in its last bit (exact-arithmetic restatement,
// tools/div_invariant_check.py: 3 of 3518 such pairs; 0 of 33824 with small normal quotients; family 5 of the device self
// test).  One unsigned minimum per element decides it: bits(n) - 1 wraps 0 to the top, and n >= 0 (beta is the bucket's
// minimum; a NaN anywhere makes alpha NaN, which fastdiv_ok refuses).  Almost every bucket of real data passes; one that
// does not takes the IEEE form as before.
__device__ __forceinline__ unsigned scale_numerator_key(float v, float b) { return __float_as_uint(v - b) - 1u; }
__device__ __forceinline__ bool scale_fast_ok(float a, unsigned min_key) {
    const float thr = fmaxf(0x1p-100f, a * 0x1p-120f);
    return fastdiv_ok(a) && min_key >= __float_as_uint(thr) - 1u;
}

// ---- the k-level quantize-dequantize of one element (quant_functions.py:106-107,189-191,142-148)
// Seven separately rounded fp32 ops; both divisions are IEEE-correct (no reciprocal shortcut):
// the level index rint(u*(s-1)) must match the reference bit for bit.
template <bool FAST = false>
__device__ __forceinline__ float qdq(float v, float a, float b, float sm1, float mean, float& level, float ry = 0.0f) {
    float u = v - b;
    u = div_alpha<FAST>(u, a, ry);
    float t = u * sm1;
    float r = rintf(t);
    level = r;
    float w = r / sm1;
    float y = w * a;
    y = y + b;
    y = y + mean;
    return y;
}

// Same as qdq() when levels <= 16, with the second division (level / (levels - 1)) looked up instead of computed:
// lane l of every 16-lane DPP row holds tab = (float)l / sm1 -- the same correctly rounded IEEE quotient -- and the
// element fetches its level's entry with one ds_bpermute from its own row (rows are active or inactive as a whole
// in the vector kernels).  Saves ~8 VALU instructions per element; bit-identical: level = rint(t) is an integer in
// [0, sm1] for finite input, and for NaN (v_cvt_i32_f32 gives 0) w = 0 still yields NaN through a or b.
template <bool FAST = false>
__device__ __forceinline__ float qdq_tab(float v, float a, float b, float sm1, float mean, float& level, float tab,
                                         float ry = 0.0f) {
    float u = v - b;
    u = div_alpha<FAST>(u, a, ry);
    float t = u * sm1;
    float r = rintf(t);
    level = r;
    const int src = (int)(threadIdx.x & 48) + (int)r;
    float w = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(tab)));
    float y = w * a;
    y = y + b;
    y = y + mean;
    return y;
}

// ---- Philox4x32-7 counter-based generator for the stochastic-rounding branch ----
// 7 rounds is the smallest Philox4x32 variant that passes BigCrush (Salmon et al., SC'11); the
// integer multiplies are quarter-rate on CDNA, so the rounds are what the stochastic kernel pays for.
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
// four uniforms in [0,1) for the 4 consecutive elements starting at element index e (e % 4 == 0
// on the vector paths; scalar paths use component e & 3 of block e >> 2)
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t block, float (&out)[4]) {
    uint32_t c[4] = {(uint32_t)block, (uint32_t)(block >> 32), 0x51ed270bu, 0x2545f491u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) out[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}
// stochastic variant (quant_functions.py:174-187): floor + Bernoulli(frac)
template <bool FAST = false>
__device__ __forceinline__ float qdq_stochastic(float v, float a, float b, float sm1, float mean, float rnd,
                                                float& level, float ry = 0.0f) {
    float u = v - b;
    u = div_alpha<FAST>(u, a, ry);
    float t = u * sm1;     // == probabilities before the subtraction (same product)
    float l = floorf(t);
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

// stochastic variant with the same per-row table for floor(t) / sm1 (levels <= 16)
template <bool FAST = false>
__device__ __forceinline__ float qdq_stochastic_tab(float v, float a, float b, float sm1, float mean, float rnd,
                                                    float& level, float tab, float ry = 0.0f) {
    float u = v - b;
    u = div_alpha<FAST>(u, a, ry);
    float t = u * sm1;
    float l = floorf(t);
    float p = t - l;
    const int src = (int)(threadIdx.x & 48) + (int)l;
    float w = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(tab)));
    float inc = (rnd <= p) ? (1.0f / sm1) : 0.0f;
    level = l + ((rnd <= p) ? 1.0f : 0.0f);
    w = w + inc;
    float y = w * a;
    y = y + b;
    y = y + mean;
    return y;
}

// ---- sorted-array searches over LDS (uniform trip count, branch-free) ----
// count of a[j] <  u (lower bound) when UPPER == false; count of a[j] <= u when UPPER == true
template <bool UPPER, typename P>      // P: pointer to float in any address space (LDS tables pass address_space(3) pointers)
__device__ __forceinline__ int count_before(P a, int n, float u) {
    int lo = 0;
    while (n > 1) {
        const int half = n >> 1;
        const float v = a[lo + half - 1];
        const bool c = UPPER ? (v <= u) : (v < u);
        lo = c ? lo + half : lo;
        n -= half;
    }
    if (n == 1) {
        const float v = a[lo];
        const bool c = UPPER ? (v <= u) : (v < u);
        lo += c ? 1 : 0;
    }
    return lo;
}

// Four searches at once over the same sorted array: the same uniform trip count, but the four LDS reads of a step are
// independent -- one LDS round trip per step and float4 instead of four dependent chains one after the other.  (A kernel
// that holds several float4s per lane at two or three waves per SIMD cannot hide those chains behind other waves: the
// pre-processed forward at bucket 100 ran at 145 us against 94 us for the same kernel without the point search.)
template <bool UPPER, typename P>
__device__ __forceinline__ void count_before4(P a, int n, const float (&u)[4], int (&lo)[4]) {
    lo[0] = 0; lo[1] = 0; lo[2] = 0; lo[3] = 0;
    while (n > 1) {
        const int half = n >> 1;
        const float v0 = a[lo[0] + half - 1], v1 = a[lo[1] + half - 1], v2 = a[lo[2] + half - 1], v3 = a[lo[3] + half - 1];
        lo[0] += (UPPER ? (v0 <= u[0]) : (v0 < u[0])) ? half : 0;
        lo[1] += (UPPER ? (v1 <= u[1]) : (v1 < u[1])) ? half : 0;
        lo[2] += (UPPER ? (v2 <= u[2]) : (v2 < u[2])) ? half : 0;
        lo[3] += (UPPER ? (v3 <= u[3]) : (v3 < u[3])) ? half : 0;
        n -= half;
    }
    if (n == 1) {
        const float v0 = a[lo[0]], v1 = a[lo[1]], v2 = a[lo[2]], v3 = a[lo[3]];
        lo[0] += (UPPER ? (v0 <= u[0]) : (v0 < u[0])) ? 1 : 0;
        lo[1] += (UPPER ? (v1 <= u[1]) : (v1 < u[1])) ? 1 : 0;
        lo[2] += (UPPER ? (v2 <= u[2]) : (v2 < u[2])) ? 1 : 0;
        lo[3] += (UPPER ? (v3 <= u[3]) : (v3 < u[3])) ? 1 : 0;
    }
}

}  // namespace qd
