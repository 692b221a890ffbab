// qd_selftest.hip -- device-side self test of the one arithmetic shortcut the quantize kernels rest on:
// the division by a bucket-invariant alpha, div_alpha<true>() of qd_common.h (ref: the IEEE division of
// quantization/quant_functions.py:106-107, whose quotient decides the level index of :189-191).
//
// qd_selftest_div_invariant() generates adversarial (n, alpha) pairs ON THE DEVICE, evaluates the shortcut exactly as the
// kernels do -- y = RN(1 / alpha) by a true division, then q = RN(n y), r = fma(-alpha, q, n), u = fma(r, y, q); this file
// is compiled with the library's flags and calls the same inline function -- and compares the bits with the IEEE quotient
// n / alpha.  tools/div_invariant_check.py drives it over >= 10^9 pairs per family (docs/history/profiles/r03_div_invariant.txt);
// tests/test_hip_parity.py::test_division_by_bucket_invariant_alpha runs a 10^8-pair slice on every GPU test run.
//
// Domain of the claim (qd_common.h): alpha in [2^-60, 2^100] (fastdiv_ok), n = 0 or 2^-100 <= n, n / alpha finite.  In
// the kernels n = x - min(bucket) and alpha = max(bucket) - min(bucket), so 0 <= n <= alpha always (rounding is monotone).
// Where only the LEVEL of the quotient is consumed (quantize-dequantize) that is all; where the quotient itself is returned
// (scale_down) the numerator must also be 0 or >= alpha 2^-120, i.e. the quotient normal (family 5).

#include "qd_common.h"
#include "../../include/qd_hip.h"

using namespace qd;

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {          // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ float from_parts(uint32_t sign, int exp_unbiased, uint32_t mant23) {
    const int e = exp_unbiased + 127;
    return __uint_as_float((sign << 31) | ((uint32_t)(e < 1 ? 1 : (e > 254 ? 254 : e)) << 23) | (mant23 & 0x7FFFFFu));
}
// significands that stress the rounding of a quotient: all ones, all ones but the last bits, powers of two and their
// neighbours, single set bits, alternating patterns; otherwise random
__device__ __forceinline__ uint32_t nasty_mantissa(uint64_t h) {
    const uint32_t r = (uint32_t)(h >> 40) & 0x7FFFFFu;
    switch ((uint32_t)h & 15u) {
        case 0: return 0x7FFFFFu;
        case 1: return 0x7FFFFFu - ((uint32_t)(h >> 8) & 7u);
        case 2: return 0u;
        case 3: return (uint32_t)(h >> 8) & 7u;
        case 4: return 1u << ((uint32_t)(h >> 8) % 23u);
        case 5: return 0x7FFFFFu ^ (1u << ((uint32_t)(h >> 8) % 23u));
        case 6: return 0x555555u;
        case 7: return 0x2AAAAAu;
        case 8: return 0x400000u + ((uint32_t)(h >> 8) & 3u) - 1u;
        case 9: return r & 0x7FF000u;              // few significant bits
        case 10: return r | 0x000FFFu;
        default: return r;
    }
}

// result[0] = pairs tested, [1] = mismatches, [2] = bits of n and alpha of the first mismatch seen (n << 32 | alpha),
// [3] = pairs skipped because they fell outside the stated domain
__global__ __launch_bounds__(256) void k_selftest_div(uint64_t seed, int64_t npairs, int family,
                                                      unsigned long long* result) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    unsigned long long tested = 0, bad = 0, skipped = 0, first = 0;
    for (int64_t i = tid; i < npairs; i += nth) {
        const uint64_t h0 = mix64(seed ^ (uint64_t)i * 0xD1342543DE82EF95ull);
        const uint64_t h1 = mix64(h0), h2 = mix64(h1);
        float n, a;
        if (family == 0) {
            // the quantizer's own domain: a bucket's min / max at a random scale, n = x - min with x inside
            const int ea = (int)(h0 % 161u) - 60;                        // exponent of alpha: -60 .. 100
            a = from_parts(0, ea, (uint32_t)(h0 >> 20));
            const float frac = (float)((h1 >> 11) & 0xFFFFFFu) * (1.0f / 16777216.0f);
            const float mn = from_parts((uint32_t)(h1 & 1), ea + (int)((h1 >> 1) % 5u) - 2, (uint32_t)(h1 >> 36));
            const float mx = mn + a;
            a = mx - mn;                                                  // what the kernel computes
            float x = mn + frac * a;
            x = x > mx ? mx : x;
            n = x - mn;
        } else if (family == 1) {
            // wide exponents, random significands, any n whose quotient stays finite and normal-or-zero-level
            const int ea = (int)(h0 % 161u) - 60;
            a = from_parts(0, ea, (uint32_t)(h0 >> 20));
            int en = ea - (int)(h1 % 60u) + 8;                            // quotient exponent -52 .. +8
            n = from_parts(0, en, (uint32_t)(h1 >> 20));
        } else if (family == 2) {
            // adversarial significands on both sides
            const int ea = (int)(h0 % 161u) - 60;
            a = from_parts(0, ea, nasty_mantissa(h1));
            int en = ea - (int)((h0 >> 32) % 40u) + 4;
            n = from_parts(0, en, nasty_mantissa(h2));
        } else if (family == 3) {
            // near-exact quotients: n = RN(c * alpha) moved by -2 .. +2 ulps, c a short fraction (level-like: j / (s-1),
            // half-way points between levels, and random short values)
            const int ea = (int)(h0 % 161u) - 60;
            a = from_parts(0, ea, (h0 & (1ull << 40)) ? nasty_mantissa(h1) : (uint32_t)(h1 >> 20));
            const uint32_t s = 2u + (uint32_t)((h2 >> 8) % 255u);
            const uint32_t j = (uint32_t)((h2 >> 20) % (2u * s));
            float c = (h2 & 1) ? (float)j / (2.0f * (float)(s - 1 ? s - 1 : 1)) : (float)((h2 >> 30) & 0xFFFu) * (1.0f / 4096.0f);
            c = c > 1.0f ? 1.0f : c;
            const float p = c * a;
            const int d = (int)((h2 >> 44) % 5u) - 2;
            n = __uint_as_float(__float_as_uint(p) + (uint32_t)d);
            if (!(n >= 0.0f)) n = 0.0f;
        } else if (family == 5) {
            // small quotients: scale_down returns u itself, so the form must hold down to the smallest NORMAL quotients --
            // numerators from 2^-100 up, quotient exponents -120 .. -50 (denormal quotients are outside the claim: there the
            // last bit can differ, and the kernels' numerator threshold max(2^-100, alpha 2^-120) keeps them out)
            const int ea = (int)(h0 % 151u) - 50;                        // exponent of alpha: -50 .. 100
            a = from_parts(0, ea, (h0 & (1ull << 40)) ? nasty_mantissa(h1) : (uint32_t)(h1 >> 20));
            const int lo = ea - 120 > -100 ? ea - 120 : -100, hi = ea - 50;
            const int en = lo + (int)((h2 >> 8) % (uint32_t)(hi - lo + 1));
            n = from_parts(0, en, (h2 & 1) ? nasty_mantissa(h2 >> 4) : (uint32_t)(h2 >> 36));
            if (n < a * 0x1p-120f) n = 0.0f;
        } else {
            // the edges of the stated ranges: alpha at 2^-60 / 2^100 (+- an ulp inside), n at 2^-100 and at alpha
            const uint32_t pick = (uint32_t)(h0 & 3u);
            a = pick == 0 ? 0x1p-60f : (pick == 1 ? 0x1p100f : from_parts(0, (h0 & 4u) ? -60 : 99, nasty_mantissa(h1)));
            const uint32_t pn = (uint32_t)((h0 >> 8) & 3u);
            if (pn == 0) n = from_parts(0, -100, nasty_mantissa(h2));
            else if (pn == 1) n = a;
            else if (pn == 2) n = __uint_as_float(__float_as_uint(a) - 1u - (uint32_t)((h2 >> 8) & 3u));
            else n = from_parts(0, -100 + (int)((h2 >> 12) % 40u), (uint32_t)(h2 >> 30));
        }
        const bool in_domain = fastdiv_ok(a) && (n == 0.0f || n >= 0x1p-100f) && n <= 0x1p127f &&
                               (n / a) <= 0x1.fffffep127f;
        if (!in_domain) { ++skipped; continue; }
        const float y = 1.0f / a;                                         // RN(1/alpha): one IEEE division per bucket
        const float got = div_alpha<true>(n, a, y);
        const float want = n / a;
        ++tested;
        if (__float_as_uint(got) != __float_as_uint(want)) {
            if (!bad) first = ((unsigned long long)__float_as_uint(n) << 32) | __float_as_uint(a);
            ++bad;
        }
    }
    // per-wave totals, then one global atomic each (a self test, not a hot path)
    for (int s = 1; s < 64; s <<= 1) {
        tested += __shfl_xor(tested, s);
        bad += __shfl_xor(bad, s);
        skipped += __shfl_xor(skipped, s);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&result[0], tested);
        atomicAdd(&result[1], bad);
        atomicAdd(&result[3], skipped);
    }
    if (first) atomicCAS(&result[2], 0ull, first);
}

}  // namespace

extern "C" int qd_selftest_div_invariant(uint64_t seed, int64_t npairs, int family, unsigned long long* result,
                                         void* stream) {
    if (npairs < 0 || family < 0 || family > 5 || !result) return QD_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(result, 0, 4 * sizeof(unsigned long long), st) != hipSuccess) return (int)hipGetLastError();
    if (npairs == 0) return 0;
    int64_t blocks = (npairs + 256 * 64 - 1) / (256 * 64);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_selftest_div, dim3((unsigned)blocks), dim3(256), 0, st, seed, npairs, family, result);
    return (int)hipGetLastError();
}
