// qd_select.hip -- a few order statistics of an fp32 array without sorting it (gfx950).
//
// initialize_quantization_points (reference: quantization/help_functions.py:140-154) needs
// np.percentile(scaled, linspace(0, 100, k)): 2k order statistics of the scaled tensor (the two neighbours of each
// virtual index), and the reference gets them by copying the tensor to the host and letting numpy partition it.  A
// device sort moves every element ~8 times (radix passes, read + write); the 2k ranks only need COUNTS:
//
//   key(v) = the 32-bit pattern of v made order-preserving (sign flip), split 12 | 10 | 10 bits.
//   pass 1: histogram of the top 12 bits over all elements            -> bin of every target rank, rank inside the bin
//   pass 2: for elements in a target's bin, histogram of the next 10  -> 22-bit prefix of every target, residual rank
//   pass 3: for elements under a target's 22-bit prefix, the last 10  -> the key itself = the value, exactly
//
// Three reads of the array (4 B/elem each), nothing written but per-block partial histograms (summed by the picking
// kernels: no global atomics, no zeroing launch, deterministic).  The counters of passes 2 and 3 are [targets][1024] in
// LDS (<= 32 targets = 128 KiB of the 160 KiB), one 1024-lane block per CU.  Exact for every float (NaNs order last,
// -0 before +0); equal keys are the same value, so ties need no care.
#include "qd_common.h"
#include <atomic>

#include "../../include/qd_hip.h"

using namespace qd;

namespace {

constexpr int TOP_BITS = 12, SUB_BITS = 10;
constexpr int TOP_BINS = 1 << TOP_BITS, SUB_BINS = 1 << SUB_BITS;
constexpr int MAX_RANKS = QD_ORDER_STATS_MAX_RANKS;
constexpr int SEL_THREADS = 1024;
constexpr int MAX_BLOCKS = 256;

struct Ranks {
    uint32_t r[MAX_RANKS];
};

__device__ __forceinline__ uint32_t order_key(float v) {
    const uint32_t b = __float_as_uint(v);
    return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u);
}
__device__ __forceinline__ float key_value(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

// The array as [head scalars][nvec float4][tail scalars] around the 16-byte-aligned middle.
struct Span {
    const f4* vec;
    int64_t nvec;
    const float* tail;
    int head, ntail;
};
__device__ __forceinline__ Span span_of(const float* u, int64_t n) {
    Span s;
    int64_t head = (int64_t)((16 - ((uintptr_t)u & 15)) & 15) >> 2;
    if (head > n) head = n;
    s.head = (int)head;
    s.vec = (const f4*)(u + head);
    s.nvec = (n - head) >> 2;
    s.tail = u + head + 4 * s.nvec;
    s.ntail = (int)(n - head - 4 * s.nvec);
    return s;
}

// Calls f(key) for every element of u[0..n) exactly once across the grid; U 16-byte loads in flight per lane.
template <int U, typename F>
__device__ __forceinline__ void for_each_key(const float* u, int64_t n, F&& f) {
    const Span s = span_of(u, n);
    const int64_t stride = (int64_t)gridDim.x * SEL_THREADS;
    int64_t i = (int64_t)blockIdx.x * SEL_THREADS + threadIdx.x;
    for (; i + (U - 1) * stride < s.nvec; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = ldg_nt(s.vec + i + j * stride);
#pragma unroll
        for (int j = 0; j < U; ++j) {
            f(order_key(v[j].x));
            f(order_key(v[j].y));
            f(order_key(v[j].z));
            f(order_key(v[j].w));
        }
    }
    for (; i < s.nvec; i += stride) {
        const f4 v = ldg_nt(s.vec + i);
        f(order_key(v.x));
        f(order_key(v.y));
        f(order_key(v.z));
        f(order_key(v.w));
    }
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < s.head) f(order_key(u[threadIdx.x]));
        if ((int)threadIdx.x < s.ntail) f(order_key(s.tail[threadIdx.x]));
    }
}

// pass 1: part[block][4096] = counts of the top 12 key bits
template <int U>
__global__ __launch_bounds__(SEL_THREADS) void k_sel_hist_top(const float* u, int64_t n, uint32_t* part) {
    __shared__ uint32_t h[TOP_BINS];
    for (int i = threadIdx.x; i < TOP_BINS; i += SEL_THREADS) h[i] = 0;
    __syncthreads();
    for_each_key<U>(u, n, [&](uint32_t key) { atomicAdd(&h[key >> (32 - TOP_BITS)], 1u); });
    __syncthreads();
    uint32_t* out = part + (size_t)blockIdx.x * TOP_BINS;
    for (int i = threadIdx.x; i < TOP_BINS; i += SEL_THREADS) out[i] = h[i];
}

// sum over the per-block partials with `nblocks` independent loads in flight (a plain loop issues them one by one:
// 35 us for 256 blocks, all of it load latency)
__device__ __forceinline__ uint32_t sum_strided(const uint32_t* p, int count, size_t stride) {
    uint32_t c = 0;
    int b = 0;
    for (; b + 16 <= count; b += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = ldg(p + (size_t)(b + j) * stride);
#pragma unroll
        for (int j = 0; j < 16; ++j) c += v[j];
    }
    for (; b < count; ++b) c += ldg(p + (size_t)b * stride);
    return c;
}

// hist[bin] = sum over blocks; 64 bins per block, the blocks of the partials split over the 4 waves
__global__ __launch_bounds__(256) void k_sel_sum_top(const uint32_t* part, int nblocks, uint32_t* hist) {
    __shared__ uint32_t s[4][64];
    const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int bin = blockIdx.x * 64 + lane;
    const int per = (nblocks + 3) / 4;
    const int first = g * per;
    int count = nblocks - first;
    if (count > per) count = per;
    s[g][lane] = count > 0 ? sum_strided(part + (size_t)first * TOP_BINS + bin, count, TOP_BINS) : 0u;
    __syncthreads();
    if (g == 0) hist[bin] = s[0][lane] + s[1][lane] + s[2][lane] + s[3][lane];
}

// exclusive scan of one value per lane over a 1024-lane block; `totals` is 16 words of LDS
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t c, uint32_t* totals) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) totals[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += totals[w];
    __syncthreads();
    return base + inc - c;
}

// the bin of each target rank among the 4096 top bins, and its rank inside that bin
__global__ __launch_bounds__(SEL_THREADS) void k_sel_pick_top(const uint32_t* hist, Ranks ranks, int m, uint32_t* prefix,
                                                              uint32_t* resid) {
    __shared__ uint32_t cum[TOP_BINS + 1];
    __shared__ uint32_t totals[16];
    constexpr int PER = TOP_BINS / SEL_THREADS;
    uint32_t c[PER], sum = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        c[j] = hist[threadIdx.x * PER + j];
        sum += c[j];
    }
    uint32_t e = block_exclusive_scan(sum, totals);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        cum[threadIdx.x * PER + j] = e;
        e += c[j];
    }
    if (threadIdx.x == SEL_THREADS - 1) cum[TOP_BINS] = e;
    __syncthreads();
    if ((int)threadIdx.x < m) {
        const uint32_t r = ranks.r[threadIdx.x];
        int lo = 0, hi = TOP_BINS;                       // largest b with cum[b] <= r
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (cum[mid] <= r) lo = mid; else hi = mid;
        }
        prefix[threadIdx.x] = (uint32_t)lo;
        resid[threadIdx.x] = r - cum[lo];
    }
}

// passes 2 and 3: part[block][row][1024] = counts of the next 10 key bits among the elements whose leading
// PREFIX_BITS bits equal a target's prefix; row = the first target with that prefix (prefix[] is non-decreasing)
template <int PREFIX_BITS, int U>
__global__ __launch_bounds__(SEL_THREADS) void k_sel_hist_sub(const float* u, int64_t n, const uint32_t* prefix, int m,
                                                              uint32_t* part) {
    extern __shared__ uint32_t h[];                      // [m][1024]
    __shared__ uint32_t s_prefix[MAX_RANKS];
    __shared__ int8_t s_first[TOP_BINS];                 // top 12 bits -> first target under them, or -1
    __shared__ uint32_t s_mask[PREFIX_BITS > TOP_BITS ? MAX_RANKS * (SUB_BINS / 32) : 1];   // pass 3: per first target, which
                                                         // next-10-bit digits some target's prefix has (rejects ~all elements)
    for (int i = threadIdx.x; i < m * SUB_BINS; i += SEL_THREADS) h[i] = 0;
    for (int i = threadIdx.x; i < TOP_BINS; i += SEL_THREADS) s_first[i] = -1;
    if ((int)threadIdx.x < m) s_prefix[threadIdx.x] = prefix[threadIdx.x];
    if (PREFIX_BITS > TOP_BITS)
        for (int i = threadIdx.x; i < m * (SUB_BINS / 32); i += SEL_THREADS) s_mask[i] = 0;
    __syncthreads();
    if ((int)threadIdx.x < m) {
        const uint32_t top = s_prefix[threadIdx.x] >> (PREFIX_BITS - TOP_BITS);
        int first = threadIdx.x;
        while (first > 0 && (s_prefix[first - 1] >> (PREFIX_BITS - TOP_BITS)) == top) --first;
        if (first == (int)threadIdx.x) s_first[top] = (int8_t)first;
        if (PREFIX_BITS > TOP_BITS) {
            const uint32_t digit = s_prefix[threadIdx.x] & (SUB_BINS - 1);
            atomicOr(&s_mask[first * (SUB_BINS / 32) + (digit >> 5)], 1u << (digit & 31));
        }
    }
    __syncthreads();
    constexpr int SHIFT = 32 - PREFIX_BITS - SUB_BITS;
    for_each_key<U>(u, n, [&](uint32_t key) {
        int row = s_first[key >> (32 - TOP_BITS)];
        if (row < 0) return;
        if (PREFIX_BITS > TOP_BITS) {
            const uint32_t p = key >> (32 - PREFIX_BITS);
            const uint32_t digit = p & (SUB_BINS - 1);
            if (!((s_mask[row * (SUB_BINS / 32) + (digit >> 5)] >> (digit & 31)) & 1u)) return;
            while (row < m && s_prefix[row] < p) ++row;
            if (row >= m || s_prefix[row] != p) return;
        }
        atomicAdd(&h[row * SUB_BINS + ((key >> SHIFT) & (SUB_BINS - 1))], 1u);
    });
    __syncthreads();
    uint32_t* out = part + (size_t)blockIdx.x * m * SUB_BINS;
    for (int i = threadIdx.x; i < m * SUB_BINS; i += SEL_THREADS) out[i] = h[i];
}

// One block per target: sum its row over the blocks, scan, find the 10-bit digit holding the residual rank.
// Lane (g, j) of the 4 x 256 sums digits 4j..4j+3 (one 16-byte load) over the blocks b = g mod 4, 16 loads in flight.
// LAST: the prefix is now the whole key -> write the value.
template <bool LAST>
__global__ __launch_bounds__(SEL_THREADS) void k_sel_pick_sub(const uint32_t* part, int nblocks, int m,
                                                              const uint32_t* prefix_in, uint32_t* prefix_out, uint32_t* resid,
                                                              float* out) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    __shared__ uint32_t totals[16];
    __shared__ u4 s_sum[4][256];
    const int t = blockIdx.x;
    const uint32_t p = prefix_in[t];
    int row = t;
    while (row > 0 && prefix_in[row - 1] == p) --row;
    const uint32_t r = resid[t];
    const int g = threadIdx.x >> 8, j = threadIdx.x & 255;
    const size_t stride = (size_t)m * SUB_BINS / 4;                                   // in 16-byte units
    const u4* col = (const u4*)(part + (size_t)row * SUB_BINS) + j;
    u4 acc = {0, 0, 0, 0};
    int b = g;
    for (; b + 4 * 15 < nblocks; b += 4 * 16) {
        u4 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = ldg(col + (size_t)(b + 4 * q) * stride);
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
    }
    for (; b < nblocks; b += 4) acc += ldg(col + (size_t)b * stride);
    s_sum[g][j] = acc;
    __syncthreads();
    uint32_t c[4] = {0, 0, 0, 0};
    if (g == 0) {
        const u4 a = s_sum[0][j] + s_sum[1][j] + s_sum[2][j] + s_sum[3][j];
        c[0] = a.x; c[1] = a.y; c[2] = a.z; c[3] = a.w;
    }
    uint32_t e = block_exclusive_scan(c[0] + c[1] + c[2] + c[3], totals);            // lanes >= 256 contribute 0
    if (g == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e <= r && r - e < c[q]) {
                const uint32_t np = (p << SUB_BITS) | (uint32_t)(4 * j + q);
                if (LAST) out[t] = key_value(np);
                else {
                    prefix_out[t] = np;
                    resid[t] = r - e;
                }
            }
            e += c[q];
        }
    }
}

struct Layout {
    size_t part, hist, prefix_a, prefix_b, resid, total;
};
inline Layout layout_for(int m) {
    Layout l;
    const size_t words = (size_t)MAX_BLOCKS * (m * SUB_BINS > TOP_BINS ? m * SUB_BINS : TOP_BINS);
    l.part = 0;
    l.hist = words * 4;
    l.prefix_a = l.hist + TOP_BINS * 4;
    l.prefix_b = l.prefix_a + MAX_RANKS * 4;
    l.resid = l.prefix_b + MAX_RANKS * 4;
    l.total = l.resid + MAX_RANKS * 4;
    return l;
}

}  // namespace

extern "C" {

size_t qd_order_stats_workspace_bytes(int m) {
    if (m < 1) m = 1;
    if (m > MAX_RANKS) m = MAX_RANKS;
    return layout_for(m).total;
}

int qd_order_stats_f32(const float* x, int64_t n, const int64_t* ranks, int m, float* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    if (!x || !ranks || !out || n < 1 || m < 1 || ((uintptr_t)x & 3)) return QD_ERR_INVALID_ARGUMENT;
    if (m > MAX_RANKS || n >= ((int64_t)1 << 32)) return QD_ERR_UNSUPPORTED;
    Ranks rk;
    for (int t = 0; t < m; ++t) {
        if (ranks[t] < 0 || ranks[t] >= n || (t > 0 && ranks[t] < ranks[t - 1])) return QD_ERR_INVALID_ARGUMENT;
        rk.r[t] = (uint32_t)ranks[t];
    }
    for (int t = m; t < MAX_RANKS; ++t) rk.r[t] = 0;
    const Layout l = layout_for(m);
    if (!workspace || workspace_bytes < l.total) return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    uint32_t* part = (uint32_t*)(ws + l.part);
    uint32_t* hist = (uint32_t*)(ws + l.hist);
    uint32_t* prefix_a = (uint32_t*)(ws + l.prefix_a);
    uint32_t* prefix_b = (uint32_t*)(ws + l.prefix_b);
    uint32_t* resid = (uint32_t*)(ws + l.resid);

    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); dev = 0; }
    const int cus = device_cus();
    constexpr int U = 4;
    int64_t want = (n + (int64_t)SEL_THREADS * 4 * U - 1) / ((int64_t)SEL_THREADS * 4 * U);
    const int cap = cus < MAX_BLOCKS ? cus : MAX_BLOCKS;
    const int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    const size_t lds = (size_t)m * SUB_BINS * sizeof(uint32_t);
    // the attribute is per DEVICE: raise it once for each device this process launches on
    static std::atomic<unsigned long long> lds_raised{0};
    const unsigned long long dev_bit = 1ull << (dev & 63);
    if (!(lds_raised.load(std::memory_order_relaxed) & dev_bit)) {
        const hipError_t e1 = hipFuncSetAttribute((const void*)k_sel_hist_sub<TOP_BITS, U>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAX_RANKS * SUB_BINS * 4);
        const hipError_t e2 = hipFuncSetAttribute((const void*)k_sel_hist_sub<TOP_BITS + SUB_BITS, U>,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAX_RANKS * SUB_BINS * 4);
        if (e1 != hipSuccess || e2 != hipSuccess) {
            (void)hipGetLastError();
            return (int)(e1 != hipSuccess ? e1 : e2);          // positive hipError_t, as every entry point reports launch failures
        }
        lds_raised.fetch_or(dev_bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((k_sel_hist_top<U>), dim3(blocks), dim3(SEL_THREADS), 0, st, x, n, part);
    hipLaunchKernelGGL(k_sel_sum_top, dim3(TOP_BINS / 64), dim3(256), 0, st, part, blocks, hist);
    hipLaunchKernelGGL(k_sel_pick_top, dim3(1), dim3(SEL_THREADS), 0, st, hist, rk, m, prefix_a, resid);
    hipLaunchKernelGGL((k_sel_hist_sub<TOP_BITS, U>), dim3(blocks), dim3(SEL_THREADS), lds, st, x, n, prefix_a, m, part);
    hipLaunchKernelGGL((k_sel_pick_sub<false>), dim3(m), dim3(SEL_THREADS), 0, st, part, blocks, m, prefix_a, prefix_b, resid,
                       (float*)nullptr);
    hipLaunchKernelGGL((k_sel_hist_sub<TOP_BITS + SUB_BITS, U>), dim3(blocks), dim3(SEL_THREADS), lds, st, x, n, prefix_b, m,
                       part);
    hipLaunchKernelGGL((k_sel_pick_sub<true>), dim3(m), dim3(SEL_THREADS), 0, st, part, blocks, m, prefix_b, (uint32_t*)nullptr,
                       resid, out);
    return (int)hipGetLastError();
}

}  // extern "C"
