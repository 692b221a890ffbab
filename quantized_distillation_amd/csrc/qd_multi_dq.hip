// qd_multi_dq.hip -- multi-tensor kernels for the differentiable-quantization step (gfx950).
//
// The reference's optimize_quantization_points loop (cnn_models/conv_forward_model.py:524-545)
// calls, for EVERY parameter tensor EVERY step,
//     p_quantized.data = quantizationFunctions[i].forward(None, points[i].data)      (:532)
//     points[i].grad.data = quantizationFunctions[i].backward(p.grad.data)[1]        (:545)
// With 60 tensors that is 120+ kernel launches per step from Python; the two entry points here
// do each sweep in ONE launch over a device table of per-tensor descriptors.  Arithmetic is
// identical to qd_nearest_point_f32(prescaled, QD_ASSIGN_MIDPOINT) / qd_point_grad_f32.
#include "qd_common.h"
#include "../../include/qd_hip.h"

using namespace qd;

namespace {

constexpr int kMaxK = 64;

__device__ __forceinline__ int find_owner(const QdDiffQuantDesc* table, int ntensors, int64_t item, bool by_block) {
    int lo = 0, hi = ntensors - 1;                     // last tensor whose prefix <= item
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        const int64_t first = by_block ? table[mid].first_block : table[mid].first_tile;
        if (first <= item) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// forward: a tile = 4 buckets of one tensor = one wave iteration; a DPP row owns a bucket
template <int ROW>
__global__ __launch_bounds__(256) void k_multi_nearest(const QdDiffQuantDesc* __restrict__ table, int ntensors, int64_t total_tiles,
                                                       int64_t bucket, const float* points, int k) {
    __shared__ float s_pts[4][kMaxK];
    __shared__ float s_mid[4][kMaxK];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int sub = lane >> 4, l = lane & 15;
    const int64_t wave = uniform_wave_index();      // scalar: find_owner runs on s_load
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave; t < total_tiles; t += nwaves) {
        const int ti = find_owner(table, ntensors, t, false);          // wave-uniform
        const QdDiffQuantDesc d = table[ti];
        // this tensor's points and fp32 midpoints (quant_functions.py:533) into the wave's LDS slot;
        // LDS operations of one wave complete in order, the barriers only stop compiler reordering
        __builtin_amdgcn_wave_barrier();
        if (lane < k) s_pts[w][lane] = points[(int64_t)ti * k + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane + 1 < k) {
            float df = s_pts[w][lane + 1] - s_pts[w][lane];
            df = df / 2.0f;
            s_mid[w][lane] = s_pts[w][lane] + df;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int64_t row = d.n < bucket ? d.n : bucket;
        const int64_t nb = (d.n + row - 1) / row;
        const int64_t bkt = (t - d.first_tile) * 4 + sub;
        if (bkt >= nb) continue;
        const int64_t lo = bkt * row;
        const int64_t hi = lo + row < d.n ? lo + row : d.n;
        const float a = ldg(d.alpha + bkt), b = ldg(d.beta + bkt);
        const bool fast = ROW > 0 && (hi - lo) == ROW &&
                          (((((uintptr_t)d.u) | ((uintptr_t)d.q)) & 15) == 0) && ((((uintptr_t)d.idx) & 3) == 0);
        if (fast) {
            constexpr int V = ROW > 0 ? ROW / 64 : 1;
            const f4* src = (const f4*)(d.u + lo) + l;
            f4* dst = (f4*)(d.q + lo) + l;
            f4 v[V];
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = ldg_nt(src + j * 16);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                int id[4];
                f4 r;
#define QD_ONE(c, n_)                                                         \
                {                                                             \
                    id[n_] = count_before<true>(s_mid[w], k - 1, v[j].c);     \
                    float y = s_pts[w][id[n_]] * a;                           \
                    y = y + b;                                                \
                    r.c = y + 0.0f;                                           \
                }
                QD_ONE(x, 0) QD_ONE(y, 1) QD_ONE(z, 2) QD_ONE(w, 3)
#undef QD_ONE
                stg_nt(r, dst + j * 16);
                const uint32_t pk = (uint32_t)id[0] | ((uint32_t)id[1] << 8) | ((uint32_t)id[2] << 16) | ((uint32_t)id[3] << 24);
                stg(pk, (uint32_t*)(d.idx + lo + ((int64_t)(j * 16 + l) << 2)));
            }
        } else {
            for (int64_t i = lo + l; i < hi; i += 16) {
                const int id = count_before<true>(s_mid[w], k - 1, d.u[i]);
                float y = s_pts[w][id] * a;
                y = y + b;
                d.q[i] = y + 0.0f;
                d.idx[i] = (uint8_t)id;
            }
        }
    }
}

// ---- backward: grad of the points of every tensor, one launch + one fold --------------------------------------------
// Work unit: a GRADIENT TILE = 1024 consecutive elements of one tensor (a wave iteration: four 1 KiB rows of gradient, four
// 256 B rows of uint8 indices).  QdDiffQuantDesc.first_block is the prefix of FULL gradient tiles over the tensors; the
// tensors' full tiles form ONE sequence 0 ... T - 1, and the main grid strides over it as qd_point_grad_f32's does over its
// one tensor: wave g of the W = 4 B waves takes tiles g, g + W, g + 2 W, ... -- every wave streams the same number of bytes
// whatever the tensor sizes are, and the whole grid moves through memory front to back together.  What is left of a tensor
// after its full tiles (n mod 1024 elements: every bias and batch-norm vector is only that) goes to EXTRA blocks behind the
// main grid, one wave per tensor: short-lived waves that run beside the sweep instead of inside one of its waves.
//   round 4: every tensor its own ceil(n / 128 Ki) blocks -- 631 blocks of 65 Ki ... 128 Ki elements on the WRN-16-22 shape
//            list, 60 tensors swept concurrently: 71.8 us = 72.1 % of the HBM peak; 47.7 us on the 1 M-parameter CIFAR student
//   round 5 (profiles/r05_ab_k6m.txt): 512 equal blocks, each its own CONTIGUOUS piece of the sequence: 73.5 us -- 512 separate
//            streaming windows are worse than 60; the wave-strided sequence WITH the partial tiles in it: 74.2 us (on ONE 64 Mi
//            tensor 54.7 us, the single-tensor kernel's time, against 65.2 us for round 4's) -- 47 of the 60 WRN tensors end
//            in a partial tile, 43 are nothing else, and each cost one of the 40-tile waves a slow masked tile + two flushes
// Partial sums: a wave keeps its bins while its tiles stay in one tensor and writes them out -- one row of k floats -- when they
// move on to the next one.  Tensor ti with c full tiles from f0 on is visited by the waves (f0 + i) mod W, i < min(c, W), each
// once: wave g writes row first_row[ti] + ((g - f0) mod W) and the extra wave of the tensor row first_row[ti] + min(c, W), so a
// tensor owns min(c, W) + 1 rows (round 6: it owned W + 1 whatever its size -- 105 MB of scratch for 200 small tensors at
// k = 64) and the fold reads them front to back.  Which rows exist is a function of the table alone, so nothing needs zeroing.
// Fixed tile -> wave assignment, fixed fold order: deterministic.
constexpr int kGradTile = 1024;
constexpr int kGradBlocks = 512;                       // the main grid: 2048 waves x 4 independent 1 KiB streams, the shape that
constexpr int64_t kGradWaves = 4 * kGradBlocks;        // measured best for qd_point_grad_f32; blocks beyond the last tile return at once

// T is not an argument of the entry point (the host holds no copy of the device table): the last tensor's prefix + its tiles
__device__ __forceinline__ int64_t total_grad_tiles(const QdDiffQuantDesc* table, int ntensors) {
    return table[ntensors - 1].first_block + table[ntensors - 1].n / kGradTile;
}

// last tensor whose tile prefix is <= t, for a whole wave at once: every lane looks at one tensor's prefix, a ballot counts
// (one memory round trip per 64 tensors instead of log2(ntensors) dependent scalar loads before the wave's first tile)
__device__ __forceinline__ int owner_of_tile(const QdDiffQuantDesc* table, int ntensors, int64_t t, int lane) {
    int cnt = 0;
    for (int base = 0; base < ntensors; base += 64) {
        const int i = base + lane;
        const bool le = i < ntensors && table[i].first_block <= t;
        cnt += __popcll(__ballot(le));
    }
    return __builtin_amdgcn_readfirstlane(cnt - 1);
}

// KR > 0: k <= KR bins in registers (compare-select-add); KR == 0: lane-private LDS columns [k][256]
template <int KR>
__global__ __launch_bounds__(256) void k_multi_point_grad(const QdDiffQuantDesc* __restrict__ table, int ntensors,
                                                          int64_t bucket, int row_shift, int k, float* part /* [rows][k] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // KR == 0: [k][256]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    constexpr int B = kGradBlocks;
    constexpr int64_t W = kGradWaves;
    float acc[KR > 0 ? KR : 1];
    float* col = lds + threadIdx.x;                                    // this lane's column; a wave owns columns 64 w ... 64 w + 63
    auto clear = [&]() {
        if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) acc[j] = 0.0f;
        } else {
            for (int j = 0; j < k; ++j) col[j * 256] = 0.0f;
        }
    };
    auto add = [&](int id, float m) {
        if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) acc[j] += (id == j) ? m : 0.0f;
        } else {
            col[id * 256] += m;                                        // private column: plain LDS read-add-write
        }
    };
    auto flush = [&](const QdDiffQuantDesc& dd, int64_t r) {           // this wave's sums for tensor dd -> its row for wave r (W: the extra wave), bins cleared
        const int64_t full = dd.n / kGradTile;
        int64_t c = r - dd.first_block % W;                            // (r - f0) mod W for a wave of the main grid
        c = c < 0 ? c + W : c;
        c = r == W ? (full < W ? full : W) : c;
        float* row = part + (dd.first_row + c) * k;
        if (KR > 0) {
#pragma unroll
            for (int j = 0; j < (KR > 0 ? KR : 1); ++j) {
                // wave_sum() as lane 0 sees it -- (row 0 + row 1) + (row 2 + row 3) -- with the four row sums read as scalars
                const float rs = row16_sum(acc[j]);
                const float s0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 0));
                const float s1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 16));
                const float s2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 32));
                const float s3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rs), 48));
                const float sum = (s0 + s1) + (s2 + s3);
                if (lane == 0 && j < k) row[j] = sum;
            }
        } else {
            // LDS operations of one wave complete in order; the barriers only pin the compiler.  Lane j folds bin j's 64 columns
            // of this wave (rotated start: bank-conflict free) in a fixed order.
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < k) {
                const float* base = lds + lane * 256 + 64 * w;
                float s0 = 0.0f, s1 = 0.0f;
                for (int c = 0; c < 64; c += 2) { s0 += base[(c + lane) & 63]; s1 += base[(c + 1 + lane) & 63]; }
                row[lane] = s0 + s1;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        clear();
    };
    auto scalar_span = [&](const QdDiffQuantDesc& d, int64_t lo, int64_t hi) {        // element by element (a view at an odd offset)
        const bool single = d.n <= bucket;
        for (int64_t e = lo + lane; e < hi; e += 64)
            add((int)d.idx[e], d.grad[e] * d.alpha[single ? 0 : (e >> row_shift)]);
    };

    if ((int)blockIdx.x >= B) {
        // ---- the extra waves: what is left of tensor ti after its full tiles
        const int ti = ((int)blockIdx.x - B) * 4 + w;
        if (ti >= ntensors) return;
        const QdDiffQuantDesc d = table[ti];
        const int64_t rem = d.n % kGradTile;
        if (rem == 0) return;
        clear();
        const int64_t e0 = d.n - rem;
        const bool single = d.n <= bucket;
        const bool aligned = ((((uintptr_t)d.grad) & 15) == 0) && ((((uintptr_t)d.idx) & 3) == 0);
        if (aligned) {
            // every lane issues its loads up front; the float4 that straddles the end is assembled from scalar loads; elements past
            // the end add an exact +0 to bin 0 (not 0 x alpha, which is NaN for the alpha = inf of a bucket that holds an infinity)
            const f4* g4 = (const f4*)(d.grad + e0) + lane;
            const uint32_t* i4 = (const uint32_t*)(d.idx + e0) + lane;
            const int64_t left = rem - 4 * lane;                        // elements from this lane's first float4 to the end
            f4 gv[4]; uint32_t pk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t lu = left - 256 * u;
                const float* gs = (const float*)(g4 + 64 * u);
                const uint8_t* is = (const uint8_t*)(i4 + 64 * u);
                const float al = lu > 0 ? ldg(d.alpha + (single ? 0 : ((e0 + 256 * u + 4 * lane) >> row_shift))) : 0.0f;
                f4 gq = {0.0f, 0.0f, 0.0f, 0.0f};
                uint32_t pq = 0;
                if (lu >= 4) {
                    gq = ldg_nt(g4 + 64 * u);
                    pq = ldg_nt(i4 + 64 * u);
                } else {
                    if (lu > 0) { gq.x = gs[0]; pq |= (uint32_t)is[0]; }
                    if (lu > 1) { gq.y = gs[1]; pq |= (uint32_t)is[1] << 8; }
                    if (lu > 2) { gq.z = gs[2]; pq |= (uint32_t)is[2] << 16; }
                }
                gv[u].x = lu > 0 ? gq.x * al : 0.0f; gv[u].y = lu > 1 ? gq.y * al : 0.0f;     // one fp32 multiply each, quant_functions.py:495
                gv[u].z = lu > 2 ? gq.z * al : 0.0f; gv[u].w = lu > 3 ? gq.w * al : 0.0f;
                pk[u] = pq;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                add(pk[u] & 255, gv[u].x);
                add((pk[u] >> 8) & 255, gv[u].y);
                add((pk[u] >> 16) & 255, gv[u].z);
                add(pk[u] >> 24, gv[u].w);
            }
        } else {
            scalar_span(d, e0, d.n);
        }
        flush(d, W);
        return;
    }

    // ---- the main grid: full tiles only
    const int64_t T = total_grad_tiles(table, ntensors);
    const int64_t g = (int64_t)blockIdx.x * 4 + w;                     // a block's waves take four adjacent tiles
    if (g >= T) return;
    int ti = owner_of_tile(table, ntensors, g, lane);                  // once per wave
    clear();
    QdDiffQuantDesc d = table[ti];
    int64_t next_first = ti + 1 < ntensors ? table[ti + 1].first_block : T;
    for (int64_t t = g; t < T; t += W) {
        if (t >= next_first) {                                          // this wave's tiles have moved on to a later tensor
            flush(d, g);
            do {
                ++ti;
                next_first = ti + 1 < ntensors ? table[ti + 1].first_block : T;
            } while (t >= next_first);
            d = table[ti];
        }
        const int64_t e0 = (t - d.first_block) * kGradTile;
        const bool aligned = ((((uintptr_t)d.grad) & 15) == 0) && ((((uintptr_t)d.idx) & 3) == 0);
        if (aligned) {                                                  // four (gradient float4, four indices, alpha) loads up front
            const f4* g4 = (const f4*)(d.grad + e0) + lane;
            const uint32_t* i4 = (const uint32_t*)(d.idx + e0) + lane;
            f4 gv[4]; uint32_t pk[4]; float a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                gv[u] = ldg_nt(g4 + 64 * u);
                pk[u] = ldg_nt(i4 + 64 * u);
                a[u] = ldg(d.alpha + (d.n <= bucket ? 0 : ((e0 + 256 * u + 4 * lane) >> row_shift)));      // n <= bucket: one alpha
            }
            __builtin_amdgcn_sched_barrier(0);                          // keep the loads together (not sunk to their uses)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                add(pk[u] & 255, gv[u].x * a[u]);                       // one fp32 multiply each, quant_functions.py:495
                add((pk[u] >> 8) & 255, gv[u].y * a[u]);
                add((pk[u] >> 16) & 255, gv[u].z * a[u]);
                add(pk[u] >> 24, gv[u].w * a[u]);
            }
        } else {
            scalar_span(d, e0, e0 + kGradTile);
        }
    }
    flush(d, g);
}

// backward stage 2: block (tensor, bin) folds the rows that tensor's waves wrote, front to back.  Up to 2048 + 1 rows per
// tensor: 256 threads, eight independent loads in flight each (one dependent load per row and thread was 32 round trips =
// ~12 us on the WRN shape list, more than the sweep gained).
__global__ __launch_bounds__(256) void k_multi_point_grad_final(const QdDiffQuantDesc* __restrict__ table, int ntensors,
                                                                int k, const float* part,
                                                                float* grad_points /* [ntensors][k] */) {
    __shared__ double s_w[4];
    constexpr int64_t W = kGradWaves;
    const int ti = blockIdx.x / k, j = blockIdx.x % k;
    const int64_t full = table[ti].n / kGradTile;
    const int64_t rows = full < W ? full : W;                           // row i: wave (f0 + i) mod W, the i-th of the waves that had a tile of this tensor
    const float* base = part + table[ti].first_row * k + j;
    auto row = [&](int64_t i) { return base[i * k]; };
    double acc = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 7 * 256 < rows; i += 8 * 256) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = row(i + u * 256);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += (double)v[u];
    }
    for (; i < rows; i += 256) acc += (double)row(i);
    if (threadIdx.x == 0 && (table[ti].n % kGradTile) != 0) acc += (double)base[rows * k];   // the extra wave's row
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) grad_points[(int64_t)ti * k + j] = (float)((s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

}  // namespace

extern "C" {

int64_t qd_multi_dq_plan(QdDiffQuantDesc* host_table, int ntensors, int64_t bucket, int64_t* total_blocks_out) {
    if (!host_table || ntensors <= 0 || bucket <= 0 || !total_blocks_out) return -1;
    int64_t tiles = 0, gtiles = 0, rows = 0;
    for (int i = 0; i < ntensors; ++i) {
        const int64_t n = host_table[i].n;
        const int64_t row = n < bucket ? (n > 0 ? n : 1) : bucket;
        const int64_t nb = n > 0 ? (n + row - 1) / row : 0;
        host_table[i].first_tile = tiles;
        host_table[i].first_block = gtiles;              // prefix of FULL 1024-element gradient tiles (k_multi_point_grad)
        host_table[i].first_row = rows;                  // prefix of partial rows: min(full tiles, 2048) + 1 per tensor
        const int64_t full = n > 0 ? n / kGradTile : 0;
        tiles += (nb + 3) / 4;
        gtiles += full;
        rows += (full < kGradWaves ? full : kGradWaves) + 1;
    }
    *total_blocks_out = rows;                            // partial rows of the gradient sweep, see k_multi_point_grad
    return tiles;
}

int qd_multi_nearest_f32(const QdDiffQuantDesc* table, int ntensors, int64_t total_tiles, int64_t bucket,
                         const float* points, int k, void* stream) {
    if (!table || ntensors <= 0 || total_tiles < 0 || bucket <= 0 || !points || k < 1 || k > kMaxK)
        return QD_ERR_INVALID_ARGUMENT;
    if (total_tiles == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    int64_t b = (total_tiles + 3) / 4;
    const int blocks = (int)(b < (1 << 20) ? b : (1 << 20));
    if (bucket == 256) hipLaunchKernelGGL((k_multi_nearest<256>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, points, k);
    else if (bucket == 128) hipLaunchKernelGGL((k_multi_nearest<128>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, points, k);
    else if (bucket == 64) hipLaunchKernelGGL((k_multi_nearest<64>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, points, k);
    else hipLaunchKernelGGL((k_multi_nearest<0>), dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, bucket, points, k);
    return (int)hipGetLastError();
}

int qd_multi_point_grad_f32(const QdDiffQuantDesc* table, int ntensors, int64_t total_blocks, int64_t bucket, int k,
                            float* grad_points, void* workspace, size_t workspace_bytes, void* stream) {
    if (!table || ntensors <= 0 || total_blocks <= 0 || bucket <= 0 || (bucket & (bucket - 1)) || k < 1 || k > kMaxK ||
        !grad_points)
        return QD_ERR_INVALID_ARGUMENT;
    // total_blocks is what qd_multi_dq_plan wrote: between 1 and 2048 + 1 partial rows per tensor
    if (total_blocks < ntensors || total_blocks > (kGradWaves + 1) * (int64_t)ntensors) return QD_ERR_INVALID_ARGUMENT;
    if (!workspace || (((uintptr_t)workspace) & 15) || workspace_bytes < (size_t)total_blocks * k * sizeof(float))
        return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    int row_shift = 0;
    while (((int64_t)1 << row_shift) < bucket) ++row_shift;
    float* part = (float*)workspace;
    // grid: the 512 blocks of the sweep over the full tiles (a wave beyond the last tile returns at once) + one wave per
    // tensor for what is left after them
    const unsigned grid = (unsigned)(kGradBlocks + (ntensors + 3) / 4);
    if (k <= 4)
        hipLaunchKernelGGL((k_multi_point_grad<4>), dim3(grid), dim3(256), 0, st, table, ntensors, bucket, row_shift, k, part);
    else
        hipLaunchKernelGGL((k_multi_point_grad<0>), dim3(grid), dim3(256), (size_t)k * 256 * sizeof(float), st, table, ntensors,
                           bucket, row_shift, k, part);
    hipLaunchKernelGGL(k_multi_point_grad_final, dim3((unsigned)(ntensors * k)), dim3(256), 0, st, table, ntensors, k, part,
                       grad_points);
    return (int)hipGetLastError();
}

}  // extern "C"
