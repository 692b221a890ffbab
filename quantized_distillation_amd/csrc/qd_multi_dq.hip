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

// backward stage 1: block -> tensor; lane-private LDS columns bins[k][256], plain read-add-write
__global__ __launch_bounds__(256) void k_multi_point_grad(const QdDiffQuantDesc* __restrict__ table, int ntensors, int64_t bucket,
                                                          int row_shift, int k, float* part /* [blocks][k] */) {
    extern __shared__ __attribute__((aligned(16))) float lds[];        // [k][256]
    const int ti = find_owner(table, ntensors, blockIdx.x, true);
    const QdDiffQuantDesc d = table[ti];
    const int64_t nblk = (ti + 1 < ntensors ? table[ti + 1].first_block : (int64_t)gridDim.x) - d.first_block;
    const int64_t bl = blockIdx.x - d.first_block;
    for (int j = threadIdx.x; j < k * 256; j += 256) lds[j] = 0.0f;
    __syncthreads();
    float* col = lds + threadIdx.x;
    const bool single = d.n <= bucket;                  // one bucket: alpha[0] for every element
    const float a_single = single ? d.alpha[0] : 0.0f;
    const int64_t tid = bl * 256 + threadIdx.x, nth = nblk * 256;
    const bool vec = (((((uintptr_t)d.grad)) & 15) == 0) && ((((uintptr_t)d.idx) & 3) == 0);
    int64_t done = 0;
    if (vec) {
        const int64_t n4 = d.n >> 2;
        auto add4 = [&](const f4& gv, uint32_t pk, float a) {
            col[(pk & 255) * 256] += gv.x * a;           // one fp32 multiply each, quant_functions.py:495
            col[((pk >> 8) & 255) * 256] += gv.y * a;
            col[((pk >> 16) & 255) * 256] += gv.z * a;
            col[(pk >> 24) * 256] += gv.w * a;
        };
        constexpr int U = 4;                              // float4 in flight per lane (see qd_point_grad_f32's grid note)
        int64_t i = tid;
        for (; i + (U - 1) * nth < n4; i += U * nth) {
            f4 gv[U]; uint32_t pk[U]; float a[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t iu = i + u * nth;
                gv[u] = ldg_nt((const f4*)d.grad + iu);
                pk[u] = ldg_nt((const uint32_t*)d.idx + iu);
                a[u] = single ? a_single : ldg(d.alpha + ((iu << 2) >> row_shift));
            }
            __builtin_amdgcn_sched_barrier(0);            // keep the loads together (not sunk to their uses)
#pragma unroll
            for (int u = 0; u < U; ++u) add4(gv[u], pk[u], a[u]);
        }
        for (; i < n4; i += nth) {
            const f4 gv = ldg_nt((const f4*)d.grad + i);
            const uint32_t pk = ldg_nt((const uint32_t*)d.idx + i);
            add4(gv, pk, single ? a_single : ldg(d.alpha + ((i << 2) >> row_shift)));
        }
        done = n4 << 2;
    }
    for (int64_t e = done + tid; e < d.n; e += nth)
        col[(int)d.idx[e] * 256] += d.grad[e] * (single ? a_single : d.alpha[e >> row_shift]);
    __syncthreads();
    for (int t = threadIdx.x; t < ((k * 4 + 3) & ~3); t += 256) {        // 4 threads per bin, fixed fold
        const int j = t >> 2, qd4 = t & 3;
        float acc = 0.0f;
        if (j < k)
            for (int c = 0; c < 64; ++c) acc += lds[j * 256 + qd4 * 64 + ((c + j) & 63)];
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (qd4 == 0 && j < k) part[(int64_t)blockIdx.x * k + j] = acc;
    }
}

// backward stage 2: block (tensor, bin) folds that tensor's partial rows in a fixed order
__global__ __launch_bounds__(64) void k_multi_point_grad_final(const QdDiffQuantDesc* __restrict__ table, int ntensors,
                                                               int64_t total_blocks, int k, const float* part,
                                                               float* grad_points /* [ntensors][k] */) {
    const int ti = blockIdx.x / k, j = blockIdx.x % k;
    const int64_t b0 = table[ti].first_block;
    const int64_t b1 = ti + 1 < ntensors ? table[ti + 1].first_block : total_blocks;
    double acc = 0.0;
    for (int64_t bI = b0 + threadIdx.x; bI < b1; bI += 64) acc += (double)part[bI * k + j];
    acc = wave_sum_d(acc);
    if (threadIdx.x == 0) grad_points[(int64_t)ti * k + j] = (float)acc;
}

}  // namespace

extern "C" {

int64_t qd_multi_dq_plan(QdDiffQuantDesc* host_table, int ntensors, int64_t bucket, int64_t* total_blocks_out) {
    if (!host_table || ntensors <= 0 || bucket <= 0 || !total_blocks_out) return -1;
    int64_t tiles = 0, blocks = 0;
    for (int i = 0; i < ntensors; ++i) {
        const int64_t n = host_table[i].n;
        const int64_t row = n < bucket ? (n > 0 ? n : 1) : bucket;
        const int64_t nb = n > 0 ? (n + row - 1) / row : 0;
        host_table[i].first_tile = tiles;
        host_table[i].first_block = blocks;
        tiles += (nb + 3) / 4;
        // ~512 elements per thread: a whole model then runs on a few hundred blocks, each lane streaming four
        // independent float4 at a time -- the grid shape that measured best for qd_point_grad_f32
        int64_t nblk = (n + 256 * 4 * 128 - 1) / (256 * 4 * 128);
        if (nblk < 1) nblk = 1;
        if (nblk > 512) nblk = 512;
        blocks += nblk;
    }
    *total_blocks_out = blocks;
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
    if (!workspace || (((uintptr_t)workspace) & 15) || workspace_bytes < (size_t)total_blocks * k * sizeof(float))
        return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    int row_shift = 0;
    while (((int64_t)1 << row_shift) < bucket) ++row_shift;
    float* part = (float*)workspace;
    hipLaunchKernelGGL(k_multi_point_grad, dim3((unsigned)total_blocks), dim3(256), (size_t)k * 256 * sizeof(float), st,
                       table, ntensors, bucket, row_shift, k, part);
    hipLaunchKernelGGL(k_multi_point_grad_final, dim3((unsigned)(ntensors * k)), dim3(64), 0, st, table, ntensors,
                       total_blocks, k, part, grad_points);
    return (int)hipGetLastError();
}

}  // extern "C"
