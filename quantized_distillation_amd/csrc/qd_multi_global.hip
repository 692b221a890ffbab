// qd_multi_global.hip -- multi-tensor uniform quantization WITHOUT buckets (bucket_size=None):
// every tensor is one bucket with its own global min/max.  The reference's per-parameter loop
// (cnn_models/conv_forward_model.py:235-247 with bucket_size=None, e.g. cifar10_test.py:113) costs
// three launches per tensor through qd_uniform_f32 (reduce, fold, apply); here the whole model
// takes three launches in total.  Arithmetic identical to qd_uniform_f32(bucket = 0).
#include "qd_common.h"
#include "../../include/qd_hip.h"

using namespace qd;

namespace {

constexpr int kTile = 1024;     // elements per wave tile: 64 lanes x 4 float4

__device__ __forceinline__ int owner_of(const QdTensorDesc* table, int ntensors, int64_t tile) {
    int lo = 0, hi = ntensors - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_tile <= tile) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// phase 1: per-tile min/max -> part[2*tile], part[2*tile+1]
__global__ __launch_bounds__(256) void k_mg_minmax(const QdTensorDesc* __restrict__ table, int ntensors, int64_t total_tiles, float* part) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = uniform_wave_index();      // scalar: owner_of runs on s_load
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave; t < total_tiles; t += nwaves) {
        const QdTensorDesc d = table[owner_of(table, ntensors, t)];
        const int64_t lo = (t - d.first_tile) * kTile;
        const int64_t hi = lo + kTile < d.n ? lo + kTile : d.n;
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        if (hi - lo == kTile && ((((uintptr_t)d.x) & 15) == 0)) {
            const f4* src = (const f4*)(d.x + lo) + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f4 v = ldg(src + j * 64);           // plain loads: phase 3 re-reads from L2 / MALL
                mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
                mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
                nan |= has_nan4(v);
            }
        } else {
            for (int64_t i = lo + lane; i < hi; i += 64) { const float v = d.x[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); nan |= (v != v); }
        }
        mn = wave_min(mn); mx = wave_max(mx);
        if (group_any<64>(nan)) { mn = NAN; mx = NAN; }       // NaN poisons the tile and, in phase 2, the tensor
        if (lane == 0) { part[2 * t] = mn; part[2 * t + 1] = mx; }
    }
}

// phase 2: one block per tensor folds its tiles into (alpha, beta); the 1e-10 guard on the device
__global__ __launch_bounds__(256) void k_mg_fold(const QdTensorDesc* __restrict__ table, int ntensors, int64_t total_tiles,
                                                 const float* part, float* ab /* [ntensors][2] */) {
    __shared__ float red[32];
    const int ti = blockIdx.x;
    const int64_t t0 = table[ti].first_tile;
    const int64_t t1 = ti + 1 < ntensors ? table[ti + 1].first_tile : total_tiles;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int64_t t = t0 + threadIdx.x; t < t1; t += 256) {
        const float pm = part[2 * t];
        nan |= (pm != pm);
        mn = fminf(mn, pm); mx = fmaxf(mx, part[2 * t + 1]);
    }
    block_minmax(mn, mx, red);
    if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }
    if (threadIdx.x == 0) {
        float a, b;
        alpha_beta(mn, mx, a, b);
        ab[2 * ti] = a; ab[2 * ti + 1] = b;
    }
}

// phase 3: apply with the tensor's single (alpha, beta)
__global__ __launch_bounds__(256) void k_mg_apply(const QdTensorDesc* __restrict__ table, int ntensors, int64_t total_tiles,
                                                  const float* ab, float sm1) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = uniform_wave_index();      // scalar: owner_of runs on s_load
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t t = wave; t < total_tiles; t += nwaves) {
        const int ti = owner_of(table, ntensors, t);
        const QdTensorDesc d = table[ti];
        const float a = ab[2 * ti], b = ab[2 * ti + 1];
        const int64_t lo = (t - d.first_tile) * kTile;
        const int64_t hi = lo + kTile < d.n ? lo + kTile : d.n;
        float lev;
        if (hi - lo == kTile && (((((uintptr_t)d.x) | ((uintptr_t)d.q)) & 15) == 0)) {
            const f4* src = (const f4*)(d.x + lo) + lane;
            f4* dst = (f4*)(d.q + lo) + lane;
            f4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ldg_nt(src + j * 64);
            __builtin_amdgcn_sched_barrier(0);          // all four loads in flight before the first use
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f4 r;
                r.x = qdq(v[j].x, a, b, sm1, 0.0f, lev); r.y = qdq(v[j].y, a, b, sm1, 0.0f, lev);
                r.z = qdq(v[j].z, a, b, sm1, 0.0f, lev); r.w = qdq(v[j].w, a, b, sm1, 0.0f, lev);
                stg_nt(r, dst + j * 64);
            }
        } else {
            for (int64_t i = lo + lane; i < hi; i += 64) d.q[i] = qdq(d.x[i], a, b, sm1, 0.0f, lev);
        }
    }
}

}  // namespace

extern "C" {

int64_t qd_multi_global_plan(QdTensorDesc* host_table, int ntensors) {
    if (!host_table || ntensors < 0) return -1;
    int64_t tiles = 0;
    for (int i = 0; i < ntensors; ++i) {
        host_table[i].first_tile = tiles;
        tiles += (host_table[i].n + kTile - 1) / kTile;
    }
    return tiles;
}

int qd_multi_uniform_global_f32(const QdTensorDesc* table, int ntensors, int64_t total_tiles, int levels,
                                float* alpha_beta, void* workspace, size_t workspace_bytes, void* stream) {
    if (!table || ntensors <= 0 || total_tiles < 0 || levels < 2 || !alpha_beta) return QD_ERR_INVALID_ARGUMENT;
    if (total_tiles == 0) return 0;
    if (!workspace || (((uintptr_t)workspace) & 15) || workspace_bytes < (size_t)total_tiles * 2 * sizeof(float))
        return QD_ERR_WORKSPACE_TOO_SMALL;
    hipStream_t st = (hipStream_t)stream;
    float* part = (float*)workspace;
    int64_t b = (total_tiles + 3) / 4;
    const int blocks = (int)(b < (1 << 20) ? b : (1 << 20));
    hipLaunchKernelGGL(k_mg_minmax, dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, part);
    hipLaunchKernelGGL(k_mg_fold, dim3(ntensors), dim3(256), 0, st, table, ntensors, total_tiles, part, alpha_beta);
    hipLaunchKernelGGL(k_mg_apply, dim3(blocks), dim3(256), 0, st, table, ntensors, total_tiles, alpha_beta,
                       (float)(levels - 1));
    return (int)hipGetLastError();
}

}  // extern "C"
