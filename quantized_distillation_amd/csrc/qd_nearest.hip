// qd_nearest.hip -- K4 / K5, nearest-point (non-uniform) quantization (quantization/quant_functions.py:196-290, :531-573): the
// MODE_NEAREST instantiation of the bucket kernels of qd_transform.h, as its own translation unit (parallel build).
#include "qd_transform.h"

namespace {

// ---- K5 at ANY bucket size: the pre-processed forward as a plain stream ------------------------------------------------------
// In the pre-processed call (quant_functions.py:449-469 -> :531-563) the tensor is already scaled: u, alpha, beta are inputs
// and nothing has to be reduced over a bucket.  The bucket kernels that serve odd bucket sizes are built around that
// reduction (k_bucket_chunk(_any): a chunk staged in registers / LDS, three phases per chunk, 2-6 waves per SIMD; at
// bucket 33 / 250 the call ran at 132 / 125 us, 57-61 % of the HBM peak, at 64 / 256 points 210-295 us).  Here every lane
// simply takes float4s of u in memory order -- a contiguous 4 KiB tile per wave, kStreamU float4s per lane in flight -- assigns the four elements (joint search up to 32
// points, the 2048-cell table above that: qd_transform.h), and rescales each with the (alpha, beta) of ITS bucket: the
// float4's first element lies in bucket b0 = e / row (one 32-bit division per lane and tile, then advanced by 256 elements), the others in
// b0 or b0 + 1 -- two loads per array, served by L1 / L2 as neighbouring lanes ask for the same buckets.  One coalesced
// 16-byte load and store per float4, indices as one uint32 (uint8) or two 16-byte stores (int64, exchanged inside the DPP
// row like the vector kernel does) -- whatever the bucket size, from 4 elements up.  The bucket sizes of the vector kernel
// (64 ... 2048, powers of two) stay there: same stream, no per-element bucket choice.
// TWO = false: the bucket size is a multiple of 4, a float4 never straddles two buckets: one (alpha, beta) pair per float4.
constexpr int kStreamU = 4;
template <bool TWO>
__global__ __launch_bounds__(256) void k_nearest_prescaled_stream(KParams p) {
    PointStore Ts;
    PointTable Tc;
    load_points(Tc, Ts, p.pts, p.k, p.fine);
    const PointTable* T = &Tc;
    const float mean = p.mean ? *p.mean : 0.0f;
    const int64_t n4 = p.n >> 2;
    const int64_t row = p.row, last_b = p.nb - 1;
    const f4* src = (const f4*)p.x;
    f4* dst = (f4*)p.out;
    // one CONTIGUOUS 4 KiB tile per wave and step (lane l holds float4 tile * 256 + 64 u + l), as the vector kernels stream;
    // four far-apart streams per lane (float4 i, i + nth, ...) measured 6 % slower on the same scheme in K3
    const int lane = threadIdx.x & 63;
    const int64_t wave = uniform_wave_index(), nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (n4 + 255) >> 8;
    const int64_t dq = 256 / row, dr = 256 % row;                            // 64 float4 = 256 elements further
    const bool small = p.n < ((int64_t)1 << 31);
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t base = (t << 8) + lane;                                // this lane's first float4 of the tile
        int64_t bkt, rem;
        if (small) { const uint32_t e = (uint32_t)(base << 2), rw = (uint32_t)row; bkt = e / rw; rem = e - (uint32_t)bkt * rw; }
        else { bkt = (base << 2) / row; rem = (base << 2) - bkt * row; }
        f4 v[kStreamU];
        float a0[kStreamU], a1[kStreamU], b0[kStreamU], b1[kStreamU];
        int split[kStreamU];                                                 // elements of the float4 that lie in the first bucket
#pragma unroll
        for (int u = 0; u < kStreamU; ++u) {
            const int64_t ii = base + 64 * u;
            const bool live = ii < n4;
            v[u] = __builtin_nontemporal_load(src + (live ? ii : (n4 > 0 ? n4 - 1 : 0)));     // always issued, clamped
            const int64_t bb = bkt < last_b ? bkt : last_b, bn = bkt + 1 < last_b ? bkt + 1 : last_b;
            a0[u] = p.alpha[bb]; b0[u] = p.beta[bb];
            a1[u] = TWO ? p.alpha[bn] : a0[u]; b1[u] = TWO ? p.beta[bn] : b0[u];
            const int64_t left = row - rem;                                  // >= 1
            split[u] = (!TWO || left >= 4) ? 4 : (int)left;
            bkt += dq; rem += dr;
            if (rem >= row) { rem -= row; ++bkt; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kStreamU; ++u) {
            const int64_t ii = base + 64 * u;
            const bool live = ii < n4;
            const float uu[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            int id[4];
            float pt[4];
            if (p.k <= 32) {
                assign_point4(*T->s, p.k, p.assign_mode, uu, id);
#pragma unroll
                for (int c = 0; c < 4; ++c) pt[c] = T->s->pts[id[c]];
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    id[c] = T->fine ? midpoint_index_fine(*T->s, uu[c]) : assign_point(*T->s, p.k, p.assign_mode, uu[c]);
                    pt[c] = T->s->pts[id[c]];
                }
            }
            float o[4], side[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool first = c < split[u];
                float y = pt[c] * (first ? a0[u] : a1[u]);                   // quant_functions.py:142-143,148: mul, add, add
                y = y + (first ? b0[u] : b1[u]);
                y = y + mean;
                o[c] = y;
                side[c] = (float)id[c];
            }
            const int64_t e = ii << 2;
            const bool row_in = !group_any<16>(!live);                       // the 16 float4s of the DPP row are all in range
            if (row_in) {
                const f4 rr = {o[0], o[1], o[2], o[3]};
                if (dst) __builtin_nontemporal_store(rr, dst + ii);          // (q == NULL: indices only, SearchSorted.query)
                store_side4_row<MODE_NEAREST>(p, e, side);
            } else if (live) {
                const f4 rr = {o[0], o[1], o[2], o[3]};
                if (dst) __builtin_nontemporal_store(rr, dst + ii);
                store_side4<MODE_NEAREST>(p, e, side);
            }
        }
    }
    // the n % 4 last elements: one lane
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t e = n4 << 2; e < p.n; ++e) {
            const int64_t bb = e / row < last_b ? e / row : last_b;
            float side = 0.0f;
            const float y = transform<MODE_NEAREST>(p, T, p.x[e], p.alpha[bb], p.beta[bb], mean, 0.0f, side);
            if (p.out) p.out[e] = y;
            store_side1<MODE_NEAREST>(p, e, side);
        }
    }
}

// true if the call was taken by the streaming kernel
bool launch_prescaled_stream(KParams& p, int64_t bucket, hipStream_t st, int& rc) {
    if (!p.prescaled || p.n < 4) return false;
    geometry(p.n, bucket, p.nb, p.row);
    if (p.row < 4) return false;
    if (p.nb <= 1 && p.out) return false;          // one bucket: the single-bucket apply kernel already is a plain stream
    if (((((uintptr_t)p.x) | ((uintptr_t)p.out)) & kDataAlign) != 0) return false;
    if (p.idx && (p.idx_bytes == 8 ? (((uintptr_t)p.idx) & 15) != 0 : (((uintptr_t)p.idx) & 3) != 0)) return false;
    const size_t tb = point_table_bytes(p.k, p.fine);
    int blocks = blocks_for(p.n >> 2, 256 * kStreamU);                      // one 4 KiB tile per wave
    const int cap = p.fine ? kFineBlocksPerCu * num_cus() : (1 << 30);
    if (blocks > cap) blocks = cap;
    if (p.row & 3) hipLaunchKernelGGL(k_nearest_prescaled_stream<true>, dim3(blocks), dim3(256), tb, st, p);
    else hipLaunchKernelGGL(k_nearest_prescaled_stream<false>, dim3(blocks), dim3(256), tb, st, p);
    rc = check_launch();
    return true;
}

}  // namespace

extern "C" {

int qd_nearest_point_f32(const float* x, int prescaled, const float* points, int k, int assign_mode, float* q,
                         void* idx, int idx_bytes, int64_t n, int64_t bucket, float* alpha, float* beta,
                         const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (n < 0 || bucket < 0 || k < 1 || k > kMaxPoints || !points || (n > 0 && !x)) return QD_ERR_INVALID_ARGUMENT;
    if (n > 0 && !q && !(prescaled && idx)) return QD_ERR_INVALID_ARGUMENT;   // indices only: the pre-scaled stream kernel
    if (idx && idx_bytes != 8 && idx_bytes != 1) return QD_ERR_INVALID_ARGUMENT;
    if (idx && idx_bytes == 1 && k > 256) return QD_ERR_INVALID_ARGUMENT;
    if (assign_mode != QD_ASSIGN_DISTANCE && assign_mode != QD_ASSIGN_MIDPOINT) return QD_ERR_INVALID_ARGUMENT;
    if (prescaled && (!alpha || !beta)) return QD_ERR_INVALID_ARGUMENT;
    KParams p = {};
    p.x = x; p.out = q; p.n = n; p.alpha = alpha; p.beta = beta; p.mean = mean;
    p.me = clamp ? max_element : INFINITY;
    p.idx = idx; p.idx_bytes = idx_bytes; p.pts = points; p.k = k; p.assign_mode = assign_mode;
    p.prescaled = prescaled ? 1 : 0;
    p.fine = (k > 32 && assign_mode == QD_ASSIGN_MIDPOINT) ? 1 : 0;      // the fine cell table of qd_transform.h
    int rc = 0;
    if (n > 0 && launch_prescaled_stream(p, bucket, (hipStream_t)stream, rc)) return rc;
    if (n > 0 && !q) return QD_ERR_INVALID_ARGUMENT;                          // (n < 4, a bucket below 4 elements, or a misaligned base)
    return run_transform<MODE_NEAREST>(p, bucket, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
