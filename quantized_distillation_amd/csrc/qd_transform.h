// qd_transform.h -- the MODE-templated half of the gfx950 fake-quantization kernels: the per-element transforms, every
// bucket kernel family (vector, chunk, chunk_any, one wave per bucket, lane groups, block per bucket), the single-bucket
// kernels (three launches / one launch) and their launchers, ending in run_transform<MODE>().
//
// Included by THREE translation units, one per mode, so that hipcc builds them in parallel (as one file the kernels took
// 55 s to compile):
//     qd_kernels.hip   MODE_QDQ      qd_uniform_f32, the multi-tensor launch, and every non-templated kernel (K3, K6, K7, K8, ...)
//     qd_scale.hip     MODE_SCALE    qd_scale_down_f32
//     qd_nearest.hip   MODE_NEAREST  qd_nearest_point_f32
// Everything here sits in an anonymous namespace: each unit gets its own copy of the few non-templated helpers (min/max
// partials, launch helpers) and its own barrier slots and epoch counter for the one-launch kernel -- no device symbol
// crosses a translation unit (no -fgpu-rdc).  The only shared state is the host-side switch qd_fused_mode_state.
//
// Reference being replaced (paths relative to the reference root):
//   quantization/quant_functions.py:56-152   ScalingFunction.scale_down / inv_scale_down
//   quantization/quant_functions.py:155-194  uniformQuantization
//   quantization/quant_functions.py:196-290  nonUniformQuantization (+ SearchSorted :509-573)
//   quantization/help_functions.py:67-94     create_bucket_tensor (semantics folded in)
//
// No CUDA compatibility layer, no dual code paths: gfx950 only.
#pragma once

#include "qd_common.h"
#include "../../include/qd_hip.h"

#include <math.h>
#include <atomic>
#include <type_traits>
#include <stdlib.h>

using namespace qd;

// qd_set_single_fused_mode(): 0 = always the three-launch path, 1 = default, 2 / 3 / 4 = test hooks (see qd_hip.h).  Defined in
// qd_kernels.hip; hidden: not part of the ABI.
extern "C" __attribute__((visibility("hidden"))) int qd_fused_mode_state;

namespace {

enum Mode { MODE_QDQ = 0, MODE_SCALE = 1, MODE_NEAREST = 2 };

constexpr int kMaxPoints = 1024;        // LDS table of quantization points
constexpr int kPartialBlocks = 1024;    // stage-1 blocks of every two-stage reduction

// Alignment the 16-byte (float4) global accesses of the single-tensor entry points need from their fp32 data pointers:
// FOUR bytes.  global_load / global_store_dwordx4 take any dword-aligned address (the HSA queues run in unaligned access
// mode), and a wave's 1 KiB per instruction covers the same lines either way: a tensor view that starts 4, 8 or 12 bytes
// into a 16-byte granule runs at the speed of an aligned one (bucket 256, 64 Mi elements: 87.3 us against 85.3 us; 214 us
// on the scalar two-pass kernels that the 16-byte requirement of round 1 sent it to).  tests/test_hip_parity.py::
// test_views_at_every_4_byte_offset runs every entry point on such views.  (int64 index outputs and the workspace keep their
// 16-byte requirement; the multi-tensor tables are built from 256-byte-aligned slots anyway.)
constexpr uintptr_t kDataAlign = 3;

struct KParams {
    const float* x;      // input [n]
    float* out;          // QDQ/NEAREST: [n]; SCALE: [padded]
    int64_t n;
    int64_t row;         // elements per bucket row (>= 1)
    int64_t nb;          // number of buckets
    float* alpha;        // [nb] outputs, or inputs when prescaled
    float* beta;
    const float* mean;   // device scalar or null
    float me;            // clamp limit, +inf when off
    float sm1;           // levels - 1
    uint8_t* lev8;       // optional [n] level index
    void* idx;           // NEAREST: optional index output
    int idx_bytes;       // 8 or 1
    const float* pts;    // NEAREST: [k] sorted points
    int k;
    int assign_mode;
    int prescaled;       // NEAREST: x is already u, alpha/beta are inputs
    int fine;            // NEAREST, midpoint rule, k > 32: the kernel builds and uses the fine cell table (see PointStore)
    int stochastic;
    uint64_t seed;
    int64_t nvec;        // number of leading full buckets handled by the vector path
};

// LDS-resident point table (+ midpoints, quant_functions.py:533)
constexpr int kCells = 256;             // uniform grid over [0,1] that narrows the midpoint search (k > 32)

// The table lives at the START of the kernel's dynamic LDS (every nearest-point launch asks for point_table_bytes(k) more),
// sized by the number of points: 256 bytes up to 32 points -- every configuration of the reference's scripts -- instead of
// a static 10.3 KB for the 1024 points the ABI allows.  (The static table left the chunk kernels 6 / 5 blocks per CU, bound
// by LDS; the point search is a chain of dependent LDS reads, so resident waves are what hides it.)
// (pointers in the LDS address space: 32-bit addresses and ds_read straight away, instead of 64-bit generic pointers that
// the compiler first has to prove to be LDS -- the k = 256 search is VALU- and bank-conflict-bound, every instruction counts)
typedef __attribute__((address_space(3))) float lds_f32;
typedef __attribute__((address_space(3))) int lds_i32;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));      // (a native vector: HIP's uint2 class has no LDS-qualified operators)
typedef __attribute__((address_space(3))) u32x2 lds_u2;
struct PointStore {
    lds_f32* pts;                       // [cap]
    lds_f32* mid;                       // [cap]
    lds_i32* start;                     // [kCells + 4], k > 32 only: start[c] = #{ j : cell(mid_j) < c }, c = 0..kCells
    lds_i32* startp;                    // same for the points themselves (distance rule)
    lds_u2* cell;                       // [kFineCells], `fine` only: {start | count << 16, first midpoint of the cell}
};
extern __shared__ __attribute__((aligned(16))) unsigned char qd_dyn_lds[];     // every kernel's dynamic LDS starts here
constexpr int kSmallTable = 32;
// Many points, midpoint rule (the per-step call): a FINE grid of 2048 cells over [0, 1] whose entry answers an element by
// itself when its cell holds at most one midpoint -- index = start + (first midpoint <= u) -- with ONE 8-byte LDS read, no
// loop; only cells with two or more midpoints fall back to the binary search inside the cell.  With k = 256 points spread
// like uniform samples 0.7 % of the elements need the fallback (a wave executes it for an element slot in which any of its
// 64 lanes does: about once per three float4 instead of for every element).  Why: at k = 256 the kernel is bound by VALU
// issue and LDS bank conflicts, not by latency -- 61 M VALU wave-instructions per 64 Mi-element launch (25 M at k = 4), half
// of its 35.7 M LDS-active cycles lost to conflicts of five to six random table reads per element
// (docs/history/profiles/r03_sq_counters.txt).  Points crowded into few cells (percentile-initialised, bell-shaped weights) keep taking the
// fallback: never slower than the narrowed search it replaces, only no faster.  16 KB more LDS, so only the kernels without
// LDS staging of their own use it (`fine` is cleared for the chunk kernels), on a capped grid (kFineBlocksPerCu) so that
// the table is built a few thousand times, not once per 16 KB of data.
constexpr int kFineCells = 2048;
// Grid cap of the kernels that build the fine table, in blocks per CU (6 are resident next to 26.6 KB of LDS).  Measured, K5
// k = 256 at bucket 256 / 1000, 64 Mi elements (round 2's narrowed search: 117 / 144 us): 6 -> 102 / 110 us (every resident
// block builds its table at the same moment, nothing to overlap it with), 12 -> 98 / 105, 24 -> 95 / 102, 48 -> 101 / 111,
// uncapped (a table per 16 KB of data) -> 106 / 117.  (A compile-time constant; tools/ build variants of it for A/B runs.)
#ifndef QD_FINE_BLOCKS_PER_CU
#define QD_FINE_BLOCKS_PER_CU 24
#endif
constexpr int kFineBlocksPerCu = QD_FINE_BLOCKS_PER_CU;
constexpr size_t kCoarseTableBytes = (size_t)2 * kMaxPoints * sizeof(float) + (size_t)2 * (kCells + 4) * sizeof(int);   // 10272
__host__ __device__ constexpr size_t point_table_bytes(int k, int fine = 0) {
    return k <= kSmallTable ? (size_t)2 * kSmallTable * sizeof(float)
                            : kCoarseTableBytes + (fine ? (size_t)kFineCells * 8 : 0);
}

// What the kernels pass around: a handle on the LDS table.  (Tried in round 3 and dropped: for small point sets the points
// and midpoints themselves in registers, the assignment as k - 1 compares and selects without any LDS access.  With k <= 8
// and the values forced into scalar registers, 23 more live SGPRs put the kernels 88-179 SGPR spills over the limit and the
// pre-processed forward got slower at every bucket size -- k = 4, bucket 256 / 100 / 1000: 107 / 192 / 130 us against
// 93 / 114 / 97 us with the joint LDS search of count_before4.  With k <= 4 in vector registers (+26 VGPRs in the chunk
// kernel) it tied or lost on the per-step call (bucket 100: 119 us against 105 us) and gained 5-10 % on the one-off
// nonUniformQuantization call only.  docs/history/profiles/r03_side_outputs.txt.)
struct PointTable {
    const PointStore* s;
    bool fine;
};

// cell of a scaled value: monotone non-decreasing in u (x256 is exact in fp32, then truncation and a
// clamp), which is all the narrowing below relies on; NaN lands in cell 0
__device__ __forceinline__ int cell_of(float u) {
    const float t = u * (float)kCells;
    int c = t > 0.0f ? (int)t : 0;
    return c < kCells - 1 ? c : kCells - 1;
}

__device__ __forceinline__ int fine_cell_of(float u) {           // as cell_of: exact scaling by a power of two, monotone, NaN -> 0
    const float t = u * (float)kFineCells;
    int c = t > 0.0f ? (int)t : 0;
    return c < kFineCells - 1 ? c : kFineCells - 1;
}

__device__ __forceinline__ void load_points(PointTable& C, PointStore& T, const float* pts, int k, int fine = 0) {
    const int cap = k <= kSmallTable ? kSmallTable : kMaxPoints;
    // midpoints FIRST and the cell tables at a constant distance: the searches of the per-step call (midpoint rule) then
    // address LDS with compile-time offsets again, whatever the table size; only the points sit at a k-dependent offset
    // (with both arrays behind a run-time offset the k = 256 search, which is VALU-bound, issued 5 % more instructions)
    T.mid = (lds_f32*)qd_dyn_lds;
    T.pts = T.mid + cap;
    T.start = (lds_i32*)(T.mid + 2 * kMaxPoints);            // (k > 32 only, where cap == kMaxPoints)
    T.startp = T.start + (kCells + 4);
    T.cell = (lds_u2*)(qd_dyn_lds + kCoarseTableBytes);      // (`fine` only)
    for (int j = threadIdx.x; j < k; j += blockDim.x) T.pts[j] = pts[j];
    __syncthreads();
    for (int j = threadIdx.x; j + 1 < k; j += blockDim.x) {
        float d = T.pts[j + 1] - T.pts[j];   // np.diff(k)
        d = d / 2.0f;                        //  / 2
        T.mid[j] = T.pts[j] + d;             // k[:-1] + ...
    }
    __syncthreads();
    if (k > 32 && !fine) {                                   // (the fine table below replaces both coarse ones: `fine` is midpoint-rule only)
        // start[c] by binary search on the monotone predicate cell(mid_j) < c
        for (int c = threadIdx.x; c <= kCells; c += blockDim.x) {
            int lo = 0, n = k - 1;
            while (n > 0) {
                const int half = n >> 1;
                if (cell_of(T.mid[lo + half]) < c) { lo += half + 1; n -= half + 1; } else n = half;
            }
            T.start[c] = lo;
            lo = 0; n = k;
            while (n > 0) {
                const int half = n >> 1;
                if (cell_of(T.pts[lo + half]) < c) { lo += half + 1; n -= half + 1; } else n = half;
            }
            T.startp[c] = lo;
        }
        __syncthreads();
    }
    if (k > 32 && fine) {
        // count the midpoints per fine cell (integer LDS atomics), exclusive scan over the cells, then the entries
        lds_i32* cnt = (lds_i32*)T.cell;                     // the first kFineCells dwords of the entry array, for now
        lds_i32* wsum = cnt + kFineCells;                    // wave totals of the scan (<= 16)
        for (int c = threadIdx.x; c < kFineCells; c += blockDim.x) cnt[c] = 0;
        __syncthreads();
        for (int j = threadIdx.x; j + 1 < k; j += blockDim.x)
            __hip_atomic_fetch_add(&cnt[fine_cell_of(T.mid[j])], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        // thread t owns cells [t * per, (t + 1) * per); blockDim is 128, 256 or 1024: per = 16, 8, 2
        const int per = kFineCells / (int)blockDim.x;
        int mine[16];
        int local = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) { mine[q] = q < per ? cnt[threadIdx.x * per + q] : 0; local += mine[q]; }
        int incl = local;                                    // inclusive scan over the lanes of the wave
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) {
            const int o = __shfl_up(incl, sft);
            incl += (int)(threadIdx.x & 63) >= sft ? o : 0;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();                                     // every thread has read its counters: the array may be overwritten
        int base = incl - local;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wsum[w];
        __syncthreads();                                     // (wsum sits inside the entry array)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (q < per) {
                const int n_here = mine[q];
                u32x2 e;
                e.x = (uint32_t)base | ((uint32_t)n_here << 16);
                e.y = n_here > 0 ? __float_as_uint(T.mid[base]) : 0x7FC00000u;      // NaN: no midpoint, the compare is false
                T.cell[threadIdx.x * per + q] = e;
                base += n_here;
            }
        }
        __syncthreads();
    }
    C.s = &T;
    C.fine = k > 32 && fine;
}

// #{ midpoints <= u }.  For many points the search is narrowed to the midpoints that fall in u's
// grid cell: every midpoint in an earlier cell is < u and every one in a later cell is > u (cell_of
// is monotone), so the count is exact whatever the point distribution; with roughly uniform points
// the remaining range holds 0-2 midpoints instead of k-1 (LDS reads per element: ~4 instead of ~9
// at k = 256, and fewer bank conflicts).
__device__ __forceinline__ int midpoint_index_fine(const PointStore& T, float u) {
    const u32x2 e = T.cell[fine_cell_of(u)];
    const int s0 = (int)(e.x & 0xFFFFu), n_here = (int)(e.x >> 16);
    int i = s0 + ((__uint_as_float(e.y) <= u) ? 1 : 0);                   // right for a cell with 0 or 1 midpoints
    if (n_here > 1) i = s0 + count_before<true>(T.mid + s0, n_here, u);   // crowded cell: search inside it
    return i;
}
__device__ __forceinline__ int midpoint_index(const PointStore& T, int k, float u) {
    if (k <= 32) return count_before<true>(T.mid, k - 1, u);
    const int c = cell_of(u);
    const int s0 = T.start[c];
    return s0 + count_before<true>(T.mid + s0, T.start[c + 1] - s0, u);
}

// nearest point of u (quant_functions.py:267-273 or :531-573)
__device__ __forceinline__ int assign_point(const PointStore& T, int k, int mode, float u) {
    if (mode == QD_ASSIGN_MIDPOINT) return midpoint_index(T, k, u);
    int i;                                           // searchsorted(side='left'): #{ points < u }
    if (k <= 32) {
        i = count_before<false>(T.pts, k, u);
    } else {                                         // narrowed to u's grid cell, exact for the same reason
        const int c = cell_of(u);
        const int s0 = T.startp[c];
        i = s0 + count_before<false>(T.pts + s0, T.startp[c + 1] - s0, u);
    }
    i = i > k - 1 ? k - 1 : i;                       // .clip(max=k-1)
    if (i > 0) {
        const float dl = fabsf(u - T.pts[i - 1]);
        const float dh = fabsf(u - T.pts[i]);
        i -= (dl < dh) ? 1 : 0;                      // strictly closer to the lower point
    }
    return i;
}

// ---- per-element transform shared by every bucket kernel -----------------------------------
// v: prepared value (mean subtracted, clamped) -- or u itself when prescaled.
// e: global element index (for the side outputs and the random stream).
template <int MODE, bool FAST = false>
__device__ __forceinline__ float transform(const KParams& p, const PointTable* T, float v, float a, float b,
                                           float mean, float rnd, float& side, float y = 0.0f) {
    if (MODE == MODE_QDQ) {
        return p.stochastic ? qdq_stochastic<FAST>(v, a, b, p.sm1, mean, rnd, side, y)
                            : qdq<FAST>(v, a, b, p.sm1, mean, side, y);
    } else if (MODE == MODE_SCALE) {
        float u = v - b;
        u = div_alpha<FAST>(u, a, y);                // FAST only where the caller has checked the bucket (scale_fast_ok)
        return u;
    } else {
        float u = v;
        if (!p.prescaled) { u = v - b; u = u / a; }
        const int i = T->fine ? midpoint_index_fine(*T->s, u) : assign_point(*T->s, p.k, p.assign_mode, u);
        const float pt = T->s->pts[i];
        side = (float)i;
        float y = pt * a;
        y = y + b;
        y = y + mean;
        return y;
    }
}

// nearest points of the four elements of a float4, up to 32 points (every configuration of the reference's scripts:
// 2^bits points, bits <= 4 in the differentiable-quantization runs): the four searches advance together (count_before4).
// Above 32 points transform_x4 keeps the element-by-element form of assign_point: there the kernel is bound by VALU issue
// and LDS bank conflicts, not by latency (at k = 256: 64 M VALU wave-instructions per 64 Mi-element launch ~ 105 us of
// issue time, half of the 35.7 M LDS-active cycles lost to conflicts of random reads in 256-entry tables), and both a
// joint narrowed search with always-issued reads (134 us against 115 us) and merely routing the four per-element searches
// through this function (125 us) measured slower.  docs/history/profiles/r03_sq_counters.txt.
__device__ __forceinline__ void assign_point4(const PointStore& T, int k, int mode, const float (&u)[4], int (&i)[4]) {
    if (mode == QD_ASSIGN_MIDPOINT) {
        count_before4<true>(T.mid, k - 1, u, i);
        return;
    }
    count_before4<false>(T.pts, k, u, i);                // searchsorted(side='left'): #{ points < u }
#pragma unroll
    for (int c = 0; c < 4; ++c) i[c] = i[c] > k - 1 ? k - 1 : i[c];                 // .clip(max=k-1)
    float pl[4], ph[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { pl[c] = T.pts[i[c] > 0 ? i[c] - 1 : 0]; ph[c] = T.pts[i[c]]; }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float dl = fabsf(u[c] - pl[c]);
        const float dh = fabsf(u[c] - ph[c]);
        i[c] -= (i[c] > 0 && dl < dh) ? 1 : 0;           // strictly closer to the lower point
    }
}

// transform<MODE>() of the four elements of a float4 that share (a, b): the same arithmetic per element; in the
// nearest-point mode the four point searches run together.
template <int MODE, bool FAST = false>
__device__ __forceinline__ void transform_x4(const KParams& p, const PointTable* T, const float (&v)[4], float a, float b,
                                             float mean, const float (&rnd)[4], float (&side)[4], float (&out)[4],
                                             float y = 0.0f) {
    if (MODE == MODE_NEAREST && p.k <= 32) {
        float u[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            u[c] = v[c];
            if (!p.prescaled) { u[c] = v[c] - b; u[c] = u[c] / a; }
        }
        int i[4];
        float pt[4];
        assign_point4(*T->s, p.k, p.assign_mode, u, i);
#pragma unroll
        for (int c = 0; c < 4; ++c) pt[c] = T->s->pts[i[c]];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            side[c] = (float)i[c];
            float o = pt[c] * a;
            o = o + b;
            o = o + mean;
            out[c] = o;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) out[c] = transform<MODE, FAST>(p, T, v[c], a, b, mean, rnd[c], side[c], y);
    }
}
template <int MODE, bool FAST = false>
__device__ __forceinline__ f4 transform_f4(const KParams& p, const PointTable* T, const f4& v, float a, float b, float mean,
                                           const float (&rnd)[4], float (&side)[4], float y = 0.0f) {
    const float xs[4] = {v.x, v.y, v.z, v.w};
    float o[4];
    transform_x4<MODE, FAST>(p, T, xs, a, b, mean, rnd, side, o, y);
    const f4 r = {o[0], o[1], o[2], o[3]};
    return r;
}

template <int MODE>
__device__ __forceinline__ void store_side1(const KParams& p, int64_t e, float side) {
    if (MODE == MODE_QDQ) {
        if (p.lev8) p.lev8[e] = (uint8_t)(int)side;
    } else if (MODE == MODE_NEAREST) {
        if (p.idx) {
            if (p.idx_bytes == 8) ((int64_t*)p.idx)[e] = (int64_t)side;
            else ((uint8_t*)p.idx)[e] = (uint8_t)(int)side;
        }
    }
}

template <int MODE>
__device__ __forceinline__ void store_side4(const KParams& p, int64_t e, const float (&s)[4]) {
    if (MODE == MODE_QDQ) {
        if (p.lev8) {
            const uint32_t pk = (uint32_t)(int)s[0] | ((uint32_t)(int)s[1] << 8) | ((uint32_t)(int)s[2] << 16) |
                                ((uint32_t)(int)s[3] << 24);
            *(uint32_t*)(p.lev8 + e) = pk;
        }
    } else if (MODE == MODE_NEAREST) {
        if (p.idx) {
            if (p.idx_bytes == 8) {
                l2* o = (l2*)((int64_t*)p.idx + e);
                l2 a = {(int64_t)s[0], (int64_t)s[1]}, b = {(int64_t)s[2], (int64_t)s[3]};
                o[0] = a;      // plain stores: the two 16-byte halves of a lane's 32 B merge in L2
                o[1] = b;
            } else {
                const uint32_t pk = (uint32_t)(int)s[0] | ((uint32_t)(int)s[1] << 8) |
                                    ((uint32_t)(int)s[2] << 16) | ((uint32_t)(int)s[3] << 24);
                *(uint32_t*)((uint8_t*)p.idx + e) = pk;
            }
        }
    }
}

// store_side4 for callers where the 16 lanes of a DPP row hold 16 CONSECUTIVE float4 and are all active
// (k_bucket_vec): int64 indices are exchanged inside the row first so that each of the two store
// instructions writes 16 x 16 B contiguous bytes (lane i: chunk i, then chunk 16 + i of the row's 512 B)
// instead of every lane writing two 16-B halves at a 32-B stride.  Indices fit 16 bits (k <= 1024).
template <int MODE>
__device__ __forceinline__ void store_side4_row(const KParams& p, int64_t e, const float (&s)[4]) {
    if (MODE == MODE_NEAREST && p.idx && p.idx_bytes == 8) {
        const int lane = threadIdx.x & 63, l16 = lane & 15, rowbase = lane & 48;
        const uint32_t lo = (uint32_t)(int)s[0] | ((uint32_t)(int)s[1] << 16);
        const uint32_t hi = (uint32_t)(int)s[2] | ((uint32_t)(int)s[3] << 16);
        const int src_a = rowbase + (l16 >> 1), src_b = src_a + 8;
        const uint32_t a_lo = __shfl(lo, src_a), a_hi = __shfl(hi, src_a);
        const uint32_t b_lo = __shfl(lo, src_b), b_hi = __shfl(hi, src_b);
        const bool odd = l16 & 1;
        const uint32_t ca = odd ? a_hi : a_lo, cb = odd ? b_hi : b_lo;
        const l2 va = {(int64_t)(ca & 0xFFFFu), (int64_t)(ca >> 16)}, vb = {(int64_t)(cb & 0xFFFFu), (int64_t)(cb >> 16)};
        l2* o = (l2*)((int64_t*)p.idx + (e - (int64_t)l16 * 4));          // the row's first element
        o[l16] = va;                  // plain stores: 175.8 us vs 180.0 us with the non-temporal hint (k = 4, round 3); round 4, index output
        o[16 + l16] = vb;             // kept alive by the caller: 201.0 vs 199.4 us, no difference either way (docs/history/profiles/r04_ab_idx_stores.txt)
    } else {
        store_side4<MODE>(p, e, s);
    }
}

// (Round 4 measured the whole-wave form of this exchange -- each store instruction one contiguous KiB instead of four 256-byte
// pieces at a 512-byte stride -- on the stream kernel: 189.7 us both ways.  The layout of the index stores is not what holds
// the int64 calls at 67-71 % of the HBM peak; torch's own fp32 -> int64 conversion, the same 1 : 2 read : write mix, is the
// yardstick: docs/history/profiles/r04_ab_idx_stores.txt.)
// ---- LANES lanes (16 = one DPP row, 64 = a wave) process one arbitrary bucket [lo, hi): scalar
// accesses, two passes (the second pass re-reads from L1/L2).  Used for the ragged last bucket,
// short buckets, odd bucket sizes and the multi-tensor kernel's unaligned cases.
// `l` = lane index inside the group (0..LANES-1).
template <int MODE, int LANES>
__device__ __forceinline__ void bucket_lanes(const KParams& p, const PointTable* T, int64_t bkt, int64_t lo,
                                             int64_t hi, int l, const Prep& pp) {
    float a, b;
    if (MODE == MODE_NEAREST && p.prescaled) {
        a = p.alpha[bkt]; b = p.beta[bkt];
    } else {
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int64_t i = lo + l; i < hi; i += LANES) {
            const float v = prep(p.x[i], pp);
            mn = fminf(mn, v); mx = fmaxf(mx, v);
            nan |= (v != v);
        }
        if (LANES == 16) { mn = row16_min(mn); mx = row16_max(mx); }
        else { mn = wave_min(mn); mx = wave_max(mx); }
        if (group_any<LANES>(nan)) { mn = NAN; mx = NAN; }
        alpha_beta(mn, mx, a, b);
        if (l == 0) {
            if (p.alpha) p.alpha[bkt] = a;
            if (p.beta) p.beta[bkt] = b;
        }
    }
    float last = 0.0f;
    for (int64_t i = lo + l; i < hi; i += LANES) {
        float v = p.x[i];
        if (!(MODE == MODE_NEAREST && p.prescaled)) v = prep(v, pp);
        float rnd = 0.0f;
        if (MODE == MODE_QDQ && p.stochastic) {
            float r4[4];
            philox_uniform4(p.seed, (uint64_t)i >> 2, r4);
            rnd = r4[i & 3];
        }
        float side = 0.0f;
        const float y = transform<MODE>(p, T, v, a, b, pp.mean, rnd, side);
        p.out[i] = y;
        store_side1<MODE>(p, i, side);
        last = y;
    }
    if (MODE == MODE_SCALE) {
        // padding of the ragged last bucket: copies of x[n-1], scaled (help_functions.py:76-86)
        const int64_t end = lo + p.row;
        if (hi < end && hi == p.n && p.nb > 1) {
            const float u_last = transform<MODE>(p, T, prep(p.x[p.n - 1], pp), a, b, pp.mean, 0.0f, last);
            for (int64_t i = hi + l; i < end; i += LANES) p.out[i] = u_last;
        }
    }
}

// Same, for a FULL bucket whose length is a multiple of 4 and whose base is 16-byte aligned:
// float4 accesses (4x fewer memory instructions than the scalar routine).  Odd bucket sizes such
// as 100 take this path; the second pass re-reads the bucket from L1/L2.
template <int MODE, int LANES>
__device__ __forceinline__ void bucket_lanes4(const KParams& p, const PointTable* T, int64_t bkt, int64_t lo,
                                              int l, const Prep& pp) {
    const int nv = (int)(p.row >> 2);
    const f4* src = (const f4*)(p.x + lo);
    f4* dst = (f4*)(p.out + lo);
    const bool prescaled = (MODE == MODE_NEAREST && p.prescaled);
    float a, b;
    if (prescaled) {
        a = p.alpha[bkt]; b = p.beta[bkt];
    } else {
        float mn = INFINITY, mx = -INFINITY;
        bool nan = false;
        for (int i = l; i < nv; i += LANES) {
            const f4 v = prep4(src[i], pp);
            mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            nan |= has_nan4(v);
        }
        if (LANES == 16) { mn = row16_min(mn); mx = row16_max(mx); }
        else { mn = wave_min(mn); mx = wave_max(mx); }
        if (group_any<LANES>(nan)) { mn = NAN; mx = NAN; }
        alpha_beta(mn, mx, a, b);
        if (l == 0) {
            if (p.alpha) p.alpha[bkt] = a;
            if (p.beta) p.beta[bkt] = b;
        }
    }
    for (int i = l; i < nv; i += LANES) {
        f4 v = src[i];
        if (!prescaled) v = prep4(v, pp);
        const int64_t e = lo + ((int64_t)i << 2);
        float rnd[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == MODE_QDQ && p.stochastic) {
            // element index need not be a multiple of 4 here: draw per element from its own block
            for (int c = 0; c < 4; ++c) {
                float r4[4];
                philox_uniform4(p.seed, (uint64_t)(e + c) >> 2, r4);
                rnd[c] = r4[(e + c) & 3];
            }
        }
        float side[4];
        const f4 r = transform_f4<MODE>(p, T, v, a, b, pp.mean, rnd, side);
        __builtin_nontemporal_store(r, dst + i);
        for (int c = 0; c < 4; ++c) store_side1<MODE>(p, e + c, side[c]);
    }
}

template <int MODE>
__device__ __forceinline__ void bucket_row16(const KParams& p, const PointTable* T, int64_t bkt, int64_t lo,
                                             int64_t hi, int l, const Prep& pp) {
    bucket_lanes<MODE, 16>(p, T, bkt, lo, hi, l, pp);
}

// The buckets a kernel's main path leaves over (from `first` on: the ragged last bucket, the full ones after the last whole
// tile / chunk), done by ONE block: a DPP row each for buckets up to 256 elements, a whole wave each above that -- a
// 16-lane group needs 2 x 500 dependent iterations for an 8000-element bucket (measured: 60-130 us at the end of a
// 64 Mi-element launch when the LAST block did it; the caller is block 0, which starts first, so the tail overlaps the bulk).
template <int MODE>
__device__ __forceinline__ void tail_buckets(const KParams& p, const PointTable* T, int64_t first, const Prep& pp) {
    if (p.row > 256) {
        for (int64_t bkt = first + (threadIdx.x >> 6); bkt < p.nb; bkt += (blockDim.x >> 6)) {
            const int64_t lo = bkt * p.row;
            const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
            bucket_lanes<MODE, 64>(p, T, bkt, lo, hi, threadIdx.x & 63, pp);
        }
    } else {
        for (int64_t bkt = first + (threadIdx.x >> 4); bkt < p.nb; bkt += (blockDim.x >> 4)) {
            const int64_t lo = bkt * p.row;
            const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
            bucket_row16<MODE>(p, T, bkt, lo, hi, threadIdx.x & 15, pp);
        }
    }
}

// ---- vector path: LPB lanes per bucket, V float4 per lane: bucket = LPB*V*4 elements ---------
// LPB == 16: a DPP row owns a bucket; LPB == 64: the whole wave owns a bucket (large buckets).
// U = consecutive buckets each lane group handles per tile, so that a wave always streams 4 KiB
// per tile (U*V == 4 float4 per lane in flight) whatever the bucket size.

// wave-uniform facts about the call that the per-bucket code branches on
struct VecFlags {
    bool prescaled;   // NEAREST with u, alpha, beta given
    bool prep_on;     // mean subtraction / clamp requested
    bool use_tab;     // QDQ, deterministic, <= 16 levels: level / (s-1) from the per-row table
    bool use_tab_s;   // same for the stochastic branch
    float tab;        // this lane's table entry: (lane & 15) / (s-1)
};

// The transform + store half of vec_bucket for one code variant: VAR 0 = deterministic with the per-row level table
// (<= 16 levels), 1 = stochastic with the table, 2 = generic transform<MODE>; FAST = bucket-invariant division.
template <int MODE, int LPB, int V, int VAR, bool FAST>
__device__ __forceinline__ void vec_apply(const KParams& p, const PointTable* T, const Prep& pp, const VecFlags& fl,
                                          f4 (&v)[V], int64_t e0, float a, float b) {
    // The seed goes through an opaque (empty) asm per instantiation: without it LLVM hoists the code the variants share --
    // the whole Philox draw of the stochastic and generic variants -- in front of the variant dispatch, where the
    // deterministic path executes it too (that is what round 1's kernel did: 12 of its 32 VALU instructions per element
    // were an unused random draw).
    uint32_t seed_lo = (uint32_t)p.seed, seed_hi = (uint32_t)(p.seed >> 32);
    if (MODE == MODE_QDQ && VAR != 0) asm volatile("; seed of variant %2" : "+s"(seed_lo), "+s"(seed_hi) : "n"(VAR * 2 + (FAST ? 1 : 0)));
    const uint64_t seed = (uint64_t)seed_lo | ((uint64_t)seed_hi << 32);
    uint32_t ctr_lo = (uint32_t)e0, ctr_hi = (uint32_t)((uint64_t)e0 >> 32);         // same for the counter (round 1 is seed-free)
    if (MODE == MODE_QDQ && VAR != 0) asm volatile("; counter of variant %2" : "+v"(ctr_lo), "+v"(ctr_hi) : "n"(VAR * 2 + (FAST ? 1 : 0)));
    const uint64_t ctr0 = (uint64_t)ctr_lo | ((uint64_t)ctr_hi << 32);
    f4* dst = (f4*)(p.out + e0);
    const float y = FAST ? 1.0f / a : 0.0f;             // RN(1/alpha): one IEEE division per bucket and lane
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int64_t e = e0 + (int64_t)j * LPB * 4;
        float rnd[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == MODE_QDQ && VAR != 0 && p.stochastic) philox_uniform4(seed, (ctr0 + (uint64_t)j * LPB * 4) >> 2, rnd);
        float side[4];
        f4 r;
        if (MODE == MODE_QDQ && VAR == 0) {
            r.x = qdq_tab<FAST>(v[j].x, a, b, p.sm1, pp.mean, side[0], fl.tab, y);
            r.y = qdq_tab<FAST>(v[j].y, a, b, p.sm1, pp.mean, side[1], fl.tab, y);
            r.z = qdq_tab<FAST>(v[j].z, a, b, p.sm1, pp.mean, side[2], fl.tab, y);
            r.w = qdq_tab<FAST>(v[j].w, a, b, p.sm1, pp.mean, side[3], fl.tab, y);
        } else if (MODE == MODE_QDQ && VAR == 1) {
            r.x = qdq_stochastic_tab<FAST>(v[j].x, a, b, p.sm1, pp.mean, rnd[0], side[0], fl.tab, y);
            r.y = qdq_stochastic_tab<FAST>(v[j].y, a, b, p.sm1, pp.mean, rnd[1], side[1], fl.tab, y);
            r.z = qdq_stochastic_tab<FAST>(v[j].z, a, b, p.sm1, pp.mean, rnd[2], side[2], fl.tab, y);
            r.w = qdq_stochastic_tab<FAST>(v[j].w, a, b, p.sm1, pp.mean, rnd[3], side[3], fl.tab, y);
        } else {
            r = transform_f4<MODE, FAST>(p, T, v[j], a, b, pp.mean, rnd, side, y);
        }
        // MODE_SCALE: the whole-tile and partial-tile copies of this loop are merged by the compiler, and the merged store
        // loses its !nontemporal flag (qd_common.h store_nt_pinned)
        if (MODE == MODE_SCALE) store_nt_pinned((QD_AS_GLOBAL f4*)dst + j * LPB, r);
        else __builtin_nontemporal_store(r, dst + j * LPB);
        store_side4_row<MODE>(p, e, side);
    }
}

// One bucket held in registers by its LPB lanes (v[0..V): this lane's float4s, already loaded): reduce, transform,
// store.  The ONLY copy of the per-bucket arithmetic of the vector kernels: both loop shapes of k_bucket_vec call it.
template <int MODE, int LPB, int V>
__device__ __forceinline__ void vec_bucket(const KParams& p, const PointTable* T, const Prep& pp, const VecFlags& fl,
                                           f4 (&v)[V], int64_t bkt, int uu, int l, float& a_keep, float& b_keep) {
    constexpr int ROW = LPB * V * 4;
    const int64_t e0 = bkt * ROW + (int64_t)l * 4;     // first element of this lane
    float a, b;
    if (fl.prescaled) {
        a = p.alpha[bkt]; b = p.beta[bkt];
    } else {
        if (MODE != MODE_QDQ || fl.prep_on) {
#pragma unroll
            for (int j = 0; j < V; ++j) v[j] = prep4(v[j], pp);
        }
        float mn = pmin4(v[0]), mx = pmax4(v[0]);           // NaN-propagating: a NaN element makes both NaN
#pragma unroll
        for (int j = 1; j < V; ++j) { mn = pmin(mn, pmin4(v[j])); mx = pmax(mx, pmax4(v[j])); }
        if (LPB == 16) { mn = row16_min(mn); mx = row16_max(mx); }
        else { mn = wave_min(mn); mx = wave_max(mx); }
        alpha_beta(mn, mx, a, b);
        if (l == uu) { a_keep = a; b_keep = b; }       // lane uu of the group keeps bucket uu's pair
    }
    // quantize-dequantize consumes only the LEVEL of u, so the bucket-invariant division form is exact there (qd_common.h);
    // wave-uniform choice: every bucket of the wave must be in the proven range.  The transform loop is instantiated per
    // (variant, division form) and chosen ONCE per bucket: with the flags tested inside the loop the compiler no longer
    // unswitched it and the kernel executed 37.9 M instead of 33.9 M VALU wave-instructions (docs/history/profiles/r02_sq_counters.txt).
    const bool fast = MODE == MODE_QDQ && !__any(!fastdiv_ok(a));
    if (MODE == MODE_SCALE) {
        unsigned key = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            key = min(key, min(scale_numerator_key(v[j].x, b), scale_numerator_key(v[j].y, b)));
            key = min(key, min(scale_numerator_key(v[j].z, b), scale_numerator_key(v[j].w, b)));
        }
        if (!__any(!scale_fast_ok(a, key))) vec_apply<MODE, LPB, V, 2, true>(p, T, pp, fl, v, e0, a, b);
        else vec_apply<MODE, LPB, V, 2, false>(p, T, pp, fl, v, e0, a, b);
    } else if (MODE != MODE_QDQ) vec_apply<MODE, LPB, V, 2, false>(p, T, pp, fl, v, e0, a, b);
    else if (fast) {
        if (fl.use_tab) vec_apply<MODE, LPB, V, 0, true>(p, T, pp, fl, v, e0, a, b);
        else if (fl.use_tab_s) vec_apply<MODE, LPB, V, 1, true>(p, T, pp, fl, v, e0, a, b);
        else vec_apply<MODE, LPB, V, 2, true>(p, T, pp, fl, v, e0, a, b);
    } else {
        if (fl.use_tab) vec_apply<MODE, LPB, V, 0, false>(p, T, pp, fl, v, e0, a, b);
        else if (fl.use_tab_s) vec_apply<MODE, LPB, V, 1, false>(p, T, pp, fl, v, e0, a, b);
        else vec_apply<MODE, LPB, V, 2, false>(p, T, pp, fl, v, e0, a, b);
    }
}

template <int MODE, int LPB, int V, int U>
__global__ __launch_bounds__(256) void k_bucket_vec(KParams p) {
    PointStore Ts;
    PointTable Tc;
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);

    constexpr int BPW = (64 / LPB) * U;           // buckets per wave tile
    constexpr int ROW = LPB * V * 4;              // elements per bucket
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPB;                   // which lane group of the wave
    const int l = lane % LPB;                     // lane inside the bucket
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    // wave-uniform shortcuts of the common configuration (no mean, no clamp, <= 16 levels, deterministic): they
    // take the kernel from ~43 to ~30 VALU instructions per element, which keeps it HBM-bound on boxes whose
    // sustained clock is lower (measured: 88 us vs 85.8 us for the leaner kbench kernel on the same box)
    // (MODE_QDQ only, so that the code generated for the other modes is untouched: the point-search kernels are
    // sensitive to it -- K5 at k = 256 went from 112 to 124 us when these branches were compiled into them.)
    VecFlags fl;
    fl.prescaled = (MODE == MODE_NEAREST && p.prescaled);
    fl.prep_on = MODE != MODE_QDQ || p.mean != nullptr || p.me != INFINITY;
    fl.use_tab = MODE == MODE_QDQ && !p.stochastic && p.sm1 <= 15.0f;
    fl.use_tab_s = MODE == MODE_QDQ && p.stochastic && p.sm1 <= 15.0f;
    fl.tab = MODE == MODE_QDQ ? (float)(lane & 15) / p.sm1 : 0.0f;

    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t ntiles = (p.nvec + BPW - 1) / BPW;

    // One loop shape around the one vec_bucket(): whole tiles unpredicated (all loads, then the buckets), the last,
    // partial tile bucket by bucket.
    for (int64_t t = wave; t < ntiles; t += nwaves) {
        const int64_t bkt0 = t * BPW + (int64_t)sub * U;          // first of this group's U buckets
        float a_keep = 0.0f, b_keep = 0.0f;
        if (bkt0 + U <= p.nvec) {
            // whole group in range (every tile but possibly the last): unpredicated.  All loads first, then the buckets
            // one after the other.
            f4 v[U][V];
#pragma unroll
            for (int uu = 0; uu < U; ++uu) {
                const f4* src = (const f4*)(p.x + (bkt0 + uu) * ROW + (int64_t)l * 4);
#pragma unroll
                for (int j = 0; j < V; ++j) v[uu][j] = __builtin_nontemporal_load(src + j * LPB);
            }
#pragma unroll
            for (int uu = 0; uu < U; ++uu) vec_bucket<MODE, LPB, V>(p, T, pp, fl, v[uu], bkt0 + uu, uu, l, a_keep, b_keep);
        } else {
            for (int uu = 0; uu < U; ++uu) {
                if (bkt0 + uu < p.nvec) {
                    f4 v[V];
                    const f4* src = (const f4*)(p.x + (bkt0 + uu) * ROW + (int64_t)l * 4);
#pragma unroll
                    for (int j = 0; j < V; ++j) v[j] = __builtin_nontemporal_load(src + j * LPB);
                    vec_bucket<MODE, LPB, V>(p, T, pp, fl, v, bkt0 + uu, uu, l, a_keep, b_keep);
                }
            }
        }
        // alpha/beta of the group's U buckets: lanes 0..U-1 write U consecutive floats (one store
        // instruction per array per tile instead of U single-lane stores)
        if (!fl.prescaled && l < U && bkt0 + l < p.nvec) {
            if (p.alpha) p.alpha[bkt0 + l] = a_keep;
            if (p.beta) p.beta[bkt0 + l] = b_keep;
        }
    }

    // buckets after the vector part (the ragged last bucket): one DPP row each, last block
    if (blockIdx.x == 0) {                                    // (the first block, which starts first: the tail overlaps the bulk)
        tail_buckets<MODE>(p, T, p.nvec, pp);
    }
}

// ---- chunk path: bucket sizes that are a multiple of 4 but not one of the vector sizes ----------
// (100, 1000, 36, 4096, ...).  A wave streams a CHUNK of m consecutive buckets (m a power of two, m * row / 4
// <= VMAX * 64 float4) with fully coalesced 16-byte loads -- lane i holds float4 i, i + 64, ... of the chunk,
// whatever the bucket boundaries are -- and keeps it in registers: one pass over HBM.  The per-float4 (min,
// max) go through LDS, where lane groups of 64 / m lanes reduce one bucket each; (alpha, beta) of the chunk's
// buckets come back through an LDS table.  Every chunk starts at a multiple of row * 4 bytes, so alignment
// only needs row % 4 == 0.  The buckets after the last whole chunk are done by the last block, a DPP row each.
template <int MODE, int VMAX>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 4)))   // the chunk lives in registers
void k_bucket_chunk(KParams p, int m, int64_t nchunks) {
    PointStore Ts;
    PointTable Tc;
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    // dynamic LDS: [point table (nearest-point mode)] then per wave: pairs[VMAX * 64], ab[256], 1/alpha[256]
    float2* chunk_lds = (float2*)(qd_dyn_lds + (MODE == MODE_NEAREST ? point_table_bytes(p.k) : 0));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float2* pr = chunk_lds + w * (VMAX * 64 + 256 + 128);
    float2* ab = pr + VMAX * 64;
    float* yt = (float*)(ab + 256);

    const int Bq = (int)(p.row >> 2);                  // float4 per bucket
    const int nf = m * Bq;                             // float4 per chunk
    const int nj = (nf + 63) >> 6;                     // rounds of 64 float4 (<= VMAX)
    const int mlanes = m < 64 ? m : 64;                // buckets reduced side by side
    const int G = 64 / mlanes;                         // lanes per bucket in the reduce step (power of two)
    const int bl = lane / G, sub = lane % G;
    const int step_q = 64 / Bq, step_r = 64 % Bq;      // bucket of float4 (lane + 64 j): advanced incrementally
    const int q0 = lane / Bq, r0 = lane % Bq;
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    const bool prescaled = (MODE == MODE_NEAREST && p.prescaled);
    const bool prep_on = p.mean != nullptr || p.me != INFINITY;
    const bool use_tab = MODE == MODE_QDQ && !p.stochastic && p.sm1 <= 15.0f;
    const float tab = (float)(lane & 15) / p.sm1;
    const int64_t wave = uniform_wave_index();
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

    for (int64_t c = wave; c < nchunks; c += nwaves) {
        const int64_t b0 = c * m;                      // first bucket of the chunk
        const int64_t e0 = b0 * p.row;                 // first element
        const f4* src = (const f4*)(p.x + e0);
        f4 v[VMAX];
        // always VMAX loads, the address clamped to the chunk's last float4 (a broadcast re-load): a load behind
        // a branch -- even a wave-uniform one -- gets an s_waitcnt vmcnt(0) at the join and the chunk would be
        // fetched one memory round trip per float4.  The launcher picks VMAX in {8, 16, 32} to bound the waste.
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int f = lane + 64 * j;
            v[j] = __builtin_nontemporal_load(src + (f < nf ? f : nf - 1));
        }
        __builtin_amdgcn_sched_barrier(0);             // all loads in flight before the first use
        bool div_ok = true;
        if (!prescaled) {
#pragma unroll
            for (int j = 0; j < VMAX; ++j) {
                const int f = lane + 64 * j;
                if (j < nj && f < nf) {
                    if (prep_on) v[j] = prep4(v[j], pp);
                    pr[f] = make_float2(pmin4(v[j]), pmax4(v[j]));     // NaN-propagating
                }
            }
            // LDS operations of one wave complete in order; the fence/barrier only stop compiler reordering
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int bb = bl; bb < m; bb += mlanes) {  // G > 1: m is a multiple of mlanes, so the trip count is uniform
                const float2* q = pr + bb * Bq;
                float mn = INFINITY, mx = -INFINITY;
                for (int t = sub; t < Bq; t += G) {
                    const float2 pm = q[t];
                    mn = pmin(mn, pm.x); mx = pmax(mx, pm.y);
                }
                for (int sft = 1; sft < G; sft <<= 1) {
                    mn = pmin(mn, __shfl_xor(mn, sft));
                    mx = pmax(mx, __shfl_xor(mx, sft));
                }
                float a, b;
                alpha_beta(mn, mx, a, b);
                div_ok &= fastdiv_ok(a);
                if (sub == 0) {
                    ab[bb] = make_float2(a, b);
                    yt[bb] = 1.0f / a;                  // RN(1/alpha), one IEEE division per bucket
                    if (p.alpha) p.alpha[b0 + bb] = a;
                    if (p.beta) p.beta[b0 + bb] = b;
                }
            }
        } else {
            for (int bb = lane; bb < m; bb += 64) ab[bb] = make_float2(p.alpha[b0 + bb], p.beta[b0 + bb]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f4* dst = (f4*)(p.out + e0);
        // only the level of u is consumed by quantize-dequantize: bucket-invariant division (qd_common.h) when every
        // bucket of the chunk is in its proven range
        const bool fast = MODE == MODE_QDQ && !prescaled && !__any(!div_ok);
        auto body = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            int q = q0, r = r0;
#pragma unroll
            for (int j = 0; j < VMAX; ++j) {
                const int f = lane + 64 * j;
                if (j < nj) {                                  // whole wave: lanes past the chunk work on their duplicate
                    const int qi = q < m ? q : m - 1;
                    const float2 s = ab[qi];
                    const float y = FAST ? yt[qi] : 0.0f;
                    const int64_t e = e0 + ((int64_t)f << 2);
                    float side[4];
                    f4 o;
                    if (use_tab) {                             // <= 16 levels, deterministic: see k_bucket_vec
                        o.x = qdq_tab<FAST>(v[j].x, s.x, s.y, p.sm1, pp.mean, side[0], tab, y);
                        o.y = qdq_tab<FAST>(v[j].y, s.x, s.y, p.sm1, pp.mean, side[1], tab, y);
                        o.z = qdq_tab<FAST>(v[j].z, s.x, s.y, p.sm1, pp.mean, side[2], tab, y);
                        o.w = qdq_tab<FAST>(v[j].w, s.x, s.y, p.sm1, pp.mean, side[3], tab, y);
                    } else {
                        float rnd[4] = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == MODE_QDQ && p.stochastic) philox_uniform4(p.seed, (uint64_t)e >> 2, rnd);
                        o = transform_f4<MODE, FAST>(p, T, v[j], s.x, s.y, pp.mean, rnd, side, y);
                    }
                    // (int64 indices: exchanged inside the DPP row -- 16 consecutive float4s, all in range -- so that each store
                    // instruction writes 256 contiguous bytes; the row that straddles the end of the chunk stores per lane)
                    const bool row_in = MODE == MODE_NEAREST && !group_any<16>(f >= nf);
                    if (MODE == MODE_NEAREST && row_in) {
                        __builtin_nontemporal_store(o, dst + f);
                        store_side4_row<MODE>(p, e, side);
                    } else if (f < nf) {
                        __builtin_nontemporal_store(o, dst + f);
                        store_side4<MODE>(p, e, side);
                    }
                }
                q += step_q; r += step_r;
                if (r >= Bq) { r -= Bq; ++q; }
            }
        };
        if (fast) body(std::true_type{}); else body(std::false_type{});
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next chunk overwrites pr / ab
        __builtin_amdgcn_wave_barrier();
    }

    if (blockIdx.x == 0) {                             // buckets after the last whole chunk (incl. the ragged one); block 0 starts first
        if (p.row > 256) {
            tail_buckets<MODE>(p, T, nchunks * m, pp);
        } else {
            const int row_id = threadIdx.x >> 4;
            for (int64_t bkt = nchunks * m + row_id; bkt < p.nb; bkt += (blockDim.x >> 4)) {
                const int64_t lo = bkt * p.row;
                const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
                if (hi - lo == p.row) bucket_lanes4<MODE, 16>(p, T, bkt, lo, threadIdx.x & 15, pp);   // full: float4 accesses
                else bucket_row16<MODE>(p, T, bkt, lo, hi, threadIdx.x & 15, pp);
            }
        }
    }
}

// Same for bucket sizes that are NOT a multiple of 4 (33, 50, 7, 3, 1, ...): m is a multiple of 4, so every chunk
// still starts 16-byte aligned and holds a whole number of float4, but a float4 may straddle two buckets.  So the
// registers only carry the chunk between HBM and LDS: the prepared values are staged in LDS (16 B per float4), the lane
// (group) that reduces a bucket goes straight on to TRANSFORM it in place in LDS -- alpha, beta and 1/alpha are then
// per-lane constants, no per-element choice between two buckets, and no register array stays live across the phases --
// and finally every lane streams its float4s from LDS to HBM, coalesced.  (The first version transformed the register
// copy and picked (alpha, beta) per element from the two candidate buckets: 48 VALU instructions per element against
// 32 in the vector kernel, VALU-bound at 107-138 us for 64 Mi elements.)
template <int MODE, int VMAX>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(1, 6)))
void k_bucket_chunk_any(KParams p, int m, int64_t nchunks, int lead) {
    PointStore Ts;
    PointTable Tc;
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    // dynamic LDS: [point table (nearest-point mode)] then per wave: vals[VMAX * 256] floats (+ side bytes)
    float2* chunk_lds = (float2*)(qd_dyn_lds + (MODE == MODE_NEAREST ? point_table_bytes(p.k) : 0));
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // Side outputs (level / point indices up to 255) are staged as ONE BYTE per element behind the values -- a quarter more
    // LDS, only when such an output is asked for -- and leave with the values: four per lane, coalesced.  Stored from the
    // transform loop they would leave element by element in LDS order (lanes a bucket apart): 216 us instead of 95 us for
    // quantize + levels at bucket 33, 405 us for int64 point indices (docs/history/profiles/r03_side_outputs.txt).  (Up to 32 points: with
    // the 10.3 KB table of a larger point set the extra bytes cost a resident block, which measured slower than the scattered
    // stores: k = 256 at bucket 33: 316 us against 281 us.)
    const bool stage8 = (MODE == MODE_QDQ && p.lev8 != nullptr) || (MODE == MODE_NEAREST && p.idx != nullptr && p.k <= 32);
    const int wave_floats = VMAX * 256 + (stage8 ? VMAX * 64 : 0);
    float* vals = (float*)chunk_lds + w * wave_floats;
    uint8_t* sidev = (uint8_t*)(vals + VMAX * 256);

    const int B = (int)p.row;
    const int nf = (m * B) >> 2;                       // float4 per chunk (m % 4 == 0)
    const int mlanes = m < 64 ? m : 64;
    const int G = 64 / mlanes;                         // lanes per bucket (a power of two whenever it is > 1)
    const int bl = lane / G, sub = lane % G;
    const int steps = (B + G - 1) / G;                 // elements each lane of a group handles
    const int nrounds = (m + mlanes - 1) / mlanes;     // buckets each lane group handles
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    const bool prescaled = (MODE == MODE_NEAREST && p.prescaled);
    const bool prep_on = p.mean != nullptr || p.me != INFINITY;
    const bool use_tab = MODE == MODE_QDQ && !p.stochastic && p.sm1 <= 15.0f;
    const float tab = (float)(lane & 15) / p.sm1;
    const int64_t wave = uniform_wave_index();
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;

    for (int64_t c = wave; c < nchunks; c += nwaves) {
        const int64_t b0 = c * m;
        const int64_t e0 = b0 * p.row;
        const f4* src = (const f4*)(p.x + e0);
        // Lane i of round j holds float4 (i + 64 j - h) of the chunk, h = the chunk's distance from the 128-byte line below
        // it: every load and store instruction then covers eight whole lines instead of straddling nine (chunks start at
        // arbitrary multiples of 16 bytes; measured on the one-wave-per-bucket kernel: 107.7 -> 100.1 us at bucket 1000).
        // The launcher leaves room for the lead-in (nf + 7 <= VMAX * 64) or switches it off (lead = 0: sizes 506 .. 511).
        const int h = lead ? (int)((e0 >> 2) & 7) : 0;
        {
            f4 v[VMAX];
#pragma unroll
            for (int j = 0; j < VMAX; ++j) {           // always-issued loads with a clamped address, as above
                const int f = lane + 64 * j - h;
                v[j] = __builtin_nontemporal_load(src + (f < 0 ? 0 : (f < nf ? f : nf - 1)));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < VMAX; ++j) {
                const int f = lane + 64 * j - h;
                if (f >= 0 && f < nf) {
                    if (!prescaled && prep_on) v[j] = prep4(v[j], pp);
                    ((f4*)vals)[f] = v[j];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        // every lane runs every round and every step (clamped indices, masked stores): the level table of qdq_tab is
        // fetched with ds_bpermute from the lanes of the own DPP row, which must all be active
        for (int rd = 0; rd < nrounds; ++rd) {
            const int bb_raw = bl + rd * mlanes;
            const bool live = bb_raw < m;
            const int bb = live ? bb_raw : m - 1;
            float* q = vals + bb * B;
            float a, b;
            if (prescaled) {
                a = p.alpha[b0 + bb]; b = p.beta[b0 + bb];
            } else {
                float mn = INFINITY, mx = -INFINITY;
                int t = sub;
                for (; t + G < B; t += 2 * G) {         // two elements per step: one min3 / max3 each (NaN-propagating)
                    const float x0 = q[t], x1 = q[t + G];
                    mn = pmin(mn, pmin(x0, x1)); mx = pmax(mx, pmax(x0, x1));
                }
                if (t < B) { const float x0 = q[t]; mn = pmin(mn, x0); mx = pmax(mx, x0); }
                for (int sft = 1; sft < G; sft <<= 1) {
                    mn = pmin(mn, __shfl_xor(mn, sft));
                    mx = pmax(mx, __shfl_xor(mx, sft));
                }
                alpha_beta(mn, mx, a, b);
                if (live && sub == 0) {
                    if (p.alpha) p.alpha[b0 + bb] = a;
                    if (p.beta) p.beta[b0 + bb] = b;
                }
            }
            // only the level of u is consumed by quantize-dequantize: bucket-invariant division (qd_common.h) when every
            // bucket of this round is in its proven range
            const bool fast = MODE == MODE_QDQ && !__any(!fastdiv_ok(a));
            const int64_t eb = e0 + (int64_t)bb * B;   // first element of the bucket
            auto body = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value;
                const float y = FAST ? 1.0f / a : 0.0f; // RN(1/alpha), one IEEE division per bucket
                // four steps at a time: the four LDS reads, and in the nearest-point mode the four point searches, are
                // independent of each other (one element per step left every lane with `steps` dependent LDS chains in a
                // row: 234 us for the pre-processed forward at bucket 33 against 95 us for the plain quantize)
                for (int i = 0; i < steps; i += 4) {
                    int tt[4];
                    bool ok[4];
                    float xv[4], o[4], side[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int t_raw = sub + (i + c) * G;
                        ok[c] = live && t_raw < B && i + c < steps;
                        tt[c] = t_raw < B ? t_raw : B - 1;
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) xv[c] = q[tt[c]];
                    if (use_tab) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] = qdq_tab<FAST>(xv[c], a, b, p.sm1, pp.mean, side[c], tab, y);
                    } else {
                        float rnd[4] = {0.f, 0.f, 0.f, 0.f};
                        if (MODE == MODE_QDQ && p.stochastic) {
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const int64_t e = eb + tt[c];
                                float r4[4];
                                philox_uniform4(p.seed, (uint64_t)e >> 2, r4);
                                rnd[c] = r4[e & 3];
                            }
                        }
                        transform_x4<MODE, FAST>(p, T, xv, a, b, pp.mean, rnd, side, o, y);
                    }
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (ok[c]) {
                            q[tt[c]] = o[c];
                            if (stage8) sidev[bb * B + tt[c]] = (uint8_t)(int)side[c];
                            else store_side1<MODE>(p, eb + tt[c], side[c]);
                        }
                    }
                }
            };
            if (fast) body(std::true_type{}); else body(std::false_type{});
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f4* dst = (f4*)(p.out + e0);
#pragma unroll
        for (int j = 0; j < VMAX; ++j) {
            const int f = lane + 64 * j - h;
            const bool in = f >= 0 && f < nf;
            if (in) __builtin_nontemporal_store(((const f4*)vals)[f], dst + f);
            if (MODE != MODE_SCALE && stage8) {
                const uint32_t pk = in ? ((const uint32_t*)sidev)[f] : 0u;
                const int64_t e = e0 + ((int64_t)f << 2);
                if (MODE == MODE_QDQ) {
                    if (in) *(uint32_t*)(p.lev8 + e) = pk;
                } else if (p.idx_bytes == 1) {
                    if (in) *(uint32_t*)((uint8_t*)p.idx + e) = pk;
                } else {
                    // int64: through the DPP-row exchange where the row's 16 float4s are all inside the chunk
                    const float sd[4] = {(float)(pk & 255u), (float)((pk >> 8) & 255u), (float)((pk >> 16) & 255u), (float)(pk >> 24)};
                    if (!group_any<16>(!in)) store_side4_row<MODE>(p, e, sd);
                    else if (in) store_side4<MODE>(p, e, sd);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next chunk overwrites vals
        __builtin_amdgcn_wave_barrier();
    }

    if (blockIdx.x == 0) tail_buckets<MODE>(p, T, nchunks * m, pp);
}

// ---- one wave per bucket, ANY bucket size above 256 (every mode): 513, 1000, 1001, 2000, 3000, ... ---------------
// The wave loads the 16-byte-aligned float4s that TOUCH its bucket [lo, hi) -- lane i holds float4 i, i + 64, ... counted
// from the aligned element at or below lo, so a bucket that does not start on a 16-byte boundary shares its first and last
// float4 with its neighbours (those two are fetched twice, the second time from L2) -- keeps them in registers, reduces
// min / max over the elements that belong to the bucket (rounds that lie wholly inside take the unmasked path: a
// wave-uniform test), and writes whole float4s with one 16-byte store, the up to 3 + 3 elements of the shared edge float4s
// one by one.  One pass over HBM at any size, the short last bucket included (the float4 that would reach past the end of
// the tensor is fetched as x[n-4 .. n-1] and rotated in registers: no byte outside the tensor is read, and nothing is
// ever stored outside the bucket) -- handing the last one or two buckets of a tensor to a 16-lane
// group, as the kernels above do for their 256-element buckets, costs 60-130 us at bucket sizes of 3000-8000.  The chunk
// kernels above stay for small buckets, where a wave per bucket would leave most lanes idle.  The float4 grid is aligned in ELEMENT index (the base pointer is 16-byte aligned), so the
// stochastic draw of element e -- Philox block e >> 2, word e & 3 -- is the one every other kernel uses.
// G = waves per bucket (1, 2 or 4 of the block's four): above 2048 elements a single wave would need more than 8 rounds --
// 115-252 VGPRs, two to four waves per SIMD -- so the bucket is spread over 2 (up to 4096 elements) or 4 waves (up to
// 8192), which exchange their (min, max) through LDS across one block barrier; every wave then keeps at most 9 float4s.
template <int MODE, int V, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))   // <= 256 VGPRs: the 32-float4 instance must keep two waves per SIMD
void k_bucket_wave_any(KParams p, int64_t nbk, int64_t amask) {
    PointStore Ts;
    PointTable Tc;
    __shared__ float red[2][4][2];                         // [iteration parity][wave of the block][min, max]
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    constexpr int GL = 64 * G;                             // lanes per bucket
    constexpr int GPB = 4 / G;                             // buckets per block and iteration
    const int lane = threadIdx.x & (GL - 1);               // lane inside the bucket's group
    const int wv = threadIdx.x >> 6;                       // wave of the block
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    const bool prescaled = (MODE == MODE_NEAREST && p.prescaled);
    const bool prep_on = !prescaled && (p.mean != nullptr || p.me != INFINITY);
    const bool use_tab = MODE == MODE_QDQ && !p.stochastic && p.sm1 <= 15.0f;
    const float tab = (float)(lane & 15) / p.sm1;
    const int64_t group = (int64_t)blockIdx.x * GPB + (threadIdx.x / GL);
    const int64_t ngroups = (int64_t)gridDim.x * GPB;
    const int64_t iters = (nbk + ngroups - 1) / ngroups;   // the same for every wave of the grid: the barriers below are uniform

    for (int64_t it = 0; it < iters; ++it) {
        const int64_t bkt = it * ngroups + group;
        const bool active = bkt < nbk;                     // uniform over the group's waves
        const int64_t lo = (active ? bkt : 0) * p.row;
        const int row = (int)(lo + p.row <= p.n ? p.row : p.n - lo);   // the last bucket may be short
        const int64_t a0 = lo & amask;                     // aligned element at or below lo (amask = ~31: a 128-byte line)
        const int off = (int)(a0 - lo);                    // -31 .. 0: position of a0 relative to the bucket
        const int nf = (int)(((lo + row + 3) >> 2) - (a0 >> 2));   // float4s from a0 to the bucket's last one
        const f4* src = (const f4*)(p.x + a0);
        // The bucket's last float4 may reach past the END OF THE TENSOR (n % 4 != 0; the last bucket, or the one before it
        // when the last one has fewer than 3 elements).  The lane that holds it fetches the 16 bytes x[n-4 .. n-1] instead
        // -- the same instruction with another address: 16-byte accesses need 4-byte alignment only -- and rotates the
        // t = n % 4 valid elements to the front; nothing outside the tensor is ever read.
        // (Instances with more than 16 float4 per lane -- buckets above 12288 elements -- sit at the 256-VGPR limit and have
        // no room for the rotation: the launcher keeps such buckets out of their range, block 0 does them with scalar
        // accesses below.)
        const bool tail_partial = V <= 16 && active && (((lo + row + 3) >> 2) << 2) > p.n;      // group-uniform
        f4 v[V];
        float a = 1.0f, b = 0.0f;
        float mn = INFINITY, mx = -INFINITY;
        if (active) {
#pragma unroll
            for (int j = 0; j < V; ++j) {                  // always issued, the index clamped (see k_bucket_chunk)
                const int f = lane + GL * j;
                const int fc = f < nf ? f : nf - 1;
                const float* ptr = (const float*)(src + fc);
                if (V <= 16 && tail_partial && fc == nf - 1) ptr = p.x + (p.n - 4);
                v[j] = __builtin_nontemporal_load((const f4*)ptr);
            }
            if (V <= 16 && tail_partial) {
                const int t = (int)(p.n & 3);
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int f = lane + GL * j;
                    if ((f < nf ? f : nf - 1) == nf - 1) {
                        const f4 w = v[j];
                        v[j].x = t == 1 ? w.w : (t == 2 ? w.z : w.y);
                        v[j].y = t == 2 ? w.w : w.z;
                        v[j].z = w.w;
                    }
                }
            }
            if (prescaled) {                               // x is u, alpha / beta are inputs
                a = p.alpha[bkt]; b = p.beta[bkt];
            } else {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (prep_on) v[j] = prep4(v[j], pp);
                    const int r0 = off + 4 * (lane + GL * j);      // position of this float4's first element in the bucket
                    if (off + 4 * GL * j >= 0 && off + 4 * GL * (j + 1) <= row) {   // group-uniform: the whole round is inside
                        mn = pmin(mn, pmin4(v[j])); mx = pmax(mx, pmax4(v[j]));
                    } else {
                        const float xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const bool in = (unsigned)(r0 + c) < (unsigned)row;
                            mn = pmin(mn, in ? xs[c] : INFINITY); mx = pmax(mx, in ? xs[c] : -INFINITY);
                        }
                    }
                }
                mn = wave_min(mn); mx = wave_max(mx);
            }
        }
        if (G > 1) {                                       // the bucket's waves exchange (min, max); parity: one barrier per iteration
            const int par = (int)(it & 1);
            if ((threadIdx.x & 63) == 0) { red[par][wv][0] = mn; red[par][wv][1] = mx; }
            __syncthreads();
            const int w0 = (wv / G) * G;
#pragma unroll
            for (int g = 0; g < G; ++g) { mn = pmin(mn, red[par][w0 + g][0]); mx = pmax(mx, red[par][w0 + g][1]); }
        }
        if (!active) continue;
        if (!prescaled) {
            alpha_beta(mn, mx, a, b);
            if (lane == 0) {
                if (p.alpha) p.alpha[bkt] = a;
                if (p.beta) p.beta[bkt] = b;
            }
        }
        const bool fast = MODE == MODE_QDQ && fastdiv_ok(a);   // a is uniform over the bucket's waves
        auto body = [&](auto fast_c) {
            constexpr bool FAST = decltype(fast_c)::value;
            const float y = FAST ? 1.0f / a : 0.0f;        // RN(1/alpha), one IEEE division per bucket
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const int f = lane + GL * j;
                const int r0 = off + 4 * f;
                const int64_t e = a0 + 4 * (int64_t)f;     // element index of this float4 (a multiple of 4)
                float side[4], o[4];
                const float xs[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                if (use_tab) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = qdq_tab<FAST>(xs[c], a, b, p.sm1, pp.mean, side[c], tab, y);
                } else {
                    float rnd[4] = {0.f, 0.f, 0.f, 0.f};
                    if (MODE == MODE_QDQ && p.stochastic) philox_uniform4(p.seed, (uint64_t)e >> 2, rnd);
#pragma unroll
                    for (int c = 0; c < 4; ++c) side[c] = 0.0f;
                    transform_x4<MODE, FAST>(p, T, xs, a, b, pp.mean, rnd, side, o, y);
                }
                if (off + 4 * GL * j >= 0 && off + 4 * GL * (j + 1) <= row) {   // group-uniform: the whole round is inside the bucket
                    const f4 r = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(r, (f4*)(p.out + e));
                    store_side4_row<MODE>(p, e, side);     // every DPP row holds 16 consecutive float4s, all lanes active
                } else if (r0 >= 0 && r0 + 4 <= row) {     // the float4 belongs to this bucket alone
                    const f4 r = {o[0], o[1], o[2], o[3]};
                    __builtin_nontemporal_store(r, (f4*)(p.out + e));
                    store_side4<MODE>(p, e, side);
                } else {                                   // shared with a neighbour (or outside the bucket): own elements only
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if ((unsigned)(r0 + c) < (unsigned)row) {
                            p.out[e + c] = o[c];
                            store_side1<MODE>(p, e + c, side[c]);
                        }
                    }
                }
            }
        };
        if (fast) body(std::true_type{}); else body(std::false_type{});
        if (MODE == MODE_SCALE && row < (int)p.row) {
            // scale_down returns the padded layout: the short last bucket is filled up with copies of the scaled last
            // element (ref: help_functions.py:76-86)
            float ul = (prep_on ? prep(p.x[p.n - 1], pp) : p.x[p.n - 1]) - b;
            ul = ul / a;
            for (int64_t i = lo + row + lane; i < lo + p.row; i += GL) p.out[i] = ul;
        }
    }
    // nbk < p.nb only for the instances above 16 float4 per lane when the tensor's length is not a multiple of 4 (see the
    // launcher): the bucket(s) whose last float4 would reach past the end of the tensor, block 0, scalar accesses
    if (V > 16 && nbk < p.nb && blockIdx.x == 0) tail_buckets<MODE>(p, T, nbk, pp);
}

// ---- generic path, small/medium rows: one lane group (16 lanes or a wave) per bucket, 256-thread
// blocks, no LDS, no barrier.  Any row length / alignment.
template <int MODE, int LANES>
__global__ __launch_bounds__(256) void k_bucket_groups(KParams p) {
    PointStore Ts;
    PointTable Tc;
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const int64_t ngroups = ((int64_t)gridDim.x * blockDim.x) / LANES;
    const int l = threadIdx.x % LANES;
    const bool vec4 = (p.row & 3) == 0 && (((((uintptr_t)p.x) | ((uintptr_t)p.out)) & kDataAlign) == 0);
    for (int64_t bkt = group; bkt < p.nb; bkt += ngroups) {
        const int64_t lo = bkt * p.row;
        const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
        if (vec4 && hi - lo == p.row) bucket_lanes4<MODE, LANES>(p, T, bkt, lo, l, pp);
        else bucket_lanes<MODE, LANES>(p, T, bkt, lo, hi, l, pp);
    }
}

// ---- generic path, huge rows: one block per bucket, any row length / alignment ----------------
template <int MODE>
__global__ __launch_bounds__(1024) void k_bucket_generic(KParams p) {
    PointStore Ts;
    PointTable Tc;
    __shared__ float red[32];
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    for (int64_t bkt = blockIdx.x; bkt < p.nb; bkt += gridDim.x) {
        const int64_t lo = bkt * p.row;
        const int64_t hi = lo + p.row < p.n ? lo + p.row : p.n;
        float a, b;
        if (MODE == MODE_NEAREST && p.prescaled) {
            a = p.alpha[bkt]; b = p.beta[bkt];
        } else {
            float mn = INFINITY, mx = -INFINITY;
            int nan = 0;
            for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const float v = prep(p.x[i], pp);
                mn = fminf(mn, v); mx = fmaxf(mx, v);
                nan |= (v != v);
            }
            block_minmax(mn, mx, red);
            if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }
            alpha_beta(mn, mx, a, b);
            if (threadIdx.x == 0) {
                if (p.alpha) p.alpha[bkt] = a;
                if (p.beta) p.beta[bkt] = b;
            }
        }
        for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
            float v = p.x[i];
            if (!(MODE == MODE_NEAREST && p.prescaled)) v = prep(v, pp);
            float rnd = 0.0f;
            if (MODE == MODE_QDQ && p.stochastic) {
                float r4[4];
                philox_uniform4(p.seed, (uint64_t)i >> 2, r4);
                rnd = r4[i & 3];
            }
            float side = 0.0f;
            const float y = transform<MODE>(p, T, v, a, b, pp.mean, rnd, side);
            p.out[i] = y;
            store_side1<MODE>(p, i, side);
        }
        if (MODE == MODE_SCALE) {
            const int64_t end = lo + p.row;
            if (hi < end && hi == p.n && p.nb > 1) {
                float dummy;
                const float u_last = transform<MODE>(p, T, prep(p.x[p.n - 1], pp), a, b, pp.mean, 0.0f, dummy);
                for (int64_t i = hi + threadIdx.x; i < end; i += blockDim.x) p.out[i] = u_last;
            }
        }
    }
}

// ---- single-bucket (bucket_size=None) path for large tensors: reduce, finalize, apply --------
// stage 1: per-block partial min/max of prep(x)
__global__ __launch_bounds__(256) void k_minmax_partial(const float* x, int64_t n, const float* mean, float me,
                                                        float* part /* [2*kPartialBlocks] */) {
    __shared__ float red[32];
    Prep pp;
    pp.mean = mean ? *mean : 0.0f;
    pp.me = me;
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    if ((((uintptr_t)x) & kDataAlign) == 0) {
        const int64_t n4 = n >> 2;
        const f4* x4 = (const f4*)x;
        for (int64_t i = tid; i < n4; i += nth) {
            const f4 v = prep4(x4[i], pp);          // plain load: keep the lines in L2/MALL for stage 3
            mn = fminf(mn, fminf(fminf(v.x, v.y), fminf(v.z, v.w)));
            mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            nan |= has_nan4(v);
        }
        for (int64_t i = (n4 << 2) + tid; i < n; i += nth) {
            const float v = prep(x[i], pp);
            mn = fminf(mn, v); mx = fmaxf(mx, v);
            nan |= (v != v);
        }
    } else {
        for (int64_t i = tid; i < n; i += nth) {
            const float v = prep(x[i], pp);
            mn = fminf(mn, v); mx = fmaxf(mx, v);
            nan |= (v != v);
        }
    }
    block_minmax(mn, mx, red);
    if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }      // NaN poisons the partial (and, in stage 2, the tensor)
    if (threadIdx.x == 0) { part[blockIdx.x] = mn; part[kPartialBlocks + blockIdx.x] = mx; }
}

// stage 2: one block folds the partials into alpha/beta (device scalars; no host sync, unlike
// the reference's `alpha[0] < tol` at quant_functions.py:96)
__global__ __launch_bounds__(256) void k_minmax_final(const float* part, int nparts, float* ab /* [2] */,
                                                      float* alpha_out, float* beta_out) {
    __shared__ float red[32];
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
        const float pm = part[i];
        nan |= (pm != pm);
        mn = fminf(mn, pm);
        mx = fmaxf(mx, part[kPartialBlocks + i]);
    }
    block_minmax(mn, mx, red);
    if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }
    if (threadIdx.x == 0) {
        float a, b;
        alpha_beta(mn, mx, a, b);
        ab[0] = a; ab[1] = b;
        if (alpha_out) alpha_out[0] = a;
        if (beta_out) beta_out[0] = b;
    }
}

// stage 3: elementwise apply with the single (alpha, beta)
// ab != null: the pair was finalised by k_minmax_final (huge tensors: many apply blocks).
// ab == null: every block folds the `nparts` stage-1 partials itself (a few KB from L2) -- one
// launch fewer for the model-sized tensors, where the call is launch-bound, not bandwidth-bound.
template <int MODE>
__global__ __launch_bounds__(256) void k_single_apply(KParams p, const float* ab, const float* part, int nparts) {
    PointStore Ts;
    PointTable Tc;
    __shared__ float red[32];
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    float a, b;
    if (ab) {
        a = ab[0]; b = ab[1];
    } else {
        float mn = INFINITY, mx = -INFINITY;
        int nan = 0;
        for (int i = threadIdx.x; i < nparts; i += blockDim.x) {
            const float pm = part[i];
            nan |= (pm != pm);
            mn = fminf(mn, pm);
            mx = fmaxf(mx, part[kPartialBlocks + i]);
        }
        block_minmax(mn, mx, red);
        if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }
        alpha_beta(mn, mx, a, b);
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (p.alpha) p.alpha[0] = a;
            if (p.beta) p.beta[0] = b;
        }
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const bool prescaled = (MODE == MODE_NEAREST && p.prescaled);
    int64_t done = 0;
    if (((((uintptr_t)p.x) | ((uintptr_t)p.out)) & kDataAlign) == 0) {
        const int64_t n4 = p.n >> 2;
        const f4* x4 = (const f4*)p.x;
        f4* o4 = (f4*)p.out;
        for (int64_t i = tid; i < n4; i += nth) {
            f4 v = __builtin_nontemporal_load(x4 + i);
            if (!prescaled) v = prep4(v, pp);
            float rnd[4] = {0.f, 0.f, 0.f, 0.f};
            if (MODE == MODE_QDQ && p.stochastic) philox_uniform4(p.seed, (uint64_t)i, rnd);
            float side[4];
            const f4 r = transform_f4<MODE>(p, T, v, a, b, pp.mean, rnd, side);
            __builtin_nontemporal_store(r, o4 + i);
            store_side4<MODE>(p, i << 2, side);
        }
        done = n4 << 2;
    }
    for (int64_t i = done + tid; i < p.n; i += nth) {
        float v = p.x[i];
        if (!prescaled) v = prep(v, pp);
        float rnd = 0.0f;
        if (MODE == MODE_QDQ && p.stochastic) {
            float r4[4];
            philox_uniform4(p.seed, (uint64_t)i >> 2, r4);
            rnd = r4[i & 3];
        }
        float side = 0.0f;
        p.out[i] = transform<MODE>(p, T, v, a, b, pp.mean, rnd, side);
        store_side1<MODE>(p, i, side);
    }
}

// ---- single-bucket path in ONE launch for tensors that fit the register files ------------------
// k_single_fused: every lane loads its share of the tensor ONCE into registers (V float4 per lane), the blocks
// publish their min/max, meet at a grid-wide barrier, and transform their registers with the folded (alpha, beta):
// the tensor is read from HBM once and written once (8 B/element instead of 12) in one launch instead of three
// (ref: quant_functions.py:85-87,95-97 with bucket_size=None).
//
// The barrier has NO counter.  Device-scope round trips cost 1-2 us on this chip (the coherence point is behind
// the XCDs' private L2s) and same-address atomics serialise (~40 ns each): a counter that 780 blocks increment
// and poll measured +35 us, a 32-way fan-out of it +22 us.  Instead every block writes its (min, max) into its
// own slot, tagged with the launch's epoch (two 8-byte atomic stores, nothing to wait for), and then polls ALL G
// slots -- each lane checks G/256 of them, one round trip per sweep -- until every slot carries this epoch; the
// last sweep IS the fold.  No read-modify-write, no fence, no serialisation.
//
// The barrier is also OPTIMISTIC, and a block that gives up depends on NOBODY.  A grid barrier needs all blocks
// resident at the same time; the launch is sized for that (<= one block per CU), but another stream or process may
// hold part of the GPU, and two such kernels could starve each other forever.  So a block that still misses a slot
// after 2 ms of the 100 MHz wall clock stops waiting and folds the min/max of the WHOLE tensor itself, from memory
// (at most 1 Mi elements = 4 MiB, served by L2 / the Infinity Cache; x is never written: in-place calls take the
// three-launch path), then transforms its own registers like everybody else.  min / max do not depend on the fold order,
// so it arrives at the same (alpha, beta) bit for bit as the blocks that did meet.  There is no departure count, no
// "last block", no flag that one block sets and another must see: every block's output depends only on x and on slot
// values it has itself observed with this launch's tag (round 2's protocol -- a relaxed gave_up counter read by the last
// block out -- could miss a departure and leave a slice unwritten).
// A slot is two 8-byte words {min bits | tag_min << 32}, {max bits | tag_max << 32}; each word is written and read with
// one atomic access, so a value can never be seen with another launch's tag.  BOTH tags are unique to the launch among
// all launches that can have touched the slot: the host draws a 64-bit epoch that never repeats and sets
// tag_min = its low half, tag_max = low half ^ (high half * odd constant), both non-zero (a never-written slot is 0 | 0).
// A slot left over from another launch matches tag_min only if its epoch has the same low half, i.e. lies 2^32 launches
// back, and then its tag_max differs because the high half does.  (Tagging the two words with the two HALVES of the epoch
// does not work: the high half is the same for 2^32 launches in a row, so a sweep could pair this launch's min with the max
// a previous launch left in the slot.  A 31-bit epoch alone wrapped after 2^31 launches, and slots that only large grids
// touch could still carry the old tag.)  Should two launches ever share a slot set while both are running (more than
// kFusedSlots of these kernels in flight), they overwrite each other's tags, their sweeps fail, and both take the give-up
// path: slow, still correct.  The slots live in a __device__ array (zero-initialised when the module is loaded, per device
// and per process), not in the caller's workspace, whose contents are undefined by contract.
constexpr int kFusedSlots = 64;
constexpr int kFusedMaxBlocks = 256;                   // <= one block per CU: every block sweeps all G slots
constexpr long long kBarrierTimeout = 200000;          // 2 ms of the 100 MHz wall clock
struct FusedCtl {
    unsigned long long slot[kFusedMaxBlocks][2];
};
__device__ FusedCtl g_fused_ctl[kFusedSlots];

// test hooks of qd_set_single_fused_mode(): which blocks skip the barrier and take the give-up path at once
enum { FUSED_GIVE_UP_NONE = 0, FUSED_GIVE_UP_ALL = 1, FUSED_GIVE_UP_EVERY_7TH = 2, FUSED_GIVE_UP_ONE = 3 };

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int MODE>
__device__ __forceinline__ f4 transform4(const KParams& p, const PointTable* T, f4 v, float a, float b, float mean,
                                         int64_t i4, float (&side)[4]) {
    float rnd[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == MODE_QDQ && p.stochastic) philox_uniform4(p.seed, (uint64_t)i4, rnd);
    return transform_f4<MODE>(p, T, v, a, b, mean, rnd, side);
}

// One sweep over the G slots (all threads of the block): true when every slot carries this launch's epoch; then
// (mn, mx) is the fold of all partials (NaN in any of them poisons both, as torch's min/max do).
__device__ __forceinline__ bool sweep_slots(const FusedCtl* ctl, unsigned G, unsigned tag_min, unsigned tag_max, float* red,
                                            float& mn, float& mx) {
    float fmn = INFINITY, fmx = -INFINITY;
    int fnan = 0, missing = 0;
    for (unsigned i = threadIdx.x; i < G; i += blockDim.x) {
        const unsigned long long a = ld_agent(&ctl->slot[i][0]);
        const unsigned long long b = ld_agent(&ctl->slot[i][1]);
        missing |= ((unsigned)(a >> 32) != tag_min) | ((unsigned)(b >> 32) != tag_max);
        const float pm = __uint_as_float((unsigned)a), px = __uint_as_float((unsigned)b);
        fnan |= (pm != pm);
        fmn = fminf(fmn, pm);
        fmx = fmaxf(fmx, px);
    }
    if (__syncthreads_or(missing)) return false;
    block_minmax(fmn, fmx, red);
    if (__syncthreads_or(fnan)) { fmn = NAN; fmx = NAN; }
    mn = fmn; mx = fmx;
    return true;
}

// V float4 per lane at W waves per SIMD (W = 4: 128 VGPRs).  The launcher uses it up to 1 Mi elements (V <= 4).
template <int MODE, int V, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
void k_single_fused(KParams p, int slot_set, unsigned tag_min, unsigned tag_max, int give_up_mode) {
    PointStore Ts;
    PointTable Tc;
    __shared__ float red[32];
    __shared__ int s_timed_out;
    const PointTable* T = &Tc;                             // (read only in the nearest-point mode)
    if (MODE == MODE_NEAREST) load_points(Tc, Ts, p.pts, p.k, p.fine);
    Prep pp;
    pp.mean = p.mean ? *p.mean : 0.0f;
    pp.me = p.me;
    const unsigned G = gridDim.x;
    FusedCtl* ctl = &g_fused_ctl[slot_set];
    const int64_t n4 = p.n >> 2;
    const f4* x4 = (const f4*)p.x;
    f4* o4 = (f4*)p.out;
    const bool forced = give_up_mode == FUSED_GIVE_UP_ALL ||
                        (give_up_mode == FUSED_GIVE_UP_EVERY_7TH && blockIdx.x % 7 == 3) ||
                        (give_up_mode == FUSED_GIVE_UP_ONE && blockIdx.x == G / 2);

    // ---- load once, reduce ----
    f4 v[V];
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int64_t i = ((int64_t)j * G + blockIdx.x) * 256 + threadIdx.x;
        if (i < n4) {
            v[j] = prep4(ldg_nt(x4 + i), pp);
            mn = fminf(mn, fminf(fminf(v[j].x, v[j].y), fminf(v[j].z, v[j].w)));
            mx = fmaxf(mx, fmaxf(fmaxf(v[j].x, v[j].y), fmaxf(v[j].z, v[j].w)));
            nan |= has_nan4(v[j]);
        }
    }
    float tail[3] = {0.f, 0.f, 0.f};
    const int ntail = (int)(p.n & 3);
    const bool owns_tail = blockIdx.x == 0 && threadIdx.x == 0;
    if (owns_tail)
        for (int t = 0; t < ntail; ++t) {
            tail[t] = prep(p.x[(n4 << 2) + t], pp);
            mn = fminf(mn, tail[t]); mx = fmaxf(mx, tail[t]);
            nan |= (tail[t] != tail[t]);
        }
    block_minmax(mn, mx, red);
    if (__syncthreads_or(nan)) { mn = NAN; mx = NAN; }      // NaN poisons the partial, hence the tensor

    // ---- publish (also by a block that is about to give up: the others must not wait for it) ----
    if (threadIdx.x == 0) {
        st_agent(&ctl->slot[blockIdx.x][0], (unsigned long long)__float_as_uint(mn) | ((unsigned long long)tag_min << 32));
        st_agent(&ctl->slot[blockIdx.x][1], (unsigned long long)__float_as_uint(mx) | ((unsigned long long)tag_max << 32));
        s_timed_out = forced ? 1 : 0;
    }
    __syncthreads();

    // ---- meet: sweep the slots until all carry this epoch (the successful sweep is the fold) ----
    bool met = false;
    if (!forced) {
        const long long t0 = wall_clock64();
        for (;;) {
            if (sweep_slots(ctl, G, tag_min, tag_max, red, mn, mx)) { met = true; break; }
            if (threadIdx.x == 0 && wall_clock64() - t0 > kBarrierTimeout) s_timed_out = 1;
            __syncthreads();
            if (s_timed_out) break;
            __builtin_amdgcn_s_sleep(4);
        }
    }
    if (!met) {
        // ---- gave up: this block folds the whole tensor alone (same min / max, whatever the order) ----
        float gmn = INFINITY, gmx = -INFINITY;
        int gnan = 0;
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            const f4 t = prep4(x4[i], pp);
            gmn = fminf(gmn, fminf(fminf(t.x, t.y), fminf(t.z, t.w)));
            gmx = fmaxf(gmx, fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)));
            gnan |= has_nan4(t);
        }
        if (threadIdx.x == 0)
            for (int t = 0; t < ntail; ++t) {
                const float e = prep(p.x[(n4 << 2) + t], pp);
                gmn = fminf(gmn, e); gmx = fmaxf(gmx, e);
                gnan |= (e != e);
            }
        block_minmax(gmn, gmx, red);
        if (__syncthreads_or(gnan)) { gmn = NAN; gmx = NAN; }
        mn = gmn; mx = gmx;
    }

    float a, b;
    alpha_beta(mn, mx, a, b);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (p.alpha) p.alpha[0] = a;
        if (p.beta) p.beta[0] = b;
    }
    // ---- transform the registers ----
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int64_t i = ((int64_t)j * G + blockIdx.x) * 256 + threadIdx.x;
        if (i < n4) {
            float side[4];
            const f4 r = transform4<MODE>(p, T, v[j], a, b, pp.mean, i, side);
            stg_nt(r, o4 + i);
            store_side4<MODE>(p, i << 2, side);
        }
    }
    if (owns_tail)
        for (int t = 0; t < ntail; ++t) {
            const int64_t e = (n4 << 2) + t;
            float rnd = 0.0f;
            if (MODE == MODE_QDQ && p.stochastic) {
                float r4[4];
                philox_uniform4(p.seed, (uint64_t)e >> 2, r4);
                rnd = r4[e & 3];
            }
            float side = 0.0f;
            p.out[e] = transform<MODE>(p, T, tail[t], a, b, pp.mean, rnd, side);
            store_side1<MODE>(p, e, side);
        }
}

// ================================ host side ====================================================

struct Workspace {           // fixed carve-up of the caller's scratch buffer
    float* minmax_part;      // [2*kPartialBlocks]
    float* ab;               // [2]
    double* sum_part;        // [kPartialBlocks]
    float* arg_pv;           // [2*kPartialBlocks]
    int64_t* arg_pi;         // [2*kPartialBlocks]
    float* pg_part;          // [kPartialBlocks * kMaxPoints]
};
constexpr size_t kWsBytes = 64 * 1024 + (size_t)kPartialBlocks * kMaxPoints * sizeof(float);

bool carve(void* ws, size_t bytes, Workspace& w) {
    if (!ws || bytes < kWsBytes || (((uintptr_t)ws) & 15)) return false;
    char* p = (char*)ws;
    w.minmax_part = (float*)p;               p += 2 * kPartialBlocks * sizeof(float);     // 8 KiB
    w.ab = (float*)p;                        p += 64;
    w.sum_part = (double*)p;                 p += kPartialBlocks * sizeof(double);        // 8 KiB
    w.arg_pv = (float*)p;                    p += 2 * kPartialBlocks * sizeof(float);     // 8 KiB
    w.arg_pi = (int64_t*)p;                  p += 2 * kPartialBlocks * sizeof(int64_t);   // 16 KiB
    p = (char*)ws + 64 * 1024;
    w.pg_part = (float*)p;
    return true;
}

inline void geometry(int64_t n, int64_t bucket, int64_t& nb, int64_t& row) {
    if (bucket <= 0 || n < bucket) { nb = 1; row = n; return; }
    row = bucket;
    nb = (n + bucket - 1) / bucket;
}

// compute units of the current device (256 on MI355X), cached PER DEVICE (qd_common.h)
inline int num_cus() { return device_cus(); }

inline int grid_cap() {
    // Measured on MI355X (tools/tune_k1.py, docs/history/profiles/r01_tune.txt): one wave-tile per wave (no grid-stride
    // reuse) streams fastest -- 85.9 us vs 97 us at 2048 persistent blocks for the 64 Mi-element
    // headline tensor -- so the cap only bounds the grid dimension.
#ifdef QD_TUNING        // launch-geometry experiments only (build with -DQD_TUNING): QD_GRID_CAP=<blocks>
    const char* e = getenv("QD_GRID_CAP");
    const int v = e ? atoi(e) : 0;
    if (v > 0) return v;
#endif
    return 1 << 20;
}
inline int blocks_for(int64_t items, int per_block) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    const int cap = grid_cap();
    return (int)(b < cap ? b : cap);
}

inline int check_launch() {
    const hipError_t e = hipGetLastError();
    return (int)e;
}

// launch the bucketed transform for nb > 1 (or a short single bucket)
template <int MODE>
int launch_bucketed(KParams& p, hipStream_t st) {
    const bool aligned = ((((uintptr_t)p.x) | ((uintptr_t)p.out)) & kDataAlign) == 0 &&
                         (MODE != MODE_NEAREST || p.idx == nullptr || p.idx_bytes != 8 || (((uintptr_t)p.idx) & 15) == 0) &&
                         (MODE != MODE_QDQ || p.lev8 == nullptr || (((uintptr_t)p.lev8) & 3) == 0);
    const int64_t nfull = p.n / p.row;                 // leading full buckets
    const size_t tb = MODE == MODE_NEAREST ? point_table_bytes(p.k, p.fine) : 0;      // dynamic LDS of every launch: the point table
    // with the fine cell table the grid is capped (the kernels loop): the table is built once per resident block, not per tile
    const int fine_cap = (MODE == MODE_NEAREST && p.fine) ? kFineBlocksPerCu * num_cus() : (1 << 30);
    const size_t tbc = MODE == MODE_NEAREST ? point_table_bytes(p.k, 0) : 0;           // the chunk kernels': always the coarse table
    // k_bucket_chunk_any stages level / point indices that fit a byte in LDS (the same expression as in the kernel)
    const bool stage8 = (MODE == MODE_QDQ && p.lev8 != nullptr) || (MODE == MODE_NEAREST && p.idx != nullptr && p.k <= 32);
#define QD_VEC(LPB, V, U)                                                                       \
    {                                                                                           \
        p.nvec = nfull;                                                                         \
        constexpr int64_t bpw = (64 / LPB) * U;                                                 \
        const int64_t tiles = (nfull + bpw - 1) / bpw;                                          \
        int blocks = blocks_for(tiles, 4) + 1; /* +1: the block that owns the tail */           \
        /* the vector kernel hides the narrowed search behind 7 waves per SIMD up to 64 points (99 us against 104 us with */ \
        /* the fine table on a capped grid); the kernels with fewer resident waves gain from 33 points on */ \
        if (MODE == MODE_NEAREST && p.k <= 64) p.fine = 0;                                      \
        if (MODE == MODE_NEAREST && p.fine && blocks > fine_cap) blocks = fine_cap;             \
        hipLaunchKernelGGL((k_bucket_vec<MODE, LPB, V, U>), dim3(blocks), dim3(256),            \
                           MODE == MODE_NEAREST ? point_table_bytes(p.k, p.fine) : 0, st, p);   \
        return check_launch();                                                                  \
    }
    if (aligned && p.nb > 1) {
        switch (p.row) {
            case 64: QD_VEC(16, 1, 4)
            case 128: QD_VEC(16, 2, 2)
            case 256: QD_VEC(16, 4, 1)
            case 512: QD_VEC(64, 2, 2)
            case 1024: QD_VEC(64, 4, 1)
            case 2048: QD_VEC(64, 8, 1)
            default: break;       // 4096, 8192: k_bucket_wave_any with two / four waves per bucket (8192: 91.2 vs 95.5 us as 32 float4 per lane)
        }
    }
#undef QD_VEC
    p.nvec = 0;
    if (aligned && p.nb > 1 && p.row > 256 && p.row <= 32768) {
        // one wave per bucket, any size (k_bucket_wave_any): sizes above 512, and sizes from 448 that are not a multiple of 4
        // (multiples of 4 up to 512 stay with the chunk kernel: 300 -> 90 us against 127 us here; the vector sizes 512 /
        // 1024 / 2048 were taken above).  The lane -> float4 mapping starts at the 128-byte line (32 elements) at or below
        // the bucket (measured against 16- and 64-element boundaries: docs/history/profiles/r02_tune_kernels.txt).
        const bool mult4 = (p.row & 3) == 0;
        constexpr int al = 32;
        const bool line_ok = (p.row * 4) % (al * 4) == 0;                        // every bucket starts on the boundary anyway
        const int64_t amask = ~(int64_t)(al - 1);
        // float4s a wave may have to hold: the bucket's own, +1 for a split first/last one, + the lead-in from the boundary
        const int nf_max = (int)(p.row >> 2) + (mult4 ? 0 : 2) + (line_ok ? 0 : al / 4 - 1);
        // every bucket, the short last one included (the float4 that would reach past the end of the tensor is fetched as
        // x[n-4 .. n-1] and rotated: nothing outside the tensor is read, whatever the alignment of the base)
        // (the two largest instances have no registers to spare for that and leave those buckets to block 0's scalar path)
        int64_t nbk_all = p.nb, nbk_whole = p.nb;
        while (nbk_whole > 0 && ((((nbk_whole * p.row < p.n ? nbk_whole * p.row : p.n) + 3) >> 2) << 2) > p.n) --nbk_whole;
        if (nf_max <= 256 * 32 && (p.row > 512 || (!mult4 && p.row >= 448))) {
#define QD_WAVE_ANY(V, G)                                                                                  \
    {                                                                                                      \
        const int64_t nbk = V > 16 ? nbk_whole : nbk_all;                                                  \
        int blocks = blocks_for(nbk > 0 ? nbk : 1, 4 / G);                                                 \
        if (blocks > fine_cap) blocks = fine_cap;                                                          \
        hipLaunchKernelGGL((k_bucket_wave_any<MODE, V, G>), dim3(blocks), dim3(256), tb, st, p, nbk, amask); \
        return check_launch();                                                                             \
    }
            // one wave per bucket up to 8 rounds (2048 elements), then two (up to 4096) and four waves per bucket
            if constexpr (MODE == MODE_QDQ) {              // the hot mode: rounds in steps of one
                if (nf_max <= 64 * 2) QD_WAVE_ANY(2, 1)
                if (nf_max <= 64 * 3) QD_WAVE_ANY(3, 1)
                if (nf_max <= 64 * 4) QD_WAVE_ANY(4, 1)
                if (nf_max <= 64 * 5) QD_WAVE_ANY(5, 1)
                if (nf_max <= 64 * 6) QD_WAVE_ANY(6, 1)
                if (nf_max <= 64 * 7) QD_WAVE_ANY(7, 1)
                if (nf_max <= 64 * 8) QD_WAVE_ANY(8, 1)
                if (nf_max <= 128 * 5) QD_WAVE_ANY(5, 2)
                if (nf_max <= 128 * 6) QD_WAVE_ANY(6, 2)
                if (nf_max <= 128 * 7) QD_WAVE_ANY(7, 2)
                if (nf_max <= 128 * 8) QD_WAVE_ANY(8, 2)
                if (nf_max <= 256 * 5) QD_WAVE_ANY(5, 4)
                if (nf_max <= 256 * 6) QD_WAVE_ANY(6, 4)
                if (nf_max <= 256 * 7) QD_WAVE_ANY(7, 4)
                if (nf_max <= 256 * 8) QD_WAVE_ANY(8, 4)
                if (nf_max <= 256 * 9) QD_WAVE_ANY(9, 4)
                // above 8192 elements: more float4s per lane again (8200 / 10000 / 16384 / 20000 took 148 / 182 / 155 / 284 us on
                // the two-pass kernels below)
                if (nf_max <= 256 * 12) QD_WAVE_ANY(12, 4)
                if (nf_max <= 256 * 16) QD_WAVE_ANY(16, 4)
                if (nf_max <= 256 * 24) QD_WAVE_ANY(24, 4)
                QD_WAVE_ANY(32, 4)
            } else {                                       // scale_down, nearest point: fewer instances
                if (nf_max <= 64 * 3) QD_WAVE_ANY(3, 1)
                if (nf_max <= 64 * 5) QD_WAVE_ANY(5, 1)
                if (nf_max <= 64 * 8) QD_WAVE_ANY(8, 1)
                if (nf_max <= 128 * 6) QD_WAVE_ANY(6, 2)
                if (nf_max <= 128 * 8) QD_WAVE_ANY(8, 2)
                if (nf_max <= 256 * 6) QD_WAVE_ANY(6, 4)
                if (nf_max <= 256 * 9) QD_WAVE_ANY(9, 4)
                // more float4s per lane: scale_down only -- the point search of the nearest-point mode keeps too much live
                // and the compiler demotes the float4 array to scratch memory (272 / 528 bytes per lane); buckets above 9216
                // elements take the two-pass kernels below in that mode
                if constexpr (MODE == MODE_SCALE) {
                    if (nf_max <= 256 * 16) QD_WAVE_ANY(16, 4)
                    QD_WAVE_ANY(32, 4)
                }
            }
#undef QD_WAVE_ANY
        }
    }
    // float4 per lane a chunk holds.  Measured at 64 Mi elements, bucket 100 / 36 / 300 / 1000 / 2000 (the one-bucket-
    // per-lane-group kernels below: 145 / 211 / 167 / 114 / 122 us): 16 -> 123 / 123 / 128 / 122 / 120 us (195 VGPRs,
    // two waves per SIMD: the load and the compute phase of a wave do not overlap); 8 -> 100 / 100 / 103 / 108 / 108;
    // 4 -> 124 / 131 / 130 / 118 / 121.  Bucket 12 / 4: 101 / 112 us instead of 571 / 1616.
    constexpr int kChunkV = 8;
    if (aligned && p.nb > 1 && (p.row & 3) == 0 && p.row <= (int64_t)kChunkV * 256) {
        const int64_t bq = p.row >> 2;
        // as many whole buckets as fit the chunk's kChunkV * 64 float4 (<= 256: the (alpha, beta) table), not the next power of
        // two below it: bucket 36 fills 504 of the 512 float4 with m = 56 instead of 288 with m = 32.  Above 64 buckets a lane
        // reduces whole buckets alone, below that 64 / m lanes share one, so m is then rounded down to a power of two.
        int m = (int)((kChunkV * 64) / bq);
        if (m > 256) m = 256;
        if (m < 64) { int p2 = 1; while (p2 * 2 <= m) p2 *= 2; if (m < 48 || p2 == m) m = p2; else { /* 48..63 lanes, one bucket each */ } }
        const int64_t nchunks = nfull / m;
        if (nchunks > 0) {
            const size_t lds = (size_t)2 * (kChunkV * 64 + 256 + 128) * sizeof(float2);   // two waves per block: pairs, (alpha, beta), 1/alpha
            const int blocks = blocks_for(nchunks, 2) + 1;                             // +1: the block that owns the tail
            p.fine = 0;                                                                // the chunk kernels stage data in LDS: coarse table
            hipLaunchKernelGGL((k_bucket_chunk<MODE, kChunkV>), dim3(blocks), dim3(128), lds + tbc, st, p, m, nchunks);
            return check_launch();
        }
    }
    if (aligned && p.nb > 1 && (p.row & 3) != 0 && p.row * 4 <= (int64_t)kChunkV * 256) {
        // a multiple of 4 (chunks start 16-byte aligned), as many buckets as fit kChunkV * 256 elements: bucket 33 fills
        // 1980 of the 2048 elements with m = 60 instead of 1056 with m = 32
        int m = (int)(((int64_t)kChunkV * 256 - 28) / p.row) & ~3;        // - 28 elements: the lead-in to the 128-byte line
        const int lead = m >= 4;
        if (!lead) m = 4;                                                  // 506 .. 511: four buckets fill the chunk, no lead-in
        // (no upper limit on m: the kernel keeps no per-bucket table; bucket sizes 1, 2, 3, 5, 7 fill the chunk too.  Above 64
        // buckets every lane reduces whole buckets alone: a multiple of 64 keeps all lanes busy in every round -- bucket 7:
        // 102 us with m = 256, 107-110 us with m = 288)
        if (m >= 64) m &= ~63;
        if (m < 64) { int p2 = 4; while (p2 * 2 <= m) p2 *= 2; if (m < 48 || p2 == m) m = p2; }
        const int64_t nchunks = nfull / m;
        if (nchunks > 0) {
            // two waves: the staged chunk, + a byte per element when level / point indices (<= 255) are asked for
            const size_t lds = (size_t)2 * (kChunkV * 256 + (stage8 ? kChunkV * 64 : 0)) * sizeof(float);
            const int blocks = blocks_for(nchunks, 2) + 1;
            p.fine = 0;
            hipLaunchKernelGGL((k_bucket_chunk_any<MODE, kChunkV>), dim3(blocks), dim3(128), lds + tbc, st, p, m, nchunks, lead);
            return check_launch();
        }
    }
    // (bucket sizes 513 .. 1023 that are not a multiple of 4 once had a 16-float4 instance of k_bucket_chunk_any here; since
    // k_bucket_wave_any takes every size above 512 the branch was unreachable -- tools/launch_coverage.py -- and is gone)
    if (p.row <= 256) {                                  // 16 buckets per block, a DPP row each
        hipLaunchKernelGGL((k_bucket_groups<MODE, 16>), dim3(blocks_for(p.nb, 16) < fine_cap ? blocks_for(p.nb, 16) : fine_cap), dim3(256), tb, st, p);
    } else if (p.row <= 16384) {                         // 4 buckets per block, one wave each
        hipLaunchKernelGGL((k_bucket_groups<MODE, 64>), dim3(blocks_for(p.nb, 4) < fine_cap ? blocks_for(p.nb, 4) : fine_cap), dim3(256), tb, st, p);
    } else {
        hipLaunchKernelGGL((k_bucket_generic<MODE>), dim3(blocks_for(p.nb, 1) < fine_cap ? blocks_for(p.nb, 1) : fine_cap), dim3(1024), tb, st, p);
    }
    return check_launch();
}

// one-launch single bucket (k_single_fused) when the tensor fits the register files of a resident grid
constexpr int kNotFused = -1000;
// qd_set_single_fused_mode(): 0 = always the three-launch path, 1 = default, 2 / 3 / 4 = every block / every block with
// blockIdx % 7 == 3 / exactly one block skips the barrier and takes the give-up path, so that the tests reach the
// contention fallback deterministically on an idle GPU.  A plain int: set it before launching, not concurrently.
// (the switch itself lives in qd_kernels.hip -- one variable for the three translation units that include this header)
template <int MODE, int V, int W>
int fused_capacity() {                                         // blocks of k_single_fused<MODE, V> resident at once, 0 if unusable
    static int caps[64];                                       // per device, stored + 1 (0 = not asked yet)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    const bool cached = dev >= 0 && dev < 64;                  // a device index beyond the table is asked every time, never aliased
    int& slot = caps[cached ? dev : 0];
    int cap = cached ? __atomic_load_n(&slot, __ATOMIC_RELAXED) - 1 : -1;
    if (cap < 0) {
        int per_cu = 0;
        hipFuncAttributes fa;
        const void* fn = (const void*)k_single_fused<MODE, V, W>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, MODE == MODE_NEAREST ? point_table_bytes(kMaxPoints, 1) : 0) != hipSuccess ||
            hipFuncGetAttributes(&fa, fn) != hipSuccess || fa.localSizeBytes != 0 /* spills: not worth it */)
            per_cu = 0;
        (void)hipGetLastError();
        int c = per_cu * num_cus();
        cap = c > kFusedMaxBlocks ? kFusedMaxBlocks : c;
        if (cached) __atomic_store_n(&slot, cap + 1, __ATOMIC_RELAXED);
    }
    return cap;
}
// one epoch sequence for ALL instantiations, 64 bits: a tag never repeats on a slot set (epoch 0 = a never-written slot)
std::atomic<uint64_t> next_launch{1};
// Measured on MI355X (docs/history/profiles/r02_k1g_fused.txt): a device-scope round trip costs 1.5-2 us, so the barrier adds ~3.5 us
// to a kernel -- about what a kernel boundary costs -- and the load and store phases of the register-resident kernel do
// not overlap, while the three-launch path's second read is served by the 256 MiB Infinity Cache.  One launch wins only
// where the call is launch-bound: up to 1 Mi elements (GPU time 7.8 vs 9.7 us at 0.1 M, 10.8 vs 10.9 at 0.8 M; one host
// launch instead of two); at 2.8 M / 5.3 M elements it measured 22 / 21 us against 16 / 17.5 us.
constexpr int64_t kFusedMaxN = (int64_t)1 << 20;
template <int MODE>
int launch_single_fused(KParams& p, hipStream_t st) {
    const int fmode = qd_fused_mode_state;
    if (fmode == 0 || ((((uintptr_t)p.x) | ((uintptr_t)p.out)) & kDataAlign) != 0) return kNotFused;
    if ((const void*)p.x == (const void*)p.out) return kNotFused;   // in place: a block that gives up re-reads x
    // a captured launch would bake its barrier slot and epoch into the graph, and two replays in flight would share them
    hipStreamCaptureStatus cap_status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap_status) != hipSuccess || cap_status != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return kNotFused;
    }
    if (MODE == MODE_NEAREST && p.idx && p.idx_bytes == 8 && (((uintptr_t)p.idx) & 15)) return kNotFused;
    if (MODE == MODE_QDQ && p.lev8 && (((uintptr_t)p.lev8) & 3)) return kNotFused;
    if (p.n > kFusedMaxN) return kNotFused;
    const int64_t n4 = p.n >> 2;
    const int64_t lanes = (n4 + 255) / 256;                    // blocks needed at one float4 per lane
#define QD_FUSED(V, W)                                                                                     \
    {                                                                                                      \
        const int cap = fused_capacity<MODE, V, W>();                                                      \
        const int64_t blocks = (lanes + V - 1) / V;                                                        \
        if (cap > 0 && blocks <= cap) {                                                                    \
            uint64_t epoch;                                                                                \
            unsigned tag_min, tag_max;                                                                     \
            do {                                            /* both tags non-zero: 0 | 0 is a never-written slot */ \
                epoch = next_launch.fetch_add(1, std::memory_order_relaxed);                               \
                tag_min = (unsigned)epoch;                                                                 \
                tag_max = tag_min ^ ((unsigned)(epoch >> 32) * 0x9E3779B9u);                               \
            } while (tag_min == 0u || tag_max == 0u);                                                      \
            const int slot = (int)(epoch % kFusedSlots);                                                   \
            p.nvec = 0;                                                                                    \
            hipLaunchKernelGGL((k_single_fused<MODE, V, W>), dim3((unsigned)(blocks < 1 ? 1 : blocks)), dim3(256), \
                               MODE == MODE_NEAREST ? point_table_bytes(p.k, p.fine) : 0, st, p, \
                               slot, tag_min, tag_max, fmode >= 2 ? fmode - 1 : 0);          \
            return check_launch();                                                                         \
        }                                                                                                  \
    }
    QD_FUSED(1, 4) QD_FUSED(4, 4)
#undef QD_FUSED
    return kNotFused;
}

// single bucket spanning a large tensor: one fused launch when it fits on chip, else reduce -> finalize -> apply
template <int MODE>
int launch_single(KParams& p, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int64_t kSmall = 16384;
    if (p.n <= kSmall) {                               // one block does both passes, one launch
        p.nvec = 0;
        hipLaunchKernelGGL((k_bucket_generic<MODE>), dim3(1), dim3(1024), MODE == MODE_NEAREST ? point_table_bytes(p.k, p.fine) : 0, st, p);
        return check_launch();
    }
    Workspace w;
    if (!carve(ws, ws_bytes, w)) return QD_ERR_WORKSPACE_TOO_SMALL;
    if (!(MODE == MODE_NEAREST && p.prescaled)) {
        const int rc = launch_single_fused<MODE>(p, st);
        if (rc != kNotFused) return rc;
    }
    const float* ab = nullptr;
    int nparts = 0;
    if (MODE == MODE_NEAREST && p.prescaled) {
        // alpha/beta are inputs: copy the pair into the scratch slot the apply kernel reads
        (void)hipMemcpyAsync(w.ab, p.alpha, sizeof(float), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(w.ab + 1, p.beta, sizeof(float), hipMemcpyDeviceToDevice, st);
        ab = w.ab;
    } else {
        int pb = blocks_for(p.n, 256 * 4 * 8);
        if (pb > kPartialBlocks) pb = kPartialBlocks;
        hipLaunchKernelGGL(k_minmax_partial, dim3(pb), dim3(256), 0, st, p.x, p.n, p.mean, p.me, w.minmax_part);
        nparts = pb;
        if (p.n > ((int64_t)8 << 20)) {      // > 32 MB: thousands of apply blocks, fold once in its own launch
            hipLaunchKernelGGL(k_minmax_final, dim3(1), dim3(256), 0, st, w.minmax_part, pb, w.ab, p.alpha, p.beta);
            ab = w.ab;
        }
    }
    const int blocks = blocks_for(p.n, 256 * 4 * 4);
    hipLaunchKernelGGL((k_single_apply<MODE>), dim3(blocks < ((MODE == MODE_NEAREST && p.fine) ? kFineBlocksPerCu * num_cus() : (1 << 30)) ? blocks : kFineBlocksPerCu * num_cus()), dim3(256),
                       MODE == MODE_NEAREST ? point_table_bytes(p.k, p.fine) : 0, st, p, ab,
                       w.minmax_part, nparts);
    return check_launch();
}

template <int MODE>
int run_transform(KParams& p, int64_t bucket, void* ws, size_t ws_bytes, hipStream_t st) {
    if (p.n == 0) return 0;
    geometry(p.n, bucket, p.nb, p.row);
    if (p.nb == 1) return launch_single<MODE>(p, ws, ws_bytes, st);
    return launch_bucketed<MODE>(p, st);
}

}  // namespace
