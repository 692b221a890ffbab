/* qd_hip.h -- C ABI of libqd_hip.so, the MI355X (gfx950) fake-quantization kernels.
 *
 * This is the drop-in boundary of the hot path.  The reference
 * (antspy/quantized_distillation) has no FFI layer: its hot path is the Python module
 * `quantization` (quantization/__init__.py:4-8), a chain of unfused torch ops.  Each entry point
 * below replaces the chain of reference ops cited next to it (paths relative to the reference
 * root); the Python package `quantization` shipped in this repository keeps the reference's
 * signatures and binds these symbols with ctypes and, for the per-call entry points, with a small
 * CPython/ATen module (csrc/qd_torch_glue.cpp).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it says "host";
 *   - tensors are contiguous fp32; `n` is the number of real elements;
 *   - `bucket`: elements per bucket, or 0 for "bucket_size=None" (one bucket = whole tensor);
 *     the bucket view follows quantization/help_functions.py:67-94:
 *       n <  bucket            -> 1 bucket of n elements
 *       n %  bucket == 0       -> n/bucket buckets
 *       otherwise              -> ceil(n/bucket) buckets, the last one padded with copies of
 *                                 x[n-1] (padding never changes min/max, so kernels ignore it);
 *   - all calls are asynchronous on `stream` (a hipStream_t passed as void*), never allocate,
 *     never synchronise with the host;
 *   - return value: 0 on success, a positive hipError_t if a launch failed, or a negative
 *     QD_ERR_* code for argument errors.  qd_error_string() describes either.
 *   - optional outputs may be NULL.
 *   - alignment: fp32 data pointers need the 4 bytes of their type and nothing more (a tensor view that starts 4, 8 or
 *     12 bytes into a 16-byte granule runs the same kernels at the same speed); int64 index outputs and workspaces 16
 *     bytes, uint8 level / index outputs 4 bytes, unless an entry point says otherwise.
 *   - `mean`: optional device scalar subtracted from every element before anything else
 *     (subtract_mean=True, quant_functions.py:66-68) and added back at the end (:148).
 *   - `clamp`/`max_element`: if clamp != 0, values are clamped to [-max_element, max_element]
 *     after the mean subtraction (quant_functions.py:72-74).
 */
#ifndef QD_HIP_H
#define QD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_ERR_INVALID_ARGUMENT (-1)
#define QD_ERR_WORKSPACE_TOO_SMALL (-2)
#define QD_ERR_UNSUPPORTED (-3)

/* nearest-point assignment rules (quant_functions.py:267-273 vs :531-573) */
#define QD_ASSIGN_DISTANCE 0
#define QD_ASSIGN_MIDPOINT 1

/* tie rule of the 'complicated' STE backward */
#define QD_STE_TIE_REFERENCE 0 /* first element at the top/bottom LEVEL of the quantized bucket */
#define QD_STE_TIE_TRUE_ARG 1  /* true argmax/argmin of the input                               */

/* Library identification: ABI version (bumped on every change of an entry point's meaning or signature) and target arch.
 * History: 1 = rounds 1-3; 2 = qd_nearest_point_f32 accepts q == NULL (indices only), qd_uniform_f32 accepts q == NULL with
 * level_idx (levels only), qd_selftest_div_invariant, qd_digitize_histogram_f32 / qd_histogram_i64 / qd_level_histogram_f32
 * added, 4-byte data alignment; 3 = qd_scale_digitize_histogram_f32 added, QdDiffQuantDesc.first_row (partial rows of
 * qd_multi_point_grad_f32 per tensor size instead of 4 B + 1 for every tensor).
 * The Python binding and _qd_glue.so compare the version THEY were built for with the library's. */
#define QD_ABI_VERSION 3
int qd_abi_version(void);
const char* qd_target_arch(void);
const char* qd_error_string(int code);

/* Bytes of device scratch the reductions need (global min/max, mean, point gradient).
 * One buffer of this size per stream is enough; contents need no initialisation. */
size_t qd_workspace_bytes(void);

/* bucket_size=None tensors (one bucket = whole tensor) of 16 Ki .. 1 Mi elements are processed by ONE launch -- load
 * once, grid-wide barrier on the per-block min/max, transform from registers (8 B/element) -- instead of reduce + fold +
 * apply.  The barrier never blocks: under contention a block gives up after 2 ms, folds the min/max of the whole tensor
 * itself and carries on; it depends on no other block (no departure counters, no "last block").
 *   mode 1 (default) = on, 0 = always the three-launch path;
 *   test hooks that reach the give-up path on an idle GPU: 2 = every block gives up at once, 3 = the blocks with
 *   blockIdx % 7 == 3, 4 = exactly one block (the middle one).
 * Returns the previous mode; any other value restores the default.  The only switch of this path (no environment
 * variable).  Process-wide, not thread-safe against concurrent launches. */
int qd_set_single_fused_mode(int mode);

/* Number of buckets / padded length of the bucket view (help_functions.py:67-94). Host only. */
int64_t qd_num_buckets(int64_t n, int64_t bucket);
int64_t qd_padded_length(int64_t n, int64_t bucket);

/* mean_out[0] = mean(x) (float64 accumulation, one rounding).  Replaces tensor.mean(),
 * quant_functions.py:67. */
int qd_mean_f32(const float* x, int64_t n, float* mean_out, void* workspace, size_t workspace_bytes, void* stream);

/* K1/K1g: uniformQuantization, linear scaling (quant_functions.py:155-194 with
 * ScalingFunction.scale_down :56-107 and inv_scale_down :131-152 fused):
 *     alpha_b = max_b - min_b (alpha < 1e-10 -> 1), beta_b = min_b
 *     q = rint((x - beta)/alpha * (levels-1)) / (levels-1) * alpha + beta   [each op rounded]
 * x, q: [n] (q may alias x: modify_in_place).  alpha, beta: [num_buckets] optional outputs.
 * level_idx: optional [n] uint8 output of the integer level rint(u*(levels-1)) (levels <= 256).
 * Levels only: q == NULL with level_idx given writes the levels (and alpha / beta if asked for) and nothing else -- 4 B read +
 * 1 B written per element; deterministic rounding without mean / clamp, bucket in {64, 128, 256, 512, 1024, 2048}, x 16-byte
 * aligned (the geometry of qd_pack_uniform_f32, whose 8-bit form this is); QD_ERR_UNSUPPORTED otherwise.
 * stochastic != 0 selects the stochastic-rounding branch (:174-187) with a counter-based
 * in-kernel generator keyed by (seed, element index).
 * workspace is only used when the tensor is a single bucket (bucket == 0 or n < bucket). */
int qd_uniform_f32(const float* x, float* q, int64_t n, int64_t bucket, int levels, float* alpha, float* beta,
                   uint8_t* level_idx, const float* mean, int clamp, float max_element, int stochastic,
                   uint64_t seed, void* workspace, size_t workspace_bytes, void* stream);

/* K2: ScalingFunction.scale_down, linear (quant_functions.py:56-107).  u has the PADDED bucket
 * layout: qd_padded_length(n, bucket) elements; padding entries hold the scaled last element,
 * exactly as the reference's cat-then-scale produces.  u may alias x only when no padding is
 * needed.  alpha, beta: [num_buckets] (required: they are what inverts the scaling). */
int qd_scale_down_f32(const float* x, float* u, int64_t n, int64_t bucket, float* alpha, float* beta,
                      const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                      void* stream);

/* K3: ScalingFunction.inv_scale_down, linear (quant_functions.py:131-152): y = u*alpha + beta
 * (+ mean), padding dropped.  u: padded layout; y: [n]; y may alias u. */
int qd_inv_scale_f32(const float* u, float* y, int64_t n, int64_t bucket, const float* alpha, const float* beta,
                     const float* mean, void* stream);

/* First-occurrence arg-min / arg-max of each bucket, relative to the bucket start (int64), the
 * idx_min_rows / idx_max_rows of ScalingFunction (quant_functions.py:85-90,103-104).  Computed
 * on demand only (nothing on the per-step path reads them except the 'complicated' STE). */
int qd_bucket_argminmax_f32(const float* x, int64_t n, int64_t bucket, const float* mean, int clamp,
                            float max_element, int64_t* argmin, int64_t* argmax, void* workspace,
                            size_t workspace_bytes, void* stream);

/* K4/K5: nearest-point (non-uniform) quantization (quant_functions.py:196-290).
 *   prescaled == 0: x is the raw tensor [n]; it is scaled per bucket first (alpha/beta are
 *                   OUTPUTS, [num_buckets]);
 *   prescaled != 0: x is u in the unpadded layout [n], already scaled; alpha/beta are INPUTS
 *                   (the per-step path of differentiable quantization: the tensor is frozen,
 *                   only `points` change, quant_functions.py:449-469).
 * points: [k] sorted ascending, device.  assign_mode: QD_ASSIGN_DISTANCE (searchsorted-left +
 * strictly-closer-lower rule, :267-273) or QD_ASSIGN_MIDPOINT (#{midpoints <= u}, the
 * SearchSorted.query formulation, :531-573).
 * q: [n] = points[idx]*alpha + beta (+mean).  idx: optional, idx_bytes 8 (int64, what the
 * reference API returns) or 1 (uint8, k <= 256).
 * Indices only (what SearchSorted.query returns, :531-563): prescaled != 0 with q == NULL and idx given -- n >= 4,
 * buckets of at least 4 elements (or bucket == 0), idx 16-byte (int64) / 4-byte (uint8) aligned; QD_ERR_INVALID_ARGUMENT
 * otherwise. */
int qd_nearest_point_f32(const float* x, int prescaled, const float* points, int k, int assign_mode, float* q,
                         void* idx, int idx_bytes, int64_t n, int64_t bucket, float* alpha, float* beta,
                         const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                         void* stream);

/* K6: nonUniformQuantization_variable.backward (quant_functions.py:471-506):
 *     grad_points[j] = sum_{i : idx_i == j} g_i * alpha_bucket(i)
 * Deterministic two-stage reduction (fixed order, no atomics).  idx_bytes 8 or 1. */
int qd_point_grad_f32(const float* g, const void* idx, int idx_bytes, const float* alpha, int64_t n, int64_t bucket,
                      int k, float* grad_points, void* workspace, size_t workspace_bytes, void* stream);

/* K7: 'complicated' straight-through backward of uniformQuantization_variable
 * (quant_functions.py:319-406, intended math): per bucket
 *     S_b = sum_i g_i*(qs_i - u_i);  out = g;  out[jmax_b] += S_b;  out[jmin_b] -= S_b.
 * Requires bucket > 0 (as the reference does, :332-334).  out may alias g. */
int qd_ste_bucket_backward_f32(const float* x, const float* g, float* out, int64_t n, int64_t bucket, int levels,
                               int tie_mode, void* stream);

/* K8: 'truncated' STE (cnn_models/conv_forward_model.py:240-241,263-264):
 * qd_clamp_f32: w = clamp(w, -limit, limit) in place; qd_truncated_ste_f32: grad[|w| > limit] = 0. */
int qd_clamp_f32(float* w, int64_t n, float limit, void* stream);
int qd_truncated_ste_f32(const float* w, float* grad, int64_t n, float limit, void* stream);

/* Multi-tensor K1: one launch quantizes every parameter tensor of a model (the per-step loop
 * `for p in model.parameters(): p.data = uniformQuantization(p.data, s, bucket_size=...)[0]`,
 * cnn_models/conv_forward_model.py:235-247).  `table` is a DEVICE array of descriptors built
 * once per model; each tensor is bucketed independently with the same `bucket`/`levels`.
 * bucket must be > 0. */
typedef struct QdTensorDesc {
    const float* x;      /* master weights            */
    float* q;            /* quantized shadow (may alias x) */
    int64_t n;           /* elements                   */
    int64_t first_tile;  /* prefix sum of work tiles (filled by qd_multi_plan) */
} QdTensorDesc;

/* Host helper: fills first_tile of `ntensors` host descriptors, returns the total tile count
 * (pass it to qd_multi_uniform_f32 as total_tiles). */
int64_t qd_multi_plan(QdTensorDesc* host_table, int ntensors, int64_t bucket);
int qd_multi_uniform_f32(const QdTensorDesc* table, int ntensors, int64_t total_tiles, int64_t bucket, int levels,
                         void* stream);

/* Multi-tensor K1g: the same per-step loop with bucket_size=None (every tensor one bucket with its
 * own global min/max; e.g. cifar10_test.py:113): three launches for the whole model (per-tile
 * min/max, per-tensor fold, apply) instead of three per tensor.  qd_multi_global_plan fills
 * first_tile (1024-element tiles); alpha_beta: [ntensors][2] output; workspace >= total_tiles*8 bytes. */
int64_t qd_multi_global_plan(QdTensorDesc* host_table, int ntensors);
int qd_multi_uniform_global_f32(const QdTensorDesc* table, int ntensors, int64_t total_tiles, int levels,
                                float* alpha_beta, void* workspace, size_t workspace_bytes, void* stream);

/* ---- 'absmax' / 'absnorm' scaling (type_scaling of ScalingFunction, quant_functions.py:109-127,144-146).
 * PARITY UNPINNED: the reference code for these two types raises on every torch version, so these
 * entry points implement the math those lines evidently intend (see qd_abs.hip):
 *     sign = sign(x), m = |x|, norm_b = max_b m (norm_kind 0) or sqrt(sum_b m^2) (norm_kind 1), < 1e-10 -> 1
 *     qd_scale_down_abs_f32 : u = m/norm_b and sign, both in the PADDED bucket layout; norm_out [num_buckets]
 *     qd_inv_scale_abs_f32  : y = u*norm_b*sign (+mean), padding dropped
 *     qd_uniform_abs_f32    : q = rint(u*(levels-1))/(levels-1)*norm_b*sign (+mean) */
int qd_uniform_abs_f32(const float* x, float* q, int64_t n, int64_t bucket, int levels, int norm_kind, float* norm_out,
                       const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                       void* stream);
int qd_scale_down_abs_f32(const float* x, float* u, float* sign, int64_t n, int64_t bucket, int norm_kind,
                          float* norm_out, const float* mean, int clamp, float max_element, void* workspace,
                          size_t workspace_bytes, void* stream);
int qd_inv_scale_abs_f32(const float* u, const float* sign, float* y, int64_t n, int64_t bucket, const float* norm,
                         const float* mean, void* stream);

/* ---- multi-tensor differentiable-quantization step: the per-tensor calls
 *     p_quantized.data = quantizationFunctions[i].forward(None, points[i].data)   (conv_forward_model.py:532)
 *     points[i].grad.data = quantizationFunctions[i].backward(p.grad.data)[1]     (conv_forward_model.py:545)
 * for ALL tensors in one launch each.  Arithmetic = qd_nearest_point_f32(prescaled, MIDPOINT) with
 * uint8 indices and qd_point_grad_f32.  k <= 64; bucket a power of two; `points` is one device
 * array [ntensors][k] (so the optimizer updates a single tensor); grad_points likewise. */
typedef struct QdDiffQuantDesc {
    const float* u;       /* scaled weights, resident [n]                                  */
    float* q;             /* quantized weights out [n]                                     */
    uint8_t* idx;         /* point index out (forward) / in (backward) [n]                 */
    const float* alpha;   /* [num_buckets]                                                 */
    const float* beta;    /* [num_buckets]                                                 */
    const float* grad;    /* dLoss/dq [n] (backward only)                                  */
    int64_t n;
    int64_t first_tile;   /* filled by qd_multi_dq_plan: prefix of 4-bucket tiles (forward) */
    int64_t first_block;  /* filled by qd_multi_dq_plan: prefix of FULL 1024-element gradient tiles (backward) */
    int64_t first_row;    /* filled by qd_multi_dq_plan: prefix of the backward sweep's partial rows (ABI 3)  */
} QdDiffQuantDesc;
/* Host helper: fills first_tile / first_block / first_row, returns the total of forward tiles, writes `total_blocks` = the
 * number of partial rows of the backward sweep: min(full 1024-element gradient tiles, 2048) + 1 per tensor (the 2048 waves of
 * the main grid stride over the ONE sequence of all tensors' full tiles and write one row per tensor they visit; one more
 * wave per tensor takes the n mod 1024 elements left).  Pass it to qd_multi_point_grad_f32 unchanged. */
int64_t qd_multi_dq_plan(QdDiffQuantDesc* host_table, int ntensors, int64_t bucket, int64_t* total_blocks_out);
int qd_multi_nearest_f32(const QdDiffQuantDesc* table, int ntensors, int64_t total_tiles, int64_t bucket,
                         const float* points, int k, void* stream);
/* workspace: at least total_blocks * k floats */
int qd_multi_point_grad_f32(const QdDiffQuantDesc* table, int ntensors, int64_t total_blocks, int64_t bucket, int k,
                            float* grad_points, void* workspace, size_t workspace_bytes, void* stream);

/* ---- packed-index codec (the compressed form whose SIZE the reference accounts for in
 * helpers/functions.py:226-262: bits*N/8 bytes of level indices + 8 bytes (alpha, beta) per
 * bucket) and the level histogram behind the Huffman accounting of
 * quantization/help_functions.py:175-232.
 * qd_pack_uniform_f32: quantize x with `levels` levels per bucket and store ONLY the level
 *   indices, `bits` (1, 2, 4 or 8; levels <= 2^bits) per element, element e in bits
 *   [e*bits, (e+1)*bits) of `packed` (qd_packed_bytes(n, bits) bytes, little endian inside a byte),
 *   plus alpha/beta [num_buckets] (optional).  bucket in {64,128,256,512,1024,2048}; x 16-byte aligned.
 * qd_pack_levels_u8: the same packing for level indices that are already there (the level_idx output of qd_uniform_f32):
 *   together they give the packed form at ANY bucket size, bucket 0 = none included.
 * qd_unpack_uniform_f32: y = (index/(levels-1))*alpha + beta -- bit-identical to the output of
 *   qd_uniform_f32 on the same input.  Any bucket size (0: one alpha / beta for the tensor).
 * qd_histogram_u8: hist[j] = #{i : idx[i] == j}, j < k <= 256 (hist is overwritten).
 * qd_histogram_u8_ws: the same with a scratch buffer (8-byte aligned, qd_workspace_bytes() is enough; contents need no
 *   initialisation): per-block totals go there and are summed per bin in a fixed order -- no global atomics. */
int64_t qd_packed_bytes(int64_t n, int bits);
int qd_pack_uniform_f32(const float* x, int64_t n, int64_t bucket, int levels, int bits, uint8_t* packed, float* alpha,
                        float* beta, void* stream);
int qd_pack_levels_u8(const uint8_t* levels_idx, int64_t n, int bits, uint8_t* packed, void* stream);
int qd_unpack_uniform_f32(const uint8_t* packed, int64_t n, int64_t bucket, int levels, int bits, const float* alpha,
                          const float* beta, float* y, void* stream);
int qd_histogram_u8(const uint8_t* idx, int64_t n, int k, uint64_t* hist, void* stream);
int qd_histogram_u8_ws(const uint8_t* idx, int64_t n, int k, uint64_t* hist, void* workspace, size_t workspace_bytes,
                       void* stream);
/* qd_level_histogram_f32: hist[j] = number of elements of x whose quantization level (levels per bucket, as qd_uniform_f32
 * computes it) is j, j < levels <= 256 -- in ONE pass, 4 B read per element: the level indices are counted in the kernel that
 * computes them and never written.  bucket in {64 ... 2048}, x 16-byte aligned (QD_ERR_UNSUPPORTED otherwise: callers then
 * write the levels with qd_uniform_f32 and count them with qd_histogram_u8_ws).  Elements of a bucket that holds a NaN count as
 * level 0 (what the uint8 level output stores for them).  workspace as for qd_histogram_u8_ws. */
int qd_level_histogram_f32(const float* x, int64_t n, int64_t bucket, int levels, uint64_t* hist, void* workspace,
                           size_t workspace_bytes, void* stream);
/* The two histograms get_huffman_encoding_mean_bit_length needs when it runs on the device (quantization/help_functions.py:
 * 175-232; only the counters cross PCIe instead of every quantized tensor):
 * qd_digitize_histogram_f32: hist[c] = #{ i : #{ j < m : edges[j] <= v[i] } == c }, c = 0 .. m (hist has m + 1 entries; a NaN
 *   counts as c = m) -- np.digitize(v, edges) of :216-218 followed by np.unique(..., return_counts=True) of :223, for
 *   `edges` a DEVICE array of m <= 256 increasing float64 values, compared in float64 as numpy does.  v: the re-scaled
 *   quantized tensor (`scal.scale_down(q_tensor)`, :215), n elements, 4-byte aligned.
 * qd_histogram_i64: hist[j] = #{ i : idx[i] == j } for j < k <= 256 and hist[k] = #{ i : idx[i] outside [0, k) } (k + 1
 *   entries) -- the counts of the int64 indices nonUniformQuantization returns (:220-221).
 * Both: workspace as for qd_histogram_u8_ws (required when n > 0), deterministic (per-block totals summed in a fixed order). */
int qd_digitize_histogram_f32(const float* v, int64_t n, const double* edges, int m, uint64_t* hist, void* workspace,
                              size_t workspace_bytes, void* stream);
int qd_histogram_i64(const int64_t* idx, int64_t n, int k, uint64_t* hist, void* workspace, size_t workspace_bytes,
                     void* stream);
/* qd_scale_digitize_histogram_f32: the re-scale and the digitize + count of :215-223 in ONE pass over the quantized tensor q
 * (4 B read per element; the two-call form -- qd_scale_down_f32, then qd_digitize_histogram_f32 over its output -- moves 12):
 * hist[c] = #{ i : #{ j < m : edges[j] <= (double)u[i] } == c } with u = ScalingFunction('linear', bucket_size=bucket)
 * .scale_down(q), i.e. per bucket (q - min) / (max - min or 1), bit-identical to qd_scale_down_f32's output.  For the plain
 * configuration only (no mean subtraction, no max_element), bucket in {64 ... 2048}, q 16-byte aligned: QD_ERR_UNSUPPORTED
 * otherwise, and the caller takes the two-call form.  edges, hist, workspace as for qd_digitize_histogram_f32. */
int qd_scale_digitize_histogram_f32(const float* q, int64_t n, int64_t bucket, const double* edges, int m, uint64_t* hist,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* ---- order statistics for initialize_quantization_points (quantization/help_functions.py:140-154: the reference
 * copies the scaled tensor to the host and calls np.percentile(a, linspace(0, 100, k)), which needs the two
 * neighbouring order statistics of each of the k virtual indices).
 * qd_order_stats_f32: out[t] (device, m floats) = the ranks[t]-th smallest element of x[0..n) (0-based; NaNs order
 *   last, -0 before +0), found by a 12|10|10-bit radix SELECT: x is read three times and never sorted or modified.
 *   ranks is a HOST array of m non-decreasing values in [0, n); 1 <= m <= QD_ORDER_STATS_MAX_RANKS; n < 2^32
 *   (QD_ERR_UNSUPPORTED beyond either).  workspace: qd_order_stats_workspace_bytes(m) bytes of device memory. */
#define QD_ORDER_STATS_MAX_RANKS 32
size_t qd_order_stats_workspace_bytes(int m);
int qd_order_stats_f32(const float* x, int64_t n, const int64_t* ranks, int m, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---- self test of the bucket-invariant division (csrc/qd_selftest.hip).  The quantize kernels replace the IEEE division
 * u = (x - beta) / alpha of quantization/quant_functions.py:106-107 by y = RN(1/alpha) once per bucket and two FMAs per
 * element (qd_common.h: div_alpha<true>), which must give the correctly rounded quotient bit for bit wherever it is used:
 * alpha in [2^-60, 2^100], n = 0 or n >= 2^-100 -- and, where the quotient itself is returned (qd_scale_down_f32), n >=
 * alpha 2^-120 as well (a normal quotient).  This entry point generates `npairs` adversarial (n, alpha) pairs of
 * `family` 0 .. 5 on the device (0: the quantizer's own domain, 1: wide exponents, 2: all-ones / near-power-of-two
 * significands, 3: near-exact quotients around level and half-level values, 4: the edges of the stated ranges, 5: small
 * normal quotients 2^-120 .. 2^-50), evaluates
 * the shortcut with the very function the kernels inline and compares it with n / alpha.
 * result (device, 4 x uint64): pairs tested, mismatches, (n bits << 32 | alpha bits) of one mismatch, pairs skipped
 * as outside the domain.  tools/div_invariant_check.py, tests/test_hip_parity.py. */
int qd_selftest_div_invariant(uint64_t seed, int64_t npairs, int family, unsigned long long* result, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QD_HIP_H */
