/* CPU oracle (plain C) for the fake-quantization hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * A scalar restatement of antspy/quantized_distillation's quantization/quant_functions.py and
 * quantization/help_functions.py (citations below are relative to /root/reference/), one
 * separately rounded IEEE fp32 operation per reference tensor op, in the reference's op order.
 * Build with -ffp-contract=off (see oracle/Makefile) so no multiply-add is fused.
 *
 * It exists so that parity can be checked at the full benchmark sizes (64 Mi elements) in
 * seconds, and so that bench.py can time a CPU baseline of the same algorithm on the GPU box's
 * host cores (OpenMP over buckets; the thread count used is reported).  Nothing in the product
 * links or loads this file.  It is pinned (tests/test_oracle_c.py) against oracle/oracle_np.py,
 * which in turn is pinned against vectors produced by running the reference
 * (tests/golden/gen_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QDO_TOL 1e-10f /* quant_functions.py:40 */

int qdo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void qdo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* help_functions.py:67-94: rows of the bucket view.  bucket <= 0 means None. */
static void geometry(int64_t n, int64_t bucket, int64_t* nb, int64_t* row) {
    if (bucket <= 0 || n < bucket) { *nb = 1; *row = n; return; }
    *row = bucket;
    *nb = (n + bucket - 1) / bucket;
}

int64_t qdo_num_buckets(int64_t n, int64_t bucket) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    return nb;
}

/* quant_functions.py:66-74: v = clamp(x - mean) (mean subtraction first, then clamp). */
static inline float prep(float x, int sub_mean, float mean, int clamp, float me) {
    if (sub_mean) x = x - mean;
    if (clamp) { if (x > me) x = me; if (x < -me) x = -me; }
    return x;
}

/* One bucket: min/max with first-occurrence indices (torch.min/max(dim) semantics),
 * alpha/beta with the 1e-10 guard.  quant_functions.py:85-99. */
static void bucket_stats(const float* x, int64_t lo, int64_t hi, int sub_mean, float mean, int clamp,
                         float me, float* alpha, float* beta, int64_t* imin, int64_t* imax) {
    float mn = prep(x[lo], sub_mean, mean, clamp, me), mx = mn;
    int64_t jmn = 0, jmx = 0;
    int has_nan = mn != mn;
    for (int64_t i = lo + 1; i < hi; ++i) {
        float v = prep(x[i], sub_mean, mean, clamp, me);
        if (v != v) has_nan = 1;
        if (v < mn) { mn = v; jmn = i - lo; }
        if (v > mx) { mx = v; jmx = i - lo; }
    }
    if (has_nan) { mn = NAN; mx = NAN; }   /* torch.min/max propagate NaN: the whole bucket becomes NaN */
    float a = mx - mn;
    if (a < QDO_TOL) a = 1.0f;
    *alpha = a; *beta = mn;
    if (imin) *imin = jmn;
    if (imax) *imax = jmx;
}

/* Global (bucket None) statistics with an OpenMP reduction that keeps first-occurrence ties. */
static void global_stats(const float* x, int64_t n, int sub_mean, float mean, int clamp, float me,
                         float* alpha, float* beta, int64_t* imin, int64_t* imax) {
    float gmn = INFINITY, gmx = -INFINITY;
    int64_t gjmn = 0, gjmx = 0;
    int any_nan = 0;
#pragma omp parallel
    {
        float mn = INFINITY, mx = -INFINITY;
        int64_t jmn = INT64_MAX, jmx = INT64_MAX;
#pragma omp for schedule(static) nowait
        for (int64_t i = 0; i < n; ++i) {
            float v = prep(x[i], sub_mean, mean, clamp, me);
            if (v != v) {
#pragma omp atomic write
                any_nan = 1;
            }
            if (v < mn) { mn = v; jmn = i; }
            if (v > mx) { mx = v; jmx = i; }
        }
#pragma omp critical
        {
            if (jmn != INT64_MAX && (mn < gmn || (mn == gmn && jmn < gjmn))) { gmn = mn; gjmn = jmn; }
            if (jmx != INT64_MAX && (mx > gmx || (mx == gmx && jmx < gjmx))) { gmx = mx; gjmx = jmx; }
        }
    }
    if (any_nan) { gmn = NAN; gmx = NAN; }
    float a = gmx - gmn;
    if (a < QDO_TOL) a = 1.0f;
    *alpha = a; *beta = gmn;
    if (imin) *imin = gjmn;
    if (imax) *imax = gjmx;
}

/* fp32 mean the way the oracle defines it: float64 accumulation, rounded once. */
float qdo_mean_f32(const float* x, int64_t n) {
    double acc = 0.0;
#pragma omp parallel for reduction(+ : acc) schedule(static)
    for (int64_t i = 0; i < n; ++i) acc += (double)x[i];
    return (float)(acc / (double)n);
}

/* uniformQuantization, linear scaling, deterministic rounding.  quant_functions.py:155-194.
 * q, lev: n elements (unpadded).  alpha/beta/imin/imax: one per bucket (may be NULL). */
void qdo_uniform_f32(const float* x, float* q, int64_t n, int64_t bucket, int s, float* alpha, float* beta,
                     int64_t* imin, int64_t* imax, int32_t* lev, int sub_mean, float mean, int clamp, float me) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    const float sm1 = (float)(s - 1); /* :172 */
    if (nb == 1) {
        float a, b;
        global_stats(x, n, sub_mean, mean, clamp, me, &a, &b, imin, imax);
        if (alpha) alpha[0] = a;
        if (beta) beta[0] = b;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            float v = prep(x[i], sub_mean, mean, clamp, me);
            float u = v - b;  u = u / a;            /* :106-107 */
            float t = u * sm1;                      /* :189 */
            float r = rintf(t);                     /* :190 half-to-even */
            float w = r / sm1;                      /* :191 */
            float y = w * a;  y = y + b;            /* :142-143 */
            if (sub_mean) y = y + mean;             /* :148 (mean 0 is an exact no-op) */
            q[i] = y;
            if (lev) lev[i] = (int32_t)r;
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int64_t b_ = 0; b_ < nb; ++b_) {
        int64_t lo = b_ * row, hi = lo + row < n ? lo + row : n;
        float a, b;
        bucket_stats(x, lo, hi, sub_mean, mean, clamp, me, &a, &b, imin ? imin + b_ : 0, imax ? imax + b_ : 0);
        if (alpha) alpha[b_] = a;
        if (beta) beta[b_] = b;
        for (int64_t i = lo; i < hi; ++i) {
            float v = prep(x[i], sub_mean, mean, clamp, me);
            float u = v - b;  u = u / a;
            float t = u * sm1;
            float r = rintf(t);
            float w = r / sm1;
            float y = w * a;  y = y + b;
            if (sub_mean) y = y + mean;
            q[i] = y;
            if (lev) lev[i] = (int32_t)r;
        }
    }
}

/* ScalingFunction.scale_down alone, unpadded output u[n].  quant_functions.py:56-107. */
void qdo_scale_down_f32(const float* x, float* u, int64_t n, int64_t bucket, float* alpha, float* beta,
                        int64_t* imin, int64_t* imax, int sub_mean, float mean, int clamp, float me) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    if (nb == 1) {
        float a, b;
        global_stats(x, n, sub_mean, mean, clamp, me, &a, &b, imin, imax);
        if (alpha) alpha[0] = a;
        if (beta) beta[0] = b;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            float v = prep(x[i], sub_mean, mean, clamp, me) - b;
            u[i] = v / a;
        }
        return;
    }
#pragma omp parallel for schedule(static)
    for (int64_t b_ = 0; b_ < nb; ++b_) {
        int64_t lo = b_ * row, hi = lo + row < n ? lo + row : n;
        float a, b;
        bucket_stats(x, lo, hi, sub_mean, mean, clamp, me, &a, &b, imin ? imin + b_ : 0, imax ? imax + b_ : 0);
        if (alpha) alpha[b_] = a;
        if (beta) beta[b_] = b;
        for (int64_t i = lo; i < hi; ++i) {
            float v = prep(x[i], sub_mean, mean, clamp, me) - b;
            u[i] = v / a;
        }
    }
}

/* nearest point, distance rule: searchsorted-left, clip, step down when STRICTLY closer to the
 * lower point.  quant_functions.py:267-273. */
static inline int assign_distance(float u, const float* p, int k) {
    int lo = 0, hi = k;               /* first i with p[i] >= u */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (p[mid] < u) lo = mid + 1; else hi = mid; }
    int i = lo > k - 1 ? k - 1 : lo;
    if (i > 0 && fabsf(u - p[i - 1]) < fabsf(u - p[i])) i -= 1;
    return i;
}

/* midpoint rule: #{m_j <= u}, m_j = p_j + (p_{j+1}-p_j)/2 in fp32.  quant_functions.py:531-573. */
static inline int assign_midpoint(float u, const float* m, int km1) {
    int lo = 0, hi = km1;             /* first j with m[j] > u */
    while (lo < hi) { int mid = (lo + hi) >> 1; if (m[mid] <= u) lo = mid + 1; else hi = mid; }
    return lo;
}

/* nonUniformQuantization (linear scaling).  mode 0 = distance rule, 1 = midpoint rule.
 * quant_functions.py:196-290.  idx: int64[n]. */
void qdo_nonuniform_f32(const float* x, const float* pts, int k, int mode, float* q, int64_t* idx, int64_t n,
                        int64_t bucket, float* alpha, float* beta) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    float* mids = (float*)malloc(sizeof(float) * (size_t)(k > 1 ? k - 1 : 1));
    for (int j = 0; j + 1 < k; ++j) { float d = pts[j + 1] - pts[j]; d = d / 2.0f; mids[j] = pts[j] + d; }
    float ga = 1.0f, gb = 0.0f;
    if (nb == 1) {
        global_stats(x, n, 0, 0.0f, 0, 0.0f, &ga, &gb, 0, 0);
        if (alpha) alpha[0] = ga;
        if (beta) beta[0] = gb;
    }
#pragma omp parallel for schedule(static)
    for (int64_t b_ = 0; b_ < nb; ++b_) {
        int64_t lo = b_ * row, hi = lo + row < n ? lo + row : n;
        float a = ga, b = gb;
        if (nb > 1) {
            bucket_stats(x, lo, hi, 0, 0.0f, 0, 0.0f, &a, &b, 0, 0);
            if (alpha) alpha[b_] = a;
            if (beta) beta[b_] = b;
        }
        for (int64_t i = lo; i < hi; ++i) {
            float u = x[i] - b;  u = u / a;
            int j = mode == 0 ? assign_distance(u, pts, k) : assign_midpoint(u, mids, k - 1);
            float y = pts[j] * a;  y = y + b;
            q[i] = y;
            idx[i] = j;
        }
    }
    free(mids);
}

/* gradPoint[j] = sum_{idx_i == j} g_i * alpha_bucket(i); float64 accumulation of the fp32
 * products.  quant_functions.py:493-503.  abs_out (optional) = sum |g_i*alpha| per bin. */
void qdo_point_grad_f32(const float* g, const int64_t* idx, const float* alpha, int64_t n, int64_t bucket, int k,
                        double* out, double* abs_out) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    for (int j = 0; j < k; ++j) { out[j] = 0.0; if (abs_out) abs_out[j] = 0.0; }
    for (int64_t i = 0; i < n; ++i) {
        float a = alpha[nb == 1 ? 0 : i / row];
        float m = g[i] * a;
        out[idx[i]] += (double)m;
        if (abs_out) abs_out[idx[i]] += fabs((double)m);
    }
}

/* 'complicated' STE backward, reference-faithful tie rule (first element at the top / bottom
 * level of the QUANTIZED bucket).  quant_functions.py:319-406 + the shape fixes of SURVEY 8c.
 * Accumulates the bucket sum in float64. */
void qdo_ste_backward_f32(const float* x, const float* g, float* out, int64_t n, int64_t bucket, int s) {
    int64_t nb, row;
    geometry(n, bucket, &nb, &row);
    float* q = (float*)malloc(sizeof(float) * (size_t)n);
    qdo_uniform_f32(x, q, n, bucket, s, 0, 0, 0, 0, 0, 0, 0.0f, 0, 0.0f);
#pragma omp parallel for schedule(static)
    for (int64_t b_ = 0; b_ < nb; ++b_) {
        int64_t lo = b_ * row, hi = lo + row < n ? lo + row : n;
        float a, b;
        int64_t jmn, jmx;
        bucket_stats(q, lo, hi, 0, 0.0f, 0, 0.0f, &a, &b, &jmn, &jmx);   /* :350 on the quantized tensor */
        double sb = 0.0;
        for (int64_t i = lo; i < hi; ++i) {
            float qs = q[i] - b;  qs = qs / a;
            float u = x[i] - b;   u = u / a;                             /* :400 */
            float d = qs - u;
            float t = g[i] * d;
            sb += (double)t;
            out[i] = g[i];
        }
        double vmax = (double)g[lo + jmx] + sb;
        if (jmx == jmn) vmax -= sb;
        out[lo + jmx] = (float)vmax;
        if (jmx != jmn) out[lo + jmn] = (float)((double)g[lo + jmn] - sb);
    }
    free(q);
}

/* float64 checksums used by the full-size property tests. */
void qdo_checksum_f32(const float* x, int64_t n, double* sum, double* sumsq) {
    double a = 0.0, b = 0.0;
#pragma omp parallel for reduction(+ : a, b) schedule(static)
    for (int64_t i = 0; i < n; ++i) { a += (double)x[i]; b += (double)x[i] * (double)x[i]; }
    *sum = a; *sumsq = b;
}
