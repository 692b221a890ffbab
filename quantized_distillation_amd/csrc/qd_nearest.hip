// qd_nearest.hip -- K4 / K5, nearest-point (non-uniform) quantization (quantization/quant_functions.py:196-290, :531-573): the
// MODE_NEAREST instantiation of the bucket kernels of qd_transform.h, as its own translation unit (parallel build).
#include "qd_transform.h"

extern "C" {

int qd_nearest_point_f32(const float* x, int prescaled, const float* points, int k, int assign_mode, float* q,
                         void* idx, int idx_bytes, int64_t n, int64_t bucket, float* alpha, float* beta,
                         const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (n < 0 || bucket < 0 || k < 1 || k > kMaxPoints || !points || (n > 0 && (!x || !q)))
        return QD_ERR_INVALID_ARGUMENT;
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
    return run_transform<MODE_NEAREST>(p, bucket, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
