// qd_scale.hip -- K2, ScalingFunction.scale_down (quantization/quant_functions.py:56-107): the MODE_SCALE instantiation of the
// bucket kernels of qd_transform.h, as its own translation unit (parallel build).
#include "qd_transform.h"

extern "C" {

int qd_scale_down_f32(const float* x, float* u, int64_t n, int64_t bucket, float* alpha, float* beta,
                      const float* mean, int clamp, float max_element, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (n < 0 || bucket < 0 || (n > 0 && (!x || !u || !alpha || !beta))) return QD_ERR_INVALID_ARGUMENT;
    KParams p = {};
    p.x = x; p.out = u; p.n = n; p.alpha = alpha; p.beta = beta; p.mean = mean;
    p.me = clamp ? max_element : INFINITY;
    return run_transform<MODE_SCALE>(p, bucket, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
