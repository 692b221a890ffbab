// qd_kernels.hip -- hand-written gfx950 (CDNA4) kernels for the fake-quantization hot path of
// antspy/quantized_distillation, behind the C ABI of include/qd_hip.h.
//
// Reference being replaced (paths relative to the reference root):
//   quantization/quant_functions.py:56-152   ScalingFunction.scale_down / inv_scale_down
//   quantization/quant_functions.py:155-194  uniformQuantization
//   quantization/quant_functions.py:196-290  nonUniformQuantization (+ SearchSorted :509-573)
//   quantization/quant_functions.py:319-406  uniformQuantization_variable.backward
//   quantization/quant_functions.py:471-506  nonUniformQuantization_variable.backward
//   quantization/help_functions.py:67-94     create_bucket_tensor (semantics folded in)
//
// Design: this is HBM-bound elementwise + small-reduction work, so there is no
// MFMA and no LDS staging of data.  A bucket of 256 fp32 is 1 KiB; one DPP row (16 lanes) owns
// one bucket and holds it in registers as 4 x float4, so a wave64 streams 4 buckets (4 KiB) per
// iteration with 16-byte coalesced non-temporal loads, reduces min/max with 4 DPP row rotations
// (no LDS, no bpermute), and writes the result once: 8 B/element of HBM traffic instead of the
// reference's ~12 unfused passes.
//
// No CUDA compatibility layer, no dual code paths: gfx950 only.

#include "qd_transform.h"      // the MODE-templated kernels and launchers (shared with qd_scale.hip, qd_nearest.hip)


extern "C" int qd_fused_mode_state = 1;       // see qd_transform.h


// ================================ C ABI ========================================================
extern "C" {

int qd_abi_version(void) { return QD_ABI_VERSION; }

int qd_set_single_fused_mode(int mode) {
    const int prev = qd_fused_mode_state;
    qd_fused_mode_state = (mode >= 0 && mode <= 4) ? mode : 1;
    return prev;
}
const char* qd_target_arch(void) { return "gfx950"; }

const char* qd_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case QD_ERR_INVALID_ARGUMENT: return "qd: invalid argument";
        case QD_ERR_WORKSPACE_TOO_SMALL: return "qd: workspace missing, misaligned or smaller than qd_workspace_bytes()";
        case QD_ERR_UNSUPPORTED: return "qd: unsupported configuration";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "qd: unknown error";
    }
}

size_t qd_workspace_bytes(void) { return kWsBytes; }

int64_t qd_num_buckets(int64_t n, int64_t bucket) {
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    return nb;
}
int64_t qd_padded_length(int64_t n, int64_t bucket) {
    int64_t nb, row;
    geometry(n, bucket, nb, row);
    return nb * row;
}

int qd_uniform_f32(const float* x, float* q, int64_t n, int64_t bucket, int levels, float* alpha, float* beta,
                   uint8_t* level_idx, const float* mean, int clamp, float max_element, int stochastic,
                   uint64_t seed, void* workspace, size_t workspace_bytes, void* stream) {
    if (n < 0 || levels < 2 || bucket < 0 || (n > 0 && !x)) return QD_ERR_INVALID_ARGUMENT;
    if (level_idx && levels > 256) return QD_ERR_INVALID_ARGUMENT;
    if (n > 0 && !q) {
        // levels only: the integer level of every element (1 B written per element instead of 4 + 1) -- what the Huffman
        // accounting and the packed codec consume.  That is the fused pack kernel with 8 bits per level (qd_codec.hip), so
        // its geometry applies: deterministic rounding, no mean / clamp, bucket in {64 ... 2048}, x 16-byte aligned;
        // QD_ERR_UNSUPPORTED otherwise (callers then take the q-writing form).
        if (!level_idx) return QD_ERR_INVALID_ARGUMENT;
        if (stochastic || mean || clamp) return QD_ERR_UNSUPPORTED;
        return qd_pack_uniform_f32(x, n, bucket, levels, 8, level_idx, alpha, beta, stream);
    }
    KParams p = {};
    p.x = x; p.out = q; p.n = n; p.alpha = alpha; p.beta = beta; p.mean = mean;
    p.me = clamp ? max_element : INFINITY;
    p.sm1 = (float)(levels - 1);
    p.lev8 = level_idx;
    p.stochastic = stochastic; p.seed = seed;
    return run_transform<MODE_QDQ>(p, bucket, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
