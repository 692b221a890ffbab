// native_check.cpp -- uses libqd_hip.so through its C ABI from plain C++/HIP: no Python, no torch.
// Shows that the boundary really is "pointers, sizes, a stream" (include/qd_hip.h) and checks the
// results against the C oracle (oracle/qd_oracle.c, linked here as the checker only).
//   hipcc -O2 -I include tests/native/native_check.cpp -L quantized_distillation_amd -lqd_hip \
//         -L oracle/_build -lqd_oracle -Wl,-rpath,... -o build/native_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "qd_hip.h"

extern "C" {   // the checker (oracle/qd_oracle.c)
void qdo_uniform_f32(const float* x, float* q, int64_t n, int64_t bucket, int s, float* alpha, float* beta,
                     int64_t* imin, int64_t* imax, int32_t* lev, int sub_mean, float mean, int clamp, float me);
void qdo_scale_down_f32(const float* x, float* u, int64_t n, int64_t bucket, float* alpha, float* beta, int64_t* imin,
                        int64_t* imax, int sub_mean, float mean, int clamp, float me);
void qdo_nonuniform_f32(const float* x, const float* pts, int k, int mode, float* q, int64_t* idx, int64_t n,
                        int64_t bucket, float* alpha, float* beta);
void qdo_point_grad_f32(const float* g, const int64_t* idx, const float* alpha, int64_t n, int64_t bucket, int k,
                        double* out, double* abs_out);
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define QD(x) do { int rc_ = (x); if (rc_ != 0) { std::printf("qd error %d (%s) line %d\n", rc_, qd_error_string(rc_), __LINE__); return 3; } } while (0)

static int failures = 0;
static void expect(bool ok, const char* what) { std::printf("%-58s %s\n", what, ok ? "ok" : "MISMATCH"); if (!ok) ++failures; }

template <class T> static T* dev_copy(const std::vector<T>& h) {
    T* d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
template <class T> static std::vector<T> host_copy(const T* d, size_t n) {
    std::vector<T> h(n);
    (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

int main() {
    std::printf("libqd_hip ABI %d for %s, workspace %zu bytes\n", qd_abi_version(), qd_target_arch(), qd_workspace_bytes());
    const int64_t n = 1000003;                      // ragged
    const int64_t bucket = 256;
    std::vector<float> x(n), g(n);
    uint32_t s = 2463534242u;
    for (int64_t i = 0; i < n; ++i) {               // xorshift -> roughly normal by summing 4 uniforms
        float acc = 0.f;
        for (int r = 0; r < 4; ++r) { s ^= s << 13; s ^= s >> 17; s ^= s << 5; acc += (s >> 8) * (1.0f / 16777216.0f); }
        x[i] = (acc - 2.0f) * 1.7f;
        s ^= s << 13; s ^= s >> 17; s ^= s << 5;
        g[i] = ((s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 0.01f;
    }
    hipStream_t st; CK(hipStreamCreate(&st));
    void* ws = nullptr; CK(hipMalloc(&ws, qd_workspace_bytes()));
    const int64_t nb = qd_num_buckets(n, bucket);
    float *dx = dev_copy(x), *dg = dev_copy(g), *dq, *da, *db, *du;
    CK(hipMalloc(&dq, n * 4)); CK(hipMalloc(&da, nb * 4)); CK(hipMalloc(&db, nb * 4));
    CK(hipMalloc(&du, qd_padded_length(n, bucket) * 4));

    // K1: bucketed 4-bit quantize-dequantize
    QD(qd_uniform_f32(dx, dq, n, bucket, 16, da, db, nullptr, nullptr, 0, 0.f, 0, 0, ws, qd_workspace_bytes(), st));
    CK(hipStreamSynchronize(st));
    std::vector<float> rq(n), ra(nb), rb(nb);
    qdo_uniform_f32(x.data(), rq.data(), n, bucket, 16, ra.data(), rb.data(), nullptr, nullptr, nullptr, 0, 0.f, 0, 0.f);
    expect(host_copy(dq, n) == rq && host_copy(da, nb) == ra && host_copy(db, nb) == rb, "K1  qd_uniform_f32 bucket 256, bit exact");

    // K1g: whole tensor = one bucket (reduce -> fold -> apply, no host sync)
    QD(qd_uniform_f32(dx, dq, n, 0, 16, da, db, nullptr, nullptr, 0, 0.f, 0, 0, ws, qd_workspace_bytes(), st));
    CK(hipStreamSynchronize(st));
    qdo_uniform_f32(x.data(), rq.data(), n, 0, 16, ra.data(), rb.data(), nullptr, nullptr, nullptr, 0, 0.f, 0, 0.f);
    expect(host_copy(dq, n) == rq && host_copy(da, 1)[0] == ra[0], "K1g qd_uniform_f32 no buckets, bit exact");

    // K2 + K5 + K6: scale once, assign to 4 points by the midpoint rule, reduce the gradient
    QD(qd_scale_down_f32(dx, du, n, bucket, da, db, nullptr, 0, 0.f, ws, qd_workspace_bytes(), st));
    const std::vector<float> pts = {0.0f, 0.41f, 0.59f, 1.0f};
    float* dp = dev_copy(pts);
    uint8_t* didx; CK(hipMalloc(&didx, n));
    QD(qd_nearest_point_f32(du, 1, dp, 4, QD_ASSIGN_MIDPOINT, dq, didx, 1, n, bucket, da, db, nullptr, 0, 0.f, ws,
                            qd_workspace_bytes(), st));
    float* dgp; CK(hipMalloc(&dgp, 4 * 4));
    QD(qd_point_grad_f32(dg, didx, 1, da, n, bucket, 4, dgp, ws, qd_workspace_bytes(), st));
    CK(hipStreamSynchronize(st));
    std::vector<int64_t> ridx(n);
    qdo_nonuniform_f32(x.data(), pts.data(), 4, 1, rq.data(), ridx.data(), n, bucket, ra.data(), rb.data());
    std::vector<uint8_t> hidx = host_copy(didx, n);
    bool idx_ok = true;
    for (int64_t i = 0; i < n; ++i) idx_ok &= (hidx[i] == (uint8_t)ridx[i]);
    expect(idx_ok && host_copy(dq, n) == rq, "K5  qd_nearest_point_f32 prescaled/midpoint, bit exact");
    double want[4], absum[4];
    qdo_point_grad_f32(g.data(), ridx.data(), ra.data(), n, bucket, 4, want, absum);
    std::vector<float> gp = host_copy(dgp, 4);
    bool gp_ok = true;
    for (int j = 0; j < 4; ++j) gp_ok &= std::fabs((double)gp[j] - want[j]) <= 4e-6 * absum[j] + 1e-30;
    expect(gp_ok, "K6  qd_point_grad_f32 within 4e-6 * sum|g*alpha|");

    // argument errors come back as negative codes, not crashes
    expect(qd_uniform_f32(nullptr, dq, n, bucket, 16, da, db, nullptr, nullptr, 0, 0.f, 0, 0, ws, qd_workspace_bytes(), st) ==
               QD_ERR_INVALID_ARGUMENT, "null input -> QD_ERR_INVALID_ARGUMENT");
    expect(qd_uniform_f32(dx, dq, n, 0, 16, da, db, nullptr, nullptr, 0, 0.f, 0, 0, nullptr, 0, st) ==
               QD_ERR_WORKSPACE_TOO_SMALL, "missing workspace -> QD_ERR_WORKSPACE_TOO_SMALL");
    std::printf(failures ? "NATIVE CHECK FAILED (%d)\n" : "NATIVE CHECK PASSED\n", failures);
    return failures ? 1 : 0;
}
