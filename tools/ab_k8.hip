// A/B of the 'truncated' STE gradient mask (K8, grad[|w| > limit] = 0) on 64 Mi elements with 32 % of |w| > 1:
//   v0  16-byte read-modify-write of g where any of four lanes is masked (round 5)
//   v1  4-byte zero stores under the lane mask, g never read
//   v2  16-byte zero store where all four are masked, 4-byte zero stores otherwise, g never read
//   v3  as v1 with non-temporal stores
// hipcc --offload-arch=gfx950 -O3 tools/ab_k8.hip -o build/ab_k8 && build/ab_k8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int V>
__global__ __launch_bounds__(256) void k(const float* w, float* grad, int64_t n, float limit) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nth = (int64_t)gridDim.x * blockDim.x;
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += nth) {
        const f4 v = __builtin_nontemporal_load((const f4*)w + i);
        const bool m0 = fabsf(v.x) > limit, m1 = fabsf(v.y) > limit, m2 = fabsf(v.z) > limit, m3 = fabsf(v.w) > limit;
        if (V == 0) {
            if (m0 | m1 | m2 | m3) {
                f4 gv = ((const f4*)grad)[i];
                gv.x = m0 ? 0.0f : gv.x; gv.y = m1 ? 0.0f : gv.y; gv.z = m2 ? 0.0f : gv.z; gv.w = m3 ? 0.0f : gv.w;
                ((f4*)grad)[i] = gv;
            }
        } else if (V == 1) {
            float* g = grad + 4 * i;
            if (m0) g[0] = 0.0f;
            if (m1) g[1] = 0.0f;
            if (m2) g[2] = 0.0f;
            if (m3) g[3] = 0.0f;
        } else if (V == 2) {
            float* g = grad + 4 * i;
            if (m0 & m1 & m2 & m3) { f4 z = {0.f, 0.f, 0.f, 0.f}; ((f4*)grad)[i] = z; }
            else {
                if (m0) g[0] = 0.0f;
                if (m1) g[1] = 0.0f;
                if (m2) g[2] = 0.0f;
                if (m3) g[3] = 0.0f;
            }
        } else {
            float* g = grad + 4 * i;
            if (m0) __builtin_nontemporal_store(0.0f, g + 0);
            if (m1) __builtin_nontemporal_store(0.0f, g + 1);
            if (m2) __builtin_nontemporal_store(0.0f, g + 2);
            if (m3) __builtin_nontemporal_store(0.0f, g + 3);
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int64_t n = 1 << 26;
    const int R = 4;
    float scale = argc > 1 ? atof(argv[1]) : 1.0f;      // w = scale * randn: 1.0 -> 32 % masked at limit 1
    std::vector<float> hw(n), hg(n);
    uint64_t s = 88172645463325252ull;
    auto u = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; };
    for (int64_t i = 0; i < n; i += 2) {
        double a = sqrt(-2.0 * log(u() + 1e-300)), b = 6.283185307179586 * u();
        hw[i] = scale * a * cos(b); hw[i + 1] = scale * a * sin(b);
    }
    for (int64_t i = 0; i < n; ++i) hg[i] = 1.0f + (i & 7);
    int64_t masked = 0;
    for (int64_t i = 0; i < n; ++i) masked += fabsf(hw[i]) > 1.0f;
    printf("n = %lld, masked %.1f %%\n", (long long)n, 100.0 * masked / n);
    float *w[R], *g[R];
    for (int r = 0; r < R; ++r) {
        CK(hipMalloc(&w[r], n * 4)); CK(hipMalloc(&g[r], n * 4));
        // rotate the host tensor so that the buffers differ
        CK(hipMemcpy(w[r], hw.data() + r * 1024, (n - r * 1024) * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(w[r] + (n - r * 1024), hw.data(), r * 1024 * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(g[r], hg.data(), n * 4, hipMemcpyHostToDevice));
    }
    const int blocks = (int)((n + 256 * 16 - 1) / (256 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto launch = [&](int v, int r) {
        switch (v) {
            case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, w[r], g[r], n, 1.0f); break;
            case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, w[r], g[r], n, 1.0f); break;
            case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, w[r], g[r], n, 1.0f); break;
            default: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, w[r], g[r], n, 1.0f); break;
        }
    };
    // correctness: every variant gives g = 0 where masked, the original elsewhere
    std::vector<float> out(n);
    for (int v = 0; v < 4; ++v) {
        CK(hipMemcpy(g[0], hg.data(), n * 4, hipMemcpyHostToDevice));
        launch(v, 0); CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), g[0], n * 4, hipMemcpyDeviceToHost));
        int64_t bad = 0;
        for (int64_t i = 0; i < n; ++i) bad += out[i] != (fabsf(hw[i]) > 1.0f ? 0.0f : hg[i]);
        printf("v%d mismatches %lld\n", v, (long long)bad);
    }
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 4; ++v) {
            for (int i = 0; i < 200; ++i) launch(v, i % R);          // preconditioning
            CK(hipDeviceSynchronize());
            std::vector<float> us;
            for (int t = 0; t < 5; ++t) {
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 40; ++i) launch(v, i % R);
                CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                us.push_back(ms * 1e3f / 40);
            }
            std::sort(us.begin(), us.end());
            printf("v%d  %.2f us (%.2f..%.2f)\n", v, us[2], us[0], us[4]);
        }
    return 0;
}
