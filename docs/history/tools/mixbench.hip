// tools/mixbench.hip -- minimal streaming kernels: what can the chip do for a given read : write mix?  (round 4: the calls that
// return int64 indices -- 4 B read, 8 or 12 B written per element -- sit at 66-71 % of the HBM peak; so do these kernels, which do
// nothing else.  hipcc --offload-arch=gfx950 -O3 tools/mixbench.hip -o build/mixbench && build/mixbench -> profiles/r04_mix_ceiling.txt)
//  One 4 KiB input tile per wave (as the library's
// kernels), 16-byte accesses, non-temporal loads; stores non-temporal (NT = 1) or plain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef long long l2 __attribute__((ext_vector_type(2)));
template <int WQ, int WI, int NT>      // WQ: write a float4 of q per input float4; WI: write 32 B of int64 indices per input float4
__global__ __launch_bounds__(256) void k_mix(const f4* __restrict__ x, f4* __restrict__ q, l2* __restrict__ idx, long long n4) {
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long long base = wave * 256 + lane;            // 4 float4 per lane: base, +64, +128, +192
    if (base + 192 >= n4) return;
    f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = __builtin_nontemporal_load(x + base + 64 * j);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long ii = base + 64 * j;
        f4 r = v[j] * 1.0001f + 0.5f;
        if (WQ) { if (NT) __builtin_nontemporal_store(r, q + ii); else q[ii] = r; }
        if (WI) {
            // the wave's 64 float4 -> 128 chunks of 16 B, contiguous: lane writes chunk lane and chunk 64 + lane
            l2 a = {(long long)(int)r.x & 15, (long long)(int)r.y & 15}, b = {(long long)(int)r.z & 15, (long long)(int)r.w & 15};
            l2* o = idx + (ii - lane) * 2;
            if (NT) { __builtin_nontemporal_store(a, o + lane); __builtin_nontemporal_store(b, o + 64 + lane); }
            else { o[lane] = a; o[64 + lane] = b; }
        }
    }
}
template <int NT>
__global__ __launch_bounds__(256) void k_fill(l2* __restrict__ idx, long long n2) {
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const l2 a = {3, 4};
#pragma unroll
    for (int j = 0; j < 8; ++j) {                           // 8 KiB per wave, 1 KiB per store instruction
        const long long ii = wave * 512 + 64 * j + lane;
        if (ii < n2) { if (NT) __builtin_nontemporal_store(a, idx + ii); else idx[ii] = a; }
    }
}
#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
template <typename F> float time_us(F launch, int iters) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 400; ++i) launch(i);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(a)); for (int i = 0; i < iters; ++i) launch(i); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms * 1e3f / iters < best) best = ms * 1e3f / iters;
    }
    return best;
}
int main() {
    const long long N = 1ll << 26, n4 = N / 4;
    f4* x[3]; f4* q[3]; l2* idx[3];
    for (int i = 0; i < 3; ++i) { CHECK(hipMalloc(&x[i], N * 4)); CHECK(hipMalloc(&q[i], N * 4)); CHECK(hipMalloc(&idx[i], N * 8)); CHECK(hipMemset(x[i], 0x3c, N * 4)); }
    const int blocks = (int)(n4 / 256 / 4);
#define RUN(label, WQ, WI, NT, bytes) { float us = time_us([&](int i) { hipLaunchKernelGGL((k_mix<WQ, WI, NT>), dim3(blocks), dim3(256), 0, 0, x[i % 3], q[i % 3], idx[i % 3], n4); }, 40); \
        printf("%-52s %7.2f us  %6.0f GB/s  %5.1f%% of 8 TB/s\n", label, us, (double)bytes * N / us / 1e3, (double)bytes * N / us / 1e3 / 80); }
    RUN("read 4, write 4 (copy)            NT stores", 1, 0, 1, 8)
    RUN("read 4, write 4 (copy)            plain stores", 1, 0, 0, 8)
    RUN("read 4, write 8 (int64 only)      NT stores", 0, 1, 1, 12)
    RUN("read 4, write 8 (int64 only)      plain stores", 0, 1, 0, 12)
    RUN("read 4, write 4 + 8 (q + int64)   NT stores", 1, 1, 1, 16)
    RUN("read 4, write 4 + 8 (q + int64)   plain stores", 1, 1, 0, 16)
    { float us = time_us([&](int i) { hipLaunchKernelGGL((k_fill<1>), dim3((unsigned)(N / 2 / 8 / 256)), dim3(256), 0, 0, idx[i % 3], N / 2); }, 40);
      printf("%-52s %7.2f us  %6.0f GB/s  %5.1f%% of 8 TB/s\n", "write 8 (fill int64)               NT stores", us, 8.0 * N / us / 1e3, 8.0 * N / us / 1e3 / 80); }
    { float us = time_us([&](int i) { hipLaunchKernelGGL((k_fill<0>), dim3((unsigned)(N / 2 / 8 / 256)), dim3(256), 0, 0, idx[i % 3], N / 2); }, 40);
      printf("%-52s %7.2f us  %6.0f GB/s  %5.1f%% of 8 TB/s\n", "write 8 (fill int64)               plain stores", us, 8.0 * N / us / 1e3, 8.0 * N / us / 1e3 / 80); }
    return 0;
}
