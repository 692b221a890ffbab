// Host cost of one kernel launch through the different HIP entry points (what bounds the drop-in API call:
// quantization.uniformQuantization on a small tensor is ~4.5 us of host time, ~3 us of it inside the launch).
//   hipcc --offload-arch=gfx950 -O2 tools/launch_probe.hip -o build/launch_probe && ./build/launch_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdint>

struct P { const float* x; float* out; int64_t n, row, nb; float *a, *b; const float* mean; float me, sm1; uint8_t* l; void* idx;
           int ib; const float* pts; int k, am, pre, st; uint64_t seed; int64_t nvec; };
__global__ void kern(P p) { if (p.n < 0) p.out[0] = 1.f; }

template <typename F> double per_launch_us(hipStream_t st, F&& f, int iters = 20000) {
    for (int i = 0; i < 200; ++i) f();
    (void)hipStreamSynchronize(st);
    double tot = 0;
    for (int done = 0; done < iters; done += 64) {          // shallow queue: sync outside the timed part
        auto a = std::chrono::steady_clock::now();
        for (int i = 0; i < 64; ++i) f();
        auto b = std::chrono::steady_clock::now();
        tot += std::chrono::duration<double, std::micro>(b - a).count();
        (void)hipStreamSynchronize(st);
    }
    return tot / iters;
}

int main() {
    float* d; (void)hipMalloc(&d, 1024);
    P p = {}; p.x = d; p.out = d; p.n = 500;
    for (int which = 0; which < 2; ++which) {
        hipStream_t st = nullptr;
        if (which == 1) (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        printf("stream: %s\n", which == 0 ? "null (torch's default current stream)" : "a non-blocking stream");
        printf("  kern<<<>>>                 %.2f us\n", per_launch_us(st, [&] { kern<<<dim3(1), dim3(256), 0, st>>>(p); }));
        void* args[] = {&p};
        printf("  hipLaunchKernel            %.2f us\n", per_launch_us(st, [&] { (void)hipLaunchKernel((const void*)kern, dim3(1), dim3(256), args, 0, st); }));
        hipFunction_t fn;
        if (hipGetFuncBySymbol(&fn, (const void*)kern) == hipSuccess) {
            printf("  hipModuleLaunchKernel      %.2f us\n", per_launch_us(st, [&] { (void)hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, st, args, nullptr); }));
            size_t sz = sizeof(P);
            void* cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &p, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            printf("  hipModuleLaunchKernel(buf) %.2f us\n", per_launch_us(st, [&] { (void)hipModuleLaunchKernel(fn, 1, 1, 1, 256, 1, 1, 0, st, nullptr, cfg); }));
            printf("  hipExtModuleLaunchKernel   %.2f us\n", per_launch_us(st, [&] { (void)hipExtModuleLaunchKernel(fn, 256, 1, 1, 256, 1, 1, 0, st, args, nullptr, nullptr, nullptr, 0); }));
        }
        hipLaunchConfig_t cfgx = {};
        cfgx.gridDim = dim3(1); cfgx.blockDim = dim3(256); cfgx.dynamicSmemBytes = 0; cfgx.stream = st; cfgx.attrs = nullptr; cfgx.numAttrs = 0;
        printf("  hipLaunchKernelExC         %.2f us\n", per_launch_us(st, [&] { (void)hipLaunchKernelExC(&cfgx, (const void*)kern, args); }));
    }
    printf("hipGetLastError: %d\n", (int)hipGetLastError());
    return 0;
}
