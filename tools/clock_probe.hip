// clock_probe.hip -- diagnostic: what shader clock does the chip hold while a stream of short
// dependent kernels runs (our step pattern)?  clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>

__global__ void probe(unsigned long long *out, int iters, float *sink) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    float x = threadIdx.x;
    for (int i = 0; i < iters; i++) x = x * 1.0001f + 0.5f;
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
    if (x == 12345.f) *sink = x;
}
__global__ void empty_kernel(float *p) { if (p && threadIdx.x == 9999) *p = 1.f; }

int main() {
    unsigned long long *d; float *sink;
    hipMalloc(&d, 16); hipMalloc(&sink, 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    unsigned long long h[2];
    for (int round = 0; round < 6; round++) {
        // ~N short kernels back to back, then a probe
        int n = 200 << round;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; i++) hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s, (float *)nullptr);
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, s, d, 20000, sink);
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
        printf("after %6d empty launches (%.1f us each): memtime %llu realtime %llu -> %.0f MHz\n", n, us / n,
               h[0], h[1], (double)h[0] / (double)h[1] * 100.0);
    }
    // event-timed empty kernel and dependent chain
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int g : {1, 40, 256, 1024}) {
        hipEventRecord(a, s);
        for (int i = 0; i < 1000; i++) hipLaunchKernelGGL(empty_kernel, dim3(g), dim3(256), 0, s, (float *)nullptr);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("1000 empty kernels grid %4d: %.2f us each\n", g, ms);
    }
    return 0;
}
