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

__global__ void pingpong(const float4 *in, float4 *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = in[i];
    v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
    out[i] = v;
}

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
    {   // dependent chain of minimal load -> store kernels: the floor of one "real" kernel
        float4 *pa, *pb;
        hipMalloc(&pa, 1 << 22); hipMalloc(&pb, 1 << 22);
        hipMemset(pa, 0, 1 << 22); hipMemset(pb, 0, 1 << 22);
        for (int g : {8, 64, 256, 1024}) {
            for (int i = 0; i < 50; i++) { hipLaunchKernelGGL(pingpong, dim3(g), dim3(256), 0, s, pa, pb); hipLaunchKernelGGL(pingpong, dim3(g), dim3(256), 0, s, pb, pa); }
            hipEventRecord(a, s);
            for (int i = 0; i < 500; i++) { hipLaunchKernelGGL(pingpong, dim3(g), dim3(256), 0, s, pa, pb); hipLaunchKernelGGL(pingpong, dim3(g), dim3(256), 0, s, pb, pa); }
            hipEventRecord(b, s); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("dependent load->store kernels, grid %4d x 256 threads (16 B per thread): %.2f us each\n", g, ms);
        }
        // the same chain captured in a graph (no host launch limit)
        hipGraph_t graph; hipGraphExec_t exec;
        hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < 100; i++) { hipLaunchKernelGGL(pingpong, dim3(256), dim3(256), 0, s, pa, pb); hipLaunchKernelGGL(pingpong, dim3(256), dim3(256), 0, s, pb, pa); }
        hipStreamEndCapture(s, &graph);
        hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphLaunch(exec, s); hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int i = 0; i < 10; i++) hipGraphLaunch(exec, s);
        hipEventRecord(b, s); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("same chain, hipGraph (200 kernels per graph): %.2f us each\n", ms * 1000.f / 2000.f);
    }
    return 0;
}
