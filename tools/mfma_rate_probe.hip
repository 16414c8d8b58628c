// mfma_rate_probe.hip -- development probe: sustained issue rate of the exact-f32 MFMAs (16x16x4 vs 32x32x2), 1..4 waves per
// SIMD, independent accumulators.  Prints cycles per instruction per SIMD and the fraction of the 256 FLOP/clk/CU peak.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1;} } while (0)

template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, int iters, unsigned long long *cyc) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float *out, int iters, unsigned long long *cyc) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; i++) for (int j = 0; j < 16; j++) acc[i][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K> int run(const char *name, K kern, int nacc, int wg_per_cu, double flop_per_inst) {
    float *out; unsigned long long *cyc, h;
    CK(hipMalloc(&out, 256 * 8 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    const int iters = 2000;
    hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters, cyc);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(256), 0, 0, out, iters * 10, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tflops = 256.0 * wg_per_cu * 4 * (double)iters * 10 * nacc * flop_per_inst / (ms * 1e-3) / 1e12;
    // one workgroup = 4 waves = one per SIMD; wg_per_cu waves per SIMD
    const double inst_per_simd = (double)iters * nacc * wg_per_cu;
    const double cpi = (double)h / inst_per_simd;
    printf("%-14s %d accumulators, %d workgroup(s) of 4 waves per CU: workgroup 0 alone %6.2f cycles per MFMA per SIMD (%5.1f %%); whole grid by events: %.1f TFLOP/s\n", name, nacc, wg_per_cu, cpi,
           100.0 * flop_per_inst / cpi / 64.0, tflops);
    (void)hipFree(out); (void)hipFree(cyc);
    return 0;
}

int main() {
    for (int w = 1; w <= 2; w++) {
        run("16x16x4 f32", k16<1>, 1, w, 2048.0);
        run("16x16x4 f32", k16<2>, 2, w, 2048.0);
        run("16x16x4 f32", k16<4>, 4, w, 2048.0);
        run("16x16x4 f32", k16<6>, 6, w, 2048.0);
        run("16x16x4 f32", k16<8>, 8, w, 2048.0);
        run("16x16x4 f32", k16<16>, 16, w, 2048.0);
        run("32x32x2 f32", k32<1>, 1, w, 4096.0);
        run("32x32x2 f32", k32<2>, 2, w, 4096.0);
        run("32x32x2 f32", k32<4>, 4, w, 4096.0);
    }
    return 0;
}
