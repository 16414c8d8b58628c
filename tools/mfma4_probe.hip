// mfma4_probe.hip -- development harness: issue rate of v_mfma_f32_4x4x1_16b_f32 on one SIMD (cycles per instruction),
// 1, 2, 4 accumulators, one wave alone and two waves sharing the SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void k(float *out, unsigned long long *cyc, int n) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc[u % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[u % NACC], 0, 0, 0);
    }
    f32x4 s = acc[0];
    for (int i = 1; i < NACC; i++) s += acc[i];
    asm volatile("" : "+v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
// dst != src C: four accumulators rotate through five register quads, as the compiler's allocation in rowblock_kernel does
__global__ void k_rot(float *out, unsigned long long *cyc, int n) {
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = r0, r2 = r0, r3 = r0, r4 = r0;
    float a = threadIdx.x * 0.001f, b = 1.0f + threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; i++) {
        asm volatile(
            "v_mfma_f32_4x4x1_16b_f32 %4, %5, %6, %0\n\tv_mfma_f32_4x4x1_16b_f32 %0, %5, %6, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %5, %6, %2\n\tv_mfma_f32_4x4x1_16b_f32 %2, %5, %6, %3\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %5, %6, %4\n\tv_mfma_f32_4x4x1_16b_f32 %4, %5, %6, %0\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %0, %5, %6, %1\n\tv_mfma_f32_4x4x1_16b_f32 %1, %5, %6, %2\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %2, %5, %6, %3\n\tv_mfma_f32_4x4x1_16b_f32 %3, %5, %6, %4\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %4, %5, %6, %0\n\tv_mfma_f32_4x4x1_16b_f32 %0, %5, %6, %1\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %1, %5, %6, %2\n\tv_mfma_f32_4x4x1_16b_f32 %2, %5, %6, %3\n\t"
            "v_mfma_f32_4x4x1_16b_f32 %3, %5, %6, %4\n\tv_mfma_f32_4x4x1_16b_f32 %4, %5, %6, %0\n\ts_nop 7\n\ts_nop 7"
            : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4) : "v"(a), "v"(b));
    }
    f32x4 s = r0 + r1 + r2 + r3 + r4;
    asm volatile("" : "+v"(s));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64);
    const int n = 1000;
    auto run = [&](const char *name, auto kern, int threads) {
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, n);
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, out, cyc, n);
        hipDeviceSynchronize();
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("%-40s %6.2f cycles per MFMA (wave 0)\n", name, (double)h / (16.0 * n));
    };
    run("4x4x1 f32, 1 acc, 1 wave", k<1>, 64);
    run("4x4x1 f32, 2 acc, 1 wave", k<2>, 64);
    run("4x4x1 f32, 4 acc, 1 wave", k<4>, 64);
    run("4x4x1 f32, 4 acc, 4 waves (1 per SIMD)", k<4>, 256);
    run("4x4x1 f32, 4 acc, 8 waves (2 per SIMD)", k<4>, 512);
    run("4x4x1 f32, 1 acc, 8 waves (2 per SIMD)", k<1>, 512);
    run("4x4x1 f32, rotating dst != C, 1 wave", k_rot, 64);
    run("4x4x1 f32, rotating dst != C, 8 waves", k_rot, 512);
    return 0;
}
