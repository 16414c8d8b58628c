// gemm_probe.hip -- development harness (not shipped): times gemm_f32_kernel instantiations (tile, waves, register
// stages) on the GEMM shapes of BASELINE configs[3] (4096-2048-2048-1024, 512 rows) and configs[4]
// (784-1024-1024-1024-10, 256 rows).  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_probe.hip -o tools/_build/gemm_probe
// Round 4's modes (tools/_build/gemm_probe <mode>; logs under profiles/r04/gemm_probe_*):
//   30, 31  bf16 tile shapes; the bf16 gradient + update products          32, 38  what one K-slice of a split-K product costs (register-staged / DMA)
//   33, 35  main-loop forms of gemm_bf16_kernel (LDS images, fragment depth, waves)   34  s_memtime stamps inside its loop (GEMM_PROBE_STAMPS=1)
//   36      that loop compiled without its global loads / MFMAs / LDS reads / LDS writes (NSTG + 10 x ablation)
//   37      gemm_bf16_dma_kernel: checked element by element against gemm_bf16_kernel, then timed      40  configs[4]'s bf16 products
//   41      gemm_f32_kernel as shipped (build with -DGNN_F32_NO_LDS_WRITES for the loop without its LDS writes)
//   42      gemm_f32_dma_kernel: checked against gemm_f32_kernel, then timed   43, 44  DMA form: K depth, a fourth image
#define GNN_GEMM_BF16_STAMPS
#include "../graph-neural-net_amd/csrc/gemm_wavek.h"
#include "../graph-neural-net_amd/csrc/gemm_bf16.h"
#include "../graph-neural-net_amd/csrc/gemm_bf16_dma.h"
#include "../graph-neural-net_amd/csrc/gemm_f32_dma.h"
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
using namespace gnn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

static float *dA, *dB, *dC, *dAux, *dW, *dV;
static int g_pad = 0; // GEMM_PROBE_PAD: extra elements on every leading dimension (is a power-of-two row stride a cost? mode 20)

template <int BM, int BN, bool AK, bool BKC, int EPI, int WM, int NSTG, int BKT = 0>
void run(const char *what, int M, int N, int K) {
    GemmParams p{};
    p.A = dA; p.lda = (AK ? K : M) + g_pad;
    p.B = dB; p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad;
    p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    p.aux = dAux; p.ldaux = N + g_pad; p.W = dW; p.V = dV; p.step_over_b = 1e-6f; p.momentum = 0.9f; p.act = 0;
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(WM * 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BKC, EPI, WM, NSTG, BKT>), grid, block, 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BKC, EPI, WM, NSTG, BKT>), grid, block, 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s %4dx%4dx%4d  tile %3dx%-3d waves %d stages %d BK %3d  %8.2f us  %6.1f TFLOP/s (%4.1f %%)  %d tiles\n", what, M, N, K, BM, BN, WM * 2, NSTG, BKT ? BKT : (BM <= 32 ? 128 : BM <= 64 ? 64 : 32), us, tf,
           100.0 * tf / 157.3, (int)(grid.x * grid.y));
    fflush(stdout);
}

template <bool AK, bool BKC, int EPI, int NW, int DEPTH, bool HEAD = true>
void runk(const char *what, int M, int N, int K) {
    GemmParams p{};
    p.A = dA; p.lda = (AK ? K : M) + g_pad;
    p.B = dB; p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad;
    p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    p.aux = dAux; p.ldaux = N + g_pad; p.W = dW; p.V = dV; p.step_over_b = 1e-6f; p.momentum = 0.9f; p.act = 0;
    dim3 grid(N / 32, M / 32), block(NW * 64);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gemm_f32_wavek_kernel<AK, BKC, EPI, NW, DEPTH, HEAD>), grid, block, 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gemm_f32_wavek_kernel<AK, BKC, EPI, NW, DEPTH, HEAD>), grid, block, 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s %4dx%4dx%4d  wave-K  32x32  waves %d depth %d   %8.2f us  %6.1f TFLOP/s (%4.1f %%)  %d tiles\n", what, M, N, K, NW, DEPTH, us, tf,
           100.0 * tf / 157.3, (int)(grid.x * grid.y));
    fflush(stdout);
}

// compares the wave-K kernel with gemm_f32_kernel on one shape (max abs difference of C, number of elements off by > 1e-5)
template <bool AK, bool BKC, int EPI, int NW, int DEPTH>
void check(int M, int N, int K) {
    GemmParams p{};
    p.A = dA; p.lda = (AK ? K : M) + g_pad;
    p.B = dB; p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad;
    p.aux = dW; p.ldaux = N;
    p.M = M; p.N = N; p.K = K; p.m_true = M - 3; p.n_true = N - 5; p.act = 0;
    hipLaunchKernelGGL((gemm_f32_kernel<32, 32, AK, BKC, EPI, 2>), dim3(N / 32, M / 32), dim3(256), 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    p.C = dAux;
    hipLaunchKernelGGL((gemm_f32_wavek_kernel<AK, BKC, EPI, NW, DEPTH>), dim3(N / 32, M / 32), dim3(NW * 64), 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
    std::vector<float> a((size_t)M * N), b((size_t)M * N);
    CK(hipMemcpy(a.data(), dC, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), dAux, b.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; size_t bad = 0, first = 0;
    for (size_t i = 0; i < a.size(); i++) {
        const double d = fabsf(a[i] - b[i]);
        if (d > 1e-5 && !bad++) first = i;
        md = std::max(md, d); mx = std::max(mx, (double)fabsf(a[i]));
    }
    printf("check A_KC=%d B_KC=%d epi %d %dx%dx%d waves %d depth %d: max |diff| %.3g (max |C| %.3g), %zu elements off", AK, BKC, EPI, M, N, K, NW, DEPTH, md, mx, bad);
    if (bad) printf(" (first at row %zu col %zu)", first / N, first % N);
    printf("\n");
}

// bf16 GEMM (gemm_bf16.h) on the same buffers reinterpreted (timing only; values are whatever the bits are)
template <int BM, int BN, bool AK, bool BKC, int EPI, int NSTG = 2, int WM = 2>
void runb(const char *what, int M, int N, int K) {
    GemmBf16Params p{};
    p.A = reinterpret_cast<const __bf16 *>(dA); p.lda = (AK ? K : M) + g_pad;
    p.B = reinterpret_cast<const __bf16 *>(dB); p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad; p.Cb = reinterpret_cast<__bf16 *>(dAux);
    p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    p.aux = dW; p.ldaux = N + g_pad; p.W = dW; p.V = dV; p.Wb = reinterpret_cast<__bf16 *>(dAux); p.step_over_b = 1e-6f; p.momentum = 0.9f; p.act = 0;
    if (EPI == EPI_SGD) { p.C = nullptr; p.Cb = nullptr; }
    constexpr size_t lds = gemm_bf16_lds_bytes<BM, BN, AK, BKC, NSTG % 10>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<BM, BN, AK, BKC, EPI, NSTG, WM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM), block(WM * 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, EPI, NSTG, WM>), grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, EPI, NSTG, WM>), grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    if (getenv("GEMM_PROBE_STAMPS") && (NSTG % 10 == 2 || NSTG % 10 == 4)) {
        unsigned long long st[16];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gnn_bf16_stamps), sizeof(st)));
        const double steps = (double)((K + GemmBf16Depth<BM>::BK - 1) / GemmBf16Depth<BM>::BK);
        printf("    cycles per tile of K (wave 0 of workgroup 0; %d tiles): write image (waits for its loads) %.0f | barrier %.0f | issue loads + multiply %.0f | barrier %.0f | loop top %.0f   (whole loop: %.0f per tile)\n", (int)steps,
               (st[1] + st[5]) / steps, (st[2] + st[6]) / steps, (st[3] + st[7]) / steps, (st[4] + st[8]) / steps, st[0] / steps, st[9] / steps);
    }
    printf("%-34s %4dx%4dx%4d  bf16 tile %3dx%-3d stages %d waves %d  %8.2f us  %6.1f TFLOP/s (%4.1f %% of 2500)  %d tiles\n", what, M, N, K, BM, BN, NSTG, WM * 2, us, tf, 100.0 * tf / 2500.0,
           (int)(grid.x * grid.y));
    fflush(stdout);
}

// the DMA form (gemm_bf16_dma.h): timing, and -- EPI_STORE / EPI_ACT -- its result against gemm_bf16_kernel's on the same operands
template <int BM, int BN, bool AK, bool BKC, int EPI, int NIMG = 3, int WM = 2, int BKO = 0>
void rund(const char *what, int M, int N, int K, bool check = false) {
    GemmBf16Params p{};
    p.A = reinterpret_cast<const __bf16 *>(dA); p.lda = (AK ? K : M) + g_pad;
    p.B = reinterpret_cast<const __bf16 *>(dB); p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad; p.Cb = reinterpret_cast<__bf16 *>(dAux);
    p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    p.aux = dW; p.ldaux = N + g_pad; p.W = dW; p.V = dV; p.Wb = reinterpret_cast<__bf16 *>(dAux); p.step_over_b = 1e-6f; p.momentum = 0.9f; p.act = 0;
    if (EPI == EPI_SGD) { p.C = nullptr; p.Cb = nullptr; }
    constexpr size_t lds = gemm_bf16_dma_lds_bytes<BM, BN, NIMG, BKO>();
    auto kern = gemm_bf16_dma_kernel<BM, BN, AK, BKC, EPI, WM, NIMG, BKO>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (M % BM || N % BN || K % GemmBf16Dma<BM, BN, BKO>::BK) { printf("%s: shape does not fit the DMA form\n", what); return; }
    dim3 grid(N / BN, M / BM), block(WM * 128);
    if (check && EPI != EPI_SGD) {
        std::vector<float> c0((size_t)M * p.ldc), c1((size_t)M * p.ldc);
        CK(hipMemset(dC, 0, (size_t)M * p.ldc * 4));
        hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dC, 0, (size_t)M * p.ldc * 4));
        constexpr size_t lds0 = gemm_bf16_lds_bytes<BM, BN, AK, BKC, 2>();
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<BM, BN, AK, BKC, EPI, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0));
        hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, AK, BKC, EPI, 2, 2>), grid, dim3(256), lds0, 0, GNN_GEMM_HEAD_ARGS(p), p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0; size_t bad = 0;
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) {
                const double a = c0[(size_t)m * p.ldc + n], b = c1[(size_t)m * p.ldc + n];
                if (!(std::fabs(a - b) <= 1e-4 * std::max(1.0, std::fabs(a)))) bad++;
                worst = std::max(worst, std::fabs(a - b)); big = std::max(big, std::fabs(a));
            }
        printf("    check against gemm_bf16_kernel: max |diff| %.3g (max |value| %.3g), %zu of %zu elements differ by more than 1e-4\n", worst, big, bad, (size_t)M * N);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s %4dx%4dx%4d  bf16 DMA  %3dx%-3d BK %3d images %d waves %d  %8.2f us  %6.1f TFLOP/s (%4.1f %% of 2500)  %d tiles\n", what, M, N, K, BM, BN, GemmBf16Dma<BM, BN, BKO>::BK, NIMG, WM * 2, us, tf, 100.0 * tf / 2500.0,
           (int)(grid.x * grid.y));
    fflush(stdout);
}

// the f32 DMA form (gemm_f32_dma.h): timing, and its result against gemm_f32_kernel's on the same operands
template <int BM, int BN, bool AK, bool BKC, int EPI, int NIMG = 3, int WM = 4>
void runfd(const char *what, int M, int N, int K, bool check = false) {
    GemmParams p{};
    p.A = dA; p.lda = (AK ? K : M) + g_pad;
    p.B = dB; p.ldb = (BKC ? K : N) + g_pad;
    p.C = dC; p.ldc = N + g_pad;
    p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    p.aux = dAux; p.ldaux = N + g_pad; p.W = dW; p.V = dV; p.step_over_b = 1e-6f; p.momentum = 0.9f; p.act = 0;
    constexpr size_t lds = gemm_f32_dma_lds_bytes<BM, BN, NIMG>();
    auto kern = gemm_f32_dma_kernel<BM, BN, AK, BKC, EPI, WM, NIMG>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (M % BM || N % BN || K % GemmF32DmaDepth<BM>::BK) { printf("%s: shape does not fit the DMA form\n", what); return; }
    dim3 grid(N / BN, M / BM), block(WM * 128);
    if (check && EPI != EPI_SGD) {
        std::vector<float> c0((size_t)M * p.ldc), c1((size_t)M * p.ldc);
        CK(hipMemset(dC, 0, (size_t)M * p.ldc * 4));
        hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c1.data(), dC, c1.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(dC, 0, (size_t)M * p.ldc * 4));
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, AK, BKC, EPI, WM>), grid, block, 0, 0, GNN_GEMM_HEAD_ARGS(p), p);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(c0.data(), dC, c0.size() * 4, hipMemcpyDeviceToHost));
        double worst = 0, big = 0; size_t bad = 0;
        for (int m = 0; m < M; m++)
            for (int n = 0; n < N; n++) {
                const double a = c0[(size_t)m * p.ldc + n], b = c1[(size_t)m * p.ldc + n];
                if (!(std::fabs(a - b) <= 2e-5 * std::max(0.05, std::fabs(a)))) bad++;
                worst = std::max(worst, std::fabs(a - b)); big = std::max(big, std::fabs(a));
            }
        printf("    check against gemm_f32_kernel: max |diff| %.3g (max |value| %.3g), %zu of %zu elements differ by more than 2e-5 relative\n", worst, big, bad, (size_t)M * N);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipDeviceSynchronize());
    const int iters = 30;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL(kern, grid, block, lds, 0, GNN_GEMM_HEAD_ARGS(p), p);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-34s %4dx%4dx%4d  f32 DMA %3dx%-3d images %d waves %d  %8.2f us  %6.1f TFLOP/s (%4.1f %%)  %d tiles\n", what, M, N, K, BM, BN, NIMG, WM * 2, us, tf, 100.0 * tf / 157.3, (int)(grid.x * grid.y));
    fflush(stdout);
}

// One wave that watches the clocks for `ticks` of the 100 MHz real-time counter: shader clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
__global__ void clock_watch(unsigned long long *out, unsigned long long ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) { __builtin_amdgcn_s_sleep(32); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

// shader clock while gemm_f32_kernel<128, 128> runs back to back on another stream (and with the chip idle)
static void clock_under_load() {
    unsigned long long *d, h[2];
    CK(hipMalloc(&d, 16));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipLaunchKernelGGL(clock_watch, dim3(1), dim3(64), 0, sb, d, 300000ull);
    CK(hipStreamSynchronize(sb)); CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
    printf("idle chip:                         %.0f MHz\n", (double)h[0] / (double)h[1] * 100.0);
    GemmParams p{};
    const int M = 4096, N = 2048, K = 512;
    p.A = dA; p.lda = M; p.B = dB; p.ldb = N; p.C = dC; p.ldc = N; p.M = M; p.N = N; p.K = K; p.m_true = M; p.n_true = N;
    for (int rep = 0; rep < 3; rep++) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, sa));
        const int iters = 300;
        for (int i = 0; i < iters; i++)
            hipLaunchKernelGGL((gemm_f32_kernel<128, 128, false, false, EPI_STORE, 2>), dim3(N / 128, M / 128), dim3(256), 0, sa, GNN_GEMM_HEAD_ARGS(p), p);
        CK(hipEventRecord(e1, sa));
        hipLaunchKernelGGL(clock_watch, dim3(1), dim3(64), 0, sb, d, 1000000ull); // 10 ms inside the ~25 ms of GEMMs
        CK(hipStreamSynchronize(sb)); CK(hipStreamSynchronize(sa));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost));
        const double mhz = (double)h[0] / (double)h[1] * 100.0, us = ms * 1e3 / iters, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
        printf("under f32 GEMM 4096x2048x512 (128x128 tiles, store epilogue, %d launches back to back): %.0f MHz; %.2f us = %.1f TFLOP/s = %.1f %% of 157.3 "
               "(%.1f %% of the peak at this clock, 256 CUs x 256 FLOP/clk)\n", iters, mhz, us, tf, 100.0 * tf / 157.3, 100.0 * tf / (256.0 * 256.0 * mhz * 1e6 / 1e12));
    }
}

int main(int argc, char **argv) {
    if (getenv("GEMM_PROBE_PAD")) g_pad = atoi(getenv("GEMM_PROBE_PAD"));
    const size_t n = (size_t)4096 * (2048 + 256) + 4096;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; i++) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 0.1f;
    float **bufs[] = {&dA, &dB, &dC, &dAux, &dW, &dV};
    for (auto b : bufs) { CK(hipMalloc(b, n * 4)); CK(hipMemcpy(*b, h.data(), n * 4, hipMemcpyHostToDevice)); }
    const int which = argc > 1 ? atoi(argv[1]) : 0;
    if (which == 10) {
        printf("---- bf16 GEMMs of configs[3] (each twice)\n");
        for (int rep = 0; rep < 2; rep++) {
            runb<64, 64, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 1", 512, 2048, 4096);
            runb<64, 64, true, false, EPI_ACT, 2, 2>("forward 2", 512, 2048, 2048);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 2", 512, 2048, 2048);
            runb<64, 64, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            runb<64, 64, true, false, EPI_STORE, 2, 4>("logits", 512, 1024, 2048);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 2", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 2, 4>("backward data 2", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 1", 512, 2048, 2048);
            runb<64, 64, true, true, EPI_DACT, 2, 4>("backward data 1", 512, 2048, 2048);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 0", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 1", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 2", 2048, 1024, 512);
        }
        return 0;
    }
    if (which == 32) {
        printf("---- round 4: what ONE K-slice of a split-K product would cost (256 workgroups of a large tile, each over K / S) against the shipped 64 x 64 tiles over all of K\n");
        for (int rep = 0; rep < 2; rep++) {
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 1 (shipped)", 512, 2048, 4096);
            runb<128, 128, true, false, EPI_ACT, 2, 2>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            runb<128, 128, true, false, EPI_ACT, 2, 4>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            runb<128, 64, true, false, EPI_ACT, 2, 2>("forward 1: 128 tiles x 2 slices as", 1024, 2048, 2048);
            runb<128, 64, true, false, EPI_ACT, 2, 4>("forward 1: 128 tiles x 2 slices as", 1024, 2048, 2048);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 2 (shipped)", 512, 2048, 2048);
            runb<128, 128, true, false, EPI_ACT, 2, 2>("forward 2: 64 tiles x 4 slices as", 2048, 2048, 512);
            runb<128, 64, true, false, EPI_ACT, 2, 2>("forward 2: 128 tiles x 2 slices as", 1024, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 1 (shipped)", 512, 2048, 2048);
            runb<128, 128, true, true, EPI_DACT, 2, 2>("backward data 1: 64 x 4 as", 2048, 2048, 512);
            runb<128, 64, true, true, EPI_DACT, 2, 2>("backward data 1: 128 x 2 as", 1024, 2048, 1024);
            runb<128, 128, true, false, EPI_ACT, 2, 2>("(one slice alone: 64 workgroups)", 512, 2048, 1024);
        }
        return 0;
    }
    if (which == 33) {
        printf("---- round 4: two LDS images and one barrier per tile (stages 3) against the shipped main loop (stages 2)\n");
        for (int rep = 0; rep < 2; rep++) {
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 1 (shipped)", 512, 2048, 4096);
            runb<64, 64, true, false, EPI_ACT, 3, 4>("forward 1", 512, 2048, 4096);
            runb<64, 64, true, false, EPI_ACT, 3, 2>("forward 1", 512, 2048, 4096);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 2 (shipped)", 512, 2048, 2048);
            runb<64, 64, true, false, EPI_ACT, 3, 4>("forward 2", 512, 2048, 2048);
            runb<64, 64, true, false, EPI_ACT, 3, 2>("forward 2", 512, 2048, 2048);
            runb<32, 64, true, false, EPI_STORE, 2, 2>("logits (shipped)", 512, 1024, 2048);
            runb<32, 64, true, false, EPI_STORE, 3, 2>("logits", 512, 1024, 2048);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 2 (shipped)", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 3, 2>("backward data 2", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 3, 4>("backward data 2", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 1 (shipped)", 512, 2048, 2048);
            runb<64, 64, true, true, EPI_DACT, 3, 2>("backward data 1", 512, 2048, 2048);
            runb<64, 64, true, true, EPI_DACT, 3, 4>("backward data 1", 512, 2048, 2048);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0 (shipped)", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 3, 2>("gradient + update 0", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1 (shipped)", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 3, 2>("gradient + update 1", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2 (shipped)", 2048, 1024, 512);
            runb<64, 64, false, false, EPI_SGD, 3, 2>("gradient + update 2", 2048, 1024, 512);
            runb<128, 128, true, false, EPI_ACT, 3, 2>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            runb<128, 128, true, false, EPI_ACT, 3, 4>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            runb<128, 128, true, true, EPI_DACT, 3, 2>("backward data 1: 64 x 4 as", 2048, 2048, 512);
        }
        return 0;
    }
    if (which == 34) {
        printf("---- round 4: where the cycles of the shipped bf16 main loop go (GEMM_PROBE_STAMPS=1; GEMM_PROBE_MODE=1: no global loads in the loop, 2: no MFMAs)\n");

        runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 1 (shipped)", 512, 2048, 4096);
        runb<64, 64, true, false, EPI_ACT, 4, 4>("forward 1, deep fragment prefetch", 512, 2048, 4096);
        runb<64, 64, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
        runb<64, 64, true, false, EPI_ACT, 4, 2>("forward 1, deep fragment prefetch", 512, 2048, 4096);
        runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 2 (shipped)", 512, 2048, 2048);
        runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 1 (shipped)", 512, 2048, 2048);
        runb<64, 64, true, true, EPI_DACT, 4, 2>("backward data 1, deep fragment prefetch", 512, 2048, 2048);
        runb<64, 64, true, true, EPI_DACT, 2, 4>("backward data 1", 512, 2048, 2048);
        runb<64, 64, true, true, EPI_DACT, 4, 4>("backward data 1, deep fragment prefetch", 512, 2048, 2048);
        runb<64, 64, false, false, EPI_SGD, 4, 2>("gradient + update 0, deep fragment prefetch", 4096, 2048, 512);
        runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0 (shipped)", 4096, 2048, 512);
        runb<128, 128, true, false, EPI_ACT, 2, 2>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
        runb<128, 128, true, false, EPI_ACT, 2, 2>("(one slice alone: 64 workgroups)", 512, 2048, 1024);
        runb<128, 128, true, true, EPI_DACT, 2, 2>("backward data 1: 64 x 4 as", 2048, 2048, 512);
        return 0;
    }
    if (which == 35) {
        printf("---- round 4: main-loop forms of the bf16 GEMM (stages 2 = shipped before; 3 = two LDS images + all fragments of a tile up front; 4 = one image, fragments up front; 5 = two images, one block ahead)\n");
        for (int rep = 0; rep < 2; rep++) {
#define FORMS(name, BM_, BN_, AK, BKC, EPI, M, N, K) \
            runb<BM_, BN_, AK, BKC, EPI, 2, 2>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 2, 4>(name, M, N, K); \
            runb<BM_, BN_, AK, BKC, EPI, 3, 2>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 3, 4>(name, M, N, K); \
            runb<BM_, BN_, AK, BKC, EPI, 4, 2>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 4, 4>(name, M, N, K); \
            runb<BM_, BN_, AK, BKC, EPI, 5, 2>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 5, 4>(name, M, N, K);
            FORMS("forward 1", 64, 64, true, false, EPI_ACT, 512, 2048, 4096)
            FORMS("forward 2", 64, 64, true, false, EPI_ACT, 512, 2048, 2048)
            FORMS("backward data 2", 64, 64, true, true, EPI_DACT, 512, 2048, 1024)
            FORMS("backward data 1", 64, 64, true, true, EPI_DACT, 512, 2048, 2048)
            FORMS("gradient + update 0", 64, 64, false, false, EPI_SGD, 4096, 2048, 512)
            FORMS("gradient + update 1", 64, 64, false, false, EPI_SGD, 2048, 2048, 512)
            FORMS("gradient + update 2", 64, 64, false, false, EPI_SGD, 2048, 1024, 512)
            runb<32, 64, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            runb<32, 64, true, false, EPI_STORE, 3, 2>("logits", 512, 1024, 2048);
            runb<32, 64, true, false, EPI_STORE, 4, 2>("logits", 512, 1024, 2048);
            runb<32, 64, true, false, EPI_STORE, 5, 2>("logits", 512, 1024, 2048);
            runb<64, 64, true, false, EPI_STORE, 3, 4>("logits", 512, 1024, 2048);
        }
        return 0;
    }
    if (which == 36) {
        printf("---- round 4: ablations of the bf16 main loop (stages 2; 12: no global loads inside the loop, 22: no MFMAs, 32: no LDS reads, 42: no LDS writes)\n");
#define ABLS(name, BM_, BN_, AK, BKC, EPI, WM_, M, N, K) \
        runb<BM_, BN_, AK, BKC, EPI, 2, WM_>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 12, WM_>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 22, WM_>(name, M, N, K); \
        runb<BM_, BN_, AK, BKC, EPI, 32, WM_>(name, M, N, K); runb<BM_, BN_, AK, BKC, EPI, 42, WM_>(name, M, N, K);
        ABLS("forward 1", 64, 64, true, false, EPI_ACT, 2, 512, 2048, 4096)
        ABLS("forward 1", 64, 64, true, false, EPI_ACT, 4, 512, 2048, 4096)
        ABLS("backward data 1", 64, 64, true, true, EPI_DACT, 2, 512, 2048, 2048)
        return 0;
    }
    if (which == 37) {
        printf("---- round 4: operand tiles by LDS DMA (gemm_bf16_dma.h) against the register-staged kernel (stages 5, eight waves: what ships)\n");
        { // operands that are numbers in BOTH halves of every float
            const size_t nb = ((size_t)4096 * (2048 + 256) + 4096) * 2;
            std::vector<unsigned short> hb(nb);
            for (size_t i = 0; i < nb; i++) { const float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; memcpy(&u, &f, 4); hb[i] = (unsigned short)(u >> 16); }
            CK(hipMemcpy(dA, hb.data(), nb * 2, hipMemcpyHostToDevice));
            for (size_t i = 0; i < nb; i++) { const float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; memcpy(&u, &f, 4); hb[i] = (unsigned short)(u >> 16); }
            CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
        }
        rund<64, 64, true, false, EPI_STORE, 3, 2>("forward 1", 512, 2048, 4096, true);
        rund<64, 64, true, false, EPI_STORE, 2, 4>("forward 1", 512, 2048, 4096, true);
        rund<64, 64, true, true, EPI_STORE, 3, 4>("backward data 1", 512, 2048, 2048, true);
        rund<64, 64, false, false, EPI_STORE, 3, 2>("gradient 1", 2048, 2048, 512, true);
        rund<32, 64, true, false, EPI_STORE, 3, 2>("logits", 512, 1024, 2048, true);
        rund<128, 128, true, false, EPI_STORE, 2, 2>("forward, 128 x 128", 2048, 2048, 1024, true);
        rund<128, 128, false, false, EPI_STORE, 2, 2>("gradient, 128 x 128", 2048, 2048, 512, true);
        rund<128, 128, true, true, EPI_STORE, 2, 2>("backward, 128 x 128", 2048, 2048, 512, true);
        for (int rep = 0; rep < 2; rep++) {
#define DFORMS(name, BM_, BN_, AK, BKC, EPI, M, N, K) \
            runb<BM_, BN_, AK, BKC, EPI, 5, 4>(name, M, N, K); \
            rund<BM_, BN_, AK, BKC, EPI, 2, 2>(name, M, N, K); rund<BM_, BN_, AK, BKC, EPI, 2, 4>(name, M, N, K); \
            rund<BM_, BN_, AK, BKC, EPI, 3, 2>(name, M, N, K); rund<BM_, BN_, AK, BKC, EPI, 3, 4>(name, M, N, K);
            DFORMS("forward 1", 64, 64, true, false, EPI_ACT, 512, 2048, 4096)
            DFORMS("forward 2", 64, 64, true, false, EPI_ACT, 512, 2048, 2048)
            DFORMS("backward data 2", 64, 64, true, true, EPI_DACT, 512, 2048, 1024)
            DFORMS("backward data 1", 64, 64, true, true, EPI_DACT, 512, 2048, 2048)
            DFORMS("gradient + update 0", 64, 64, false, false, EPI_SGD, 4096, 2048, 512)
            DFORMS("gradient + update 1", 64, 64, false, false, EPI_SGD, 2048, 2048, 512)
            DFORMS("gradient + update 2", 64, 64, false, false, EPI_SGD, 2048, 1024, 512)
            runb<32, 64, true, false, EPI_STORE, 5, 2>("logits", 512, 1024, 2048);
            rund<32, 64, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            rund<32, 64, true, false, EPI_STORE, 3, 2>("logits", 512, 1024, 2048);
            rund<64, 64, true, false, EPI_STORE, 3, 4>("logits", 512, 1024, 2048);
        }
        return 0;
    }
    if (which == 38) {
        printf("---- round 4: DMA form, 128 x 128 tiles: one K-slice of a split-K product per workgroup (256 workgroups)\n");
        for (int rep = 0; rep < 2; rep++) {
            rund<128, 128, true, false, EPI_ACT, 2, 2>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            rund<128, 128, true, false, EPI_ACT, 3, 2>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            rund<128, 128, true, false, EPI_ACT, 2, 4>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            rund<128, 128, true, false, EPI_ACT, 3, 4>("forward 1: 64 tiles x 4 slices as", 2048, 2048, 1024);
            rund<128, 128, true, false, EPI_ACT, 3, 4>("forward 2: 64 tiles x 4 slices as", 2048, 2048, 512);
            rund<128, 128, true, true, EPI_DACT, 3, 4>("backward data 1: 64 x 4 as", 2048, 2048, 512);
            rund<128, 128, true, true, EPI_DACT, 3, 2>("backward data 1: 64 x 4 as", 2048, 2048, 512);
            rund<128, 128, true, false, EPI_ACT, 3, 4>("(big square)", 4096, 2048, 2048);
            rund<64, 64, true, false, EPI_ACT, 3, 4>("(big square)", 4096, 2048, 2048);
            runb<128, 128, true, false, EPI_ACT, 2, 2>("(big square)", 4096, 2048, 2048);
        }
        return 0;
    }
    if (which == 40) {
        printf("---- round 4: configs[4]'s bf16 products (256 rows): the shipped 32 x 32 register-staged tiles against two images and the DMA form\n");
        for (int rep = 0; rep < 2; rep++) {
            runb<32, 32, true, false, EPI_ACT, 2, 2>("forward 2/3 (shipped)", 256, 1024, 1024);
            runb<32, 32, true, false, EPI_ACT, 5, 2>("forward 2/3", 256, 1024, 1024);
            rund<32, 32, true, false, EPI_ACT, 2, 2>("forward 2/3", 256, 1024, 1024);
            rund<32, 32, true, false, EPI_ACT, 3, 2>("forward 2/3", 256, 1024, 1024);
            rund<32, 64, true, false, EPI_ACT, 2, 2>("forward 2/3", 256, 1024, 1024);
            rund<64, 64, true, false, EPI_ACT, 2, 4>("forward 2/3", 256, 1024, 1024);
            runb<32, 32, true, true, EPI_DACT, 2, 2>("backward data (shipped)", 256, 1024, 1024);
            runb<32, 32, true, true, EPI_DACT, 5, 2>("backward data", 256, 1024, 1024);
            rund<32, 32, true, true, EPI_DACT, 2, 2>("backward data", 256, 1024, 1024);
            rund<32, 32, true, true, EPI_DACT, 3, 2>("backward data", 256, 1024, 1024);
            rund<64, 64, true, true, EPI_DACT, 2, 4>("backward data", 256, 1024, 1024);
            runb<32, 32, true, false, EPI_ACT, 2, 2>("forward 1 (shipped; K = 784)", 256, 1024, 784);
            runb<32, 32, true, false, EPI_ACT, 5, 2>("forward 1 (K = 784)", 256, 1024, 784);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0 (M = 784)", 784, 1024, 256);
            runb<64, 64, false, false, EPI_SGD, 5, 4>("gradient + update 0 (M = 784)", 784, 1024, 256);
        }
        return 0;
    }
    if (which == 41) {
#ifdef GNN_F32_NO_LDS_WRITES
        printf("---- f32 GEMMs of configs[3] WITHOUT the LDS writes of the main loop (wrong results: an upper bound on what staging by DMA could give)\n");
#else
        printf("---- f32 GEMMs of configs[3] as shipped\n");
#endif
        for (int rep = 0; rep < 2; rep++) {
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 2", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 2", 512, 2048, 1024);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 1", 512, 2048, 2048);
            run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 0", 4096, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 2", 2048, 1024, 512);
        }
        return 0;
    }
    if (which == 42) {
        printf("---- round 4: f32 GEMMs of configs[3], operand tiles by LDS DMA (gemm_f32_dma.h) against gemm_f32_kernel\n");
        runfd<64, 64, true, false, EPI_STORE, 3, 4>("forward 2", 512, 2048, 2048, true);
        runfd<64, 64, true, true, EPI_STORE, 3, 4>("backward data 2", 512, 2048, 1024, true);
        runfd<64, 64, false, false, EPI_STORE, 2, 4>("gradient 2", 2048, 1024, 512, true);
        runfd<128, 128, false, false, EPI_STORE, 3, 2>("gradient 0", 4096, 2048, 512, true);
        runfd<128, 128, true, false, EPI_STORE, 3, 2>("forward, 128 x 128", 4096, 2048, 2048, true);
        runfd<128, 128, true, true, EPI_STORE, 3, 2>("backward, 128 x 128", 4096, 2048, 2048, true);
        for (int rep = 0; rep < 2; rep++) {
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 1", 512, 2048, 4096);
            runfd<64, 64, true, false, EPI_ACT, 3, 4>("forward 1", 512, 2048, 4096);
            runfd<64, 64, true, false, EPI_ACT, 2, 4>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 2", 512, 2048, 2048);
            runfd<64, 64, true, false, EPI_ACT, 3, 4>("forward 2", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 2", 512, 2048, 1024);
            runfd<64, 64, true, true, EPI_DACT, 3, 4>("backward data 2", 512, 2048, 1024);
            runfd<64, 64, true, true, EPI_DACT, 2, 4>("backward data 2", 512, 2048, 1024);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 1", 512, 2048, 2048);
            runfd<64, 64, true, true, EPI_DACT, 3, 4>("backward data 1", 512, 2048, 2048);
            run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 0", 4096, 2048, 512);
            runfd<128, 128, false, false, EPI_SGD, 3, 2>("gradient + update 0", 4096, 2048, 512);
            runfd<128, 128, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runfd<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 0", 4096, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 1", 2048, 2048, 512);
            runfd<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 1", 2048, 2048, 512);
            runfd<64, 64, false, false, EPI_SGD, 3, 4>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 2", 2048, 1024, 512);
            runfd<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 2", 2048, 1024, 512);
            runfd<64, 64, false, false, EPI_SGD, 3, 4>("gradient + update 2", 2048, 1024, 512);
        }
        return 0;
    }
    if (which == 43) {
        printf("---- round 4: DMA form, deeper tiles of K (fewer barriers per K) on the forward and backward-data products of configs[3]\n");
        { // operands that are numbers in BOTH halves of every float
            const size_t nb = ((size_t)4096 * (2048 + 256) + 4096) * 2;
            std::vector<unsigned short> hb(nb);
            for (size_t i = 0; i < nb; i++) { const float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; memcpy(&u, &f, 4); hb[i] = (unsigned short)(u >> 16); }
            CK(hipMemcpy(dA, hb.data(), nb * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice));
        }
        rund<64, 64, true, false, EPI_STORE, 2, 4, 256>("forward 1", 512, 2048, 4096, true);
        rund<64, 64, true, true, EPI_STORE, 2, 4, 256>("backward data 1", 512, 2048, 2048, true);
        rund<64, 64, true, false, EPI_STORE, 3, 4, 64>("forward 1", 512, 2048, 4096, true);
        for (int rep = 0; rep < 2; rep++) {
            rund<64, 64, true, false, EPI_ACT, 3, 4>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 2, 4, 256>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 3, 4, 64>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 3, 2, 64>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 3, 4>("forward 2", 512, 2048, 2048);
            rund<64, 64, true, false, EPI_ACT, 2, 4, 256>("forward 2", 512, 2048, 2048);
            rund<64, 64, true, false, EPI_ACT, 3, 4, 64>("forward 2", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 3, 4>("backward data 1", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 2, 4, 256>("backward data 1", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 3, 4, 64>("backward data 1", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 2, 4>("backward data 2", 512, 2048, 1024);
            rund<64, 64, true, true, EPI_DACT, 2, 4, 256>("backward data 2", 512, 2048, 1024);
            rund<64, 64, true, true, EPI_DACT, 3, 4, 64>("backward data 2", 512, 2048, 1024);
        }
        return 0;
    }
    if (which == 44) {
        printf("---- round 4: DMA form with FOUR images (three tiles in flight per workgroup)\n");
        { const size_t nb = ((size_t)4096 * (2048 + 256) + 4096) * 2;
            std::vector<unsigned short> hb(nb);
            for (size_t i = 0; i < nb; i++) { const float f = (rand() / (float)RAND_MAX - 0.5f) * 0.25f; unsigned u; memcpy(&u, &f, 4); hb[i] = (unsigned short)(u >> 16); }
            CK(hipMemcpy(dA, hb.data(), nb * 2, hipMemcpyHostToDevice));
            CK(hipMemcpy(dB, hb.data(), nb * 2, hipMemcpyHostToDevice)); }
        rund<64, 64, true, false, EPI_STORE, 4, 4>("forward 1", 512, 2048, 4096, true);
        rund<64, 64, true, true, EPI_STORE, 4, 4>("backward data 2", 512, 2048, 1024, true);
        rund<64, 64, true, false, EPI_STORE, 4, 4, 64>("forward 1", 512, 2048, 4096, true);
        for (int rep = 0; rep < 3; rep++) {
            rund<64, 64, true, false, EPI_ACT, 3, 4>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 4, 4>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 4, 4, 64>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 4, 2>("forward 1", 512, 2048, 4096);
            rund<64, 64, true, false, EPI_ACT, 3, 4>("forward 2", 512, 2048, 2048);
            rund<64, 64, true, false, EPI_ACT, 4, 4>("forward 2", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 3, 4>("backward data 1", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 4, 4>("backward data 1", 512, 2048, 2048);
            rund<64, 64, true, true, EPI_DACT, 2, 4>("backward data 2", 512, 2048, 1024);
            rund<64, 64, true, true, EPI_DACT, 4, 4>("backward data 2", 512, 2048, 1024);
            rund<32, 64, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            rund<32, 64, true, false, EPI_STORE, 3, 2>("logits", 512, 1024, 2048);
        }
        return 0;
    }
    if (which == 31) {
        printf("---- round 4: the bf16 gradient + update products (each three times)\n");
        for (int rep = 0; rep < 3; rep++) {
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("cfg5 gradient + update 1", 1024, 1024, 256);
        }
        return 0;
    }
    if (which == 30) {
        printf("---- round 4: bf16 tile shapes on the products of configs[3] that run few or memory-bound workgroups (each twice)\n");
        for (int rep = 0; rep < 2; rep++) {
            runb<64, 64, true, false, EPI_STORE, 2, 4>("logits (shipped)", 512, 1024, 2048);
            runb<64, 32, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            runb<64, 32, true, false, EPI_STORE, 2, 4>("logits", 512, 1024, 2048);
            runb<32, 64, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            runb<32, 32, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 2 (shipped)", 512, 2048, 1024);
            runb<64, 32, true, true, EPI_DACT, 2, 2>("backward data 2", 512, 2048, 1024);
            runb<32, 64, true, true, EPI_DACT, 2, 2>("backward data 2", 512, 2048, 1024);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("backward data 1 (shipped)", 512, 2048, 2048);
            runb<64, 32, true, true, EPI_DACT, 2, 2>("backward data 1", 512, 2048, 2048);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("forward 2 (shipped)", 512, 2048, 2048);
            runb<64, 32, true, false, EPI_ACT, 2, 4>("forward 2", 512, 2048, 2048);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0 (shipped)", 4096, 2048, 512);
            runb<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runb<32, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runb<128, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1 (shipped)", 2048, 2048, 512);
            runb<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2 (shipped)", 2048, 1024, 512);
            runb<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
            runb<32, 32, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
        }
        return 0;
    }
    if (which == 21) {
        printf("---- wave-K kernel: main-loop arguments preloaded (head) against read from the struct (each pair three times)\n");
        for (int rep = 0; rep < 3; rep++) {
            runk<true, false, EPI_ACT, 4, 2, true>("cfg5 forward, head args", 256, 1024, 1024);
            runk<true, false, EPI_ACT, 4, 2, false>("cfg5 forward, struct", 256, 1024, 1024);
            runk<true, true, EPI_DACT, 4, 2, true>("cfg5 backward data, head args", 256, 1024, 1024);
            runk<true, true, EPI_DACT, 4, 2, false>("cfg5 backward data, struct", 256, 1024, 1024);
        }
        return 0;
    }
    if (which == 20) {
        printf("---- the production kernels of configs[3] / [4] with %d extra elements per row (each twice)\n", g_pad);
        for (int rep = 0; rep < 2; rep++) {
            runk<true, false, EPI_ACT, 4, 2>("cfg5 forward (wave-K)", 256, 1024, 1024);
            runk<true, true, EPI_DACT, 4, 2>("cfg5 backward data (wave-K)", 256, 1024, 1024);
            run<64, 64, true, false, EPI_ACT, 4, 1>("cfg4 forward 1", 512, 2048, 4096);
            run<64, 64, true, true, EPI_DACT, 4, 1>("cfg4 backward data 1", 512, 2048, 2048);
            run<128, 128, false, false, EPI_SGD, 2, 1>("cfg4 gradient + update 0", 4096, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("cfg4 gradient + update 1", 2048, 2048, 512);
            runb<64, 64, true, false, EPI_ACT, 2, 4>("cfg4 bf16 forward 1", 512, 2048, 4096);
            runb<64, 64, true, true, EPI_DACT, 2, 2>("cfg4 bf16 backward data 1", 512, 2048, 2048);
            runb<64, 64, false, false, EPI_SGD, 2, 2>("cfg4 bf16 gradient + update 0", 4096, 2048, 512);
        }
        return 0;
    }
    if (which == 6) { clock_under_load(); return 0; }
    if (which == 2) {
        printf("---- one register stage against two, same k-tile depth (each pair twice)\n");
        for (int rep = 0; rep < 2; rep++) {
            run<64, 32, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
            run<64, 32, true, false, EPI_ACT, 2, 1>("forward 1", 512, 2048, 4096);
            run<64, 32, true, false, EPI_ACT, 2, 2>("forward 2", 512, 2048, 2048);
            run<64, 32, true, false, EPI_ACT, 2, 1>("forward 2", 512, 2048, 2048);
            run<64, 32, true, true, EPI_DACT, 2, 2>("backward data 1", 512, 2048, 2048);
            run<64, 32, true, true, EPI_DACT, 2, 1>("backward data 1", 512, 2048, 2048);
            run<64, 32, true, true, EPI_DACT, 2, 2>("backward data 2", 512, 2048, 1024);
            run<64, 32, true, true, EPI_DACT, 2, 1>("backward data 2", 512, 2048, 1024);
            run<32, 32, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
            run<32, 32, true, false, EPI_STORE, 2, 1>("logits", 512, 1024, 2048);
            run<32, 32, true, false, EPI_STORE, 2, 1, 128>("logits", 512, 1024, 2048);
            run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
            run<64, 64, false, false, EPI_SGD, 2, 1>("gradient + update 2", 2048, 1024, 512);
            run<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update (configs[4])", 1024, 1024, 256);
            run<64, 32, false, false, EPI_SGD, 2, 1>("gradient + update (configs[4])", 1024, 1024, 256);
        }
        return 0;
    }
    if (which == 1) {
        printf("---- tile shapes with one register stage (each twice)\n");
        for (int rep = 0; rep < 2; rep++) {
            run<64, 32, true, false, EPI_ACT, 2, 1>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 2, 1>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 1", 512, 2048, 4096);
            run<128, 64, true, false, EPI_ACT, 4, 1>("forward 1", 512, 2048, 4096);
            run<64, 32, true, false, EPI_ACT, 2, 1>("forward 2", 512, 2048, 2048);
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 2", 512, 2048, 2048);
            run<64, 32, true, true, EPI_DACT, 2, 1>("backward data 1", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 1", 512, 2048, 2048);
            run<64, 32, true, true, EPI_DACT, 2, 1>("backward data 2", 512, 2048, 1024);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 2", 512, 2048, 1024);
            run<32, 32, true, false, EPI_STORE, 2, 1>("logits", 512, 1024, 2048);
            run<64, 32, true, false, EPI_STORE, 2, 1>("logits", 512, 1024, 2048);
            run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 0", 4096, 2048, 512);
            run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 2, 1>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 2, 1>("gradient + update 2", 2048, 1024, 512);
            run<64, 32, false, false, EPI_SGD, 2, 1>("gradient + update 2", 2048, 1024, 512);
            runk<true, false, EPI_STORE, 4, 2>("logits (wave-K)", 512, 1024, 2048);
        }
        return 0;
    }
    if (which == 11) {
        printf("---- unguarded loads: one register stage against two (each twice)\n");
        for (int rep = 0; rep < 2; rep++) {
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 4, 2>("forward 1", 512, 2048, 4096);
            run<64, 64, true, false, EPI_ACT, 4, 1>("forward 2", 512, 2048, 2048);
            run<64, 64, true, false, EPI_ACT, 4, 2>("forward 2", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 1", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 2>("backward data 1", 512, 2048, 2048);
            run<64, 64, true, true, EPI_DACT, 4, 1>("backward data 2", 512, 2048, 1024);
            run<64, 64, true, true, EPI_DACT, 4, 2>("backward data 2", 512, 2048, 1024);
            run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 0", 4096, 2048, 512);
            run<128, 128, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 2, 1>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 1", 2048, 2048, 512);
            run<64, 64, false, false, EPI_SGD, 2, 1>("gradient + update 2", 2048, 1024, 512);
            run<64, 64, false, false, EPI_SGD, 4, 1>("gradient + update 2", 2048, 1024, 512);
        }
        return 0;
    }
    if (which == 3) {
        printf("---- k-tile depth\n");
        run<64, 32, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 1, 64>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 1, 128>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 2, 128>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 2, 1, 128>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 4, 1, 128>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 4, 2, 128>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 1, 128>("forward 2", 512, 2048, 2048);
        run<64, 32, true, true, EPI_DACT, 2, 2>("backward data 1", 512, 2048, 2048);
        run<64, 32, true, true, EPI_DACT, 2, 1, 128>("backward data 1", 512, 2048, 2048);
        run<64, 32, true, true, EPI_DACT, 2, 1, 128>("backward data 2", 512, 2048, 1024);
        run<32, 32, true, false, EPI_STORE, 2, 1, 128>("logits", 512, 1024, 2048);
        run<32, 32, true, false, EPI_STORE, 2, 1, 256>("logits", 512, 1024, 2048);
        run<64, 64, false, false, EPI_SGD, 2, 1, 128>("gradient + update 2", 2048, 1024, 512);
        run<64, 64, false, false, EPI_SGD, 2, 1, 64>("gradient + update 2", 2048, 1024, 512);
        run<128, 128, false, false, EPI_SGD, 2, 1, 64>("gradient + update 0", 4096, 2048, 512);
        return 0;
    }
    if (which == 8) {
        check<true, false, EPI_ACT, 4, 2>(256, 2048, 4096); check<true, false, EPI_ACT, 4, 2>(256, 2048, 2048);
        check<true, false, EPI_STORE, 4, 2>(256, 1024, 2048); check<true, false, EPI_STORE, 4, 2>(512, 1024, 2048);
        check<true, true, EPI_DACT, 4, 2>(256, 2048, 1024); check<true, true, EPI_DACT, 4, 2>(256, 2048, 2048);
        check<true, false, EPI_ACT, 4, 2>(256, 1024, 784); check<true, true, EPI_DACT, 4, 2>(256, 1024, 1024);
        check<false, false, EPI_STORE, 4, 2>(1024, 1024, 256); check<true, false, EPI_ACT, 4, 2>(64, 2048, 4096);
        return 0;
    }
    if (which == 9) {
        check<true, false, EPI_ACT, 4, 4>(256, 1024, 1024); check<true, false, EPI_ACT, 8, 4>(256, 1024, 784); check<true, true, EPI_ACT, 4, 4>(256, 1024, 1024);
        check<true, true, EPI_ACT, 8, 2>(256, 1024, 784); check<false, false, EPI_ACT, 4, 4>(1024, 1024, 256); check<false, false, EPI_ACT, 8, 4>(800, 1024, 272);
        printf("---- wave-K kernel\n");
        runk<true, false, EPI_ACT, 4, 2>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 4, 4>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 4, 6>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 8, 2>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 8, 4>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 4, 3>("forward", 256, 1024, 1024);
        runk<true, false, EPI_ACT, 4, 2>("forward (first layer)", 256, 1024, 784);
        runk<true, false, EPI_ACT, 4, 4>("forward (first layer)", 256, 1024, 784);
        runk<true, false, EPI_ACT, 8, 4>("forward (first layer)", 256, 1024, 784);
        runk<true, true, EPI_DACT, 4, 2>("backward data", 256, 1024, 1024);
        runk<true, true, EPI_DACT, 4, 4>("backward data", 256, 1024, 1024);
        runk<true, true, EPI_DACT, 8, 2>("backward data", 256, 1024, 1024);
        runk<true, true, EPI_DACT, 8, 4>("backward data", 256, 1024, 1024);
        runk<false, false, EPI_SGD, 4, 2>("gradient + update", 1024, 1024, 256);
        runk<false, false, EPI_SGD, 8, 1>("gradient + update", 1024, 1024, 256);
        runk<true, false, EPI_STORE, 4, 4>("logits (configs[3])", 512, 1024, 2048);
        runk<true, false, EPI_STORE, 8, 4>("logits (configs[3])", 512, 1024, 2048);
        runk<true, true, EPI_DACT, 4, 4>("backward data 2 (configs[3])", 512, 2048, 1024);
        runk<true, false, EPI_ACT, 4, 4>("forward 2 (configs[3])", 512, 2048, 2048);
    }
    if (which == 0 || which == 5) {
        printf("---- configs[4]: 784-1024-1024-1024-10, 256 rows\n");
        run<32, 32, true, false, EPI_ACT, 2, 2>("forward", 256, 1024, 1024);
        run<32, 32, true, false, EPI_ACT, 2, 4>("forward", 256, 1024, 1024);
        run<32, 32, true, false, EPI_ACT, 2, 6>("forward", 256, 1024, 1024);
        run<32, 32, true, false, EPI_ACT, 2, 8>("forward", 256, 1024, 1024);
        run<64, 32, true, false, EPI_ACT, 2, 2>("forward", 256, 1024, 1024);
        run<64, 32, true, false, EPI_ACT, 2, 4>("forward", 256, 1024, 1024);
        run<32, 32, true, false, EPI_ACT, 2, 2>("forward (first layer)", 256, 1024, 784);
        run<32, 32, true, false, EPI_ACT, 2, 6>("forward (first layer)", 256, 1024, 784);
        run<32, 32, true, false, EPI_ACT, 2, 8>("forward (first layer)", 256, 1024, 784);
        run<32, 32, true, true, EPI_DACT, 2, 2>("backward data", 256, 1024, 1024);
        run<32, 32, true, true, EPI_DACT, 2, 4>("backward data", 256, 1024, 1024);
        run<32, 32, true, true, EPI_DACT, 2, 6>("backward data", 256, 1024, 1024);
        run<32, 32, true, true, EPI_DACT, 2, 8>("backward data", 256, 1024, 1024);
        run<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update", 1024, 1024, 256);
        run<64, 32, false, false, EPI_SGD, 2, 4>("gradient + update", 1024, 1024, 256);
        run<32, 32, false, false, EPI_SGD, 2, 2>("gradient + update", 1024, 1024, 256);
        run<32, 32, false, false, EPI_SGD, 2, 4>("gradient + update", 1024, 1024, 256);
        run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update", 1024, 1024, 256);
        run<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update (first layer)", 784, 1024, 256);
        run<64, 32, false, false, EPI_SGD, 2, 4>("gradient + update (first layer)", 784, 1024, 256);
        run<32, 32, false, false, EPI_SGD, 2, 4>("gradient + update (first layer)", 784, 1024, 256);
    }
    if (which == 0 || which == 4) {
        printf("---- configs[3]: 4096-2048-2048-1024, 512 rows\n");
        run<64, 32, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 3>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 4>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 2, 2>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 2, 4>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 4, 2>("forward 1", 512, 2048, 4096);
        run<64, 64, true, false, EPI_ACT, 4, 4>("forward 1", 512, 2048, 4096);
        run<64, 32, true, false, EPI_ACT, 2, 2>("forward 2", 512, 2048, 2048);
        run<64, 32, true, false, EPI_ACT, 2, 4>("forward 2", 512, 2048, 2048);
        run<32, 32, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
        run<32, 32, true, false, EPI_STORE, 2, 4>("logits", 512, 1024, 2048);
        run<32, 32, true, false, EPI_STORE, 2, 8>("logits", 512, 1024, 2048);
        run<64, 32, true, false, EPI_STORE, 2, 2>("logits", 512, 1024, 2048);
        run<64, 32, true, false, EPI_STORE, 2, 4>("logits", 512, 1024, 2048);
        run<64, 32, true, true, EPI_DACT, 2, 2>("backward data 2", 512, 2048, 1024);
        run<64, 32, true, true, EPI_DACT, 2, 4>("backward data 2", 512, 2048, 1024);
        run<64, 32, true, true, EPI_DACT, 2, 2>("backward data 1", 512, 2048, 2048);
        run<64, 32, true, true, EPI_DACT, 2, 4>("backward data 1", 512, 2048, 2048);
        run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 0", 4096, 2048, 512);
        run<128, 128, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
        run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 0", 4096, 2048, 512);
        run<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 0", 4096, 2048, 512);
        run<128, 128, false, false, EPI_SGD, 2, 1>("gradient + update 1", 2048, 2048, 512);
        run<128, 128, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
        run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 1", 2048, 2048, 512);
        run<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 1", 2048, 2048, 512);
        run<64, 64, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
        run<64, 64, false, false, EPI_SGD, 2, 4>("gradient + update 2", 2048, 1024, 512);
        run<64, 32, false, false, EPI_SGD, 2, 2>("gradient + update 2", 2048, 1024, 512);
        run<64, 32, false, false, EPI_SGD, 2, 4>("gradient + update 2", 2048, 1024, 512);
    }
    return 0;
}
