// perf_harness.hip -- development harness (not shipped): runs the fused step kernels on the
// 784-300-100-10 / B=128 shapes with random data, times them with HIP events and prints the
// middle kernel's phase stamps (STAMP build).
#include "../graph-neural-net_amd/csrc/middle4_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
using namespace gnn;
using SS = StaticShape<784, 300, 100, 10>;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)


// block-to-block spans on the device-wide 100 MHz real-time counter (slots 4 = start, 5 = end)
static void spans(const std::vector<unsigned long long> &hs, int blocks, const char *name) {
    unsigned long long t0 = ~0ull, t1 = 0, s1 = 0, e0 = ~0ull;
    for (int w = 0; w < blocks; w++) {
        if (!hs[w * 8 + 4] || !hs[w * 8 + 5]) continue;
        t0 = std::min(t0, hs[w * 8 + 4]); t1 = std::max(t1, hs[w * 8 + 5]); s1 = std::max(s1, hs[w * 8 + 4]); e0 = std::min(e0, hs[w * 8 + 5]);
    }
    printf("%s: first block start -> last block end %.2f us; last block starts %.2f us after the first; first block ends after %.2f us\n", name,
           (t1 - t0) / 100.0, (s1 - t0) / 100.0, (e0 - t0) / 100.0);
}

int main(int argc, char **argv) {
    const int L = 4, dims[4] = {784, 300, 100, 10};
    int B = argc > 1 ? atoi(argv[1]) : 128;
    int ld[4]; for (int i = 0; i < L; i++) ld[i] = pad_up(dims[i]);
    const int Bp = pad_up(B);
    size_t woff[3], np = 0; for (int l = 0; l < 3; l++) { woff[l] = np; np += (size_t)ld[l] * ld[l + 1]; }
    float *W, *V, *G, *act[4], *delta[4], *Y, *lossv; int32_t *labels; unsigned long long *stamps;
    CK(hipMalloc(&W, np * 4)); CK(hipMalloc(&V, np * 4)); CK(hipMalloc(&G, np * 4));
    std::vector<float> hw(np, 0.f);
    for (int l = 0; l < 3; l++) for (int i = 0; i < dims[l]; i++) for (int j = 0; j < dims[l + 1]; j++)
        hw[woff[l] + (size_t)i * ld[l + 1] + j] = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    CK(hipMemcpy(W, hw.data(), np * 4, hipMemcpyHostToDevice)); CK(hipMemset(V, 0, np * 4));
    for (int l = 0; l < L; l++) { CK(hipMalloc(&act[l], (size_t)Bp * ld[l] * 4)); CK(hipMemset(act[l], 0, (size_t)Bp * ld[l] * 4));
                                  CK(hipMalloc(&delta[l], (size_t)Bp * ld[l] * 4)); CK(hipMemset(delta[l], 0, (size_t)Bp * ld[l] * 4)); }
    std::vector<float> hx((size_t)Bp * ld[0], 0.f), hy((size_t)Bp * ld[3], 0.f);
    for (int b = 0; b < B; b++) { for (int i = 0; i < dims[0]; i++) hx[(size_t)b * ld[0] + i] = rand() / (float)RAND_MAX; hy[(size_t)b * ld[3] + rand() % 10] = 1.f; }
    CK(hipMemcpy(act[0], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&Y, hy.size() * 4)); CK(hipMemcpy(Y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&lossv, Bp * 4)); CK(hipMalloc(&labels, Bp * 4)); CK(hipMalloc(&stamps, 4096 * 8)); CK(hipMemset(stamps, 0, 4096 * 8));

    FwdFirstParams f{}; f.A = act[0]; f.lda = ld[0]; f.W = W; f.ldw = ld[1]; f.C = act[1]; f.ldc = ld[1];
    f.M = Bp; f.N = ld[1]; f.K = ld[0]; f.m_true = B; f.n_true = dims[1]; f.act = 0; f.apply_act = 1; f.tiling = make_xcd_tiling(f.M / 16, f.N / 16);
    GradParams g{}; g.n_layers = 3; int tiles = 0;
    for (int l = 0; l < 3; l++) { GradLayer &gl = g.layer[l]; gl.A = act[l]; gl.lda = ld[l]; gl.D = delta[l + 1]; gl.ldd = ld[l + 1];
        gl.W = W + woff[l]; gl.V = V + woff[l]; gl.G = G + woff[l]; gl.M = ld[l]; gl.N = ld[l + 1]; gl.tiling = make_xcd_tiling((gl.M + 31) / 32, (gl.N + 31) / 32); gl.block_begin = tiles; tiles += gl.tiling.blocks(); }
    g.K = Bp; g.step_over_b = 0.0125f / B; g.momentum = 0.9f;

    // middle4 plan
    Mid4Params m4{}; m4.plan = make_mid4_plan(dims, L); size_t lds4 = (size_t)m4.plan.lds_floats * 4;
    {
        printf("middle4: LDS %zu bytes, ks_fwd = %d %d, ks_bwd = %d %d\n", lds4, m4.plan.ks_fwd[2], m4.plan.ks_fwd[3], m4.plan.ks_bwd[2], m4.plan.ks_bwd[1]);
        for (int l = 1; l < 3; l++) { m4.W[l] = W + woff[l]; m4.act[l] = act[l]; }
        for (int l = 1; l < L; l++) m4.delta[l] = delta[l];
        m4.Y = Y; m4.ldy = ld[3]; m4.loss = getenv("HARNESS_AUX") ? lossv : nullptr; m4.label = getenv("HARNESS_AUX") ? labels : nullptr; m4.B = B; m4.stamps = stamps; m4.inner_act = 0;
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<RuntimeShape<4>, 0, 0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<RuntimeShape<4>, 0, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<SS, 0, 0, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<SS, 0, 0, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<SS, 0, 0, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipFuncSetAttribute((const void *)&middle4_kernel<SS, 0, 0, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        // first-layer K slabs (two-launch path): 13 slabs of [Bp][304]
        const int ns = (ld[0] + 63) / 64;
        float *slabs; CK(hipMalloc(&slabs, (size_t)ns * Bp * ld[1] * 4));
        std::vector<float> hsl((size_t)ns * Bp * ld[1]);
        for (auto &x : hsl) x = (rand() / (float)RAND_MAX - 0.5f) * 0.3f;
        CK(hipMemcpy(slabs, hsl.data(), hsl.size() * 4, hipMemcpyHostToDevice));
        m4.slabs = slabs; m4.slab_rows = Bp; m4.n_slabs = ns;
    }
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, int n, auto fn) {
        for (int i = 0; i < 20; i++) fn();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; i++) fn();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %8.2f us per call\n", name, ms * 1000.f / n);
    };
    auto k_first = [&]() { hipLaunchKernelGGL((fwd_first_kernel<8>), dim3(f.tiling.blocks()), dim3(512), 0, s, f); };
    auto k_first16 = [&]() { hipLaunchKernelGGL((fwd_first_kernel<16>), dim3(f.tiling.blocks()), dim3(1024), 0, s, f); };
    auto k_grad = [&]() { hipLaunchKernelGGL((grad_update_kernel<true>), dim3(tiles), dim3(GRAD_THREADS), 0, s, g); };
    auto k_grad_nf = [&]() { hipLaunchKernelGGL((grad_update_kernel<false>), dim3(tiles), dim3(GRAD_THREADS), 0, s, g); };
    time_it("fwd_first<8>", 500, k_first);
    time_it("fwd_first<16>", 500, k_first16);
    time_it("fwd_first<4,ACT=leaky>", 500, [&]() { hipLaunchKernelGGL((fwd_first_kernel<4, false, 0>), dim3(f.tiling.blocks()), dim3(256), 0, s, f); });
    time_it("fwd_first<3,ACT=leaky>", 500, [&]() { hipLaunchKernelGGL((fwd_first_kernel<3, false, 0>), dim3(f.tiling.blocks()), dim3(192), 0, s, f); });
    time_it("fwd_first<2,ACT=leaky>", 500, [&]() { hipLaunchKernelGGL((fwd_first_kernel<2, false, 0>), dim3(f.tiling.blocks()), dim3(128), 0, s, f); });
    time_it("fwd_first<8,ACT=leaky>", 500, [&]() { hipLaunchKernelGGL((fwd_first_kernel<8, false, 0>), dim3(f.tiling.blocks()), dim3(512), 0, s, f); });
    auto k_mid4r = [&]() { hipLaunchKernelGGL((middle4_kernel<RuntimeShape<4>, 0, 0, true, false>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4); };
    auto k_mid4 = [&]() { hipLaunchKernelGGL((middle4_kernel<SS, 0, 0, true, false>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4); };
    time_it("middle4 runtime shape", 500, k_mid4r);
    time_it("middle4 static shape", 500, k_mid4);
    time_it("step with middle4 static", 500, [&]() { k_first(); k_mid4(); k_grad(); });
    time_it("middle4 static shape, K slabs", 500, [&]() { hipLaunchKernelGGL((middle4_kernel<SS, 0, 0, true, false, true>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4); });
    time_it("grad_update<fused>", 500, k_grad);
    time_it("grad_update<store G>", 500, k_grad_nf);
    {
        f.stamps = stamps; g.stamps = stamps;
        for (int rep = 0; rep < 2; rep++) {
        if (rep) hipLaunchKernelGGL((fwd_first_kernel<8, true, 0>), dim3(f.tiling.blocks()), dim3(512), 0, s, f);
        else hipLaunchKernelGGL((fwd_first_kernel<8, true>), dim3(f.tiling.blocks()), dim3(512), 0, s, f);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hs(f.tiling.blocks() * 8);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        printf("fwd_first stamps (%s):", rep ? "ACT templated" : "ACT runtime");
        for (int w : {0, 1, 80, 150}) printf(" wg%d: loads+mfma=%llu red=%llu epi=%llu |", w, hs[w*8+1]-hs[w*8], hs[w*8+2]-hs[w*8+1], hs[w*8+3]-hs[w*8+2]);
        printf("\n");
        }
        CK(hipMemsetAsync(stamps, 0, 4096 * 8, s));
        hipLaunchKernelGGL((fwd_first_kernel<8, true>), dim3(f.tiling.blocks()), dim3(512), 0, s, f);
        CK(hipStreamSynchronize(s));
        std::vector<unsigned long long> hs(f.tiling.blocks() * 8);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        spans(hs, f.tiling.blocks(), "fwd_first");
        printf("fwd_first stamps:");
        for (int w : {0, 1, 75, 150}) printf(" wg%d: loads+mfma=%llu red=%llu epi=%llu |", w, hs[w*8+1]-hs[w*8], hs[w*8+2]-hs[w*8+1], hs[w*8+3]-hs[w*8+2]);
        printf("\n");
        CK(hipMemsetAsync(stamps, 0, 4096 * 8, s));
        hipLaunchKernelGGL((grad_update_kernel<true, true>), dim3(tiles), dim3(GRAD_THREADS), 0, s, g);
        CK(hipStreamSynchronize(s));
        hs.resize(tiles * 8);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        t0 = ~0ull; t1 = 0;
        spans(hs, tiles, "grad_update");
        printf("grad_update stamps:");
        for (int w : {0, 1, 128, 249, 281, 329}) printf(" wg%d: load=%llu mfma=%llu epi=%llu |", w, hs[w*8+1]-hs[w*8], hs[w*8+2]-hs[w*8+1], hs[w*8+3]-hs[w*8+2]);
        printf("\n");
    }
    {
        for (int variant = 0; variant < 3; variant++) {
        if (variant == 2) hipLaunchKernelGGL((middle4_kernel<SS, 0, 0, true, true, true>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4);
        else if (variant) hipLaunchKernelGGL((middle4_kernel<SS, 0, 0, true, true>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4);
        else hipLaunchKernelGGL((middle4_kernel<RuntimeShape<4>, 0, 0, true, true>), dim3((B + 3) / 4), dim3(1024), lds4, s, m4);
        CK(hipStreamSynchronize(s));
        printf("%s:\n", variant == 2 ? "STATIC shape, K slabs" : variant ? "STATIC shape" : "RUNTIME shape");
        std::vector<unsigned long long> hs(2 * 32 * 16);
        CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
        for (int wg : {0, 31, 32, 63}) printf("middle4 stamps wg %d%s: load+stage=%llu fwdL2=%llu fwdL3=%llu output=%llu bwd2=%llu bwd1=%llu total=%llu\n", wg % 32, wg >= 32 ? " (2nd pass)" : "",
            hs[wg*16+1]-hs[wg*16], hs[wg*16+6]-hs[wg*16+1], hs[wg*16+7]-hs[wg*16+6], hs[wg*16+3]-hs[wg*16+2], hs[wg*16+12]-hs[wg*16+3], hs[wg*16+11]-hs[wg*16+12], hs[wg*16+4]-hs[wg*16]);
        }
    }
    return 0;
}
