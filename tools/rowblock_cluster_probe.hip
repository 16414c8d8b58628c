// rowblock_cluster_probe.hip -- development harness (not shipped): rowblock_cluster_kernel (a row block shared by four
// workgroups, csrc/rowblock_cluster_kernel.h) against rowblock_kernel on 784-300-100-10 with random slabs: outputs, time per
// call (HIP events over 500 launches), in-kernel phase stamps.
#include "rowblock_cluster_kernel.h"
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace gnn;
#define RBHEAD(r) (r).slabs, (r).W[1], (r).W[2], (r).row_idx, (r).Y, (r).copy_idx, (r).B, (r).slab_rows, (r).ldy
using RS4 = RbStaticShape<784, 300, 100, 10>;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int L = 4, dims[4] = {784, 300, 100, 10};
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    int ld[4]; for (int i = 0; i < L; i++) ld[i] = pad_up(dims[i]);
    const int Bp = pad_up(B);
    size_t woff[3], np = 0; for (int l = 0; l < 3; l++) { woff[l] = np; np += (size_t)ld[l] * ld[l + 1]; }
    float *W, *Y, *slabs; unsigned long long *stamps;
    CK(hipMalloc(&W, np * 4));
    std::vector<float> hw(np, 0.f);
    for (int l = 0; l < 3; l++) for (int i = 0; i < dims[l]; i++) for (int j = 0; j < dims[l + 1]; j++)
        hw[woff[l] + (size_t)i * ld[l + 1] + j] = (rand() / (float)RAND_MAX - 0.5f) * 0.4f;
    CK(hipMemcpy(W, hw.data(), np * 4, hipMemcpyHostToDevice));
    std::vector<float> hy((size_t)Bp * ld[3], 0.f);
    for (int b = 0; b < B; b++) hy[(size_t)b * ld[3] + rand() % 10] = 1.f;
    CK(hipMalloc(&Y, hy.size() * 4)); CK(hipMemcpy(Y, hy.data(), hy.size() * 4, hipMemcpyHostToDevice));
    const int ns = (ld[0] + 63) / 64;
    std::vector<float> hsl((size_t)ns * Bp * ld[1], 0.f);
    for (int s = 0; s < ns; s++) for (int b = 0; b < B; b++) for (int j = 0; j < dims[1]; j++) hsl[((size_t)s * Bp + b) * ld[1] + j] = (rand() / (float)RAND_MAX - 0.45f) * 0.3f;
    CK(hipMalloc(&slabs, hsl.size() * 4)); CK(hipMemcpy(slabs, hsl.data(), hsl.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&stamps, 4096 * 16 * 8)); CK(hipMemset(stamps, 0, 4096 * 16 * 8));
    float *act[2][4], *delta[2][4], *prob[2], *loss[2];
    for (int v = 0; v < 2; v++) {
        for (int l = 1; l < L; l++) {
            CK(hipMalloc(&act[v][l], (size_t)Bp * ld[l] * 4)); CK(hipMemset(act[v][l], 0xff, (size_t)Bp * ld[l] * 4));
            CK(hipMalloc(&delta[v][l], (size_t)Bp * ld[l] * 4)); CK(hipMemset(delta[v][l], 0xff, (size_t)Bp * ld[l] * 4));
        }
        CK(hipMalloc(&prob[v], (size_t)Bp * 16 * 4)); CK(hipMalloc(&loss[v], (size_t)Bp * 4));
    }
    RbParams rb{}; rb.plan = make_rb_plan(dims, L); const size_t ldsr = (size_t)rb.plan.lds_floats * 4;
    for (int l = 1; l < 3; l++) { rb.W[l] = W + woff[l]; rb.act[l] = act[0][l]; }
    for (int l = 1; l < L; l++) rb.delta[l] = delta[0][l];
    rb.Y = Y; rb.ldy = ld[3]; rb.B = B; rb.inner_act = 0; rb.slabs = slabs; rb.slab_rows = Bp; rb.stamps = stamps; rb.prob = prob[0]; rb.loss = loss[0];
    auto k_old = rowblock_kernel<RS4, 0, 0, false>;
    CK(hipFuncSetAttribute((const void *)k_old, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsr));
    RbcParams q{};
    q.W1 = W + woff[1]; q.W2 = W + woff[2]; q.act1 = act[1][1]; q.act2 = act[1][2];
    q.delta1 = delta[1][1]; q.delta2 = delta[1][2]; q.delta3 = delta[1][3];
    q.Y = Y; q.ldy = ld[3]; q.B = B; q.slabs = slabs; q.slab_rows = Bp; q.stamps = stamps; q.prob = prob[1]; q.loss = loss[1];
    const int clusters = Bp / 4;
    CK(hipMalloc(&q.xch, (size_t)clusters * 16 * 32 * 8)); CK(hipMemset(q.xch, 0, (size_t)clusters * 16 * 32 * 8));
    CK(hipMalloc(&q.sync, sizeof(XchSync))); CK(hipMemset(q.sync, 0, sizeof(XchSync)));
    auto k_new = rowblock_cluster_kernel<784, 300, 100, 10, 0, 0, false>;
    auto k_stamp = rowblock_cluster_kernel<784, 300, 100, 10, 0, 0, true>;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const dim3 grid_old((B + 3) / 4), grid_new(rbc_grid(B));
    printf("B = %d: rowblock_kernel %d workgroups x %d threads (LDS %zu B); cluster kernel %d workgroups x %d threads\n", B, grid_old.x, RB_NT, ldsr, grid_new.x, RBC_NT);
    hipLaunchKernelGGL(k_old, grid_old, dim3(RB_NT), ldsr, s, RBHEAD(rb), rb);
    hipLaunchKernelGGL(k_new, grid_new, dim3(RBC_NT), 0, s, q);
    CK(hipStreamSynchronize(s));
    auto cmp = [&](const char *name, const float *a_d, const float *b_d, size_t n) {
        std::vector<float> a(n), b(n);
        CK(hipMemcpy(a.data(), a_d, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), b_d, n * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; int bad = 0;
        for (size_t i = 0; i < n; i++) { if (!(b[i] == b[i])) bad++; md = std::max(md, (double)fabsf(a[i] - b[i])); mx = std::max(mx, (double)fabsf(a[i])); }
        printf("  %-8s max|d| = %.3g (scale %.3g, nan %d)\n", name, md, mx, bad);
    };
    cmp("act1", act[0][1], act[1][1], (size_t)Bp * ld[1]); cmp("act2", act[0][2], act[1][2], (size_t)Bp * ld[2]);
    cmp("delta3", delta[0][3], delta[1][3], (size_t)Bp * ld[3]); cmp("delta2", delta[0][2], delta[1][2], (size_t)Bp * ld[2]);
    cmp("delta1", delta[0][1], delta[1][1], (size_t)Bp * ld[1]); cmp("prob", prob[0], prob[1], (size_t)Bp * 16); cmp("loss", loss[0], loss[1], (size_t)Bp);
    { // which of the two is right: row 0's probabilities from A_2 (identical in both) and W_2 on the host
        std::vector<float> a2((size_t)Bp * ld[2]), p0((size_t)Bp * 16), p1((size_t)Bp * 16);
        CK(hipMemcpy(a2.data(), act[1][2], a2.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(p0.data(), prob[0], p0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), prob[1], p1.size() * 4, hipMemcpyDeviceToHost));
        for (int row : {0, 1, 5}) {
            double z[10], mx = -1e30, ssum = 0;
            for (int c = 0; c < 10; c++) { z[c] = 0; for (int k = 0; k < 100; k++) z[c] += (double)a2[(size_t)row * ld[2] + k] * hw[woff[2] + (size_t)k * ld[3] + c]; mx = std::max(mx, z[c]); }
            for (int c = 0; c < 10; c++) ssum += exp(z[c] - mx);
            printf("  row %d host / rowblock / cluster:", row);
            for (int c = 0; c < 4; c++) printf("  %.4f %.4f %.4f |", exp(z[c] - mx) / ssum, p0[(size_t)row * 16 + c], p1[(size_t)row * 16 + c]);
            printf("\n");
        }
    }
    { // the A_2 image slice-1 workgroups multiplied, against A_2 in memory
        q.last_act = 99;
        hipLaunchKernelGGL(k_new, grid_new, dim3(RBC_NT), 0, s, q);
        CK(hipStreamSynchronize(s));
        q.last_act = 0;
        std::vector<float> a2((size_t)Bp * ld[2]), im((size_t)Bp * ld[2]);
        CK(hipMemcpy(a2.data(), act[1][2], a2.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(im.data(), delta[1][2], im.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int row = 0; row < B; row++) for (int n = 0; n < ld[2]; n++) if (a2[(size_t)row * ld[2] + n] != im[(size_t)row * ld[2] + n]) { if (bad < 12) printf("  image row %d col %d: %g, memory %g\n", row, n, im[(size_t)row * ld[2] + n], a2[(size_t)row * ld[2] + n]); bad++; }
        printf("  A_2 image of the slice-1 workgroups: %d of %d elements differ from A_2 in memory\n", bad, B * ld[2]);
        hipLaunchKernelGGL(k_new, grid_new, dim3(RBC_NT), 0, s, q); CK(hipStreamSynchronize(s));
    }
    XchSync hs; CK(hipMemcpy(&hs, q.sync, sizeof(hs), hipMemcpyDeviceToHost));
    printf("  sync: launch %u done %u error %u\n", hs.launch, hs.done, hs.error);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, int n, auto fn) {
        for (int i = 0; i < 20; i++) fn();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; i++) fn();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.2f us per call\n", name, ms * 1000.f / n);
    };
    for (int rep = 0; rep < 3; rep++) {
        time_it("rowblock_kernel<static> (32 workgroups)", 500, [&]() { hipLaunchKernelGGL(k_old, grid_old, dim3(RB_NT), ldsr, s, RBHEAD(rb), rb); });
        time_it("rowblock_cluster_kernel (4 per row block)", 500, [&]() { hipLaunchKernelGGL(k_new, grid_new, dim3(RBC_NT), 0, s, q); });
    }
    CK(hipMemcpy(&hs, q.sync, sizeof(hs), hipMemcpyDeviceToHost));
    printf("  sync after the timed launches: launch %u done %u error %u\n", hs.launch, hs.done, hs.error);
    // the results after 1 500 launches: still the same (the epoch tags keep launches apart)
    cmp("delta1", delta[0][1], delta[1][1], (size_t)Bp * ld[1]);
    CK(hipMemsetAsync(stamps, 0, 4096 * 16 * 8, s));
    hipLaunchKernelGGL(k_stamp, grid_new, dim3(RBC_NT), 0, s, q);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> st((size_t)grid_new.x * 16);
    CK(hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost));
    const char *names[8] = {"start", "loads issued", "A_1 barrier", "partial slices", "slice published", "peers' slices in", "tail done", "backward done"};
    for (int w : {0, 8, 16, 24, 33, 127}) {
        if (w >= (int)grid_new.x) continue;
        const unsigned long long *z = &st[(size_t)w * 16];
        printf("wg %3d (cluster %d slice %d):", w, (w >> 5) * 8 + (w & 7), (w >> 3) & 3);
        for (int k = 1; k < 8; k++) printf(" %s +%llu |", names[k], z[k] - z[0]);
        printf("\n");
    }
    unsigned long long first = ~0ull, last = 0;
    for (unsigned w = 0; w < grid_new.x; w++) { if (!st[(size_t)w * 16]) continue; first = std::min(first, st[(size_t)w * 16]); last = std::max(last, st[(size_t)w * 16 + 7]); }
    printf("first workgroup start -> last workgroup end: %llu ticks\n", last - first);
    return 0;
}
