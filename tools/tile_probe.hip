// tile_probe.hip -- development harness (not shipped): tile_step_kernel on the 784-300-100-10 / B = 128
// shapes with random data: HIP-event time per call of every mode and in-kernel phase stamps (STAMP build).
#include "../graph-neural-net_amd/csrc/tile_step_kernel.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <map>
using namespace gnn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

int main(int argc, char **argv) {
    const int L = 4, dims[4] = {784, 300, 100, 10};
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    int ld[4]; for (int i = 0; i < L; i++) ld[i] = pad_up(dims[i]);
    const int Bp = pad_up(B);
    size_t woff[3], np = 0; for (int l = 0; l < 3; l++) { woff[l] = np; np += (size_t)ld[l] * ld[l + 1]; }
    float *W, *V, *G, *act[4], *delta[4], *slabs, *An; unsigned long long *stamps;
    CK(hipMalloc(&W, np * 4)); CK(hipMalloc(&V, np * 4)); CK(hipMalloc(&G, np * 4));
    std::vector<float> hw(np, 0.f);
    for (int l = 0; l < 3; l++) for (int i = 0; i < dims[l]; i++) for (int j = 0; j < dims[l + 1]; j++)
        hw[woff[l] + (size_t)i * ld[l + 1] + j] = (rand() / (float)RAND_MAX - 0.5f) * 0.2f;
    CK(hipMemcpy(W, hw.data(), np * 4, hipMemcpyHostToDevice)); CK(hipMemset(V, 0, np * 4)); CK(hipMemset(G, 0, np * 4));
    for (int l = 0; l < L; l++) {
        std::vector<float> h((size_t)Bp * ld[l], 0.f);
        for (int b = 0; b < B; b++) for (int i = 0; i < dims[l]; i++) h[(size_t)b * ld[l] + i] = rand() / (float)RAND_MAX - 0.3f;
        CK(hipMalloc(&act[l], h.size() * 4)); CK(hipMemcpy(act[l], h.data(), h.size() * 4, hipMemcpyHostToDevice));
        for (auto &x : h) x *= 1e-3f;
        CK(hipMalloc(&delta[l], h.size() * 4)); CK(hipMemcpy(delta[l], h.data(), h.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipMalloc(&An, (size_t)Bp * ld[0] * 4)); CK(hipMemcpy(An, act[0], (size_t)Bp * ld[0] * 4, hipMemcpyDeviceToDevice));
    const int ns = (ld[0] + TS_TM - 1) / TS_TM;
    CK(hipMalloc(&slabs, (size_t)ns * Bp * ld[1] * 4));
    CK(hipMalloc(&stamps, 4096 * 16 * 8)); CK(hipMemset(stamps, 0, 4096 * 16 * 8));
    TileStepParams t{}; t.n_layers = 3; int tiles = 0, tiles0 = 0;
    for (int l = 0; l < 3; l++) { GradLayer &gl = t.layer[l]; gl.A = act[l]; gl.lda = ld[l]; gl.D = delta[l + 1]; gl.ldd = ld[l + 1];
        gl.W = W + woff[l]; gl.V = V + woff[l]; gl.G = G + woff[l]; gl.M = ld[l]; gl.N = ld[l + 1];
        gl.tiling = make_xcd_tiling((gl.M + TS_TM - 1) / TS_TM, gl.N / TS_TN); gl.block_begin = tiles; tiles += gl.tiling.blocks(); if (!l) tiles0 = tiles; }
    t.K = Bp; t.k_true = B; t.step_over_b = 1e-4f; t.momentum = 0.9f;
    t.An = An; t.ldan = ld[0]; t.next_rows = B; t.next_K = Bp; t.slabs = slabs; t.slab_rows = Bp; t.ldz = ld[1]; t.stamps = stamps;
    // workgroup -> tile: the per-layer XCD rectangles of rounds 2-3 (idle blocks included), and the host-built map
    auto old_map = [&](int n_layers_used, int n_blocks) {
        std::vector<uint32_t> m((size_t)n_blocks, ~0u);
        for (int id = 0; id < n_blocks; id++) {
            int li = 0;
            for (int i = 1; i < n_layers_used; i++) if (id >= t.layer[i].block_begin) li = i;
            const XcdTiling &x = t.layer[li].tiling;
            const int loc = id - t.layer[li].block_begin, xx = loc & 7, j = loc >> 3, q = j / x.rn;
            const int tm = (xx >> x.xs) * x.rm + q, tn = (xx & (x.xn - 1)) * x.rn + (j - q * x.rn);
            if (tm < x.tiles_m && tn < x.tiles_n && q < x.rm) m[id] = (uint32_t)li | (uint32_t)tm << 4 | (uint32_t)tn << 18;
        }
        return m;
    };
    auto to_dev = [&](const std::vector<uint32_t> &m) { uint32_t *d; hipMalloc(&d, m.size() * 4); hipMemcpy(d, m.data(), m.size() * 4, hipMemcpyHostToDevice); return d; };
    const int tiles_old = tiles, tiles0_old = tiles0;
    const uint32_t *map_old = to_dev(old_map(3, tiles_old)), *map0_old = to_dev(old_map(1, tiles0_old));
    TileMapLayer ml[3]; for (int l = 0; l < 3; l++) ml[l] = TileMapLayer{ld[l], ld[l + 1]};
    const std::vector<uint32_t> hm = make_tile_map(ml, 3, 32, true), hm0 = make_tile_map(ml, 1, 32, true);
    const uint32_t *map_new = to_dev(hm), *map0_new = to_dev(hm0);
    tiles = (int)hm.size(); tiles0 = (int)hm0.size();
    t.tile_map = map_new;
    t.map_in_args = pack_tile_map(hm, t.map_words) ? 1 : 0;
    printf("tiles: %d (layer 0: %d) with the host-built map, %d (%d) with the rectangles of every layer; %d slabs\n", tiles, tiles0, tiles_old, tiles0_old, ns);
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *name, int n, auto fn) {
        for (int i = 0; i < 20; i++) fn();
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < n; i++) fn();
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.2f us per call\n", name, ms * 1000.f / n);
    };
    {
        TileStepParams o = t; o.tile_map = map_old; o.map_in_args = 0;
        TileStepParams tg = t; tg.map_in_args = 0; // the host-built map read from memory
        time_it("tile_step<grad, update, fwd>, rectangles of every layer", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles_old), dim3(TS_THREADS), 0, s, o); });
        time_it("tile_step<grad, update, fwd>, host-built map in the kernel arguments", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
        time_it("tile_step<grad, update, fwd>, host-built map read from memory", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, tg); });
        time_it("tile_step<grad, update, fwd>, rectangles of every layer", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles_old), dim3(TS_THREADS), 0, s, o); });
        time_it("tile_step<grad, update, fwd>, host-built map in the kernel arguments", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
        time_it("tile_step<grad, update, fwd>, host-built map read from memory", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, tg); });
        TileStepParams o0 = t; o0.n_layers = 1; o0.tile_map = map0_old; o0.map_in_args = 0;
        TileStepParams n0 = t; n0.n_layers = 1; n0.tile_map = map0_new; n0.map_in_args = pack_tile_map(hm0, n0.map_words) ? 1 : 0;
        time_it("tile_step<fwd only>, rectangles", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0_old), dim3(TS_THREADS), 0, s, o0); });
        time_it("tile_step<fwd only>, host-built map", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0), dim3(TS_THREADS), 0, s, n0); });
    }
    time_it("tile_step<grad, update, fwd> slabs write-through", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 1>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<grad, update, fwd> slabs + W, V write-through", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 2>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<grad, update, fwd>", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 0>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<grad, update, fwd> slabs write-through", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 1>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<grad, update, fwd> slabs + W, V write-through", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 2>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    {   // a pair as in a real step: the tile kernel followed by a dependent small kernel (what the next launch waits for)
        time_it("pair: tile_step + dependent fwd-only launch", 300, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, t); TileStepParams u = t; u.n_layers = 1; u.tile_map = map0_new; u.map_in_args = pack_tile_map(hm0, u.map_words) ? 1 : 0; hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0), dim3(TS_THREADS), 0, s, u); });
        time_it("pair: tile_step (write-through 2) + dependent launch", 300, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 8, 2>), dim3(tiles), dim3(TS_THREADS), 0, s, t); TileStepParams u = t; u.n_layers = 1; u.tile_map = map0_new; u.map_in_args = pack_tile_map(hm0, u.map_words) ? 1 : 0; hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0), dim3(TS_THREADS), 0, s, u); });
    }
    {   // a SAMPLED next batch: rows gathered through an index vector from a 6 000-row data set, with and without the contiguous copy
        const int NR = 6000;
        float *big; CK(hipMalloc(&big, (size_t)NR * ld[0] * 4)); CK(hipMemset(big, 0, (size_t)NR * ld[0] * 4));
        std::vector<int32_t> hidx(Bp); for (int i = 0; i < Bp; i++) hidx[i] = rand() % NR;
        int32_t *didx; CK(hipMalloc(&didx, Bp * 4)); CK(hipMemcpy(didx, hidx.data(), Bp * 4, hipMemcpyHostToDevice));
        float *copy; CK(hipMalloc(&copy, (size_t)Bp * ld[0] * 4));
        TileStepParams g = t; g.An = big; g.next_idx = didx;
        TileStepParams gc = g; gc.stage_out = copy;
        TileStepParams seq = t; seq.An = big; // the same data set, rows in place
        for (int rep = 0; rep < 2; rep++) {
            time_it("next batch in place (6 000-row set)", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, seq); });
            time_it("next batch gathered by index", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, g); });
            time_it("next batch gathered + contiguous copy written", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, gc); });
        }
    }
    time_it("tile_step<grad, update, fwd> 4 waves", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 4>), dim3(tiles), dim3(256), 0, s, t); });
    time_it("tile_step<grad, update, fwd> 4 waves", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, true, false, 4>), dim3(tiles), dim3(256), 0, s, t); });
    time_it("tile_step<grad, store G> 4 waves", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 1, false, false, 4>), dim3(tiles), dim3(256), 0, s, t); });
    time_it("tile_step<fwd only> 4 waves", 500, [&]() { TileStepParams u = t; u.n_layers = 1; u.tile_map = map0_new; u.map_in_args = pack_tile_map(hm0, u.map_words) ? 1 : 0; hipLaunchKernelGGL((tile_step_kernel<0, 0, true, false, 4>), dim3(tiles0), dim3(256), 0, s, u); });
    time_it("tile_step<grad, update>", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 2, false>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<grad, store G>", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<1, 1, false>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<G, update, fwd>", 500, [&]() { hipLaunchKernelGGL((tile_step_kernel<2, 2, true>), dim3(tiles), dim3(TS_THREADS), 0, s, t); });
    time_it("tile_step<fwd only> (layer 0 tiles)", 500, [&]() { TileStepParams u = t; u.n_layers = 1; u.tile_map = map0_new; u.map_in_args = pack_tile_map(hm0, u.map_words) ? 1 : 0; hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0), dim3(TS_THREADS), 0, s, u); });
    {   // 4 waves against 8: the forward-only launch leaves W alone, so its slabs must agree bit for bit
        std::vector<float> s8((size_t)ns * Bp * ld[1]), s4(s8.size());
        TileStepParams u = t; u.n_layers = 1; u.tile_map = map0_new; u.map_in_args = pack_tile_map(hm0, u.map_words) ? 1 : 0;
        hipLaunchKernelGGL((tile_step_kernel<0, 0, true>), dim3(tiles0), dim3(TS_THREADS), 0, s, u);
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(s8.data(), slabs, s8.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(slabs, 0xff, s8.size() * 4));
        hipLaunchKernelGGL((tile_step_kernel<0, 0, true, false, 4>), dim3(tiles0), dim3(256), 0, s, u);
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(s4.data(), slabs, s4.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < s8.size(); i++) bad += memcmp(&s8[i], &s4[i], 4) != 0;
        printf("forward-only slabs, 4 waves vs 8: %zu of %zu words differ\n", bad, s8.size());
    }
    CK(hipMemsetAsync(stamps, 0, 4096 * 16 * 8, s));
    hipLaunchKernelGGL((tile_step_kernel<1, 2, true, true>), dim3(tiles), dim3(TS_THREADS), 0, s, t);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned long long> hs((size_t)tiles * 16);
    CK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0, s1 = 0;
    for (int w = 0; w < tiles; w++) { if (!hs[w * 16 + 8]) continue; t0 = std::min(t0, hs[w * 16 + 8]); t1 = std::max(t1, hs[w * 16 + 9]); s1 = std::max(s1, hs[w * 16 + 8]); }
    printf("first block start -> last block end %.2f us; last block starts %.2f us after the first\n", (t1 - t0) / 100.0, (s1 - t0) / 100.0);
    for (int w : {0, 1, 100, 200, 246, 250, 280}) {
        if (w >= tiles || !hs[w * 16 + 8]) continue;
        const unsigned long long *q = &hs[w * 16];
        printf("wg%-3d start+%.2f us: loads->LDS %llu | grad mfma %llu | reduce+update %llu | sW barrier %llu | fwd mfma %llu | store %llu | total %llu cycles (%.2f us)\n", w,
               (q[8] - t0) / 100.0, q[1] - q[0], q[2] - q[1], q[3] - q[2], q[4] ? q[4] - q[3] : 0, q[5] ? q[5] - q[4] : 0, q[6] ? q[6] - q[5] : 0, (q[6] ? q[6] : q[3]) - q[0], (q[9] - q[8]) / 100.0);
    }
    // who shares a CU with whom: workgroups per (XCC, SE, CU), and the span of the members of shared CUs against the others
    {
        std::map<unsigned, std::vector<int>> cu;
        for (int w = 0; w < tiles; w++) {
            if (!hs[w * 16 + 8] || !hs[w * 16 + 9]) continue; // idle block (left before its first stamp)
            const unsigned hw = (unsigned)hs[w * 16 + 10], xcc = (unsigned)hs[w * 16 + 11] & 15;
            const unsigned key = (xcc << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 8) & 15);
            cu[key].push_back(w);
        }
        int n1 = 0, n2 = 0, n3 = 0; double d1 = 0, d2 = 0, e1 = 0, e2 = 0;
        for (auto &kv : cu) {
            const size_t n = kv.second.size();
            (n == 1 ? n1 : n == 2 ? n2 : n3)++;
            for (int w : kv.second) {
                const double dur = (hs[w * 16 + 9] - hs[w * 16 + 8]) / 100.0, end = (hs[w * 16 + 9] - t0) / 100.0;
                if (n == 1) { d1 = std::max(d1, dur); e1 = std::max(e1, end); } else { d2 = std::max(d2, dur); e2 = std::max(e2, end); }
            }
        }
        printf("CUs with 1 / 2 / 3+ live workgroups: %d / %d / %d; longest workgroup alone on its CU %.2f us (ends +%.2f), sharing %.2f us (ends +%.2f)\n", n1, n2, n3, d1, e1, d2, e2);
        int shown = 0;
        for (auto &kv : cu) {
            if (kv.second.size() < 2 || shown >= 12) continue;
            printf("  xcc %u se %u cu %2u:", kv.first >> 16, (kv.first >> 8) & 255, kv.first & 255);
            for (int w : kv.second) printf("  wg%-3d [+%.2f, +%.2f]", w, (hs[w * 16 + 8] - t0) / 100.0, (hs[w * 16 + 9] - t0) / 100.0);
            printf("\n");
            shown++;
        }
    }
    return 0;
}
