// gemm_wavek.h -- f32 GEMM for outputs of only a few hundred 32 x 32 tiles (a 256-row batch against a 1024-wide
// layer: 256 tiles, ONE per CU), where gemm_f32_kernel is bound by its own barriers: one 4-wave workgroup per CU,
// one wave per SIMD, and every 64-deep k tile pays store -> barrier -> LDS latency -> 16 dependent MFMAs -> barrier
// (~1 400 cycles for 512 cycles of MFMAs; 256 x 1024 x 1024 in 12.2 us = 28 % of the f32 MFMA peak, whatever the
// prefetch depth: profiles/r02/gemm_probe_v1.log).
//
// Here every WAVE is a GEMM of its own: wave w multiplies the whole 32 x 32 tile over its own slice of K (K split
// NW ways inside the workgroup), with its own operand registers and its own corner of LDS -- there is no barrier in
// the main loop, the waves drift apart, and each has four independent accumulators (2 x 2 MFMA tiles) instead of
// one.  The NW partial tiles are summed once at the end, in wave order (fixed: the result does not depend on
// timing), by the epilogue.
//
// Operand paths (chunks of 32 k; MFMA j of a chunk gives k slot q the element k = 8q + j on BOTH operands):
//   k-contiguous operand (X[index][k]):  lane (fr, fq) loads X[2 fr + i][kc + 8 fq .. + 7] -- the fragment of the
//       chunk's eight MFMAs, straight from global memory to registers, no LDS at all; the four k slots of a row
//       make one whole 128-B line (with 16-k chunks every line was fetched twice, the second time after it had
//       left the L1: deeper prefetch made the kernel SLOWER, profiles/r02/gemm_probe_wavek1.log);
//   index-contiguous operand (X[k][index]): the wave copies the [32 k][32] block to its private LDS image
//       (16-B loads and writes) and reads one b64 per MFMA (rows/columns interleaved: tile i holds 2 rho + i).
// LDS operations of one wave execute in order, so a wave needs no barrier between its own writes and reads.
#pragma once
#include "kernels.h"

namespace gnn {

constexpr int WK_CH = 32;   // k per chunk: 128 B of every row of a k-contiguous operand -- whole cache lines, used once
constexpr int WK_LDT = 36;  // private [32 k][32 + 4] image: rows 8 apart are 32 banks apart (b64 reads conflict-free)
constexpr int WK_LDP = 36;  // partial tiles [32][32 + 4]
constexpr int WK_WAVE_FLOATS = 2 * WK_CH * WK_LDT; // per wave: two operand images; the partial tile reuses them
static_assert(32 * WK_LDP <= WK_WAVE_FLOATS, "the partial tile reuses the wave's operand images");

// (GNN_GEMM_HEAD_PARAMS, kernels.h: the main loop's first values preloaded into SGPRs.  HEAD = false: the struct's own copies --
//  tools/gemm_probe 21 compares the two: 8.85 against 9.07 us forward, 10.0 against 10.2 backward at 256 x 1024 x 1024.)
// RAGGED (round 4): M and / or N need not be multiples of 32 (they are multiples of 16: 608 rows of a validation block, the 304
// columns of a 300-wide layer).  Loads of rows past M and of columns past N are CLAMPED to the last row / the last float4 of the
// row -- valid addresses whose products land in accumulator elements nobody stores -- and the epilogue guards its accesses.
template <bool A_KC, bool B_KC, int EPI, int NW, int DEPTH, bool HEAD = true, bool RAGGED = false>
__global__ __launch_bounds__(NW * 64) void gemm_f32_wavek_kernel(GNN_GEMM_HEAD_PARAMS(float), GemmParams p) {
    if constexpr (HEAD) GNN_GEMM_TAKE_HEAD(p);
    __shared__ __attribute__((aligned(16))) float lds[NW * WK_WAVE_FLOATS];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32; // M and N are multiples of 32 (checked by the host) unless RAGGED
    float *sA = lds + wave * WK_WAVE_FLOATS, *sB = sA + WK_CH * WK_LDT;

    // this wave's chunks; K is a multiple of 16: the last chunk may be half a chunk ("tail")
    const int nch = (p.K + WK_CH - 1) / WK_CH;
    const bool half_tail = (p.K & 16) != 0;
    const int cw = (nch + NW - 1) / NW;
    const int c_begin = wave * cw;
    const int c_end = (c_begin + cw < nch) ? c_begin + cw : nch;
    const int c_last = (c_end > c_begin) ? c_end - 1 : nch - 1; // loads past the slice repeat its last chunk (cache hits, never used)

    // this lane's loads of a chunk: four 16-B pieces per operand
    //   k-contiguous X[index][k]: piece 2 i + h = X[2 fr + i][kc + 8 fq + 4 h .. + 3]   (MFMA j: k slot q holds k = 8 q + j)
    //   index-contiguous X[k][index]: piece u = float4 number lane + 64 u of the [32 k][32] block: k = 8 u + (lane >> 3)
    // In a half chunk the pieces of k >= 16 are re-read from k - 16 (inside the operand) and zeroed before use.
    const float *a_src[4], *a_tail[4], *b_src[4], *b_tail[4];
    size_t a_step, b_step;
    auto in_m = [&](int m) { return (RAGGED && m >= p.M) ? p.M - 1 : m; };   // (index of an A row / C row)
    auto in_n = [&](int n) { return (RAGGED && n >= p.N) ? p.N - 1 : n; };   // (index of a B row when B is k-contiguous)
    auto in_m4 = [&](int m) { return (RAGGED && m + 4 > p.M) ? p.M - 4 : m; }; // (first of four consecutive indices)
    auto in_n4 = [&](int n) { return (RAGGED && n + 4 > p.N) ? p.N - 4 : n; };
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (A_KC) {
            a_src[u] = p.A + (size_t)in_m(m0 + 2 * fr + (u >> 1)) * p.lda + 8 * fq + 4 * (u & 1);
            a_tail[u] = a_src[u] - (fq >= 2 ? 16 : 0);
        } else {
            a_src[u] = p.A + (size_t)(8 * u + (lane >> 3)) * p.lda + in_m4(m0 + 4 * (lane & 7));
            a_tail[u] = a_src[u] - (u >= 2 ? (size_t)16 * p.lda : 0);
        }
        if (B_KC) {
            b_src[u] = p.B + (size_t)in_n(n0 + 2 * fr + (u >> 1)) * p.ldb + 8 * fq + 4 * (u & 1);
            b_tail[u] = b_src[u] - (fq >= 2 ? 16 : 0);
        } else {
            b_src[u] = p.B + (size_t)(8 * u + (lane >> 3)) * p.ldb + in_n4(n0 + 4 * (lane & 7));
            b_tail[u] = b_src[u] - (u >= 2 ? (size_t)16 * p.ldb : 0);
        }
    }
    a_step = A_KC ? (size_t)WK_CH : (size_t)WK_CH * p.lda;
    b_step = B_KC ? (size_t)WK_CH : (size_t)WK_CH * p.ldb;
    struct Stage { f32x4 a[4], b[4]; };
    auto load = [&](Stage &s, int c) { // unconditional (c clamped): the load sequence is the same on every path
        const int cc = c < c_last ? c : c_last;
        const bool tail = half_tail && cc == nch - 1; // wave-uniform
#pragma unroll
        for (int u = 0; u < 4; u++) {
            s.a[u] = *reinterpret_cast<const f32x4 *>((tail ? a_tail[u] : a_src[u]) + cc * a_step);
            s.b[u] = *reinterpret_cast<const f32x4 *>((tail ? b_tail[u] : b_src[u]) + cc * b_step);
        }
    };
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int w_off = (lane >> 3) * WK_LDT + 4 * (lane & 7); // image write: piece u at row 8 u + (lane >> 3)
    const int r_off = 8 * fq * WK_LDT + 2 * fr;              // image read of MFMA j: row 8 fq + j
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    auto multiply = [&](Stage &s, int c) {
        if (half_tail && c == nch - 1) { // wave-uniform: the upper half of the chunk is past K
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (A_KC ? fq >= 2 : u >= 2) s.a[u] = zero4;
                if (B_KC ? fq >= 2 : u >= 2) s.b[u] = zero4;
            }
        }
        if (!A_KC) {
#pragma unroll
            for (int u = 0; u < 4; u++) *reinterpret_cast<f32x4 *>(&sA[w_off + 8 * u * WK_LDT]) = s.a[u];
        }
        if (!B_KC) {
#pragma unroll
            for (int u = 0; u < 4; u++) *reinterpret_cast<f32x4 *>(&sB[w_off + 8 * u * WK_LDT]) = s.b[u];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int jh = 0; jh < 2; jh++) { // four MFMA steps at a time: their LDS reads first
            float av[4][2], bv[4][2];
#pragma unroll
            for (int jl = 0; jl < 4; jl++) {
                const int j = jh * 4 + jl;
                if (A_KC) { av[jl][0] = s.a[jh][jl]; av[jl][1] = s.a[2 + jh][jl]; }
                else { const float2 v = *reinterpret_cast<const float2 *>(&sA[r_off + j * WK_LDT]); av[jl][0] = v.x; av[jl][1] = v.y; }
                if (B_KC) { bv[jl][0] = s.b[jh][jl]; bv[jl][1] = s.b[2 + jh][jl]; }
                else { const float2 v = *reinterpret_cast<const float2 *>(&sB[r_off + j * WK_LDT]); bv[jl][0] = v.x; bv[jl][1] = v.y; }
            }
#pragma unroll
            for (int jl = 0; jl < 4; jl++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int jj = 0; jj < 2; jj++) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jl][i], bv[jl][jj], acc[i][jj], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
    };

    // DEPTH chunks ahead in registers.  The steady loop issues only loads of real chunks; the last trips are peeled
    // (loads past the slice -- even as cache hits -- cost issue slots and hold the wave at its end: with them
    // depth 6 ran 13.4 us against 9.0 us for depth 2)
    Stage st[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH; s++) load(st[s], c_begin + s);
    int c = c_begin;
    for (; c + 2 * DEPTH <= c_end; c += DEPTH) {
#pragma unroll
        for (int s = 0; s < DEPTH; s++) {
            multiply(st[s], c + s);
            load(st[s], c + s + DEPTH);
        }
    }
#pragma unroll
    for (int s = 0; s < DEPTH; s++) {
        if (c + s < c_end) { // wave-uniform
            multiply(st[s], c + s);
            if (c + s + DEPTH < c_end) load(st[s], c + s + DEPTH);
        }
    }
    c += DEPTH;
#pragma unroll
    for (int s = 0; s < DEPTH; s++)
        if (c + s < c_end) multiply(st[s], c + s);

    // partial tile of this wave -> LDS [m][n]: m = 2 (4 fq + r) + i, n = 2 fr + j  (C/D map: row 4 fq + r, column fr)
    float *sP = sA;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            *reinterpret_cast<float2 *>(&sP[(2 * (4 * fq + r) + i) * WK_LDP + 2 * fr]) = make_float2(acc[i][0][r], acc[i][1][r]);
    __syncthreads();
    if (t >= 256) return;
    const int row = t >> 3, cq = t & 7;
    f32x4 v = *reinterpret_cast<const f32x4 *>(&lds[row * WK_LDP + 4 * cq]);
#pragma unroll
    for (int w = 1; w < NW; w++) v += *reinterpret_cast<const f32x4 *>(&lds[w * WK_WAVE_FLOATS + row * WK_LDP + 4 * cq]);

    const int m = m0 + row, n = n0 + 4 * cq;
    if (RAGGED && (m >= p.M || n >= p.N)) return; // (N is a multiple of 16: the four columns are all inside or all outside)
    const size_t off = (size_t)m * p.ldc + n;
    f32x4 out0, out1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 aux = {0.f, 0.f, 0.f, 0.f}, vold = aux, wold = aux;
    if (EPI == EPI_DACT) aux = *reinterpret_cast<const f32x4 *>(p.aux + (size_t)m * p.ldaux + n);
    if (EPI == EPI_SGD) { vold = *reinterpret_cast<const f32x4 *>(p.V + off); wold = *reinterpret_cast<const f32x4 *>(p.W + off); }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool live = (m < p.m_true) && (n + j < p.n_true);
        if (EPI == EPI_STORE) out0[j] = live ? v[j] : 0.f;
        else if (EPI == EPI_ACT) out0[j] = live ? act_fn(p.act, v[j]) : 0.f;
        else if (EPI == EPI_DACT) out0[j] = live ? v[j] * act_prime_from_a(p.act, aux[j]) : 0.f;
        else { // EPI_SGD: ((step*G)/B) + (momentum*prev), SCE:333; padding elements stay as they are (zeros)
            const float adj = sgd_adj(p.step_over_b, v[j], p.momentum, vold[j]);
            out0[j] = live ? wold[j] - adj : wold[j];
            out1[j] = live ? adj : vold[j];
        }
    }
    if (EPI == EPI_SGD) {
        *reinterpret_cast<f32x4 *>(p.W + off) = out0;
        *reinterpret_cast<f32x4 *>(p.V + off) = out1;
    } else {
        *reinterpret_cast<f32x4 *>(p.C + off) = out0;
    }
}

} // namespace gnn
