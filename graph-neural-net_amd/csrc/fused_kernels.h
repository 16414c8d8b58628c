// fused_kernels.h -- the latency-oriented path for SMALL nets (the headline 784-300-100-10 at
// batch 128 is 0.14 GFLOP per step: every kernel is bound by dependent-launch and memory
// latency, not by MFMA or HBM throughput).  One gradientStep (SCE:297-346) is three launches:
//
//   fwd_first_kernel   A_1 = f(A_0 . W_0)            one 16x16 output tile per workgroup, the K
//                                                     dimension split over the workgroup's waves
//                                                     (in-LDS reduction, no global partials)
//   middle4_kernel     (middle4_kernel.h) per 4-row block of the batch: forward through every
//                      remaining layer, softmax / loss / delta_{L-1}, and the whole backward-data
//                      chain down to delta_1 -- all per-sample independent (SCE:164-198,
//                      SCE:249-278), so a workgroup needs no other workgroup's data
//   grad_update_kernel every layer's G_l = A_l^T . delta_{l+1} (the only cross-sample sum,
//                      SCE:305-322) with the momentum update (SCE:333-339) fused into the
//                      epilogue, one launch for all layers
//
// All contractions run on v_mfma_f32_16x16x4_f32.  Operands that are k-contiguous in memory are
// read 16 B per lane (4 consecutive k); the four k-slots of one MFMA then hold k = base+4q+j
// (q = lane>>4) -- any assignment of distinct k to slots is valid as long as A and B agree.
#pragma once
#include "kernels.h"

namespace gnn {

constexpr int MAX_LAYERS = 8;

// ------------------------------------------------------------------------------------------
// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (blocks b and b+8 share
// an L2), and every kernel starts with a cold L2, so a tile's operand panels should be shared
// with the OTHER tiles of the same XCD: the tile grid is cut into an xm x xn grid of rectangles
// (xm*xn = 8), one rectangle per XCD label.  Placement is a speed matter only; a different
// dispatch order changes nothing but L2 hit rates.
// ------------------------------------------------------------------------------------------
struct XcdTiling {
    int tiles_m, tiles_n; // real tile counts
    int xn;               // rectangles along n (xm = 8 / xn), a power of two
    int rm, rn;           // rectangle extent in tiles
    int xs;               // log2(xn)
    float inv_rn;         // 1 / rn
    __host__ __device__ int blocks() const { return 8 * rm * rn; }
    // local block id -> tile; false = idle block.  Every workgroup runs this before its first load: no integer division
    // (two of them were ~50 instructions): xn is a power of two, and j / rn through the reciprocal is exact for these sizes
    // ((j + 0.5) / rn is at least 0.5 / rn away from an integer; j, rn < 2^16).
    __device__ __forceinline__ bool tile_of(int id, int &tm, int &tn) const {
        const int x = id & 7, j = id >> 3;
        const int q = (int)(((float)j + 0.5f) * inv_rn);
        tm = (x >> xs) * rm + q;
        tn = (x & (xn - 1)) * rn + (j - q * rn);
        return tm < tiles_m && tn < tiles_n && q < rm;
    }
};
__host__ inline XcdTiling make_xcd_tiling(int tiles_m, int tiles_n) {
    XcdTiling best{};
    long best_cost = -1;
    for (int xn = 1; xn <= 8; xn *= 2) {
        const int xm = 8 / xn;
        const int rm = (tiles_m + xm - 1) / xm, rn = (tiles_n + xn - 1) / xn;
        const long cost = (long)(rm + rn) * 1000 + (long)rm * rn; // panels fetched per XCD, then idle blocks
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            int xs = 0;
            while ((1 << xs) < xn) xs++;
            best = XcdTiling{tiles_m, tiles_n, xn, rm, rn, xs, 1.0f / (float)rn};
        }
    }
    return best;
}

#define GNN_STAMP_AT(ptr, i)                                                                  \
    do {                                                                                      \
        if (STAMP && threadIdx.x == 0) (ptr)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
// s_memtime counts per-CU clocks with unrelated origins; the 100 MHz real-time counter is shared by
// the whole device and is what block-to-block spans are measured with (slots 4 and 5)
#define GNN_STAMP_REAL(ptr, i)                                                                \
    do {                                                                                      \
        if (STAMP && threadIdx.x == 0) (ptr)[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// ------------------------------------------------------------------------------------------
// fwd_first_kernel: C[M x N] = epi(A[M x K] . W[K x N]), A k-contiguous, W n-contiguous.
// One 16x16 output tile per workgroup; NW waves split K in 16-wide chunks, partial tiles are
// summed through LDS.  Operands go straight to registers (no wave shares another's K slice).
// ------------------------------------------------------------------------------------------
struct FwdFirstParams {
    const float *A; int lda;
    const float *W; int ldw;
    float *C; int ldc;
    int M, N, K;          // padded extents (multiples of 16)
    int m_true, n_true;
    int act, apply_act;
    XcdTiling tiling;
    unsigned long long *stamps; // STAMP builds only
    const int32_t *row_idx;     // optional: row m of A is dataset row row_idx[m] (sampled batches, NNT:143-158)
};

template <int NW, bool STAMP = false, int ACT = -1>
__global__ __launch_bounds__(NW * 64) void fwd_first_kernel(FwdFirstParams p) {
    constexpr int MAXC = NW >= 8 ? 8 : NW >= 4 ? 13 : NW == 3 ? 17 : 25; // chunks whose loads are in flight at once (MAXC * (4+4) VGPRs)
    constexpr int RLD = 20; // row stride of a partial tile in LDS
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * RLD];
    int tm, tn;
    if (!p.tiling.tile_of(blockIdx.x, tm, tn)) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int n0 = tn * 16, m0 = tm * 16;
    const int k16 = p.K / 16;
    const int c_begin = (int)((long)wave * k16 / NW), c_end = (int)((long)(wave + 1) * k16 / NW);

    int a_row = m0 + fr;
    if (p.row_idx) a_row = a_row < p.m_true ? p.row_idx[a_row] : 0; // rows past the batch: any valid row, masked at the end
    const float *arow = p.A + (size_t)a_row * p.lda + 4 * fq;
    const float *wcol = p.W + (size_t)(4 * fq) * p.ldw + n0 + fr;

    GNN_STAMP_AT(p.stamps, 0);
    GNN_STAMP_REAL(p.stamps, 4);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int cb = c_begin; cb < c_end; cb += MAXC) {
        float4 a[MAXC];
        float b[MAXC][4];
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
            const int c = cb + i;
            if (c < c_end) {
                a[i] = *reinterpret_cast<const float4 *>(arow + c * 16);
                const float *w = wcol + (size_t)(c * 16) * p.ldw;
                b[i][0] = w[0];
                b[i][1] = w[p.ldw];
                b[i][2] = w[2 * p.ldw];
                b[i][3] = w[3 * p.ldw];
            } else {
                a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                b[i][0] = b[i][1] = b[i][2] = b[i][3] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
            if (i & 1) {
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i][0], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i][1], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i][2], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i][3], acc1, 0, 0, 0);
            } else {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i][0], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i][1], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i][2], acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i][3], acc0, 0, 0, 0);
            }
        }
    }
    const f32x4 acc = acc0 + acc1;
    GNN_STAMP_AT(p.stamps, 1);
    // partial tile, row-major: row = fq*4 + r, col = fr
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + fq * 4 + r) * RLD + fr] = acc[r];
    __syncthreads();
    GNN_STAMP_AT(p.stamps, 2);
    if (t < 64) { // thread -> (row m = t/4, 4 columns): one 16-B store per lane
        const int m = t >> 2, q = t & 3;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; w++) s += *reinterpret_cast<const f32x4 *>(&red[(w * 16 + m) * RLD + q * 4]);
        float4 o;
        float *ov = reinterpret_cast<float *>(&o);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool live = (m0 + m < p.m_true) && (n0 + q * 4 + j < p.n_true);
            float v = s[j];
            if (ACT >= 0) v = act_fn(ACT, v);
            else if (p.apply_act) v = act_fn(p.act, v);
            ov[j] = live ? v : 0.f;
        }
        *reinterpret_cast<float4 *>(p.C + (size_t)(m0 + m) * p.ldc + n0 + q * 4) = o;
    }
    GNN_STAMP_AT(p.stamps, 3);
    GNN_STAMP_REAL(p.stamps, 5);
}

// ------------------------------------------------------------------------------------------
// grad_update_kernel: G_l = A_l^T . D_{l+1} for every layer in one launch, K = batch rows.
// One 32x32 tile of one layer per workgroup (4 waves, 2x2 MFMA tiles); the K extent is staged
// to LDS in chunks of 128 rows.  FUSED: the momentum update replaces the store of G.
// All global traffic is 16 B per lane: the accumulator tile is transposed through LDS so that
// W, V (or G) move as whole 128-B rows.
// ------------------------------------------------------------------------------------------
struct GradLayer {
    const float *A; int lda;   // activations of layer l   [K][lda]
    const float *D; int ldd;   // deltas of layer l+1      [K][ldd]
    float *W; float *V; float *G; // all [M][ldd]
    int M, N;                  // padded extents of W_l
    XcdTiling tiling;
    int block_begin;           // first block of this layer (a multiple of 8)
};
struct GradParams {
    GradLayer layer[MAX_LAYERS];
    int n_layers;
    int K;                     // padded batch rows
    float step_over_b, momentum;
    unsigned long long *stamps; // STAMP builds only
    const int32_t *row_idx;     // optional, layer 0 only: batch row k of A_0 is dataset row row_idx[k]
    int k_true;                 // live batch rows (rows past them are zeros when row_idx is given)
};

constexpr int GRAD_THREADS = 512; // default: 8 waves; 256 (4 waves, no K halves) for grids of > ~1000 tiles
template <bool FUSED, bool STAMP = false, int NTHR = GRAD_THREADS>
__global__ __launch_bounds__(NTHR) void grad_update_kernel(GradParams p) {
    constexpr int KSPLIT = NTHR / 256;
    constexpr int KC = 128, LDS_LD = 48; // row stride = 16 (mod 32) floats
    constexpr int CLD = 36;
    __shared__ __attribute__((aligned(16))) float As[KC * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Ds[KC * LDS_LD];
    __shared__ __attribute__((aligned(16))) float Cs[KSPLIT * 32 * CLD];
    // 8 waves: 2x2 MFMA tiles x 2 halves of every K chunk.  The f32 MFMA work of a tile is fixed
    // (128 instructions of 32 cycles over 4 SIMDs), but with one wave per SIMD its LDS reads and
    // the dependent accumulator chain are exposed; two waves per SIMD cover each other.
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4, wm = (wave >> 1) & 1, wn = wave & 1, kh = wave >> 2;

    int li = 0;
#pragma unroll
    for (int i = 1; i < MAX_LAYERS; i++)
        if (i < p.n_layers && (int)blockIdx.x >= p.layer[i].block_begin) li = i;
    const GradLayer &L = p.layer[li];
    int tm, tn;
    if (!L.tiling.tile_of(blockIdx.x - L.block_begin, tm, tn)) return;
    const int m0 = tm * 32, n0 = tn * 32;

    // this thread's 16 B of the output tile: row er, columns 4*eq .. 4*eq+3
    const int er = (t >> 3) & 31, eq = t & 7;
    const bool e_ok = t < 256 && (m0 + er < L.M) && (n0 + eq * 4 < L.N);
    const size_t e_off = (size_t)(m0 + er) * L.ldd + n0 + eq * 4;
    GNN_STAMP_AT(p.stamps, 0);
    GNN_STAMP_REAL(p.stamps, 4);
    float4 w_old = make_float4(0.f, 0.f, 0.f, 0.f), v_old = w_old;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        const int kc = (p.K - k0 < KC) ? p.K - k0 : KC;
        if (k0) __syncthreads();
        // 32 floats (8 float4) per row per operand; thread -> (row = idx / 8, q = idx % 8).  The operand
        // panels are on the critical path and are requested FIRST; the W / V rows of the update are needed
        // only after the GEMM and queue behind them (their latency hides under the MFMAs)
        float4 va[KC * 8 / NTHR], vd[KC * 8 / NTHR];
#pragma unroll
        for (int i = 0; i < KC * 8 / NTHR; i++) {
            const int idx = t + i * NTHR, k = idx >> 3, q = idx & 7;
            va[i] = make_float4(0.f, 0.f, 0.f, 0.f); vd[i] = va[i];
            if (k < kc) {
                size_t a_row = (size_t)(k0 + k);
                bool a_live = true;
                if (li == 0 && p.row_idx) { // sampled batch: the input rows are gathered here, not copied first
                    a_live = k0 + k < p.k_true;
                    a_row = a_live ? (size_t)p.row_idx[k0 + k] : 0;
                }
                if (a_live && m0 + q * 4 < L.M) va[i] = *reinterpret_cast<const float4 *>(L.A + a_row * L.lda + m0 + q * 4);
                if (n0 + q * 4 < L.N) vd[i] = *reinterpret_cast<const float4 *>(L.D + (size_t)(k0 + k) * L.ldd + n0 + q * 4);
            }
        }
        if (FUSED && e_ok && k0 == 0) {
            w_old = *reinterpret_cast<const float4 *>(L.W + e_off);
            v_old = *reinterpret_cast<const float4 *>(L.V + e_off);
        }
#pragma unroll
        for (int i = 0; i < KC * 8 / NTHR; i++) {
            const int idx = t + i * NTHR, k = idx >> 3, q = idx & 7;
            *reinterpret_cast<float4 *>(&As[k * LDS_LD + q * 4]) = va[i];
            *reinterpret_cast<float4 *>(&Ds[k * LDS_LD + q * 4]) = vd[i];
        }
        __syncthreads();
        GNN_STAMP_AT(p.stamps, 1);
        const float *ap = &As[fq * LDS_LD + wm * 16 + fr];
        const float *dp = &Ds[fq * LDS_LD + wn * 16 + fr];
        // 32 k (8 MFMAs) per trip, the trip's 16 LDS reads issued before its first MFMA; a full
        // 128-row chunk is a compile-time trip count so the compiler overlaps consecutive trips
        auto trip = [&](int kk) {
            float a[8], d[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                a[j] = ap[(kk + 4 * j) * LDS_LD];
                d[j] = dp[(kk + 4 * j) * LDS_LD];
            }
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], d[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1], d[j + 1], acc1, 0, 0, 0);
            }
        };
        if (kc == KC) { // this wave's half of the chunk
#pragma unroll
            for (int kk = 0; kk < KC / KSPLIT; kk += 32) trip(kh * (KC / KSPLIT) + kk);
        } else {        // kc is a multiple of 16: 4-k steps dealt alternately to the two halves
            for (int kk = 4 * kh; kk < kc; kk += 4 * KSPLIT)
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kk * LDS_LD], dp[kk * LDS_LD], acc0, 0, 0, 0);
        }
    }
    const f32x4 acc = acc0 + acc1;
    GNN_STAMP_AT(p.stamps, 2);
#pragma unroll
    for (int r = 0; r < 4; r++) Cs[(kh * 32 + wm * 16 + fq * 4 + r) * CLD + wn * 16 + fr] = acc[r];
    __syncthreads();
    if (e_ok) {
        float4 gsum = *reinterpret_cast<const float4 *>(&Cs[er * CLD + eq * 4]);
        if (KSPLIT == 2) {
            const float4 g1 = *reinterpret_cast<const float4 *>(&Cs[(32 + er) * CLD + eq * 4]);
            gsum = make_float4(gsum.x + g1.x, gsum.y + g1.y, gsum.z + g1.z, gsum.w + g1.w);
        }
        if (FUSED) { // ((step*G)/B) + (momentum*prev), SCE:333
            float4 adj, wn_;
            adj.x = sgd_adj(p.step_over_b, gsum.x, p.momentum, v_old.x);
            adj.y = sgd_adj(p.step_over_b, gsum.y, p.momentum, v_old.y);
            adj.z = sgd_adj(p.step_over_b, gsum.z, p.momentum, v_old.z);
            adj.w = sgd_adj(p.step_over_b, gsum.w, p.momentum, v_old.w);
            wn_.x = w_old.x - adj.x; wn_.y = w_old.y - adj.y; wn_.z = w_old.z - adj.z; wn_.w = w_old.w - adj.w;
            *reinterpret_cast<float4 *>(L.W + e_off) = wn_;
            *reinterpret_cast<float4 *>(L.V + e_off) = adj;
        } else {
            *reinterpret_cast<float4 *>(L.G + e_off) = gsum;
        }
    }
    GNN_STAMP_AT(p.stamps, 3);
    GNN_STAMP_REAL(p.stamps, 5);
}

// ------------------------------------------------------------------------------------------
// grad_update64_kernel: the same one-launch gradient (+ update) with 64x64 tiles, for nets whose
// 32x32 grid runs to thousands of tiles: a 32x32 tile reads its two K x 32 panels for 1024 outputs,
// a 64x64 tile two K x 64 panels for 4096 -- half the L2 traffic per output (784-1024^3-10 at 256
// rows moved 210 MB through L2 per step with 32x32 tiles, ~6 TB/s, and was bound by it).
// 8 waves = 2x2 sub-tiles of 32x32 (2x2 MFMA tiles each, rows/columns interleaved so that an operand
// fragment is one ds_read_b64) x 2 halves of every 64-row K chunk; partial tiles meet in LDS.
// ------------------------------------------------------------------------------------------
template <bool FUSED, bool PF = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(6, 6))) void grad_update64_kernel(GradParams p) {
    constexpr int T = 64, KC = 64, LD = T + 16, CLD = T + 4;
    __shared__ __attribute__((aligned(16))) float sm[2 * KC * LD]; // A and D panels; later the two partial tiles
    static_assert(2 * T * CLD <= 2 * KC * LD, "partial tiles reuse the operand panels");
    float *As = sm, *Ds = sm + KC * LD;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4, wm = (wave >> 1) & 1, wn = wave & 1, kh = wave >> 2;

    int li = 0;
#pragma unroll
    for (int i = 1; i < MAX_LAYERS; i++)
        if (i < p.n_layers && (int)blockIdx.x >= p.layer[i].block_begin) li = i;
    const GradLayer &L = p.layer[li];
    int tm, tn;
    if (!L.tiling.tile_of(blockIdx.x - L.block_begin, tm, tn)) return;
    const int m0 = tm * T, n0 = tn * T;

    // this thread's two 16-B pieces of the output tile: rows er and er + 32, columns 4*eq .. 4*eq+3
    const int er = t >> 4, eq = t & 15;
    bool e_ok[2];
    size_t e_off[2];
    float4 w_old[2], v_old[2];
#pragma unroll
    for (int i = 0; i < 2; i++) {
        e_ok[i] = (m0 + er + 32 * i < L.M) && (n0 + eq * 4 < L.N);
        e_off[i] = (size_t)(m0 + er + 32 * i) * L.ldd + n0 + eq * 4;
        w_old[i] = make_float4(0.f, 0.f, 0.f, 0.f); v_old[i] = w_old[i];
    }
    // tiles wholly inside their layer with the batch rows in place load without bounds tests (see gemm_f32_kernel)
    const bool interior = (m0 + T <= L.M) && (n0 + T <= L.N) && (p.K % KC == 0) && !(li == 0 && p.row_idx) &&
                          (unsigned long long)p.K * (unsigned)(L.lda > L.ldd ? L.lda : L.ldd) < 0xffffffffull; // 32-bit offsets
    // PF, interior tiles: the NEXT chunk's operands are requested before this chunk's MFMAs and the W / V rows take those
    // registers' place during the last chunk -- held from the start they cost the 16 VGPRs the prefetch needs (3 -> 2
    // workgroups per CU with both)
    const bool late_wv = PF && interior;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (FUSED && e_ok[i] && !late_wv) { // W / V rows first: their latency hides under the GEMM
            w_old[i] = *reinterpret_cast<const float4 *>(L.W + e_off[i]);
            v_old[i] = *reinterpret_cast<const float4 *>(L.V + e_off[i]);
        }
    }

    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (late_wv) {
        // (named registers and macros, not arrays captured by lambdas: those were given a home in scratch memory)
        float4 va0, va1, vd0, vd1;
        const unsigned rk = (unsigned)(t >> 4), rq = (unsigned)(t & 15) * 4;
#define GNN_GU64_REQUEST(k0_)                                                                                                   \
    do {                                                                                                                        \
        va0 = *reinterpret_cast<const float4 *>(L.A + (((unsigned)(k0_) + rk) * (unsigned)L.lda + m0 + rq));                    \
        va1 = *reinterpret_cast<const float4 *>(L.A + (((unsigned)(k0_) + rk + 32) * (unsigned)L.lda + m0 + rq));               \
        vd0 = *reinterpret_cast<const float4 *>(L.D + (((unsigned)(k0_) + rk) * (unsigned)L.ldd + n0 + rq));                    \
        vd1 = *reinterpret_cast<const float4 *>(L.D + (((unsigned)(k0_) + rk + 32) * (unsigned)L.ldd + n0 + rq));               \
    } while (0)
#define GNN_GU64_LAND()                                                                                                         \
    do {                                                                                                                        \
        *reinterpret_cast<float4 *>(&As[rk * LD + rq]) = va0;                                                                   \
        *reinterpret_cast<float4 *>(&As[(rk + 32) * LD + rq]) = va1;                                                            \
        *reinterpret_cast<float4 *>(&Ds[rk * LD + rq]) = vd0;                                                                   \
        *reinterpret_cast<float4 *>(&Ds[(rk + 32) * LD + rq]) = vd1;                                                            \
    } while (0)
        GNN_GU64_REQUEST(0);
        const float *ap = &As[(kh * (KC / 2) + fq) * LD + wm * 32 + fr * 2];
        const float *dp = &Ds[(kh * (KC / 2) + fq) * LD + wn * 32 + fr * 2];
        // this wave's half of the chunk in FOUR groups of eight k (two MFMA k steps, eight MFMAs); a group's operand fragments are
        // read from LDS while the group before it multiplies (round 4: the compiler's order was read - wait - multiply per group,
        // every group's LDS latency in front of its MFMAs: the ISA of profiles/r04/grad_update64_isa_before.txt)
        auto multiply = [&]() {
            constexpr int NG = KC / 2 / 8;
            f32x2 a[2][2], d[2][2];
            auto fetch = [&](int g, int buf) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    a[buf][u] = *reinterpret_cast<const f32x2 *>(ap + (8 * g + 4 * u) * LD);
                    d[buf][u] = *reinterpret_cast<const f32x2 *>(dp + (8 * g + 4 * u) * LD);
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int g = 0; g < NG; g++) {
                if (g + 1 < NG) fetch(g + 1, (g + 1) & 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int i = 0; i < 2; i++)
#pragma unroll
                        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][u][i], d[g & 1][u][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        int k0 = 0;
        for (; k0 + KC < p.K; k0 += KC) {
            if (k0) __syncthreads();
            GNN_GU64_LAND();
            __syncthreads();
            GNN_GU64_REQUEST(k0 + KC);
            __builtin_amdgcn_sched_barrier(0); // (the scheduler otherwise sinks the requests behind the MFMAs, next to their use)
            multiply();
        }
        if (k0) __syncthreads();
        GNN_GU64_LAND();
        __syncthreads();
#undef GNN_GU64_REQUEST
#undef GNN_GU64_LAND
        if (FUSED) { // (the last chunk: the W / V rows ride where the operand prefetch did)
#pragma unroll
            for (int i = 0; i < 2; i++) {
                w_old[i] = *reinterpret_cast<const float4 *>(L.W + e_off[i]);
                v_old[i] = *reinterpret_cast<const float4 *>(L.V + e_off[i]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        multiply();
    } else
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        const int kc = (p.K - k0 < KC) ? p.K - k0 : KC; // a multiple of 16
        if (k0) __syncthreads();
        // 64 floats (16 float4) per row per operand; thread -> (row = idx / 16, q = idx % 16)
#pragma unroll
        for (int i = 0; i < KC * 16 / 512; i++) {
            const int idx = t + i * 512, k = idx >> 4, q = idx & 15;
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vd = va;
            if (interior) { // block-uniform
                va = *reinterpret_cast<const float4 *>(L.A + ((unsigned)(k0 + k) * (unsigned)L.lda + m0 + q * 4));
                vd = *reinterpret_cast<const float4 *>(L.D + ((unsigned)(k0 + k) * (unsigned)L.ldd + n0 + q * 4));
            } else if (k < kc) {
                size_t a_row = (size_t)(k0 + k);
                bool a_live = true;
                if (li == 0 && p.row_idx) {
                    a_live = k0 + k < p.k_true;
                    a_row = a_live ? (size_t)p.row_idx[k0 + k] : 0;
                }
                if (a_live && m0 + q * 4 < L.M) va = *reinterpret_cast<const float4 *>(L.A + a_row * L.lda + m0 + q * 4);
                if (n0 + q * 4 < L.N) vd = *reinterpret_cast<const float4 *>(L.D + (size_t)(k0 + k) * L.ldd + n0 + q * 4);
            }
            *reinterpret_cast<float4 *>(&As[k * LD + q * 4]) = va;
            *reinterpret_cast<float4 *>(&Ds[k * LD + q * 4]) = vd;
        }
        __syncthreads();
        // this wave's half of the chunk; MFMA tile i holds rows wm*32 + 2*rho + i, tile j columns wn*32 + 2*gamma + j
        const float *ap = &As[fq * LD + wm * 32 + fr * 2];
        const float *dp = &Ds[fq * LD + wn * 32 + fr * 2];
        const int kk_end = (kh + 1) * (KC / 2) < kc ? (kh + 1) * (KC / 2) : kc;
        for (int kk = kh * (KC / 2); kk < kk_end; kk += 16) { // kc, KC/2 are multiples of 16
            f32x2 a[4], d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                a[u] = *reinterpret_cast<const f32x2 *>(ap + (kk + 4 * u) * LD);
                d[u] = *reinterpret_cast<const f32x2 *>(dp + (kk + 4 * u) * LD);
            }
#pragma unroll
            for (int u = 0; u < 4; u++)
#pragma unroll
                for (int i = 0; i < 2; i++)
#pragma unroll
                    for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], d[u][j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads(); // the panels are free: partial tile of K half kh at sm[kh*T*CLD]
    float *Cs = sm + kh * (T * CLD);
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = wm * 32 + (fq * 4 + r) * 2 + i;
            *reinterpret_cast<f32x2 *>(&Cs[row * CLD + wn * 32 + fr * 2]) = (f32x2){acc[i][0][r], acc[i][1][r]};
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (!e_ok[i]) continue;
        const float4 g0 = *reinterpret_cast<const float4 *>(&sm[(er + 32 * i) * CLD + eq * 4]);
        const float4 g1 = *reinterpret_cast<const float4 *>(&sm[T * CLD + (er + 32 * i) * CLD + eq * 4]);
        const float4 gsum = make_float4(g0.x + g1.x, g0.y + g1.y, g0.z + g1.z, g0.w + g1.w);
        if (FUSED) { // ((step*G)/B) + (momentum*prev), SCE:333
            float4 adj, wn_;
            adj.x = sgd_adj(p.step_over_b, gsum.x, p.momentum, v_old[i].x);
            adj.y = sgd_adj(p.step_over_b, gsum.y, p.momentum, v_old[i].y);
            adj.z = sgd_adj(p.step_over_b, gsum.z, p.momentum, v_old[i].z);
            adj.w = sgd_adj(p.step_over_b, gsum.w, p.momentum, v_old[i].w);
            wn_.x = w_old[i].x - adj.x; wn_.y = w_old[i].y - adj.y; wn_.z = w_old[i].z - adj.z; wn_.w = w_old[i].w - adj.w;
            *reinterpret_cast<float4 *>(L.W + e_off[i]) = wn_;
            *reinterpret_cast<float4 *>(L.V + e_off[i]) = adj;
        } else {
            *reinterpret_cast<float4 *>(L.G + e_off[i]) = gsum;
        }
    }
}

// ------------------------------------------------------------------------------------------
// tail_kernel: for nets with at most 16 outputs (padded width 16) the LAST layer, the output rule and the
// first backward product in one launch, per block of 16 batch rows:
//   z = A_{L-2} . W_{L-2}           K split over the 8 waves in 16-wide chunks (as fwd_first_kernel)
//   p = softmax(z), delta_{L-1} = p - y, loss, `>=` argmax          (SCE:357-376, 249-251, 213-217; MT:166-168)
//   delta_{L-2} = (delta_{L-1} . W_{L-2}^T) * f'(a_{L-2})           (SCE:272-278; only when layer L-2 is hidden)
// Off the row-block path these were three launches of ~4 us each around a few KB of data.
// ------------------------------------------------------------------------------------------
struct TailParams {
    const float *A; int lda;      // activations of layer L-2 [rows][lda] (the inputs when L = 2)
    const float *W;               // W_{L-2} [K][16]
    const float *Y; int ldy;      // expected rows (may be null)
    float *prob; float *delta_out; // [rows][16], may be null
    float *loss; int32_t *label;  // [rows], may be null
    float *delta_prev; int ldp;   // delta_{L-2} [rows][ldp], null = not wanted (forward only, or L = 2)
    int K;                        // padded width of layer L-2
    int k_true;                   // its logical width
    int B, n_true;                // live rows, logical outputs
    int act;                      // inner activation (for f')
    __bf16 *delta_out_b;          // BF16 form: the bf16 roundings of delta_out / delta_prev (same strides; may be null)
    __bf16 *delta_prev_b;
};

// The value a bf16 GEMM operand has (GNN_DTYPE_BF16's contract: every product operand rounded to bf16, RNE -- the same conversion
// that writes actb / Wb / deltab); a product of two such values is exact in f32, so the f32 MFMA below IS bf16 x bf16 -> f32.
__device__ __forceinline__ float as_bf16_operand(float x) { return (float)(__bf16)x; }

// grid = (row blocks, column splits): every workgroup of a row block forms the logits and the output rule (redundantly:
// 16 x K x 16 MACs), split 0 stores them, and each split makes its share of delta_{L-2}'s columns -- 16 workgroups for a
// 256-row batch became 128, and the column tiles' operands are requested together instead of one dependent round trip per
// tile (784-1024^3-10 at 256 rows: 12.3 -> see profiles/r02).
// BF16: the two products take bf16-rounded operands (A_{L-2}, W_{L-2}, delta_{L-1}); f' is taken from the unrounded activation and
// the bf16 copies of both deltas are written beside the f32 ones, as the bf16 per-layer path's other kernels do (gemm_bf16.h).
template <bool BF16 = false>
__global__ __launch_bounds__(512) void tail_kernel(TailParams p) {
    constexpr int NW = 8, MAXC = 8, RLD = 20;
    __shared__ __attribute__((aligned(16))) float red[NW * 16 * RLD];
    __shared__ __attribute__((aligned(16))) float dl[16 * RLD];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.x * 16;
    const int k16 = p.K / 16;
    const int c_begin = (int)((long)wave * k16 / NW), c_end = (int)((long)(wave + 1) * k16 / NW);
    const float *arow = p.A + (size_t)(m0 + fr) * p.lda + 4 * fq;
    const float *wcol = p.W + (size_t)(4 * fq) * 16 + fr;

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    for (int cb = c_begin; cb < c_end; cb += MAXC) {
        float4 a[MAXC];
        float b[MAXC][4];
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
            const int c = cb + i;
            if (c < c_end) {
                a[i] = *reinterpret_cast<const float4 *>(arow + c * 16);
                const float *w = wcol + (size_t)(c * 16) * 16;
                b[i][0] = w[0]; b[i][1] = w[16]; b[i][2] = w[32]; b[i][3] = w[48];
                if constexpr (BF16) {
                    a[i].x = as_bf16_operand(a[i].x); a[i].y = as_bf16_operand(a[i].y); a[i].z = as_bf16_operand(a[i].z); a[i].w = as_bf16_operand(a[i].w);
#pragma unroll
                    for (int j = 0; j < 4; j++) b[i][j] = as_bf16_operand(b[i][j]);
                }
            } else {
                a[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                b[i][0] = b[i][1] = b[i][2] = b[i][3] = 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < MAXC; i++) {
            f32x4 &acc = (i & 1) ? acc1 : acc0;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[i][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[i][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[i][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[i][3], acc, 0, 0, 0);
        }
    }
    // The operands of this wave's first delta_{L-2} tiles (W rows, the activations f' is taken from) depend on nothing computed
    // here: requested NOW, their round trip runs under the reduction, the barrier and the output rule instead of after them.
    constexpr int TB = 4;
    const int nt_begin = (int)((long)blockIdx.y * k16 / gridDim.y), nt_end = (int)((long)(blockIdx.y + 1) * k16 / gridDim.y);
    f32x4 wb[TB];
    float av[TB][4];
    auto load_tiles = [&](int base) {
#pragma unroll
        for (int i = 0; i < TB; i++) {
            const int nt = base + i * NW;
            const int n = (nt < nt_end ? nt : base) * 16 + fr; // (a tile past the end re-reads the first: never used)
            wb[i] = *reinterpret_cast<const f32x4 *>(p.W + (size_t)n * 16 + 4 * fq); // W[n][4fq..4fq+3]
            if constexpr (BF16) {
#pragma unroll
                for (int j = 0; j < 4; j++) wb[i][j] = as_bf16_operand(wb[i][j]);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) av[i][r] = p.A[(size_t)(m0 + fq * 4 + r) * p.lda + n];
        }
    };
    const bool have_first = p.delta_prev && nt_begin + wave < nt_end; // (wave-uniform)
    if (have_first) load_tiles(nt_begin + wave);

    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int r = 0; r < 4; r++) red[(wave * 16 + fq * 4 + r) * RLD + fr] = acc[r];
    __syncthreads();

    // ---- output rule: thread (m = t/4, q = t%4) holds logits 4q..4q+3 of row m; a row is one quad of lanes ----
    if (t < 64) {
        const int m = t >> 2, q = t & 3;
        const int row = m0 + m;
        const bool live_row = row < p.B;
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; w++) z += *reinterpret_cast<const f32x4 *>(&red[(w * 16 + m) * RLD + q * 4]);
        float4 yv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.Y && live_row) yv = *reinterpret_cast<const float4 *>(p.Y + (size_t)row * p.ldy + q * 4);
        const float yy[4] = {yv.x, yv.y, yv.z, yv.w};
        // `>=` scan in ascending index (MT:166-168): ties -> highest index; NaN anywhere -> label 0 (see output_layer_kernel)
        float mx = -__builtin_inff();
        int best = -1;
        bool has_nan = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = 4 * q + j;
            if (c < p.n_true) {
                has_nan |= (z[j] != z[j]);
                if (z[j] >= mx) { mx = z[j]; best = c; }
            }
        }
#pragma unroll
        for (int o = 1; o <= 2; o <<= 1) {
            const float ov = __shfl_xor(mx, o);
            const int ob = __shfl_xor(best, o);
            if (ov > mx || (ov == mx && ob > best)) { mx = ov; best = ob; }
            has_nan |= (__shfl_xor((int)has_nan, o) != 0);
        }
        float e[4], s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            e[j] = (4 * q + j < p.n_true) ? __expf(z[j] - mx) : 0.f;
            s += e[j];
        }
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        const float inv = 1.f / s, lse = mx + __logf(s);
        float l = 0.f;
        f32x4 pr, dd;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool live = live_row && 4 * q + j < p.n_true;
            pr[j] = live ? e[j] * inv : 0.f;
            const float y1 = live ? yy[j] : 0.f;
            dd[j] = live ? pr[j] - y1 : 0.f;                     // SCE:250
            if (live && y1 != 0.f) l += y1 * (lse - z[j]);       // -y ln p, SCE:216
        }
        l += __shfl_xor(l, 1);
        l += __shfl_xor(l, 2);
        const bool writer = blockIdx.y == 0;
        if (p.prob && writer) *reinterpret_cast<f32x4 *>(p.prob + (size_t)row * 16 + q * 4) = pr;
        if (p.delta_out && writer) *reinterpret_cast<f32x4 *>(p.delta_out + (size_t)row * 16 + q * 4) = dd;
        if constexpr (BF16) {
            typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
            const bf16x4_t db = {(__bf16)dd[0], (__bf16)dd[1], (__bf16)dd[2], (__bf16)dd[3]};
            if (p.delta_out_b && writer) *reinterpret_cast<bf16x4_t *>(p.delta_out_b + (size_t)row * 16 + q * 4) = db;
#pragma unroll
            for (int j = 0; j < 4; j++) dd[j] = (float)db[j]; // the backward product's operand
        }
        *reinterpret_cast<f32x4 *>(&dl[m * RLD + q * 4]) = dd;
        if (has_nan) best = 0;
        if (q == 0 && writer) {
            if (p.loss) p.loss[row] = live_row ? l : 0.f;
            if (p.label) p.label[row] = live_row ? best : -1;
        }
    }
    if (!p.delta_prev) return;
    __syncthreads();

    // ---- delta_{L-2}[16 x K]: this split's 16-column tiles, dealt to the waves; k = 4*fq + i for the i-th MFMA on both
    // operands.  Up to TB tiles per trip, every operand of the trip requested before the first is used (the first trip's: above).
    const f32x4 da = *reinterpret_cast<const f32x4 *>(&dl[fr * RLD + 4 * fq]); // delta_{L-1}[m = fr][4fq..4fq+3]
    for (int base = nt_begin + wave; base < nt_end; base += NW * TB) {
        if (base != nt_begin + wave) load_tiles(base);
#pragma unroll
        for (int i = 0; i < TB; i++) {
            const int nt = base + i * NW;
            if (nt >= nt_end) break; // wave-uniform
            const int n = nt * 16 + fr;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(da[j], wb[i][j], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = m0 + fq * 4 + r;
                const bool live = row < p.B && n < p.k_true;
                const float dv = live ? c[r] * act_prime_from_a(p.act, av[i][r]) : 0.f;
                p.delta_prev[(size_t)row * p.ldp + n] = dv;
                if constexpr (BF16) { if (p.delta_prev_b) p.delta_prev_b[(size_t)row * p.ldp + n] = (__bf16)dv; }
            }
        }
    }
}

} // namespace gnn
