// gemm_f32_dma.h -- gemm_f32_kernel's three forms with the operand tiles brought into LDS by DMA (global_load_lds_dwordx4), as
// gemm_bf16_dma.h does for bf16: three images, one barrier per tile of K, two tiles in flight, no ds_write and no register
// staging.  gemm_f32_kernel writes a 32-KB tile pair per step with ds_write_b32 (the transposing stores of a k-contiguous
// operand, 64 B/clk per CU) and ds_write_b128 (~79 B/clk): ~460 cycles per step in which its waves do not multiply, beside
// 2 048 cycles of f32 MFMAs -- tools/gemm_probe 41 (the loop WITHOUT its LDS writes, timing only) puts that at 6-22 % of the
// products of configs[3] (backward data, both operands transposed on the way in, the most).
//
// Images (16-B pieces = 4 floats, XOR-permuted as in gemm_bf16_dma.h because a DMA instruction writes consecutive LDS):
//   k-contiguous operand: [row][BK], NOT transposed.  A fragment read is one ds_read_b128 -- lane (fr, fq) takes
//       k = kk + 4 fq .. + 3 of its row -- and serves FOUR MFMAs: step j of a 16-k block multiplies k = kk + 4 fq + j on slot fq
//       (any dealing of k to slots is a valid product when both operands use it).  Piece c of row r at c ^ kc_swz(r).
//       Image row 16 i + fr of a wave's block holds the operand's row base + fr * T + i (T = MFMA tiles per wave along that
//       edge): the rows are INTERLEAVED on the way in, so that the accumulators come out in gemm_f32_kernel's arrangement
//       (a lane owns TN consecutive columns) and the epilogue is shared.
//   k-major operand: [BK][W] as it lies in memory.  Step j reads row kk + 4 fq + j: T consecutive floats per lane (b32 / b64 /
//       b128).  A 32-lane group of such a read spans two rows 4 apart: they are moved apart by XOR-ing the piece index with
//       bit (k >> 2) & 1 shifted to where that read's banks want it (b32: 64 B, b64: 128 B; b128 reads need nothing).
// Summation order: an MFMA adds k = kk + j, kk + 4 + j, kk + 8 + j, kk + 12 + j where gemm_f32_kernel's adds kk .. kk + 3 --
// another order of the same f32 sums (the parity tests' bounds hold for both; shapes choose the kernel, so equal calls still
// give equal bits).  Whole tiles only (M % BM == 0, N % BN == 0, K % BK == 0).
#pragma once
#include "gemm_bf16_dma.h" // lds_dma16, wait_vmcnt, IntC

namespace gnn {

template <int BM> struct GemmF32DmaDepth { static constexpr int BK = BM <= 64 ? 64 : 32; };
template <int BM, int BN, int NIMG> constexpr size_t gemm_f32_dma_lds_bytes() { return (size_t)NIMG * (BM + BN) * GemmF32DmaDepth<BM>::BK * 4; }

// k-contiguous image, rows of 4 BK bytes: piece position = piece ^ this
template <int BK> __device__ __forceinline__ constexpr int f32_kc_swz(int r) { return BK >= 64 ? (r & 15) : ((r >> 1) & 7); }
// k-major image read T floats at a time
template <int T> __device__ __forceinline__ constexpr int f32_km_swz(int k) { return T >= 4 ? 0 : T == 2 ? (((k >> 2) & 1) << 3) : (((k >> 2) & 1) << 2); }

template <int T> struct F32Vec { typedef float type __attribute__((ext_vector_type(T))); };
template <> struct F32Vec<1> { typedef float type; };

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 4, int NIMG = 3>
__global__ __launch_bounds__(WM * 128) void gemm_f32_dma_kernel(GNN_GEMM_HEAD_PARAMS(float), GemmParams p) {
    GNN_GEMM_TAKE_HEAD(p);
    constexpr int BK = GemmF32DmaDepth<BM>::BK, NW = WM * 2;
    constexpr int TM = BM / (WM * 16), TN = BN / 32; // 16x16 MFMA tiles per wave (waves are WM x 2)
    constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 4, IMG_BYTES = A_BYTES + B_BYTES;
    constexpr int NIA = A_BYTES / 1024, NI = IMG_BYTES / 1024, NPW = (NI + NW - 1) / NW;
    constexpr int PD = NIMG - 1;
    static_assert(NIMG == 2 || NIMG == 3, "two or three images");
    static_assert(TM == 1 || TM == 2 || TM == 4, "MFMA tiles per wave along M");
    static_assert(TN == 1 || TN == 2 || TN == 4, "MFMA tiles per wave along N");
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole DMA instructions");
    extern __shared__ __attribute__((aligned(1024))) float gemm_f32_dma_smem[];
    float *smem = gemm_f32_dma_smem;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;

    // ---- this wave's DMA instructions of a tile (gemm_bf16_dma.h): lane addresses, LDS destinations, steps along K
    const char *src[NPW];
    unsigned dst[NPW];
    long long step[NPW];
#pragma unroll
    for (int j = 0; j < NPW; j++) {
        int i = wave + NW * j;
        if (i >= NI) i = NI - 1;
        const bool is_a = i < NIA; // (wave-uniform)
        const int li = is_a ? i : i - NIA;
        const int P = li * 1024 + lane * 16; // this lane's byte position in the operand's image
        auto kc_elem = [&](auto T_, int tile0, int ld) { // k-contiguous image (rows interleaved within a wave's block of T * 16)
            constexpr int T = decltype(T_)::value;
            const int R = P / (BK * 4), s = (P % (BK * 4)) / 16;
            const int blk = R / (T * 16), rem = R % (T * 16);
            const int row = blk * (T * 16) + (rem % 16) * T + rem / 16;
            return (size_t)(tile0 + row) * ld + (s ^ f32_kc_swz<BK>(R)) * 4;
        };
        auto km_elem = [&](auto T_, auto W_, int tile0, int ld) {
            constexpr int T = decltype(T_)::value, W = decltype(W_)::value;
            const int k = P / (W * 4), s = (P % (W * 4)) / 16;
            return (size_t)k * ld + tile0 + (s ^ f32_km_swz<T>(k)) * 4;
        };
        size_t e;
        if (is_a) e = A_KC ? kc_elem(IntC<TM>{}, m0, p.lda) : km_elem(IntC<TM>{}, IntC<BM>{}, m0, p.lda);
        else e = B_KC ? kc_elem(IntC<TN>{}, n0, p.ldb) : km_elem(IntC<TN>{}, IntC<BN>{}, n0, p.ldb);
        src[j] = reinterpret_cast<const char *>(is_a ? p.A : p.B) + 4 * e;
        step[j] = is_a ? (A_KC ? 4LL * BK : 4LL * BK * p.lda) : (B_KC ? 4LL * BK : 4LL * BK * p.ldb);
        dst[j] = (is_a ? 0u : (unsigned)A_BYTES) + (unsigned)li * 1024u;
    }
    auto issue = [&](int img) {
#pragma unroll
        for (int j = 0; j < NPW; j++) {
            lds_dma16(src[j], lds0 + (unsigned)img * (unsigned)IMG_BYTES + dst[j]);
            src[j] += step[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragments of a 16-k block: av[i][j] / bv[jj][j] = the value MFMA step j takes for tile i / jj
    typedef typename F32Vec<TM>::type vec_m;
    typedef typename F32Vec<TN>::type vec_n;
    struct Frags { float a[TM][4], b[TN][4]; };
    auto multiply = [&](auto IMG_) {
        constexpr int IMG = decltype(IMG_)::value;
        const float *As = smem + IMG * (IMG_BYTES / 4), *Bs = As + A_BYTES / 4;
        auto read_block = [&](int kk, Frags &f) {
            if constexpr (A_KC) {
#pragma unroll
                for (int i = 0; i < TM; i++) {
                    const int R = wm * (TM * 16) + 16 * i + fr;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(As + R * BK + (((kk >> 2) + fq) ^ f32_kc_swz<BK>(R)) * 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) f.a[i][j] = v[j];
                }
            } else {
                const int col = wm * (TM * 16) + fr * TM;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = kk + 4 * fq + j;
                    const float *q = As + k * BM + (((col >> 2) ^ f32_km_swz<TM>(k)) << 2) + (col & 3);
                    if constexpr (TM == 1) f.a[0][j] = *q;
                    else {
                        const vec_m v = *reinterpret_cast<const vec_m *>(q);
#pragma unroll
                        for (int i = 0; i < TM; i++) f.a[i][j] = v[i];
                    }
                }
            }
            if constexpr (B_KC) {
#pragma unroll
                for (int jj = 0; jj < TN; jj++) {
                    const int R = wn * (TN * 16) + 16 * jj + fr;
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(Bs + R * BK + (((kk >> 2) + fq) ^ f32_kc_swz<BK>(R)) * 4);
#pragma unroll
                    for (int j = 0; j < 4; j++) f.b[jj][j] = v[j];
                }
            } else {
                const int col = wn * (TN * 16) + fr * TN;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int k = kk + 4 * fq + j;
                    const float *q = Bs + k * BN + (((col >> 2) ^ f32_km_swz<TN>(k)) << 2) + (col & 3);
                    if constexpr (TN == 1) f.b[0][j] = *q;
                    else {
                        const vec_n v = *reinterpret_cast<const vec_n *>(q);
#pragma unroll
                        for (int jj = 0; jj < TN; jj++) f.b[jj][j] = v[jj];
                    }
                }
            }
        };
        auto mfma_block = [&](const Frags &f) {
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int jj = 0; jj < TN; jj++)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.a[i][j], f.b[jj][j], acc[i][jj], 0, 0, 0);
        };
        Frags f0, f1; // the next 16-k block's fragments are read while this block's MFMAs issue
        read_block(0, f0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            if (kk + 16 < BK) read_block(kk + 16, f1);
            mfma_block(f0);
            if (kk + 16 < BK) {
                if (kk + 32 < BK) read_block(kk + 32, f0);
                mfma_block(f1);
            }
        }
    };

    // ---- main loop (gemm_bf16_dma.h): image i % NIMG holds tile i; tiles i + 1 .. i + PD are in flight or landed
    const int nt = p.K / BK;
#pragma unroll
    for (int d = 0; d < PD; d++)
        if (d < nt) issue(d);
    auto tile = [&](auto IMG_, int i) {
        constexpr int IMG = decltype(IMG_)::value;
        if (NIMG == 3 && i + 1 < nt) wait_vmcnt<NPW>();
        else wait_vmcnt<0>();
        __syncthreads();
        if (i + PD < nt) issue((IMG + PD) % NIMG);
        multiply(IMG_);
    };
    for (int i = 0; i < nt; i += NIMG) {
        tile(IntC<0>{}, i);
        if (i + 1 < nt) tile(IntC<1>{}, i + 1);
        if (NIMG == 3 && i + 2 < nt) tile(IntC<NIMG == 3 ? 2 : 0>{}, i + 2);
    }
    gemm_f32_epilogue<TM, TN, EPI>(acc, p, m0, n0, wm, wn, fr, fq);
}

} // namespace gnn
