// gemm_bf16_dma.h -- the bf16 GEMM forms of gemm_bf16.h with the operand tiles brought from global memory STRAIGHT INTO LDS
// (global_load_lds_dwordx4, the gfx950 "LDS DMA": 64 lanes x 16 B land in 1 KB of consecutive LDS; no VGPRs, no ds_write).
//
// Why: gemm_bf16_kernel stages a tile through registers and writes it with ds_write_b128, which the LDS takes at ~79 B/clk per
// CU -- 32 KB per tile of K on a 64 x 64 tile (BK 128) = ~415 cycles in which no wave multiplies, a quarter of the loop; leaving
// the writes out (tools/gemm_probe 36, wrong results, timing only) takes 512 x 2048 x 4096 from 27.2 to 17.3 us.  The DMA has no
// such phase: the loop is  wait for tile i - barrier - issue tile i + 2 - multiply tile i  over THREE images in LDS, one
// barrier per tile, two tiles in flight.
//
// An instruction's 64 lanes write CONSECUTIVE 16-B pieces, so the images cannot be padded against bank conflicts; the pieces
// are permuted instead (every lane gives its own global address, so any piece can go anywhere):
//   k-contiguous image [row][BK] (rows of 2 BK bytes): piece c (8 k) of row r sits at piece position c ^ kc_swz(r).  A fragment
//       is one ds_read_b128 (lane (fr, fg): row fr, k = kk + 8 fg .. + 7); its 16-lane groups {0-3, 12-15, 20-27}, ... then fall on
//       16 different 4-bank groups (BK >= 128: position = c ^ (r & 15); BK 64, two rows per 64 banks: c ^ ((r >> 1) & 7)).
//   k-major image [BK][W] (rows of 2 W bytes, W = the tile edge): piece c (8 columns) of row rho at c ^ km_swz(rho), which moves
//       PAIRS of pieces (32 B: what a 16-lane group of ds_read_b64_tr_b16 reads of a row) so that the eight rows of a 32-lane
//       group cover all 64 banks.  In mixed products (one operand of each kind) the ROWS are permuted as well, as in
//       gemm_bf16.h: row 4 g + (e & 3) + 16 (e >> 2) of a 32-k block holds k = 8 g + e, and both operands deal k to the MFMA's
//       slots alike.
// Shapes: whole tiles only (M % BM == 0, N % BN == 0, K % BK == 0: the host checks; everything else takes gemm_bf16_kernel).
// The DMA is written as inline assembly: through the builtin the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of
// every barrier and LDS read that follows a DMA (it cannot tell the images apart) -- no tile would ever be in flight across a
// multiplication.  Unknown to the compiler, the instructions only make its own counts conservative (vmcnt retires in order).
#pragma once
#include "gemm_bf16.h"

namespace gnn {

// lds_addr: wave-uniform LDS byte address; lane l's 16 bytes land at lds_addr + 16 l.  M0 (the destination base) belongs to the
// compiler and is put back inside the statement; the s_nop is the wait state between an SALU write of M0 and the DMA reading it.
__device__ __forceinline__ void lds_dma16(const void *src, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_addr)
                 : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BK> __device__ __forceinline__ constexpr int kc_swz(int r) { return BK >= 128 ? (r & 15) : ((r >> 1) & 7); }
template <int W> __device__ __forceinline__ constexpr int km_swz(int rho) {
    return W >= 128 ? ((rho & 7) << 1) : W == 64 ? (((rho >> 1) & 3) << 1) : (((rho >> 2) & 1) << 1);
}
// k-major image of a mixed product: the k (inside its 32-k block) that image row rho holds
__device__ __forceinline__ constexpr int km_row_k(int rho) { return (rho & ~31) + 8 * ((rho >> 2) & 3) + 4 * ((rho >> 4) & 1) + (rho & 3); }

// (BKO: a K depth other than GemmBf16Depth's, a multiple of 64 -- tools/gemm_probe)
template <int BM, int BN, int BKO = 0> struct GemmBf16Dma {
    static constexpr int BK = BKO ? BKO : GemmBf16Depth<BM>::BK;
    static constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, IMG_BYTES = A_BYTES + B_BYTES;
    static_assert(A_BYTES % 1024 == 0 && B_BYTES % 1024 == 0, "whole DMA instructions");
};
template <int BM, int BN, int NIMG, int BKO = 0> constexpr size_t gemm_bf16_dma_lds_bytes() { return (size_t)NIMG * GemmBf16Dma<BM, BN, BKO>::IMG_BYTES; }
template <int V> struct IntC { static constexpr int value = V; };

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2, int NIMG = 3, int BKO = 0>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_dma_kernel(GNN_GEMM_HEAD_PARAMS(__bf16), GemmBf16Params p) {
    GNN_GEMM_TAKE_HEAD(p);
    using D = GemmBf16Dma<BM, BN, BKO>;
    constexpr int BK = D::BK, NW = WM * 2, NBLK = BK / 32;
    constexpr int TM = BM / (WM * 16), TN = BN / 32; // 16x16 MFMA tiles per wave (waves are WM x 2)
    constexpr int NIA = D::A_BYTES / 1024, NI = (D::A_BYTES + D::B_BYTES) / 1024, NPW = (NI + NW - 1) / NW; // DMA instructions: A's, all, per wave
    constexpr bool PERMUTE = (A_KC != B_KC);
    constexpr int PD = NIMG - 1; // tiles in flight ahead of the one being multiplied
    static_assert(NIMG >= 2 && NIMG <= 4, "two to four images");
    static_assert(TM >= 1 && TN >= 1, "tile too small for this many waves");
    extern __shared__ __attribute__((aligned(1024))) __bf16 gemm_bf16_dma_smem[];
    __bf16 *smem = gemm_bf16_dma_smem;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void *)smem;

    // ---- this wave's DMA instructions of a tile: number wave + NW j (past the last: the last again -- the same bytes to the
    // same place), each lane's global address, the wave's LDS destination, and the step from one tile of K to the next
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
        auto kc_elem = [&](int tile0, int ld) { // k-contiguous image: the global element this position holds (at k0 = 0)
            const int row = P / (BK * 2), s = (P % (BK * 2)) / 16;
            return (size_t)(tile0 + row) * ld + (s ^ kc_swz<BK>(row)) * 8;
        };
        auto km_elem = [&](auto W_, int tile0, int ld) {
            constexpr int W = decltype(W_)::value;
            const int rho = P / (W * 2), s = (P % (W * 2)) / 16;
            const int k = PERMUTE ? km_row_k(rho) : rho;
            return (size_t)k * ld + tile0 + (s ^ km_swz<W>(rho)) * 8;
        };
        size_t e;
        if (is_a) e = A_KC ? kc_elem(m0, p.lda) : km_elem(IntC<BM>{}, m0, p.lda);
        else e = B_KC ? kc_elem(n0, p.ldb) : km_elem(IntC<BN>{}, n0, p.ldb);
        src[j] = reinterpret_cast<const char *>(is_a ? p.A : p.B) + 2 * e;
        step[j] = is_a ? (A_KC ? 2LL * BK : 2LL * BK * p.lda) : (B_KC ? 2LL * BK : 2LL * BK * p.ldb);
        dst[j] = (is_a ? 0u : (unsigned)D::A_BYTES) + (unsigned)li * 1024u;
    }
    auto issue = [&](int img) {
#pragma unroll
        for (int j = 0; j < NPW; j++) {
            lds_dma16(src[j], lds0 + (unsigned)img * (unsigned)D::IMG_BYTES + dst[j]);
            src[j] += step[j];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- fragments (k slots as in gemm_bf16.h: both operands k-major: j < 4: k = kk + 4 fg + j, else kk + 16 + 4 fg + j - 4;
    // otherwise k = kk + 8 fg + j)
    auto frag_kc = [&](const __bf16 *img, int r0, int kk) {
        const int row = r0 + fr;
        return *reinterpret_cast<const bf16x8 *>(img + row * BK + (((kk >> 3) + fg) ^ kc_swz<BK>(row)) * 8);
    };
    auto frag_tr = [&](const __bf16 *img, auto W_, int c0, int kk) {
        constexpr int W = decltype(W_)::value;
        const int rho = kk + 4 * fg + (fr >> 2); // (km_swz is the same for rho and rho + 16)
        const __bf16 *q = img + rho * W + (((c0 >> 3) + ((fr & 3) >> 1)) ^ km_swz<W>(rho)) * 8 + (fr & 1) * 4;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + 16 * W));
        return join8(lo, hi);
    };
    auto multiply = [&](auto IMG_) {
        constexpr int IMG = decltype(IMG_)::value;
        const __bf16 *As = smem + IMG * (D::IMG_BYTES / 2), *Bs = As + D::A_BYTES / 2;
        auto read_block = [&](int kk, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; i++)
                a[i] = A_KC ? frag_kc(As, wm * (TM * 16) + i * 16, kk) : frag_tr(As, IntC<BM>{}, wm * (TM * 16) + i * 16, kk);
#pragma unroll
            for (int j = 0; j < TN; j++)
                b[j] = B_KC ? frag_kc(Bs, wn * (TN * 16) + j * 16, kk) : frag_tr(Bs, IntC<BN>{}, wn * (TN * 16) + j * 16, kk);
        };
        auto mfma_block = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        };
        bf16x8 a0[TM], b0[TN], a1[TM], b1[TN]; // the next 32-k block's fragments are read while this block's MFMAs issue
        read_block(0, a0, b0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 64) {
            if (kk + 32 < BK) read_block(kk + 32, a1, b1);
            mfma_block(a0, b0);
            if (kk + 32 < BK) {
                if (kk + 64 < BK) read_block(kk + 64, a0, b0);
                mfma_block(a1, b1);
            }
        }
    };

    // ---- main loop: image i % NIMG holds tile i; tiles i + 1 .. i + PD are in flight or landed
    const int nt = p.K / BK;
#pragma unroll
    for (int d = 0; d < PD; d++)
        if (d < nt) issue(d);
    auto tile = [&](auto IMG_, int i) {
        constexpr int IMG = decltype(IMG_)::value;
        // all but the instructions of the tiles issued after tile i (at most PD - 1 of them): tile i has landed
        const int younger = nt - 1 - i;
        if (PD >= 3 && younger >= 2) wait_vmcnt<2 * NPW>();
        else if (PD >= 2 && younger >= 1) wait_vmcnt<NPW>();
        else wait_vmcnt<0>();
        __syncthreads(); // every wave's share of tile i is in LDS, and every wave is past its reads of tile i - 1 ...
        if (i + PD < nt) issue((IMG + PD) % NIMG); // ... whose image takes tile i + PD
        multiply(IMG_);
    };
    for (int i = 0; i < nt; i += NIMG) {
        tile(IntC<0>{}, i);
        if (i + 1 < nt) tile(IntC<1>{}, i + 1);
        if (NIMG >= 3 && i + 2 < nt) tile(IntC<NIMG >= 3 ? 2 : 0>{}, i + 2);
        if (NIMG >= 4 && i + 3 < nt) tile(IntC<NIMG >= 4 ? 3 : 0>{}, i + 3);
    }
    __syncthreads(); // the epilogue stages through the images

    static_assert(2 * WM * 16 * (TN * 16 + 4) * sizeof(float) <= gemm_bf16_dma_lds_bytes<BM, BN, NIMG, BKO>(), "epilogue staging fits the images");
    gemm_bf16_epilogue<TM, TN, EPI>(acc, p, reinterpret_cast<float *>(smem), m0, n0, wave, lane);
}

} // namespace gnn
