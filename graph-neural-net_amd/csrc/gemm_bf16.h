// gemm_bf16.h -- GNN_DTYPE_BF16: the three GEMM forms of kernels.h with bf16 OPERANDS IN HBM and
// f32 accumulation on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, 16x the f32 MFMA rate).
//
// What lives where (bf16 mode):
//   W, V (momentum), G      f32 masters, as in f32 mode (the update is SCE:333-339 in f32)
//   Wb                      bf16 shadow of W, same padded [in][out] layout, rewritten by whatever updates W
//   act[l] / actb[l]        f(z_l) in f32 (f' is taken from it, host exports) and its bf16 rounding (GEMM operand)
//   delta[l] / deltab[l]    dE/dz_l in f32 and bf16
// Every producer rounds ONCE (v_cvt_pk_bf16_f32, round-to-nearest-even) in its epilogue, so a GEMM
// moves half the operand bytes and converts nothing.  Arithmetic contract (what the bf16 parity tests restate
// in fp64): every GEMM operand rounded to bf16, products exact, f32 accumulation.
//
// All three products read their operands in the NATURAL layouts (no transposed copies in HBM):
//   forward   Z  = A . W        A [B][d_in] k-contiguous,   W [d_in][d_out] k-major
//   backward  D' = D . W^T      D [B][d_out] k-contiguous,  W as stored is k-contiguous for this product
//   gradient  G  = A^T . D      both operands k-major ([B][d] with k = batch row)
// A k-contiguous tile is staged as [row][k] (row stride BK + 16 bf16 = 8 banks mod 16) and a fragment is ONE ds_read_b128
// (lane (fr, fg): k = kk + 8 fg .. + 7 of row fr; see LDK in the kernel for why not two ds_read_b64); a k-major tile is staged
// AS IT IS, [k][n] with a row stride of 32*odd bytes, and read with ds_read_b64_tr_b16, the gfx950 transpose read (per 16
// lanes: a 4 x 16 block, lane i gets column i).  So that those reads are conflict-free the k values of one MFMA are dealt to
// the four 16-lane groups as  element j < 4: k = 4g + j,  j >= 4: k = 16 + 4g + (j - 4)  -- any assignment of distinct k to
// slots is a valid product as long as both operands use the same one; a 32-lane half then reads rows 0..7 (16..23) of the
// block, eight rows x 32 B = all 64 banks once.  A product with one operand of each kind (forward) writes the k-major image
// with its rows permuted so that the same reads deliver k = 8g + j, the k-contiguous operand's assignment.
// gemm_bf16_dma.h holds the form that brings whole tiles into LDS by DMA; this kernel takes every shape.
#pragma once
#include "kernels.h"

// tools/gemm_probe defines GNN_GEMM_BF16_STAMPS: wave 0 of workgroup (0, 0) adds up the cycles between the marks of the main loop
#ifdef GNN_GEMM_BF16_STAMPS
__device__ unsigned long long gnn_bf16_stamps[16];
#define GNN_STAMP_DECL unsigned long long st_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, st_prev = __builtin_readcyclecounter(); const unsigned long long st_begin = st_prev
#define GNN_STAMP(i) do { asm volatile("" ::: "memory"); const unsigned long long st_now = __builtin_readcyclecounter(); st_sum[i] += st_now - st_prev; st_prev = st_now; } while (0)
#define GNN_STAMP_FLUSH do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { for (int i = 0; i < 9; i++) gnn_bf16_stamps[i] = st_sum[i]; gnn_bf16_stamps[9] = __builtin_readcyclecounter() - st_begin; } } while (0)
#else
#define GNN_STAMP_DECL
#define GNN_STAMP(i)
#define GNN_STAMP_FLUSH
#endif

namespace gnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct GemmBf16Params {
    const __bf16 *A; int lda;
    const __bf16 *B; int ldb;
    float *C; int ldc;            // f32 result (may be null when only the bf16 copy is wanted)
    __bf16 *Cb;                   // bf16 rounding of what C receives, same ld (may be null)
    int M, N, K;                  // padded extents (multiples of 16)
    int m_true, n_true;           // logical extents (zeros are stored beyond them)
    const float *aux; int ldaux;  // EPI_DACT: a = f(z) of this layer (f32)
    float *W; float *V; __bf16 *Wb; // EPI_SGD: masters and the bf16 shadow (same ld as C)
    float step_over_b, momentum;
    int act;
};

__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// Fragment of a 16x16x32 bf16 MFMA from a k-major LDS image img[k][c] (row stride ldt elements, 32*odd bytes):
// the 16 columns c0..c0+15 of the 32-wide k block at kk, k dealt to the lane groups as described above.
// ds_read_b64_tr_b16: lane 4q + p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4 x 16 block and
// receives column (lane & 15) of its four rows.  EXEC must be all ones (every lane of the wave executes this).
__device__ __forceinline__ bf16x8 tr_frag(const __bf16 *img, int ldt, int c0, int kk, int lane) {
    const int fr = lane & 15, fg = lane >> 4;
    const __bf16 *q = img + (kk + 4 * fg + (fr >> 2)) * ldt + c0 + 4 * (fr & 3);
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4 *)(q + 16 * ldt));
    return join8(lo, hi);
}

// ---- epilogue (shared by gemm_bf16_kernel and gemm_bf16_dma_kernel) ------------------------------------
// acc[i][j]: the 16 x 16 MFMA tile at rows m0 + wm * (TM * 16) + i * 16, columns n0 + wn * (TN * 16) + j * 16.  `smem`: LDS the
// waves may overwrite (every wave is past its last operand read): 2 * WM * 16 * (TN * 16 + 4) floats.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void gemm_bf16_epilogue(f32x4 (&acc)[TM][TN], const GemmBf16Params &p, float *smem, int m0, int n0, int wave, int lane) {
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;
    // The accumulator of a 16x16 MFMA tile has its column on the lane (lane & 15) and four rows in the registers:
    // stored as it stands that is 4 B per lane (64-B segments; W, V, Wb of the update: five such accesses per
    // element).  Each wave therefore turns one row of its MFMA tiles (16 rows x TN*16 columns) at a time through a
    // private LDS area and handles it as 16-B pieces: lane -> (row, 4 consecutive columns), 256-B runs per row.
    constexpr int EW = TN * 16, ELD = EW + 4, C4 = EW / 4, RPP = 64 / C4, PASSES = 16 / RPP; // float4s per row, rows per pass
    float *est = smem + wave * (16 * ELD);
    const int erow = lane / C4, ec4 = lane % C4;
#pragma unroll
    for (int i = 0; i < TM; i++) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < TN; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) est[(fg * 4 + r) * ELD + j * 16 + fr] = acc[i][j][r];
        __builtin_amdgcn_wave_barrier(); // one wave, LDS operations complete in order: a scheduling fence is enough
        asm volatile("" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < PASSES; ps++) {
            const int row = ps * RPP + erow;
            const int m = m0 + wm * (TM * 16) + i * 16 + row;
            const int n = n0 + wn * EW + ec4 * 4;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(est + row * ELD + ec4 * 4);
            if (m < p.M && n < p.N) { // N is a multiple of 16: the four columns are all inside or all outside
                const size_t off = (size_t)m * p.ldc + n;
                f32x4 out = {0.f, 0.f, 0.f, 0.f};
                if (EPI == EPI_SGD) { // on the f32 masters, SCE:333; the bf16 shadow follows the master
                    f32x4 w = *reinterpret_cast<const f32x4 *>(p.W + off);
                    f32x4 vv = *reinterpret_cast<const f32x4 *>(p.V + off);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        if (m < p.m_true && n + e < p.n_true) {
                            const float adj = sgd_adj(p.step_over_b, v[e], p.momentum, vv[e]);
                            w[e] -= adj;
                            vv[e] = adj;
                        }
                    }
                    *reinterpret_cast<f32x4 *>(p.W + off) = w;
                    *reinterpret_cast<f32x4 *>(p.V + off) = vv;
                    *reinterpret_cast<bf16x4 *>(p.Wb + off) = (bf16x4){(__bf16)w[0], (__bf16)w[1], (__bf16)w[2], (__bf16)w[3]};
                } else {
                    f32x4 aux = {0.f, 0.f, 0.f, 0.f};
                    if (EPI == EPI_DACT) aux = *reinterpret_cast<const f32x4 *>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const bool live = (m < p.m_true) && (n + e < p.n_true);
                        if (EPI == EPI_STORE) out[e] = live ? v[e] : 0.f;
                        else if (EPI == EPI_ACT) out[e] = live ? act_fn(p.act, v[e]) : 0.f;
                        else out[e] = live ? v[e] * act_prime_from_a(p.act, aux[e]) : 0.f;
                    }
                    if (p.C) *reinterpret_cast<f32x4 *>(p.C + off) = out;
                    if (p.Cb) *reinterpret_cast<bf16x4 *>(p.Cb + off) = (bf16x4){(__bf16)out[0], (__bf16)out[1], (__bf16)out[2], (__bf16)out[3]};
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// K depth of one staged tile.  A 4-wave workgroup keeps TWO tiles ahead in registers while it multiplies the
// staged one (the loads of tile t+2 are issued before tile t is multiplied); with one workgroup per CU (grids
// of 256..511 tiles) that is all the CU has in flight, and it must cover a memory round trip (~1 500 cycles
// under load) at the CU's load rate: 2 x 32 KB for 64 x 64 tiles (BK 128), 2 x 32 KB for 128 x 128 (BK 64).
// With ONE 16-KB tile ahead (BK 64) the 512-row products of 4096-2048-2048-1024 ran at 11 B/clk/CU.
template <int BM> struct GemmBf16Depth { static constexpr int BK = BM >= 128 ? 64 : BM >= 64 ? 128 : 256; };

// NSTG: 2 = one operand image in LDS, fragments one 32-k block ahead; 3 = TWO images (one barrier per tile, see the main loop),
// every fragment of a tile requested up front; 4 = one image, fragments up front; 5 = two images, fragments one block ahead
constexpr bool gemm_bf16_two_images(int nstg) { return nstg == 3 || nstg == 5; }
template <int BM, int BN, bool A_KC, bool B_KC, int NSTG = 2>
constexpr size_t gemm_bf16_lds_bytes() {
    constexpr int BK = GemmBf16Depth<BM>::BK, LDK = BK + 16;
    constexpr int LDTA = BM + ((BM / 16) % 2 == 0 ? 16 : 0), LDTB = BN + ((BN / 16) % 2 == 0 ? 16 : 0);
    return sizeof(__bf16) * ((A_KC ? BM * LDK : BK * LDTA) + (B_KC ? BN * LDK : BK * LDTB)) * (gemm_bf16_two_images(NSTG) ? 2 : 1);
}

// WM: wave rows (waves are WM x 2, WM * 128 threads); WM = 4 puts eight waves on a tile (see gemm_f32_kernel)
// (tools/gemm_probe only: NSTG_ = NSTG + 10 x ablation -- 1: no global loads inside the loop, 2: no MFMAs, 3: no LDS reads, 4: no LDS writes)
// (the body: one BM x BN tile of product `p`, tile column bx, tile row by -- gemm_bf16_kernel's grid gives them, the grouped
//  kernel below finds them from a flat workgroup number)
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int NSTG_, int WM>
__device__ __forceinline__ void gemm_bf16_tile(const GemmBf16Params &p, const int bx, const int by) {
    constexpr int NSTG = NSTG_ % 10, ABL = NSTG_ / 10;
    constexpr int NT = WM * 128;
    constexpr int BK = GemmBf16Depth<BM>::BK;
    constexpr int TM = BM / (WM * 16), TN = BN / 32; // 16x16 MFMA tiles per wave (waves are WM x 2)
    // k-contiguous image [row][k]: a fragment is ONE ds_read_b128 -- lane (fr, fg) takes k = kk + 8 fg .. + 7 of row fr.  (Two
    // ds_read_b64 per fragment, k = 4 fg + j and 16 + 4 fg + j at a row stride of BK + 8, were conflict-free as written -- and
    // the compiler merges such a pair into ds_read2_b64, which the LDS serves in 16-lane groups over 32 banks at HALF the rate,
    // where rows 8 apart collide: 16 LDS cycles per fragment instead of 4, and the multiplication of a tile ran at the LDS's
    // pace, 1 040 cycles for 256 cycles of MFMAs -- tools/gemm_probe 34.)  ds_read_b128 is served in the 16-lane groups
    // {0-3, 12-15, 20-27}, ... over 64 banks: a row stride of 8 banks mod 16 (BK + 16 elements = 72 or 40 banks) is conflict-free.
    constexpr int LDK = BK + 16;
    static_assert((LDK / 2) % 16 == 8, "row stride of the k-contiguous image: 8 banks mod 16");
    // Mixed products (forward: A k-contiguous, W k-major) need BOTH operands to deal k to the slots alike: the k-major image is
    // then written with its rows permuted inside every 32-k block -- k = 8 g + e sits in row 4 g + (e & 3) + 16 (e >> 2) -- so
    // that the transpose reads (rows 4 fg + j and 16 + 4 fg + j, conflict-free) deliver k = 8 fg + e as well.
    constexpr bool PERMUTE_KMAJOR = (A_KC != B_KC);
    // k-major image: row stride 32*odd bytes (BM = 128: 288 B, 64: 160 B, 32: 96 B)
    constexpr int LDTA = BM + ((BM / 16) % 2 == 0 ? 16 : 0), LDTB = BN + ((BN / 16) % 2 == 0 ? 16 : 0);
    constexpr int A_ELEMS = A_KC ? BM * LDK : BK * LDTA, B_ELEMS = B_KC ? BN * LDK : BK * LDTB;
    extern __shared__ __attribute__((aligned(16))) __bf16 gemm_bf16_smem[]; // up to 70 KB: dynamic (opt-in above 64 KB)
    __bf16 *As = gemm_bf16_smem, *Bs = gemm_bf16_smem + A_ELEMS;
    static_assert(sizeof(__bf16) * (A_ELEMS + B_ELEMS) * (gemm_bf16_two_images(NSTG) ? 2 : 1) == gemm_bf16_lds_bytes<BM, BN, A_KC, B_KC, NSTG>(), "LDS size");
    auto use_image = [&](int which) { As = gemm_bf16_smem + which * (A_ELEMS + B_ELEMS); Bs = As + A_ELEMS; }; // (NSTG == 3)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = by * BM, n0 = bx * BN;
    constexpr int NA = BM * BK / 8 / NT, NB = BN * BK / 8 / NT; // 16-B chunks (8 bf16) per thread per tile
    static_assert(NA >= 1 && NB >= 1 && TM >= 1, "tile too small for this many threads");
    bf16x8 ra0[NA], rb0[NB], ra1[NA], rb1[NB]; // register stages

    // interior tiles (wholly inside M x N, K a multiple of BK): unguarded loads through 32-bit element offsets (see gemm_f32_kernel)
    const bool fits32 = (unsigned long long)(A_KC ? p.M : p.K) * (unsigned)p.lda < 0xffffffffull &&
                        (unsigned long long)(B_KC ? p.N : p.K) * (unsigned)p.ldb < 0xffffffffull;
    const bool interior = fits32 && (m0 + BM <= p.M) && (n0 + BN <= p.N) && (p.K % BK == 0);
    unsigned oa[NA], ob[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int c = t + i * NT;
        oa[i] = A_KC ? (unsigned)(m0 + c / (BK / 8)) * (unsigned)p.lda + (c % (BK / 8)) * 8 : (unsigned)(c / (BM / 8)) * (unsigned)p.lda + m0 + (c % (BM / 8)) * 8;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int c = t + i * NT;
        ob[i] = B_KC ? (unsigned)(n0 + c / (BK / 8)) * (unsigned)p.ldb + (c % (BK / 8)) * 8 : (unsigned)(c / (BN / 8)) * (unsigned)p.ldb + n0 + (c % (BN / 8)) * 8;
    }
    const unsigned a_kstride = A_KC ? 1u : (unsigned)p.lda, b_kstride = B_KC ? 1u : (unsigned)p.ldb;
    auto load_tiles = [&](int k0, bf16x8 (&ra)[NA], bf16x8 (&rb)[NB], auto inside) {
        if constexpr (decltype(inside)::value) {
#pragma unroll
            for (int i = 0; i < NA; i++) ra[i] = *reinterpret_cast<const bf16x8 *>(p.A + (oa[i] + (unsigned)k0 * a_kstride));
#pragma unroll
            for (int i = 0; i < NB; i++) rb[i] = *reinterpret_cast<const bf16x8 *>(p.B + (ob[i] + (unsigned)k0 * b_kstride));
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int c = t + i * NT;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (A_KC) {
                const int row = c / (BK / 8), kq = c % (BK / 8);
                if (m0 + row < p.M && k0 + kq * 8 < p.K)
                    v = *reinterpret_cast<const bf16x8 *>(p.A + (size_t)(m0 + row) * p.lda + k0 + kq * 8);
            } else {
                const int k = c / (BM / 8), mq = c % (BM / 8);
                if (k0 + k < p.K && m0 + mq * 8 < p.M)
                    v = *reinterpret_cast<const bf16x8 *>(p.A + (size_t)(k0 + k) * p.lda + m0 + mq * 8);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int c = t + i * NT;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (B_KC) {
                const int row = c / (BK / 8), kq = c % (BK / 8);
                if (n0 + row < p.N && k0 + kq * 8 < p.K)
                    v = *reinterpret_cast<const bf16x8 *>(p.B + (size_t)(n0 + row) * p.ldb + k0 + kq * 8);
            } else {
                const int k = c / (BN / 8), nq = c % (BN / 8);
                if (k0 + k < p.K && n0 + nq * 8 < p.N)
                    v = *reinterpret_cast<const bf16x8 *>(p.B + (size_t)(k0 + k) * p.ldb + n0 + nq * 8);
            }
            rb[i] = v;
        }
    };
    auto image_row = [](int k) { // row of a k-major image that holds k
        if (!PERMUTE_KMAJOR) return k;
        const int e = k & 7, g = (k >> 3) & 3;
        return (k & ~31) + 4 * g + (e & 3) + 16 * (e >> 2);
    };
    auto store_tiles = [&](const bf16x8 (&ra)[NA], const bf16x8 (&rb)[NB]) {
        if constexpr (ABL == 4) { asm volatile("" ::"v"(ra[0]), "v"(rb[0])); return; }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int c = t + i * NT;
            if (A_KC) *reinterpret_cast<bf16x8 *>(&As[(c / (BK / 8)) * LDK + (c % (BK / 8)) * 8]) = ra[i];
            else *reinterpret_cast<bf16x8 *>(&As[image_row(c / (BM / 8)) * LDTA + (c % (BM / 8)) * 8]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int c = t + i * NT;
            if (B_KC) *reinterpret_cast<bf16x8 *>(&Bs[(c / (BK / 8)) * LDK + (c % (BK / 8)) * 8]) = rb[i];
            else *reinterpret_cast<bf16x8 *>(&Bs[image_row(c / (BN / 8)) * LDTB + (c % (BN / 8)) * 8]) = rb[i];
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    // fragment of MFMA tile `tile` (16 rows / columns starting at r0) for the 32-wide k block at kk:
    //   both operands k-major:   element j < 4: k = kk + 4 fg + j ; j >= 4: k = kk + 16 + 4 fg + (j - 4)
    //   otherwise:               element j: k = kk + 8 fg + j
    auto frag_kc = [&](const __bf16 *img, int r0, int kk) {
        return *reinterpret_cast<const bf16x8 *>(img + (r0 + fr) * LDK + kk + 8 * fg);
    };
    auto frag_tr = [&](const __bf16 *img, int ldt, int c0, int kk) { return tr_frag(img, ldt, c0, kk, lane); };

    // Fragments of the NEXT 32-wide k block are read while this block's MFMAs issue (as in gemm_f32_kernel): a bf16 block is
    // only TM*TN MFMAs of 16 cycles, far less than an LDS round trip, and with one workgroup per CU nothing else covers it.
    auto read_block = [&](int kk, bf16x8 (&a)[TM], bf16x8 (&b)[TN]) {
        if constexpr (ABL == 3) {
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = ra0[0];
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = rb0[0];
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
            a[i] = A_KC ? frag_kc(As, wm * (TM * 16) + i * 16, kk) : frag_tr(As, LDTA, wm * (TM * 16) + i * 16, kk);
#pragma unroll
        for (int j = 0; j < TN; j++)
            b[j] = B_KC ? frag_kc(Bs, wn * (TN * 16) + j * 16, kk) : frag_tr(Bs, LDTB, wn * (TN * 16) + j * 16, kk);
    };
    auto mfma_block = [&](const bf16x8 (&a)[TM], const bf16x8 (&b)[TN]) {
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int j = 0; j < TN; j++)
                if constexpr (ABL == 2) { acc[i][j][0] += (float)a[i][0]; acc[i][j][1] += (float)b[j][0]; }
                else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    };
    constexpr int NBLK = BK / 32;
    constexpr bool DEEP = NSTG == 3 || NSTG == 4; // (NSTG 4: the two-barrier loop with the deep fragment prefetch)
    auto multiply = [&](int k0) {
        const int kmax = (p.K - k0 < BK) ? p.K - k0 : BK; // K is a multiple of 16; rows past it were staged as zeros
        if (kmax < BK) { // ragged last tile: plain loop
            for (int kk = 0; kk < kmax; kk += 32) {
                bf16x8 a[TM], b[TN];
                read_block(kk, a, b);
                mfma_block(a, b);
            }
            return;
        }
        if constexpr (DEEP) {
            // every fragment of up to four blocks is requested before the first MFMA: the LDS serves the four waves' reads of a
            // block in turn, a round trip of ~130 cycles where a block's MFMAs take 64, and with ONE block ahead every block
            // waited for the difference
            constexpr int G = NBLK < 4 ? NBLK : 4;
#pragma unroll
            for (int g0 = 0; g0 < NBLK; g0 += G) {
                bf16x8 a[G][TM], b[G][TN];
#pragma unroll
                for (int g = 0; g < G; g++) read_block((g0 + g) * 32, a[g], b[g]);
#pragma unroll
                for (int g = 0; g < G; g++) mfma_block(a[g], b[g]);
            }
            return;
        }
        bf16x8 a0[TM], b0[TN], a1[TM], b1[TN];
        read_block(0, a0, b0);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 64) {
            if (kk + 32 < BK) read_block(kk + 32, a1, b1);
            mfma_block(a0, b0);
            if (kk + 32 < BK) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0); // the next block's LDS reads first
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                            // then this block's MFMAs
            if (kk + 32 < BK) {
                if (kk + 64 < BK) read_block(kk + 64, a0, b0);
                mfma_block(a1, b1);
                if (kk + 64 < BK) __builtin_amdgcn_sched_group_barrier(0x100, 2 * (TM + TN), 0);
                __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
            }
        }
    };
    auto main_loop = [&](auto inside) {
        if constexpr (gemm_bf16_two_images(NSTG)) {
            // Two images in LDS, ONE barrier per tile: while image i & 1 is multiplied, tile i + 1 -- in registers since two tiles
            // ago -- is written to the other image (nobody reads it: its last readers passed the barrier that ended tile i - 1),
            // and the registers it frees take the loads of tile i + 3.
            const int nt = (p.K + BK - 1) / BK;
            load_tiles(0, ra0, rb0, inside);
            if (nt > 1) load_tiles(BK, ra1, rb1, inside);
            use_image(0);
            store_tiles(ra0, rb0);
            if (nt > 2) load_tiles(2 * BK, ra0, rb0, inside);
            __syncthreads();
            for (int i = 0; i < nt; i += 2) { // two tiles per trip so that the stages keep their names
                if (i + 1 < nt) {
                    use_image(1);
                    store_tiles(ra1, rb1);
                    if (i + 3 < nt) load_tiles((i + 3) * BK, ra1, rb1, inside);
                }
                use_image(0);
                multiply(i * BK);
                __syncthreads();
                if (i + 1 < nt) {
                    if (i + 2 < nt) {
                        use_image(0);
                        store_tiles(ra0, rb0);
                        if (i + 4 < nt) load_tiles((i + 4) * BK, ra0, rb0, inside);
                    }
                    use_image(1);
                    multiply((i + 1) * BK);
                    __syncthreads();
                }
            }
        } else if constexpr (NSTG >= 2) {
            GNN_STAMP_DECL;
            load_tiles(0, ra0, rb0, inside);
            if (BK < p.K) load_tiles(BK, ra1, rb1, inside);
            for (int k0 = 0; k0 < p.K; k0 += 2 * BK) { // two tiles per trip so that the stages keep their names
                GNN_STAMP(0);
                store_tiles(ra0, rb0);
                GNN_STAMP(1);
                __syncthreads();
                GNN_STAMP(2);
                if (k0 + 2 * BK < p.K && ABL != 1) load_tiles(k0 + 2 * BK, ra0, rb0, inside);
                multiply(k0);
                GNN_STAMP(3);
                __syncthreads();
                GNN_STAMP(4);
                if (k0 + BK < p.K) {
                    store_tiles(ra1, rb1);
                    GNN_STAMP(5);
                    __syncthreads();
                    GNN_STAMP(6);
                    if (k0 + 3 * BK < p.K && ABL != 1) load_tiles(k0 + 3 * BK, ra1, rb1, inside);
                    multiply(k0 + BK);
                    GNN_STAMP(7);
                    __syncthreads();
                    GNN_STAMP(8);
                }
            }
            GNN_STAMP_FLUSH;
        } else {
            load_tiles(0, ra0, rb0, inside);
            for (int k0 = 0; k0 < p.K; k0 += BK) {
                store_tiles(ra0, rb0);
                __syncthreads();
                if (k0 + BK < p.K) load_tiles(k0 + BK, ra0, rb0, inside);
                multiply(k0);
                __syncthreads();
            }
        }
    };
    if (interior) main_loop(BoolC<true>{});
    else main_loop(BoolC<false>{});

    static_assert(2 * WM * 16 * (TN * 16 + 4) * sizeof(float) <= gemm_bf16_lds_bytes<BM, BN, A_KC, B_KC, NSTG>(), "epilogue staging fits the operand images");
    gemm_bf16_epilogue<TM, TN, EPI>(acc, p, reinterpret_cast<float *>(gemm_bf16_smem), m0, n0, wave, lane);
}

template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int NSTG_ = 2, int WM = 2>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_kernel(GNN_GEMM_HEAD_PARAMS(__bf16), GemmBf16Params p) {
    GNN_GEMM_TAKE_HEAD(p);
    gemm_bf16_tile<BM, BN, A_KC, B_KC, EPI, NSTG_, WM>(p, blockIdx.x, blockIdx.y);
}

// SEVERAL products of one form in ONE launch: the gradient (+ update) products of all layers of a net whose products are each
// a few hundred tiles (784-1024-1024-1024-10 at 256 rows: 208 + 256 + 256 + 16 tiles of 64 x 64) -- four launches of 4-8 us,
// each mostly launch and ramp, become one.  Workgroup w takes tile w - first[q] of product q (products in the order given:
// the host puts the largest first).
constexpr int GNN_GEMM_GROUP_MAX = 6;
struct GemmBf16Group {
    GemmBf16Params p[GNN_GEMM_GROUP_MAX];
    int first[GNN_GEMM_GROUP_MAX + 1]; // first[q] = number of tiles before product q; first[n] = all
    int tiles_x[GNN_GEMM_GROUP_MAX];
    int n;
};
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int NSTG_ = 2, int WM = 2>
__global__ __launch_bounds__(WM * 128) void gemm_bf16_group_kernel(GemmBf16Group g) {
    const int w = blockIdx.x;
    GemmBf16Params p = g.p[0]; // (selected with scalar moves: an index into the argument block would put the array in scratch)
    int first = 0, tiles_x = g.tiles_x[0];
#pragma unroll
    for (int i = 1; i < GNN_GEMM_GROUP_MAX; i++)
        if (i < g.n && w >= g.first[i]) { p = g.p[i]; first = g.first[i]; tiles_x = g.tiles_x[i]; }
    const int local = w - first;
    gemm_bf16_tile<BM, BN, A_KC, B_KC, EPI, NSTG_, WM>(p, local % tiles_x, local / tiles_x);
}

// f32 -> bf16 (RNE) over a flat buffer: the shadow of W after set_weights / init / load, dataset rows, ...
static __global__ __launch_bounds__(256) void to_bf16_kernel(const float4 *__restrict__ src, bf16x4 *__restrict__ dst, int64_t n4) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = src[i];
        dst[i] = (bf16x4){(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    }
}

} // namespace gnn
