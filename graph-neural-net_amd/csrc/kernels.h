// kernels.h -- gfx950 (CDNA4, MI355X) device code of the MLP mini-batch SGD path.
//
// Every dense contraction of the reference's per-sample loop nests is a batched GEMM here:
//   forward      Z_l  = A_{l-1} . W_{l-1}            (SCE:187-192)          "NN"
//   backward     D_l  = (D_{l+1} . W_l^T) * f'(Z_l)  (SCE:272-278)          "NT"
//   weight grad  G_l  = A_l^T . D_{l+1}              (SCE:253-258,279-283,
//                                                     summed over the batch by SCE:305-322) "TN"
// all on the exact-f32 matrix instruction v_mfma_f32_16x16x4_f32 (one k-ordered fmaf chain per
// output element).  Operands are staged through LDS in a k-major image so that every MFMA
// fragment read is a conflict-free ds_read_b32.
//
// Layout in HBM: every matrix is dense row-major f32 with its leading dimension rounded up to
// PAD = 16 floats (64 B) and the padding kept at exactly 0, so no GEMM needs a remainder path:
//   W_l   [ld(d_l)] rows x ld(d_{l+1}) cols  (the reference's weights[l][in][out], SCE:44-47)
//   A_l   [B_pad]   rows x ld(d_l)           activations f(z_l), rows >= B are 0
//   D_l   [B_pad]   rows x ld(d_l)           dE/dz_l
#pragma once
#ifdef __HIPCC_RTC__ // compiled at run time by hiprtc (jit.h): its built-in runtime header is implicit
typedef signed char int8_t;
typedef unsigned char uint8_t;
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long long int64_t;
typedef unsigned long long uint64_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace gnn {

constexpr int PAD = 16;
__host__ __device__ constexpr int pad_up(int x, int m = PAD) { return (x + m - 1) / m * m; }

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum Act { ACT_LEAKY_RELU = 0, ACT_SIGMOID = 1, ACT_TANH = 2, ACT_RELU = 3, ACT_IDENTITY = 4 };

// f(z): the reference's innerActivationFunc lambdas as a closed enum (MT:234).
__device__ __forceinline__ float act_fn(int kind, float z) {
    switch (kind) {
    case ACT_LEAKY_RELU: return z > 0.f ? z : 0.01f * z;
    case ACT_SIGMOID: return 1.f / (1.f + __expf(-z));
    case ACT_TANH: return tanhf(z);
    case ACT_RELU: return z > 0.f ? z : 0.f;
    default: return z;
    }
}
// f'(z) written in terms of a = f(z), which is what stays in HBM (MT:235: z<=0 -> 0.01;
// a = f(z) has the sign of z and f(0) = 0, so `a > 0` decides the same branch).
__device__ __forceinline__ float act_prime_from_a(int kind, float a) {
    switch (kind) {
    case ACT_LEAKY_RELU: return a > 0.f ? 1.f : 0.01f;
    case ACT_SIGMOID: return a * (1.f - a);
    case ACT_TANH: return 1.f - a * a;
    case ACT_RELU: return a > 0.f ? 1.f : 0.f;
    default: return 1.f;
    }
}

// adj = ((step*G)/B) + (momentum*prev), SCE:333, with step/B folded into one factor.  ONE spelling
// for every kernel that updates -- an explicit fma of the first product onto the rounded second --
// so that the fused update, the flat update after an all-reduce and the tile-owner kernel give the
// same bits for the same G (left to the compiler, `a*b + c*d` contracts differently from kernel to kernel).
__device__ __forceinline__ float sgd_adj(float step_over_b, float g, float momentum, float prev) {
    return __builtin_fmaf(step_over_b, g, momentum * prev);
}

// ------------------------------------------------------------------------------------------
// GEMM  C[M x N] = opA(A)[M x K] . opB(B)[K x N]   with a fused epilogue.
//   A_KC: A element (m,k) at A[m*lda + k]   (k contiguous)  else at A[k*lda + m]
//   B_KC: B element (k,n) at B[n*ldb + k]   (k contiguous)  else at B[k*ldb + n]
// M, N, K are the PADDED extents (multiples of 16); m_true / n_true are the logical extents:
// outside them the epilogue stores exact zeros so the padding invariant survives every kernel.
// ------------------------------------------------------------------------------------------
enum Epi {
    EPI_STORE = 0,       // C = acc                          (logits; gradients)
    EPI_ACT = 1,         // C = f(acc)                       (hidden forward, SCE:184-186 of the next layer)
    EPI_DACT = 2,        // C = acc * f'(aux)                (backward data, SCE:277)
    EPI_SGD = 3          // v = step_over_b*acc + mu*V; W -= v; V = v   (SCE:333-339 fused into G_l)
};

// The values a GEMM kernel's main loop starts from travel AHEAD of its parameter struct: leading scalar / pointer arguments are
// preloaded into SGPRs by the dispatch (-amdgpu-kernarg-preload-count, build.py), so the first operand loads do not wait for a
// scalar-cache miss on the kernel-argument segment -- 0.2 us per launch (tools/gemm_probe 21; rowblock_kernel.h, GNN_RB_HEAD_PARAMS).
// The struct's own copies are overwritten from them; a launch passes GNN_GEMM_HEAD_ARGS(p), p.
#define GNN_GEMM_HEAD_PARAMS(T) const T *hd_A, const T *hd_B, int hd_lda, int hd_ldb, int hd_M, int hd_N, int hd_K
#define GNN_GEMM_TAKE_HEAD(p) do { (p).A = hd_A; (p).B = hd_B; (p).lda = hd_lda; (p).ldb = hd_ldb; (p).M = hd_M; (p).N = hd_N; (p).K = hd_K; } while (0)
#define GNN_GEMM_HEAD_ARGS(p) (p).A, (p).B, (p).lda, (p).ldb, (p).M, (p).N, (p).K
struct GemmParams {
    const float *A; int lda;
    const float *B; int ldb;
    float *C; int ldc;
    int M, N, K;          // padded extents
    int m_true, n_true;   // logical extents (zeros are stored beyond them)
    const float *aux; int ldaux; // EPI_DACT: a = f(z) of this layer
    float *W; float *V;   // EPI_SGD (same ld as C)
    float step_over_b, momentum;
    int act;
};

// ---- epilogue of the f32 GEMM kernels (gemm_f32_kernel, gemm_f32_dma_kernel) -------------------------------------
// C/D map of the 16x16 MFMA: tile column gamma = lane & 15, tile row rho = (lane >> 4) * 4 + reg; the kernels interleave
// their fragments so that MFMA tile (i, j) of a wave holds rows base + TM * rho + i and columns base + TN * gamma + j:
// a lane stores TN consecutive columns.
template <int TM, int TN, int EPI>
__device__ __forceinline__ void gemm_f32_epilogue(f32x4 (&acc)[TM][TN], const GemmParams &p, int m0, int n0, int wm, int wn, int fr, int fq) {
    typedef float vec_b __attribute__((ext_vector_type(TN == 1 ? 2 : TN)));
    const int n = n0 + wn * (TN * 16) + fr * TN; // this lane's TN consecutive columns (all inside or all outside N)
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int m = m0 + wm * (TM * 16) + (fq * 4 + r) * TM + i;
            if (m < p.M && n < p.N) {
                const size_t off = (size_t)m * p.ldc + n;
                float v[TN], aux[TN], vold[TN], wold[TN];
                if (EPI == EPI_DACT) {
                    if constexpr (TN == 1) aux[0] = p.aux[(size_t)m * p.ldaux + n];
                    else {
                        const vec_b t_ = *reinterpret_cast<const vec_b *>(p.aux + (size_t)m * p.ldaux + n);
#pragma unroll
                        for (int j = 0; j < TN; j++) aux[j] = t_[j];
                    }
                }
                if (EPI == EPI_SGD) {
                    if constexpr (TN == 1) { vold[0] = p.V[off]; wold[0] = p.W[off]; }
                    else {
                        const vec_b tv = *reinterpret_cast<const vec_b *>(p.V + off), tw = *reinterpret_cast<const vec_b *>(p.W + off);
#pragma unroll
                        for (int j = 0; j < TN; j++) { vold[j] = tv[j]; wold[j] = tw[j]; }
                    }
                }
                float out0[TN], out1[TN]; // out0: C (or W), out1: V
#pragma unroll
                for (int j = 0; j < TN; j++) {
                    const bool live = (m < p.m_true) && (n + j < p.n_true);
                    v[j] = acc[i][j][r];
                    if (EPI == EPI_STORE) out0[j] = live ? v[j] : 0.f;
                    else if (EPI == EPI_ACT) out0[j] = live ? act_fn(p.act, v[j]) : 0.f;
                    else if (EPI == EPI_DACT) out0[j] = live ? v[j] * act_prime_from_a(p.act, aux[j]) : 0.f;
                    else { // EPI_SGD: ((step*G)/B) + (momentum*prev), SCE:333; padding elements stay as they are (zeros)
                        const float adj = sgd_adj(p.step_over_b, v[j], p.momentum, vold[j]);
                        out0[j] = live ? wold[j] - adj : wold[j];
                        out1[j] = live ? adj : vold[j];
                    }
                }
                float *dst0 = (EPI == EPI_SGD) ? p.W + off : p.C + off;
                if constexpr (TN == 1) {
                    dst0[0] = out0[0];
                    if (EPI == EPI_SGD) p.V[off] = out1[0];
                } else {
                    vec_b o0, o1;
#pragma unroll
                    for (int j = 0; j < TN; j++) { o0[j] = out0[j]; if (EPI == EPI_SGD) o1[j] = out1[j]; }
                    *reinterpret_cast<vec_b *>(dst0) = o0;
                    if (EPI == EPI_SGD) *reinterpret_cast<vec_b *>(p.V + off) = o1;
                }
            }
        }
    }
}

// WM: wave rows (waves are WM x 2, WM * 128 threads).  WM = 4 puts EIGHT waves on a tile -- two per SIMD from one
// workgroup, which is what covers barriers and LDS latency when the grid has only one workgroup per CU.
// NSTG: k tiles held in registers ahead of the one being multiplied (see the main loop).  ONE for every tile: with two, the
// guarded loads of the stages sit in branches and the compiler's wait-count pass puts s_waitcnt vmcnt(0) at the head of the
// loop -- both stages drained, the same prefetch distance as one stage, for more registers and code: one stage is 2-8 %
// faster on every product of configs[3] / [4] (profiles/r02/gemm_probe_stages.log).
constexpr int gemm_f32_stages(int BM, int BN) { return 1; }
template <bool B> struct BoolC { static constexpr bool value = B; };
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2, int NSTG = gemm_f32_stages(BM, BN), int BK_ = 0>
__global__ __launch_bounds__(WM * 128) void gemm_f32_kernel(GNN_GEMM_HEAD_PARAMS(float), GemmParams p) {
    GNN_GEMM_TAKE_HEAD(p);
    constexpr int NT = WM * 128;
    // k depth of a staged tile: small tiles do few MFMAs per wave per k -- more K per barrier pair, and per prefetch distance
    // (32 x 32: 128 deep measured 26.5 us against 29.2 us at 64 on 512 x 1024 x 2048)
    constexpr int BK = BK_ ? BK_ : (BM <= 32) ? 128 : (BM <= 64) ? 64 : 32;
    constexpr int TM = BM / (WM * 16), TN = BN / 32;   // 16x16 MFMA tiles per wave (waves are WM x 2)
    constexpr int LDAS = BM + 16, LDBS = BN + 16; // row stride = 16 (mod 32) floats: lanes 0-15 / 16-31 hit disjoint banks
    __shared__ __attribute__((aligned(16))) float As[BK * LDAS];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDBS];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

    constexpr int NA = BM * BK / 4 / NT, NB = BN * BK / 4 / NT; // float4 per thread per tile
    static_assert(NA >= 1 && NB >= 1 && TM >= 1, "tile too small for this many threads");
    float4 ra[NSTG][NA], rb[NSTG][NB];

    // A tile wholly inside M x N with K a multiple of BK loads through precomputed per-thread offsets with no bounds test: the
    // guarded form costs ~13 instructions and an exec-mask branch per 16-B load (~80 instructions per k tile of a 64-wide tile).
    // With the single register stage this is worth 2-4 % on the gradient form and 4-12 % on the 8-wave 64 x 64 tiles of the
    // forward / backward-data forms (profiles/r02/gemm_probe_tiles2.log against gemm_probe_tiles1.log).  (With TWO stages the
    // unguarded k-contiguous loads had been slower: the compiler copied part of the second stage for the loop back-edge and
    // waited for it right behind the loads.)
    const bool fits32 = (unsigned long long)(A_KC ? p.M : p.K) * (unsigned)p.lda < 0xffffffffull &&
                        (unsigned long long)(B_KC ? p.N : p.K) * (unsigned)p.ldb < 0xffffffffull; // 32-bit element offsets below
    const bool interior = fits32 && (m0 + BM <= p.M) && (n0 + BN <= p.N) && (p.K % BK == 0);
    // (32-bit element offsets from the operand's base: base in SGPRs + one VGPR offset per load; operands are < 2^32 B)
    unsigned oa[NA], ob[NB];
#pragma unroll
    for (int i = 0; i < NA; i++) {
        const int idx = t + i * NT;
        oa[i] = A_KC ? (unsigned)(m0 + idx % BM) * (unsigned)p.lda + (idx / BM) * 4 : (unsigned)(idx / (BM / 4)) * (unsigned)p.lda + m0 + (idx % (BM / 4)) * 4;
    }
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const int idx = t + i * NT;
        ob[i] = B_KC ? (unsigned)(n0 + idx % BN) * (unsigned)p.ldb + (idx / BN) * 4 : (unsigned)(idx / (BN / 4)) * (unsigned)p.ldb + n0 + (idx % (BN / 4)) * 4;
    }
    const unsigned a_kstride = A_KC ? 1u : (unsigned)p.lda, b_kstride = B_KC ? 1u : (unsigned)p.ldb;
    auto load_tiles = [&](int k0, float4 (&ra)[NA], float4 (&rb)[NB], auto inside) {
        if constexpr (decltype(inside)::value) {
#pragma unroll
            for (int i = 0; i < NA; i++) ra[i] = *reinterpret_cast<const float4 *>(p.A + (oa[i] + (unsigned)k0 * a_kstride));
#pragma unroll
            for (int i = 0; i < NB; i++) rb[i] = *reinterpret_cast<const float4 *>(p.B + (ob[i] + (unsigned)k0 * b_kstride));
            return;
        }
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = t + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (A_KC) {
                const int m = idx % BM, kq = idx / BM;
                if (m0 + m < p.M && k0 + kq * 4 < p.K)
                    v = *reinterpret_cast<const float4 *>(p.A + (size_t)(m0 + m) * p.lda + k0 + kq * 4);
            } else {
                const int k = idx / (BM / 4), mq = idx % (BM / 4);
                if (k0 + k < p.K && m0 + mq * 4 < p.M)
                    v = *reinterpret_cast<const float4 *>(p.A + (size_t)(k0 + k) * p.lda + m0 + mq * 4);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = t + i * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (B_KC) {
                const int n = idx % BN, kq = idx / BN;
                if (n0 + n < p.N && k0 + kq * 4 < p.K)
                    v = *reinterpret_cast<const float4 *>(p.B + (size_t)(n0 + n) * p.ldb + k0 + kq * 4);
            } else {
                const int k = idx / (BN / 4), nq = idx % (BN / 4);
                if (k0 + k < p.K && n0 + nq * 4 < p.N)
                    v = *reinterpret_cast<const float4 *>(p.B + (size_t)(k0 + k) * p.ldb + n0 + nq * 4);
            }
            rb[i] = v;
        }
    };
    auto store_tiles = [&](const float4 (&ra)[NA], const float4 (&rb)[NB]) {
#ifdef GNN_F32_NO_LDS_WRITES // (tools/gemm_probe, timing only: what the write phase of the loop costs)
        asm volatile("" ::"v"(ra[0].x), "v"(rb[0].x));
        return;
#endif
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = t + i * NT;
            if (A_KC) {
                const int m = idx % BM, kq = idx / BM;
                As[(kq * 4 + 0) * LDAS + m] = ra[i].x;
                As[(kq * 4 + 1) * LDAS + m] = ra[i].y;
                As[(kq * 4 + 2) * LDAS + m] = ra[i].z;
                As[(kq * 4 + 3) * LDAS + m] = ra[i].w;
            } else {
                const int k = idx / (BM / 4), mq = idx % (BM / 4);
                *reinterpret_cast<float4 *>(&As[k * LDAS + mq * 4]) = ra[i];
            }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = t + i * NT;
            if (B_KC) {
                const int n = idx % BN, kq = idx / BN;
                Bs[(kq * 4 + 0) * LDBS + n] = rb[i].x;
                Bs[(kq * 4 + 1) * LDBS + n] = rb[i].y;
                Bs[(kq * 4 + 2) * LDBS + n] = rb[i].z;
                Bs[(kq * 4 + 3) * LDBS + n] = rb[i].w;
            } else {
                const int k = idx / (BN / 4), nq = idx % (BN / 4);
                *reinterpret_cast<float4 *>(&Bs[k * LDBS + nq * 4]) = rb[i];
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Fragment rows are INTERLEAVED: MFMA tile i of this wave holds the rows base + TM*rho + i
    // (rho = the tile's own row index 0..15), tile j the columns base + TN*gamma + j.  Any assignment of
    // rows to tiles is a valid GEMM, and with this one a lane's TM (TN) operand values for a k sit
    // next to each other in the [k][m] LDS image: ONE b128 / b64 read per operand per k step instead
    // of TM (TN) b32 reads, and the epilogue stores TN consecutive columns per lane (16 B for TN = 4).
    const int fr = lane & 15, fq = lane >> 4;
    const int a_base = fq * LDAS + wm * (TM * 16) + fr * TM;
    const int b_base = fq * LDBS + wn * (TN * 16) + fr * TN;
    typedef float vec_a __attribute__((ext_vector_type(TM == 1 ? 2 : TM))); // (a 1-vector is not a type; b32 read below)
    typedef float vec_b __attribute__((ext_vector_type(TN == 1 ? 2 : TN)));

    // The fragments of k step j+1 are read while the MFMAs of step j issue: with one workgroup per CU (one wave per
    // SIMD) nothing else covers the ~100 cycles between a ds_read and its first use, and the compiler, left alone,
    // emits read - wait - 8 MFMAs - read - wait ... (a quarter of the MFMA time exposed on 64 x 64 tiles).
    auto read_frag = [&](int kk, float (&a)[TM], float (&b)[TN]) {
        if constexpr (TM == 1) {
            a[0] = As[kk * LDAS + a_base];
        } else {
            const vec_a va = *reinterpret_cast<const vec_a *>(&As[kk * LDAS + a_base]);
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = va[i];
        }
        if constexpr (TN == 1) {
            b[0] = Bs[kk * LDBS + b_base];
        } else {
            const vec_b vb = *reinterpret_cast<const vec_b *>(&Bs[kk * LDBS + b_base]);
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = vb[j];
        }
    };
    auto multiply = [&]() {
        // blocks of two k steps; the scheduler is told the order (next block's LDS reads, then this block's MFMAs)
        float a0[2][TM], b0[2][TN], a1[2][TM], b1[2][TN];
        read_frag(0, a0[0], b0[0]);
        read_frag(4, a0[1], b0[1]);
        auto mfma_block = [&](float (&a)[2][TM], float (&b)[2][TN]) {
#pragma unroll
            for (int u = 0; u < 2; u++)
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int j = 0; j < TN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
        };
#pragma unroll
        for (int kk = 0; kk < BK; kk += 16) {
            read_frag(kk + 8, a1[0], b1[0]);
            read_frag(kk + 12, a1[1], b1[1]);
            mfma_block(a0, b0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);            // the four DS reads of the next block first
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);  // then this block's MFMAs
            if (kk + 16 < BK) {
                read_frag(kk + 16, a0[0], b0[0]);
                read_frag(kk + 20, a0[1], b0[1]);
            }
            mfma_block(a1, b1);
            if (kk + 16 < BK) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN, 0);
        }
    };
    // NSTG register stages: the loads of tile t + NSTG are issued before tile t is multiplied, so NSTG - 1 tiles
    // (plus the one being staged) are in flight per workgroup.  What a CU must keep in flight is latency x rate: a
    // 32 x 32 tile with two stages had 32 KB out per CU and ran at 9 B/clk (256 x 1024 x 1024 in 12.6 us);
    // 128 x 128 tiles take one stage (a second would cost a wave of occupancy).
    auto main_loop = [&](auto inside) {
        if constexpr (NSTG >= 2) {
#pragma unroll
            for (int s = 0; s < NSTG; s++)
                if (s * BK < p.K) load_tiles(s * BK, ra[s], rb[s], inside);
            for (int k0 = 0; k0 < p.K; k0 += NSTG * BK) { // NSTG tiles per trip so that the stages keep their names
#pragma unroll
                for (int s = 0; s < NSTG; s++) {
                    if (k0 + s * BK < p.K) {
                        store_tiles(ra[s], rb[s]);
                        __syncthreads();
                        if (k0 + (s + NSTG) * BK < p.K) load_tiles(k0 + (s + NSTG) * BK, ra[s], rb[s], inside);
                        multiply();
                        __syncthreads();
                    }
                }
            }
        } else {
            load_tiles(0, ra[0], rb[0], inside);
            for (int k0 = 0; k0 < p.K; k0 += BK) {
                store_tiles(ra[0], rb[0]);
                __syncthreads();
                if (k0 + BK < p.K) load_tiles(k0 + BK, ra[0], rb[0], inside); // next tile's latency hides under the MFMAs
                multiply();
                __syncthreads();
            }
        }
    };
    if (interior) main_loop(BoolC<true>{});
    else main_loop(BoolC<false>{});

    gemm_f32_epilogue<TM, TN, EPI>(acc, p, m0, n0, wm, wn, fr, fq);
}

// ------------------------------------------------------------------------------------------
// Output layer, one wave per sample row.
//   SOFTMAX_CE: p = softmax(z) (SCE:357-376; the row max IS subtracted here -- mathematically
//   identical, and expf would overflow at the reference's |z| ~ 100 logits: SURVEY H2),
//   delta = p - y (SCE:249-251), loss = -sum y ln p (SCE:213-217) evaluated in log space,
//   label = argmax with `>=` (MT:166-168: ties -> highest index).
//   ACT_LOSS  : a = f_last(z) (GNN:215-218), delta = loss'(a,y) * f_last'(z) (GNN:267-271),
//   loss = sum loss(a,y) (GNN:236-239).
// Rows >= B and columns >= n_true of `delta`/`prob` are written as zeros.
// ------------------------------------------------------------------------------------------
struct OutParams {
    const float *Z; int ldz;    // logits [B_pad][ld]
    const float *Y; int ldy;    // expected [rows][ld] (may be null when !want_delta && !want_loss)
    float *prob; int ldp;       // may be null
    float *delta; int ldd;      // may be null
    float *loss;                // [B_pad] may be null
    int32_t *label;             // [B_pad] may be null
    int B, B_pad, n_true, n_pad;
    int out_kind, last_act;
    __bf16 *delta_b;            // bf16 mode: the rounding of `delta`, same ld (may be null)
};

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

static __global__ __launch_bounds__(256) void output_layer_kernel(OutParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.B_pad) return;
    const bool live_row = row < p.B;
    const float *z = p.Z + (size_t)row * p.ldz;
    const float *y = p.Y ? p.Y + (size_t)row * p.ldy : nullptr;

    // NaN rule of MT:166-168: the scan starts with actual = 0 and `x >= NaN` is false, so a NaN
    // at index 0 is sticky (label 0) while a NaN elsewhere is never selected.  With softmax any
    // NaN logit makes EVERY probability NaN, i.e. label 0.
    bool has_nan = false;
    constexpr int CV = 4; // a row of up to 256*CV logits is held in registers: ONE round trip to memory
    if (p.out_kind == 0 && p.n_pad <= 256 * CV) {
        // (a row read three times with dependent loads -- max, sum, probabilities -- cost 19.6 us at 512 x 1024; with one 4-B access
        //  per lane and column the 512 x 1024 rows of configs[3] took 6.6 us, mostly 4-B stores at ~7 B/clk/CU: a lane now owns FOUR
        //  consecutive columns per 256-column chunk -- 16-B loads and stores; leading dimensions are multiples of 16 floats)
        const int nv = (p.n_pad + 255) / 256; // wave-uniform
        f32x4 zc[CV], yc[CV];
#pragma unroll
        for (int i = 0; i < CV; i++) {
            const int c = 4 * lane + 256 * i;
            zc[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; yc[i] = zc[i];
            if (i < nv && c < p.n_pad) {
                zc[i] = *reinterpret_cast<const f32x4 *>(z + c);
                if (y) yc[i] = *reinterpret_cast<const f32x4 *>(y + c);
            }
        }
        float mx = -__builtin_inff();
        int best = -1;
#pragma unroll
        for (int i = 0; i < CV; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int c = 4 * lane + 256 * i + j;
                const float v = zc[i][j];
                if (i < nv && c < p.n_true) {
                    has_nan |= (v != v);
                    if (v >= mx) { mx = v; best = c; } // ascending c within a lane: `>=` keeps the highest
                }
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { // larger value wins, equal values -> higher index (MT:166-168)
            const float ov = __shfl_xor(mx, o);
            const int ob = __shfl_xor(best, o);
            if (ov > mx || (ov == mx && ob > best)) { mx = ov; best = ob; }
        }
        float s = 0.f;
        f32x4 e[CV];
#pragma unroll
        for (int i = 0; i < CV; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float ev = (i < nv && 4 * lane + 256 * i + j < p.n_true) ? __expf(zc[i][j] - mx) : 0.f;
                e[i][j] = ev; s += ev;
            }
        s = wave_sum(s);
        const float inv = 1.f / s;
        const float lse = mx + __logf(s);
        float l = 0.f;
#pragma unroll
        for (int i = 0; i < CV; i++) {
            const int c = 4 * lane + 256 * i;
            if (i < nv && c < p.n_pad) {
                f32x4 pr, dd;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const bool live = live_row && c + j < p.n_true;
                    pr[j] = live ? e[i][j] * inv : 0.f;
                    const float yy = live ? yc[i][j] : 0.f;
                    dd[j] = live ? pr[j] - yy : 0.f;
                    if (live && yy != 0.f) l += yy * (lse - zc[i][j]); // -y ln p
                }
                if (p.prob) *reinterpret_cast<f32x4 *>(p.prob + (size_t)row * p.ldp + c) = pr;
                if (p.delta) *reinterpret_cast<f32x4 *>(p.delta + (size_t)row * p.ldd + c) = dd;
                if (p.delta_b) {
                    typedef __bf16 out_bf16x4 __attribute__((ext_vector_type(4)));
                    *reinterpret_cast<out_bf16x4 *>(p.delta_b + (size_t)row * p.ldd + c) = (out_bf16x4){(__bf16)dd[0], (__bf16)dd[1], (__bf16)dd[2], (__bf16)dd[3]};
                }
            }
        }
        l = wave_sum(l);
        if (__any(has_nan)) best = 0;
        if (lane == 0) {
            if (p.loss) p.loss[row] = live_row ? l : 0.f;
            if (p.label) p.label[row] = live_row ? best : -1;
        }
    } else if (p.out_kind == 0) { // wider rows: three passes over memory
        float mx = -__builtin_inff();
        int best = -1;
        for (int c = lane; c < p.n_true; c += 64) {
            const float v = z[c];
            has_nan |= (v != v);
            if (v >= mx) { mx = v; best = c; } // ascending c within a lane: `>=` keeps the highest
        }
        // wave argmax: larger value wins, equal values -> higher index (MT:166-168)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mx, o);
            const int ob = __shfl_xor(best, o);
            if (ov > mx || (ov == mx && ob > best)) { mx = ov; best = ob; }
        }
        float s = 0.f;
        for (int c = lane; c < p.n_true; c += 64) s += __expf(z[c] - mx);
        s = wave_sum(s);
        const float inv = 1.f / s;
        const float lse = mx + __logf(s);
        float l = 0.f;
        for (int c = lane; c < p.n_pad; c += 64) {
            const bool live = live_row && c < p.n_true;
            const float pr = live ? __expf(z[c] - mx) * inv : 0.f;
            const float yy = (live && y) ? y[c] : 0.f;
            if (p.prob) p.prob[(size_t)row * p.ldp + c] = pr;
            if (p.delta) p.delta[(size_t)row * p.ldd + c] = live ? pr - yy : 0.f;
            if (p.delta_b) p.delta_b[(size_t)row * p.ldd + c] = (__bf16)(live ? pr - yy : 0.f);
            if (live && yy != 0.f) l += yy * (lse - z[c]); // -y ln p
        }
        l = wave_sum(l);
        if (__any(has_nan)) best = 0;
        if (lane == 0) {
            if (p.loss) p.loss[row] = live_row ? l : 0.f;
            if (p.label) p.label[row] = live_row ? best : -1;
        }
    } else {
        float mx = -__builtin_inff();
        int best = -1;
        float l = 0.f;
        for (int c = lane; c < p.n_pad; c += 64) {
            const bool live = live_row && c < p.n_true;
            const float zz = z[c];
            const float a = act_fn(p.last_act, zz);
            const float yy = (live && y) ? y[c] : 0.f;
            const float d = a - yy; // loss' of 0.5*(a-y)^2
            if (p.prob) p.prob[(size_t)row * p.ldp + c] = live ? a : 0.f;
            if (p.delta) p.delta[(size_t)row * p.ldd + c] = live ? d * act_prime_from_a(p.last_act, a) : 0.f;
            if (p.delta_b) p.delta_b[(size_t)row * p.ldd + c] = (__bf16)(live ? d * act_prime_from_a(p.last_act, a) : 0.f);
            if (live) {
                l += 0.5f * d * d;
                if (c == 0) has_nan = (a != a); // element-wise output: only a NaN at index 0 is sticky
                if (a >= mx) { mx = a; best = c; }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(mx, o);
            const int ob = __shfl_xor(best, o);
            if (ov > mx || (ov == mx && ob > best)) { mx = ov; best = ob; }
        }
        l = wave_sum(l);
        if (__any(has_nan)) best = 0;
        if (lane == 0) {
            if (p.loss) p.loss[row] = live_row ? l : 0.f;
            if (p.label) p.label[row] = live_row ? best : -1;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Momentum update over the flat padded parameter buffer (SCE:327-342), 16 B per lane:
//   adj = (step*G)/B + momentum*prev ; W -= adj ; prev = adj
// Padding elements have G = 0 and prev = 0, so they stay exactly 0.
// ------------------------------------------------------------------------------------------
typedef __bf16 sgd_bf16x4 __attribute__((ext_vector_type(4)));
struct SgdParams {
    float4 *W; float4 *V; const float4 *G;
    int64_t n4;
    float step_over_b, momentum;
    sgd_bf16x4 *Wb; // bf16 mode: the shadow of W (may be null)
};
static __global__ __launch_bounds__(256) void sgd_momentum_kernel(SgdParams p) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n4; i += (int64_t)gridDim.x * 256) {
        const float4 g = p.G[i];
        float4 v = p.V[i], w = p.W[i];
        v.x = sgd_adj(p.step_over_b, g.x, p.momentum, v.x);
        v.y = sgd_adj(p.step_over_b, g.y, p.momentum, v.y);
        v.z = sgd_adj(p.step_over_b, g.z, p.momentum, v.z);
        v.w = sgd_adj(p.step_over_b, g.w, p.momentum, v.w);
        w.x -= v.x; w.y -= v.y; w.z -= v.z; w.w -= v.w;
        p.V[i] = v;
        p.W[i] = w;
        if (p.Wb) p.Wb[i] = (sgd_bf16x4){(__bf16)w.x, (__bf16)w.y, (__bf16)w.z, (__bf16)w.w};
    }
}

// ------------------------------------------------------------------------------------------
// Input encoding.  Host fp64 rows -> padded f32 rows; apply_act: the reference applies the
// inner activation to the raw input too (SCE:183-186 with l-1 = 0), so A_0 = f(x).
// ------------------------------------------------------------------------------------------
static __global__ __launch_bounds__(256) void convert_rows_f64_kernel(const double *__restrict__ src, int d,
                                                              float *__restrict__ dst, int ld, int64_t rows,
                                                              int64_t rows_pad, int act, int apply_act) {
    const int64_t total = rows_pad * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.f;
        if (r < rows && c < d) {
            v = (float)src[r * d + c];
            if (apply_act) v = act_fn(act, v);
        }
        dst[i] = v;
    }
}

// A host batch as the boundary hands it over (NN:16-51 take double[] rows): the library has already rounded the rows
// to f32 into a PINNED host buffer (the same RNE conversion as above, done by the calling thread while it copies), and
// this kernel pulls them across PCIe itself -- no copy-engine command, no staging buffer in HBM: X [rows][d0] then
// Y [rows][dl], unpadded.  Writes A_0 = f(x) (SCE:183-186) and the expected rows in the padded layout, zeros elsewhere.
struct StageBatchParams {
    const float *src_x; const float *src_y; // pinned host memory (device-visible); src_y may be null
    int d0, dl;
    float *dst_x; int ld0;
    float *dst_y; int ldl;
    int64_t rows, rows_pad;
    int act;
    __bf16 *dst_xb;                         // bf16 mode: the rounding of A_0, same ld (may be null)
};
static __global__ __launch_bounds__(256) void stage_batch_kernel(StageBatchParams p) {
    const int64_t nx = p.rows_pad * p.ld0, total = nx + (p.src_y ? p.rows_pad * p.ldl : 0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < nx) {
            const int64_t r = i / p.ld0;
            const int c = (int)(i - r * p.ld0);
            float v = 0.f;
            if (r < p.rows && c < p.d0) v = act_fn(p.act, p.src_x[r * p.d0 + c]);
            p.dst_x[i] = v;
            if (p.dst_xb) p.dst_xb[i] = (__bf16)v;
        } else {
            const int64_t j = i - nx, r = j / p.ldl;
            const int c = (int)(j - r * p.ldl);
            p.dst_y[j] = (r < p.rows && c < p.dl) ? p.src_y[r * p.dl + c] : 0.f;
        }
    }
}

// raw IDX bytes -> A_0 = f(pixel/255.0) (MT:98) and one-hot labels (MT:112-118)
static __global__ __launch_bounds__(256) void encode_u8_kernel(const uint8_t *__restrict__ pix, int d,
                                                       float *__restrict__ dst, int ld, int64_t rows,
                                                       int64_t rows_pad, int act) {
    const int64_t total = rows_pad * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        float v = 0.f;
        if (r < rows && c < d) v = act_fn(act, (float)((double)pix[r * d + c] / 255.0));
        dst[i] = v;
    }
}
static __global__ __launch_bounds__(256) void onehot_u8_kernel(const uint8_t *__restrict__ lab, int n_classes,
                                                       float *__restrict__ dst, int ld, int64_t rows,
                                                       int64_t rows_pad) {
    const int64_t total = rows_pad * ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld;
        const int c = (int)(i - r * ld);
        dst[i] = (r < rows && c < n_classes && (int)lab[r] == c) ? 1.f : 0.f;
    }
}

// gather dataset rows by index (one NNT.sample draw, NNT:143-158) into a dense batch
// rows idx[0..B) of TWO row-aligned matrices (inputs and expected outputs) in one launch
static __global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src_x, int ld_x, float *__restrict__ dst_x,
                                                         const float *__restrict__ src_y, int ld_y, float *__restrict__ dst_y,
                                                         const int32_t *__restrict__ idx, int B, int B_pad) {
    const int x4 = ld_x / 4, y4 = ld_y / 4, row4 = x4 + y4;
    const int64_t total = (int64_t)B_pad * row4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / row4);
        const int c = (int)(i - (int64_t)r * row4);
        const bool in_x = c < x4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < B) {
            const size_t row = (size_t)idx[r];
            v = in_x ? reinterpret_cast<const float4 *>(src_x + row * ld_x)[c]
                     : reinterpret_cast<const float4 *>(src_y + row * ld_y)[c - x4];
        }
        if (in_x) reinterpret_cast<float4 *>(dst_x + (size_t)r * ld_x)[c] = v;
        else reinterpret_cast<float4 *>(dst_y + (size_t)r * ld_y)[c - x4] = v;
    }
}

// f32 rows (padded) -> host-bound fp64 rows (unpadded)
static __global__ __launch_bounds__(256) void export_rows_f64_kernel(const float *__restrict__ src, int ld, int d,
                                                             int64_t rows, double *__restrict__ dst) {
    const int64_t total = rows * d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / d;
        const int c = (int)(i - r * d);
        dst[i] = (double)src[r * ld + c];
    }
}

} // namespace gnn
