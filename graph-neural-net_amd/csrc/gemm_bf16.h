// gemm_bf16.h -- GNN_DTYPE_BF16: the same three GEMM forms as kernels.h with bf16 OPERANDS and
// f32 accumulation on v_mfma_f32_16x16x32_bf16 (K = 32 per instruction, 16x the f32 MFMA rate).
// Master weights, momentum, activations and deltas stay f32 in HBM; a tile is rounded to bf16
// (v_cvt_pk_bf16_f32, round-to-nearest-even) while it is staged into LDS, so no shadow copies
// exist and the epilogues (activation, f', fused momentum update on the f32 masters) are the
// f32 ones.  LDS image: [row][k] bf16, k contiguous, row stride 72 (144 B = 9 x 16 B, odd): a
// fragment (8 consecutive k of one row) is one ds_read_b128 and 16 lanes hit 16 different slots.
#pragma once
#include "kernels.h"

namespace gnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int BM, int BN, bool A_KC, bool B_KC, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr int BK = 64, LDK = BK + 8;
    constexpr int TM = BM / 32, TN = BN / 32; // 16x16 MFMA tiles per wave (waves are 2 x 2)
    __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDK];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * LDK];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    constexpr int NA = BM * BK / 4 / 256, NB = BN * BK / 4 / 256; // float4 per thread per tile
    float4 ra[NA], rb[NB];

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = t + i * 256;
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
            const int idx = t + i * 256;
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
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < NA; i++) {
            const int idx = t + i * 256;
            if (A_KC) {
                const int m = idx % BM, kq = idx / BM;
                bf16x4 q = {(__bf16)ra[i].x, (__bf16)ra[i].y, (__bf16)ra[i].z, (__bf16)ra[i].w};
                *reinterpret_cast<bf16x4 *>(&As[m * LDK + kq * 4]) = q;
            } else { // 4 rows at one k: transposing 2-byte writes
                const int k = idx / (BM / 4), mq = idx % (BM / 4);
                As[(mq * 4 + 0) * LDK + k] = (__bf16)ra[i].x;
                As[(mq * 4 + 1) * LDK + k] = (__bf16)ra[i].y;
                As[(mq * 4 + 2) * LDK + k] = (__bf16)ra[i].z;
                As[(mq * 4 + 3) * LDK + k] = (__bf16)ra[i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < NB; i++) {
            const int idx = t + i * 256;
            if (B_KC) {
                const int n = idx % BN, kq = idx / BN;
                bf16x4 q = {(__bf16)rb[i].x, (__bf16)rb[i].y, (__bf16)rb[i].z, (__bf16)rb[i].w};
                *reinterpret_cast<bf16x4 *>(&Bs[n * LDK + kq * 4]) = q;
            } else {
                const int k = idx / (BN / 4), nq = idx % (BN / 4);
                Bs[(nq * 4 + 0) * LDK + k] = (__bf16)rb[i].x;
                Bs[(nq * 4 + 1) * LDK + k] = (__bf16)rb[i].y;
                Bs[(nq * 4 + 2) * LDK + k] = (__bf16)rb[i].z;
                Bs[(nq * 4 + 3) * LDK + k] = (__bf16)rb[i].w;
            }
        }
    };

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int j = 0; j < TN; j++) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    // operand maps of mfma_f32_16x16x32_bf16: lane l holds A[row l&15][8*(l>>4) + j] and
    // B[8*(l>>4) + j][col l&15], j = 0..7
    const __bf16 *ap = &As[(wm * (TM * 16) + fr) * LDK + fq * 8];
    const __bf16 *bp = &Bs[(wn * (TN * 16) + fr) * LDK + fq * 8];

    load_tiles(0);
    for (int k0 = 0; k0 < p.K; k0 += BK) {
        store_tiles();
        __syncthreads();
        if (k0 + BK < p.K) load_tiles(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 32) {
            bf16x8 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) a[i] = *reinterpret_cast<const bf16x8 *>(ap + i * 16 * LDK + kk);
#pragma unroll
            for (int j = 0; j < TN; j++) b[j] = *reinterpret_cast<const bf16x8 *>(bp + j * 16 * LDK + kk);
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int j = 0; j < TN; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // epilogue (f32, identical to gemm_f32_kernel): col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int j = 0; j < TN; j++) {
            const int n = n0 + wn * (TN * 16) + j * 16 + fr;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + wm * (TM * 16) + i * 16 + fq * 4 + r;
                if (m < p.M && n < p.N) {
                    const bool live = (m < p.m_true) && (n < p.n_true);
                    const float v = acc[i][j][r];
                    const size_t off = (size_t)m * p.ldc + n;
                    if (EPI == EPI_STORE) {
                        p.C[off] = live ? v : 0.f;
                    } else if (EPI == EPI_ACT) {
                        p.C[off] = live ? act_fn(p.act, v) : 0.f;
                    } else if (EPI == EPI_DACT) {
                        const float a = p.aux[(size_t)m * p.ldaux + n];
                        p.C[off] = live ? v * act_prime_from_a(p.act, a) : 0.f;
                    } else { // EPI_SGD on the f32 masters: ((step*G)/B) + (momentum*prev), SCE:333
                        if (live) {
                            const float adj = sgd_adj(p.step_over_b, v, p.momentum, p.V[off]);
                            p.W[off] -= adj;
                            p.V[off] = adj;
                        }
                    }
                }
            }
        }
    }
}

} // namespace gnn
