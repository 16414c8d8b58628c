// middle4_kernel.h -- the per-sample chain between A_1 and delta_1 for FOUR batch rows per
// workgroup, with every middle weight matrix resident in LDS.
//
// Why four rows: the chain (SCE:172-198 forward, SCE:249-251 output delta, SCE:262-278 backward)
// is per-sample independent, and a workgroup must stream ALL middle weights (W_1 is 136 KB for
// 784-300-100-10) through its CU.  A CU pulls only ~10-15 B/clk from beyond L2 and ~30 B/clk
// from L2, and every kernel starts with a cold L2, so the step time of this kernel is the weight
// stream.  With B/4 workgroups, four of them share an XCD (blocks b and b+8 land on one L2):
// each starts its stream at a different quarter, so three quarters of its bytes are L2 hits.
// The weights are read ONCE into LDS ([k][ld+1] image, odd row stride) and serve both the
// forward product (lanes along n: consecutive addresses) and the backward product with W^T
// (lanes along the image's rows: stride ld+1, conflict-free because it is odd).
//
// MFMA: v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of (4x1).(1x4) per instruction.  Every block gets
// the same A column (the 4 batch rows' activations at k; lane l supplies row l&3) and its own
// 4 B values (lane l supplies W[k][n0+l]), so one instruction is the rank-1 update
// Z[0..3][n0..n0+63] += a[0..3] (x) W[k][n0..n0+63]; accumulator register i of lane l is
// Z[row i][n0+l].  Exact f32, same rate per FLOP as the 16x16x4 form, no padding rows.
#pragma once
#include "fused_kernels.h"

namespace gnn {

struct Mid4Params {
    int L;
    int d[MAX_LAYERS], ld[MAX_LAYERS];
    int kr[MAX_LAYERS];          // rows of layer l kept in LDS / contracted over: round_up(d[l], 4)
    const float *W[MAX_LAYERS];  // global W_l, l = 1..L-2
    float *act[MAX_LAYERS];      // act[1] in; act[2..L-2] out
    float *delta[MAX_LAYERS];    // delta[1..L-1] out
    int off_w[MAX_LAYERS];       // LDS float offsets: weight images l = 1..L-2, [kr[l]][ld[l+1]+1]
    int off_act[MAX_LAYERS];     // activation images l = 1..L-2, [4][ld[l]+4]
    int off_dl[MAX_LAYERS];      // delta images l = 2..L-1, [4][ld[l]+4]
    int off_logits, off_y, off_scratch;
    int w_total4;                // float4s in all weight images
    int w_begin4[MAX_LAYERS];    // first float4 of layer l's image in that index space
    unsigned w_inv_c4[MAX_LAYERS]; // ceil(2^32 / (ld[l+1]/4)): row = umulhi(idx, inv)
    int ks_fwd[MAX_LAYERS];      // K splits of the product giving layer l (l = 2..L-1)
    int ks_bwd[MAX_LAYERS];      // K splits of the product giving delta_l (l = 1..L-2)
    const float *Y; int ldy;
    float *prob; float *loss; int32_t *label;
    int B;
    int last_act;
    unsigned long long *stamps;  // STAMP builds only
};

// 16-lane (one DPP row) butterfly: every lane ends with the reduction over its row of 16.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true);
}
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    return v;
}
// (value, index) argmax over the row of 16: larger value wins, ties -> higher index (MT:166-168)
__device__ __forceinline__ void row16_argmax(float &v, int &ix) {
#define GNN_ARGMAX_STEP(CTRL)                                          \
    {                                                                  \
        const float ov = dpp_f<CTRL>(v);                               \
        const int oi = dpp_i<CTRL>(ix);                                \
        if (ov > v || (ov == v && oi > ix)) { v = ov; ix = oi; }       \
    }
    GNN_ARGMAX_STEP(DPP_XOR1)
    GNN_ARGMAX_STEP(DPP_XOR2)
    GNN_ARGMAX_STEP(DPP_HALF_MIRROR)
    GNN_ARGMAX_STEP(DPP_MIRROR)
#undef GNN_ARGMAX_STEP
}

#define GNN_STAMP4(i)                                                                             \
    do {                                                                                          \
        if (STAMP && threadIdx.x == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// partial[ks][4][GW] = A_img[4 x Kpart] . B   for this wave's (column group g, K part ks)
//   TRANS = false: B[k][n] = Wimg[k*ldw + n]          (forward,  n = output neuron)
//   TRANS = true : B[k][n] = Wimg[n*ldw + k]          (backward, n = input neuron; n < n_rows)
template <bool TRANS>
__device__ __forceinline__ void rowblock_product(const float *A_img, int lda, const float *Wimg, int ldw, int k4_begin,
                                                 int k4_end, int n0, int n_rows, float *partial, int gw, int lane) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float *arow = A_img + (lane & 3) * lda;
    const float *wp;
    int wstep;
    if (TRANS) {
        int n = n0 + lane;
        n = n < n_rows ? n : n_rows - 1; // columns past the image compute garbage nobody reads
        wp = Wimg + n * ldw;
        wstep = 1;
    } else {
        wp = Wimg + n0 + lane;
        wstep = ldw;
    }
    // U groups of 4 k per trip: all 5*U LDS reads are issued before the first MFMA waits
    constexpr int U = 4;
    for (int kb = k4_begin; kb < k4_end; kb += U) { // k4_begin / k4_end are wave-uniform
        f32x4 a[U];
        float b[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k4 = (kb + u < k4_end) ? kb + u : k4_end - 1; // tail: re-read the last group, weight 0
            a[u] = *reinterpret_cast<const f32x4 *>(arow + 4 * k4);
            const float *w = wp + (4 * k4) * wstep;
            b[u][0] = w[0]; b[u][1] = w[wstep]; b[u][2] = w[2 * wstep]; b[u][3] = w[3 * wstep];
            if (kb + u >= k4_end) a[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (u & 1) {
#pragma unroll
                for (int j = 0; j < 4; j++) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][j], b[u][j], acc1, 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][j], b[u][j], acc0, 0, 0, 0);
            }
        }
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int i = 0; i < 4; i++) partial[i * gw + n0 + lane] = acc[i];
}

template <int ACT, int OUTK, bool BACKWARD, bool STAMP = false>
__global__ __launch_bounds__(1024) void middle4_kernel(Mid4Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NW = 16, NT_ = 1024;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // provably wave-uniform: scalar branches
    const int row0 = blockIdx.x * 4;
    const int Lm = p.L - 1;
    // STAMP builds run the body twice and stamp the SECOND pass too (slots 16..31 of the
    // next record): the difference is what a cold instruction cache / cold L2 costs.
    for (int pass = 0; pass < (STAMP ? 2 : 1); pass++) {
    if (STAMP && pass == 1) { __syncthreads(); p.stamps += 16 * gridDim.x; }
    GNN_STAMP4(0);

    // ---- phase 0: everything this block reads from memory, issued before anything waits ----
    {   // A_1 rows and expected rows
        const int q1 = p.ld[1] / 4, qy = p.ld[Lm] / 4;
        for (int e = t; e < 4 * q1; e += NT_) {
            const int m = e / q1, q = e - m * q1;
            *reinterpret_cast<float4 *>(smem + p.off_act[1] + m * (p.ld[1] + 4) + q * 4) =
                *reinterpret_cast<const float4 *>(p.act[1] + (size_t)(row0 + m) * p.ld[1] + q * 4);
        }
        if (p.Y) {
            for (int e = NT_ - 1 - t; e < 4 * qy; e += NT_) { // the last threads, so that both loads fly together
                const int m = e / qy, q = e - m * qy;
                *reinterpret_cast<float4 *>(smem + p.off_y + m * p.ld[Lm] + q * 4) =
                    *reinterpret_cast<const float4 *>(p.Y + (size_t)(row0 + m) * p.ldy + q * 4);
            }
        }
    }
    // weight images of every middle layer as ONE index space of float4s, all loads in flight at
    // once; the start is rotated by the block's slot on its XCD so that the four blocks sharing
    // an L2 fetch different quarters first
    {
        const int total = p.w_total4;
        const int rot = (int)(((long)total * ((blockIdx.x >> 3) & 3)) >> 2);
        constexpr int MAXF = 10;
        for (int base = 0; base < total; base += MAXF * NT_) {
            float4 v[MAXF];
            int dsto[MAXF];
#pragma unroll
            for (int i = 0; i < MAXF; i++) {
                int idx = base + i * NT_ + t;
                dsto[i] = -1;
                if (idx < total) {
                    idx += rot;
                    if (idx >= total) idx -= total;
                    int l = 1;
#pragma unroll
                    for (int j = 2; j < MAX_LAYERS - 1; j++)
                        if (j < Lm && idx >= p.w_begin4[j]) l = j;
                    idx -= p.w_begin4[l];
                    const int c4 = p.ld[l + 1] >> 2;
                    const int row = (int)__umulhi((unsigned)idx, p.w_inv_c4[l]);
                    const int col = (idx - row * c4) * 4;
                    v[i] = *reinterpret_cast<const float4 *>(p.W[l] + (size_t)row * p.ld[l + 1] + col);
                    dsto[i] = p.off_w[l] + row * (p.ld[l + 1] + 1) + col;
                }
            }
#pragma unroll
            for (int i = 0; i < MAXF; i++) {
                if (dsto[i] >= 0) {
                    float *dst = smem + dsto[i];
                    dst[0] = v[i].x; dst[1] = v[i].y; dst[2] = v[i].z; dst[3] = v[i].w;
                }
            }
        }
    }
    __syncthreads();
    GNN_STAMP4(1);

    const int rr = t >> 8, rc = t & 255; // epilogue thread -> (row, column + 256*i)
    const bool live_row = row0 + rr < p.B;

    // ---- forward: layers 2 .. L-1 (SCE:172-194) ----
    for (int l = 2; l <= Lm; l++) {
        const int N = p.ld[l], G = (N + 63) / 64, gw = G * 64, KS = p.ks_fwd[l], k4n = p.kr[l - 1] / 4;
        if (wave < G * KS) {
            const int g = wave % G, ks = wave / G;
            rowblock_product<false>(smem + p.off_act[l - 1], p.ld[l - 1] + 4, smem + p.off_w[l - 1], N + 1,
                                    ks * k4n / KS, (ks + 1) * k4n / KS, g * 64, 0,
                                    smem + p.off_scratch + ks * 4 * gw, gw, lane);
        }
        __syncthreads();
        const bool last = (l == Lm);
        float *img = smem + (last ? p.off_logits : p.off_act[l]);
        for (int n = rc; n < N; n += 256) {
            float v = 0.f;
            for (int ks = 0; ks < KS; ks++) v += smem[p.off_scratch + (ks * 4 + rr) * gw + n];
            const bool live = live_row && n < p.d[l];
            if (last) {
                img[rr * (N + 4) + n] = live ? v : 0.f;
            } else {
                const float a = live ? act_fn(ACT, v) : 0.f;
                img[rr * (N + 4) + n] = a;
                p.act[l][(size_t)(row0 + rr) * N + n] = a;
            }
        }
        __syncthreads();
        GNN_STAMP4(4 + l);
    }
    GNN_STAMP4(2);

    // ---- output layer: wave 0, one DPP row of 16 lanes per batch row ----
    if (wave == 0) {
        const int N = p.ld[Lm], nt = p.d[Lm];
        const int m = lane >> 4, c0 = lane & 15;
        const int row = row0 + m;
        const bool lrow = row < p.B;
        const float *z = smem + p.off_logits + m * (N + 4);
        const float *y = smem + p.off_y + m * N;
        float *dimg = smem + p.off_dl[Lm] + m * (N + 4);
        float mx = -INFINITY, lsum = 0.f;
        int best = -1;
        if (OUTK == 0) {
            for (int c = c0; c < nt; c += 16) {
                const float v = z[c];
                if (v >= mx) { mx = v; best = c; }
            }
            row16_argmax(mx, best);
            float s = 0.f;
            for (int c = c0; c < nt; c += 16) s += __expf(z[c] - mx);
            s = row16_sum(s);
            const float inv = 1.f / s, lse = mx + __logf(s);
            for (int c = c0; c < N; c += 16) {
                const bool live = lrow && c < nt;
                const float pr = live ? __expf(z[c] - mx) * inv : 0.f;
                const float yy = (live && p.Y) ? y[c] : 0.f;
                const float dd = live ? pr - yy : 0.f;          // SCE:250
                if (p.prob) p.prob[(size_t)row * N + c] = pr;
                dimg[c] = dd;
                if (BACKWARD) p.delta[Lm][(size_t)row * N + c] = dd;
                if (live && yy != 0.f) lsum += yy * (lse - z[c]); // -y ln p, SCE:216
            }
        } else {
            for (int c = c0; c < N; c += 16) {
                const bool live = lrow && c < nt;
                const float a = act_fn(p.last_act, z[c]);
                const float yy = (live && p.Y) ? y[c] : 0.f;
                const float df = a - yy;
                const float dd = live ? df * act_prime_from_a(p.last_act, a) : 0.f; // GNN:267-271
                if (p.prob) p.prob[(size_t)row * N + c] = live ? a : 0.f;
                dimg[c] = dd;
                if (BACKWARD) p.delta[Lm][(size_t)row * N + c] = dd;
                if (live) {
                    lsum += 0.5f * df * df;
                    if (a >= mx) { mx = a; best = c; }
                }
            }
            row16_argmax(mx, best);
        }
        lsum = row16_sum(lsum);
        if (c0 == 0) {
            if (p.loss) p.loss[row] = lrow ? lsum : 0.f;
            if (p.label) p.label[row] = lrow ? best : -1;
        }
    }
    if (!BACKWARD) { if (STAMP) continue; return; }
    __syncthreads();
    GNN_STAMP4(3);

    // ---- backward data: delta_l = (delta_{l+1} . W_l^T) * f'(z_l), l = L-2 .. 1 (SCE:262-278) ----
    for (int l = Lm - 1; l >= 1; l--) {
        const int N = p.ld[l], NR = p.kr[l], G = (NR + 63) / 64, gw = G * 64, KS = p.ks_bwd[l];
        const int k4n = p.kr[l + 1] / 4;
        if (wave < G * KS) {
            const int g = wave % G, ks = wave / G;
            rowblock_product<true>(smem + p.off_dl[l + 1], p.ld[l + 1] + 4, smem + p.off_w[l], p.ld[l + 1] + 1,
                                   ks * k4n / KS, (ks + 1) * k4n / KS, g * 64, NR,
                                   smem + p.off_scratch + ks * 4 * gw, gw, lane);
        }
        __syncthreads();
        const float *aimg = smem + p.off_act[l] + rr * (N + 4);
        float *dimg = smem + p.off_dl[l] + rr * (N + 4);
        for (int n = rc; n < N; n += 256) {
            float v = 0.f;
            if (n < NR)
                for (int ks = 0; ks < KS; ks++) v += smem[p.off_scratch + (ks * 4 + rr) * gw + n];
            const bool live = live_row && n < p.d[l];
            const float dd = live ? v * act_prime_from_a(ACT, aimg[n]) : 0.f;
            if (l > 1) dimg[n] = dd;
            p.delta[l][(size_t)(row0 + rr) * N + n] = dd;
        }
        __syncthreads();
        GNN_STAMP4(10 + l);
    }
    GNN_STAMP4(4);
    } // pass
}

} // namespace gnn
