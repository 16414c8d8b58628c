// middle4_kernel.h -- the per-sample chain between A_1 and delta_1 for FOUR batch rows per
// workgroup, with every middle weight matrix resident in LDS.
//
// Why four rows: the chain (SCE:172-198 forward, SCE:249-251 output delta, SCE:262-278 backward)
// is per-sample independent, and a workgroup must pull ALL middle weights (W_1 is 120 KB for
// 784-300-100-10) through its CU.  Every kernel starts with a cold L2, so the weights are read
// from memory exactly ONCE per workgroup, all loads in flight together, into an LDS image that
// then serves both the forward product (A . W) and the backward product (delta . W^T).  With
// B/4 workgroups four of them share an XCD's L2 (blocks b and b+8): each starts its stream at a
// different quarter, so three quarters of its bytes are L2 hits.
//
// LDS image of W_l: [k][lw] floats, lw = 4*(odd) >= round_up(d_{l+1},4) + 4; only the logical
// columns are copied (120 KB instead of 134 KB for W_1).  The row stride is a multiple of 4
// floats, so the stream is written with ds_write_b128, and lw/4 is odd, so
//   forward : lane l reads W[k][n0+l]              -- consecutive addresses, conflict-free b32
//   backward: lane l reads W[n0+l][4q..4q+3] (b128) -- 16 lanes hit 16 different 16-B slots
//
// MFMA: v_mfma_f32_4x4x1_16b_f32 -- 16 blocks of (4x1).(1x4) per instruction.  Every block gets
// the same A column (the 4 batch rows' value at k; lane l supplies row l&3) and its own 4 B
// values (lane l supplies B[k][n0+l]), so one instruction is the rank-1 update
// Z[0..3][n0..n0+63] += a[0..3] (x) B[k][n0..n0+63]; accumulator register i of lane l is
// Z[row i][n0+l].  Exact f32, same FLOP rate as the 16x16x4 form, no padding rows.
//
// The kernel is written once over a "shape" policy: RuntimeShape reads the plan from the kernel
// arguments (any net that fits), StaticShape<d0,d1,...> makes every extent, LDS offset and K
// split a compile-time constant.  At this problem size a workgroup's critical path is a few
// thousand instructions, so kernarg loads, integer divisions and layer loops with runtime
// bounds are a large share of the step; the static instantiations remove them.
#pragma once
#include "fused_kernels.h"

namespace gnn {

// ---- LDS plan, computed identically on the host (runtime) and at compile time -----------------
struct Mid4Plan {
    int L;
    int d[MAX_LAYERS], ld[MAX_LAYERS], kr[MAX_LAYERS]; // kr = round_up(d, 4): rows kept / contracted
    int lw[MAX_LAYERS];      // row stride of weight image l: 4*(odd) floats >= kr[l+1] + 4
    int off_w[MAX_LAYERS];   // weight images l = 1..L-2: [kr[l]][lw[l]], only kr[l+1] columns are copied
    int off_act[MAX_LAYERS]; // activation images l = 1..L-2: [4][ld[l]+4]
    int off_dl[MAX_LAYERS];  // delta images l = 2..L-1: [4][ld[l]+4]
    int off_logits, off_y, off_scratch;
    int ks_fwd[MAX_LAYERS];  // K splits of the product giving layer l (l = 2..L-1)
    int ks_bwd[MAX_LAYERS];  // K splits of the product giving delta_l (l = 1..L-2)
    // staging of weight image l: slabs of st_rpt[l] rows x c4 = kr[l+1]/4 float4s, one float4 per thread
    int st_rpt[MAX_LAYERS], st_trips[MAX_LAYERS], st_begin[MAX_LAYERS], st_total; // slabs per layer / first slot / all
    unsigned inv_n4[MAX_LAYERS];    // ceil(2^22 / (ld[l]/4)): epilogue thread -> row = (t * inv) >> 22 (t < 1024, ld <= 1024)
    unsigned st_inv_c4[MAX_LAYERS]; // ceil(2^22 / c4): thread -> slab row = (t * inv) >> 22 (exact for t < 1024, c4 <= 256)
    // first-layer K slabs (SLABS kernels): ns slabs of 64 input neurons; the 4 x ld[1]/4 float4 elements of the four
    // A_1 rows are each summed by sgrp threads (thread t -> element t % ld[1], group t / ld[1]; group g takes the
    // slabs g, g + sgrp, ..), the group sums meet in the scratch area: every thread has loads in flight
    int ns, sgrp;
    unsigned inv_e;          // ceil(2^22 / ld[1])
    // BF16 kernels: f'(a_1) image [4][ld[1]+4] (for l >= 2 the f' image shares the slot of delta_l, which is free until
    // the backward pass writes it), and the staging of the bf16 weight shadow: c8 = ceil(kr[l+1]/8) 16-B pieces per row
    int off_fp1;
    int st8_rpt[MAX_LAYERS], st8_trips[MAX_LAYERS], st8_begin[MAX_LAYERS], st8_total;
    unsigned st8_inv_c8[MAX_LAYERS];
    int lds_floats;          // total dynamic LDS, floats
    bool ok;
};

__host__ __device__ constexpr int mid4_min(int a, int b) { return a < b ? a : b; }

__host__ __device__ constexpr Mid4Plan make_mid4_plan(const int *dims, int L, bool bf16 = false) {
    Mid4Plan m{};
    m.L = L;
    m.ok = false;
    if (L < 3 || L > MAX_LAYERS) return m;
    const int Lm = L - 1;
    for (int l = 0; l < L; l++) {
        m.d[l] = dims[l];
        m.ld[l] = (dims[l] + PAD - 1) / PAD * PAD;
        m.kr[l] = (dims[l] + 3) / 4 * 4;
    }
    for (int l = 0; l < L; l++) m.inv_n4[l] = ((1u << 22) + m.ld[l] / 4 - 1) / (m.ld[l] / 4);
    int off = 0;
    for (int l = 1; l < Lm; l++) {
        int lw4 = m.kr[l + 1] / 4 + 1;
        if (lw4 % 2 == 0) lw4++; // (row stride / 4) odd: b128 reads down a column of rows hit 16 different slots
        m.lw[l] = 4 * lw4;
        m.off_w[l] = off;
        off += m.kr[l] * m.lw[l];
    }
    for (int l = 1; l < Lm; l++) { m.off_act[l] = off; off += 4 * (m.ld[l] + 4); }
    m.off_logits = off; off += 4 * (m.ld[Lm] + 4);
    for (int l = 2; l <= Lm; l++) { m.off_dl[l] = off; off += 4 * (m.ld[l] + 4); }
    m.off_y = off; off += 4 * m.ld[Lm];
    m.off_fp1 = off;
    if (bf16) off += 4 * (m.ld[1] + 4);
    m.off_scratch = off;
    m.st_total = 0;
    for (int l = 1; l < Lm; l++) {
        const int c4 = m.kr[l + 1] / 4; // float4s copied per row (the padding columns stay in HBM)
        if (c4 > 256) return m;         // (the column-group limit below says the same)
        m.st_rpt[l] = 1024 / c4;
        m.st_trips[l] = (m.kr[l] + m.st_rpt[l] - 1) / m.st_rpt[l];
        m.st_begin[l] = m.st_total;
        m.st_total += m.st_trips[l];
        m.st_inv_c4[l] = ((1u << 22) + c4 - 1) / c4;
    }
    m.st8_total = 0;
    for (int l = 1; l < Lm; l++) {
        const int c8 = (m.kr[l + 1] + 7) / 8; // (the image row has room for the up to 4 extra zero columns: lw >= kr + 4)
        m.st8_rpt[l] = 1024 / c8;
        m.st8_trips[l] = (m.kr[l] + m.st8_rpt[l] - 1) / m.st8_rpt[l];
        m.st8_begin[l] = m.st8_total;
        m.st8_total += m.st8_trips[l];
        m.st8_inv_c8[l] = ((1u << 22) + c8 - 1) / c8;
    }
    const int budget = (160 * 1024) / 4 - 64 - off; // floats left for the K-split partials
    if (budget <= 0) return m;
    int scratch = 0;
    for (int l = 2; l <= Lm; l++) {
        const int G = (m.kr[l] + 63) / 64, gw = G * 64, k4n = m.kr[l - 1] / 4;
        if (G > 16 || 4 * gw > budget) return m;
        const int ks = mid4_min(mid4_min(16 / G, k4n), budget / (4 * gw));
        m.ks_fwd[l] = ks;
        if (ks * 4 * gw > scratch) scratch = ks * 4 * gw;
    }
    for (int l = Lm - 1; l >= 1; l--) {
        const int G = (m.kr[l] + 63) / 64, gw = G * 64, k4n = m.kr[l + 1] / 4;
        if (G > 16 || 4 * gw > budget) return m;
        const int ks = mid4_min(mid4_min(16 / G, k4n), budget / (4 * gw));
        m.ks_bwd[l] = ks;
        if (ks * 4 * gw > scratch) scratch = ks * 4 * gw;
    }
    m.ns = (m.ld[0] + 63) / 64;
    m.inv_e = ((1u << 22) + m.ld[1] - 1) / m.ld[1];
    m.sgrp = mid4_min(mid4_min(m.ns, 1024 / m.ld[1]), 4);
    if (m.sgrp < 1) m.sgrp = 1;
    while (m.sgrp > 1 && m.sgrp * 4 * m.ld[1] > budget) m.sgrp--;
    if (m.sgrp > 1 && m.sgrp * 4 * m.ld[1] > scratch) scratch = m.sgrp * 4 * m.ld[1];
    m.lds_floats = off + scratch + 64;
    m.ok = true;
    return m;
}

struct Mid4Params {
    Mid4Plan plan;               // used by RuntimeShape only
    const float *W[MAX_LAYERS];  // global W_l, l = 1..L-2
    float *act[MAX_LAYERS];      // act[1] in; act[2..L-2] out
    float *delta[MAX_LAYERS];    // delta[1..L-1] out
    const float *Y; int ldy;
    float *prob; float *loss; int32_t *label;
    int B;
    int inner_act, last_act;     // inner_act is read only by kernels built with ACT = -1
    unsigned long long *stamps;  // STAMP builds only
    const int32_t *row_idx;      // optional: expected row of batch row r is Y row row_idx[r] (sampled batches)
    // SLABS kernels: A_1 = f(sum of the K slabs of the first-layer sums) instead of reading act[1];
    // slab s of batch row b at slabs[(s * slab_rows + b) * ld[1]] (tile_step_kernel.h)
    const float *slabs; int slab_rows; int n_slabs;
    // BF16 kernels: the bf16 shadow of W_l (l = 1..L-2) and the bf16 outputs the tile kernel reads (same lds as the f32 ones)
    const __bf16 *Wb[MAX_LAYERS];
    __bf16 *actb[MAX_LAYERS];   // l = 1..L-2
    __bf16 *deltab[MAX_LAYERS]; // l = 1..L-1
};

typedef __bf16 m4_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 m4_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned m4_u32x4 __attribute__((ext_vector_type(4)));
// x rounded to bf16 (nearest even) and widened again: the value a bf16 GEMM operand carries
__device__ __forceinline__ float bf16_value(float x) { return (float)(__bf16)x; }

constexpr int MID4_MAX_SLABS = 16;

// NL > 0: the layer COUNT is a compile-time constant (extents stay kernel arguments), so the
// per-layer loops unroll with constant trip counts and constant kernarg offsets; NL = 0: any L.
template <int NL> struct RuntimeShape {
    static constexpr bool is_static = false;
    static constexpr int kL = NL;
};
template <int... DIMS> struct StaticShape {
    static constexpr bool is_static = true;
    static constexpr int kL = (int)sizeof...(DIMS);
    static constexpr int kDims[sizeof...(DIMS)] = {DIMS...};
    template <bool BF = false> __host__ __device__ static constexpr Mid4Plan make() {
        constexpr int dims[sizeof...(DIMS)] = {DIMS...};
        return make_mid4_plan(dims, (int)sizeof...(DIMS), BF);
    }
};

// ---- 16-lane (one DPP row) butterflies ----------------------------------------------------------
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
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
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

// partial[4][gw] (this wave's K part) = A_img[4 x Kpart] . B for column group n0..n0+63
//   TRANS = false: B[k][n] = Wimg[k*ldw + n]   (forward:  n = neuron of the layer being produced)
//   TRANS = true : B[k][n] = Wimg[n*ldw + k]   (backward: n = neuron of the layer receiving delta)
template <bool TRANS>
__device__ __forceinline__ void rowblock_product(const float *A_img, int lda, const float *Wimg, int ldw, int k4_begin,
                                                 int k4_end, int n0, int n_rows, float *partial, int gw, int lane) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float *arow = A_img + (lane & 3) * lda;
    const float *wp;
    if (TRANS) {
        int n = n0 + lane;
        n = n < n_rows ? n : n_rows - 1; // columns past the image compute garbage nobody reads
        wp = Wimg + n * ldw;
    } else {
        wp = Wimg + n0 + lane;
    }
    // U groups of 4 k per trip: every LDS read of the trip is issued before the first MFMA waits.
    // Full trips carry no tail masks and the last 0..U-1 groups run one by one: the kernel is
    // issue-bound (16 waves: every instruction of the stream costs 16 SIMD cycles), and masking a
    // partial trip cost 4 selects per group in EVERY trip.
    constexpr int U = 4;
    auto read_group = [&](int k4, f32x4 &a, f32x4 &b) {
        a = *reinterpret_cast<const f32x4 *>(arow + 4 * k4);
        if (TRANS) {
            b = *reinterpret_cast<const f32x4 *>(wp + 4 * k4);
        } else {
            const float *w = wp + (4 * k4) * ldw;
            b = (f32x4){w[0], w[ldw], w[2 * ldw], w[3 * ldw]};
        }
    };
    int kb = k4_begin; // bounds are wave-uniform
    for (; kb + U <= k4_end; kb += U) {
        f32x4 a[U], b[U];
#pragma unroll
        for (int u = 0; u < U; u++) read_group(kb + u, a[u], b[u]);
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
    for (; kb < k4_end; kb++) {
        f32x4 a, b;
        read_group(kb, a, b);
#pragma unroll
        for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc0, 0, 0, 0);
    }
    const f32x4 acc = acc0 + acc1;
#pragma unroll
    for (int i = 0; i < 4; i++) partial[i * gw + n0 + lane] = acc[i];
}

// NSLOT > 0: static shape, the number of weight slabs (Mid4Plan::st_total); 0: runtime extents
// NS: 0 = A_1 is read from act[1]; > 0 = that many first-layer K slabs (static shape); < 0 = p.n_slabs of them
// BF: GNN_DTYPE_BF16 -- every operand of the row-block products carries a bf16 value: the weights are staged from
// the bf16 shadow (half the bytes of the kernel's dominant load) and widened into the same f32 LDS image, activations
// and deltas are rounded when they are written to their operand images (f' is kept from the unrounded activation),
// and the products run on the same exact-f32 MFMA -- a product of two bf16 values is exact in f32, so this IS bf16
// operands with f32 accumulation; the kernel is bound by issue and loads, not by the MFMA rate.  Outputs for the
// tile kernel (A_l, delta_l) are written as bf16.  NSLOT8: static weight slabs of the bf16 staging.
// SGV: static shape, the number of slab groups (Mid4Plan::sgrp); 0: read from the plan
// 16 B at element offset `off` of `base`, the offset taken as a 32-bit BYTE offset (operands below 4 GB): the load is
// "SGPR base + one VGPR offset" with no 64-bit address arithmetic -- one instruction less per load in phases bound by issue
__device__ __forceinline__ f32x4 m4_load16(const float *base, unsigned off) {
    return *reinterpret_cast<const f32x4 *>(reinterpret_cast<const char *>(base) + (size_t)(off * 4u));
}
template <int NL, int ACT_T, int OUTK, bool BACKWARD, bool STAMP, int NSLOT, int NS, bool BF = false, int NSLOT8 = 0, int SGV = 0>
__device__ __forceinline__ void middle4_body(const Mid4Plan &m, Mid4Params &p) {
    const int ACT = (ACT_T >= 0) ? ACT_T : p.inner_act;
    auto opv = [](float x) { return BF ? bf16_value(x) : x; };  // the value an operand image holds
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int NT_ = 1024;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6); // provably wave-uniform: scalar branches
    const int row0 = blockIdx.x * 4;
    const int L = (NL > 0) ? NL : m.L, Lm = L - 1;
    // STAMP builds run the body twice and stamp the second (warm) pass into the next record
    for (int pass = 0; pass < (STAMP ? 2 : 1); pass++) {
    if (STAMP && pass == 1) { __syncthreads(); p.stamps += 16 * gridDim.x; }
    GNN_STAMP4(0);

    // ---- phase 0: everything this block reads from memory, issued before anything waits ----
    // A_1 rows and expected rows: one float4 per thread (4*ld/4 <= 1024 floats4 since kr <= 1024),
    // held in registers until the weight loads are in flight too -- written `lds = global` the
    // compiler waits for the load before it issues anything else
    const int q1 = m.ld[1] / 4, qy = m.ld[Lm] / 4;
    const bool a1_on = t < 4 * q1, y_on = p.Y != nullptr && NT_ - 1 - t < 4 * qy; // Y: the LAST threads
    const int a1_r = t / q1, a1_q = t - a1_r * q1, y_e = NT_ - 1 - t, y_r = y_e / qy, y_q = y_e - y_r * qy;
    // (lanes without an element re-read element 0 and never store it)
    f32x4 a1v = {0.f, 0.f, 0.f, 0.f};
    const int ns = NS > 0 ? NS : (NS < 0 ? p.n_slabs : 0); // block-uniform
    const int sg = (NS != 0) ? m.sgrp : 1;                 // threads per A_1 element (slab groups)
    // most slabs one thread may have to take: with SGV slab groups known at compile time ceil(NS / SGV) (13 slabs on 3 groups:
    // 5 registers, not 13 -- the unused ones were zeroed again at every scheduling fence below, ~30 instructions of this phase)
    constexpr int NSV = (NS > 0 && SGV > 0) ? (NS + SGV - 1) / SGV : NS > 0 ? NS : (NS < 0 ? MID4_MAX_SLABS : 1);
    f32x4 zs[NSV];
    // thread -> (element se of the 4 x ld[1]/4 float4s, slab group sgi); with sg = 1 this is (t, 0) for t < 4*q1
    const int sgi = (NS != 0 && sg > 1) ? (NSLOT > 0 ? t / m.ld[1] : (int)(((unsigned)t * m.inv_e) >> 22)) : 0;
    const int se = t - sgi * m.ld[1];
    const bool s_on = (sg > 1) ? sgi < sg : a1_on;
    const int s_r = (sg > 1) ? se / q1 : a1_r, s_q = (sg > 1) ? se - s_r * q1 : a1_q;
    if constexpr (NS == 0) {
        a1v = *reinterpret_cast<const f32x4 *>(p.act[1] + (a1_on ? (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4 : (size_t)0));
    } else {
        // 32-bit element offsets (the slab buffer is far below 2^32 floats: plan_chain checks): a 64-bit product per slab
        // load was ~7 instructions each in a phase that is bound by instruction issue (16 waves per CU)
        const unsigned zoff = s_on ? (unsigned)(row0 + s_r) * (unsigned)m.ld[1] + (unsigned)(s_q * 4) : 0u;
        const unsigned sstride = (unsigned)p.slab_rows * (unsigned)m.ld[1];
#pragma unroll
        for (int i = 0; i < NSV; i++) {
            const int s_ = sgi + i * sg;
            zs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s_ < ns) zs[i] = m4_load16(p.slabs, zoff + (s_on ? (unsigned)s_ * sstride : 0u));
        }
    }
    // this thread's slabs in ascending order; with one thread per element that is the whole sum: then f;
    // rows past the batch and columns past d_1 are zeros (f(0) != 0 for the sigmoid)
    auto finish_a1 = [&]() {
        if constexpr (NS != 0) {
#pragma unroll
            for (int i = 0; i < NSV; i++) asm volatile("" : "+v"(zs[i])); // every slab load issued before the first add waits
            f32x4 z = zs[0];
#pragma unroll
            for (int i = 1; i < NSV; i++)
                if (sgi + i * sg < ns) z += zs[i];
            if (sg > 1) { // group sum to the scratch area; the element's first thread finishes after the barrier
                if (s_on) *reinterpret_cast<f32x4 *>(smem + m.off_scratch + (sgi * m.ld[1] + se) * 4) = z;
            } else {
                const bool lrow = row0 + a1_r < p.B;
#pragma unroll
                for (int j = 0; j < 4; j++) a1v[j] = (lrow && a1_q * 4 + j < m.d[1]) ? act_fn(ACT, z[j]) : 0.f;
                if (BF) {
                    f32x4 fp;
#pragma unroll
                    for (int j = 0; j < 4; j++) { fp[j] = act_prime_from_a(ACT, a1v[j]); a1v[j] = bf16_value(a1v[j]); }
                    if (a1_on) {
                        *reinterpret_cast<f32x4 *>(smem + m.off_fp1 + a1_r * (m.ld[1] + 4) + a1_q * 4) = fp;
                        *reinterpret_cast<m4_bf16x4 *>(p.actb[1] + (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4) =
                            (m4_bf16x4){(__bf16)a1v[0], (__bf16)a1v[1], (__bf16)a1v[2], (__bf16)a1v[3]};
                    }
                } else if (a1_on) {
                    *reinterpret_cast<f32x4 *>(p.act[1] + (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4) = a1v; // the gradient kernel reads A_1
                }
            }
        }
    };
    size_t y_row = (size_t)(row0 + y_r);
    if (y_on && p.row_idx) y_row = row0 + y_r < p.B ? (size_t)p.row_idx[row0 + y_r] : 0; // rows past the batch are masked below
    f32x4 yv = *reinterpret_cast<const f32x4 *>((p.Y ? p.Y : p.act[1]) + (y_on ? y_row * p.ldy + y_q * 4 : (size_t)0));
    // Weight images. Layer j is walked in slabs of rpt rows x c4 float4s with thread -> (r0, c) fixed
    // inside the slab: a slab then costs each thread ONE add for its LDS address and none for the
    // global one (wave-uniform slab base + per-thread offset), where a flat index space over all
    // layers cost ~25 VALU instructions per float4 -- and with 16 waves per CU every instruction of
    // the stream is 16 issue cycles, which made staging issue-bound (2 us of the kernel).
    // The four blocks that share an XCD start at different quarters of the slab sequence.
    {
        const int q = (blockIdx.x >> 3) & 3; // wave-uniform
        if constexpr (BF) {
            // bf16 shadow: rows of c8 = ceil(kr/8) 16-B pieces (8 weights), widened to f32 into the same [k][lw] image
            auto widen_store = [&](float *dst, m4_bf16x8 w) {
                const m4_u32x4 u = __builtin_bit_cast(m4_u32x4, w);
                f32x4 lo, hi;
                lo[0] = __builtin_bit_cast(float, u[0] << 16); lo[1] = __builtin_bit_cast(float, u[0] & 0xffff0000u);
                lo[2] = __builtin_bit_cast(float, u[1] << 16); lo[3] = __builtin_bit_cast(float, u[1] & 0xffff0000u);
                hi[0] = __builtin_bit_cast(float, u[2] << 16); hi[1] = __builtin_bit_cast(float, u[2] & 0xffff0000u);
                hi[2] = __builtin_bit_cast(float, u[3] << 16); hi[3] = __builtin_bit_cast(float, u[3] & 0xffff0000u);
                *reinterpret_cast<f32x4 *>(dst) = lo;
                *reinterpret_cast<f32x4 *>(dst + 4) = hi;
            };
            constexpr int NV = NSLOT8 > 0 ? NSLOT8 : 10; // static: every slab of every layer in flight; runtime: up to 10 per batch
            m4_bf16x8 v[NV];
            bool rows_done = false;
            auto rows_to_lds = [&]() {
                if (rows_done) return;
                rows_done = true;
                asm volatile("" : "+v"(a1v), "+v"(yv));
                finish_a1();
                if (a1_on && sg == 1) *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a1v;
                if (y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv;
            };
            if constexpr (NSLOT8 > 0) {
#pragma unroll
                for (int j = 1; j < MAX_LAYERS - 1; j++) {
                    if (j < Lm) {
                        const int c8 = (m.kr[j + 1] + 7) >> 3, rpt = m.st8_rpt[j], trips = m.st8_trips[j];
                        const int r0 = t / c8, c = t - r0 * c8;
                        const unsigned goff = (unsigned)(r0 * m.ld[j + 1] + 8 * c);
                        const int first = (trips * q) >> 2;
#pragma unroll
                        for (int i = 0; i < NSLOT8; i++) {
                            if (i < trips) {
                                int tr = i + first;
                                tr = tr >= trips ? tr - trips : tr;
                                const int rs = tr * rpt;
                                const bool ok = r0 < rpt && r0 + rs < m.kr[j];
                                v[m.st8_begin[j] + i] = *reinterpret_cast<const m4_bf16x8 *>(p.Wb[j] + (ok ? (unsigned)(rs * m.ld[j + 1]) + goff : 0u));
                            }
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < NSLOT8; i++) asm volatile("" : "+v"(v[i]));
                rows_to_lds();
#pragma unroll
                for (int j = 1; j < MAX_LAYERS - 1; j++) {
                    if (j < Lm) {
                        const int c8 = (m.kr[j + 1] + 7) >> 3, rpt = m.st8_rpt[j], trips = m.st8_trips[j];
                        const int r0 = t / c8, c = t - r0 * c8;
                        float *dst = smem + m.off_w[j] + r0 * m.lw[j] + 8 * c;
                        const int first = (trips * q) >> 2;
#pragma unroll
                        for (int i = 0; i < NSLOT8; i++) {
                            if (i < trips) {
                                int tr = i + first;
                                tr = tr >= trips ? tr - trips : tr;
                                const int rs = tr * rpt;
                                if (r0 < rpt && r0 + rs < m.kr[j]) widen_store(dst + rs * m.lw[j], v[m.st8_begin[j] + i]);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
                for (int j = 1; j < MAX_LAYERS - 1; j++) {
                    if (j < Lm) {
                        const int c8 = (m.kr[j + 1] + 7) >> 3, rpt = m.st8_rpt[j], trips = m.st8_trips[j];
                        const int r0 = (int)(((unsigned)t * m.st8_inv_c8[j]) >> 22), c = t - r0 * c8;
                        const unsigned goff = (unsigned)(r0 * m.ld[j + 1] + 8 * c);
                        float *dst = smem + m.off_w[j] + r0 * m.lw[j] + 8 * c;
                        const int first = (trips * q) >> 2;
                        for (int tb = 0; tb < trips; tb += NV) {
#pragma unroll
                            for (int i = 0; i < NV; i++) {
                                int tr = tb + i + first;
                                tr = tr >= trips ? tr - trips : tr;
                                const int rs = tr * rpt;
                                const bool ok = tb + i < trips && r0 < rpt && r0 + rs < m.kr[j];
                                v[i] = *reinterpret_cast<const m4_bf16x8 *>(p.Wb[j] + (ok ? (unsigned)(rs * m.ld[j + 1]) + goff : 0u));
                            }
#pragma unroll
                            for (int i = 0; i < NV; i++) asm volatile("" : "+v"(v[i]));
                            rows_to_lds(); // the first weight loads are in flight: now the rows
#pragma unroll
                            for (int i = 0; i < NV; i++) {
                                int tr = tb + i + first;
                                tr = tr >= trips ? tr - trips : tr;
                                const int rs = tr * rpt;
                                if (tb + i < trips && r0 < rpt && r0 + rs < m.kr[j]) widen_store(dst + rs * m.lw[j], v[i]);
                            }
                        }
                    }
                }
                rows_to_lds();
            }
        } else if constexpr (NSLOT > 0) {
            // static shape: every load of every layer in flight before the first LDS write
            f32x4 v[NSLOT];
#pragma unroll
            for (int j = 1; j < MAX_LAYERS - 1; j++) {
                if (j < Lm) {
                    const int c4 = m.kr[j + 1] >> 2, rpt = m.st_rpt[j], trips = m.st_trips[j];
                    const int r0 = t / c4, c = t - r0 * c4;
                    const unsigned goff = (unsigned)(r0 * m.ld[j + 1] + 4 * c);
                    const int first = (trips * q) >> 2;
#pragma unroll
                    for (int i = 0; i < NSLOT; i++) {
                        if (i < trips) {
                            int tr = i + first;
                            tr = tr >= trips ? tr - trips : tr;
                            const int rs = tr * rpt;
                            const bool ok = r0 < rpt && r0 + rs < m.kr[j];
                            // unconditional: lanes without a row re-read the image's first float4 and
                            // never store it (a load under `if` gets sunk next to its store, serialising)
                            v[m.st_begin[j] + i] = m4_load16(p.W[j], ok ? (unsigned)(rs * m.ld[j + 1]) + goff : 0u);
                        }
                    }
                }
            }
            // pin every load above this point: left alone, the compiler sinks each load next to its
            // (conditional) LDS store and the stream degenerates to load - wait - store per slab
            asm volatile("" : "+v"(a1v), "+v"(yv));
#pragma unroll
            for (int i = 0; i < NSLOT; i++) asm volatile("" : "+v"(v[i]));
            finish_a1();
            if (a1_on && sg == 1) *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a1v;
            if (y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv;
#pragma unroll
            for (int j = 1; j < MAX_LAYERS - 1; j++) {
                if (j < Lm) {
                    const int c4 = m.kr[j + 1] >> 2, rpt = m.st_rpt[j], trips = m.st_trips[j];
                    const int r0 = t / c4, c = t - r0 * c4;
                    float *dst = smem + m.off_w[j] + r0 * m.lw[j] + 4 * c;
                    const int first = (trips * q) >> 2;
#pragma unroll
                    for (int i = 0; i < NSLOT; i++) {
                        if (i < trips) {
                            int tr = i + first;
                            tr = tr >= trips ? tr - trips : tr;
                            const int rs = tr * rpt;
                            if (r0 < rpt && r0 + rs < m.kr[j]) *reinterpret_cast<f32x4 *>(dst + rs * m.lw[j]) = v[m.st_begin[j] + i];
                        }
                    }
                }
            }
        } else {
            // runtime extents: layer by layer, up to MAXF slabs in flight
            constexpr int MAXF = 10;
#pragma unroll
            for (int j = 1; j < MAX_LAYERS - 1; j++) {
                if (j < Lm) {
                    const int c4 = m.kr[j + 1] >> 2, rpt = m.st_rpt[j], trips = m.st_trips[j];
                    const int r0 = (int)(((unsigned)t * m.st_inv_c4[j]) >> 22), c = t - r0 * c4;
                    const unsigned goff = (unsigned)(r0 * m.ld[j + 1] + 4 * c);
                    float *dst = smem + m.off_w[j] + r0 * m.lw[j] + 4 * c;
                    const int first = (trips * q) >> 2;
                    for (int tb = 0; tb < trips; tb += MAXF) {
                        f32x4 v[MAXF];
#pragma unroll
                        for (int i = 0; i < MAXF; i++) {
                            int tr = tb + i + first;
                            tr = tr >= trips ? tr - trips : tr;
                            const int rs = tr * rpt;
                            const bool ok = tb + i < trips && r0 < rpt && r0 + rs < m.kr[j];
                            v[i] = *reinterpret_cast<const f32x4 *>(p.W[j] + (ok ? (unsigned)(rs * m.ld[j + 1]) + goff : 0u));
                        }
#pragma unroll
                        for (int i = 0; i < MAXF; i++) asm volatile("" : "+v"(v[i])); // pin the loads (see above)
                        if (j == 1 && tb == 0) { // the first weight loads are in flight: now the rows
                            asm volatile("" : "+v"(a1v), "+v"(yv));
                            finish_a1();
                            if (a1_on && sg == 1) *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a1v;
                            if (y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv;
                        }
#pragma unroll
                        for (int i = 0; i < MAXF; i++) {
                            int tr = tb + i + first;
                            tr = tr >= trips ? tr - trips : tr;
                            const int rs = tr * rpt;
                            if (tb + i < trips && r0 < rpt && r0 + rs < m.kr[j]) *reinterpret_cast<f32x4 *>(dst + rs * m.lw[j]) = v[i];
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    if (NS != 0 && sg > 1) {
        // group sums in group order (fixed), then f: the first thread of every element
        if (a1_on) {
            f32x4 z = *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + t * 4);
            for (int g = 1; g < sg; g++) z += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (g * m.ld[1] + t) * 4);
            const bool lrow = row0 + a1_r < p.B;
#pragma unroll
            for (int j = 0; j < 4; j++) a1v[j] = (lrow && a1_q * 4 + j < m.d[1]) ? act_fn(ACT, z[j]) : 0.f;
            if (BF) {
                f32x4 fp;
#pragma unroll
                for (int j = 0; j < 4; j++) { fp[j] = act_prime_from_a(ACT, a1v[j]); a1v[j] = bf16_value(a1v[j]); }
                *reinterpret_cast<f32x4 *>(smem + m.off_fp1 + a1_r * (m.ld[1] + 4) + a1_q * 4) = fp;
                *reinterpret_cast<m4_bf16x4 *>(p.actb[1] + (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4) =
                    (m4_bf16x4){(__bf16)a1v[0], (__bf16)a1v[1], (__bf16)a1v[2], (__bf16)a1v[3]};
            } else {
                *reinterpret_cast<f32x4 *>(p.act[1] + (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4) = a1v; // the gradient kernel reads A_1
            }
            *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a1v;
        }
        __syncthreads();
    }
    GNN_STAMP4(1);

    constexpr bool IS_STATIC = NSLOT > 0;

    // ---- forward: layers 2 .. L-1 (SCE:172-194) ----
    // Row tail: with at most 16 classes and a last hidden layer of at most 128 neurons, ONE wave per batch
    // row does the last layer, the output rule and delta_{L-2} on its own (vector FMAs, the wave's own LDS
    // traffic, no workgroup barrier in between): as three barrier-separated phases these took 3 400 cycles
    // in which one to six waves worked and the rest were parked (SQ_WAIT_ANY was 61 % of the wave cycles).
    const bool rowtail = Lm >= 2 && m.ld[Lm] == 16 && m.kr[Lm - 1] <= 128;
#pragma unroll
    for (int l = 2; l < MAX_LAYERS; l++) {
        if (l > Lm) break;
        if (rowtail && l == Lm) break;
        const int N = m.ld[l], G = (m.kr[l] + 63) / 64, gw = G * 64, KS = m.ks_fwd[l], k4n = m.kr[l - 1] / 4;
        if (wave < G * KS) {
            const int g = wave % G, ks = wave / G;
            rowblock_product<false>(smem + m.off_act[l - 1], m.ld[l - 1] + 4, smem + m.off_w[l - 1], m.lw[l - 1],
                                    ks * k4n / KS, (ks + 1) * k4n / KS, g * 64, 0,
                                    smem + m.off_scratch + ks * 4 * gw, gw, lane);
        }
        __syncthreads();
        if (l < Lm) {
            // K parts summed, f applied: one float4 per thread (4 rows x ld/4 <= 1024 float4s); the
            // logits of the last layer are summed by the output wave itself, below
            const int n4 = N >> 2;
            const int er = IS_STATIC ? t / n4 : (int)(((unsigned)t * m.inv_n4[l]) >> 22), n = 4 * (t - er * n4);
            if (t < 4 * n4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < m.kr[l])
                    for (int ks = 0; ks < KS; ks++) v += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (ks * 4 + er) * gw + n);
                const bool lrow = row0 + er < p.B;
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (lrow && n + j < m.d[l]) ? act_fn(ACT, v[j]) : 0.f;
                if (BF) { // f'(a_l) parks in delta_l's image until the backward pass overwrites it with delta_l
                    f32x4 fp;
#pragma unroll
                    for (int j = 0; j < 4; j++) { fp[j] = act_prime_from_a(ACT, v[j]); v[j] = bf16_value(v[j]); }
                    *reinterpret_cast<f32x4 *>(smem + m.off_dl[l] + er * (N + 4) + n) = fp;
                    *reinterpret_cast<m4_bf16x4 *>(p.actb[l] + (size_t)(row0 + er) * N + n) = (m4_bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                } else {
                    *reinterpret_cast<f32x4 *>(p.act[l] + (size_t)(row0 + er) * N + n) = v;
                }
                *reinterpret_cast<f32x4 *>(smem + m.off_act[l] + er * (N + 4) + n) = v;
            }
            __syncthreads();
        }
        GNN_STAMP4(4 + l);
    }
    GNN_STAMP4(2);

    // ---- output layer: wave 0, one DPP row of 16 lanes per batch row ----
    if (rowtail) {
        if (wave < 4) {
            const int r = wave, row = row0 + r;
            const int K = m.kr[Lm - 1], nt = m.d[Lm], lwl = m.lw[Lm - 1], ldp = m.ld[Lm - 1];
            const float *a = smem + m.off_act[Lm - 1] + r * (ldp + 4);
            const float *Wl = smem + m.off_w[Lm - 1];
            const int ks = lane >> 4, c = lane & 15;
            // logits: lane (ks, c) sums k = ks, ks+4, .. of column c; the four partial sums meet by lane exchange
            float zv = 0.f;
            if (c < m.kr[Lm]) {
                constexpr int UK = 8; // reads of 8 steps in flight before their FMAs: a lone wave hides no LDS latency
                int k = ks;
                for (; k + 4 * (UK - 1) < K; k += 4 * UK) {
                    float av[UK], wv[UK];
#pragma unroll
                    for (int u = 0; u < UK; u++) { av[u] = a[k + 4 * u]; wv[u] = Wl[(k + 4 * u) * lwl + c]; }
#pragma unroll
                    for (int u = 0; u < UK; u++) zv = __builtin_fmaf(av[u], wv[u], zv);
                }
                for (; k < K; k += 4) zv = __builtin_fmaf(a[k], Wl[k * lwl + c], zv);
            }
            zv += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, __builtin_bit_cast(int, zv)));
            zv += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, zv)));
            // output rule on the DPP row of 16 (all four rows of the wave hold the same 16 logits)
            const bool valid = c < nt, live = row < p.B && valid;
            const float yy = (live && p.Y) ? smem[m.off_y + r * 16 + c] : 0.f;
            float out, dd, lterm, mx;   // reported output, delta, this lane's loss term, argmax key
            int best = valid ? c : -1;
            float nan_flag = 0.f;
            float lse = 0.f;
            if (OUTK == 0) {
                mx = valid ? zv : -__builtin_inff();
                if (p.label) { // MT:166-168 incl. the NaN rule (see output_layer_kernel): any NaN logit -> label 0
                    nan_flag = (valid && zv != zv) ? 1.f : 0.f;
                    row16_argmax(mx, best);
                } else {
                    mx = row16_max(mx);
                }
                const float e = valid ? __expf(zv - mx) : 0.f;
                const float s = row16_sum(e);
                out = live ? e * (1.f / s) : 0.f;
                dd = live ? out - yy : 0.f;                          // SCE:250
                if (p.loss) lse = mx + __logf(s);
                lterm = (live && yy != 0.f) ? yy * (lse - zv) : 0.f; // -y ln p, SCE:216
            } else {
                const float av = act_fn(p.last_act, zv);             // GNN:215-218
                const float df = av - yy;
                out = live ? av : 0.f;
                dd = live ? df * act_prime_from_a(p.last_act, av) : 0.f; // GNN:267-271
                lterm = live ? 0.5f * df * df : 0.f;
                const bool in_scan = live && av == av;               // `x >= NaN` and `NaN >= x` are false: a NaN is never selected
                mx = in_scan ? av : -__builtin_inff();
                if (!in_scan) best = -1;
                if (p.label) { // element-wise output: only a NaN at index 0 is sticky (MT:166-168)
                    nan_flag = (live && c == 0 && av != av) ? 1.f : 0.f;
                    row16_argmax(mx, best);
                }
            }
            if (p.label && row16_sum(nan_flag) > 0.f) best = 0;
            float *dlast = smem + m.off_dl[Lm] + r * (16 + 4);
            if (ks == 0) {
                if (p.prob) p.prob[(size_t)row * 16 + c] = out;
                dlast[c] = opv(dd);
                if (BACKWARD) {
                    if (BF) p.deltab[Lm][(size_t)row * 16 + c] = (__bf16)dd;
                    else p.delta[Lm][(size_t)row * 16 + c] = dd;
                }
            }
            if (p.loss) {
                const float lsum = row16_sum(lterm);
                if (lane == 0) p.loss[row] = row < p.B ? lsum : 0.f;
            }
            if (p.label && lane == 0) p.label[row] = row < p.B ? best : -1;
            if (BACKWARD) {
                // delta_{L-2}[n] = (sum_c delta_{L-1}[c] W[n][c]) f'(a[n]): lane n and n + 64; the wave reads back
                // its own 16 deltas (LDS keeps a wave's accesses in order); only the copied columns of W are used
                f32x4 d4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) d4[q] = *reinterpret_cast<const f32x4 *>(dlast + 4 * q);
#pragma unroll
                for (int half = 0; half < 2; half++) {
                    const int n = lane + 64 * half;
                    if (n < ldp) {
                        float acc = 0.f;
                        if (n < K) {
#pragma unroll
                            for (int q = 0; q < 4; q++) {
                                if (4 * q < m.kr[Lm]) {
                                    const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wl + n * lwl + 4 * q);
#pragma unroll
                                    for (int j = 0; j < 4; j++) acc = __builtin_fmaf(d4[q][j], w4[j], acc);
                                }
                            }
                        }
                        // (BF: f' of the unrounded activation was parked by the forward pass -- in delta_{L-2}'s own slot, or
                        //  in the f'(a_1) image when layer L-2 is layer 1)
                        const float fpv = BF ? smem[((Lm - 1 > 1) ? m.off_dl[Lm - 1] : m.off_fp1) + r * (ldp + 4) + n] : act_prime_from_a(ACT, a[n]);
                        const float v = (row < p.B && n < m.d[Lm - 1]) ? acc * fpv : 0.f;
                        if (Lm - 1 > 1) smem[m.off_dl[Lm - 1] + r * (ldp + 4) + n] = opv(v);
                        if (BF) p.deltab[Lm - 1][(size_t)row * ldp + n] = (__bf16)v;
                        else p.delta[Lm - 1][(size_t)row * ldp + n] = v;
                    }
                }
            }
        }
    } else if (wave == 0 && OUTK == 0 && m.ld[Lm] == 16) {
        // at most 16 classes: ONE logit per lane, kept in registers -- the general form below walks the
        // row three times through LDS and always reduces loss, label and NaN flag; this wave works alone
        // while the other fifteen wait, so its dependent chain is kernel time (1 900 -> ~700 cycles)
        const int nt = m.d[Lm];
        const int mr = lane >> 4, c0 = lane & 15;
        const int row = row0 + mr;
        const bool valid = c0 < nt, live = row < p.B && valid;
        float zv = 0.f;
        {
            const int KS = m.ks_fwd[Lm]; // (kr <= 16: one column group of 64)
            if (c0 < m.kr[Lm])
                for (int ks = 0; ks < KS; ks++) zv += smem[m.off_scratch + (ks * 4 + mr) * 64 + c0];
        }
        float mx = valid ? zv : -__builtin_inff();
        int best = valid ? c0 : -1;
        if (p.label) { // MT:166-168 incl. the NaN rule (see output_layer_kernel)
            const float nan_flag = (valid && zv != zv) ? 1.f : 0.f;
            row16_argmax(mx, best);
            if (row16_sum(nan_flag) > 0.f) best = 0;
        } else {
            mx = row16_max(mx);
        }
        const float e = valid ? __expf(zv - mx) : 0.f;
        const float s = row16_sum(e);
        const float pr = live ? e * (1.f / s) : 0.f;
        const float yy = (live && p.Y) ? smem[m.off_y + mr * 16 + c0] : 0.f;
        const float dd = live ? pr - yy : 0.f;                       // SCE:250
        if (p.prob) p.prob[(size_t)row * 16 + c0] = pr;
        smem[m.off_dl[Lm] + mr * (16 + 4) + c0] = opv(dd);
        if (BACKWARD) {
            if (BF) p.deltab[Lm][(size_t)row * 16 + c0] = (__bf16)dd;
            else p.delta[Lm][(size_t)row * 16 + c0] = dd;
        }
        if (p.loss) {
            const float lse = mx + __logf(s);
            const float lsum = row16_sum((live && yy != 0.f) ? yy * (lse - zv) : 0.f); // -y ln p, SCE:216
            if (c0 == 0) p.loss[row] = row < p.B ? lsum : 0.f;
        }
        if (p.label && c0 == 0) p.label[row] = row < p.B ? best : -1;
    } else if (wave == 0) {
        const int N = m.ld[Lm], nt = m.d[Lm];
        const int mr = lane >> 4, c0 = lane & 15;
        const int row = row0 + mr;
        const bool lrow = row < p.B;
        float *z = smem + m.off_logits + mr * (N + 4);
        {   // logits = sum of the K parts of the last product; every lane reads back only what it wrote
            const int G = (m.kr[Lm] + 63) / 64, gw = G * 64, KS = m.ks_fwd[Lm];
            for (int c = c0; c < N; c += 16) {
                float v = 0.f;
                if (c < m.kr[Lm])
                    for (int ks = 0; ks < KS; ks++) v += smem[m.off_scratch + (ks * 4 + mr) * gw + c];
                z[c] = (lrow && c < nt) ? v : 0.f;
            }
        }
        const float *y = smem + m.off_y + mr * N;
        float *dimg = smem + m.off_dl[Lm] + mr * (N + 4);
        float mx = -__builtin_inff(), lsum = 0.f, nan_flag = 0.f; // MT:166-168 NaN rule, see output_layer_kernel
        int best = -1;
        if (OUTK == 0) {
            for (int c = c0; c < nt; c += 16) {
                const float v = z[c];
                if (v != v) nan_flag = 1.f;
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
                dimg[c] = opv(dd);
                if (BACKWARD) {
                    if (BF) p.deltab[Lm][(size_t)row * N + c] = (__bf16)dd;
                    else p.delta[Lm][(size_t)row * N + c] = dd;
                }
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
                dimg[c] = opv(dd);
                if (BACKWARD) {
                    if (BF) p.deltab[Lm][(size_t)row * N + c] = (__bf16)dd;
                    else p.delta[Lm][(size_t)row * N + c] = dd;
                }
                if (live) {
                    lsum += 0.5f * df * df;
                    if (c == 0 && a != a) nan_flag = 1.f;
                    if (a >= mx) { mx = a; best = c; }
                }
            }
            row16_argmax(mx, best);
        }
        if (row16_sum(nan_flag) > 0.f) best = 0;
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
#pragma unroll
    for (int li = 0; li < MAX_LAYERS; li++) {
        const int l = Lm - 1 - li;
        if (l < 1) break;
        if (rowtail && l == Lm - 1) continue; // done by the row tail
        const int N = m.ld[l], NR = m.kr[l], G = (NR + 63) / 64, gw = G * 64, KS = m.ks_bwd[l];
        const int k4n = m.kr[l + 1] / 4;
        if (wave < G * KS) {
            const int g = wave % G, ks = wave / G;
            rowblock_product<true>(smem + m.off_dl[l + 1], m.ld[l + 1] + 4, smem + m.off_w[l], m.lw[l],
                                   ks * k4n / KS, (ks + 1) * k4n / KS, g * 64, NR,
                                   smem + m.off_scratch + ks * 4 * gw, gw, lane);
        }
        __syncthreads();
        {
            const int n4 = N >> 2;
            const int er = IS_STATIC ? t / n4 : (int)(((unsigned)t * m.inv_n4[l]) >> 22), n = 4 * (t - er * n4);
            if (t < 4 * n4) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < NR)
                    for (int ks = 0; ks < KS; ks++) v += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (ks * 4 + er) * gw + n);
                // f'(z_l): from a_l = f(z_l); BF: parked by the forward pass (from the unrounded a_l) in delta_l's slot / the f'(a_1) image
                const f32x4 a = *reinterpret_cast<const f32x4 *>(smem + (BF ? (l > 1 ? m.off_dl[l] : m.off_fp1) : m.off_act[l]) + er * (N + 4) + n);
                const bool lrow = row0 + er < p.B;
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (lrow && n + j < m.d[l]) ? v[j] * (BF ? a[j] : act_prime_from_a(ACT, a[j])) : 0.f;
                if (BF) {
                    *reinterpret_cast<m4_bf16x4 *>(p.deltab[l] + (size_t)(row0 + er) * N + n) = (m4_bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
#pragma unroll
                    for (int j = 0; j < 4; j++) v[j] = bf16_value(v[j]);
                } else {
                    *reinterpret_cast<f32x4 *>(p.delta[l] + (size_t)(row0 + er) * N + n) = v;
                }
                if (l > 1) *reinterpret_cast<f32x4 *>(smem + m.off_dl[l] + er * (N + 4) + n) = v;
            }
        }
        __syncthreads();
        GNN_STAMP4(10 + l);
    }
    GNN_STAMP4(4);
    } // pass
}

// SLABS: A_1 comes as K slabs from tile_step_kernel (64 input neurons per slab) instead of from act[1]
// BF16 : GNN_DTYPE_BF16 (see middle4_body); built only as the training kernel of the two-launch path (BACKWARD, SLABS)
template <class SH, int ACT, int OUTK, bool BACKWARD, bool STAMP = false, bool SLABS = false, bool BF16 = false>
__global__ __launch_bounds__(1024) void middle4_kernel(Mid4Params p) {
    static_assert(!BF16 || (BACKWARD && SLABS), "the bf16 row-block kernel exists for the two-launch training path only");
    if constexpr (SH::is_static) {
        // a LOCAL constexpr object: every member access with a compile-time index folds to an
        // immediate (a namespace-scope constant would be loaded from memory)
        constexpr Mid4Plan m = SH::template make<BF16>();
        constexpr int ns = SLABS ? (m.ld[0] + 63) / 64 : 0;
        static_assert(ns <= MID4_MAX_SLABS, "too many first-layer slabs for the register-resident sum");
        middle4_body<SH::kL, ACT, OUTK, BACKWARD, STAMP, m.st_total, ns, BF16, m.st8_total, (ns > 0 ? m.sgrp : 0)>(m, p);
    } else {
        middle4_body<SH::kL, ACT, OUTK, BACKWARD, STAMP, 0, SLABS ? -1 : 0, BF16, 0>(p.plan, p);
    }
}

} // namespace gnn
