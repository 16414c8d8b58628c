// rowblock_kernel.h -- the TRAINING row-block kernel of the two-launch step (round 3): the per-sample chain
// A_1 -> delta_1 (SCE:172-198 forward, SCE:249-251 output delta, SCE:262-278 backward) for FOUR batch rows per
// workgroup, eight waves, with the middle weights streamed from memory straight into the registers of the wave that
// multiplies them.
//
// What was wrong with the first form (middle4_kernel.h, still the inference / bf16 / three-launch kernel): a workgroup
// pulled ALL middle weights (124 KB for 784-300-100-10) into an LDS image, waited at a barrier, and only then started the
// layer-2 product -- load+stage was 7 400 cycles and the product another 4 200 of an 18 600-cycle critical path that IS
// the kernel's duration (profiles/r02, perf_harness stamps).  A workgroup's load path moves ~39 B/clk, so the 190 KB it
// needs take ~4 800 cycles whatever else happens; everything that waits for ALL of them to land is added on top.
//
// Here the forward product Z_{l+1}[4 x N] = A_l[4 x K] . W_l[K x N] is cut along K into eight slices, one per wave, and
// a wave loads exactly the rows of W_l it multiplies: 16 B per lane, lanes 0..31 -> 128 columns of row k, lanes 32..63 ->
// 128 columns of row k + 4 (v_mfma_f32_4x4x1_16b_f32 is sixteen independent (4x1).(1x4) blocks, so the two half-waves
// may sit at different k).  Nothing but the slab sums (A_1) stands between a wave's loads and its MFMAs: no barrier on
// the weights, no LDS round trip.  The same registers are then written to the [k][lw] LDS image that the BACKWARD
// product (delta . W^T, rows of the image down a column of lanes) reads later -- off the critical path.  K partial
// tiles meet in LDS in slice order (fixed: results do not depend on timing).
//
// Eight waves (512 threads): two per SIMD, 256 VGPRs per lane -- a thread holds its 13 slab pieces (52 registers) and up
// to six 8-row units of weights (96) at once, so every load of the kernel is in flight before anything waits; with 16
// waves the same data needed three threads per A_1 element and an extra LDS pass + barrier.
//
// Applies when the net has at most 16 outputs and a last hidden layer of at most 128 neurons (the "row tail": one wave
// per batch row does the last layer, the output rule and delta_{L-2}), every middle layer is at most 1024 wide and the
// plan fits LDS and the register budget (RbPlan::ok); otherwise the handle keeps middle4_kernel.
#pragma once
#include "middle4_kernel.h"

namespace gnn {

constexpr int RB_NW = 8, RB_NT = RB_NW * 64;
constexpr int RB_MAXU = 6; // 8-row weight units a wave may hold in registers at once (16 VGPRs each)

struct RbPlan {
    int L;
    int d[MAX_LAYERS], ld[MAX_LAYERS], kr[MAX_LAYERS];
    int lw[MAX_LAYERS];      // row stride of LDS weight image l (l = 1..L-2): 4*(odd) >= kr[l+1]
    int off_w[MAX_LAYERS];   // image l: [kr[l]][lw[l]]
    int off_act[MAX_LAYERS]; // activation images l = 1..L-2: [4][ld[l]+4]
    int off_dl[MAX_LAYERS];  // delta images l = 2..L-1: [4][ld[l]+4]
    int off_y, off_scratch;
    // forward register products giving layer l+1 from image-less W_l, l = 1..L-3 (the last product is the row tail's)
    int cs[MAX_LAYERS];      // column slices of 128 (waves = ksf * cs = 8)
    int ksf[MAX_LAYERS];     // K slices = partial tiles per output element
    int units[MAX_LAYERS];   // 8-row units along K = ld[l] / 8
    int upw[MAX_LAYERS];     // most units one wave takes
    // backward LDS products giving delta_l, l = L-3..1: column groups of 64 x K slices, dealt to the waves round robin
    int gb[MAX_LAYERS], ksb[MAX_LAYERS];
    int ns;                  // first-layer K slabs
    int lds_floats;
    bool ok;
};

__host__ __device__ constexpr int rb_min(int a, int b) { return a < b ? a : b; }
__host__ __device__ constexpr int rb_max(int a, int b) { return a > b ? a : b; }

__host__ __device__ constexpr RbPlan make_rb_plan(const int *dims, int L) {
    RbPlan m{};
    m.L = L;
    m.ok = false;
    if (L < 3 || L > MAX_LAYERS) return m;
    const int Lm = L - 1;
    for (int l = 0; l < L; l++) {
        m.d[l] = dims[l];
        m.ld[l] = (dims[l] + PAD - 1) / PAD * PAD;
        m.kr[l] = (dims[l] + 3) / 4 * 4;
    }
    if (m.ld[Lm] != 16 || m.kr[Lm - 1] > 128) return m; // the row tail's conditions
    for (int l = 1; l < Lm; l++) if (m.ld[l] > 1024) return m;
    m.ns = (m.ld[0] + 63) / 64;
    if (m.ns > MID4_MAX_SLABS) return m;
    if (4 * (m.ld[1] / 4) > RB_NT) return m; // one A_1 float4 per thread
    int off = 0;
    for (int l = 1; l < Lm; l++) {
        int lw4 = m.kr[l + 1] / 4;
        if (lw4 % 2 == 0) lw4++; // (row stride / 4) odd: b128 reads down a column of rows hit different 16-B slots
        m.lw[l] = 4 * lw4;
        m.off_w[l] = off;
        off += m.kr[l] * m.lw[l];
    }
    for (int l = 1; l < Lm; l++) { m.off_act[l] = off; off += 4 * (m.ld[l] + 4); }
    for (int l = 2; l <= Lm; l++) { m.off_dl[l] = off; off += 4 * (m.ld[l] + 4); }
    m.off_y = off; off += 4 * m.ld[Lm];
    m.off_scratch = off;
    int scratch = 0;
    for (int l = 1; l + 1 < Lm; l++) { // forward products from registers
        const int N = m.ld[l + 1];
        int cs = 1;
        while (cs * 128 < N) cs *= 2;
        if (cs > RB_NW) return m;
        m.cs[l] = cs;
        m.ksf[l] = RB_NW / cs;
        m.units[l] = m.ld[l] / 8;
        m.upw[l] = (m.units[l] + m.ksf[l] - 1) / m.ksf[l];
        if (m.upw[l] > RB_MAXU) return m;
        scratch = rb_max(scratch, m.ksf[l] * 4 * N);
    }
    for (int l = Lm - 2; l >= 1; l--) { // backward products from the LDS images
        const int G = (m.kr[l] + 63) / 64, k4n = m.kr[l + 1] / 4;
        int ks = rb_min(rb_max(1, (2 * RB_NW) / G), k4n); // about two tasks per wave
        if (ks > 8) ks = 8;
        m.gb[l] = G;
        m.ksb[l] = ks;
        scratch = rb_max(scratch, ks * 4 * G * 64);
    }
    m.lds_floats = off + scratch + 64;
    if (m.lds_floats * 4 > 160 * 1024 - 256) return m;
    m.ok = true;
    return m;
}

struct RbParams {
    RbPlan plan;                 // used by runtime-shape instantiations only
    const float *W[MAX_LAYERS];  // global W_l, l = 1..L-2
    float *act[MAX_LAYERS];      // act[1..L-2] out (A_l, read by the tile kernel)
    float *delta[MAX_LAYERS];    // delta[1..L-1] out
    const float *Y; int ldy;
    float *prob; float *loss; int32_t *label;
    int B;
    int inner_act, last_act;     // inner_act is read only by kernels built with ACT = -1
    const int32_t *row_idx;      // optional: expected row of batch row r is Y row row_idx[r] (sampled batches)
    const float *slabs; int slab_rows; // slab s of batch row b at slabs[(s * slab_rows + b) * ld[1]] (tile_step_kernel.h)
    unsigned long long *stamps;  // STAMP builds only: 16 slots per workgroup
};

template <int... DIMS> struct RbStaticShape {
    static constexpr bool is_static = true;
    static constexpr int kL = (int)sizeof...(DIMS);
    static constexpr int kDims[sizeof...(DIMS)] = {DIMS...};
    __host__ __device__ static constexpr RbPlan make() {
        constexpr int dims[sizeof...(DIMS)] = {DIMS...};
        return make_rb_plan(dims, (int)sizeof...(DIMS));
    }
};
template <int NL> struct RbRuntimeShape {
    static constexpr bool is_static = false;
    static constexpr int kL = NL;
};

#define GNN_RB_STAMP(i)                                                                                  \
    do {                                                                                                 \
        if (STAMP && threadIdx.x == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();   \
    } while (0)

// x + (x of the lane 32 away): both halves of the wave end up with the sum.
// (The two results are copied to scalars before they are reinterpreted: `__builtin_bit_cast(float, r[1])` on the builtin's
//  vector result read element 0 twice with hipcc 7.2 -- every sum came out as 2 x one half.)
__device__ __forceinline__ float rb_sum32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned lo = r[0], hi = r[1];
    return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}

// NL > 0: layer count fixed at compile time; IS_STATIC: `m` is a compile-time constant (every extent folds)
template <int NL, bool IS_STATIC, int ACT_T, int OUTK, bool STAMP, int NSV, int UPW1>
__device__ __forceinline__ void rowblock_body(const RbPlan &m, RbParams &p) {
    const int ACT = (ACT_T >= 0) ? ACT_T : p.inner_act;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = blockIdx.x * 4;
    const int L = (NL > 0) ? NL : m.L, Lm = L - 1;
    GNN_RB_STAMP(0);

    // ---- phase 0: every load of the kernel is issued here, in the order the data is needed -------------------------
    // (the row index of a sampled batch's expected row is a DEPENDENT load: issued first, so that waiting for it does not
    //  drain the queue of everything issued before it -- loads return in order)
    const int qy = m.ld[Lm] >> 2; // 4
    const int y_e = RB_NT - 1 - t, y_r = y_e / qy, y_q = y_e - y_r * qy; // the expected rows: the LAST threads
    const bool y_on = p.Y != nullptr && y_e < 4 * qy;
    // (unconditional, from an address that is always valid: a load under a branch is waited for at the end of its block)
    const bool y_ix = y_on && p.row_idx != nullptr && row0 + y_r < p.B;
    const int y_ld = *(y_ix ? p.row_idx + (row0 + y_r) : reinterpret_cast<const int32_t *>(p.slabs));
    __builtin_amdgcn_sched_barrier(0);
    // (a) the first-layer K slabs: one float4 of the four A_1 rows per thread, all slabs of it
    const int q1 = m.ld[1] >> 2;
    const bool a1_on = t < 4 * q1;
    const int a1_r = IS_STATIC ? t / q1 : (int)(((unsigned)t * (((1u << 22) + q1 - 1) / q1)) >> 22), a1_q = t - a1_r * q1;
    f32x4 zs[NSV];
    {
        const unsigned zoff = a1_on ? (unsigned)(row0 + a1_r) * (unsigned)m.ld[1] + (unsigned)(a1_q * 4) : 0u;
        const unsigned sstride = (unsigned)p.slab_rows * (unsigned)m.ld[1];
#pragma unroll
        for (int i = 0; i < NSV; i++) {
            zs[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (i < m.ns) zs[i] = m4_load16(p.slabs, zoff + (a1_on ? (unsigned)i * sstride : 0u));
        }
    }
    // Issue order = the order the data is needed in.  Within a wave loads return in order; ACROSS the waves the memory
    // pipeline serves requests roughly as they arrive, so every wave's slab requests are put in front of any wave's weight
    // requests: a bare barrier (nothing is waited for) between the two groups.  Without it A_1 was ready only when the
    // weights had landed too (5 500 cycles after the start instead of ~2 500: tools/rowblock_probe).
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // (b) this wave's K slice of the first register product (W_1 -> layer 2): 8-row units, 16 B per lane
    //     lane (hq, lq): rows 8u + 4hq + 0..3 of the unit, columns 128 cslice + 4 lq .. +3
    const int hq = lane >> 5, lq = lane & 31;
    const int rot = (blockIdx.x >> 3) & 3; // the four workgroups of an XCD start their weight streams at different slices
    f32x4 w1[UPW1 > 0 ? UPW1 : 1][4];
    int u0_1 = 0, nu_1 = 0, s_1 = 0, c_1 = 0;
    if (UPW1 > 0 && Lm >= 3) { // (a net of three layers has no register product: its only middle matrix is the row tail's)
        const int l = 1;
        const int CS = m.cs[l], KS = m.ksf[l], U = m.units[l];
        c_1 = wave & (CS - 1);
        s_1 = (wave / CS + rot * ((KS + 3) >> 2)) % KS;
        u0_1 = (s_1 * U) / KS;
        nu_1 = ((s_1 + 1) * U) / KS - u0_1;
        const int col = 128 * c_1 + 4 * lq;
        const bool lane_on = col < m.ld[l + 1];
#pragma unroll
        for (int uu = 0; uu < UPW1; uu++) {
            const bool on = lane_on && uu < nu_1;
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const int k = 8 * (u0_1 + uu) + 4 * hq + tt;
                w1[uu][tt] = m4_load16(p.W[l], on ? (unsigned)(k * m.ld[l + 1] + col) : 0u);
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // (c) the expected rows and the last weight image (the row tail reads it from LDS): small, needed last, issued last
    const int y_idx = p.row_idx ? (y_ix ? y_ld : 0) : row0 + y_r; // rows past the batch are masked below
    const f32x4 yv = *reinterpret_cast<const f32x4 *>((p.Y ? p.Y : p.slabs) + (y_on ? (size_t)y_idx * p.ldy + y_q * 4 : (size_t)0));
    const int c4l = m.kr[Lm] >> 2, nl4 = m.kr[Lm - 1] * c4l; // float4s of the last image's logical columns (<= 128 * 4)
    const int wl_r = IS_STATIC ? t / c4l : (int)(((unsigned)t * (((1u << 22) + c4l - 1) / c4l)) >> 22), wl_c = t - wl_r * c4l;
    const f32x4 wl = m4_load16(p.W[Lm - 1], t < nl4 ? (unsigned)(wl_r * m.ld[Lm] + 4 * wl_c) : 0u);
    // (no register pins here: an `asm volatile("" : "+v"(x))` READS x, i.e. waits for its load -- the compiler cannot move
    //  these loads below the global store of A_1 in phase 1, which may alias them)
    __builtin_amdgcn_sched_barrier(0);
    auto tail_operands_to_lds = [&]() {
        if (y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv;
        if (t < nl4) *reinterpret_cast<f32x4 *>(smem + m.off_w[Lm - 1] + wl_r * m.lw[Lm - 1] + 4 * wl_c) = wl;
    };

    // ---- phase 1: A_1 = f(sum of the slabs, slab order) -------------------------------------------------------------
    {
        f32x4 z = zs[0];
#pragma unroll
        for (int i = 1; i < NSV; i++)
            if (i < m.ns) z += zs[i];
        const bool lrow = row0 + a1_r < p.B;
        f32x4 a;
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = (lrow && a1_q * 4 + j < m.d[1]) ? act_fn(ACT, z[j]) : 0.f;
        if (a1_on) {
            *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a;
            *reinterpret_cast<f32x4 *>(p.act[1] + (size_t)(row0 + a1_r) * m.ld[1] + a1_q * 4) = a; // the tile kernel reads A_1
        }
        if (Lm < 3) tail_operands_to_lds(); // (no register product in front of the row tail)
    }
    __syncthreads();
    GNN_RB_STAMP(1);

    // ---- forward: layers 2 .. L-2 from registers (SCE:172-194) ---------------------------------------------------------
#pragma unroll
    for (int l = 1; l < MAX_LAYERS - 2; l++) {
        if (l + 1 >= Lm) break;
        const int N = m.ld[l + 1], CS = m.cs[l], KS = m.ksf[l], U = m.units[l];
        const int c = wave & (CS - 1), s = (wave / CS + rot * ((KS + 3) >> 2)) % KS;
        const int u0 = (s * U) / KS, nu = ((s + 1) * U) / KS - u0;
        const int col = 128 * c + 4 * lq;
        const bool lane_on = col < N;
        const float *arow = smem + m.off_act[l] + (lane & 3) * (m.ld[l] + 4) + 4 * hq;
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float *img = smem + m.off_w[l];
        auto unit = [&](int u, const f32x4 (&w)[4]) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 8 * u);
#pragma unroll
            for (int tt = 0; tt < 4; tt++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[tt], w[tt][j], acc[j], 0, 0, 0);
        };
        auto to_image = [&](int u, const f32x4 (&w)[4]) { // the rows this wave holds, for the backward product
            if (lane_on && col < m.kr[l + 1]) {
#pragma unroll
                for (int tt = 0; tt < 4; tt++) {
                    const int k = 8 * u + 4 * hq + tt;
                    if (k < m.kr[l]) *reinterpret_cast<f32x4 *>(img + k * m.lw[l] + col) = w[tt];
                }
            }
        };
        if (l == 1 && UPW1 > 0) {
            // a unit's rows go to the LDS image right behind its MFMAs: the writes (13 cycles of the LDS store path each) run
            // under the matrix pipe's 128 cycles per unit and under the wait for the next unit's loads
#pragma unroll
            for (int uu = 0; uu < UPW1; uu++)
                if (uu < nu) { unit(u0 + uu, w1[uu]); to_image(u0 + uu, w1[uu]); }
        } else {
            // later layers (nets of five and more layers): the slice is loaded here, RB_MAXU units at a time
            for (int ub = 0; ub < nu; ub += RB_MAXU) {
                f32x4 w[RB_MAXU][4];
#pragma unroll
                for (int uu = 0; uu < RB_MAXU; uu++) {
                    const bool on = lane_on && ub + uu < nu;
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
                        const int k = 8 * (u0 + ub + uu) + 4 * hq + tt;
                        w[uu][tt] = m4_load16(p.W[l], on ? (unsigned)(k * N + col) : 0u);
                    }
                }
#pragma unroll
                for (int uu = 0; uu < RB_MAXU; uu++)
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) asm volatile("" : "+v"(w[uu][tt]));
#pragma unroll
                for (int uu = 0; uu < RB_MAXU; uu++)
                    if (ub + uu < nu) { unit(u0 + ub + uu, w[uu]); to_image(u0 + ub + uu, w[uu]); }
            }
        }
        if (l == 1) tail_operands_to_lds(); // behind this wave's weight loads in the queue: waiting for them here costs nothing
        // the two half-waves sat at different k: add them, then the slice's partial tile, 16 B per row and lane
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc[j][r] = rb_sum32(acc[j][r]);
        if (hq == 0 && lane_on) {
            float *part = smem + m.off_scratch + (s * 4) * N + col;
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<f32x4 *>(part + r * N) = (f32x4){acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        }
        __syncthreads();
        GNN_RB_STAMP(2 * l);
        // K slices summed in slice order, f applied: 4 x N/4 float4s over the threads
        const int n4 = N >> 2;
        for (int e = t; e < 4 * n4; e += RB_NT) {
            const int er = IS_STATIC ? e / n4 : (int)(((unsigned)e * (((1u << 22) + n4 - 1) / n4)) >> 22), n = 4 * (e - er * n4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < m.kr[l + 1])
                for (int ks = 0; ks < KS; ks++) v += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (ks * 4 + er) * N + n);
            const bool lrow = row0 + er < p.B;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (lrow && n + j < m.d[l + 1]) ? act_fn(ACT, v[j]) : 0.f;
            *reinterpret_cast<f32x4 *>(p.act[l + 1] + (size_t)(row0 + er) * N + n) = v;
            *reinterpret_cast<f32x4 *>(smem + m.off_act[l + 1] + er * (N + 4) + n) = v;
        }
        __syncthreads();
        GNN_RB_STAMP(2 * l + 1);
    }

    // ---- row tail: one wave per batch row does the last layer, the output rule and delta_{L-2} ------------------------
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
            const float ssum = row16_sum(e);
            out = live ? e * (1.f / ssum) : 0.f;
            dd = live ? out - yy : 0.f;                          // SCE:250
            if (p.loss) lse = mx + __logf(ssum);
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
            dlast[c] = dd;
            p.delta[Lm][(size_t)row * 16 + c] = dd;
        }
        if (p.loss) {
            const float lsum = row16_sum(lterm);
            if (lane == 0) p.loss[row] = row < p.B ? lsum : 0.f;
        }
        if (p.label && lane == 0) p.label[row] = row < p.B ? best : -1;
        if (Lm - 1 >= 1) {
            // delta_{L-2}[n] = (sum_c delta_{L-1}[c] W[n][c]) f'(a[n]): lane n and n + 64; the wave reads back
            // its own 16 deltas (LDS keeps a wave's accesses in order); only the copied columns of W are used
            f32x4 d4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) d4[q] = *reinterpret_cast<const f32x4 *>(dlast + 4 * q);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int n = lane + 64 * half;
                if (n < ldp) {
                    float accd = 0.f;
                    if (n < K) {
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            if (4 * q < m.kr[Lm]) {
                                const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wl + n * lwl + 4 * q);
#pragma unroll
                                for (int j = 0; j < 4; j++) accd = __builtin_fmaf(d4[q][j], w4[j], accd);
                            }
                        }
                    }
                    const float v = (row < p.B && n < m.d[Lm - 1]) ? accd * act_prime_from_a(ACT, a[n]) : 0.f;
                    if (Lm - 1 > 1) smem[m.off_dl[Lm - 1] + r * (ldp + 4) + n] = v;
                    p.delta[Lm - 1][(size_t)row * ldp + n] = v;
                }
            }
        }
    }
    __syncthreads();
    GNN_RB_STAMP(12);

    // ---- backward data: delta_l = (delta_{l+1} . W_l^T) * f'(z_l), l = L-3 .. 1 (SCE:262-278), from the LDS images ------
#pragma unroll
    for (int li = 0; li < MAX_LAYERS; li++) {
        const int l = Lm - 2 - li;
        if (l < 1) break;
        const int N = m.ld[l], NR = m.kr[l], G = m.gb[l], gw = G * 64, KS = m.ksb[l];
        const int k4n = m.kr[l + 1] / 4;
        for (int task = wave; task < G * KS; task += RB_NW) { // (wave-uniform)
            const int g = task % G, ks = task / G;
            rowblock_product<true>(smem + m.off_dl[l + 1], m.ld[l + 1] + 4, smem + m.off_w[l], m.lw[l],
                                   ks * k4n / KS, (ks + 1) * k4n / KS, g * 64, NR,
                                   smem + m.off_scratch + ks * 4 * gw, gw, lane);
        }
        __syncthreads();
        const int n4 = N >> 2;
        for (int e = t; e < 4 * n4; e += RB_NT) {
            const int er = IS_STATIC ? e / n4 : (int)(((unsigned)e * (((1u << 22) + n4 - 1) / n4)) >> 22), n = 4 * (e - er * n4);
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (n < NR)
                for (int ks = 0; ks < KS; ks++) v += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (ks * 4 + er) * gw + n);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(smem + m.off_act[l] + er * (N + 4) + n); // f'(z_l) from a_l = f(z_l)
            const bool lrow = row0 + er < p.B;
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (lrow && n + j < m.d[l]) ? v[j] * act_prime_from_a(ACT, a[j]) : 0.f;
            *reinterpret_cast<f32x4 *>(p.delta[l] + (size_t)(row0 + er) * N + n) = v;
            if (l > 1) *reinterpret_cast<f32x4 *>(smem + m.off_dl[l] + er * (N + 4) + n) = v;
        }
        if (l > 1) __syncthreads();
        GNN_RB_STAMP(13);
    }
    GNN_RB_STAMP(14);
}

template <class SH, int ACT, int OUTK, bool STAMP = false>
__global__ __launch_bounds__(RB_NT) void rowblock_kernel(RbParams p) {
    if constexpr (SH::is_static) {
        constexpr RbPlan m = SH::make(); // a LOCAL constexpr object: member accesses with constant indices fold to immediates
        static_assert(m.ok, "this shape does not fit the row-block kernel");
        rowblock_body<SH::kL, true, ACT, OUTK, STAMP, m.ns, (SH::kL >= 4 ? m.upw[1] : 0)>(m, p);
    } else {
        rowblock_body<SH::kL, false, ACT, OUTK, STAMP, MID4_MAX_SLABS, (SH::kL == 3 ? 0 : RB_MAXU)>(p.plan, p);
    }
}

} // namespace gnn
