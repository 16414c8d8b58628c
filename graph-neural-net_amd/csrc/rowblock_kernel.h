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
    int off_fp1;             // bf16 kernels: f'(a_1) from the unrounded activation, [4][ld[1]+4] (middle4_kernel.h keeps the same image)
    // forward register products giving layer l+1 from image-less W_l, l = 1..L-3 (the last product is the row tail's)
    int cs[MAX_LAYERS];      // column slices of 128 (waves = ksf * cs = 8)
    int ksf[MAX_LAYERS];     // K slices = partial tiles per output element
    int units[MAX_LAYERS];   // 8-row units along K = ld[l] / 8
    int upw[MAX_LAYERS];     // most units one wave takes
    // backward LDS products giving delta_l, l = L-3..1: column groups of 64, dealt to the waves round robin (ksb = 1: no K split)
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
    m.off_fp1 = off; off += 4 * (m.ld[1] + 4);
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
    for (int l = Lm - 2; l >= 1; l--) { // backward products from the LDS images: one wave per group of 64 neurons, whole K
        m.gb[l] = (m.kr[l] + 63) / 64;
        m.ksb[l] = 1;
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
    // A SAMPLED batch (row_idx != null): the four input rows of this row block, gathered from the data set, are also written
    // to a contiguous copy [padded batch rows][ldx] by waves 4..7 while the row tail runs -- the tile kernel that follows reads
    // its gradient operand A_0 from the copy with plain addressing (the same copy written by the PREVIOUS tile kernel cost it
    // 0.8 us: tools/tile_probe).  Null = no copy.  (Beside row_idx and slabs: the kernel's top reads these fields, and a field
    // in a kernel-argument cache line of its own is one more scalar-cache miss in front of the first load.)
    // Which rows: copy_idx[r] for r < B (zeros behind them) -- this batch's own rows (copy_idx == row_idx), or the NEXT batch's
    // when the caller has announced it: the tile kernel that follows then reads the next batch's rows for its first-layer
    // product with plain addressing too, and the copy is this batch's gradient operand one step later (plan.hip, chain_gradient).
    float *xcopy; __bf16 *xcopyb; const float *X; const __bf16 *Xb; int ldx;
    const int32_t *copy_idx;
    unsigned long long *stamps;  // STAMP builds only: 16 slots per workgroup
    // bf16 kernels (BF): the bf16 shadow of W_l and the bf16 outputs the tile kernel reads, as in Mid4Params
    const __bf16 *Wb[MAX_LAYERS];
    __bf16 *actb[MAX_LAYERS];
    __bf16 *deltab[MAX_LAYERS];
};

// four bf16 weights (8 B of the shadow) widened to the f32 values they stand for
__device__ __forceinline__ f32x4 rb_widen4(uint2 u) {
    f32x4 w;
    w[0] = __builtin_bit_cast(float, u.x << 16); w[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    w[2] = __builtin_bit_cast(float, u.y << 16); w[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
    return w;
}
// the weights a thread's float4 slot holds: 16 B of the f32 masters, or (BF) 8 B of the bf16 shadow at the same element offset
template <bool BF> __device__ __forceinline__ f32x4 rb_load_w4(const float *W, const __bf16 *Wb, unsigned off) {
    if constexpr (BF) return rb_widen4(*reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(Wb) + (size_t)(off * 2u)));
    else return m4_load16(W, off);
}

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

// per-wave stamps (STAMP builds): region r, 16 slots per workgroup behind the first 16 * gridDim.x
#define GNN_RB_WSTAMP(r)                                                                                                         \
    do {                                                                                                                         \
        if (STAMP && lane == 0) p.stamps[(size_t)(16 * gridDim.x) * (1 + (r)) + blockIdx.x * 16 + wave] = __builtin_amdgcn_s_memtime(); \
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

// x + (x of the lane 16 away, lane ^ 16): rows 1 and 3 of one copy change places with rows 0 and 2 of the other
__device__ __forceinline__ float rb_sum16(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned lo = r[0], hi = r[1];
    return __builtin_bit_cast(float, lo) + __builtin_bit_cast(float, hi);
}

// NL > 0: layer count fixed at compile time; IS_STATIC: `m` is a compile-time constant (every extent folds)
template <int NL, bool IS_STATIC, int ACT_T, int OUTK, bool STAMP, int NSV, int UPW1, int TUNE, bool BF>
__device__ __forceinline__ void rowblock_body(const RbPlan &m, RbParams &p) {
    const int ACT = (ACT_T >= 0) ? ACT_T : p.inner_act;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int row0 = blockIdx.x * 4;
    int x_ld_top = 0; // waves 4..7: the data-set row this wave copies during the row tail (below), a scalar load
    if (wave >= 4 && p.copy_idx && row0 + (wave - 4) < p.B)
        x_ld_top = *reinterpret_cast<const __attribute__((address_space(4))) int32_t *>(reinterpret_cast<unsigned long long>(p.copy_idx + (row0 + wave - 4)));
    const int L = (NL > 0) ? NL : m.L, Lm = L - 1;
    // BF (GNN_DTYPE_BF16, nets of three and four layers): every operand of the products carries a bf16 value -- weights from
    // the bf16 shadow (half the bytes of the kernel's dominant load), activations and deltas rounded as they enter their
    // operand images, f' kept from the UNROUNDED activation -- on the same exact-f32 MFMA, outputs for the tile kernel in
    // bf16: middle4_kernel<.., BF16>'s arithmetic in this kernel's schedule.
    auto opv = [](float x) { return BF ? bf16_value(x) : x; };
    GNN_RB_STAMP(0);

    // ---- phase 0 / 1: loads, and A_1 = f(sum of the slabs, slab order) ------------------------------------------------
    // What bounds this kernel is the CU's vector-memory pipe: it takes a wave-wide 16-B load in 16 cycles WHATEVER its
    // lanes carry, and a wave whose requests are queued behind others' STALLS AT ISSUE -- with all 36 loads of a thread
    // issued up front, issuing alone took 4 000 cycles and the slab sums, long landed, waited behind it
    // (tools/rowblock_probe stamps).  So the pipe must be kept busy from the first cycle with the fewest instructions:
    //   * the weights are requested RB_PF units ahead of the MFMAs that consume them, PF0 units before A_1 is formed;
    //   * only the waves that own an A_1 element issue slab loads (five of eight for 784-300-100-10), in ONE block with the
    //     sum that consumes them;
    //   * every load OUTSIDE that block is unconditional (clamped address, never `cond ? offset : 0` on a wave-uniform
    //     condition): after a branch that contains a load the compiler's wait-count pass drains the whole queue
    //     (s_waitcnt vmcnt(0)) at the next use of anything loaded.
    // (Re-measured after the head arguments moved into SGPRs -- the first loads now issue ~500 cycles earlier and the best order
    //  changed with it: the slabs first and ONE weight unit ahead, 6.22 us per launch; two units ahead 6.30; the weights in front of the
    //  slabs, the order of the builds before, 6.30 / 6.50 with one / two units; three units 7.44.  tools/rowblock_probe 128 1.)
    constexpr int RB_PF = (TUNE & 0x600) ? ((TUNE >> 9) & 3) : 2; // weight units requested ahead of the one being multiplied
    constexpr int PF0 = (TUNE & 7) ? (TUNE & 7) : 1; // units requested before A_1 is formed
    constexpr bool W_FIRST = (TUNE & 8) != 0;      // (probe) the first weight units in FRONT of the slabs in every wave's queue
    constexpr int PF1 = (TUNE & 0x800) ? PF0 + RB_PF : PF0; // (probe) units requested by the time of the A_1 barrier: the product's first RB_PF units right in front of it
    // (the row index of a sampled batch's expected row is a DEPENDENT load: issued first)
    const int qy = m.ld[Lm] >> 2; // 4
    const int y_e = RB_NT - 1 - t, y_r = y_e / qy, y_q = y_e - y_r * qy; // the expected rows: the LAST threads
    const bool y_on = p.Y != nullptr && y_e < 4 * qy;
    const bool y_ix = y_on && p.row_idx != nullptr && row0 + y_r < p.B;
    const int y_ld = *(y_ix ? p.row_idx + (row0 + y_r) : reinterpret_cast<const int32_t *>(p.slabs));
    // the last weight image (the row tail reads it from LDS): one small load
    const int c4l = m.kr[Lm] >> 2, nl4 = m.kr[Lm - 1] * c4l; // float4s of the last image's logical columns (<= 128 * 4)
    const int wl_r = IS_STATIC ? t / c4l : (int)(((unsigned)t * (((1u << 22) + c4l - 1) / c4l)) >> 22), wl_c = t - wl_r * c4l;
    const f32x4 wl = rb_load_w4<BF>(p.W[Lm - 1], p.Wb[Lm - 1], t < nl4 ? (unsigned)(wl_r * m.ld[Lm] + 4 * wl_c) : 0u);
    __builtin_amdgcn_sched_barrier(0);
    // this wave's K slice of the first register product (W_1 -> layer 2): 8-row units, 16 B per lane
    //     lane (hq, lq): rows 8u + 4hq + 0..3 of the unit, columns 128 cslice + 4 lq .. +3
    const int hq = lane >> 5, lq = lane & 31;
    const int rot = (blockIdx.x >> 3) & 3; // the four workgroups of an XCD start their weight streams at different slices
    f32x4 w1[UPW1 > 0 ? UPW1 : 1][4];
    int u0_1 = 0, nu_1 = 0;
    // (uu: compile-time after unrolling.  No select on the address: a unit past the wave's slice is CLAMPED to the last
    //  unit of the matrix and a lane past the last column to the last float4 of the row -- valid addresses whose data is
    //  never used.)
    int ulast_1 = 0, colc_1 = 0;
    auto load_unit_1 = [&](int uu) {
        const int u = u0_1 + uu < ulast_1 ? u0_1 + uu : ulast_1;
#pragma unroll
        for (int tt = 0; tt < 4; tt++) {
            const int k = 8 * u + 4 * hq + tt;
            if (TUNE & (1 << 20)) w1[UPW1 > 0 ? uu : 0][tt] = (f32x4){0.01f * (float)k, 0.f, 0.02f, 0.f}; // (probe: what the weight stream costs -- no loads, wrong results)
            else w1[UPW1 > 0 ? uu : 0][tt] = rb_load_w4<BF>(p.W[1], p.Wb[1], (unsigned)(k * m.ld[2] + colc_1));
        }
    };
    // The copy of a wave's weight rows to the LDS image (for the backward product) costs the LDS store path 13 cycles per
    // 16-B write, 160 writes in all.  The waves whose threads sum the K slices afterwards (the first ceil(4 N / 4 / 64) of
    // them) do it between their MFMAs; the others wait until THEY are idle -- the slice sum -- so that neither the
    // product nor the row tail (whose LDS reads queued behind these writes when they ran beside it) pays for it.
    constexpr bool DEFER = (TUNE & 16) == 0;
    constexpr int n_sum_waves = 4; // the row-tail waves copy between their MFMAs, waves 4..7 while the row tail runs
    int col_1 = 0;
    auto to_image_1 = [&](int uu) { // the rows this wave holds of unit uu, for the backward product
        if (col_1 < m.kr[2]) {
#pragma unroll
            for (int tt = 0; tt < 4; tt++) {
                const int k = 8 * (u0_1 + uu) + 4 * hq + tt;
                if (k < m.kr[1]) *reinterpret_cast<f32x4 *>(smem + m.off_w[1] + k * m.lw[1] + col_1) = w1[UPW1 > 0 ? uu : 0][tt];
            }
        }
    };
    if (UPW1 > 0 && Lm >= 3) { // (a net of three layers has no register product: its only middle matrix is the row tail's)
        const int CS = m.cs[1], KS = m.ksf[1], U = m.units[1];
        const int c_1 = wave & (CS - 1), s_1 = (wave / CS + rot * ((KS + 3) >> 2)) % KS;
        u0_1 = (s_1 * U) / KS;
        nu_1 = ((s_1 + 1) * U) / KS - u0_1;
        col_1 = 128 * c_1 + 4 * lq;
        colc_1 = col_1 < m.ld[2] ? col_1 : m.ld[2] - 4;
        ulast_1 = U - 1;
    }
    if constexpr (UPW1 > 0 && W_FIRST) {
#pragma unroll
        for (int uu = 0; uu < UPW1; uu++)
            if (uu < PF0) load_unit_1(uu);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the first-layer K slabs: one float4 of the four A_1 rows per thread, all slabs of it; slab order; then f.
    // Rows past the batch and columns past d_1 are zeros (f(0) != 0 for the sigmoid).
    const int q1 = m.ld[1] >> 2;
    f32x4 a1v = {0.f, 0.f, 0.f, 0.f}; // this thread's float4 of A_1 and where the tile kernel reads it (stored below, see there)
    unsigned a1_goff = 0xffffffffu;
    if (wave * 64 < 4 * q1) { // (wave-uniform)
        if (TUNE & 64) { if (wave >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); } // the slab waves over the others; the younger one of a SIMD first
        const bool a1_on = t < 4 * q1;
        const int a1_r = IS_STATIC ? t / q1 : (int)(((unsigned)t * (((1u << 22) + q1 - 1) / q1)) >> 22), a1_q = t - a1_r * q1;
        const unsigned zoff = a1_on ? (unsigned)(row0 + a1_r) * (unsigned)m.ld[1] + (unsigned)(a1_q * 4) : 0u;
        const unsigned sstride = (unsigned)p.slab_rows * (unsigned)m.ld[1];
        f32x4 zs[NSV];
#pragma unroll
        for (int i = 0; i < NSV; i++) {
            const unsigned o = zoff + (a1_on ? (unsigned)i * sstride : 0u);
            if ((TUNE & (1 << 21)) && i > 0) zs[i] = (f32x4){0.001f * (float)i, 0.f, 0.f, 0.f}; // (probe: what twelve of the thirteen slab loads cost)
            else zs[i] = m4_load16(p.slabs, i < m.ns ? o : 0u); // (runtime shapes: the slots past n_slabs re-read offset 0 and are not summed)
        }
        GNN_RB_STAMP(6); // this wave's phase-0 loads issued
        f32x4 z = zs[0];
#pragma unroll
        for (int i = 1; i < NSV; i++)
            if (i < m.ns) z += zs[i];
        if (STAMP) { asm volatile("" : "+v"(z)); GNN_RB_STAMP(7); } // this wave's slabs have landed
        const bool lrow = row0 + a1_r < p.B;
        f32x4 a;
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = (lrow && a1_q * 4 + j < m.d[1]) ? act_fn(ACT, z[j]) : 0.f;
        if (BF) { // f' from the unrounded activation, then the operand value
            f32x4 fp;
#pragma unroll
            for (int j = 0; j < 4; j++) { fp[j] = act_prime_from_a(ACT, a[j]); a[j] = bf16_value(a[j]); }
            if (a1_on) *reinterpret_cast<f32x4 *>(smem + m.off_fp1 + a1_r * (m.ld[1] + 4) + a1_q * 4) = fp;
        }
        if (a1_on) {
            *reinterpret_cast<f32x4 *>(smem + m.off_act[1] + a1_r * (m.ld[1] + 4) + a1_q * 4) = a;
            a1v = a;
            a1_goff = (unsigned)(row0 + a1_r) * (unsigned)m.ld[1] + (unsigned)(a1_q * 4);
        }
        if (TUNE & 64) __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (UPW1 > 0 && !W_FIRST) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < UPW1; uu++)
            if (uu < PF0) load_unit_1(uu);
    }
    __builtin_amdgcn_sched_barrier(0);
    // the row tail's weight image (requested first) and the expected rows (the index of a sampled batch's row has long landed)
    if (t < nl4) *reinterpret_cast<f32x4 *>(smem + m.off_w[Lm - 1] + wl_r * m.lw[Lm - 1] + 4 * wl_c) = wl;
    const int y_idx = p.row_idx ? (y_ix ? y_ld : 0) : row0 + y_r; // rows past the batch are masked below
    const f32x4 yv = *reinterpret_cast<const f32x4 *>((p.Y ? p.Y : p.slabs) + (y_on ? (size_t)y_idx * p.ldy + y_q * 4 : (size_t)0));
    if (Lm < 3 && y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv; // (no register product in front of the row tail)
    // the weight units requested in phase 0 have landed in front of the slabs (or this wave has no slabs and nothing else to
    // do until A_1 exists): their rows go to the LDS image NOW, in time the wave would spend waiting at the barrier
    constexpr bool EARLY_IMG = (TUNE & 128) == 0;
    // The waves that sum slabs do not wait for their first weight unit in front of the barrier: no early copy for them (the
    // unit's rows go to the image inside the product like the others'), and A_1's store moves behind the product so that the
    // unit's wait does not stand behind the store's round trip (6.26 -> 6.19 us; probe bit 0x1000 = every wave copies early)
    constexpr bool SLAB_NOWAIT = (TUNE & 0x1000) == 0;
    const bool early_here = !SLAB_NOWAIT || wave * 64 >= 4 * q1; // (wave-uniform)
    if (EARLY_IMG && early_here && UPW1 > 0 && Lm >= 3) {
#pragma unroll
        for (int uu = 0; uu < UPW1; uu++)
            if (uu < PF0 && uu < nu_1) to_image_1(uu);
    }
    if constexpr (UPW1 > 0 && PF1 > PF0) {
#pragma unroll
        for (int uu = PF0; uu < UPW1; uu++)
            if (uu < PF1) load_unit_1(uu);
        __builtin_amdgcn_sched_barrier(0);
    }
    // A_1 for the tile kernel: stored LAST.  The memory counter counts stores too, and the copies above wait for "all but
    // the youngest request" (what a wave without slabs needs): with the store in front of them the slab waves sat out its
    // round trip to L2 in front of the barrier.
    auto store_a1 = [&]() {
        if (a1_goff != 0xffffffffu) {
            if (BF) *reinterpret_cast<m4_bf16x4 *>(reinterpret_cast<char *>(p.actb[1]) + (size_t)(a1_goff * 2u)) = (m4_bf16x4){(__bf16)a1v[0], (__bf16)a1v[1], (__bf16)a1v[2], (__bf16)a1v[3]};
            else *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(p.act[1]) + (size_t)(a1_goff * 4u)) = a1v;
        }
    };
    if (!SLAB_NOWAIT || !(UPW1 > 0 && Lm >= 3)) store_a1();
    GNN_RB_WSTAMP(0); // this wave at the A_1 barrier
    __syncthreads();
    GNN_RB_STAMP(1);

    // The row tail's operands that do not depend on the activations -- the last weight image, once in the logits' layout
    // (lane (kg, q): rows kg + 16 i, columns 4q..) and once by row (lanes n and n + 64) for delta_{L-2} -- are read by the tail
    // waves at the START of the register product in front of the tail (right in front of the tail's barrier they delayed it:
    // a barrier waits for the wave's LDS reads): behind the barrier each set was an LDS round trip in the middle of a lone
    // wave's chain (the image has been complete since the A_1 barrier).
    constexpr int TK = 8; // K <= 128 = 16 k groups x 8
    f32x4 tw4[TK], twn[2][4];
    auto load_tail_weights = [&]() {
        const float *Wl = smem + m.off_w[Lm - 1];
        const int K = m.kr[Lm - 1], lwl = m.lw[Lm - 1];
        const int kg = lane >> 2, q = lane & 3;
        const bool q_on = 4 * q < m.kr[Lm];
#pragma unroll
        for (int i = 0; i < TK; i++) {
            const int k = kg + 16 * i;
            tw4[i] = *reinterpret_cast<const f32x4 *>(Wl + ((k < K && q_on) ? k * lwl + 4 * q : 0));
        }
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int n = lane + 64 * half;
#pragma unroll
            for (int qq = 0; qq < 4; qq++) twn[half][qq] = *reinterpret_cast<const f32x4 *>(Wl + ((n < K && 4 * qq < m.kr[Lm]) ? n * lwl + 4 * qq : 0));
        }
    };

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
            // the A operands of every unit (4 rows x 8 k each) are read at once, behind the barrier: one LDS round trip
            f32x4 av1[UPW1 > 0 ? UPW1 : 1];
#pragma unroll
            for (int uu = 0; uu < UPW1; uu++) av1[uu] = *reinterpret_cast<const f32x4 *>(arow + 8 * (u0 + (uu < nu ? uu : 0)));
            if (STAMP) { asm volatile("" : "+v"(av1[UPW1 > 0 ? UPW1 - 1 : 0])); GNN_RB_WSTAMP(5); } // A operands here
            if (l + 1 == Lm - 1 && wave < 4) load_tail_weights(); // (wave-uniform; see there: LDS has nothing else to do here)
            // software pipeline over the wave's units: request unit uu + RB_PF, multiply unit uu, copy its rows to the LDS
            // image for the backward product (13 cycles of the LDS store path per write, under the matrix pipe's 128 per unit)
#pragma unroll
            for (int uu = 0; uu < UPW1; uu++) {
                if (uu + RB_PF >= PF1 && uu + RB_PF < UPW1) load_unit_1(uu + RB_PF);
                if (uu == 0) { // (PF1 < RB_PF: catch up)
#pragma unroll
                    for (int v = PF1; v < RB_PF && v < UPW1; v++) load_unit_1(v);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (uu < nu) {
#pragma unroll
                    for (int tt = 0; tt < 4; tt++)
#pragma unroll
                        for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(av1[uu][tt], w1[uu][tt][j], acc[j], 0, 0, 0);
                    if (!(EARLY_IMG && early_here && uu < PF0) && (!DEFER || wave < n_sum_waves)) to_image_1(uu);
                }
                if (STAMP && uu == 1) { asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])); GNN_RB_WSTAMP(6); } // units 0, 1 multiplied
                __builtin_amdgcn_sched_barrier(0);
            }
            if (SLAB_NOWAIT) store_a1();
        } else {
            if (l + 1 == Lm - 1 && wave < 4) load_tail_weights(); // (as in the branch above)
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
                    if (ub + uu < nu) { unit(u0 + ub + uu, w[uu]); to_image(u0 + ub + uu, w[uu]); }
            }
        }
        if (STAMP) { asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])); GNN_RB_STAMP(8); if (l == 1) GNN_RB_WSTAMP(7); } // MFMAs + image writes issued, results in
        if (l == 1 && y_on) *reinterpret_cast<f32x4 *>(smem + m.off_y + y_r * m.ld[Lm] + y_q * 4) = yv;
        // the two half-waves sat at different k: add them -- one row swap serves TWO values: rows r and r + 2 change halves
        // (v_permlane32_swap: the upper half of one register against the lower half of the other), after which the lower
        // half-wave holds the sums (lower + upper, as before) of rows 0 and 1 and the upper one those of rows 2 and 3 -- and each
        // half writes its two rows of the slice's partial tile, 16 B per row and lane.  (One swap per value and four rows
        // from the lower half alone took 1 000-1 300 cycles between the last MFMA and the barrier: tools/rowblock_probe.)
        // (Inline asm: with two DIFFERENT inputs hipcc 7.2 lowered __builtin_amdgcn_permlane32_swap to a swap of the first input
        //  with a copy of itself -- half the swaps gone, results wrong.  The asm is opaque to the hazard recogniser: the wait
        //  states between the last MFMA and the first read of its result are supplied here.)
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        f32x4 half_sum[2];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float x = acc[j][r], y = acc[j][r + 2];
                // x = (x.lower, y.lower), y = (x.upper, y.upper).  (s_nop 1: the two wait states between a vector instruction that
                //  writes a register -- the compiler may copy x or y right in front of this statement -- and a swap that reads it.)
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));
                half_sum[r][j] = x + y;
            }
        if (l == 1) GNN_RB_WSTAMP(1); // this wave's MFMAs done
        if (lane_on) {
            float *part = smem + m.off_scratch + (s * 4 + 2 * hq) * N + col;
#pragma unroll
            for (int r = 0; r < 2; r++) *reinterpret_cast<f32x4 *>(part + r * N) = half_sum[r];
        }
        if (l == 1) GNN_RB_WSTAMP(2); // this wave at the partial-tile barrier
        __syncthreads();
        GNN_RB_STAMP(2 * l);
        if (l + 1 == Lm - 1) break; // the layer the row tail reads: its wave sums its own row's slices (no barrier: same wave)
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
    // A lone wave hides no latency, so the chain is kept short: lane (kg = lane / 4, q = lane % 4) multiplies rows
    // k = kg, kg + 16, .. of the last weight image with the float4 of columns 4q..4q+3 -- every LDS read of the logits is in
    // flight at once (<= 8 + 8 reads instead of 25 dependent rounds of two), the 16 k groups meet by two DPP rotations and
    // two row swaps, and the output rule runs on the quad (q = 0..3 holds the 16 padded classes).
    // Every load of this wave was issued thousands of cycles ago, but the clamped weight units a wave does not multiply are
    // never waited for, so the compiler still counts them as pending: it then guards the first reuse of their registers with
    // s_waitcnt vmcnt(0..3) -- AFTER the stores below, which the same counter counts, so the wave sat out a store's round
    // trip to L2 three times (logits, delta_{L-2}, the backward product's start).  Waiting here costs nothing and clears it.
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0) only (expcnt, lgkmcnt: no wait)
    // Waves 4..7 copy one input row each of a sampled batch while the row tail runs (below).  Whether a copy is asked for, and
    // the row's index, are read HERE and by these waves only: the copy's fields sit in a kernel-argument line nothing else has
    // touched, and the scalar-cache miss that the test costs stood in front of the first vector load when it was made at the
    // kernel's top (0.3 us whether a copy was asked for or not) and in front of the A_1 barrier when it was made there (0.15).
    // The index is a SCALAR load (one address per wave; the index vector is not written while this kernel runs).
    // (The row's index: requested at the kernel's top, x_ld_top -- copy_idx and B arrive in SGPRs with the dispatch, so that
    //  test costs no miss; the NEXT batch's indices and rows are cold in L2, and index -> row behind each other did not fit under the row tail.)
    const int x_ld = x_ld_top;
    bool x_copy = false;
    if (wave >= 4) x_copy = !(TUNE & (1 << 22)) && (p.xcopy || p.xcopyb) && p.copy_idx; // (wave-uniform)
    if (wave < 4) {
        if (TUNE & 32) __builtin_amdgcn_s_setprio(3); // the four waves on the critical path, over the image copies of the other four
        const int r = wave, row = row0 + r;
        const int K = m.kr[Lm - 1], nt = m.d[Lm], ldp = m.ld[Lm - 1];
        // (the expected row: requested first, used ~2 000 cycles from here)
        const f32x4 y4 = (row < p.B && p.Y) ? *reinterpret_cast<const f32x4 *>(smem + m.off_y + r * 16 + 4 * (lane & 3)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        if (Lm >= 3) {
            // this row of the last hidden layer: K slices summed in slice order, f applied (N <= 128: one float4 per lane)
            const int l = Lm - 2, N = ldp, KS = m.ksf[l], n = 4 * lane;
            if (n < N) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (n < m.kr[l + 1])
                    for (int ks = 0; ks < KS; ks++) v += *reinterpret_cast<const f32x4 *>(smem + m.off_scratch + (ks * 4 + r) * N + n);
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = (row < p.B && n + j < m.d[l + 1]) ? act_fn(ACT, v[j]) : 0.f;
                if (BF) { // f'(a_l) parks in delta_l's image until delta_l itself is formed below
                    f32x4 fp;
#pragma unroll
                    for (int j = 0; j < 4; j++) { fp[j] = act_prime_from_a(ACT, v[j]); v[j] = bf16_value(v[j]); }
                    *reinterpret_cast<f32x4 *>(smem + m.off_dl[l + 1] + r * (N + 4) + n) = fp;
                    *reinterpret_cast<m4_bf16x4 *>(p.actb[l + 1] + (size_t)row * N + n) = (m4_bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                } else {
                    *reinterpret_cast<f32x4 *>(p.act[l + 1] + (size_t)row * N + n) = v;
                }
                *reinterpret_cast<f32x4 *>(smem + m.off_act[l + 1] + r * (N + 4) + n) = v; // (read back below by this very wave)
                if (STAMP) asm volatile("" : "+v"(v));
            }
            GNN_RB_STAMP(3); // this row's slices summed
        }
        const float *a = smem + m.off_act[Lm - 1] + r * (ldp + 4);
        const int kg = lane >> 2, q = lane & 3;
        const bool q_on = 4 * q < m.kr[Lm];        // (the image holds kr[Lm] <= 16 columns)
        if (Lm < 3) load_tail_weights();           // (no register product, no barrier in front of the tail: read here)
        float av[TK];
#pragma unroll
        for (int i = 0; i < TK; i++) {
            const int k = kg + 16 * i;
            av[i] = (k < K && q_on) ? a[k] : 0.f;
        }
        // (for delta_{L-2}, far below: this lane's two activations -- or, BF, the parked f' -- read now, in the same LDS round trip)
        float an[2];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int n = lane + 64 * half;
            an[half] = BF ? smem[((Lm - 1 > 1) ? m.off_dl[Lm - 1] : m.off_fp1) + r * (ldp + 4) + (n < ldp ? n : 0)] : a[n < ldp ? n : 0];
        }
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TK; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) z4[j] = __builtin_fmaf(av[i], tw4[i][j], z4[j]);
        if (STAMP) { asm volatile("" : "+v"(z4)); GNN_RB_STAMP(9); } // logits: reads + FMAs
#pragma unroll
        for (int j = 0; j < 4; j++) { // the 16 k groups: lanes q, q+4, q+8, q+12 of every row of 16, then the four rows
            float v = z4[j];
            v += dpp_f<0x128>(v);  // row_ror:8
            v += dpp_f<0x124>(v);  // row_ror:4
            v = rb_sum16(v);       // lane ^ 16, lane ^ 32: register swaps (a ds_bpermute pair per value was two more
            v = rb_sum32(v);       // LDS round trips on this chain)
            z4[j] = v;
        }
        if (STAMP) { asm volatile("" : "+v"(z4)); GNN_RB_STAMP(10); } // k groups reduced
        // output rule on the quad: lane q holds classes 4q..4q+3
        constexpr int QX1 = 0xB1, QX2 = 0x4E; // quad_perm [1,0,3,2] / [2,3,0,1]
        f32x4 out4 = {0.f, 0.f, 0.f, 0.f}, dd4 = {0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f, mx = -__builtin_inff(), nan_flag = 0.f;
        int best = -1;
        bool valid[4], live[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { valid[j] = 4 * q + j < nt; live[j] = valid[j] && row < p.B; }
        auto quad_argmax = [&](float &v, int &ix) { // larger value wins, ties -> higher index (MT:166-168)
#define GNN_RB_QSTEP(CTRL)                                              \
            {                                                            \
                const float ov = dpp_f<CTRL>(v);                         \
                const int oi = dpp_i<CTRL>(ix);                          \
                const bool tk = (ov > v) | ((ov == v) & (oi > ix));         \
                v = tk ? ov : v; ix = tk ? oi : ix;                      \
            }
            GNN_RB_QSTEP(QX1)
            GNN_RB_QSTEP(QX2)
#undef GNN_RB_QSTEP
        };
        // (selects, not branches: every `if` on a lane value here became a save-exec region with its own waits)
        if (OUTK == 0) {
            if (p.label) { // (kernel argument: a scalar branch)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float zj = z4[j];
                    nan_flag = (valid[j] & (zj != zj)) ? 1.f : nan_flag;      // any NaN logit -> label 0 (see output_layer_kernel)
                    const bool take = valid[j] & (zj >= mx);                  // `>=`: ties -> the higher index
                    mx = take ? zj : mx;
                    best = take ? 4 * q + j : best;
                }
                quad_argmax(mx, best);
            } else { // (no label asked for, every training step: the maximum as a tree -- NaNs drop out of fmaxf as they do out of `>=`)
                const float m01 = fmaxf(valid[0] ? z4[0] : -__builtin_inff(), valid[1] ? z4[1] : -__builtin_inff());
                const float m23 = fmaxf(valid[2] ? z4[2] : -__builtin_inff(), valid[3] ? z4[3] : -__builtin_inff());
                mx = fmaxf(m01, m23);
                mx = fmaxf(mx, dpp_f<QX1>(mx)); mx = fmaxf(mx, dpp_f<QX2>(mx));
            }
            f32x4 e4;
            float ssum = 0.f;
#pragma unroll
            for (int j = 0; j < 4; j++) { e4[j] = valid[j] ? __expf(z4[j] - mx) : 0.f; ssum += e4[j]; }
            ssum += dpp_f<QX1>(ssum);
            ssum += dpp_f<QX2>(ssum);
            const float inv = 1.f / ssum;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                out4[j] = live[j] ? e4[j] * inv : 0.f;
                dd4[j] = live[j] ? out4[j] - y4[j] : 0.f;                              // SCE:250
            }
            if (p.loss) { // (kernel argument: a scalar branch)
                const float lse = mx + __logf(ssum);
#pragma unroll
                for (int j = 0; j < 4; j++) lsum += (live[j] & (y4[j] != 0.f)) ? y4[j] * (lse - z4[j]) : 0.f; // -y ln p, SCE:216
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float avj = act_fn(p.last_act, z4[j]);                           // GNN:215-218
                const float df = avj - y4[j];
                out4[j] = live[j] ? avj : 0.f;
                dd4[j] = live[j] ? df * act_prime_from_a(p.last_act, avj) : 0.f;       // GNN:267-271
                lsum += live[j] ? 0.5f * df * df : 0.f;
                nan_flag = (live[j] & (4 * q + j == 0) & (avj != avj)) ? 1.f : nan_flag; // only a NaN at index 0 is sticky
                const bool take = live[j] & (avj >= mx);                               // a NaN is never selected
                mx = take ? avj : mx;
                best = take ? 4 * q + j : best;
            }
            if (p.label) quad_argmax(mx, best);
        }
        if (p.label) {
            nan_flag += dpp_f<QX1>(nan_flag);
            nan_flag += dpp_f<QX2>(nan_flag);
            if (nan_flag > 0.f) best = 0;
        }
        if (p.loss) {
            lsum += dpp_f<QX1>(lsum);
            lsum += dpp_f<QX2>(lsum);
        }
        if (STAMP) { asm volatile("" : "+v"(dd4)); GNN_RB_STAMP(11); } // output rule done
        float *dlast = smem + m.off_dl[Lm] + r * (16 + 4);
        if (kg == 0) {
            if (p.prob) *reinterpret_cast<f32x4 *>(p.prob + (size_t)row * 16 + 4 * q) = out4;
            if (BF) {
                *reinterpret_cast<m4_bf16x4 *>(p.deltab[Lm] + (size_t)row * 16 + 4 * q) = (m4_bf16x4){(__bf16)dd4[0], (__bf16)dd4[1], (__bf16)dd4[2], (__bf16)dd4[3]};
#pragma unroll
                for (int j = 0; j < 4; j++) dd4[j] = bf16_value(dd4[j]);
            } else {
                *reinterpret_cast<f32x4 *>(p.delta[Lm] + (size_t)row * 16 + 4 * q) = dd4;
            }
            *reinterpret_cast<f32x4 *>(dlast + 4 * q) = dd4;
        }
        if (lane == 0) {
            if (p.loss) p.loss[row] = row < p.B ? lsum : 0.f;
            if (p.label) p.label[row] = row < p.B ? best : -1;
        }
        {
            // delta_{L-2}[n] = (sum_c delta_{L-1}[c] W[n][c]) f'(a[n]): lane n and n + 64, both halves' reads in flight together;
            // the wave reads back its own 16 deltas (LDS keeps a wave's accesses in order); only the copied columns of W are used
            f32x4 d4[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) d4[qq] = *reinterpret_cast<const f32x4 *>(dlast + 4 * qq);
            if (STAMP) { asm volatile("" : "+v"(d4[3])); GNN_RB_STAMP(5); } // delta_{L-2}'s operands read
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int n = lane + 64 * half;
                if (n < ldp) {
                    float accd = 0.f;
                    if (n < K) {
#pragma unroll
                        for (int qq = 0; qq < 4; qq++)
                            if (4 * qq < m.kr[Lm])
#pragma unroll
                                for (int j = 0; j < 4; j++) accd = __builtin_fmaf(d4[qq][j], twn[half][qq][j], accd);
                    }
                    // (BF: f' of the unrounded activation was parked by the forward pass -- in delta_{L-2}'s own slot, or in the
                    //  f'(a_1) image when layer L-2 is layer 1)
                    const float fpv = BF ? an[half] : act_prime_from_a(ACT, an[half]);
                    const float v = (row < p.B && n < m.d[Lm - 1]) ? accd * fpv : 0.f;
                    if (Lm - 1 > 1) smem[m.off_dl[Lm - 1] + r * (ldp + 4) + n] = opv(v);
                    if (BF) p.deltab[Lm - 1][(size_t)row * ldp + n] = (__bf16)v;
                    else p.delta[Lm - 1][(size_t)row * ldp + n] = v;
                }
            }
        }
    }
    else if (DEFER && UPW1 > 0 && Lm >= 3) {
#pragma unroll
        for (int uu = 0; uu < UPW1; uu++)
            if (!(EARLY_IMG && early_here && uu < PF0) && uu < nu_1) to_image_1(uu);
    }
    if (wave >= 4 && x_copy) { // (wave-uniform) wave 4 + r: input row r of this block
        // (the row's index was fetched at the top; every load of the row in flight before the first store: as a loop of
        //  load - wait - store the copy took as long as the row tail and held up the barrier behind it)
        const int r = wave - 4, row = row0 + r;
        const bool live = row < p.B; // (a copied batch has this batch's row count: chain_gradient)
        const size_t src = live ? (size_t)x_ld * p.ldx : 0, dst = (size_t)row * p.ldx;
        constexpr int XC = 4; // 4 x 64 lanes x 4 elements = 1024 columns
        if (p.xcopy) {
            f32x4 v[XC];
#pragma unroll
            for (int i = 0; i < XC; i++) {
                const int c = 4 * lane + 256 * i;
                v[i] = *reinterpret_cast<const f32x4 *>(p.X + src + (c < p.ldx ? c : 0));
            }
            __builtin_amdgcn_s_waitcnt(0x0F70); // (every load of the row waited for HERE: one that a lane does not store stays "pending" for
                                                //  the compiler, which then drains the memory counter -- the tail's stores included -- behind the barrier)
#pragma unroll
            for (int i = 0; i < XC; i++) {
                const int c = 4 * lane + 256 * i;
                if (c < p.ldx) *reinterpret_cast<f32x4 *>(p.xcopy + dst + c) = live ? v[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        } else {
            uint2 v[XC];
#pragma unroll
            for (int i = 0; i < XC; i++) {
                const int c = 4 * lane + 256 * i;
                v[i] = *reinterpret_cast<const uint2 *>(p.Xb + src + (c < p.ldx ? c : 0));
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
            for (int i = 0; i < XC; i++) {
                const int c = 4 * lane + 256 * i;
                if (c < p.ldx) *reinterpret_cast<uint2 *>(p.xcopyb + dst + c) = live ? v[i] : make_uint2(0u, 0u);
            }
        }
    }
    if (TUNE & 32) __builtin_amdgcn_s_setprio(0);
    GNN_RB_STAMP(4);  // wave 0's tail done
    GNN_RB_WSTAMP(3); // this wave at the end of the row-tail phase
    __syncthreads();
    GNN_RB_STAMP(12);

    // ---- backward data: delta_l = (delta_{l+1} . W_l^T) * f'(z_l), l = L-3 .. 1 (SCE:262-278), from the LDS images ------
    // One wave per group of 64 neurons over the WHOLE contraction: the four rows' sums stay in the accumulators and f',
    // the mask and the stores follow at once -- no K split, no partial tiles, no barrier before the epilogue (the split
    // form spent 3 400 cycles on a product whose MFMAs take ~1 000).
#pragma unroll
    for (int li = 0; li < MAX_LAYERS; li++) {
        const int l = Lm - 2 - li;
        if (l < 1) break;
        const int N = m.ld[l], NR = m.kr[l], G = m.gb[l];
        const int k4n = m.kr[l + 1] / 4;
        for (int g = wave; g < G; g += RB_NW) { // (wave-uniform)
            const int n = g * 64 + lane;
            const float *arow = smem + m.off_dl[l + 1] + (lane & 3) * (m.ld[l + 1] + 4);
            const float *wp = smem + m.off_w[l] + (n < NR ? n : NR - 1) * m.lw[l]; // columns past the image compute garbage nobody stores
            // (the epilogue's operand -- f'(z_l) from a_l, or the parked f' -- requested in front of the product)
            float alr[4];
#pragma unroll
            for (int r = 0; r < 4; r++) alr[r] = smem[(BF ? (l > 1 ? m.off_dl[l] : m.off_fp1) : m.off_act[l]) + r * (N + 4) + (n < N ? n : 0)];
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            constexpr int U = 4;
            int kb = 0;
            for (; kb + U <= k4n; kb += U) {
                f32x4 a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; u++) { a[u] = *reinterpret_cast<const f32x4 *>(arow + 4 * (kb + u)); b[u] = *reinterpret_cast<const f32x4 *>(wp + 4 * (kb + u)); }
#pragma unroll
                for (int u = 0; u < U; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (u & 1) acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][j], b[u][j], acc1, 0, 0, 0);
                        else acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][j], b[u][j], acc0, 0, 0, 0);
                    }
            }
            for (; kb < k4n; kb++) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(arow + 4 * kb), b = *reinterpret_cast<const f32x4 *>(wp + 4 * kb);
#pragma unroll
                for (int j = 0; j < 4; j++) acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[j], b[j], acc0, 0, 0, 0);
            }
            const f32x4 acc = acc0 + acc1; // register r of lane: delta_l[row r][n] before f'
            if (n < N) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    // f'(z_l) from a_l = f(z_l); BF: parked by the forward pass (from the unrounded a_l) in delta_l's slot / the f'(a_1) image
                    const float al = alr[r];
                    const float v = (row0 + r < p.B && n < m.d[l]) ? acc[r] * (BF ? al : act_prime_from_a(ACT, al)) : 0.f;
                    if (BF) p.deltab[l][(size_t)(row0 + r) * N + n] = (__bf16)v;
                    else p.delta[l][(size_t)(row0 + r) * N + n] = v;
                    if (l > 1) smem[m.off_dl[l] + r * (N + 4) + n] = opv(v);
                }
            }
        }
        if (l > 1) __syncthreads();
        GNN_RB_STAMP(13);
    }
    GNN_RB_WSTAMP(4); // this wave done
    GNN_RB_STAMP(14);
}

// TUNE: development knob of tools/rowblock_probe (0 = the shipped schedule)
// The arguments the kernel's FIRST instructions need travel ahead of the struct (RbHead: 13 dwords): the library is built with
// -mllvm -amdgpu-kernarg-preload-count=16, so the dispatch hands them over in SGPRs and the first vector loads do not wait
// for a scalar-cache miss on the kernel-argument segment (6.63 -> 6.45 us, tools/rowblock_probe).  The struct's own copies
// of these fields are overwritten from them.
#define GNN_RB_HEAD_PARAMS const float *slabs, const float *W1, const float *Wlast, const int32_t *row_idx, const float *Y, const int32_t *copy_idx, int B, int slab_rows, int ldy
template <class SH, int ACT, int OUTK, bool STAMP = false, int TUNE = 0, bool BF = false>
__global__ __launch_bounds__(RB_NT) void rowblock_kernel(GNN_RB_HEAD_PARAMS, RbParams p) {
    static_assert(!BF || SH::kL == 3 || SH::kL == 4, "the bf16 row-block kernel: nets of three and four layers");
    p.slabs = slabs; p.row_idx = row_idx; p.copy_idx = copy_idx;
    if constexpr (BF) { // (the bf16 kernel streams the bf16 shadows: THEIR pointers travel in the two weight slots)
        p.Wb[1] = reinterpret_cast<const __bf16 *>(W1);
        if constexpr (SH::kL > 0) p.Wb[SH::kL - 2] = reinterpret_cast<const __bf16 *>(Wlast);
    } else {
        p.W[1] = W1;
        if constexpr (SH::kL > 0) p.W[SH::kL - 2] = Wlast; // (the instance for any layer count keeps the struct's own pointer)
    }
    p.Y = Y; p.B = B; p.slab_rows = slab_rows; p.ldy = ldy;
    if constexpr (SH::is_static) {
        constexpr RbPlan m = SH::make(); // a LOCAL constexpr object: member accesses with constant indices fold to immediates
        static_assert(m.ok, "this shape does not fit the row-block kernel");
        rowblock_body<SH::kL, true, ACT, OUTK, STAMP, m.ns, (SH::kL >= 4 ? m.upw[1] : 0), TUNE, BF>(m, p);
    } else {
        rowblock_body<SH::kL, false, ACT, OUTK, STAMP, MID4_MAX_SLABS, (SH::kL == 3 ? 0 : RB_MAXU), TUNE, BF>(p.plan, p);
    }
}

} // namespace gnn
