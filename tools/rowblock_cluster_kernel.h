// rowblock_cluster_kernel.h (tools/: NOT part of the library -- measured 49 % slower, profiles/r04/rowblock_cluster_experiment.md) -- round 4's STRUCTURAL EXPERIMENT on the training row-block kernel (rowblock_kernel.h): a row
// block of four batch rows shared by a CLUSTER of four workgroups.
//
// rowblock_kernel runs 32 workgroups on a 256-CU chip at B = 128, and every one of them pulls all middle weights (124 KB for
// 784-300-100-10) through its own CU's load path: ~3 700 cycles of load-pipe time per workgroup is that decomposition's floor
// (DESIGN 3.1).  Here workgroup s of a cluster owns
//   * forward:  the column slice [SL s, SL s + SL) of layer 2 (SL = ld_2 / 4 = 28): Z_2[4 x SL] = A_1[4 x K] . W_1[K x SL] -- a
//               quarter of W_1 (33 KB), straight into registers: 5 float4 per lane;
//   * backward: the slice [RS s, RS s + RS) of delta_1 (RS = ld_1 / 4 = 76): delta_1[4 x RS] = delta_2[4 x d_2] . W_1[RS rows]^T
//               -- another quarter of W_1 (30 KB), requested at the kernel's top, used at its end: 7 float4 per lane;
// and the four A_2 slices (4 x 28 floats each) are EXCHANGED through memory, after which every workgroup of the cluster holds
// the whole of A_2 and runs the row tail (logits, softmax / cross-entropy or the element-wise output rule, delta_3, delta_2)
// redundantly -- bitwise the same in all four, because the slices are summed by their owners only, in a fixed order.
//
// The exchange (tools/exchange_probe.hip measured the primitive first): a memory-side round trip costs ~1 000 cycles here even
// mid-kernel, and "data, wait for the acknowledgement, flag -- poll the flag, then read the data" is four of them in a row
// (~4 500 cycles = 1.9 us: more than the split saves).  So every exchanged float travels WITH its validity tag in one 64-bit
// store -- {value, epoch}, a relaxed agent-scope atomic, single-copy atomic by definition (the scheme of RCCL's LL protocol) --
// and a consumer polls the DATA: one propagation + one load round trip when the producer was there first.  `epoch` is a
// launch counter that lives on the DEVICE (XchSync): the last workgroup of a launch advances it, so that a launch replayed
// from a hipGraph (constant kernel arguments) still sees a fresh value.  Polls are bounded (XCH_SPIN_LIMIT): a peer that never
// arrives ends the wait, raises XchSync::error, and the kernel terminates with garbage rather than hanging the device.
// Residency: cluster c = workgroups (c / 8) * 32 + 8 s + c % 8 -- within 32 consecutive workgroup ids, dispatched together
// (128 workgroups for B = 128: all resident at once), and the same id mod 8 = the same XCD, though nothing depends on that.
//
// Applies to nets of FOUR layers (two middle matrices) with <= 16 outputs, ld_2 <= 128, ld_1 <= 512.  f32 only.
#pragma once
#include "../graph-neural-net_amd/csrc/rowblock_kernel.h"

namespace gnn {

constexpr int RBC_NW = 8, RBC_NT = RBC_NW * 64;
constexpr unsigned XCH_SPIN_LIMIT = 1u << 20;

struct XchSync {          // one per handle, device memory, zero-initialised
    unsigned launch;      // launches completed: this launch's epoch = launch + 1
    unsigned done;        // workgroups of the running launch that are past their last poll
    unsigned error;       // != 0: a poll ran into XCH_SPIN_LIMIT in some launch (results of that launch are garbage)
    unsigned pad;
};

struct RbcParams {
    const float *W1, *W2;        // middle matrices [ld1][ld2], [ld2][ld3]
    float *act1, *act2;          // A_1, A_2 out (the tile kernel reads them)
    float *delta1, *delta2, *delta3;
    const float *Y; int ldy;
    float *prob; float *loss; int32_t *label;
    int B;
    int last_act;
    const int32_t *row_idx;
    const float *slabs; int slab_rows;
    unsigned long long *xch;     // [clusters][4 slices][4 rows][SL] {value, epoch} pairs
    XchSync *sync;
    unsigned long long *stamps;  // STAMP builds: 16 slots per workgroup
};

#define GNN_RBC_STAMP(i) do { if (STAMP && threadIdx.x == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)

template <int D0, int D1, int D2, int D3, int ACT, int OUTK, bool STAMP = false>
__global__ __launch_bounds__(RBC_NT) void rowblock_cluster_kernel(RbcParams p) {
    constexpr int ld1 = pad_up(D1), ld2 = pad_up(D2), ld3 = pad_up(D3);
    static_assert(ld3 == 16 && ld2 <= 128 && ld1 <= 512, "rowblock_cluster_kernel: shape outside the plan");
    constexpr int NS = (pad_up(D0) + 63) / 64;        // first-layer K slabs
    constexpr int SL = ld2 / 4, SLQ = SL / 4;         // columns (float4s) of a forward slice: 28 (7)
    constexpr int RS = ld1 / 4;                       // rows of W_1 in a backward slice: 76
    constexpr int KW = ((ld1 + 63) / 64) * 8;         // k values a wave takes in the slice product, 8 per step: 40
    constexpr int KST = KW / 8;                       // 5
    constexpr int MB = (ld2 / 4 + 3) / 4;             // float4s of a W_1 row per backward lane (4 lanes per row): 7
    constexpr int kr3 = (D3 + 3) / 4 * 4, c4l = kr3 / 4, lw2 = 4 * ((c4l % 2) ? c4l : c4l + 1); // last image: [ld2 rows][lw2]
    constexpr int q1 = ld1 / 4;
    static_assert(4 * q1 <= RBC_NT && 4 * RS <= RBC_NT && D2 * c4l <= RBC_NT, "one element per thread");
    // LDS
    constexpr int off_a1 = 0;                          // A_1 image [4][ld1 + 4]
    constexpr int off_part = off_a1 + 4 * (ld1 + 4);   // partial slices [8 waves][4][SL]
    constexpr int off_a2 = off_part + RBC_NW * 4 * SL; // A_2 image [4][ld2 + 4]
    constexpr int off_d2 = off_a2 + 4 * (ld2 + 4);     // delta_2 image [4][ld2 + 4]
    constexpr int off_w2 = off_d2 + 4 * (ld2 + 4);     // W_2 image [ld2][lw2]
    constexpr int off_y = off_w2 + ld2 * lw2;          // expected rows [4][16]
    constexpr int off_d3 = off_y + 64;                 // delta_3 [4][20]
    constexpr int lds_floats = off_d3 + 80;
    __shared__ __attribute__((aligned(16))) float smem[lds_floats];

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int i = blockIdx.x;
    const int c = (i >> 5) * 8 + (i & 7), s = (i >> 3) & 3;   // cluster, slice
    const int row0 = 4 * c;
    if (row0 >= pad_up(p.B)) return;                   // (a whole cluster leaves together: nobody waits for it)
    const unsigned epoch = __hip_atomic_load(&p.sync->launch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u; // (written only by the LAST workgroup of a launch)
    GNN_RBC_STAMP(0);

    // ---- phase 0: every load of the kernel, in the order of use ---------------------------------------------------------
    const int qy = 4; // ld3 / 4
    const int y_e = RBC_NT - 1 - t, y_r = y_e / qy, y_q = y_e - y_r * qy; // the expected rows: the LAST threads
    const bool y_on = p.Y != nullptr && y_e < 4 * qy;
    const bool y_ix = y_on && p.row_idx != nullptr && row0 + y_r < p.B;
    const int y_ld = *(y_ix ? p.row_idx + (row0 + y_r) : reinterpret_cast<const int32_t *>(p.slabs));
    // W_2 (the row tail's image)
    const int wl_r = t / c4l, wl_c = t - wl_r * c4l;
    const bool wl_on = t < D2 * c4l;
    const f32x4 wl = m4_load16(p.W2, wl_on ? (unsigned)(wl_r * ld3 + 4 * wl_c) : 0u);
    // the first-layer K slabs of the four rows
    const bool a1_on = t < 4 * q1;
    const int a1_r = t / q1, a1_q = t - a1_r * q1;
    f32x4 zs[NS];
    {
        const unsigned zoff = a1_on ? (unsigned)(row0 + a1_r) * (unsigned)ld1 + (unsigned)(a1_q * 4) : 0u;
        const unsigned sstride = (unsigned)p.slab_rows * (unsigned)ld1;
        if (wave * 64 < 4 * q1) { // (wave-uniform: only the waves that own an A_1 element issue slab loads)
#pragma unroll
            for (int k = 0; k < NS; k++) zs[k] = m4_load16(p.slabs, zoff + (a1_on ? (unsigned)k * sstride : 0u));
        }
    }
    // W_1, forward slice: lane (kq = lane / 8, g = lane % 8): row k = KW wave + 8 st + kq, columns SL s + 4 g .. + 3
    const int kq = lane >> 3, g = lane & 7;
    const int gc = g < SLQ ? g : SLQ - 1;  // (g = 7 has no columns: clamped, never used)
    f32x4 wf[KST];
#pragma unroll
    for (int st = 0; st < KST; st++) {
        const int k = KW * wave + 8 * st + kq;
        wf[st] = m4_load16(p.W1, (unsigned)((k < ld1 ? k : ld1 - 1) * ld2 + SL * s + 4 * gc));
    }
    // W_1, backward slice: thread (n = t / 4, bq = t % 4): row RS s + n, float4s bq + 4 m
    const int bn = t >> 2, bq = t & 3;
    const bool b_on = t < 4 * RS;
    f32x4 wb[MB];
    if (wave * 64 < 4 * RS) { // (wave-uniform: a wave-load costs the CU's load pipe 16 cycles whatever its lanes carry)
#pragma unroll
        for (int m = 0; m < MB; m++) {
            const int c4 = bq + 4 * m;
            wb[m] = m4_load16(p.W1, (unsigned)((RS * s + (b_on ? bn : 0)) * ld2 + 4 * (c4 < ld2 / 4 ? c4 : ld2 / 4 - 1)));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    GNN_RBC_STAMP(1); // loads issued

    // ---- phase 1: A_1 = f(sum of the slabs, slab order) ---------------------------------------------------------------------
    if (wave * 64 < 4 * q1) {
        f32x4 z = zs[0];
#pragma unroll
        for (int k = 1; k < NS; k++) z += zs[k];
        const bool lrow = row0 + a1_r < p.B;
        f32x4 a;
#pragma unroll
        for (int j = 0; j < 4; j++) a[j] = (lrow && a1_q * 4 + j < D1) ? act_fn(ACT, z[j]) : 0.f;
        if (a1_on) {
            *reinterpret_cast<f32x4 *>(smem + off_a1 + a1_r * (ld1 + 4) + a1_q * 4) = a;
            if (a1_r == s) *reinterpret_cast<f32x4 *>(p.act1 + (size_t)(row0 + a1_r) * ld1 + a1_q * 4) = a; // (row s of the block: the four workgroups hold the same A_1)
        }
    }
    if (wl_on) *reinterpret_cast<f32x4 *>(smem + off_w2 + wl_r * lw2 + 4 * wl_c) = wl;
    {
        const int y_idx = p.row_idx ? (y_ix ? y_ld : 0) : row0 + y_r;
        if (y_on) *reinterpret_cast<f32x4 *>(smem + off_y + y_r * 16 + y_q * 4) = *reinterpret_cast<const f32x4 *>(p.Y + (size_t)y_idx * p.ldy + y_q * 4);
    }
    __syncthreads();
    GNN_RBC_STAMP(2); // A_1 barrier

    // ---- phase 2: the slice product on v_mfma_f32_4x4x1: the sixteen blocks of the instruction sit at EIGHT different k ---------
    // lane = (blk = lane / 4, e = lane % 4); kq = blk / 2: A operand = A_1[row e][k(kq)], B operand = W_1[k(kq)][column 4 g + j] with
    // g = (blk % 2) * 4 + e -- component j of the float4 the lane loaded.  acc[j][row] = this lane's column 4 g + j over its k's.
    {
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float av[KST];
#pragma unroll
        for (int st = 0; st < KST; st++) {
            const int k = KW * wave + 8 * st + kq;
            av[st] = k < ld1 ? smem[off_a1 + (lane & 3) * (ld1 + 4) + k] : 0.f; // (A_1's pad columns are zeros; past ld1: no operand)
        }
#pragma unroll
        for (int st = 0; st < KST; st++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_4x4x1f32(av[st], wf[st][j], acc[j], 0, 0, 0);
        // the eight k groups of the wave: lanes 8, 16 and 32 apart
        f32x4 rowv[4]; // rowv[r] = columns 4 g .. 4 g + 3 of row r
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                float v = acc[j][r];
                v += dpp_f<0x128>(v); // row_ror:8 = lane ^ 8 within a row of 16
                v = rb_sum16(v);
                v = rb_sum32(v);
                rowv[r][j] = v;
            }
        if (lane < SLQ) {
#pragma unroll
            for (int r = 0; r < 4; r++) *reinterpret_cast<f32x4 *>(smem + off_part + (wave * 4 + r) * SL + 4 * lane) = rowv[r];
        }
    }
    __syncthreads();
    GNN_RBC_STAMP(3); // partial slices in LDS

    // ---- phase 3: row tail, wave r = row r: this slice of A_2 (owner's sum, wave order), exchange, logits .. delta_2 ----------------
    if (wave < 4) {
        const int r = wave, row = row0 + r;
        const bool lrow = row < p.B;
        unsigned long long *xrow = p.xch + ((size_t)(c * 4) * 4) * SL; // this cluster's [slice][row][SL]
        if (lane < SLQ) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < RBC_NW; w++) v += *reinterpret_cast<const f32x4 *>(smem + off_part + (w * 4 + r) * SL + 4 * lane);
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = (lrow && SL * s + 4 * lane + j < D2) ? act_fn(ACT, v[j]) : 0.f;
            unsigned long long *dst = xrow + ((size_t)s * 4 + r) * SL + 4 * lane;
            // (each element copied to a scalar before it is reinterpreted: `__builtin_bit_cast(unsigned, v[j])` on a vector ELEMENT read
            //  element 0 four times with hipcc 7.2 -- the bug rowblock_kernel.h's rb_sum32 notes; every peer then saw column 4 g in all four)
            const float v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
            const unsigned long long tag = (unsigned long long)epoch << 32;
            __hip_atomic_store(dst + 0, tag | (unsigned long long)__float_as_uint(v0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 1, tag | (unsigned long long)__float_as_uint(v1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 2, tag | (unsigned long long)__float_as_uint(v2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dst + 3, tag | (unsigned long long)__float_as_uint(v3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *reinterpret_cast<f32x4 *>(p.act2 + (size_t)row * ld2 + SL * s + 4 * lane) = v;
            *reinterpret_cast<f32x4 *>(smem + off_a2 + r * (ld2 + 4) + SL * s + 4 * lane) = v;
        }
        if (STAMP && wave == 0) GNN_RBC_STAMP(4); // own slice published
        // the three peers' slices of this row: 3 SL elements over the lanes, two rounds; the DATA is polled
        bool ok = true;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int e = lane + 64 * h;
            if (e < 3 * SL) {
                const int q = e / SL, col = e - q * SL, peer = q + (q >= s ? 1 : 0);
                const unsigned long long *src = xrow + ((size_t)peer * 4 + r) * SL + col;
                unsigned long long u = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while ((unsigned)(u >> 32) != epoch) {
                    if (++spins > XCH_SPIN_LIMIT) { ok = false; break; }
                    __builtin_amdgcn_s_sleep(1);
                    u = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                smem[off_a2 + r * (ld2 + 4) + SL * peer + col] = __uint_as_float((unsigned)u);
            }
        }
        if (!__all(ok) && lane == 0) __hip_atomic_store(&p.sync->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (STAMP && wave == 0) GNN_RBC_STAMP(5); // peers' slices here
        // ---- the tail proper (rowblock_kernel.h's, on the LDS images): lane (kg = lane / 4, q = lane % 4)
        constexpr int TK = (ld2 + 15) / 16; // k = kg + 16 i
        const float *a = smem + off_a2 + r * (ld2 + 4);
        const float *Wl = smem + off_w2;
        const int kg = lane >> 2, q = lane & 3;
        const bool q_on = 4 * q < kr3;
        const f32x4 y4 = (lrow && p.Y) ? *reinterpret_cast<const f32x4 *>(smem + off_y + r * 16 + 4 * q) : (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ii = 0; ii < TK; ii++) {
            const int k = kg + 16 * ii;
            const bool on = k < D2 && q_on;
            const float avv = on ? a[k] : 0.f;
            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wl + (on ? k * lw2 + 4 * q : 0));
#pragma unroll
            for (int j = 0; j < 4; j++) z4[j] = __builtin_fmaf(avv, w4[j], z4[j]);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v = z4[j];
            v += dpp_f<0x128>(v);
            v += dpp_f<0x124>(v);
            v = rb_sum16(v);
            v = rb_sum32(v);
            z4[j] = v;
        }
        constexpr int QX1 = 0xB1, QX2 = 0x4E;
        f32x4 out4 = {0.f, 0.f, 0.f, 0.f}, dd4 = {0.f, 0.f, 0.f, 0.f};
        float lsum = 0.f, mx = -__builtin_inff(), nan_flag = 0.f;
        int best = -1;
        bool valid[4], live[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { valid[j] = 4 * q + j < D3; live[j] = valid[j] && lrow; }
        auto quad_argmax = [&](float &v, int &ix) {
#define GNN_RBC_QSTEP(CTRL) { const float ov = dpp_f<CTRL>(v); const int oi = dpp_i<CTRL>(ix); const bool tk = (ov > v) | ((ov == v) & (oi > ix)); v = tk ? ov : v; ix = tk ? oi : ix; }
            GNN_RBC_QSTEP(QX1)
            GNN_RBC_QSTEP(QX2)
#undef GNN_RBC_QSTEP
        };
        if (OUTK == 0) {
            if (p.label) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float zj = z4[j];
                    nan_flag = (valid[j] & (zj != zj)) ? 1.f : nan_flag;
                    const bool take = valid[j] & (zj >= mx);
                    mx = take ? zj : mx;
                    best = take ? 4 * q + j : best;
                }
                quad_argmax(mx, best);
            } else {
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
            if (p.loss) {
                const float lse = mx + __logf(ssum);
#pragma unroll
                for (int j = 0; j < 4; j++) lsum += (live[j] & (y4[j] != 0.f)) ? y4[j] * (lse - z4[j]) : 0.f; // SCE:216
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float avj = act_fn(p.last_act, z4[j]);                           // GNN:215-218
                const float df = avj - y4[j];
                out4[j] = live[j] ? avj : 0.f;
                dd4[j] = live[j] ? df * act_prime_from_a(p.last_act, avj) : 0.f;       // GNN:267-271
                lsum += live[j] ? 0.5f * df * df : 0.f;
                nan_flag = (live[j] & (4 * q + j == 0) & (avj != avj)) ? 1.f : nan_flag;
                const bool take = live[j] & (avj >= mx);
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
        float *dlast = smem + off_d3 + r * 20;
        if (kg == 0) {
            if (s == 0) { // (one workgroup of the cluster writes what all four computed)
                if (p.prob) *reinterpret_cast<f32x4 *>(p.prob + (size_t)row * 16 + 4 * q) = out4;
                *reinterpret_cast<f32x4 *>(p.delta3 + (size_t)row * 16 + 4 * q) = dd4;
            }
            *reinterpret_cast<f32x4 *>(dlast + 4 * q) = dd4;
        }
        if (lane == 0 && s == 0) {
            if (p.loss) p.loss[row] = lrow ? lsum : 0.f;
            if (p.label) p.label[row] = lrow ? best : -1;
        }
        {
            // delta_2[n] = (sum_c delta_3[c] W_2[n][c]) f'(a_2[n]): lanes n and n + 64 -- the WHOLE row in every workgroup (the
            // backward slice contracts over all of it); this workgroup's slice of it goes to memory
            f32x4 d4[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) d4[qq] = *reinterpret_cast<const f32x4 *>(dlast + 4 * qq);
#pragma unroll
            for (int half = 0; half < 2; half++) {
                const int n = lane + 64 * half;
                if (n < ld2) {
                    float accd = 0.f;
                    if (n < D2) {
#pragma unroll
                        for (int qq = 0; qq < c4l; qq++) {
                            const f32x4 w4 = *reinterpret_cast<const f32x4 *>(Wl + n * lw2 + 4 * qq);
#pragma unroll
                            for (int j = 0; j < 4; j++) accd = __builtin_fmaf(d4[qq][j], w4[j], accd);
                        }
                    }
                    const float v = (lrow && n < D2) ? accd * act_prime_from_a(ACT, a[n]) : 0.f;
                    smem[off_d2 + r * (ld2 + 4) + n] = v;
                    if (p.last_act == 99) { if (s == 1) p.delta2[(size_t)row * ld2 + n] = a[n]; } // (probe: the A_2 image this workgroup multiplied)
                    else if (n >= SL * s && n < SL * s + SL) p.delta2[(size_t)row * ld2 + n] = v;
                }
            }
        }
    }
    __syncthreads();
    GNN_RBC_STAMP(6); // tail done

    // ---- phase 4: this workgroup's slice of delta_1 from the W_1 rows it has held in registers since the top -------------------
    if (b_on) { // (whole waves but the last)
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MB; m++) {
            const int c4 = bq + 4 * m;
            if (c4 < ld2 / 4) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const f32x4 d = *reinterpret_cast<const f32x4 *>(smem + off_d2 + r * (ld2 + 4) + 4 * c4);
#pragma unroll
                    for (int j = 0; j < 4; j++) acc[r] = __builtin_fmaf(d[j], wb[m][j], acc[r]);
                }
            }
        }
        // the four lanes of a row: quad butterfly; lane bq keeps batch row bq
#pragma unroll
        for (int r = 0; r < 4; r++) {
            acc[r] += dpp_f<0xB1>(acc[r]);
            acc[r] += dpp_f<0x4E>(acc[r]);
        }
        const float mine = bq == 0 ? acc[0] : bq == 1 ? acc[1] : bq == 2 ? acc[2] : acc[3];
        const int n = RS * s + bn;
        const float al = smem[off_a1 + bq * (ld1 + 4) + n];
        const float v = (row0 + bq < p.B && n < D1) ? mine * act_prime_from_a(ACT, al) : 0.f;
        p.delta1[(size_t)(row0 + bq) * ld1 + n] = v;
    }
    GNN_RBC_STAMP(7);
    // ---- the launch counter: the last workgroup past its polls advances the epoch (all live workgroups: 4 x clusters) ---------
    if (t == 0) {
        const unsigned live_wgs = 4u * (unsigned)(pad_up(p.B) / 4);
        const unsigned before = __hip_atomic_fetch_add(&p.sync->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == live_wgs) {
            __hip_atomic_store(&p.sync->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.sync->launch, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// workgroups of a launch for B rows: clusters rounded up to whole groups of eight (32 workgroup ids)
inline int rbc_grid(int B) { const int clusters = pad_up(B) / 4; return ((clusters + 7) / 8) * 32; }

} // namespace gnn
