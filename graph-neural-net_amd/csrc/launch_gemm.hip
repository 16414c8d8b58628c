// launch_gemm.hip -- host side of the per-layer f32 GEMM path (kernels.h, gemm_wavek.h) and of the three-launch
// small-net kernels (fused_kernels.h): tile choice and launches.
#include "handle.h"
#include "gemm_f32_dma.h"

using namespace gnn;
using namespace gnn::host;

namespace gnn {
namespace host {

// ---- GEMM dispatch ----------------------------------------------------------------------
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2>
void launch_gemm_t(gnn_mlp *h, int cls, const GemmParams &p) {
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    launch_timed(h, cls, gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, WM>, grid, dim3(WM * 128), 0, GNN_GEMM_HEAD_ARGS(p), p);
}

// Operand tiles by LDS DMA (gemm_f32_dma.h): whole 64 x 64 tiles of the forms with a k-contiguous operand, whose transposing
// ds_write_b32 stores it removes -- backward data 512 x 2048 x 2048 47.0 -> 39.4 us, 512 x 2048 x 1024 23.7 -> 21.0, forward
// 512 x 2048 x 4096 78.7 -> 76.6 (profiles/r04/gemm_probe_f32_dma.log).  The gradient forms (both operands k-major, already
// written with ds_write_b128) gain nothing from it and stay on gemm_f32_kernel.
template <bool A_KC, bool B_KC, int EPI>
bool launch_gemm_dma64(gnn_mlp *h, int cls, const GemmParams &p) {
    if (h->env_f32_dma_off || p.M % 64 || p.N % 64 || p.K % GemmF32DmaDepth<64>::BK) return false;
    constexpr size_t lds = gemm_f32_dma_lds_bytes<64, 64, 3>();
    auto kern = gemm_f32_dma_kernel<64, 64, A_KC, B_KC, EPI, 4, 3>;
    static bool opted_in[kMaxOptInDevices] = {}; // (per instantiation, per device)
    opt_in_dynamic_lds(h, kern, lds, opted_in);
    launch_timed(h, cls, kern, dim3(p.N / 64, p.M / 64), dim3(512), lds, GNN_GEMM_HEAD_ARGS(p), p);
    return true;
}

// tile edge: keep >= ~256 workgroups in flight where the problem allows it (256 CUs)
int pick_tile(int M, int N) {
    auto tiles = [&](int b) { return (int64_t)((M + b - 1) / b) * ((N + b - 1) / b); };
    if (tiles(128) >= 256) return 128;
    if (tiles(64) >= 256) return 64;
    return 32;
}

// 32 x 32 tiles, K split over the waves of a workgroup (gemm_wavek.h): for outputs too small for 64-wide tiles to fill
// the chip.  256 x 1024 x 1024: 9.1 us against 12.2 us for gemm_f32_kernel<32, 32> (profiles/r02/gemm_probe_wavek3.log).
bool wavek_fits(int M, int N, int K) { return M % 32 == 0 && N % 32 == 0 && K >= 128; }
template <bool A_KC, bool B_KC, int EPI>
void launch_gemm_wavek(gnn_mlp *h, int cls, const GemmParams &p) {
    launch_timed(h, cls, gemm_f32_wavek_kernel<A_KC, B_KC, EPI, 4, 2>, dim3(p.N / 32, p.M / 32), dim3(256), 0, GNN_GEMM_HEAD_ARGS(p), p);
}
// the ragged form (extents that are multiples of 16 only): forward products of a few hundred rows against a 300-wide layer
template <bool A_KC, bool B_KC, int EPI>
void launch_gemm_wavek_ragged(gnn_mlp *h, int cls, const GemmParams &p) {
    launch_timed(h, cls, gemm_f32_wavek_kernel<A_KC, B_KC, EPI, 4, 2, true, true>, dim3((p.N + 31) / 32, (p.M + 31) / 32), dim3(256), 0, GNN_GEMM_HEAD_ARGS(p), p);
}

template <bool A_KC, bool B_KC, int EPI>
void launch_gemm(gnn_mlp *h, int cls, const GemmParams &p) {
    int tile = pick_tile(p.M, p.N);
    if (tile == 128) {
        // a 128-wide tile grid that overhangs the output by much more than the 64-wide one does multiplies zeros: 16 384 x 304 (the
        // first layer of 784-300-100-10 over an evaluation block) is 3 tile columns of 128 = 384 for 304 (79 %) against 5 of 64 = 320 (95 %)
        auto eff = [&](int b) { return (double)p.M * p.N / ((double)((p.M + b - 1) / b * b) * ((p.N + b - 1) / b * b)); };
        if (eff(64) > 1.1 * eff(128)) tile = 64;
    }
    if (tile == 32 && !h->env_wavek_off && wavek_fits(p.M, p.N, p.K)) { launch_gemm_wavek<A_KC, B_KC, EPI>(h, cls, p); return; }
    // A square grid of 256..511 tiles is ONE 4-wave workgroup per CU: nothing covers its barriers and LDS latencies.  Measured
    // per form (profiles/r02/gemm_probe_tiles2.log, 512-row products of 4096-2048-2048-1024, one register stage, unguarded loads):
    //   forward and backward data (a k-contiguous operand): 64 x 64 tiles with EIGHT waves -- two per SIMD from one workgroup and
    //     a third less operand traffic than 64 x 32 (512 x 2048 x 4096: 81.3 against 88.3 us; backward 512 x 2048 x 1024: 23.8
    //     against 30.0);
    //   gradient (both k-major): 64 x 32 below 512 tiles of 64 x 64; 128 x 128 tiles only from 512 of them up, 256..511 of them
    //     run as 64 x 64 (2048 x 2048 x 512 with the update: 45.4 against 48.1 us).
    const int64_t t64 = (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64), t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if constexpr (A_KC) {
        if (tile == 64 && launch_gemm_dma64<A_KC, B_KC, EPI>(h, cls, p)) return;
    }
    if (tile == 64 && t64 < 512) {
        if constexpr (A_KC) launch_gemm_t<64, 64, A_KC, B_KC, EPI, 4>(h, cls, p);
        else launch_gemm_t<64, 32, A_KC, B_KC, EPI>(h, cls, p);
        return;
    }
    // (64 x 64 tiles always with eight waves: 2048 x 2048 x 512 with the update 42.5 against 44.7 us, gemm_probe_stages2.log)
    if (tile == 128 && t128 < 512 && !A_KC && !B_KC) { launch_gemm_t<64, 64, A_KC, B_KC, EPI, 4>(h, cls, p); return; }
    switch (tile) {
    case 128: launch_gemm_t<128, 128, A_KC, B_KC, EPI>(h, cls, p); break;
    case 64: launch_gemm_t<64, 64, A_KC, B_KC, EPI, 4>(h, cls, p); break;
    default: launch_gemm_t<32, 32, A_KC, B_KC, EPI>(h, cls, p); break;
    }
}

constexpr int FIRST_NW = 8;  // waves per fwd_first_kernel workgroup (K split in-LDS)

// ---- forward (SCE:164-198): a0 = f(x) rows, B live rows ------------------------------------
// leaves act[1..L-2], logits; the output kernel is launched by the caller via run_output.
void forward(gnn_mlp *h, const float *a0, int B, int first_l, bool stop_before_last) {
    const int B_pad = pad_up(B);
    const float *in = (first_l == 1) ? a0 : h->act[first_l - 1];
    for (int l = first_l; l < h->L - (stop_before_last ? 1 : 0); l++) {
        GemmParams p{};
        p.A = in; p.lda = h->ld[l - 1];
        p.B = h->W + h->w_off[l - 1]; p.ldb = h->ld[l];
        p.M = B_pad; p.N = h->ld[l]; p.K = h->ld[l - 1];
        p.m_true = B; p.n_true = h->dims[l];
        p.act = h->inner_act;
        if (l < h->L - 1) {
            p.C = h->act[l]; p.ldc = h->ld[l];
            launch_gemm<true, false, EPI_ACT>(h, l == 1 ? GNN_K_FWD_GEMM0 : -1, p);
            in = h->act[l];
        } else if (h->dtype == GNN_DTYPE_F32 && p.N <= 32 && p.K >= 128) {
            // narrow logits (10 classes -> one or two 16-column tiles): a tiled GEMM would run a handful of
            // workgroups down the whole K; one 16x16 tile per workgroup with K split over its 8 waves
            // instead (784-1024^3-10 at 256 rows: 15.3 -> ~5 us)
            FwdFirstParams f{};
            f.A = p.A; f.lda = p.lda;
            f.W = p.B; f.ldw = p.ldb;
            f.C = h->logits; f.ldc = h->ld[l];
            f.M = p.M; f.N = p.N; f.K = p.K;
            f.m_true = p.m_true; f.n_true = p.n_true;
            f.act = 0; f.apply_act = 0;
            f.tiling = make_xcd_tiling(f.M / 16, f.N / 16);
            launch_timed(h, -1, fwd_first_kernel<FIRST_NW, false, -1>, dim3(f.tiling.blocks()), dim3(FIRST_NW * 64), 0, f);
        } else {
            p.C = h->logits; p.ldc = h->ld[l];
            launch_gemm<true, false, EPI_STORE>(h, l == 1 ? GNN_K_FWD_GEMM0 : -1, p);
        }
    }
}

void run_output(gnn_mlp *h, const float *y, int B, bool want_prob, bool want_delta, bool want_loss,
                bool want_label) {
    const int Lm = h->L - 1;
    OutParams o{};
    o.Z = h->logits; o.ldz = h->ld[Lm];
    o.Y = y; o.ldy = h->ld[Lm];
    o.prob = want_prob ? h->prob : nullptr; o.ldp = h->ld[Lm];
    o.delta = want_delta ? h->delta[Lm] : nullptr; o.ldd = h->ld[Lm];
    o.loss = want_loss ? h->lossv : nullptr;
    o.label = want_label ? h->labels : nullptr;
    o.B = B; o.B_pad = pad_up(B); o.n_true = h->dims[Lm]; o.n_pad = h->ld[Lm];
    o.out_kind = h->out_kind; o.last_act = h->last_act;
    o.delta_b = (want_delta && h->dtype == GNN_DTYPE_BF16) ? h->deltab[Lm] : nullptr;
    hipLaunchKernelGGL(output_layer_kernel, dim3((o.B_pad + 3) / 4), dim3(256), 0, h->stream, o);
}

// ---- backward (SCE:229-287) + gradient / update -------------------------------------------
// fused_update: G_l is consumed by the SGD epilogue and never written (single GPU);
// otherwise G_l goes to the flat gradient buffer for the caller's all-reduce.
void backward(gnn_mlp *h, const float *a0, int B, bool fused_update, float step_over_b, float momentum,
              bool data_only, bool have_last_delta) {
    const int B_pad = pad_up(B);
    for (int l = h->L - 2; l >= 0; l--) {
        if (l >= 1 && !(have_last_delta && l == h->L - 2)) { // delta_l = (delta_{l+1} . W_l^T) * f'(z_l)   -- before W_l is touched
            GemmParams p{};
            p.A = h->delta[l + 1]; p.lda = h->ld[l + 1];
            p.B = h->W + h->w_off[l]; p.ldb = h->ld[l + 1];
            p.C = h->delta[l]; p.ldc = h->ld[l];
            p.M = B_pad; p.N = h->ld[l]; p.K = h->ld[l + 1];
            p.m_true = B; p.n_true = h->dims[l];
            p.aux = h->act[l]; p.ldaux = h->ld[l];
            p.act = h->inner_act;
            launch_gemm<true, true, EPI_DACT>(h, -1, p);
        }
        if (data_only) continue; // the caller forms every G_l in one grad_update_kernel launch
        GemmParams g{}; // G_l = A_l^T . delta_{l+1}
        g.A = (l == 0) ? a0 : h->act[l]; g.lda = h->ld[l];
        g.B = h->delta[l + 1]; g.ldb = h->ld[l + 1];
        g.ldc = h->ld[l + 1];
        g.M = h->ld[l]; g.N = h->ld[l + 1]; g.K = B_pad;
        g.m_true = h->dims[l]; g.n_true = h->dims[l + 1];
        const int cls = (l == 0) ? GNN_K_GRAD_GEMM0 : -1;
        if (fused_update) {
            g.C = nullptr;
            g.W = h->W + h->w_off[l]; g.V = h->V + h->w_off[l];
            g.step_over_b = step_over_b; g.momentum = momentum;
            launch_gemm<false, false, EPI_SGD>(h, cls, g);
        } else {
            g.C = h->G + h->w_off[l];
            launch_gemm<false, false, EPI_STORE>(h, cls, g);
        }
    }
}

// first layer in one launch: a 16x16 tile per workgroup, K split over the waves
void launch_fwd_first(gnn_mlp *h, const float *a0, int B) {
    const int B_pad = pad_up(B);
    // Blocks of thousands of rows (evaluation over the training set, MT:181-197, walks 60 000 of them in blocks of max_batch): the
    // one-tile-per-workgroup kernel below is built for a FEW 16 x 16 tiles with K split over the waves -- at 4 096 rows it ran
    // the 784 x 300 product at 36 TFLOP/s (53 us of an 80-us block, profiles/r04/inference_kernel_stats_before.csv); the tiled
    // GEMM takes over from first_gemm_rows rows (rows addressed directly only: it does not gather).
    if (h->first_gemm_rows > 0 && B_pad >= h->first_gemm_rows && !h->cur_idx) {
        GemmParams p{};
        p.A = a0; p.lda = h->ld[0];
        p.B = h->W; p.ldb = h->ld[1];
        p.C = h->act[1]; p.ldc = h->ld[1];
        p.M = B_pad; p.N = h->ld[1]; p.K = h->ld[0];
        p.m_true = B; p.n_true = h->dims[1];
        p.act = h->inner_act;
        launch_gemm<true, false, EPI_ACT>(h, GNN_K_FWD_GEMM0, p);
        return;
    }
    // Blocks of a few hundred rows (the trainer's validation pass: 601 rows at MNIST's size, NNT:102-113): 16 x 16 tiles re-read
    // their operand panels 4 FLOP per byte -- 722 workgroups moved 72 MB through L2 -> L1 for 283 MFLOP, 12.6 us
    // (profiles/r04/observed_loop_kernel_stats.csv) -- where the wave-K kernel's 32 x 32 tiles (ragged form: 608 x 304) halve that.
    if (h->first_wavek_rows > 0 && B_pad >= h->first_wavek_rows && !h->cur_idx && h->ld[0] >= 128 && !h->env_wavek_off) {
        GemmParams p{};
        p.A = a0; p.lda = h->ld[0];
        p.B = h->W; p.ldb = h->ld[1];
        p.C = h->act[1]; p.ldc = h->ld[1];
        p.M = B_pad; p.N = h->ld[1]; p.K = h->ld[0];
        p.m_true = B; p.n_true = h->dims[1];
        p.act = h->inner_act;
        if (wavek_fits(p.M, p.N, p.K)) launch_gemm_wavek<true, false, EPI_ACT>(h, GNN_K_FWD_GEMM0, p);
        else launch_gemm_wavek_ragged<true, false, EPI_ACT>(h, GNN_K_FWD_GEMM0, p);
        return;
    }
    FwdFirstParams f{};
    f.A = a0; f.lda = h->ld[0];
    f.W = h->W; f.ldw = h->ld[1];
    f.C = h->act[1]; f.ldc = h->ld[1];
    f.M = B_pad; f.N = h->ld[1]; f.K = h->ld[0];
    f.m_true = B; f.n_true = h->dims[1];
    f.act = h->inner_act; f.apply_act = 1;
    f.row_idx = h->cur_idx;
    f.tiling = make_xcd_tiling(f.M / 16, f.N / 16);
    // activation as a template argument: a runtime switch in the epilogue costs ~1000 cycles of
    // instruction fetch on branch targets (measured 1400-2100 vs 650 cycles)
    // 4 waves when each can keep its whole K share in flight at once (<= 13 chunks of 16: K <= 832),
    // else 8: waves are launched at ~2 100 per us chip-wide, so at this size halving the wave count
    // is worth more than the shorter per-wave chain (4.2 vs 4.6 us at 784x300, B = 128)
    const dim3 fg(f.tiling.blocks());
    if (f.K / 16 <= 4 * 13) {
        const dim3 fb(4 * 64);
        switch (h->inner_act) {
        case 0: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<4, false, 0>, fg, fb, 0, f); break;
        case 1: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<4, false, 1>, fg, fb, 0, f); break;
        case 2: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<4, false, 2>, fg, fb, 0, f); break;
        case 3: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<4, false, 3>, fg, fb, 0, f); break;
        default: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<4, false, 4>, fg, fb, 0, f); break;
        }
    } else {
        const dim3 fb(FIRST_NW * 64);
        switch (h->inner_act) {
        case 0: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<FIRST_NW, false, 0>, fg, fb, 0, f); break;
        case 1: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<FIRST_NW, false, 1>, fg, fb, 0, f); break;
        case 2: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<FIRST_NW, false, 2>, fg, fb, 0, f); break;
        case 3: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<FIRST_NW, false, 3>, fg, fb, 0, f); break;
        default: launch_timed(h, GNN_K_FWD_GEMM0, fwd_first_kernel<FIRST_NW, false, 4>, fg, fb, 0, f); break;
        }
    }
}

void fused_gradient(gnn_mlp *h, const float *a0, int B, bool fused_update, float step_over_b, float momentum) {
    // grids of thousands of 32x32 tiles are bound by L2 traffic: 64x64 tiles halve it
    const bool big = h->grad_tiles > 1024;
    GradParams g = big ? h->grad64 : h->grad;
    g.layer[0].A = a0;
    for (int l = 0; l < g.n_layers; l++) g.layer[l].G = h->G + h->w_off[l];
    g.K = pad_up(B);
    g.row_idx = h->cur_idx; g.k_true = B;
    g.step_over_b = step_over_b; g.momentum = momentum;
    if (big) {
        // <true, true>: interior tiles request the next chunk's operands ahead of this chunk's MFMAs (784-1024^3-10 at 256 rows: 78.3 -> 76.8 us
        // per step in one run, profiles/r03/config5_gradient_prefetch.log)
        if (fused_update) launch_timed(h, GNN_K_GRAD_GEMM0, grad_update64_kernel<true, true>, dim3(h->grad_tiles64), dim3(512), 0, g);
        else launch_timed(h, GNN_K_GRAD_GEMM0, grad_update64_kernel<false>, dim3(h->grad_tiles64), dim3(512), 0, g);
    } else {
        if (fused_update) launch_timed(h, GNN_K_GRAD_GEMM0, grad_update_kernel<true>, dim3(h->grad_tiles), dim3(GRAD_THREADS), 0, g);
        else launch_timed(h, GNN_K_GRAD_GEMM0, grad_update_kernel<false>, dim3(h->grad_tiles), dim3(GRAD_THREADS), 0, g);
    }
}

void launch_tail(gnn_mlp *h, const float *a0, const float *y, int B, bool backward, bool want_prob, bool want_loss, bool want_label) {
    const int Lm = h->L - 1;
    TailParams t{};
    t.A = (Lm == 1) ? a0 : h->act[Lm - 1]; t.lda = h->ld[Lm - 1];
    t.W = h->W + h->w_off[Lm - 1];
    t.Y = y; t.ldy = h->ld[Lm];
    t.prob = want_prob ? h->prob : nullptr;
    t.delta_out = backward ? h->delta[Lm] : nullptr;
    t.loss = want_loss ? h->lossv : nullptr;
    t.label = want_label ? h->labels : nullptr;
    t.delta_prev = (backward && Lm >= 2) ? h->delta[Lm - 1] : nullptr; t.ldp = h->ld[Lm - 1];
    t.K = h->ld[Lm - 1]; t.k_true = h->dims[Lm - 1];
    t.B = B; t.n_true = h->dims[Lm];
    t.act = h->inner_act;
    // column splits of the delta_{L-2} phase: towards ~128 workgroups, at least 8 column tiles (one per wave) per split
    const int row_blocks = pad_up(B) / 16, k16 = t.K / 16;
    int splits = 1;
    if (t.delta_prev) splits = std::max(1, std::min({8, 128 / row_blocks, k16 / 8}));
    if (h->dtype == GNN_DTYPE_BF16) {
        t.delta_out_b = backward ? h->deltab[Lm] : nullptr;
        t.delta_prev_b = (backward && Lm >= 2) ? h->deltab[Lm - 1] : nullptr;
        launch_timed(h, -1, tail_kernel<true>, dim3(row_blocks, splits), dim3(512), 0, t);
        return;
    }
    launch_timed(h, -1, tail_kernel<false>, dim3(row_blocks, splits), dim3(512), 0, t);
}

} // namespace host
} // namespace gnn
