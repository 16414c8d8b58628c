// launch_small.hip -- host side of the small-net path: the row-block kernel (middle4_kernel.h, with its run-time
// instantiation, jit.h) and the tile-owner kernel (tile_step_kernel.h).
#include "static_shapes.h"
#include "jit.h"

using namespace gnn;
using namespace gnn::host;

namespace gnn {
namespace host {

// ---- middle4_kernel plan ----------------------------------------------------------------------
// (the prebuilt instances and their tables: static_shapes.h; the GeneralNeuralNet ones live in launch_small_gnn.hip)
template <class SH> bool shape_matches(const gnn_mlp *h) {
    constexpr int n = (int)(sizeof(SH::kDims) / sizeof(int));
    if (h->L != n) return false;
    for (int i = 0; i < n; i++) if (h->dims[i] != SH::kDims[i]) return false;
    return true;
}

// 0 / 1: the net has one of the two prebuilt shapes (either output kind); -1: it does not, or GNN_MLP_STATIC=0
int static_shape_of(const gnn_mlp *h) {
    if (h->env_static_off) return -1;
    if (shape_matches<ShapeMnistA>(h)) return 0;
    if (shape_matches<ShapeMnistB>(h)) return 1;
    return -1;
}

const void *mid4_function(const gnn_mlp *h, int variant) {
    const int which = static_shape_of(h);
    if (which >= 0)
        return h->out_kind == GNN_OUT_SOFTMAX_CE ? mid4_static_table<0>(which, h->inner_act, variant) : mid4_static_general(which, h->inner_act, variant);
    // runtime extents: layer count templated (3..6, else generic), activation read from the arguments
#define GNN_M4RO(NL, OK) (variant == 3 ? reinterpret_cast<const void *>(&middle4_kernel<RuntimeShape<NL>, -1, OK, true, false, true, true>) \
                          : variant == 2 ? reinterpret_cast<const void *>(&middle4_kernel<RuntimeShape<NL>, -1, OK, true, false, true>) \
                          : variant == 1 ? reinterpret_cast<const void *>(&middle4_kernel<RuntimeShape<NL>, -1, OK, true>)           \
                                         : reinterpret_cast<const void *>(&middle4_kernel<RuntimeShape<NL>, -1, OK, false>))
#define GNN_M4R(NL) (h->out_kind == GNN_OUT_SOFTMAX_CE ? GNN_M4RO(NL, 0) : GNN_M4RO(NL, 1))
    switch (h->L) {
    case 3: return GNN_M4R(3);
    case 4: return GNN_M4R(4);
    case 5: return GNN_M4R(5);
    case 6: return GNN_M4R(6);
    default: return GNN_M4R(0);
    }
#undef GNN_M4R
#undef GNN_M4RO
}

void plan_mid4(gnn_mlp *h) {
    h->mid4 = false;
    if (h->env_path == 2) { h->plan_note = "row-block kernel switched off (GNN_MLP_PATH=nomid4)"; return; } // tests: force the per-layer middle
    const int L = h->L, Lm = L - 1;
    Mid4Params &m = h->mid4p;
    m = Mid4Params{};
    const bool bf16 = h->dtype == GNN_DTYPE_BF16;
    m.plan = make_mid4_plan(h->dims.data(), L, bf16);
    if (!m.plan.ok) return; // (plan_fused notes that the middle weights do not fit LDS)
    h->mid4_lds_bytes = (size_t)m.plan.lds_floats * sizeof(float);
    for (int l = 1; l < Lm; l++) { m.W[l] = h->W + h->w_off[l]; m.act[l] = h->act[l]; }
    for (int l = 1; l <= Lm; l++) m.delta[l] = h->delta[l];
    if (bf16) {
        for (int l = 1; l < Lm; l++) { m.Wb[l] = h->Wb + h->w_off[l]; m.actb[l] = h->actb[l]; }
        for (int l = 1; l <= Lm; l++) m.deltab[l] = h->deltab[l];
    }
    m.last_act = h->last_act;
    m.inner_act = h->inner_act;
    h->specialization = static_shape_of(h) >= 0 ? 1 : 0;
    for (int bwd = (bf16 ? 2 : 0); bwd < 3; bwd++) {
        h->mid4_fn[bwd] = mid4_function(h, (bf16 && bwd == 2) ? 3 : bwd);
        if (hipFuncSetAttribute(h->mid4_fn[bwd], hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)h->mid4_lds_bytes) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
    }
    h->mid4 = true;
}

// ---- rowblock_kernel plan -----------------------------------------------------------------------
template <int NL, bool BF> const void *rb_fn_runtime(int out_kind) {
    return out_kind == GNN_OUT_SOFTMAX_CE ? reinterpret_cast<const void *>(&rowblock_kernel<RbRuntimeShape<NL>, -1, 0, false, 0, BF>)
                                          : reinterpret_cast<const void *>(&rowblock_kernel<RbRuntimeShape<NL>, -1, 1, false, 0, BF>);
}

void plan_rowblock(gnn_mlp *h) {
    h->rb = false;
    h->rb_fn = nullptr; h->rb_jit = nullptr; h->rb_static = 0;
    const bool bf = h->dtype == GNN_DTYPE_BF16;
    if (!h->chain || h->env_rb_off) return;
    if (bf && h->L > 4) return; // (the bf16 form covers nets of three and four layers; deeper ones keep middle4_kernel<.., BF16>)
    RbParams &r = h->rbp;
    r = RbParams{};
    r.plan = make_rb_plan(h->dims.data(), h->L);
    if (!r.plan.ok) return; // (the two-launch step then keeps middle4_kernel as its row-block kernel)
    const int L = h->L, Lm = L - 1;
    h->rb_lds_bytes = (size_t)r.plan.lds_floats * sizeof(float);
    for (int l = 1; l < Lm; l++) { r.W[l] = h->W + h->w_off[l]; r.act[l] = h->act[l]; }
    for (int l = 1; l <= Lm; l++) r.delta[l] = h->delta[l];
    if (bf) {
        for (int l = 1; l < Lm; l++) { r.Wb[l] = h->Wb + h->w_off[l]; r.actb[l] = h->actb[l]; }
        for (int l = 1; l <= Lm; l++) r.deltab[l] = h->deltab[l];
    }
    r.last_act = h->last_act;
    r.inner_act = h->inner_act;
    r.slabs = h->slabs; r.slab_rows = h->cap_rows;
    const int which = static_shape_of(h);
    if (which >= 0) {
        h->rb_fn = h->out_kind == GNN_OUT_SOFTMAX_CE ? rb_static_table<0>(which, h->inner_act, bf) : rb_static_general(which, h->inner_act, bf);
        h->rb_static = 1;
    }
    else if (bf) h->rb_fn = L == 3 ? rb_fn_runtime<3, true>(h->out_kind) : rb_fn_runtime<4, true>(h->out_kind);
    else {
        switch (L) {
        case 3: h->rb_fn = rb_fn_runtime<3, false>(h->out_kind); break;
        case 4: h->rb_fn = rb_fn_runtime<4, false>(h->out_kind); break;
        case 5: h->rb_fn = rb_fn_runtime<5, false>(h->out_kind); break;
        case 6: h->rb_fn = rb_fn_runtime<6, false>(h->out_kind); break;
        default: h->rb_fn = rb_fn_runtime<0, false>(h->out_kind); break;
        }
    }
    if (hipFuncSetAttribute(h->rb_fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->rb_lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        h->rb_fn = nullptr;
        return;
    }
    h->rb = true;
}

// Run-time instantiation of middle4_kernel for this net's shape (jit.h); silent no-op when the
// net is already specialised, does not take the middle4 path, or hiprtc is unavailable.
void try_specialize(gnn_mlp *h) {
    if (!h->mid4 || h->specialization != 0 || h->jit_tried) return;
    h->jit_tried = true;
    if (h->env_jit_off) return;
    const jit::Specialised *sp = jit::get_middle4(h->device, h->dims.data(), h->L, h->inner_act, h->out_kind, h->chain,
                                                  h->dtype == GNN_DTYPE_BF16, h->mid4_lds_bytes);
    if (!sp) return;
    h->mid4_jit[0] = sp->fn[0];
    h->mid4_jit[1] = sp->fn[1];
    h->mid4_jit[2] = sp->fn[2];
    h->specialization = 2;
    if (h->rb && !h->rb_static) { // the training row-block kernel for this shape, from the same embedded sources
        const jit::Specialised *rs = jit::get_rowblock(h->device, h->dims.data(), h->L, h->inner_act, h->out_kind, h->dtype == GNN_DTYPE_BF16, h->rb_lds_bytes);
        if (rs) h->rb_jit = rs->fn[0];
    }
}

// forward of the middle4 path; backward = also delta_1..delta_{L-1}
// from_slabs: A_1 = f(sum of the K slabs) (tile_step_kernel made them); else fwd_first_kernel writes act[1] first
void fused_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool backward, bool want_prob,
                   bool want_loss, bool want_label, bool from_slabs, int copy_rows) {
    if (!from_slabs) launch_fwd_first(h, a0, B);
    if (from_slabs && h->rb) { // the two-launch step's training kernel (rowblock_kernel.h)
        RbParams r = h->rbp;
        r.slabs = h->slabs; r.slab_rows = h->cap_rows;
        r.Y = y; r.ldy = h->ld[h->L - 1];
        r.prob = want_prob ? h->prob : nullptr;
        r.loss = want_loss ? h->lossv : nullptr;
        r.label = want_label ? h->labels : nullptr;
        r.B = B;
        r.row_idx = h->cur_idx;
        if (copy_rows == RB_COPY_CURRENT && h->cur_idx) { // the sampled batch's rows, contiguous, for the tile kernel that follows (RbParams::xcopy)
            r.ldx = h->ld[0]; r.copy_idx = h->cur_idx;
            if (h->dtype == GNN_DTYPE_BF16) { r.Xb = a0_bf16(h, a0); r.xcopyb = h->xstage_b[h->xstage_cur]; }
            else { r.X = a0; r.xcopy = h->xstage[h->xstage_cur]; }
        } else if (copy_rows == RB_COPY_NEXT && h->have_next && h->next_idx) { // the announced next batch's rows, to the OTHER buffer
            r.ldx = h->ld[0]; r.copy_idx = h->next_idx; // (next_B == B: chain_gradient)
            if (h->dtype == GNN_DTYPE_BF16) { r.Xb = a0_bf16(h, h->next_a0); r.xcopyb = h->xstage_b[h->xstage_cur ^ 1]; }
            else { r.X = h->next_a0; r.xcopy = h->xstage[h->xstage_cur ^ 1]; }
        }
        // (the head arguments: rowblock_kernel.h, GNN_RB_HEAD_PARAMS -- in this order)
        const bool bf = h->dtype == GNN_DTYPE_BF16; // (the bf16 kernel takes its bf16 shadows' pointers in the two weight slots)
        const float *hd_W1 = bf ? reinterpret_cast<const float *>(r.Wb[1]) : r.W[1];
        const float *hd_Wl = bf ? reinterpret_cast<const float *>(r.Wb[h->L - 2]) : r.W[h->L - 2];
        void *args[] = {&r.slabs, &hd_W1, &hd_Wl, &r.row_idx, &r.Y, &r.copy_idx, &r.B, &r.slab_rows, &r.ldy, &r};
        const unsigned grid = (unsigned)(pad_up(B) / 4);
        TimerClass &tc = h->timers[GNN_K_MIDDLE];
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->timing && tc.used < 8192) {
            if (tc.used >= tc.start.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { tc.start.push_back(a); tc.stop.push_back(b); }
            }
            if (tc.used < tc.start.size()) { ev0 = tc.start[tc.used]; ev1 = tc.stop[tc.used]; tc.used++; }
        }
        hipError_t le;
        if (h->rb_jit) le = hipExtModuleLaunchKernel(h->rb_jit, grid * (unsigned)RB_NT, 1, 1, RB_NT, 1, 1, h->rb_lds_bytes, h->stream, args, nullptr, ev0, ev1, 0);
        else if (ev0) le = hipExtLaunchKernel(const_cast<void *>(h->rb_fn), dim3(grid), dim3(RB_NT), args, h->rb_lds_bytes, h->stream, ev0, ev1, 0);
        else le = hipLaunchKernel(h->rb_fn, dim3(grid), dim3(RB_NT), args, h->rb_lds_bytes, h->stream);
        if (le != hipSuccess && h->launch_error == hipSuccess) h->launch_error = le;
        return;
    }
    {
        Mid4Params m4 = h->mid4p;
        for (int l = 1; l < h->L - 1; l++) m4.act[l] = h->act[l]; // (the evaluation workspace may stand in: plan.hip, EvalScope)
        m4.slabs = h->slabs; m4.slab_rows = h->cap_rows; m4.n_slabs = h->n_slabs;
        m4.Y = y; m4.ldy = h->ld[h->L - 1];
        m4.prob = want_prob ? h->prob : nullptr;
        m4.loss = want_loss ? h->lossv : nullptr;
        m4.label = want_label ? h->labels : nullptr;
        m4.B = B;
        m4.row_idx = h->cur_idx;
        void *args[] = {&m4};
        // every padded row is processed: rows >= B become zeros
        const int bw = from_slabs ? 2 : backward ? 1 : 0;
        const unsigned grid = (unsigned)(pad_up(B) / 4);
        TimerClass &tc = h->timers[GNN_K_MIDDLE];
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (h->timing && tc.used < 8192) {
            if (tc.used >= tc.start.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { tc.start.push_back(a); tc.stop.push_back(b); }
            }
            if (tc.used < tc.start.size()) { ev0 = tc.start[tc.used]; ev1 = tc.stop[tc.used]; tc.used++; }
        }
        hipError_t le;
        if (h->mid4_jit[bw]) { // module function: global size is given in threads
            le = hipExtModuleLaunchKernel(h->mid4_jit[bw], grid * 1024u, 1, 1, 1024, 1, 1, h->mid4_lds_bytes, h->stream, args,
                                          nullptr, ev0, ev1, 0);
        } else if (ev0) {
            le = hipExtLaunchKernel(const_cast<void *>(h->mid4_fn[bw]), dim3(grid), dim3(1024), args, h->mid4_lds_bytes,
                                    h->stream, ev0, ev1, 0);
        } else {
            le = hipLaunchKernel(h->mid4_fn[bw], dim3(grid), dim3(1024), args, h->mid4_lds_bytes, h->stream);
        }
        // a refused launch must not pass for a step: callers read it back through hipGetLastError / launch_error
        if (le != hipSuccess && h->launch_error == hipSuccess) h->launch_error = le;
    }
}

// ---- tile_step_kernel launches ------------------------------------------------------------------
// gsrc / gdst / fwd as in tile_step_kernel.h; fwd_only_layer0: the grid covers layer 0's tiles only
// staged: the current batch's rows come from the contiguous copy xstage[xstage_cur] instead of (a0, cur_idx)
void launch_tile_step(gnn_mlp *h, int gsrc, int gdst, const NextBatch *next, const float *a0, int B, float step_over_b, float momentum,
                      bool staged, const PeerGradients *peers, bool next_staged) {
    TileStepParams t = h->tsp;
    if (peers) {
        for (int r = 0; r < peers->n; r++) t.Gpeer[r] = peers->G[r];
        t.n_peer = peers->n; t.slice = peers->slice; t.Gself = h->G;
    }
    t.layer[0].A = staged ? h->xstage[h->xstage_cur] : a0;
    for (int l = 0; l < t.n_layers; l++) t.layer[l].G = h->G + h->w_off[l];
    t.K = pad_up(B); t.k_true = B;
    t.row_idx = staged ? nullptr : h->cur_idx;
    const int stage_dst = h->xstage_cur ^ 1; // a sampled next batch is copied to the OTHER buffer (this launch may be reading the current one)
    if (next && next->idx && !h->rb) { t.stage_out = h->xstage[stage_dst]; t.stage_out_b = h->xstage_b[stage_dst]; } // (with the row-block kernel on the path IT makes the copy)
    t.step_over_b = step_over_b; t.momentum = momentum;
    const bool fwd = next != nullptr;
    if (fwd) { t.An = next->a0; t.ldan = h->ld[0]; t.next_idx = next->idx; t.next_rows = next->B; t.next_K = pad_up(next->B); }
    if (fwd && next_staged) { t.An = h->xstage[stage_dst]; t.next_idx = nullptr; } // (the row-block kernel in front of this launch copied the rows)
    const bool fwd_only = gsrc == 0;
    const dim3 grid(fwd_only ? h->ts_tiles0 : h->ts_tiles), block(TS_THREADS);
    if (fwd_only) t.n_layers = 1;
    t.tile_map = fwd_only ? h->ts_map0 : h->ts_map;
    t.map_in_args = h->ts_map_args ? 1 : 0;
    if (h->ts_map_args) std::memcpy(t.map_words, h->ts_map_words[fwd_only ? 1 : 0], sizeof(t.map_words));
    const int cls = fwd_only ? GNN_K_FWD_GEMM0 : gsrc >= 2 ? GNN_K_UPDATE : GNN_K_GRAD_GEMM0;
    if (h->dtype == GNN_DTYPE_BF16) {
        if (staged) t.Ab[0] = h->xstage_b[h->xstage_cur];
        else if (a0) t.Ab[0] = a0_bf16(h, a0);
        if (fwd) t.Anb = next_staged ? h->xstage_b[stage_dst] : a0_bf16(h, next->a0);
        if (fwd_only) launch_timed(h, cls, tile_step_bf16_kernel<0, 0, true>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 1) launch_timed(h, cls, tile_step_bf16_kernel<1, 1, false>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_bf16_kernel<1, 2, false>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 2 && fwd) launch_timed(h, cls, tile_step_bf16_kernel<1, 2, true>, grid, block, 0, t);
        else if (gsrc == 2 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_bf16_kernel<2, 2, false>, grid, block, 0, t);
        else if (gsrc == 2) launch_timed(h, cls, tile_step_bf16_kernel<2, 2, true>, grid, block, 0, t);
        else if (gsrc == 3 && !fwd) launch_timed(h, cls, tile_step_bf16_kernel<3, 2, false>, grid, block, 0, t);
        else if (gsrc == 3) launch_timed(h, cls, tile_step_bf16_kernel<3, 2, true>, grid, block, 0, t);
        else if (!fwd) launch_timed(h, cls, tile_step_bf16_kernel<4, 2, false>, grid, block, 0, t);
        else launch_timed(h, cls, tile_step_bf16_kernel<4, 2, true>, grid, block, 0, t);
        return;
    }
    if (fwd_only) launch_timed(h, cls, tile_step_kernel<0, 0, true>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 1) launch_timed(h, cls, tile_step_kernel<1, 1, false>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_kernel<1, 2, false>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 2 && fwd) launch_timed(h, cls, tile_step_kernel<1, 2, true>, grid, block, 0, t);
    else if (gsrc == 2 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_kernel<2, 2, false>, grid, block, 0, t);
    else if (gsrc == 2) launch_timed(h, cls, tile_step_kernel<2, 2, true>, grid, block, 0, t);
    else if (gsrc == 3 && !fwd) launch_timed(h, cls, tile_step_kernel<3, 2, false>, grid, block, 0, t);
    else if (gsrc == 3) launch_timed(h, cls, tile_step_kernel<3, 2, true>, grid, block, 0, t);
    else if (!fwd) launch_timed(h, cls, tile_step_kernel<4, 2, false>, grid, block, 0, t);
    else launch_timed(h, cls, tile_step_kernel<4, 2, true>, grid, block, 0, t);
}

} // namespace host
} // namespace gnn
