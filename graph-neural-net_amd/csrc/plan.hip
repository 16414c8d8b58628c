// plan.hip -- which kernels a net takes (decided once at gnn_mlp_create) and one gradientStep (SCE:297-346) made of
// them: forward GEMM chain / row-block kernel, output rule, backward data, weight gradient with the momentum update
// fused in (single GPU) or written to the flat gradient buffer (data parallel).  Host logic only: the launches are in
// launch_*.hip.
#include "handle.h"

#include <algorithm>

using namespace gnn;
using namespace gnn::host;

namespace gnn {
namespace host {

// Decides whether the net fits the fused path and lays out the middle kernel's LDS.
void plan_fused(gnn_mlp *h) {
    h->fused = false;
    h->mid4 = false;
    h->mid_generic = false;
    h->plan_note.clear();
    if (h->env_path == 1) { h->plan_note = "per-layer GEMMs forced (GNN_MLP_PATH=generic)"; return; }
    const int L = h->L, Lm = L - 1;
    if (L < 3 || L > MAX_LAYERS) { h->plan_note = "fewer than 3 or more than 8 layers: per-layer GEMMs"; return; }
    if (h->dtype != GNN_DTYPE_F32) {
        // bf16 operands: per-layer GEMMs (gemm_bf16.h) for inference and for nets off the row-block path; training
        // of a net that fits the row-block kernel takes the two-launch path in bf16 (tile_step_bf16_kernel + the bf16
        // instance of rowblock_kernel, or of middle4_kernel where that one does not apply)
        plan_mid4(h);
        if (h->mid4) { plan_chain(h); plan_rowblock(h); }
        if (!h->chain) h->mid4 = false;
        if (!h->chain && h->plan_note.empty()) h->plan_note = "bf16: the row-block kernel exists for the two-launch path only; per-layer bf16 GEMMs";
        return;
    }
    // gradient tiles: every layer's 32x32 tiles in one grid (shared by both middle kernels)
    {
        GradParams &g = h->grad;
        g = GradParams{};
        g.n_layers = L - 1;
        int tiles = 0;
        for (int l = 0; l < L - 1; l++) {
            GradLayer &gl = g.layer[l];
            gl.A = h->act[l]; gl.lda = h->ld[l];
            gl.D = h->delta[l + 1]; gl.ldd = h->ld[l + 1];
            gl.W = h->W + h->w_off[l]; gl.V = h->V + h->w_off[l]; gl.G = h->G + h->w_off[l];
            gl.M = h->ld[l]; gl.N = h->ld[l + 1];
            gl.tiling = make_xcd_tiling((gl.M + 31) / 32, (gl.N + 31) / 32);
            gl.block_begin = tiles;
            tiles += gl.tiling.blocks();
        }
        h->grad_tiles = tiles;
        GradParams &g64 = h->grad64;
        g64 = g;
        int tiles64 = 0;
        for (int l = 0; l < L - 1; l++) {
            GradLayer &gl = g64.layer[l];
            gl.tiling = make_xcd_tiling((gl.M + 63) / 64, (gl.N + 63) / 64);
            gl.block_begin = tiles64;
            tiles64 += gl.tiling.blocks();
        }
        h->grad_tiles64 = tiles64;
    }
    // preferred: 4-row blocks with LDS-resident middle weights
    plan_mid4(h);
    if (h->mid4) { h->fused = true; plan_chain(h); plan_rowblock(h); return; }
    // the middle weights do not fit LDS: per-layer tiled GEMMs for the middle, still bracketed by
    // the one-launch first layer and the one-launch gradient+update (a 16-row kernel that streamed
    // the middle weights from L2 was 15-40 % slower than this on every such shape and was removed)
    h->mid_generic = true;
    h->fused = true;
    if (h->plan_note.empty()) h->plan_note = "the middle weights do not fit one workgroup's LDS: per-layer GEMMs between the one-launch first layer and gradient";
}

// ---- two-launch step: tile_step_kernel plan ----------------------------------------------------
// Every layer's weight matrix in 64 x 16 tiles, one grid; layer 0's tiles first (they also make the next
// batch's first-layer K slabs).  Needs the row-block kernel (middle4) and at most MID4_MAX_SLABS slabs.
void plan_chain(gnn_mlp *h) {
    h->chain = false;
    h->n_slabs = 0;
    h->tsp = TileStepParams{};
    if (h->env_chain_off) { h->plan_note = "two-launch path switched off (GNN_MLP_CHAIN=0)"; return; }
    if (!h->mid4) return; // (plan_mid4 said why)
    const int L = h->L;
    const int n_slabs = (h->ld[0] + TS_TM - 1) / TS_TM;
    if (n_slabs > MID4_MAX_SLABS) { h->plan_note = "more than 16 first-layer K slabs (input width above 1024): three launches per step"; return; }
    const size_t n = (size_t)n_slabs * h->cap_rows * h->ld[1];
    if (n >= (1ull << 30)) { h->plan_note = "slab buffer beyond 2^30 floats (the row-block kernel addresses it with 32-bit byte offsets)"; return; }
    // every allocation of the path, or none: a failure leaves the handle on the three-launch / per-layer path with nothing held
    auto give_up = [&](const char *what) {
        (void)hipGetLastError();
        if (h->slabs) { (void)hipFree(h->slabs); h->slabs = nullptr; }
        if (h->ts_map) { (void)hipFree(h->ts_map); h->ts_map = nullptr; }
        if (h->ts_map0) { (void)hipFree(h->ts_map0); h->ts_map0 = nullptr; }
        for (int i = 0; i < 2; i++) {
            if (h->xstage[i]) { (void)hipFree(h->xstage[i]); h->xstage[i] = nullptr; }
            if (h->xstage_b[i]) { (void)hipFree(h->xstage_b[i]); h->xstage_b[i] = nullptr; }
        }
        h->plan_note = std::string("two-launch path not taken: ") + what;
    };
    if (hipMalloc(reinterpret_cast<void **>(&h->slabs), sizeof(float) * n) != hipSuccess) { h->slabs = nullptr; give_up("hipMalloc of the slab buffer failed"); return; }
    if (hipMemsetAsync(h->slabs, 0, sizeof(float) * n, h->stream) != hipSuccess) { give_up("hipMemsetAsync of the slab buffer failed"); return; }
    for (int i = 0; i < 2; i++) {
        const size_t xn = (size_t)h->cap_rows * h->ld[0];
        if (h->dtype == GNN_DTYPE_BF16) {
            if (hipMalloc(reinterpret_cast<void **>(&h->xstage_b[i]), sizeof(__bf16) * xn) != hipSuccess) { h->xstage_b[i] = nullptr; give_up("hipMalloc of a row staging buffer failed"); return; }
        } else {
            if (hipMalloc(reinterpret_cast<void **>(&h->xstage[i]), sizeof(float) * xn) != hipSuccess) { h->xstage[i] = nullptr; give_up("hipMalloc of a row staging buffer failed"); return; }
        }
    }
    h->n_slabs = n_slabs;
    TileStepParams &t = h->tsp;
    t.n_layers = L - 1;
    int tiles = 0;
    for (int l = 0; l < L - 1; l++) {
        GradLayer &gl = t.layer[l];
        gl.A = h->act[l]; gl.lda = h->ld[l];
        gl.D = h->delta[l + 1]; gl.ldd = h->ld[l + 1];
        gl.W = h->W + h->w_off[l]; gl.V = h->V + h->w_off[l]; gl.G = h->G + h->w_off[l];
        gl.M = h->ld[l]; gl.N = h->ld[l + 1];
        if (h->dtype == GNN_DTYPE_BF16) { t.Ab[l] = h->actb[l]; t.Db[l] = h->deltab[l + 1]; t.Wb[l] = h->Wb + h->w_off[l]; }
        gl.tiling = make_xcd_tiling((gl.M + TS_TM - 1) / TS_TM, gl.N / TS_TN);
        gl.block_begin = tiles;
        tiles += gl.tiling.blocks();
    }
    {
        // workgroup -> tile of the two grids (all layers; layer 0 alone for the chain's first launch)
        TileMapLayer ml[MAX_LAYERS];
        for (int l = 0; l < L - 1; l++) ml[l] = TileMapLayer{h->ld[l], h->ld[l + 1]};
        hipDeviceProp_t prop{};
        int cus = 256;
        if (hipGetDeviceProperties(&prop, h->device) == hipSuccess && prop.multiProcessorCount >= 8) cus = prop.multiProcessorCount;
        else (void)hipGetLastError();
        const std::vector<uint32_t> all = make_tile_map(ml, L - 1, cus / 8, true), first = make_tile_map(ml, 1, cus / 8, true);
        auto upload = [&](const std::vector<uint32_t> &m, uint32_t **dst) {
            if (hipMalloc(reinterpret_cast<void **>(dst), sizeof(uint32_t) * m.size()) != hipSuccess) { *dst = nullptr; return false; }
            return hipMemcpy(*dst, m.data(), sizeof(uint32_t) * m.size(), hipMemcpyHostToDevice) == hipSuccess;
        };
        if (!upload(all, &h->ts_map) || !upload(first, &h->ts_map0)) { give_up("the workgroup -> tile maps could not be placed in device memory"); return; }
        h->ts_tiles = (int)all.size();
        h->ts_tiles0 = (int)first.size();
        h->ts_map_args = pack_tile_map(all, h->ts_map_words[0]) && pack_tile_map(first, h->ts_map_words[1]);
    }
    t.slabs = h->slabs; t.slab_rows = h->cap_rows; t.ldz = h->ld[1];
    h->chain = true;
}

bool slabs_hold(const gnn_mlp *h, const float *a0, const int32_t *idx, int B) {
    return h->slab_valid && h->slab_a0 == a0 && h->slab_idx == idx && h->slab_B == B;
}
// the hint is good for ONE weight update
bool take_next(gnn_mlp *h, NextBatch *nb) {
    if (!h->have_next) return false;
    h->have_next = false;
    *nb = NextBatch{h->next_a0, h->next_idx, h->next_B};
    return true;
}
// staged_copy: the launch that made these slabs also wrote the batch's rows to the other staging buffer
void slabs_now_hold(gnn_mlp *h, const NextBatch &nb, bool staged_copy, bool by_rowblock) {
    if (h->rb && !by_rowblock) staged_copy = false; // (on the row-block kernel's path the tile kernel writes no copy: that kernel does)
    h->slab_valid = true; h->slab_a0 = nb.a0; h->slab_idx = nb.idx; h->slab_B = nb.B;
    if (staged_copy) h->xstage_cur ^= 1;
    h->xstage_valid = staged_copy;
}

// One gradient computation on the two-launch path.  `resident`: the rows live in the dataset (a staging
// buffer holds other data under the same address at the next call, so its slabs are never reused).
void chain_gradient(gnn_mlp *h, const float *a0, const float *y, int B, bool fused_update, float step_over_b, float momentum, bool resident) {
    if (!slabs_hold(h, a0, h->cur_idx, B)) {
        const NextBatch self{a0, h->cur_idx, B};
        launch_tile_step(h, 0, 0, &self, a0, B, 0.f, 0.f); // chain start: the slabs of this batch from the weights as they are
        slabs_now_hold(h, self, self.idx != nullptr);
    }
    // A sampled batch on the row-block kernel's path: that kernel makes the contiguous copies the tile kernel reads with plain
    // addressing.  In a training loop that announces its next batch (train_sampled) it copies the NEXT batch's rows: the tile
    // kernel's first-layer product reads them without an index -> address -> row chain at its top (7.25 -> 5.6 us per launch at
    // 784-300-100-10), and one step later the same copy is the gradient operand.  Without an announced sampled batch of the same
    // size it copies its own rows, for the gradient product only (the first step of a chain gathers that operand by index).
    const bool have_copy = h->xstage_valid && h->cur_idx != nullptr; // (made one step ago, or by the tile launch that made the slabs)
    const bool rb_next = h->rb && fused_update && h->have_next && h->next_idx != nullptr && h->next_B == B;
    const bool rb_cur = h->rb && !rb_next && !have_copy && h->cur_idx != nullptr;
    const bool staged = have_copy || rb_cur;
    h->slab_valid = false;
    fused_forward(h, a0, y, B, true, false, false, false, true, rb_next ? RB_COPY_NEXT : rb_cur ? RB_COPY_CURRENT : RB_COPY_NONE);
    NextBatch nb{};
    if (fused_update) {
        const bool fwd = take_next(h, &nb);
        launch_tile_step(h, 1, 2, fwd ? &nb : nullptr, a0, B, step_over_b, momentum, staged, nullptr, rb_next);
        if (fwd) slabs_now_hold(h, nb, rb_next || nb.idx != nullptr, rb_next);
        else h->xstage_valid = false;
    } else {
        launch_tile_step(h, 1, 1, nullptr, a0, B, 0.f, 0.f, staged);
        if (resident) { // weights unchanged: the slabs (and the staged rows) still describe this batch
            const bool keep = staged;
            slabs_now_hold(h, NextBatch{a0, h->cur_idx, B});
            h->xstage_valid = keep;
        } else {
            h->xstage_valid = false;
        }
    }
}

// Nets whose middle weights exceed LDS: per-layer GEMMs for the middle, and per CALL which of the two
// one-launch kernels still pays. Both trade operand reuse for launch count and occupancy -- a
// 16x16 (first layer) or 32x32 (gradient) tile re-reads its operands from L2 4-16x more often than
// the 64/128-wide GEMM tiles -- so they win while the GEMM grids cannot fill the chip and lose once
// they can (4096-2048-2048-1024 at 512 rows: 692 us with both, 517 us with neither).
struct HybridChoice { bool first, grad; };
HybridChoice hybrid_choice(const gnn_mlp *h, int B) {
    HybridChoice c{true, true};
    if (h->env_hybrid >= 0) { c.first = (h->env_hybrid & 1) != 0; c.grad = (h->env_hybrid & 2) != 0; return c; } // tests/development
    const int B_pad = pad_up(B);
    c.first = pick_tile(B_pad, h->ld[1]) == 32;
    // ... unless the 32 x 32 wave-K GEMM has enough tiles of its own (256 x 784 x 1024: 8.3 us against 13.1 us)
    if (c.first && !h->env_wavek_off && wavek_fits(B_pad, h->ld[1], h->ld[0]) && (B_pad / 32) * (h->ld[1] / 32) >= 192) c.first = false;
    int64_t big = 0, all = 0; // gradient elements in layers whose GEMM grid would fill the chip on its own
    for (int l = 0; l + 1 < h->L; l++) {
        const int64_t e = (int64_t)h->ld[l] * h->ld[l + 1];
        all += e;
        if (pick_tile(h->ld[l], h->ld[l + 1]) == 128) big += e;
    }
    c.grad = big * 2 < all;
    return c;
}

// Nets with at most 16 outputs, off the row-block path: last layer + output rule (+ delta_{L-2}) in one launch
bool use_tail(const gnn_mlp *h) {
    return !h->env_tail_off && h->out_kind == GNN_OUT_SOFTMAX_CE && h->ld[h->L - 1] == 16; // (both dtypes: tail_kernel<BF16>)
}

// bf16 twin of an A_0 row pointer: the staging rows or the resident dataset
const __bf16 *a0_bf16(const gnn_mlp *h, const float *a0) {
    if (a0 == h->act[0]) return h->actb[0];
    return h->DXb + (a0 - h->DX);
}

// the three shapes every entry point is made of
void do_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool want_prob, bool want_loss, bool want_label) {
    if (h->dtype == GNN_DTYPE_BF16) {
        const bool tail = use_tail(h);
        forward_bf16(h, a0_bf16(h, a0), B, tail);
        if (tail) launch_tail(h, a0, y, B, false, want_prob, want_loss, want_label);
        else run_output(h, y, B, want_prob, false, want_loss, want_label);
        return;
    }
    // Blocks of thousands of rows (evaluation over a whole data set, gnn_mlp_count_hits_range): the row-block kernel re-reads every
    // middle weight per FOUR rows and the one-tile first layer is built for a few tiles -- from first_gemm_rows rows on the forward
    // pass is the per-layer GEMM chain (+ the tail kernel), each weight matrix read once per 64-row tile
    // (784-300-100-10, 60 000 rows in blocks of 16 384: 52.9 -> 75 M rows/s with the first layer alone, profiles/r04/inference_sweep.log)
    const bool big_block = h->first_gemm_rows > 0 && pad_up(B) >= h->first_gemm_rows && !h->cur_idx;
    if (h->mid4 && !big_block) { fused_forward(h, a0, y, B, false, want_prob, want_loss, want_label); return; }
    const bool tail = use_tail(h);
    if (h->mid_generic && hybrid_choice(h, B).first) {
        launch_fwd_first(h, a0, B);
        forward(h, a0, B, 2, tail);
    } else {
        forward(h, a0, B, 1, tail);
    }
    if (tail) launch_tail(h, a0, y, B, false, want_prob, want_loss, want_label);
    else run_output(h, y, B, want_prob, false, want_loss, want_label);
}
void do_gradient(gnn_mlp *h, const float *a0, const float *y, int B, bool fused_update, float step_over_b, float momentum,
                 bool resident) {
    if (h->chain) { chain_gradient(h, a0, y, B, fused_update, step_over_b, momentum, resident); return; }
    h->have_next = false;
    if (h->dtype == GNN_DTYPE_BF16) {
        const __bf16 *a0b = a0_bf16(h, a0);
        const bool tail = use_tail(h); // last layer + output rule + delta_{L-2} in one launch (three of the twelve of configs[4])
        forward_bf16(h, a0b, B, tail);
        if (tail) launch_tail(h, a0, y, B, true, false, false, false);
        else run_output(h, y, B, false, true, false, false);
        backward_bf16(h, a0b, B, fused_update, step_over_b, momentum, tail);
        return;
    }
    if (h->mid4) {
        fused_forward(h, a0, y, B, true, false, false, false);
        fused_gradient(h, a0, B, fused_update, step_over_b, momentum);
        return;
    }
    const HybridChoice c = h->mid_generic ? hybrid_choice(h, B) : HybridChoice{false, false};
    const bool tail = use_tail(h);
    if (c.first) {
        launch_fwd_first(h, a0, B);
        forward(h, a0, B, 2, tail);
    } else {
        forward(h, a0, B, 1, tail);
    }
    if (tail) launch_tail(h, a0, y, B, true, false, false, false);
    else run_output(h, y, B, false, true, false, false);
    if (c.grad) {
        backward(h, a0, B, false, 0.f, 0.f, true, tail); // delta_1..delta_{L-2} only (delta_{L-2} came from the tail kernel)
        fused_gradient(h, a0, B, fused_update, step_over_b, momentum);
    } else {
        backward(h, a0, B, fused_update, step_over_b, momentum, false, tail);
    }
}

// A handle that keeps stepping repays the ~0.4 s run-time instantiation (jit.h); never while the
// stream is being captured into a graph (module loading is not a capturable operation).
void maybe_specialize(gnn_mlp *h) {
    if (h->jit_tried || h->specialization != 0 || !h->mid4) return;
    if (++h->steps_seen < 16) return;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) { (void)hipGetLastError(); return; }
    if (st != hipStreamCaptureStatusNone) return;
    try_specialize(h);
}

int step_on_rows(gnn_mlp *h, const float *a0, const float *y, int B, double step, double momentum, bool resident) {
    maybe_specialize(h);
    ScopedTimer tm(h, GNN_K_STEP);
    do_gradient(h, a0, y, B, true, (float)(step / (double)B), (float)momentum, resident);
    h->time++;
    TRY_LAUNCHES(h);
    return GNN_OK;
}

// ---- evaluation workspace (handle.h: EvalWorkspace) -----------------------------------------------------------------------
int eval_block_rows(const gnn_mlp *h, int64_t rows_wanted) {
    int cap = h->max_batch;
    if (h->dtype == GNN_DTYPE_F32 && !h->cur_idx) cap = std::max(cap, h->eval_rows_cap);
    return (int)std::min<int64_t>(cap, rows_wanted);
}
void free_eval_workspace(gnn_mlp *h) {
    gnn_mlp::EvalWorkspace &w = h->evalws;
    for (float *p : w.act) if (p) (void)hipFree(p);
    w.act.clear();
    if (w.logits) (void)hipFree(w.logits);
    if (w.prob) (void)hipFree(w.prob);
    if (w.lossv) (void)hipFree(w.lossv);
    if (w.labels) (void)hipFree(w.labels);
    w.logits = w.prob = w.lossv = nullptr; w.labels = nullptr; w.rows = 0;
}
static int ensure_eval_workspace(gnn_mlp *h, int rows) {
    gnn_mlp::EvalWorkspace &w = h->evalws;
    const int want = pad_up(rows);
    if (w.rows >= want) return GNN_OK;
    HIP_TRY(hipStreamSynchronize(h->stream)); // (a smaller workspace may still be read by a queued pass)
    free_eval_workspace(h);
    w.act.assign((size_t)h->L, nullptr);
    const size_t r = (size_t)want;
    for (int l = 1; l < h->L - 1; l++) TRY(dev_alloc(&w.act[l], r * h->ld[l], h->stream));
    TRY(dev_alloc(&w.logits, r * h->ld[h->L - 1], h->stream));
    TRY(dev_alloc(&w.prob, r * h->ld[h->L - 1], h->stream));
    TRY(dev_alloc(&w.lossv, r, h->stream));
    TRY(dev_alloc(&w.labels, r, h->stream));
    w.rows = want;
    return GNN_OK;
}
EvalScope::EvalScope(gnn_mlp *h_, int rows, int *rc) : h(h_) {
    *rc = GNN_OK;
    if (rows <= h->max_batch) return; // the handle's own buffers hold the block
    *rc = ensure_eval_workspace(h, rows);
    if (*rc != GNN_OK) return;
    swap();
    on = true;
}
EvalScope::~EvalScope() { if (on) swap(); }
void EvalScope::swap() {
    gnn_mlp::EvalWorkspace &w = h->evalws;
    for (int l = 1; l < h->L - 1; l++) std::swap(h->act[l], w.act[l]);
    std::swap(h->logits, w.logits); std::swap(h->prob, w.prob); std::swap(h->lossv, w.lossv); std::swap(h->labels, w.labels);
}

// ---- host batches with the update deferred into the next call (handle.h: PendingUpdate) ----------------------------------
// f32 nets on the two-launch path, outside stream capture (a captured sequence must be self-contained)
bool can_defer_update(const gnn_mlp *h) {
    if (!h->chain || h->dtype != GNN_DTYPE_F32 || !h->act0_alt || h->env_defer_off) return false;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) != hipSuccess) { (void)hipGetLastError(); return false; }
    return st == hipStreamCaptureStatusNone;
}

// The pending step's update by itself (tile_step_kernel<1, 2, false>): its gradient operands are where the row-block kernel
// left them, its A_0 in act[0].
void flush_pending_update(gnn_mlp *h) {
    if (!h->pend.on) return;
    h->pend.on = false;
    h->cur_idx = nullptr;
    launch_tile_step(h, 1, 2, nullptr, h->act[0], h->pend.B, h->pend.step_over_b, h->pend.momentum);
    h->slab_valid = false; h->have_next = false; h->xstage_valid = false;
}

// One gradientStep on the batch staged in act[0] / ybuf: (update of the PENDING step + this batch's first-layer sums) in one
// tile launch -- or the chain's forward-only launch when nothing is pending --, then the row-block kernel; this step's own
// update stays pending.  The arithmetic is train_range's (tile_step_kernel<1, 2, true> between two row-block launches):
// the same weights, bit for bit, as with the update in the call that computed it.
int step_on_host_batch_deferred(gnn_mlp *h, int B, double step, double momentum) {
    maybe_specialize(h);
    const float *a0 = h->act[0];
    h->cur_idx = nullptr;
    h->have_next = false; // (a hint refers to dataset rows; this batch came from the host)
    const NextBatch self{a0, nullptr, B};
    if (h->pend.on) {
        h->pend.on = false;
        launch_tile_step(h, 1, 2, &self, h->act0_alt, h->pend.B, h->pend.step_over_b, h->pend.momentum);
    } else {
        launch_tile_step(h, 0, 0, &self, a0, B, 0.f, 0.f);
    }
    h->slab_valid = false; h->xstage_valid = false; // (a staging buffer holds other rows under the same address at the next call)
    fused_forward(h, a0, h->ybuf, B, true, false, false, false, true, RB_COPY_NONE);
    h->pend.on = true; h->pend.B = B; h->pend.step_over_b = (float)(step / (double)B); h->pend.momentum = (float)momentum;
    h->time++;
    TRY_LAUNCHES(h);
    return GNN_OK;
}

// the next gradient computation runs on dataset rows [row0, row0 + B)
void hint_range(gnn_mlp *h, int64_t row0, int B) {
    h->have_next = true; h->next_a0 = h->DX + (size_t)row0 * h->ld[0]; h->next_idx = nullptr; h->next_B = B;
}

int step_on_device_indices(gnn_mlp *h, const int32_t *d_idx, int B, double step, double momentum) {
    if (h->mid4) {
        // fused path: its three kernels read the sampled rows of the resident dataset through the
        // index vector themselves (two gather launches cost 14 us of a 33-us step)
        h->cur_idx = d_idx;
        const int rc = step_on_rows(h, h->DX, h->DY, B, step, momentum, true);
        h->cur_idx = nullptr;
        return rc;
    }
    launch_gather(h, d_idx, B);
    if (h->dtype == GNN_DTYPE_BF16) to_bf16(h, h->act[0], h->actb[0], (size_t)pad_up(B) * h->ld[0]);
    return step_on_rows(h, h->act[0], h->ybuf, B, step, momentum, false);
}

} // namespace host
} // namespace gnn
