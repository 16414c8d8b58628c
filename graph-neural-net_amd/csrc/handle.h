// handle.h -- internal: the handle behind gnn_mlp_t, the error convention, and the functions the
// translation units of the library call in each other (abi / plan / launch_* / checkpoint / sampler / dp).
// Nothing here is part of the C ABI (include/gnn_mlp.h).
#pragma once
#include "../../include/gnn_mlp.h"
#include "fused_kernels.h"
#include "gemm_bf16.h"
#include "gemm_wavek.h"
#include "middle4_kernel.h"
#include "rowblock_kernel.h"
#include "tile_step_kernel.h"
#include "kernels.h"
#include "eval_kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

struct TimerClass {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
};

struct gnn_mlp {
    int device = 0;
    int L = 0;                 // layerDims.length
    std::vector<int> dims, ld; // logical / padded widths
    std::vector<size_t> w_off; // offset of W_l in the flat padded buffers
    int64_t n_params = 0;      // unpadded
    int64_t n_pad = 0;         // padded flat length
    int out_kind = 0, inner_act = 0, last_act = 0, loss = 0, dtype = 0;
    int max_batch = 0, cap_rows = 0;
    int time = 0;

    float *W = nullptr, *V = nullptr, *G_own = nullptr, *G = nullptr;
    std::vector<float *> act;   // act[l] = f(z_l), l = 0..L-2 ; act[0] = f(x)
    std::vector<float *> delta; // delta[l] = dE/dz_l, l = 1..L-1
    float *logits = nullptr, *prob = nullptr, *ybuf = nullptr, *lossv = nullptr;
    int32_t *labels = nullptr, *idxbuf = nullptr;
    double *stage_out = nullptr;
    // host batches (gnn_mlp_propagate / _loss / _gradient_step ...): pinned f32 slots the calling thread converts the
    // caller's fp64 rows into and the staging kernel reads across PCIe; a slot is rewritten only after the kernel that
    // read it has finished (its event), so call s+1's conversion overlaps step s's kernels (launch_misc.hip)
    static constexpr int kHostSlots = 3;
    float *pin[kHostSlots] = {nullptr, nullptr, nullptr};
    hipEvent_t pin_done[kHostSlots] = {nullptr, nullptr, nullptr};
    bool pin_busy[kHostSlots] = {false, false, false};
    int pin_next = 0;

    float *DX = nullptr, *DY = nullptr; // device-resident dataset (A_0 = f(x) and y)
    // GNN_DTYPE_BF16 (gemm_bf16.h): bf16 roundings of every GEMM operand, written once by its producer
    __bf16 *Wb = nullptr;               // shadow of W, same padded layout
    std::vector<__bf16 *> actb, deltab; // actb[l] l = 0..L-2, deltab[l] l = 1..L-1
    __bf16 *DXb = nullptr;              // dataset inputs
    int64_t dataset_n = 0;

    hipStream_t stream = nullptr, own_stream = nullptr;

    // fused small-net path (fused_kernels.h): plan made once at create
    bool fused = false;
    gnn::GradParams grad{};
    int grad_tiles = 0;
    gnn::GradParams grad64{};      // the same layers cut into 64x64 tiles (grad_update64_kernel), used when grad_tiles is large
    int grad_tiles64 = 0;
    bool mid_generic = false; // middle weights exceed LDS: per-layer GEMMs, fwd_first / grad_update chosen per call (hybrid_choice)
    bool mid4 = false;        // middle4_kernel: every middle weight matrix resident in LDS
    gnn::Mid4Params mid4p{};
    size_t mid4_lds_bytes = 0;
    const void *mid4_fn[3] = {nullptr, nullptr, nullptr}; // forward only / forward + backward / the same with A_1 from K slabs
                                                          // (bf16 nets: only slot 2, the bf16 training kernel)
    hipFunction_t mid4_jit[3] = {nullptr, nullptr, nullptr}; // run-time instantiation (jit.h), preferred when set

    // the TRAINING row-block kernel of the two-launch step (rowblock_kernel.h): f32 nets whose plan fits; else mid4_fn[2]
    bool rb = false;
    gnn::RbParams rbp{};
    size_t rb_lds_bytes = 0;
    const void *rb_fn = nullptr;
    hipFunction_t rb_jit = nullptr; // run-time instantiation (jit.h), preferred when set
    int rb_static = 0;              // 1: rb_fn is a prebuilt static-shape instantiation

    // two-launch step (tile_step_kernel.h): the tile kernel of step s also makes the first-layer K slabs of step s+1
    bool chain = false;
    gnn::TileStepParams tsp{};
    int ts_tiles = 0, ts_tiles0 = 0; // blocks of all layers / of layer 0 alone
    uint32_t *ts_map = nullptr, *ts_map0 = nullptr; // workgroup -> tile of the two grids (make_tile_map), device memory
    bool ts_map_args = false; uint32_t ts_map_words[2][gnn::TS_MAP_ARGS / 2]; // ... and packed for the kernel arguments ([0]: all layers, [1]: layer 0)
    float *slabs = nullptr;
    int n_slabs = 0;
    std::string plan_note;    // why the net is NOT on the two-launch path (empty when it is): gnn_mlp_plan_note
    // the batch whose first-layer sums (for the CURRENT weights) the slabs hold
    bool slab_valid = false; const float *slab_a0 = nullptr; const int32_t *slab_idx = nullptr; int slab_B = 0;
    // the batch the next gradient computation will run on (gnn_mlp_hint_next_range, train loops); consumed by the
    // next kernel that updates the weights
    bool have_next = false; const float *next_a0 = nullptr; const int32_t *next_idx = nullptr; int next_B = 0;
    // contiguous copies of SAMPLED batches (two, used alternately): the tile kernel that forms a sampled batch's slabs also
    // writes the rows it gathered; the next step's gradient product reads them in place of the index-gathered rows
    float *xstage[2] = {nullptr, nullptr}; __bf16 *xstage_b[2] = {nullptr, nullptr};
    int xstage_cur = 0; bool xstage_valid = false; // xstage[xstage_cur] holds the rows of the batch the slabs describe
    int specialization = 0;   // 0 runtime-shape kernels, 1 prebuilt static shape, 2 run-time instantiation
    bool jit_tried = false;
    int steps_seen = 0;       // gradient computations so far: the 16th triggers the specialisation

    // train_range graph: one pass over the dataset's batches captured once, replayed many times
    hipGraphExec_t tr_exec = nullptr;
    hipGraph_t tr_graph = nullptr;
    int64_t tr_first_batch = -1; int tr_B = 0; int64_t tr_nb = 0; double tr_step = 0, tr_mom = 0;
    const float *tr_dx = nullptr;

    // Host batches on the two-launch path (gnn_mlp_gradient_step, the reference's own call shape NNT:83): the UPDATE of a step is
    // deferred into the next call's tile launch, which then also forms the new batch's first-layer sums from the weights it has
    // just written -- three dependent launches per call (staging, tile, row-block) instead of four.  `pend` describes the step
    // whose gradient operands (A_0 in act[0], A_l / delta_l in their buffers) wait for that launch; every other entry point
    // applies it first (check_handle).  act0_alt: the second A_0 buffer (the next batch is staged while the pending one is read).
    struct PendingUpdate { bool on = false; int B = 0; float step_over_b = 0.f, momentum = 0.f; } pend;
    float *act0_alt = nullptr;
    bool env_defer_off = false;  // GNN_MLP_DEFER=0: the update in the call that computed it (four launches; development)

    // Evaluation workspace (f32 nets): forward-only buffers for blocks of MORE rows than max_batch -- evaluation over a data set
    // (gnn_mlp_count_hits_range) and the trainer's validation pass (validate(), NNT:102-113: 601 rows at MNIST's size) then run
    // in blocks of up to kEvalRows rows whatever the handle's max_batch, which is sized for TRAINING batches.  Allocated on first
    // use, grown on demand; EvalScope (plan.hip) swaps the buffers in for the duration of a forward pass.
    static constexpr int kEvalRows = 16384;
    int eval_rows_cap = kEvalRows; // GNN_MLP_EVAL_ROWS (development; 0 = blocks of max_batch as until round 4)
    struct EvalWorkspace {
        int rows = 0;
        std::vector<float *> act;   // act[1..L-2]
        float *logits = nullptr, *prob = nullptr, *lossv = nullptr;
        int32_t *labels = nullptr;
    } evalws;

    // one process per GPU with the exchange inside the library's step loop (gnn_mlp_rccl_*, dp.hip): this rank's communicator
    void *rccl_comm = nullptr; int rccl_ranks = 0, rccl_rank = 0;

    hipError_t launch_error = hipSuccess; // first refused launch of a module / function-pointer kernel since the last check
    const int32_t *cur_idx = nullptr; // device row indices of the batch being stepped (fused path reads rows through them)

    bool timing = false;
    TimerClass timers[5];

    // development / test switches, read ONCE at create (never on the step path)
    int env_path = 0;          // GNN_MLP_PATH: 0 default, 1 "generic", 2 "nomid4"
    int env_hybrid = -1;       // GNN_MLP_HYBRID: -1 unset, else bit 0 = fwd_first, bit 1 = grad_update
    bool env_tail_off = false; // GNN_MLP_TAIL=0: the three-launch form instead of tail_kernel
    bool env_f32_dma_off = false;  // GNN_MLP_F32_DMA=0: gemm_f32_kernel for every shape (development)
    bool env_bf16_group_off = false; // GNN_MLP_BF16_GROUP=0: one launch per layer's gradient (+ update) product (development)
    bool env_bf16_dma_off = false; // GNN_MLP_BF16_DMA=0: the register-staged bf16 GEMM for every shape (development)
    bool env_wavek_off = false; // GNN_MLP_WAVEK=0: gemm_f32_kernel<32, 32> instead of the wave-K kernel (development)
    bool env_graph = false;    // GNN_MLP_GRAPH=1: train_range replays a captured pass
    bool env_jit_off = false;  // GNN_MLP_JIT=0
    bool env_static_off = false; // GNN_MLP_STATIC=0
    bool env_chain_off = false;  // GNN_MLP_CHAIN=0: three launches per step (fwd_first / middle4 / grad_update)
    int first_wavek_rows = 256;  // launch_fwd_first: blocks of at least this many rows take the (ragged) wave-K GEMM (GNN_MLP_FIRST_WAVEK_ROWS; 0 = never)
    int first_gemm_rows = 2048;  // launch_fwd_first: blocks of at least this many rows take gemm_f32_kernel (GNN_MLP_FIRST_GEMM_ROWS; 0 = never)
    bool env_rb_off = false;     // GNN_MLP_ROWBLOCK=0: middle4_kernel<.., SLABS> as the two-launch step's row-block kernel (round 2's form)
};

namespace gnn {
namespace host {

// ---- error convention (abi.hip) ---------------------------------------------------------------
int fail(int code, const std::string &msg);   // records the calling thread's message, returns `code`
const char *last_error_message();
int check_launches(gnn_mlp *h);
int check_handle(gnn_mlp *h, bool apply_pending = true); // (apply_pending: a host-batch step's deferred update runs first, plan.hip)
int check_batch(const gnn_mlp *h, int B);
int check_range(const gnn_mlp *h, int64_t first, int B);
int check_step_args(gnn_mlp *h, int B, double step, int noise);
int get_flat(gnn_mlp *h, const float *dev, double *flat);
int set_flat(gnn_mlp *h, float *dev, const double *flat);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::gnn::host::fail(GNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define TRY_LAUNCHES(h)                              \
    do {                                             \
        int rc_ = ::gnn::host::check_launches(h);    \
        if (rc_ != GNN_OK) return rc_;               \
    } while (0)
#define TRY(expr)                      \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != GNN_OK) return rc_; \
    } while (0)

// Every entry point of the C ABI runs its body through this: the header promises that nothing throws or aborts
// across the boundary (a JVM caller would be taken down by std::terminate), and the bodies use std::vector,
// std::string, std::thread and new.  An exception becomes a status + message like any other failure.
int fail_from_exception(const char *what) noexcept;
template <class F> int guarded(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail_from_exception("out of host memory");
    } catch (const std::exception &e) {
        return fail_from_exception(e.what());
    } catch (...) {
        return fail_from_exception("unknown C++ exception");
    }
}

inline int grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ---- timing ---------------------------------------------------------------------------
struct ScopedTimer {
    gnn_mlp *h; int cls; bool on = false; size_t slot = 0;
    ScopedTimer(gnn_mlp *h_, int c) : h(h_), cls(c) {
        if (!h->timing) return;
        TimerClass &t = h->timers[cls];
        if (t.used >= 8192) return;
        if (t.used >= t.start.size()) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            t.start.push_back(a); t.stop.push_back(b);
        }
        slot = t.used++;
        on = true;
        (void)hipEventRecord(t.start[slot], h->stream);
    }
    ~ScopedTimer() {
        if (on) (void)hipEventRecord(h->timers[cls].stop[slot], h->stream);
    }
};

// Launch with the dispatch's OWN begin/end timestamps (hipExtLaunchKernel start/stop events): the
// elapsed time between them is the kernel's execution time, the same quantity rocprofv3's
// kernel trace reports -- unlike events recorded around a launch, which add marker overhead.
template <class K, class... P> void launch_timed(gnn_mlp *h, int cls, K kernel, dim3 grid, dim3 block, size_t lds, const P &...params) {
    if (h->timing && cls >= 0) {
        TimerClass &t = h->timers[cls];
        if (t.used < 8192) {
            if (t.used >= t.start.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { t.start.push_back(a); t.stop.push_back(b); }
            }
            if (t.used < t.start.size()) {
                const size_t slot = t.used++;
                hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, h->stream, t.start[slot], t.stop[slot], 0, params...);
                return;
            }
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, h->stream, params...);
}

// More than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize -- per kernel AND per device (a handle over N
// devices, dp.hip, launches the same instantiation on each): `done` is the instantiation's own flag array, indexed by the handle's
// device (the handle's device is current when its kernels are launched).
constexpr int kMaxOptInDevices = 64;
template <class K> void opt_in_dynamic_lds(gnn_mlp *h, K kernel, size_t bytes, bool (&done)[kMaxOptInDevices]) {
    const bool tracked = h->device >= 0 && h->device < kMaxOptInDevices;
    if (tracked && done[h->device]) return;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        if (h->launch_error == hipSuccess) h->launch_error = hipGetLastError();
    }
    if (tracked) done[h->device] = true;
}

// Zero-filled device allocation.  The fill is enqueued on the HANDLE's stream: that stream is
// non-blocking, so a legacy-stream hipMemset would not be ordered with the kernels that follow.
template <typename T> int dev_alloc(T **p, size_t n, hipStream_t s) {
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), sizeof(T) * (n ? n : 1)));
    HIP_TRY(hipMemsetAsync(*p, 0, sizeof(T) * (n ? n : 1), s));
    return GNN_OK;
}

// scratch device allocation released on every exit path
struct DevScratch {
    void *p = nullptr;
    ~DevScratch() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 1));
        return GNN_OK;
    }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- launch_gemm.hip: per-layer f32 GEMMs and the three-launch small-net kernels -----------------
int pick_tile(int M, int N);
bool wavek_fits(int M, int N, int K);
void forward(gnn_mlp *h, const float *a0, int B, int first_l = 1, bool stop_before_last = false);
void run_output(gnn_mlp *h, const float *y, int B, bool want_prob, bool want_delta, bool want_loss, bool want_label);
void backward(gnn_mlp *h, const float *a0, int B, bool fused_update, float step_over_b, float momentum,
              bool data_only = false, bool have_last_delta = false);
void launch_fwd_first(gnn_mlp *h, const float *a0, int B);
void fused_gradient(gnn_mlp *h, const float *a0, int B, bool fused_update, float step_over_b, float momentum);
void launch_tail(gnn_mlp *h, const float *a0, const float *y, int B, bool backward, bool want_prob, bool want_loss, bool want_label);

// ---- launch_bf16.hip ---------------------------------------------------------------------------
void forward_bf16(gnn_mlp *h, const __bf16 *a0b, int B, bool stop_before_last = false);
void backward_bf16(gnn_mlp *h, const __bf16 *a0b, int B, bool fused_update, float step_over_b, float momentum, bool have_tail = false);
void to_bf16(gnn_mlp *h, const float *src, __bf16 *dst, size_t n);

// ---- launch_small.hip: the row-block kernel and the tile-owner kernel ----------------------------
struct NextBatch { const float *a0; const int32_t *idx; int B; };
// which rows the row-block kernel copies to a staging buffer while its row tail runs (RbParams::xcopy): none, its own
// sampled batch's (to xstage[xstage_cur]), or the announced NEXT sampled batch's (to the other buffer)
enum { RB_COPY_NONE = 0, RB_COPY_CURRENT = 1, RB_COPY_NEXT = 2 };
struct PeerGradients { const float *G[TS_MAX_PEERS]; int n; unsigned slice; }; // gsrc 3 / 4 of launch_tile_step (dp.hip)
void plan_mid4(gnn_mlp *h);
void plan_rowblock(gnn_mlp *h); // (after plan_chain: the kernel exists for the two-launch step only)
void try_specialize(gnn_mlp *h);
void fused_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool backward, bool want_prob,
                   bool want_loss, bool want_label, bool from_slabs = false, int copy_rows = RB_COPY_NONE);
void launch_tile_step(gnn_mlp *h, int gsrc, int gdst, const NextBatch *next, const float *a0, int B, float step_over_b, float momentum,
                      bool staged = false, const PeerGradients *peers = nullptr, bool next_staged = false);

// ---- sampler.hip ---------------------------------------------------------------------------------
// validate(validation_size) (NNT:102-113) on the device: the summed loss of dataset rows [0, n) into *d_out (fp64, device)
int validation_loss_sum(gnn_mlp *h, int n, double *d_out);
// the same pass with the per-sample losses LEFT in loss_row[0..n) (device; the caller sums rows later): one block, no reduction launch.
// false when the pass needs more than one block (n above the evaluation block size): the caller then takes validation_loss_sum
bool validation_losses_to_row(gnn_mlp *h, int n, float *loss_row, int *rc);

// ---- launch_misc.hip: encodings, gathers, the flat update ---------------------------------------
void launch_convert_rows(gnn_mlp *h, const double *src, int d, float *dst, int ld, int64_t rows, int64_t rows_pad, int act, int apply_act);
void launch_encode_u8(gnn_mlp *h, const uint8_t *pix, int d, float *dst, int ld, int64_t rows, int act);
void launch_onehot_u8(gnn_mlp *h, const uint8_t *lab, int n_classes, float *dst, int ld, int64_t rows);
int stage_batch(gnn_mlp *h, const double *X, const double *Y, int B); // host fp64 rows -> act[0] = f(x), ybuf (Y may be null)
void release_host_staging(gnn_mlp *h);
int export_rows(gnn_mlp *h, const float *src, int ld, int d, int B, double *host_dst);
void launch_gather(gnn_mlp *h, const int32_t *d_idx, int B);
void launch_flat_update(gnn_mlp *h, int B_global, double step, double momentum);

// ---- plan.hip: which kernels a net takes, and one step made of them -------------------------------
void plan_fused(gnn_mlp *h);
void plan_chain(gnn_mlp *h);
const __bf16 *a0_bf16(const gnn_mlp *h, const float *a0);
bool slabs_hold(const gnn_mlp *h, const float *a0, const int32_t *idx, int B);
bool take_next(gnn_mlp *h, NextBatch *nb);
void slabs_now_hold(gnn_mlp *h, const NextBatch &nb, bool staged_copy = false, bool by_rowblock = false);
void hint_range(gnn_mlp *h, int64_t row0, int B);
void do_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool want_prob, bool want_loss, bool want_label);
void do_gradient(gnn_mlp *h, const float *a0, const float *y, int B, bool fused_update, float step_over_b, float momentum,
                 bool resident = false);
void maybe_specialize(gnn_mlp *h);
// rows a forward-only block may have on this handle: max_batch, or kEvalRows through the evaluation workspace (f32)
int eval_block_rows(const gnn_mlp *h, int64_t rows_wanted);
struct EvalScope { // for blocks above max_batch: the evaluation workspace's buffers stand in for act[1..], logits, prob, lossv, labels
    gnn_mlp *h; bool on = false;
    EvalScope(gnn_mlp *h_, int rows, int *rc);
    ~EvalScope();
    void swap();
};
void free_eval_workspace(gnn_mlp *h);
void rccl_detach_handle(gnn_mlp *h); // (dp.hip; gnn_mlp_destroy calls it)
bool can_defer_update(const gnn_mlp *h);
int step_on_host_batch_deferred(gnn_mlp *h, int B, double step, double momentum); // act[0] / ybuf hold the staged batch
void flush_pending_update(gnn_mlp *h);
int step_on_rows(gnn_mlp *h, const float *a0, const float *y, int B, double step, double momentum, bool resident);
int step_on_device_indices(gnn_mlp *h, const int32_t *d_idx, int B, double step, double momentum);

} // namespace host
} // namespace gnn
