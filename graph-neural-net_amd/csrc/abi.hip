// abi.hip -- the C ABI of include/gnn_mlp.h for ONE net on ONE GPU: argument checks, staging of host rows,
// the handle's lifetime.  What a step is made of: plan.hip; kernels: launch_*.hip.
#include "handle.h"

#include <algorithm>

#include <chrono>
#include "java_random.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace gnn;
using namespace gnn::host;

namespace {
thread_local std::string g_last_error;
void read_env(gnn_mlp *h) {
    auto is = [](const char *name, const char *val) { const char *e = getenv(name); return e && !strcmp(e, val); };
    h->env_path = is("GNN_MLP_PATH", "generic") ? 1 : is("GNN_MLP_PATH", "nomid4") ? 2 : 0;
    const char *hy = getenv("GNN_MLP_HYBRID");
    h->env_hybrid = hy ? (atoi(hy) & 3) : -1;
    h->env_tail_off = is("GNN_MLP_TAIL", "0");
    h->env_graph = is("GNN_MLP_GRAPH", "1");
    h->env_jit_off = is("GNN_MLP_JIT", "0");
    h->env_static_off = is("GNN_MLP_STATIC", "0");
    h->env_chain_off = is("GNN_MLP_CHAIN", "0");
    h->env_wavek_off = is("GNN_MLP_WAVEK", "0");
    h->env_bf16_dma_off = is("GNN_MLP_BF16_DMA", "0");
    h->env_f32_dma_off = is("GNN_MLP_F32_DMA", "0");
    h->env_bf16_group_off = is("GNN_MLP_BF16_GROUP", "0");
    h->env_rb_off = is("GNN_MLP_ROWBLOCK", "0");
    h->env_defer_off = is("GNN_MLP_DEFER", "0");
    const char *fg = getenv("GNN_MLP_FIRST_GEMM_ROWS"); // development: from how many rows on the first layer runs as a tiled GEMM (0 = never)
    if (fg) h->first_gemm_rows = atoi(fg);
    const char *fw = getenv("GNN_MLP_FIRST_WAVEK_ROWS");
    if (fw) h->first_wavek_rows = atoi(fw);
    const char *er = getenv("GNN_MLP_EVAL_ROWS");
    if (er) h->eval_rows_cap = atoi(er);
}
} // namespace

namespace gnn {
namespace host {

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}
const char *last_error_message() { return g_last_error.c_str(); }
int fail_from_exception(const char *what) noexcept {
    try {
        g_last_error = std::string("C++ exception inside the library: ") + (what ? what : "");
    } catch (...) { // not even the message could be stored: keep the status
        g_last_error.clear();
    }
    return GNN_ERR_STATE;
}

// every launch since the last check was accepted: the runtime's sticky error and the return codes of
// the kernels launched through function pointers / hiprtc modules (fused_forward)
int check_launches(gnn_mlp *h) {
    const hipError_t sticky = hipGetLastError();
    const hipError_t mine = h->launch_error;
    h->launch_error = hipSuccess;
    if (mine != hipSuccess) return fail(GNN_ERR_HIP, std::string("kernel launch refused: ") + hipGetErrorString(mine));
    if (sticky != hipSuccess) return fail(GNN_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(sticky));
    return GNN_OK;
}

int check_handle(gnn_mlp *h, bool apply_pending) {
    if (!h) return fail(GNN_ERR_BAD_ARG, "null handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(GNN_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    // a host-batch step whose update waits for the next call's tile launch: any OTHER entry point sees the updated weights
    if (apply_pending && h->pend.on) flush_pending_update(h);
    return GNN_OK;
}

int check_batch(const gnn_mlp *h, int B) {
    if (B <= 0) return fail(GNN_ERR_BAD_ARG, "batch must be non-empty (reference: assert !batch.isEmpty(), SCE:300)");
    if (B > h->max_batch) return fail(GNN_ERR_BAD_ARG, "B exceeds max_batch given to gnn_mlp_create");
    return GNN_OK;
}

int check_range(const gnn_mlp *h, int64_t first, int B) {
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    if (first < 0 || first + B > h->dataset_n) return fail(GNN_ERR_BAD_ARG, "dataset rows out of range");
    return GNN_OK;
}

// padded f32 flat <-> unpadded fp64 flat
void pack_params(const gnn_mlp *h, const double *flat, std::vector<float> &out) {
    out.assign((size_t)h->n_pad, 0.f);
    size_t src = 0;
    for (int l = 0; l < h->L - 1; l++) {
        const int rows = h->dims[l], cols = h->dims[l + 1], ldc = h->ld[l + 1];
        float *dst = out.data() + h->w_off[l];
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) dst[(size_t)i * ldc + j] = (float)flat[src++];
    }
}
void unpack_params(const gnn_mlp *h, const std::vector<float> &in, double *flat) {
    size_t dst = 0;
    for (int l = 0; l < h->L - 1; l++) {
        const int rows = h->dims[l], cols = h->dims[l + 1], ldc = h->ld[l + 1];
        const float *src = in.data() + h->w_off[l];
        for (int i = 0; i < rows; i++)
            for (int j = 0; j < cols; j++) flat[dst++] = (double)src[(size_t)i * ldc + j];
    }
}

int get_flat(gnn_mlp *h, const float *dev, double *flat) {
    if (!flat) return fail(GNN_ERR_BAD_ARG, "null output");
    std::vector<float> tmp((size_t)h->n_pad);
    HIP_TRY(hipMemcpyAsync(tmp.data(), dev, sizeof(float) * (size_t)h->n_pad, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    unpack_params(h, tmp, flat);
    return GNN_OK;
}
int set_flat(gnn_mlp *h, float *dev, const double *flat) {
    if (!flat) return fail(GNN_ERR_BAD_ARG, "null input");
    if (dev == h->W) h->slab_valid = false; // first-layer sums made with the old weights
    std::vector<float> tmp;
    pack_params(h, flat, tmp);
    HIP_TRY(hipMemcpyAsync(dev, tmp.data(), sizeof(float) * (size_t)h->n_pad, hipMemcpyHostToDevice, h->stream));
    if (dev == h->W && h->Wb) to_bf16(h, h->W, h->Wb, (size_t)h->n_pad); // the bf16 shadow follows the masters
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GNN_OK;
}

int check_step_args(gnn_mlp *h, int B, double step, int noise) {
    TRY(check_batch(h, B));
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    return GNN_OK;
}

} // namespace host
} // namespace gnn

extern "C" {

const char *gnn_mlp_last_error(void) { return last_error_message(); }

int gnn_mlp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act, int loss,
                   int64_t seed, int dtype, int device, int max_batch, gnn_mlp_t **out) { return guarded([&]() -> int {
    if (!out) return fail(GNN_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!dims || n_dims < 2) return fail(GNN_ERR_BAD_ARG, "layerDims must hold at least 2 entries (SCE:105)");
    for (int i = 0; i < n_dims; i++)
        if (dims[i] <= 0) return fail(GNN_ERR_BAD_ARG, "layer dimensions must be positive (SCE:142)");
    if (out_kind != GNN_OUT_SOFTMAX_CE && out_kind != GNN_OUT_ACT_LOSS) return fail(GNN_ERR_BAD_ARG, "bad out_kind");
    if (inner_act < 0 || inner_act > GNN_ACT_IDENTITY) return fail(GNN_ERR_BAD_ARG, "bad inner_act");
    if (out_kind == GNN_OUT_ACT_LOSS && (last_act < 0 || last_act > GNN_ACT_IDENTITY))
        return fail(GNN_ERR_BAD_ARG, "bad last_act");
    if (out_kind == GNN_OUT_ACT_LOSS && loss != GNN_LOSS_HALF_SQUARED) return fail(GNN_ERR_BAD_ARG, "bad loss");
    if (dtype != GNN_DTYPE_F32 && dtype != GNN_DTYPE_BF16) return fail(GNN_ERR_BAD_ARG, "bad dtype");
    if (max_batch <= 0) return fail(GNN_ERR_BAD_ARG, "max_batch must be positive");

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GNN_ERR_NO_DEVICE, "no HIP device visible: this library has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(GNN_ERR_BAD_ARG, "device index out of range");
    HIP_TRY(hipSetDevice(device));

    gnn_mlp *h = new gnn_mlp();
    h->device = device;
    h->L = n_dims;
    h->dims.assign(dims, dims + n_dims);
    h->ld.resize(n_dims);
    for (int i = 0; i < n_dims; i++) h->ld[i] = pad_up(dims[i]);
    h->out_kind = out_kind; h->inner_act = inner_act; h->last_act = last_act; h->loss = loss; h->dtype = dtype;
    h->max_batch = max_batch;
    h->cap_rows = pad_up(max_batch);
    read_env(h);
    h->w_off.resize(n_dims - 1);
    size_t off = 0;
    for (int l = 0; l < n_dims - 1; l++) {
        h->w_off[l] = off;
        off += (size_t)h->ld[l] * h->ld[l + 1];
        h->n_params += (int64_t)dims[l] * dims[l + 1];
    }
    h->n_pad = (int64_t)off;

    auto cleanup = [&](int rc) { gnn_mlp_destroy(h); return rc; };
#define CTRY(expr) do { int rc_ = (expr); if (rc_ != GNN_OK) return cleanup(rc_); } while (0)
    {
        hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return cleanup(fail(GNN_ERR_HIP, std::string("hipStreamCreate: ") + hipGetErrorString(e)));
        h->stream = h->own_stream;
    }
    CTRY(dev_alloc(&h->W, (size_t)h->n_pad, h->stream));
    CTRY(dev_alloc(&h->V, (size_t)h->n_pad, h->stream));
    CTRY(dev_alloc(&h->G_own, (size_t)h->n_pad, h->stream));
    h->G = h->G_own;
    h->act.assign(n_dims, nullptr);
    h->delta.assign(n_dims, nullptr);
    const size_t rows = (size_t)h->cap_rows;
    for (int l = 0; l < n_dims - 1; l++) CTRY(dev_alloc(&h->act[l], rows * h->ld[l], h->stream));
    for (int l = 1; l < n_dims; l++) CTRY(dev_alloc(&h->delta[l], rows * h->ld[l], h->stream));
    if (dtype == GNN_DTYPE_F32) CTRY(dev_alloc(&h->act0_alt, rows * h->ld[0], h->stream)); // (host batches with a deferred update: handle.h)
    const int ldo = h->ld[n_dims - 1];
    CTRY(dev_alloc(&h->logits, rows * ldo, h->stream));
    CTRY(dev_alloc(&h->prob, rows * ldo, h->stream));
    CTRY(dev_alloc(&h->ybuf, rows * ldo, h->stream));
    CTRY(dev_alloc(&h->lossv, rows, h->stream));
    CTRY(dev_alloc(&h->labels, rows, h->stream));
    CTRY(dev_alloc(&h->idxbuf, rows, h->stream));
    if (dtype == GNN_DTYPE_BF16) {
        CTRY(dev_alloc(&h->Wb, (size_t)h->n_pad, h->stream));
        h->actb.assign(n_dims, nullptr);
        h->deltab.assign(n_dims, nullptr);
        for (int l = 0; l < n_dims - 1; l++) CTRY(dev_alloc(&h->actb[l], rows * h->ld[l], h->stream));
        for (int l = 1; l < n_dims; l++) CTRY(dev_alloc(&h->deltab[l], rows * h->ld[l], h->stream));
    }
    {
        size_t so = (size_t)max_batch * dims[n_dims - 1];
        if (so < (size_t)max_batch) so = (size_t)max_batch;
        CTRY(dev_alloc(&h->stage_out, so, h->stream));
    }
#undef CTRY

    // appendLayer (SCE:139-156): Random(seed), layer by layer, row-major, nextDouble() - 0.5
    {
        JavaRandom rnd(seed);
        std::vector<double> flat((size_t)h->n_params);
        size_t k = 0;
        for (int l = 1; l < n_dims; l++)
            for (int i = 0; i < dims[l - 1]; i++)
                for (int j = 0; j < dims[l]; j++) flat[k++] = rnd.next_double() - 0.5;
        int rc = set_flat(h, h->W, flat.data());
        if (rc != GNN_OK) return cleanup(rc);
    }
    plan_fused(h);
    *out = h;
    return GNN_OK;
}); }

int gnn_mlp_destroy(gnn_mlp_t *h) { return guarded([&]() -> int {
    if (!h) return GNN_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    rccl_detach_handle(h);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(h->W); fr(h->V); fr(h->G_own); fr(h->act0_alt); free_eval_workspace(h);
    for (float *p : h->act) fr(p);
    for (float *p : h->delta) fr(p);
    fr(h->logits); fr(h->prob); fr(h->ybuf); fr(h->lossv); fr(h->labels); fr(h->idxbuf);
    fr(h->stage_out); release_host_staging(h); fr(h->DX); fr(h->DY); fr(h->slabs); fr(h->ts_map); fr(h->ts_map0);
    fr(h->Wb); fr(h->DXb);
    for (int i = 0; i < 2; i++) { fr(h->xstage[i]); fr(h->xstage_b[i]); }
    for (__bf16 *p : h->actb) fr(p);
    for (__bf16 *p : h->deltab) fr(p);
    if (h->tr_exec) (void)hipGraphExecDestroy(h->tr_exec);
    if (h->tr_graph) (void)hipGraphDestroy(h->tr_graph);
    for (TimerClass &t : h->timers) {
        for (hipEvent_t e : t.start) (void)hipEventDestroy(e);
        for (hipEvent_t e : t.stop) (void)hipEventDestroy(e);
    }
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
    return GNN_OK;
}); }

int gnn_mlp_input_dim(const gnn_mlp_t *h) { return h ? h->dims[0] : -1; }
int gnn_mlp_output_dim(const gnn_mlp_t *h) { return h ? h->dims[h->L - 1] : -1; }
int64_t gnn_mlp_num_params(const gnn_mlp_t *h) { return h ? h->n_params : -1; }
int gnn_mlp_time(const gnn_mlp_t *h) { return h ? h->time : -1; }
int64_t gnn_mlp_dataset_size(const gnn_mlp_t *h) { return h ? h->dataset_n : -1; }
int64_t gnn_mlp_grad_elems(const gnn_mlp_t *h) { return h ? h->n_pad : -1; }

int gnn_mlp_propagate(gnn_mlp_t *h, const double *X, int B, double *out) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!X || !out) return fail(GNN_ERR_BAD_ARG, "null argument (reference: assert input != null, SCE:165)");
    TRY(check_batch(h, B));
    TRY(stage_batch(h, X, nullptr, B));
    do_forward(h, h->act[0], nullptr, B, true, false, false);
    TRY_LAUNCHES(h);
    return export_rows(h, h->prob, h->ld[h->L - 1], h->dims[h->L - 1], B, out);
}); }

static int read_loss(gnn_mlp *h, int B, double *loss_per_sample) {
    std::vector<float> tmp((size_t)B);
    HIP_TRY(hipMemcpyAsync(tmp.data(), h->lossv, sizeof(float) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < B; i++) loss_per_sample[i] = (double)tmp[i];
    return GNN_OK;
}
static int read_labels(gnn_mlp *h, int B, int32_t *labels) {
    HIP_TRY(hipMemcpyAsync(labels, h->labels, sizeof(int32_t) * (size_t)B, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GNN_OK;
}

int gnn_mlp_loss(gnn_mlp_t *h, const double *X, const double *Y, int B, double *loss_per_sample) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!X || !Y || !loss_per_sample) return fail(GNN_ERR_BAD_ARG, "null argument (SCE:209)");
    TRY(check_batch(h, B));
    TRY(stage_batch(h, X, Y, B));
    do_forward(h, h->act[0], h->ybuf, B, false, true, false);
    TRY_LAUNCHES(h);
    return read_loss(h, B, loss_per_sample);
}); }

int gnn_mlp_argmax(gnn_mlp_t *h, const double *X, int B, int32_t *labels) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!X || !labels) return fail(GNN_ERR_BAD_ARG, "null argument");
    TRY(check_batch(h, B));
    TRY(stage_batch(h, X, nullptr, B));
    do_forward(h, h->act[0], nullptr, B, false, false, true);
    TRY_LAUNCHES(h);
    return read_labels(h, B, labels);
}); }

int gnn_mlp_compute_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!X || !Y) return fail(GNN_ERR_BAD_ARG, "null argument (SCE:231)");
    TRY(check_batch(h, B));
    TRY(stage_batch(h, X, Y, B));
    do_gradient(h, h->act[0], h->ybuf, B, false, 0.f, 0.f, false);
    TRY_LAUNCHES(h);
    return GNN_OK;
}); }

int gnn_mlp_weight_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B, double *flat_grad) { return guarded([&]() -> int {
    if (!flat_grad) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(gnn_mlp_compute_gradient(h, X, Y, B));
    return get_flat(h, h->G, flat_grad);
}); }

int gnn_mlp_gradient_step(gnn_mlp_t *h, const double *X, const double *Y, int B, double step, double momentum,
                          int noise) { return guarded([&]() -> int {
    TRY(check_handle(h, false)); // (a pending update is taken up by THIS call's tile launch, below)
    int rc = GNN_OK;
    if (!X || !Y) rc = fail(GNN_ERR_BAD_ARG, "null argument (reference: assert batch != null, SCE:299)");
    if (rc == GNN_OK) rc = check_batch(h, B);
    if (rc == GNN_OK && noise) rc = fail(GNN_ERR_UNSUPPORTED, "noise=true is NaN-producing in the reference (SCE:335 sqrt of a negative draw) and is not built on the GPU");
    if (rc == GNN_OK && !(step > 0)) rc = fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    if (rc != GNN_OK) { flush_pending_update(h); return rc; } // (a refused call leaves nothing pending either)
    if (can_defer_update(h)) {
        // the new batch goes to the OTHER A_0 buffer: the pending step's gradient still reads its own
        if (h->pend.on) std::swap(h->act[0], h->act0_alt);
        rc = stage_batch(h, X, Y, B);   // the caller's rows are not read after this returns
        if (rc != GNN_OK) { if (h->pend.on) std::swap(h->act[0], h->act0_alt); flush_pending_update(h); return rc; }
        return step_on_host_batch_deferred(h, B, step, momentum);
    }
    TRY(stage_batch(h, X, Y, B));   // the caller's rows are not read after this returns
    h->have_next = false; // (a hint refers to dataset rows; this batch came from the host)
    return step_on_rows(h, h->act[0], h->ybuf, B, step, momentum, false);
}); }

int gnn_mlp_get_weights(gnn_mlp_t *h, double *flat) { return guarded([&]() -> int { TRY(check_handle(h)); return get_flat(h, h->W, flat); }); }
int gnn_mlp_set_weights(gnn_mlp_t *h, const double *flat) { return guarded([&]() -> int { TRY(check_handle(h)); return set_flat(h, h->W, flat); }); }
int gnn_mlp_get_momentum(gnn_mlp_t *h, double *flat) { return guarded([&]() -> int { TRY(check_handle(h)); return get_flat(h, h->V, flat); }); }
int gnn_mlp_set_momentum(gnn_mlp_t *h, const double *flat) { return guarded([&]() -> int { TRY(check_handle(h)); return set_flat(h, h->V, flat); }); }

// ---- dataset ------------------------------------------------------------------------------
static int alloc_dataset(gnn_mlp *h, int64_t N) { return guarded([&]() -> int {
    HIP_TRY(hipStreamSynchronize(h->stream)); // nothing in flight may still read the old dataset
    if (h->DX) { (void)hipFree(h->DX); h->DX = nullptr; }
    if (h->DY) { (void)hipFree(h->DY); h->DY = nullptr; }
    if (h->DXb) { (void)hipFree(h->DXb); h->DXb = nullptr; }
    h->dataset_n = 0;
    h->slab_valid = false; h->have_next = false; // they name rows of the old dataset
    const size_t rows = (size_t)N + PAD; // PAD zero rows behind the last sample: a batch's padding rows read them
    TRY(dev_alloc(&h->DX, rows * h->ld[0], h->stream));
    TRY(dev_alloc(&h->DY, rows * h->ld[h->L - 1], h->stream));
    if (h->dtype == GNN_DTYPE_BF16) TRY(dev_alloc(&h->DXb, rows * h->ld[0], h->stream));
    return GNN_OK;
}); }

int gnn_mlp_upload_dataset(gnn_mlp_t *h, const double *X, const double *Y, int64_t N) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!X || !Y || N <= 0) return fail(GNN_ERR_BAD_ARG, "bad dataset");
    TRY(alloc_dataset(h, N));
    const int d0 = h->dims[0], dl = h->dims[h->L - 1];
    const int64_t chunk = 4096;
    DevScratch bx, by;
    TRY(bx.alloc(sizeof(double) * chunk * d0));
    TRY(by.alloc(sizeof(double) * chunk * dl));
    double *sx = bx.as<double>(), *sy = by.as<double>();
    hipError_t err = hipSuccess;
    for (int64_t r0 = 0; r0 < N && err == hipSuccess; r0 += chunk) {
        const int64_t n = (N - r0 < chunk) ? N - r0 : chunk;
        err = hipMemcpyAsync(sx, X + r0 * d0, sizeof(double) * n * d0, hipMemcpyHostToDevice, h->stream);
        if (err == hipSuccess)
            err = hipMemcpyAsync(sy, Y + r0 * dl, sizeof(double) * n * dl, hipMemcpyHostToDevice, h->stream);
        if (err != hipSuccess) break;
        launch_convert_rows(h, sx, d0, h->DX + (size_t)r0 * h->ld[0], h->ld[0], n, n, h->inner_act, 1);
        launch_convert_rows(h, sy, dl, h->DY + (size_t)r0 * h->ld[h->L - 1], h->ld[h->L - 1], n, n, 0, 0);
        err = hipStreamSynchronize(h->stream); // the staging buffers are reused by the next chunk
    }
    if (err != hipSuccess) return fail(GNN_ERR_HIP, std::string("dataset upload: ") + hipGetErrorString(err));
    if (h->DXb) { to_bf16(h, h->DX, h->DXb, ((size_t)N + PAD) * h->ld[0]); HIP_TRY(hipStreamSynchronize(h->stream)); }
    TRY_LAUNCHES(h);
    h->dataset_n = N;
    return GNN_OK;
}); }

int gnn_mlp_upload_dataset_u8(gnn_mlp_t *h, const uint8_t *pixels, const uint8_t *labels, int64_t N) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!pixels || !labels || N <= 0) return fail(GNN_ERR_BAD_ARG, "bad dataset");
    TRY(alloc_dataset(h, N));
    const int d0 = h->dims[0], dl = h->dims[h->L - 1];
    DevScratch bp, bl;
    TRY(bp.alloc((size_t)N * d0));
    TRY(bl.alloc((size_t)N));
    uint8_t *sp = bp.as<uint8_t>(), *sl = bl.as<uint8_t>();
    HIP_TRY(hipMemcpyAsync(sp, pixels, (size_t)N * d0, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(sl, labels, (size_t)N, hipMemcpyHostToDevice, h->stream));
    launch_encode_u8(h, sp, d0, h->DX, h->ld[0], N, h->inner_act);
    launch_onehot_u8(h, sl, dl, h->DY, h->ld[h->L - 1], N);
    HIP_TRY(hipStreamSynchronize(h->stream)); // the scratch buffers are released when this function returns
    if (h->DXb) { to_bf16(h, h->DX, h->DXb, ((size_t)N + PAD) * h->ld[0]); HIP_TRY(hipStreamSynchronize(h->stream)); }
    TRY_LAUNCHES(h);
    h->dataset_n = N;
    return GNN_OK;
}); }


int gnn_mlp_gradient_step_range(gnn_mlp_t *h, int64_t first, int B, double step, double momentum, int noise) { return guarded([&]() -> int {
    TRY(check_handle(h));
    TRY(check_step_args(h, B, step, noise));
    TRY(check_range(h, first, B));
    return step_on_rows(h, h->DX + (size_t)first * h->ld[0], h->DY + (size_t)first * h->ld[h->L - 1], B, step,
                        momentum, true);
}); }


int gnn_mlp_train_range(gnn_mlp_t *h, int64_t first, int B, int n_steps, double step, double momentum) { return guarded([&]() -> int {
    TRY(check_handle(h));
    TRY(check_step_args(h, B, step, 0));
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    if (n_steps <= 0) return fail(GNN_ERR_BAD_ARG, "n_steps must be positive (NNT:62)");
    const int64_t nb = h->dataset_n / B;
    if (nb <= 0 || first < 0 || first % B != 0) return fail(GNN_ERR_BAD_ARG, "first must be a multiple of B inside the dataset");
    if (n_steps >= 64) try_specialize(h); // a long run repays the ~0.4 s instantiation
    int s = 0;
    // hipGraph replay (opt-in, GNN_MLP_GRAPH=1): when the request covers whole passes over the nb
    // batches, one pass (nb steps, 3 launches each on the fused path) is captured ONCE from this
    // very stream and replayed with a single launch per pass; the remainder runs eagerly.  It is
    // off by default because it buys nothing on one GPU (22.38 vs 22.32 us/step: the kernels
    // already run back to back) while the capture costs a few ms on the first call.  Never while
    // per-kernel timing is on (timed launches carry events) or on a caller-provided stream that is
    // itself being captured.
    bool caller_capturing = false; // (a caller-provided stream may itself be under capture: never nest)
    if (h->env_graph && h->stream != h->own_stream) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        caller_capturing = hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
        if (caller_capturing) (void)hipGetLastError();
    }
    const bool want_graph = h->env_graph && !h->timing && !caller_capturing &&
                            nb >= 2 && nb <= 1024 && n_steps >= 2 * nb;
    if (want_graph) {
        const int64_t fb = (first / B) % nb;
        const bool hit = h->tr_exec && h->tr_first_batch == fb && h->tr_B == B && h->tr_nb == nb &&
                         h->tr_step == step && h->tr_mom == momentum && h->tr_dx == h->DX;
        if (!hit) {
            if (h->tr_exec) { (void)hipGraphExecDestroy(h->tr_exec); h->tr_exec = nullptr; }
            if (h->tr_graph) { (void)hipGraphDestroy(h->tr_graph); h->tr_graph = nullptr; }
            if (h->chain) { // two-launch path: the pass is captured as a closed chain -- every step, the last one too, also
                            // makes the first-layer slabs of the batch after it -- so its first batch's slabs must exist before
                const float *a0 = h->DX + (size_t)(fb * B) * h->ld[0];
                if (!slabs_hold(h, a0, nullptr, B)) {
                    const NextBatch self{a0, nullptr, B};
                    launch_tile_step(h, 0, 0, &self, a0, B, 0.f, 0.f);
                    slabs_now_hold(h, self, false);
                }
            }
            if (hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                const int t0 = h->time;
                int rc = GNN_OK;
                for (int64_t b = 0; b < nb && rc == GNN_OK; b++) {
                    const int64_t row0 = ((fb + b) % nb) * B;
                    hint_range(h, ((fb + b + 1) % nb) * B, B);
                    rc = step_on_rows(h, h->DX + (size_t)row0 * h->ld[0], h->DY + (size_t)row0 * h->ld[h->L - 1], B, step, momentum, true);
                }
                h->time = t0; // captured, not executed (the slabs of batch fb, made above, are still the current ones)
                hipGraph_t g = nullptr;
                const hipError_t e = hipStreamEndCapture(h->stream, &g);
                if (rc == GNN_OK && e == hipSuccess && g && hipGraphInstantiate(&h->tr_exec, g, nullptr, nullptr, 0) == hipSuccess) {
                    h->tr_graph = g;
                    h->tr_first_batch = fb; h->tr_B = B; h->tr_nb = nb; h->tr_step = step; h->tr_mom = momentum; h->tr_dx = h->DX;
                } else {
                    if (g) (void)hipGraphDestroy(g);
                    h->tr_exec = nullptr;
                    (void)hipGetLastError();
                }
            } else {
                (void)hipGetLastError();
            }
        }
        if (h->tr_exec && h->chain && n_steps - s >= nb) { // a replay starts from batch fb's slabs and leaves them behind again
            const float *a0 = h->DX + (size_t)(fb * B) * h->ld[0];
            if (!slabs_hold(h, a0, nullptr, B)) {
                const NextBatch self{a0, nullptr, B};
                launch_tile_step(h, 0, 0, &self, a0, B, 0.f, 0.f);
                slabs_now_hold(h, self, false);
            }
        }
        while (h->tr_exec && n_steps - s >= nb) {
            HIP_TRY(hipGraphLaunch(h->tr_exec, h->stream));
            h->time += (int)nb;
            s += (int)nb;
        }
    }
    for (; s < n_steps; s++) {
        const int64_t row0 = ((first / B + s) % nb) * B;
        // the step's tile kernel also starts the next step -- the LAST step's too: it prepares the batch that follows the range in
        // the data set, so that a caller who walks the data set call by call (an epoch, or a slice of one, per call) keeps the
        // two-launch chain across calls instead of opening every call with a forward-only launch (the sums are used only if the
        // next gradient is on exactly those rows with the weights as this step leaves them: slabs_hold)
        hint_range(h, ((first / B + s + 1) % nb) * B, B);
        TRY(step_on_rows(h, h->DX + (size_t)row0 * h->ld[0], h->DY + (size_t)row0 * h->ld[h->L - 1], B, step,
                         momentum, true));
    }
    return GNN_OK;
}); }


int gnn_mlp_gradient_step_indexed(gnn_mlp_t *h, const int32_t *idx, int B, double step, double momentum, int noise) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!idx) return fail(GNN_ERR_BAD_ARG, "null index list");
    TRY(check_step_args(h, B, step, noise));
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    for (int i = 0; i < B; i++)
        if (idx[i] < 0 || idx[i] >= h->dataset_n) return fail(GNN_ERR_BAD_ARG, "sample index out of range");
    HIP_TRY(hipMemcpyAsync(h->idxbuf, idx, sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    h->slab_valid = false; h->have_next = false; // idxbuf is reused: its address does not identify a batch
    return step_on_device_indices(h, h->idxbuf, B, step, momentum);
}); }

int gnn_mlp_loss_range(gnn_mlp_t *h, int64_t first, int B, double *loss_per_sample) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!loss_per_sample) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    do_forward(h, h->DX + (size_t)first * h->ld[0], h->DY + (size_t)first * h->ld[h->L - 1], B, false, true, false);
    TRY_LAUNCHES(h);
    return read_loss(h, B, loss_per_sample);
}); }

int gnn_mlp_argmax_range(gnn_mlp_t *h, int64_t first, int B, int32_t *labels) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!labels) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    do_forward(h, h->DX + (size_t)first * h->ld[0], nullptr, B, false, false, true);
    TRY_LAUNCHES(h);
    return read_labels(h, B, labels);
}); }

int gnn_mlp_count_hits_range(gnn_mlp_t *h, int64_t first, int64_t n, int64_t *hits) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!hits) return fail(GNN_ERR_BAD_ARG, "null output");
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    if (n <= 0 || first < 0 || first + n > h->dataset_n) return fail(GNN_ERR_BAD_ARG, "rows outside the dataset");
    // one block of max_batch rows after the other on the stream: forward + output rule + `>=` argmax (labels), then the
    // comparison with the expected class -- no host work and no readback between blocks
    DevScratch cnt;
    TRY(cnt.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(unsigned long long), h->stream));
    const int Lm = h->L - 1;
    const int block = eval_block_rows(h, n); // up to 16 384 rows through the evaluation workspace (f32), else max_batch
    int rc_ws = GNN_OK;
    EvalScope scope(h, block, &rc_ws);
    if (rc_ws != GNN_OK) return rc_ws;
    for (int64_t off = 0; off < n; off += block) {
        const int B = (int)std::min<int64_t>(block, n - off);
        const float *y = h->DY + (size_t)(first + off) * h->ld[Lm];
        do_forward(h, h->DX + (size_t)(first + off) * h->ld[0], nullptr, B, false, false, true);
        hipLaunchKernelGGL(count_hits_kernel, dim3((B + 255) / 256), dim3(256), 0, h->stream,
                           HitsParams{h->labels, y, h->ld[Lm], h->dims[Lm], B, cnt.as<unsigned long long>()});
    }
    TRY_LAUNCHES(h);
    unsigned long long got = 0;
    HIP_TRY(hipMemcpyAsync(&got, cnt.p, sizeof(got), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    *hits = (int64_t)got;
    return GNN_OK;
}); }

// ---- data-parallel hooks --------------------------------------------------------------------
int gnn_mlp_grad_device_ptr(gnn_mlp_t *h, void **dev_ptr) { return guarded([&]() -> int {
    if (!h || !dev_ptr) return fail(GNN_ERR_BAD_ARG, "null argument");
    *dev_ptr = h->G;
    return GNN_OK;
}); }

int gnn_mlp_bind_grad_buffer(gnn_mlp_t *h, void *dev_ptr, int64_t n_elems) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!dev_ptr) { h->G = h->G_own; return GNN_OK; }
    if (n_elems < h->n_pad) return fail(GNN_ERR_BAD_ARG, "gradient buffer shorter than gnn_mlp_grad_elems()");
    if (reinterpret_cast<uintptr_t>(dev_ptr) % 16) return fail(GNN_ERR_BAD_ARG, "gradient buffer must be 16-byte aligned");
    h->G = static_cast<float *>(dev_ptr);
    return GNN_OK;
}); }

int gnn_mlp_set_stream(gnn_mlp_t *h, void *hip_stream) { return guarded([&]() -> int {
    TRY(check_handle(h));
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
    if (!capturing) HIP_TRY(hipStreamSynchronize(h->stream)); // (a capturing stream cannot be waited on)
    else (void)hipGetLastError();
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return GNN_OK;
}); }

int gnn_mlp_compute_gradient_range(gnn_mlp_t *h, int64_t first, int B) { return guarded([&]() -> int {
    TRY(check_handle(h));
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    const float *a0 = h->DX + (size_t)first * h->ld[0];
    maybe_specialize(h);
    do_gradient(h, a0, h->DY + (size_t)first * h->ld[h->L - 1], B, false, 0.f, 0.f, true);
    TRY_LAUNCHES(h);
    return GNN_OK;
}); }

int gnn_mlp_hint_next_range(gnn_mlp_t *h, int64_t first, int B) { return guarded([&]() -> int {
    TRY(check_handle(h));
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    hint_range(h, first, B);
    return GNN_OK;
}); }

int gnn_mlp_apply_update(gnn_mlp_t *h, int B_global, double step, double momentum) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (B_global <= 0) return fail(GNN_ERR_BAD_ARG, "B_global must be positive");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    NextBatch nb{};
    if (h->chain && take_next(h, &nb)) {
        // the update by weight tiles, each tile going straight on to the next batch's first-layer slab
        launch_tile_step(h, 2, 2, &nb, nullptr, PAD, (float)(step / (double)B_global), (float)momentum);
        slabs_now_hold(h, nb, nb.idx != nullptr);
    } else {
        launch_flat_update(h, B_global, step, momentum);
        h->slab_valid = false; h->have_next = false;
    }
    h->time++;
    TRY_LAUNCHES(h);
    return GNN_OK;
}); }

int gnn_mlp_forget_lookahead(gnn_mlp_t *h) { return guarded([&]() -> int {
    if (!h) return fail(GNN_ERR_BAD_ARG, "null handle");
    h->slab_valid = false; h->have_next = false; h->xstage_valid = false;
    return GNN_OK;
}); }

int gnn_mlp_advance_time(gnn_mlp_t *h, int steps) { return guarded([&]() -> int {
    if (!h || h->time + steps < 0) return fail(GNN_ERR_BAD_ARG, "bad argument");
    h->time += steps; // negative: steps that were only CAPTURED (enqueued into a graph, not run)
    return GNN_OK;
}); }

int gnn_mlp_recover_stream(gnn_mlp_t *h) { return guarded([&]() -> int {
    TRY(check_handle(h));
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        hipGraph_t g = nullptr;
        (void)hipStreamEndCapture(h->stream, &g); // an invalidated capture returns an error and no graph
        if (g) (void)hipGraphDestroy(g);
        st = hipStreamCaptureStatusNone;
        // a stream the runtime keeps in the invalidated state is given up: the handle falls back to
        // its own stream and the caller binds a fresh one with gnn_mlp_set_stream
        if (hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone) h->stream = h->own_stream;
    }
    for (int i = 0; i < 8 && hipGetLastError() != hipSuccess; i++) {}
    return GNN_OK;
}); }

int gnn_mlp_synchronize(gnn_mlp_t *h) { return guarded([&]() -> int {
    TRY(check_handle(h));
    // A caller that waits for a few hundred microseconds of kernels should not pay an interrupt's wake-up (tens of
    // microseconds) on top: the stream is polled for a bounded time first, then the call blocks.
    // (hipErrorNotReady may be left behind as the thread's "last error", which check_launches reads: an error of an earlier
    //  launch is moved to the handle first, and the polling's own status is cleared afterwards)
    const hipError_t before = hipGetLastError();
    if (before != hipSuccess && h->launch_error == hipSuccess) h->launch_error = before;
    const auto t0 = std::chrono::steady_clock::now();
    bool done = false;
    for (;;) {
        const hipError_t e = hipStreamQuery(h->stream);
        if (e == hipSuccess) { done = true; break; }
        if (e != hipErrorNotReady) { (void)hipGetLastError(); return fail(GNN_ERR_HIP, std::string("hipStreamQuery: ") + hipGetErrorString(e)); }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
        __builtin_ia32_pause();
    }
    (void)hipGetLastError();
    if (!done) HIP_TRY(hipStreamSynchronize(h->stream));
    return GNN_OK;
}); }

// ---- shape specialisation ---------------------------------------------------------------------
int gnn_mlp_specialize(gnn_mlp_t *h) { return guarded([&]() -> int {
    TRY(check_handle(h));
    h->jit_tried = false;
    try_specialize(h);
    return GNN_OK;
}); }
int gnn_mlp_specialization(const gnn_mlp_t *h) { return h ? h->specialization : -1; }
int gnn_mlp_step_launches(const gnn_mlp_t *h) { return !h ? -1 : h->chain ? 2 : h->mid4 ? 3 : 0; }
int gnn_mlp_rowblock_state(const gnn_mlp_t *h) { return !h ? -1 : !h->rb ? 0 : h->rb_jit ? 3 : h->rb_static ? 2 : 1; }
const char *gnn_mlp_plan_note(const gnn_mlp_t *h) { return h ? h->plan_note.c_str() : ""; }

// ---- measurement ---------------------------------------------------------------------------
int gnn_mlp_timing_enable(gnn_mlp_t *h, int on) { return guarded([&]() -> int {
    TRY(check_handle(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->timing = on != 0;
    for (TimerClass &t : h->timers) t.used = 0;
    return GNN_OK;
}); }

int gnn_mlp_timing_read(gnn_mlp_t *h, int which, double *mean_us, int64_t *count) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (which < 0 || which > 4 || !mean_us || !count) return fail(GNN_ERR_BAD_ARG, "bad timing query");
    HIP_TRY(hipStreamSynchronize(h->stream));
    TimerClass &t = h->timers[which];
    double total = 0.0;
    for (size_t i = 0; i < t.used; i++) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t.start[i], t.stop[i]));
        total += ms;
    }
    *count = (int64_t)t.used;
    *mean_us = t.used ? total * 1000.0 / (double)t.used : 0.0;
    return GNN_OK;
}); }

} // extern "C"
