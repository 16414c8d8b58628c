// launch_misc.hip -- host side of the element-wise kernels (kernels.h): input / label encodings (MT:92-118), row
// gathers (NNT:143-158), exports, and the flat momentum update after an all-reduce (SCE:327-342).
#include "handle.h"

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace gnn;
using namespace gnn::host;

namespace gnn {
namespace host {

void launch_convert_rows(gnn_mlp *h, const double *src, int d, float *dst, int ld, int64_t rows, int64_t rows_pad, int act, int apply_act) {
    hipLaunchKernelGGL(convert_rows_f64_kernel, dim3(grid_for(rows_pad * ld)), dim3(256), 0, h->stream, src, d, dst, ld, rows, rows_pad, act, apply_act);
}
void launch_encode_u8(gnn_mlp *h, const uint8_t *pix, int d, float *dst, int ld, int64_t rows, int act) {
    hipLaunchKernelGGL(encode_u8_kernel, dim3(grid_for(rows * ld)), dim3(256), 0, h->stream, pix, d, dst, ld, rows, rows, act);
}
void launch_onehot_u8(gnn_mlp *h, const uint8_t *lab, int n_classes, float *dst, int ld, int64_t rows) {
    hipLaunchKernelGGL(onehot_u8_kernel, dim3(grid_for(rows * ld)), dim3(256), 0, h->stream, lab, n_classes, dst, ld, rows, rows);
}

// ---- host batches --------------------------------------------------------------------------------------------
// The reference's call shape is gradientStep(double[] rows) once per iteration (NNT:83): 0.81 MB of fp64 per call at
// 784-300-100-10 / B = 128.  A pageable hipMemcpyAsync of that is a synchronous copy into the runtime's own staging
// memory plus a DMA plus a convert kernel (66-97 us per step in round 2).  Here the calling thread rounds the rows to
// f32 straight into a pinned slot (half the bytes written and half the bytes crossing PCIe; the same RNE rounding the
// GPU's cvt applies), the staging kernel reads the slot across PCIe, and the call returns: the NEXT call's conversion
// runs while this step's kernels do.  No host pointer outlives the call.
namespace {
void rows_to_f32_plain(const double *src, float *dst, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (float)src[i];
}
__attribute__((target("avx2"))) void rows_to_f32_avx2(const double *src, float *dst, size_t n) {
    for (size_t i = 0; i < n; i++) dst[i] = (float)src[i]; // (vectorised by the compiler: vcvtpd2ps)
}
void rows_to_f32(const double *src, float *dst, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) rows_to_f32_avx2(src, dst, n);
    else rows_to_f32_plain(src, dst, n);
}

// The conversion is the call's largest host cost (0.8 MB read, 0.4 MB written per 128-row batch of 784 inputs: ~20 us on
// one core).  ONE helper thread per process takes the second half of a large batch while the calling thread does the first;
// it sleeps on a condition variable between calls (no spinning while the caller is elsewhere).  If the thread cannot be
// created the caller converts everything itself.
class ConvertHelper {
  public:
    ~ConvertHelper() {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
        cv_.notify_all();
        if (th_.joinable()) th_.join();
    }
    // converts [src, src + n) into dst using the helper for the upper part; returns when ALL of it is done
    void run(const double *src, float *dst, size_t n) {
        std::unique_lock<std::mutex> call(call_mu_, std::try_to_lock); // one batch at a time; a second caller converts alone
        if (n < (size_t)1 << 15 || !call.owns_lock() || !start()) { rows_to_f32(src, dst, n); return; }
        const size_t mine = (n * 9 / 16) & ~(size_t)15; // a little more than half: the helper has to wake up first
        {
            std::lock_guard<std::mutex> lk(mu_);
            src_ = src + mine; dst_ = dst + mine; n_ = n - mine;
            done_.store(false, std::memory_order_relaxed);
            pending_ = true;
        }
        cv_.notify_one();
        rows_to_f32(src, dst, mine);
        while (!done_.load(std::memory_order_acquire)) __builtin_ia32_pause(); // (microseconds: the helper started long ago)
    }

  private:
    bool start() {
        if (started_) return ok_;
        started_ = true;
        try {
            th_ = std::thread([this]() {
                std::unique_lock<std::mutex> lk(mu_);
                for (;;) {
                    cv_.wait(lk, [this] { return pending_ || quit_; });
                    if (quit_) return;
                    pending_ = false;
                    const double *s = src_; float *d = dst_; const size_t n = n_;
                    lk.unlock();
                    rows_to_f32(s, d, n);
                    done_.store(true, std::memory_order_release);
                    lk.lock();
                }
            });
            ok_ = true;
        } catch (...) {
            ok_ = false;
        }
        return ok_;
    }
    std::mutex mu_, call_mu_;
    std::condition_variable cv_;
    std::thread th_;
    bool started_ = false, ok_ = false, pending_ = false, quit_ = false;
    const double *src_ = nullptr; float *dst_ = nullptr; size_t n_ = 0;
    std::atomic<bool> done_{true};
};
ConvertHelper g_convert;
} // namespace

int stage_batch(gnn_mlp *h, const double *X, const double *Y, int B) {
    const int d0 = h->dims[0], dl = h->dims[h->L - 1], Lm = h->L - 1;
    const size_t cap = (size_t)h->max_batch * (size_t)(d0 + dl);
    const int slot = h->pin_next;
    if (!h->pin[slot]) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&h->pin[slot]), sizeof(float) * cap, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&h->pin_done[slot], hipEventDisableTiming));
    }
    if (h->pin_busy[slot]) { // the staging kernel of kHostSlots calls ago: long finished unless the caller outruns the GPU
        HIP_TRY(hipEventSynchronize(h->pin_done[slot]));
        h->pin_busy[slot] = false;
    }
    float *px = h->pin[slot], *py = px + (size_t)B * d0;
    g_convert.run(X, px, (size_t)B * d0);
    if (Y) rows_to_f32(Y, py, (size_t)B * dl);
    StageBatchParams p{};
    p.src_x = px; p.src_y = Y ? py : nullptr;
    p.d0 = d0; p.dl = dl;
    p.dst_x = h->act[0]; p.ld0 = h->ld[0];
    p.dst_y = h->ybuf; p.ldl = h->ld[Lm];
    p.rows = B; p.rows_pad = pad_up(B);
    p.act = h->inner_act;
    p.dst_xb = (h->dtype == GNN_DTYPE_BF16) ? h->actb[0] : nullptr;
    const int64_t total = p.rows_pad * ((int64_t)p.ld0 + (Y ? p.ldl : 0));
    hipLaunchKernelGGL(stage_batch_kernel, dim3(grid_for(total)), dim3(256), 0, h->stream, p);
    HIP_TRY(hipEventRecord(h->pin_done[slot], h->stream));
    h->pin_busy[slot] = true;
    h->pin_next = (slot + 1) % gnn_mlp::kHostSlots;
    return GNN_OK;
}

void release_host_staging(gnn_mlp *h) {
    for (int i = 0; i < gnn_mlp::kHostSlots; i++) {
        if (h->pin_done[i]) { (void)hipEventDestroy(h->pin_done[i]); h->pin_done[i] = nullptr; }
        if (h->pin[i]) { (void)hipHostFree(h->pin[i]); h->pin[i] = nullptr; }
        h->pin_busy[i] = false;
    }
}

int export_rows(gnn_mlp *h, const float *src, int ld, int d, int B, double *host_dst) {
    hipLaunchKernelGGL(export_rows_f64_kernel, dim3(grid_for((int64_t)B * d)), dim3(256), 0, h->stream, src, ld, d,
                       (int64_t)B, h->stage_out);
    HIP_TRY(hipMemcpyAsync(host_dst, h->stage_out, sizeof(double) * (size_t)B * d, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GNN_OK;
}

// inputs and expected rows of a sampled batch in ONE launch (nets off the fused path)
void launch_gather(gnn_mlp *h, const int32_t *d_idx, int B) {
    const int B_pad = pad_up(B), Lm = h->L - 1;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)B_pad * (h->ld[0] + h->ld[Lm]) / 4)), dim3(256), 0, h->stream,
                       h->DX, h->ld[0], h->act[0], h->DY, h->ld[Lm], h->ybuf, d_idx, B, B_pad);
}

void launch_flat_update(gnn_mlp *h, int B_global, double step, double momentum) {
    const int64_t n4 = h->n_pad / 4;
    launch_timed(h, GNN_K_UPDATE, sgd_momentum_kernel, dim3(grid_for(n4)), dim3(256), 0,
                 SgdParams{reinterpret_cast<float4 *>(h->W), reinterpret_cast<float4 *>(h->V),
                           reinterpret_cast<const float4 *>(h->G), n4, (float)(step / (double)B_global), (float)momentum,
                           reinterpret_cast<sgd_bf16x4 *>(h->Wb)});
}

} // namespace host
} // namespace gnn
