// launch_misc.hip -- host side of the element-wise kernels (kernels.h): input / label encodings (MT:92-118), row
// gathers (NNT:143-158), exports, and the flat momentum update after an all-reduce (SCE:327-342).
#include "handle.h"
#include "convert_helper.h"

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
