// launch_misc.hip -- host side of the element-wise kernels (kernels.h): input / label encodings (MT:92-118), row
// gathers (NNT:143-158), exports, and the flat momentum update after an all-reduce (SCE:327-342).
#include "handle.h"

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

// host fp64 rows -> device staging -> padded f32 (A_0 = f(x) when apply_act)
int stage_rows(gnn_mlp *h, const double *src, int d, int ld, int B, double *stage, float *dst, bool apply_act) {
    HIP_TRY(hipMemcpyAsync(stage, src, sizeof(double) * (size_t)B * d, hipMemcpyHostToDevice, h->stream));
    const int64_t rows_pad = pad_up(B);
    launch_convert_rows(h, stage, d, dst, ld, (int64_t)B, rows_pad, h->inner_act, apply_act ? 1 : 0);
    if (h->dtype == GNN_DTYPE_BF16 && dst == h->act[0]) to_bf16(h, dst, h->actb[0], (size_t)rows_pad * ld);
    return GNN_OK;
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
