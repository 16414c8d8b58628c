// launch_bf16.hip -- host side of GNN_DTYPE_BF16's per-layer GEMMs (gemm_bf16.h).
#include "handle.h"
#include "gemm_bf16_dma.h"

using namespace gnn;
using namespace gnn::host;

namespace gnn {
namespace host {

// bf16 operands (gemm_bf16.h): the same tile choice; 128x128 tiles only when they alone fill the chip
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2, int NSTG = 2>
void launch_gemm_bf16_t(gnn_mlp *h, int cls, const GemmBf16Params &p) {
    constexpr size_t lds = gemm_bf16_lds_bytes<BM, BN, A_KC, B_KC, NSTG>();
    static bool opted_in[kMaxOptInDevices] = {}; // (per instantiation, per device)
    opt_in_dynamic_lds(h, gemm_bf16_kernel<BM, BN, A_KC, B_KC, EPI, NSTG, WM>, lds, opted_in);
    launch_timed(h, cls, gemm_bf16_kernel<BM, BN, A_KC, B_KC, EPI, NSTG, WM>, dim3((p.N + BN - 1) / BN, (p.M + BM - 1) / BM), dim3(WM * 128), lds, GNN_GEMM_HEAD_ARGS(p), p);
}
// the DMA form (gemm_bf16_dma.h): whole tiles only
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM, int NIMG>
bool launch_gemm_bf16_dma_t(gnn_mlp *h, int cls, const GemmBf16Params &p) {
    if (h->env_bf16_dma_off || p.M % BM || p.N % BN || p.K % GemmBf16Depth<BM>::BK) return false;
    constexpr size_t lds = gemm_bf16_dma_lds_bytes<BM, BN, NIMG>();
    auto kern = gemm_bf16_dma_kernel<BM, BN, A_KC, B_KC, EPI, WM, NIMG>;
    static bool opted_in[kMaxOptInDevices] = {};
    opt_in_dynamic_lds(h, kern, lds, opted_in);
    launch_timed(h, cls, kern, dim3(p.N / BN, p.M / BM), dim3(WM * 128), lds, GNN_GEMM_HEAD_ARGS(p), p);
    return true;
}
template <bool A_KC, bool B_KC, int EPI>
void launch_gemm_bf16(gnn_mlp *h, int cls, const GemmBf16Params &p) {
    // bytes per MAC fall with the tile edge and these kernels are bound by operand traffic per CU, so 64 x 64 tiles
    // already from 128 tiles up (f32 wants 256): the 512 x 1024 logits of 4096-2048-2048-1024 took 13.5 us on 32 x 32 tiles
    int tile = pick_tile(p.M, p.N);
    if (tile == 32 && (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64) >= 128) tile = 64;
    // the gradient form with the update is bound by the masters' traffic in its epilogue, and there more, smaller workgroups
    // keep more of it in flight: 4096 x 2048 x 512 with the update 41.5 us on 128 x 128 tiles (one workgroup per CU at its
    // register count), 32.4 us on 64 x 64 (profiles/r02/gemm_probe_bf16_interior.log)
    if (tile == 128 && !A_KC && !B_KC && EPI == EPI_SGD) tile = 64;
    // Main-loop forms (gemm_bf16.h, NSTG; profiles/r04/gemm_probe_bf16_main_loop_forms.log, configs[3]'s products): TWO operand
    // images in LDS and one barrier per tile (5) with eight waves on a 64 x 64 tile is 15-22 % faster than one image (2) on the
    // forward and backward-data products (512 x 2048 x 4096: 24.8 -> 20.5 us; backward 512 x 2048 x 2048: 15.3 -> 11.9 -- which
    // was 18.5 before the k-contiguous image moved to one ds_read_b128 per fragment), 4-7 % on the two smaller gradient +
    // update products, and even on the largest (2 048 tiles, bound by the masters' traffic).
    constexpr bool fwd_or_bwd = A_KC;
    const int64_t t64 = (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64);
    // a forward product whose 64 x 64 grid is 128..255 tiles (the 512 x 1024 logits of configs[3]: 128) leaves half the chip
    // without a workgroup: 32 x 64 tiles, twice as many, 13.2 -> 10.1 us there (9.3 with two images).  Every other shape tried in
    // round 4 -- 64 x 32, 32 x 64, 32 x 32, 128 x 64 on the backward-data, forward-2 and gradient + update products -- is 5-45 %
    // SLOWER than 64 x 64 (profiles/r04/gemm_probe_bf16_tile_shapes.log).
    // Operand tiles by LDS DMA (gemm_bf16_dma.h; whole tiles only): three images and two tiles in flight for the long products,
    // two images where a second workgroup per CU matters more (the gradient + update products, whose epilogue moves the masters)
    // or K is short.  profiles/r04/gemm_probe_bf16_dma.log, against the two-image register-staged form: 512 x 2048 x 4096
    // 20.8 -> 16.9 us, backward 512 x 2048 x 2048 11.9 -> 10.7, the 512 x 1024 logits 9.6 -> 8.0, gradient + update
    // 2048 x 1024 x 512 11.3 -> 9.6; 4096 x 2048 x 2048 on 128 x 128 tiles 501 -> 717 TFLOP/s.
    const int nt = p.K / 128;
    if constexpr (A_KC && !B_KC) {
        if (tile == 64 && t64 < 256 && p.M % 32 == 0) {
            if (launch_gemm_bf16_dma_t<32, 64, A_KC, B_KC, EPI, 2, 2>(h, cls, p)) return;
            launch_gemm_bf16_t<32, 64, A_KC, B_KC, EPI, 2, 5>(h, cls, p);
            return;
        }
    }
    if (tile == 128 && launch_gemm_bf16_dma_t<128, 128, A_KC, B_KC, EPI, 4, 3>(h, cls, p)) return;
    if (tile == 64) {
        if constexpr (fwd_or_bwd) {
            if (nt >= 16 ? launch_gemm_bf16_dma_t<64, 64, A_KC, B_KC, EPI, 4, 3>(h, cls, p) : launch_gemm_bf16_dma_t<64, 64, A_KC, B_KC, EPI, 4, 2>(h, cls, p)) return;
        } else {
            if (launch_gemm_bf16_dma_t<64, 64, A_KC, B_KC, EPI, 4, 2>(h, cls, p)) return;
        }
    }
    switch (tile) {
    case 128: launch_gemm_bf16_t<128, 128, A_KC, B_KC, EPI>(h, cls, p); break;
    case 64:
        if constexpr (fwd_or_bwd) launch_gemm_bf16_t<64, 64, A_KC, B_KC, EPI, 4, 5>(h, cls, p);
        else if (t64 <= 1024) launch_gemm_bf16_t<64, 64, A_KC, B_KC, EPI, 4, 5>(h, cls, p);
        else launch_gemm_bf16_t<64, 64, A_KC, B_KC, EPI>(h, cls, p);
        break;
    default:
        // 256 x 1024 x 1024 (configs[4]): 6.2 -> 5.0 us forward, 5.4 -> 5.0 backward data by DMA (profiles/r04/gemm_probe_bf16_configs4.log);
        // ragged K (784) keeps the register-staged kernel, two images there for the forward form (6.4 -> 6.1)
        if (launch_gemm_bf16_dma_t<32, 32, A_KC, B_KC, EPI, 2, 2>(h, cls, p)) break;
        if constexpr (A_KC && !B_KC) launch_gemm_bf16_t<32, 32, A_KC, B_KC, EPI, 2, 5>(h, cls, p);
        else launch_gemm_bf16_t<32, 32, A_KC, B_KC, EPI>(h, cls, p);
        break;
    }
}

// ---- bf16 mode: forward / backward over the bf16 operand copies -------------------------------------
void forward_bf16(gnn_mlp *h, const __bf16 *a0b, int B, bool stop_before_last) {
    const int B_pad = pad_up(B);
    const __bf16 *in = a0b;
    for (int l = 1; l < h->L - (stop_before_last ? 1 : 0); l++) {
        GemmBf16Params p{};
        p.A = in; p.lda = h->ld[l - 1];
        p.B = h->Wb + h->w_off[l - 1]; p.ldb = h->ld[l];
        p.M = B_pad; p.N = h->ld[l]; p.K = h->ld[l - 1];
        p.m_true = B; p.n_true = h->dims[l];
        p.act = h->inner_act;
        p.ldc = h->ld[l];
        if (l < h->L - 1) {
            p.C = h->act[l]; p.Cb = h->actb[l];
            launch_gemm_bf16<true, false, EPI_ACT>(h, l == 1 ? GNN_K_FWD_GEMM0 : -1, p);
            in = h->actb[l];
        } else {
            p.C = h->logits; p.Cb = nullptr;
            launch_gemm_bf16<true, false, EPI_STORE>(h, l == 1 ? GNN_K_FWD_GEMM0 : -1, p);
        }
    }
}

// the gradient (+ update) products of all layers as ONE launch (gemm_bf16_group_kernel): when each is a few hundred tiles
template <int EPI>
void launch_gradient_group(gnn_mlp *h, GemmBf16Group &g) {
    constexpr size_t lds = gemm_bf16_lds_bytes<64, 64, false, false, 5>();
    auto kern = gemm_bf16_group_kernel<64, 64, false, false, EPI, 5, 4>;
    static bool opted_in[kMaxOptInDevices] = {};
    opt_in_dynamic_lds(h, kern, lds, opted_in);
    launch_timed(h, GNN_K_GRAD_GEMM0, kern, dim3(g.first[g.n]), dim3(512), lds, g);
}
bool gradients_in_one_launch(const gnn_mlp *h) {
    if (h->env_bf16_group_off || h->L - 1 < 2 || h->L - 1 > GNN_GEMM_GROUP_MAX) return false;
    int64_t tiles = 0;
    for (int l = 0; l < h->L - 1; l++) {
        const int64_t t = (int64_t)((h->ld[l] + 63) / 64) * ((h->ld[l + 1] + 63) / 64);
        if (t > 512) return false; // a product that fills the chip twice over gains nothing from sharing a launch, and has its own tile choice
        tiles += t;
    }
    return tiles <= 2048;
}

// have_tail: delta_{L-2} (and its bf16 copy) came from tail_kernel<true>
void backward_bf16(gnn_mlp *h, const __bf16 *a0b, int B, bool fused_update, float step_over_b, float momentum, bool have_tail) {
    const int B_pad = pad_up(B);
    const bool grouped = gradients_in_one_launch(h);
    GemmBf16Group grp{};
    for (int l = h->L - 2; l >= 0; l--) {
        if (l >= 1 && !(have_tail && l == h->L - 2)) { // delta_l = (delta_{l+1} . W_l^T) * f'(z_l)   -- before W_l is touched
            GemmBf16Params p{};
            p.A = h->deltab[l + 1]; p.lda = h->ld[l + 1];
            p.B = h->Wb + h->w_off[l]; p.ldb = h->ld[l + 1];
            p.C = h->delta[l]; p.Cb = h->deltab[l]; p.ldc = h->ld[l];
            p.M = B_pad; p.N = h->ld[l]; p.K = h->ld[l + 1];
            p.m_true = B; p.n_true = h->dims[l];
            p.aux = h->act[l]; p.ldaux = h->ld[l];
            p.act = h->inner_act;
            launch_gemm_bf16<true, true, EPI_DACT>(h, -1, p);
        }
        GemmBf16Params g{}; // G_l = A_l^T . delta_{l+1}
        g.A = (l == 0) ? a0b : h->actb[l]; g.lda = h->ld[l];
        g.B = h->deltab[l + 1]; g.ldb = h->ld[l + 1];
        g.ldc = h->ld[l + 1];
        g.M = h->ld[l]; g.N = h->ld[l + 1]; g.K = B_pad;
        g.m_true = h->dims[l]; g.n_true = h->dims[l + 1];
        const int cls = (l == 0) ? GNN_K_GRAD_GEMM0 : -1;
        if (fused_update) {
            g.W = h->W + h->w_off[l]; g.V = h->V + h->w_off[l]; g.Wb = h->Wb + h->w_off[l];
            g.step_over_b = step_over_b; g.momentum = momentum;
            if (!grouped) launch_gemm_bf16<false, false, EPI_SGD>(h, cls, g);
        } else {
            g.C = h->G + h->w_off[l];
            if (!grouped) launch_gemm_bf16<false, false, EPI_STORE>(h, cls, g);
        }
        if (grouped) grp.p[l] = g; // (slot l: layer 0, the largest at MNIST-like shapes, first.  Every backward-data product
                                   //  above reads its W before ANY update runs: the grouped launch comes after the loop.)
    }
    if (grouped) {
        grp.n = h->L - 1;
        for (int l = 0; l < grp.n; l++) {
            grp.tiles_x[l] = (grp.p[l].N + 63) / 64;
            grp.first[l + 1] = grp.first[l] + grp.tiles_x[l] * ((grp.p[l].M + 63) / 64);
        }
        if (fused_update) launch_gradient_group<EPI_SGD>(h, grp);
        else launch_gradient_group<EPI_STORE>(h, grp);
    }
}

// f32 rows -> their bf16 rounding (n floats, a multiple of 4)
void to_bf16(gnn_mlp *h, const float *src, __bf16 *dst, size_t n) {
    const int64_t n4 = (int64_t)(n / 4);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid_for(n4)), dim3(256), 0, h->stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<bf16x4 *>(dst), n4);
}

} // namespace host
} // namespace gnn
