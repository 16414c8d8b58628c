// gnn_mlp.hip -- C ABI (include/gnn_mlp.h) over the gfx950 kernels in kernels.h.
// Host orchestration of one gradientStep (SCE:297-346): forward GEMM chain, fused output
// layer, backward-data GEMM chain, weight-gradient GEMMs with the momentum update fused into
// their epilogue (single GPU) or written to the flat gradient buffer (data parallel).
#include "../../include/gnn_mlp.h"
#include "java_random.h"
#include "jit.h"
#include "fused_kernels.h"
#include "gemm_bf16.h"
#include "gemm_wavek.h"
#include "middle4_kernel.h"
#include "tile_step_kernel.h"
#include "kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace gnn;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(GNN_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

struct TimerClass {
    std::vector<hipEvent_t> start, stop;
    size_t used = 0;
};

} // namespace

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
    double *stage_x = nullptr, *stage_y = nullptr, *stage_out = nullptr;

    float *DX = nullptr, *DY = nullptr; // device-resident dataset (A_0 = f(x) and y)
    // GNN_DTYPE_BF16 (gemm_bf16.h): bf16 roundings of every GEMM operand, written once by its producer
    __bf16 *Wb = nullptr;               // shadow of W, same padded layout
    std::vector<__bf16 *> actb, deltab; // actb[l] l = 0..L-2, deltab[l] l = 1..L-1
    __bf16 *DXb = nullptr;              // dataset inputs
    int64_t dataset_n = 0;

    hipStream_t stream = nullptr, own_stream = nullptr;

    // fused small-net path (fused_kernels.h): plan made once at create
    bool fused = false;
    GradParams grad{};
    int grad_tiles = 0;
    GradParams grad64{};      // the same layers cut into 64x64 tiles (grad_update64_kernel), used when grad_tiles is large
    int grad_tiles64 = 0;
    bool mid_generic = false; // middle weights exceed LDS: per-layer GEMMs, fwd_first / grad_update chosen per call (hybrid_choice)
    bool mid4 = false;        // middle4_kernel: every middle weight matrix resident in LDS
    Mid4Params mid4p{};
    size_t mid4_lds_bytes = 0;
    const void *mid4_fn[3] = {nullptr, nullptr, nullptr}; // forward only / forward + backward / the same with A_1 from K slabs
                                                          // (bf16 nets: only slot 2, the bf16 training kernel)
    hipFunction_t mid4_jit[3] = {nullptr, nullptr, nullptr}; // run-time instantiation (jit.h), preferred when set

    // two-launch step (tile_step_kernel.h): the tile kernel of step s also makes the first-layer K slabs of step s+1
    bool chain = false;
    TileStepParams tsp{};
    int ts_tiles = 0, ts_tiles0 = 0; // blocks of all layers / of layer 0 alone
    float *slabs = nullptr;
    int n_slabs = 0;
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

    hipError_t launch_error = hipSuccess; // first refused launch of a module / function-pointer kernel since the last check
    const int32_t *cur_idx = nullptr; // device row indices of the batch being stepped (fused path reads rows through them)

    bool timing = false;
    TimerClass timers[5];

    // development / test switches, read ONCE at create (never on the step path)
    int env_path = 0;          // GNN_MLP_PATH: 0 default, 1 "generic", 2 "nomid4"
    int env_hybrid = -1;       // GNN_MLP_HYBRID: -1 unset, else bit 0 = fwd_first, bit 1 = grad_update
    bool env_tail_off = false; // GNN_MLP_TAIL=0: the three-launch form instead of tail_kernel
    bool env_wavek_off = false; // GNN_MLP_WAVEK=0: gemm_f32_kernel<32, 32> instead of the wave-K kernel (development)
    bool env_graph = false;    // GNN_MLP_GRAPH=1: train_range replays a captured pass
    bool env_jit_off = false;  // GNN_MLP_JIT=0
    bool env_static_off = false; // GNN_MLP_STATIC=0
    bool env_chain_off = false;  // GNN_MLP_CHAIN=0: three launches per step (fwd_first / middle4 / grad_update)
};

namespace {

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
#define TRY_LAUNCHES(h)                 \
    do {                                \
        int rc_ = check_launches(h);    \
        if (rc_ != GNN_OK) return rc_;  \
    } while (0)

int check_handle(const gnn_mlp *h) {
    if (!h) return fail(GNN_ERR_BAD_ARG, "null handle");
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return fail(GNN_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
    return GNN_OK;
}

int grid_for(int64_t n) {
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
template <class K, class P> void launch_timed(gnn_mlp *h, int cls, K kernel, dim3 grid, dim3 block, size_t lds, const P &params) {
    if (h->timing && cls >= 0) {
        TimerClass &t = h->timers[cls];
        if (t.used < 8192) {
            if (t.used >= t.start.size()) {
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess) { t.start.push_back(a); t.stop.push_back(b); }
            }
            if (t.used < t.start.size()) {
                const size_t slot = t.used++;
                hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)lds, h->stream, t.start[slot], t.stop[slot], 0, params);
                return;
            }
        }
    }
    hipLaunchKernelGGL(kernel, grid, block, lds, h->stream, params);
}

// ---- GEMM dispatch ----------------------------------------------------------------------
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2>
void launch_gemm_t(gnn_mlp *h, int cls, const GemmParams &p) {
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM);
    launch_timed(h, cls, gemm_f32_kernel<BM, BN, A_KC, B_KC, EPI, WM>, grid, dim3(WM * 128), 0, p);
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
    launch_timed(h, cls, gemm_f32_wavek_kernel<A_KC, B_KC, EPI, 4, 2>, dim3(p.N / 32, p.M / 32), dim3(256), 0, p);
}

template <bool A_KC, bool B_KC, int EPI>
void launch_gemm(gnn_mlp *h, int cls, const GemmParams &p) {
    const int tile = pick_tile(p.M, p.N);
    if (tile == 32 && !h->env_wavek_off && wavek_fits(p.M, p.N, p.K)) { launch_gemm_wavek<A_KC, B_KC, EPI>(h, cls, p); return; }
    // A square grid of 256..511 tiles is ONE 4-wave workgroup per CU: nothing covers its barriers and LDS latencies.  Measured
    // per form (profiles/r02/gemm_probe_tiles2.log, 512-row products of 4096-2048-2048-1024, one register stage, unguarded loads):
    //   forward and backward data (a k-contiguous operand): 64 x 64 tiles with EIGHT waves -- two per SIMD from one workgroup and
    //     a third less operand traffic than 64 x 32 (512 x 2048 x 4096: 81.3 against 88.3 us; backward 512 x 2048 x 1024: 23.8
    //     against 30.0);
    //   gradient (both k-major): 64 x 32 below 512 tiles of 64 x 64; 128 x 128 tiles only from 512 of them up, 256..511 of them
    //     run as 64 x 64 (2048 x 2048 x 512 with the update: 45.4 against 48.1 us).
    const int64_t t64 = (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64), t128 = (int64_t)((p.M + 127) / 128) * ((p.N + 127) / 128);
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

// bf16 operands (gemm_bf16.h): the same tile choice; 128x128 tiles only when they alone fill the chip
template <int BM, int BN, bool A_KC, bool B_KC, int EPI, int WM = 2>
void launch_gemm_bf16_t(gnn_mlp *h, int cls, const GemmBf16Params &p) {
    constexpr size_t lds = gemm_bf16_lds_bytes<BM, BN, A_KC, B_KC>();
    static bool opted_in = false; // more than 64 KB of dynamic LDS needs the opt-in, once per instantiation
    if (!opted_in) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm_bf16_kernel<BM, BN, A_KC, B_KC, EPI, 2, WM>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            if (h->launch_error == hipSuccess) h->launch_error = hipGetLastError();
        }
        opted_in = true;
    }
    launch_timed(h, cls, gemm_bf16_kernel<BM, BN, A_KC, B_KC, EPI, 2, WM>, dim3((p.N + BN - 1) / BN, (p.M + BM - 1) / BM), dim3(WM * 128), lds, p);
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
    switch (tile) {
    case 128: launch_gemm_bf16_t<128, 128, A_KC, B_KC, EPI>(h, cls, p); break;
    case 64:
        // the forward form gains ~6 % from eight waves on the tile (512 x 2048 x 4096: 28.0 -> 26.3 us); backward data loses
        // 3-10 %, the gradient form is even (profiles/r02/gemm_probe_bf16_waves.log)
        if constexpr (A_KC && !B_KC) launch_gemm_bf16_t<64, 64, A_KC, B_KC, EPI, 4>(h, cls, p);
        else launch_gemm_bf16_t<64, 64, A_KC, B_KC, EPI>(h, cls, p);
        break;
    default: launch_gemm_bf16_t<32, 32, A_KC, B_KC, EPI>(h, cls, p); break;
    }
}

// ---- bf16 mode: forward / backward over the bf16 operand copies -------------------------------------
void forward_bf16(gnn_mlp *h, const __bf16 *a0b, int B) {
    const int B_pad = pad_up(B);
    const __bf16 *in = a0b;
    for (int l = 1; l < h->L; l++) {
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

void backward_bf16(gnn_mlp *h, const __bf16 *a0b, int B, bool fused_update, float step_over_b, float momentum) {
    const int B_pad = pad_up(B);
    for (int l = h->L - 2; l >= 0; l--) {
        if (l >= 1) { // delta_l = (delta_{l+1} . W_l^T) * f'(z_l)   -- before W_l is touched
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
            launch_gemm_bf16<false, false, EPI_SGD>(h, cls, g);
        } else {
            g.C = h->G + h->w_off[l];
            launch_gemm_bf16<false, false, EPI_STORE>(h, cls, g);
        }
    }
}

// f32 rows -> their bf16 rounding (n floats, a multiple of 4)
void to_bf16(gnn_mlp *h, const float *src, __bf16 *dst, size_t n) {
    const int64_t n4 = (int64_t)(n / 4);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(grid_for(n4)), dim3(256), 0, h->stream, reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<bf16x4 *>(dst), n4);
}

constexpr int FIRST_NW = 8;  // waves per fwd_first_kernel workgroup (K split in-LDS)

// ---- forward (SCE:164-198): a0 = f(x) rows, B live rows ------------------------------------
// leaves act[1..L-2], logits; the output kernel is launched by the caller via run_output.
void forward(gnn_mlp *h, const float *a0, int B, int first_l = 1, bool stop_before_last = false) {
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
              bool data_only = false, bool have_last_delta = false) {
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

// ---- fused small-net path ---------------------------------------------------------------------

void plan_mid4(gnn_mlp *h);
void plan_chain(gnn_mlp *h);

// Decides whether the net fits the fused path and lays out the middle kernel's LDS.
void plan_fused(gnn_mlp *h) {
    h->fused = false;
    h->mid4 = false;
    h->mid_generic = false;
    if (h->env_path == 1) return;
    const int L = h->L, Lm = L - 1;
    if (L < 3 || L > MAX_LAYERS) return;
    if (h->dtype != GNN_DTYPE_F32) {
        // bf16 operands: per-layer GEMMs (gemm_bf16.h) for inference and for nets off the row-block path; training
        // of a net that fits the row-block kernel takes the two-launch path in bf16 (tile_step_bf16_kernel + the bf16
        // instance of middle4_kernel)
        plan_mid4(h);
        if (h->mid4) plan_chain(h);
        if (!h->chain) h->mid4 = false;
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
    if (h->mid4) { h->fused = true; plan_chain(h); return; }
    // the middle weights do not fit LDS: per-layer tiled GEMMs for the middle, still bracketed by
    // the one-launch first layer and the one-launch gradient+update (a 16-row kernel that streamed
    // the middle weights from L2 was 15-40 % slower than this on every such shape and was removed)
    h->mid_generic = true;
    h->fused = true;
}

// ---- middle4_kernel plan ----------------------------------------------------------------------
// kernel table: [shape policy][activation][output kind][backward]
// variant: 0 forward only, 1 forward + backward, 2 forward + backward with A_1 from the K slabs of tile_step_kernel
template <class SH, int OUTK> const void *mid4_fn_sh(int act, int variant) {
#define GNN_M4(A) (variant == 3 ? reinterpret_cast<const void *>(&middle4_kernel<SH, A, OUTK, true, false, true, true>) \
                   : variant == 2 ? reinterpret_cast<const void *>(&middle4_kernel<SH, A, OUTK, true, false, true>)  \
                   : variant == 1 ? reinterpret_cast<const void *>(&middle4_kernel<SH, A, OUTK, true>)             \
                                  : reinterpret_cast<const void *>(&middle4_kernel<SH, A, OUTK, false>))
    switch (act) {
    case 0: return GNN_M4(0);
    case 1: return GNN_M4(1);
    case 2: return GNN_M4(2);
    case 3: return GNN_M4(3);
    default: return GNN_M4(4);
    }
#undef GNN_M4
}
// shapes with compile-time plans (BASELINE.json configs that take the fused path)
using ShapeMnistA = StaticShape<784, 300, 100, 10>;
using ShapeMnistB = StaticShape<784, 100, 50, 10>;

template <class SH> bool shape_matches(const gnn_mlp *h) {
    constexpr int n = (int)(sizeof(SH::kDims) / sizeof(int));
    if (h->L != n) return false;
    for (int i = 0; i < n; i++) if (h->dims[i] != SH::kDims[i]) return false;
    return true;
}

const void *mid4_function(const gnn_mlp *h, int variant) {
    const bool allow_static = !h->env_static_off;
    if (allow_static && h->out_kind == GNN_OUT_SOFTMAX_CE) {
        if (shape_matches<ShapeMnistA>(h)) return mid4_fn_sh<ShapeMnistA, 0>(h->inner_act, variant);
        if (shape_matches<ShapeMnistB>(h)) return mid4_fn_sh<ShapeMnistB, 0>(h->inner_act, variant);
    }
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
    if (h->env_path == 2) return; // tests: force the per-layer middle
    const int L = h->L, Lm = L - 1;
    Mid4Params &m = h->mid4p;
    m = Mid4Params{};
    const bool bf16 = h->dtype == GNN_DTYPE_BF16;
    m.plan = make_mid4_plan(h->dims.data(), L, bf16);
    if (!m.plan.ok) return;
    h->mid4_lds_bytes = (size_t)m.plan.lds_floats * sizeof(float);
    for (int l = 1; l < Lm; l++) { m.W[l] = h->W + h->w_off[l]; m.act[l] = h->act[l]; }
    for (int l = 1; l <= Lm; l++) m.delta[l] = h->delta[l];
    if (bf16) {
        for (int l = 1; l < Lm; l++) { m.Wb[l] = h->Wb + h->w_off[l]; m.actb[l] = h->actb[l]; }
        for (int l = 1; l <= Lm; l++) m.deltab[l] = h->deltab[l];
    }
    m.last_act = h->last_act;
    m.inner_act = h->inner_act;
    {
        const bool allow_static = !h->env_static_off;
        h->specialization = (allow_static && h->out_kind == GNN_OUT_SOFTMAX_CE &&
                             (shape_matches<ShapeMnistA>(h) || shape_matches<ShapeMnistB>(h))) ? 1 : 0;
    }
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

// ---- two-launch step: tile_step_kernel plan ----------------------------------------------------
// Every layer's weight matrix in 64 x 16 tiles, one grid; layer 0's tiles first (they also make the next
// batch's first-layer K slabs).  Needs the row-block kernel (middle4) and at most MID4_MAX_SLABS slabs.
void plan_chain(gnn_mlp *h) {
    h->chain = false;
    if (h->env_chain_off || !h->mid4) return;
    const int L = h->L;
    h->n_slabs = (h->ld[0] + TS_TM - 1) / TS_TM;
    if (h->n_slabs > MID4_MAX_SLABS) return;
    TileStepParams &t = h->tsp;
    t = TileStepParams{};
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
        if (l == 0) h->ts_tiles0 = tiles;
    }
    h->ts_tiles = tiles;
    const size_t n = (size_t)h->n_slabs * h->cap_rows * h->ld[1];
    if (n >= (1ull << 30)) return; // middle4 addresses the slabs with 32-bit byte offsets
    if (hipMalloc(reinterpret_cast<void **>(&h->slabs), sizeof(float) * n) != hipSuccess) { (void)hipGetLastError(); h->slabs = nullptr; return; }
    if (hipMemsetAsync(h->slabs, 0, sizeof(float) * n, h->stream) != hipSuccess) { (void)hipGetLastError(); return; }
    t.slabs = h->slabs; t.slab_rows = h->cap_rows; t.ldz = h->ld[1];
    for (int i = 0; i < 2; i++) {
        const size_t xn = (size_t)h->cap_rows * h->ld[0];
        if (h->dtype == GNN_DTYPE_BF16) {
            if (hipMalloc(reinterpret_cast<void **>(&h->xstage_b[i]), sizeof(__bf16) * xn) != hipSuccess) { (void)hipGetLastError(); h->xstage_b[i] = nullptr; return; }
        } else {
            if (hipMalloc(reinterpret_cast<void **>(&h->xstage[i]), sizeof(float) * xn) != hipSuccess) { (void)hipGetLastError(); h->xstage[i] = nullptr; return; }
        }
    }
    h->chain = true;
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
}

// first layer in one launch: a 16x16 tile per workgroup, K split over the waves
void launch_fwd_first(gnn_mlp *h, const float *a0, int B) {
    const int B_pad = pad_up(B);
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

// forward of the middle4 path; backward = also delta_1..delta_{L-1}
// from_slabs: A_1 = f(sum of the K slabs) (tile_step_kernel made them); else fwd_first_kernel writes act[1] first
void fused_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool backward, bool want_prob,
                   bool want_loss, bool want_label, bool from_slabs = false) {
    if (!from_slabs) launch_fwd_first(h, a0, B);
    {
        Mid4Params m4 = h->mid4p;
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
        if (fused_update) launch_timed(h, GNN_K_GRAD_GEMM0, grad_update64_kernel<true>, dim3(h->grad_tiles64), dim3(512), 0, g);
        else launch_timed(h, GNN_K_GRAD_GEMM0, grad_update64_kernel<false>, dim3(h->grad_tiles64), dim3(512), 0, g);
    } else {
        if (fused_update) launch_timed(h, GNN_K_GRAD_GEMM0, grad_update_kernel<true>, dim3(h->grad_tiles), dim3(GRAD_THREADS), 0, g);
        else launch_timed(h, GNN_K_GRAD_GEMM0, grad_update_kernel<false>, dim3(h->grad_tiles), dim3(GRAD_THREADS), 0, g);
    }
}

// ---- tile_step_kernel launches ------------------------------------------------------------------
struct NextBatch { const float *a0; const int32_t *idx; int B; };
const __bf16 *a0_bf16(const gnn_mlp *h, const float *a0);

// gsrc / gdst / fwd as in tile_step_kernel.h; fwd_only_layer0: the grid covers layer 0's tiles only
// staged: the current batch's rows come from the contiguous copy xstage[xstage_cur] instead of (a0, cur_idx)
void launch_tile_step(gnn_mlp *h, int gsrc, int gdst, const NextBatch *next, const float *a0, int B, float step_over_b, float momentum,
                      bool staged = false) {
    TileStepParams t = h->tsp;
    t.layer[0].A = staged ? h->xstage[h->xstage_cur] : a0;
    for (int l = 0; l < t.n_layers; l++) t.layer[l].G = h->G + h->w_off[l];
    t.K = pad_up(B); t.k_true = B;
    t.row_idx = staged ? nullptr : h->cur_idx;
    const int stage_dst = h->xstage_cur ^ 1; // a sampled next batch is copied to the OTHER buffer (this launch may be reading the current one)
    if (next && next->idx) { t.stage_out = h->xstage[stage_dst]; t.stage_out_b = h->xstage_b[stage_dst]; }
    t.step_over_b = step_over_b; t.momentum = momentum;
    const bool fwd = next != nullptr;
    if (fwd) { t.An = next->a0; t.ldan = h->ld[0]; t.next_idx = next->idx; t.next_rows = next->B; t.next_K = pad_up(next->B); }
    const bool fwd_only = gsrc == 0;
    const dim3 grid(fwd_only ? h->ts_tiles0 : h->ts_tiles), block(TS_THREADS);
    if (fwd_only) t.n_layers = 1;
    const int cls = fwd_only ? GNN_K_FWD_GEMM0 : gsrc == 2 ? GNN_K_UPDATE : GNN_K_GRAD_GEMM0;
    if (h->dtype == GNN_DTYPE_BF16) {
        if (staged) t.Ab[0] = h->xstage_b[h->xstage_cur];
        else if (a0) t.Ab[0] = a0_bf16(h, a0);
        if (fwd) t.Anb = a0_bf16(h, next->a0);
        if (fwd_only) launch_timed(h, cls, tile_step_bf16_kernel<0, 0, true>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 1) launch_timed(h, cls, tile_step_bf16_kernel<1, 1, false>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_bf16_kernel<1, 2, false>, grid, block, 0, t);
        else if (gsrc == 1 && gdst == 2 && fwd) launch_timed(h, cls, tile_step_bf16_kernel<1, 2, true>, grid, block, 0, t);
        else if (gsrc == 2 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_bf16_kernel<2, 2, false>, grid, block, 0, t);
        else launch_timed(h, cls, tile_step_bf16_kernel<2, 2, true>, grid, block, 0, t);
        return;
    }
    if (fwd_only) launch_timed(h, cls, tile_step_kernel<0, 0, true>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 1) launch_timed(h, cls, tile_step_kernel<1, 1, false>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_kernel<1, 2, false>, grid, block, 0, t);
    else if (gsrc == 1 && gdst == 2 && fwd) launch_timed(h, cls, tile_step_kernel<1, 2, true>, grid, block, 0, t);
    else if (gsrc == 2 && gdst == 2 && !fwd) launch_timed(h, cls, tile_step_kernel<2, 2, false>, grid, block, 0, t);
    else launch_timed(h, cls, tile_step_kernel<2, 2, true>, grid, block, 0, t);
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
void slabs_now_hold(gnn_mlp *h, const NextBatch &nb, bool staged_copy = false) {
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
    const bool staged = h->xstage_valid && h->cur_idx != nullptr; // the launch that made the slabs left a contiguous copy of these rows
    h->slab_valid = false;
    fused_forward(h, a0, y, B, true, false, false, false, true);
    NextBatch nb{};
    if (fused_update) {
        const bool fwd = take_next(h, &nb);
        launch_tile_step(h, 1, 2, fwd ? &nb : nullptr, a0, B, step_over_b, momentum, staged);
        if (fwd) slabs_now_hold(h, nb, nb.idx != nullptr);
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
    return !h->env_tail_off && h->dtype == GNN_DTYPE_F32 && h->out_kind == GNN_OUT_SOFTMAX_CE && h->ld[h->L - 1] == 16;
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
    launch_timed(h, -1, tail_kernel, dim3(row_blocks, splits), dim3(512), 0, t);
}

// bf16 twin of an A_0 row pointer: the staging rows or the resident dataset
const __bf16 *a0_bf16(const gnn_mlp *h, const float *a0) {
    if (a0 == h->act[0]) return h->actb[0];
    return h->DXb + (a0 - h->DX);
}

// the three shapes every entry point is made of
void do_forward(gnn_mlp *h, const float *a0, const float *y, int B, bool want_prob, bool want_loss, bool want_label) {
    if (h->dtype == GNN_DTYPE_BF16) {
        forward_bf16(h, a0_bf16(h, a0), B);
        run_output(h, y, B, want_prob, false, want_loss, want_label);
        return;
    }
    if (h->mid4) { fused_forward(h, a0, y, B, false, want_prob, want_loss, want_label); return; }
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
                 bool resident = false) {
    if (h->chain) { chain_gradient(h, a0, y, B, fused_update, step_over_b, momentum, resident); return; }
    h->have_next = false;
    if (h->dtype == GNN_DTYPE_BF16) {
        const __bf16 *a0b = a0_bf16(h, a0);
        forward_bf16(h, a0b, B);
        run_output(h, y, B, false, true, false, false);
        backward_bf16(h, a0b, B, fused_update, step_over_b, momentum);
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

// host fp64 rows -> device staging -> padded f32 (A_0 = f(x) when apply_act)
int stage_rows(gnn_mlp *h, const double *src, int d, int ld, int B, double *stage, float *dst, bool apply_act) {
    HIP_TRY(hipMemcpyAsync(stage, src, sizeof(double) * (size_t)B * d, hipMemcpyHostToDevice, h->stream));
    const int64_t rows_pad = pad_up(B);
    hipLaunchKernelGGL(convert_rows_f64_kernel, dim3(grid_for(rows_pad * ld)), dim3(256), 0, h->stream, stage, d,
                       dst, ld, (int64_t)B, rows_pad, h->inner_act, apply_act ? 1 : 0);
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

#define TRY(expr)                      \
    do {                               \
        int rc_ = (expr);              \
        if (rc_ != GNN_OK) return rc_; \
    } while (0)

} // namespace

extern "C" {

const char *gnn_mlp_last_error(void) { return g_last_error.c_str(); }

int gnn_mlp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act, int loss,
                   int64_t seed, int dtype, int device, int max_batch, gnn_mlp_t **out) {
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
    CTRY(dev_alloc(&h->stage_x, (size_t)max_batch * dims[0], h->stream));
    CTRY(dev_alloc(&h->stage_y, (size_t)max_batch * dims[n_dims - 1], h->stream));
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
}

int gnn_mlp_destroy(gnn_mlp_t *h) {
    if (!h) return GNN_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    auto fr = [](void *p) { if (p) (void)hipFree(p); };
    fr(h->W); fr(h->V); fr(h->G_own);
    for (float *p : h->act) fr(p);
    for (float *p : h->delta) fr(p);
    fr(h->logits); fr(h->prob); fr(h->ybuf); fr(h->lossv); fr(h->labels); fr(h->idxbuf);
    fr(h->stage_x); fr(h->stage_y); fr(h->stage_out); fr(h->DX); fr(h->DY); fr(h->slabs);
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
}

int gnn_mlp_input_dim(const gnn_mlp_t *h) { return h ? h->dims[0] : -1; }
int gnn_mlp_output_dim(const gnn_mlp_t *h) { return h ? h->dims[h->L - 1] : -1; }
int64_t gnn_mlp_num_params(const gnn_mlp_t *h) { return h ? h->n_params : -1; }
int gnn_mlp_time(const gnn_mlp_t *h) { return h ? h->time : -1; }
int64_t gnn_mlp_dataset_size(const gnn_mlp_t *h) { return h ? h->dataset_n : -1; }
int64_t gnn_mlp_grad_elems(const gnn_mlp_t *h) { return h ? h->n_pad : -1; }

int gnn_mlp_propagate(gnn_mlp_t *h, const double *X, int B, double *out) {
    TRY(check_handle(h));
    if (!X || !out) return fail(GNN_ERR_BAD_ARG, "null argument (reference: assert input != null, SCE:165)");
    TRY(check_batch(h, B));
    TRY(stage_rows(h, X, h->dims[0], h->ld[0], B, h->stage_x, h->act[0], true));
    do_forward(h, h->act[0], nullptr, B, true, false, false);
    TRY_LAUNCHES(h);
    return export_rows(h, h->prob, h->ld[h->L - 1], h->dims[h->L - 1], B, out);
}

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

int gnn_mlp_loss(gnn_mlp_t *h, const double *X, const double *Y, int B, double *loss_per_sample) {
    TRY(check_handle(h));
    if (!X || !Y || !loss_per_sample) return fail(GNN_ERR_BAD_ARG, "null argument (SCE:209)");
    TRY(check_batch(h, B));
    const int Lm = h->L - 1;
    TRY(stage_rows(h, X, h->dims[0], h->ld[0], B, h->stage_x, h->act[0], true));
    TRY(stage_rows(h, Y, h->dims[Lm], h->ld[Lm], B, h->stage_y, h->ybuf, false));
    do_forward(h, h->act[0], h->ybuf, B, false, true, false);
    TRY_LAUNCHES(h);
    return read_loss(h, B, loss_per_sample);
}

int gnn_mlp_argmax(gnn_mlp_t *h, const double *X, int B, int32_t *labels) {
    TRY(check_handle(h));
    if (!X || !labels) return fail(GNN_ERR_BAD_ARG, "null argument");
    TRY(check_batch(h, B));
    TRY(stage_rows(h, X, h->dims[0], h->ld[0], B, h->stage_x, h->act[0], true));
    do_forward(h, h->act[0], nullptr, B, false, false, true);
    TRY_LAUNCHES(h);
    return read_labels(h, B, labels);
}

int gnn_mlp_compute_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B) {
    TRY(check_handle(h));
    if (!X || !Y) return fail(GNN_ERR_BAD_ARG, "null argument (SCE:231)");
    TRY(check_batch(h, B));
    const int Lm = h->L - 1;
    TRY(stage_rows(h, X, h->dims[0], h->ld[0], B, h->stage_x, h->act[0], true));
    TRY(stage_rows(h, Y, h->dims[Lm], h->ld[Lm], B, h->stage_y, h->ybuf, false));
    do_gradient(h, h->act[0], h->ybuf, B, false, 0.f, 0.f, false);
    TRY_LAUNCHES(h);
    return GNN_OK;
}

int gnn_mlp_weight_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B, double *flat_grad) {
    if (!flat_grad) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(gnn_mlp_compute_gradient(h, X, Y, B));
    return get_flat(h, h->G, flat_grad);
}

int gnn_mlp_gradient_step(gnn_mlp_t *h, const double *X, const double *Y, int B, double step, double momentum,
                          int noise) {
    TRY(check_handle(h));
    if (!X || !Y) return fail(GNN_ERR_BAD_ARG, "null argument (reference: assert batch != null, SCE:299)");
    TRY(check_batch(h, B));
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is NaN-producing in the reference (SCE:335 sqrt of a negative draw) and is not built on the GPU");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    const int Lm = h->L - 1;
    TRY(stage_rows(h, X, h->dims[0], h->ld[0], B, h->stage_x, h->act[0], true));
    TRY(stage_rows(h, Y, h->dims[Lm], h->ld[Lm], B, h->stage_y, h->ybuf, false));
    // the staging buffers are reused by the next call: pageable hipMemcpyAsync has returned
    // only once the host data was consumed, and the convert kernels are stream-ordered.
    h->have_next = false; // (a hint refers to dataset rows; this batch came from the host)
    return step_on_rows(h, h->act[0], h->ybuf, B, step, momentum, false);
}

int gnn_mlp_get_weights(gnn_mlp_t *h, double *flat) { TRY(check_handle(h)); return get_flat(h, h->W, flat); }
int gnn_mlp_set_weights(gnn_mlp_t *h, const double *flat) { TRY(check_handle(h)); return set_flat(h, h->W, flat); }
int gnn_mlp_get_momentum(gnn_mlp_t *h, double *flat) { TRY(check_handle(h)); return get_flat(h, h->V, flat); }
int gnn_mlp_set_momentum(gnn_mlp_t *h, const double *flat) { TRY(check_handle(h)); return set_flat(h, h->V, flat); }

// ---- checkpoint -----------------------------------------------------------------------------
// File (little endian): "GNNMLP2\0", int32 L, int32 dims[L], int32 out_kind, inner_act, last_act, loss,
// dtype, int32 time, int64 n_params, fp64 weights[P], fp64 momentum[P], uint64 FNV-1a of every byte
// before it.  A file written for another net (dims OR any of the five enums) is refused, and so is a
// truncated or altered one.
namespace {
struct Fnv {
    uint64_t h = 1469598103934665603ull;
    void add(const void *p, size_t n) {
        const unsigned char *b = static_cast<const unsigned char *>(p);
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    }
};
bool put(FILE *f, Fnv &c, const void *p, size_t n) { c.add(p, n); return fwrite(p, 1, n, f) == n; }
bool get(FILE *f, Fnv &c, void *p, size_t n) { if (fread(p, 1, n, f) != n) return false; c.add(p, n); return true; }
const char kCkptMagic[8] = {'G', 'N', 'N', 'M', 'L', 'P', '2', 0};
} // namespace

int gnn_mlp_save_checkpoint(gnn_mlp_t *h, const char *path) {
    TRY(check_handle(h));
    if (!path) return fail(GNN_ERR_BAD_ARG, "null path");
    std::vector<double> w((size_t)h->n_params), v((size_t)h->n_params);
    TRY(get_flat(h, h->W, w.data()));
    TRY(get_flat(h, h->V, v.data()));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(GNN_ERR_BAD_ARG, std::string("cannot open ") + path);
    Fnv c;
    const int32_t L = h->L;
    bool ok = put(f, c, kCkptMagic, 8) && put(f, c, &L, 4);
    for (int l = 0; ok && l < L; l++) { const int32_t d = h->dims[l]; ok = put(f, c, &d, 4); }
    const int32_t cfg[6] = {h->out_kind, h->inner_act, h->last_act, h->loss, h->dtype, h->time};
    const int64_t np = h->n_params;
    ok = ok && put(f, c, cfg, sizeof cfg) && put(f, c, &np, 8) && put(f, c, w.data(), 8 * w.size()) &&
         put(f, c, v.data(), 8 * v.size());
    const uint64_t sum = c.h;
    ok = ok && fwrite(&sum, 8, 1, f) == 1;
    ok = (fclose(f) == 0) && ok;
    return ok ? GNN_OK : fail(GNN_ERR_BAD_ARG, std::string("short write to ") + path);
}

int gnn_mlp_load_checkpoint(gnn_mlp_t *h, const char *path) {
    TRY(check_handle(h));
    if (!path) return fail(GNN_ERR_BAD_ARG, "null path");
    FILE *f = fopen(path, "rb");
    if (!f) return fail(GNN_ERR_BAD_ARG, std::string("cannot open ") + path);
    Fnv c;
    char magic[8];
    int32_t L = 0, cfg[6] = {0, 0, 0, 0, 0, 0};
    int64_t np = 0;
    const char *why = nullptr;
    bool ok = get(f, c, magic, 8) && !memcmp(magic, kCkptMagic, 8) && get(f, c, &L, 4);
    if (!ok) why = "not a GNNMLP2 checkpoint";
    if (ok && L != h->L) { ok = false; why = "layer count differs"; }
    for (int l = 0; ok && l < L; l++) {
        int32_t d = 0;
        ok = get(f, c, &d, 4);
        if (ok && d != h->dims[l]) { ok = false; why = "layer dimensions differ"; }
    }
    if (ok) {
        ok = get(f, c, cfg, sizeof cfg) && get(f, c, &np, 8);
        if (!ok) why = "truncated header";
        else if (cfg[0] != h->out_kind || cfg[1] != h->inner_act || cfg[4] != h->dtype ||
                 (h->out_kind == GNN_OUT_ACT_LOSS && (cfg[2] != h->last_act || cfg[3] != h->loss))) {
            ok = false; why = "net configuration differs (output kind / activations / loss / dtype)";
        } else if (np != h->n_params || cfg[5] < 0) { ok = false; why = "parameter count differs"; }
    }
    std::vector<double> w, v;
    if (ok) {
        w.resize((size_t)h->n_params); v.resize((size_t)h->n_params);
        uint64_t sum = 0;
        ok = get(f, c, w.data(), 8 * w.size()) && get(f, c, v.data(), 8 * v.size());
        const uint64_t want = c.h;
        ok = ok && fread(&sum, 8, 1, f) == 1 && sum == want && fgetc(f) == EOF;
        if (!ok) why = "payload truncated, altered or followed by extra bytes (checksum)";
    }
    fclose(f);
    if (!ok) return fail(GNN_ERR_BAD_ARG, std::string(path) + ": " + (why ? why : "not a checkpoint of this net"));
    TRY(set_flat(h, h->W, w.data()));
    TRY(set_flat(h, h->V, v.data()));
    h->time = cfg[5];
    return GNN_OK;
}

// ---- dataset ------------------------------------------------------------------------------
static int alloc_dataset(gnn_mlp *h, int64_t N) {
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
}

int gnn_mlp_upload_dataset(gnn_mlp_t *h, const double *X, const double *Y, int64_t N) {
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
        hipLaunchKernelGGL(convert_rows_f64_kernel, dim3(grid_for(n * h->ld[0])), dim3(256), 0, h->stream, sx, d0,
                           h->DX + (size_t)r0 * h->ld[0], h->ld[0], n, n, h->inner_act, 1);
        hipLaunchKernelGGL(convert_rows_f64_kernel, dim3(grid_for(n * h->ld[h->L - 1])), dim3(256), 0, h->stream, sy,
                           dl, h->DY + (size_t)r0 * h->ld[h->L - 1], h->ld[h->L - 1], n, n, 0, 0);
        err = hipStreamSynchronize(h->stream); // the staging buffers are reused by the next chunk
    }
    if (err != hipSuccess) return fail(GNN_ERR_HIP, std::string("dataset upload: ") + hipGetErrorString(err));
    if (h->DXb) { to_bf16(h, h->DX, h->DXb, ((size_t)N + PAD) * h->ld[0]); HIP_TRY(hipStreamSynchronize(h->stream)); }
    TRY_LAUNCHES(h);
    h->dataset_n = N;
    return GNN_OK;
}

int gnn_mlp_upload_dataset_u8(gnn_mlp_t *h, const uint8_t *pixels, const uint8_t *labels, int64_t N) {
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
    hipLaunchKernelGGL(encode_u8_kernel, dim3(grid_for(N * h->ld[0])), dim3(256), 0, h->stream, sp, d0, h->DX, h->ld[0],
                       N, N, h->inner_act);
    hipLaunchKernelGGL(onehot_u8_kernel, dim3(grid_for(N * h->ld[h->L - 1])), dim3(256), 0, h->stream, sl, dl, h->DY,
                       h->ld[h->L - 1], N, N);
    HIP_TRY(hipStreamSynchronize(h->stream)); // the scratch buffers are released when this function returns
    if (h->DXb) { to_bf16(h, h->DX, h->DXb, ((size_t)N + PAD) * h->ld[0]); HIP_TRY(hipStreamSynchronize(h->stream)); }
    TRY_LAUNCHES(h);
    h->dataset_n = N;
    return GNN_OK;
}

static int check_step_args(gnn_mlp *h, int B, double step, int noise) {
    TRY(check_batch(h, B));
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    return GNN_OK;
}

int gnn_mlp_gradient_step_range(gnn_mlp_t *h, int64_t first, int B, double step, double momentum, int noise) {
    TRY(check_handle(h));
    TRY(check_step_args(h, B, step, noise));
    TRY(check_range(h, first, B));
    return step_on_rows(h, h->DX + (size_t)first * h->ld[0], h->DY + (size_t)first * h->ld[h->L - 1], B, step,
                        momentum, true);
}

// the next gradient computation runs on dataset rows [row0, row0 + B)
static void hint_range(gnn_mlp *h, int64_t row0, int B) {
    h->have_next = true; h->next_a0 = h->DX + (size_t)row0 * h->ld[0]; h->next_idx = nullptr; h->next_B = B;
}

int gnn_mlp_train_range(gnn_mlp_t *h, int64_t first, int B, int n_steps, double step, double momentum) {
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
    // per-kernel timing is on (timed launches carry events) or on a caller-provided stream (the
    // caller may be capturing itself).
    const bool want_graph = h->env_graph && !h->timing && h->stream == h->own_stream &&
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
        if (s + 1 < n_steps) hint_range(h, ((first / B + s + 1) % nb) * B, B); // the step's tile kernel also starts the next step
        TRY(step_on_rows(h, h->DX + (size_t)row0 * h->ld[0], h->DY + (size_t)row0 * h->ld[h->L - 1], B, step,
                         momentum, true));
    }
    return GNN_OK;
}

// inputs and expected rows of a sampled batch in ONE launch (nets off the fused path)
static void launch_gather(gnn_mlp *h, const int32_t *d_idx, int B) {
    const int B_pad = pad_up(B), Lm = h->L - 1;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)B_pad * (h->ld[0] + h->ld[Lm]) / 4)), dim3(256), 0, h->stream,
                       h->DX, h->ld[0], h->act[0], h->DY, h->ld[Lm], h->ybuf, d_idx, B, B_pad);
}

static int step_on_device_indices(gnn_mlp *h, const int32_t *d_idx, int B, double step, double momentum);

int gnn_mlp_gradient_step_indexed(gnn_mlp_t *h, const int32_t *idx, int B, double step, double momentum, int noise) {
    TRY(check_handle(h));
    if (!idx) return fail(GNN_ERR_BAD_ARG, "null index list");
    TRY(check_step_args(h, B, step, noise));
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    for (int i = 0; i < B; i++)
        if (idx[i] < 0 || idx[i] >= h->dataset_n) return fail(GNN_ERR_BAD_ARG, "sample index out of range");
    HIP_TRY(hipMemcpyAsync(h->idxbuf, idx, sizeof(int32_t) * (size_t)B, hipMemcpyHostToDevice, h->stream));
    h->slab_valid = false; h->have_next = false; // idxbuf is reused: its address does not identify a batch
    return step_on_device_indices(h, h->idxbuf, B, step, momentum);
}

int gnn_mlp_loss_range(gnn_mlp_t *h, int64_t first, int B, double *loss_per_sample) {
    TRY(check_handle(h));
    if (!loss_per_sample) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    do_forward(h, h->DX + (size_t)first * h->ld[0], h->DY + (size_t)first * h->ld[h->L - 1], B, false, true, false);
    TRY_LAUNCHES(h);
    return read_loss(h, B, loss_per_sample);
}

int gnn_mlp_argmax_range(gnn_mlp_t *h, int64_t first, int B, int32_t *labels) {
    TRY(check_handle(h));
    if (!labels) return fail(GNN_ERR_BAD_ARG, "null output");
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    do_forward(h, h->DX + (size_t)first * h->ld[0], nullptr, B, false, false, true);
    TRY_LAUNCHES(h);
    return read_labels(h, B, labels);
}

// ---- trainer-side sampling (NNT:143-168) ----------------------------------------------------
} // extern "C"

struct gnn_sampler {
    int32_t master = 0, remaining = 0;
    std::vector<int32_t> fen; // Fenwick tree over "row still in dataSampler": the r-th remaining
                              // row in master order is what ArrayList.get(r) returns after removals
    JavaRandom rnd{1};
    int log2n = 0;
    void refill() { // refillSampler NNT:164-168
        fen.assign((size_t)master + 1, 0);
        for (int32_t i = 1; i <= master; i++) {
            fen[i] += 1;
            const int32_t j = i + (i & -i);
            if (j <= master) fen[j] += fen[i];
        }
        remaining = master;
    }
    int32_t take(int32_t r) { // remove and return the r-th (0-based) remaining row
        int32_t pos = 0, k = r + 1;
        for (int32_t pw = 1 << log2n; pw > 0; pw >>= 1)
            if (pos + pw <= master && fen[pos + pw] < k) { pos += pw; k -= fen[pos]; }
        for (int32_t i = pos + 1; i <= master; i += i & -i) fen[i] -= 1;
        remaining--;
        return pos; // 0-based row
    }
};

extern "C" {

int gnn_sampler_create(int32_t master_size, int64_t seed, gnn_sampler_t **out) {
    if (!out || master_size <= 0) return fail(GNN_ERR_BAD_ARG, "bad sampler arguments");
    gnn_sampler *s = new gnn_sampler();
    s->master = master_size;
    s->rnd.set_seed(seed);
    while ((1 << (s->log2n + 1)) <= master_size) s->log2n++;
    s->refill();
    *out = s;
    return GNN_OK;
}

int gnn_sampler_destroy(gnn_sampler_t *s) { delete s; return GNN_OK; }

int gnn_sampler_sample(gnn_sampler_t *s, int batch, int32_t *out_idx, int *n_out) {
    if (!s || !out_idx || !n_out || batch <= 0) return fail(GNN_ERR_BAD_ARG, "bad sampler arguments");
    int n = 0;
    for (int i = 0; i < batch; i++) {
        if (s->remaining == 0) s->refill();                       // NNT:149-151
        const int32_t r = s->rnd.next_int(s->remaining);           // NNT:152
        const int32_t row = s->take(r);                            // NNT:153-154
        bool dup = false;                                          // HashMap.put, NNT:155
        for (int k = 0; k < n; k++) if (out_idx[k] == row) { dup = true; break; }
        if (!dup) out_idx[n++] = row;
    }
    *n_out = n;
    return GNN_OK;
}

static int step_on_device_indices(gnn_mlp *h, const int32_t *d_idx, int B, double step, double momentum) {
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

int gnn_mlp_train_sampled(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step, double momentum,
                          int noise) {
    TRY(check_handle(h));
    if (!s) return fail(GNN_ERR_BAD_ARG, "null sampler");
    TRY(check_step_args(h, batch, step, noise));
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    if (iterations <= 0) return fail(GNN_ERR_BAD_ARG, "iterations must be positive (NNT:62)");
    if (s->master != h->dataset_n) return fail(GNN_ERR_BAD_ARG, "sampler size differs from the dataset");
    if (batch >= s->master) return fail(GNN_ERR_BAD_ARG, "batchSize must be below the data size (NNT:63)");
    if (iterations >= 64) try_specialize(h);
    // The exact epoch sampler is serial host work (~10 us per batch of 128: two Fenwick walks per
    // draw) of the same order as a step on the GPU, so it runs AHEAD on a worker thread, chunk by
    // chunk, while this thread uploads finished chunks and enqueues their steps.  The first chunks are short
    // (16, 32, 64, 128, then 256 iterations): nothing runs on the GPU until the first one is sampled, and a 256-batch
    // first chunk kept it idle for 2.5 ms (0.85 us per step of a 3 000-step call).
    std::vector<int> bounds{0};
    for (int sz = 16; bounds.back() < iterations; sz = std::min(256, sz * 2)) bounds.push_back(std::min(iterations, bounds.back() + sz));
    const int n_chunks = (int)bounds.size() - 1;
    std::vector<int32_t> idx((size_t)iterations * batch);
    std::vector<int> cnt((size_t)iterations);
    std::mutex mu;
    std::condition_variable cv;
    int ready = 0, sampler_rc = GNN_OK; // chunks sampled so far (guarded by mu)
    std::string sampler_msg;
    std::thread producer([&]() {
        for (int c = 0; c < n_chunks; c++) {
            int rc = GNN_OK;
            const int i1 = bounds[c + 1];
            for (int i = bounds[c]; i < i1 && rc == GNN_OK; i++) rc = gnn_sampler_sample(s, batch, idx.data() + (size_t)i * batch, &cnt[i]);
            std::lock_guard<std::mutex> lk(mu);
            if (rc != GNN_OK) { sampler_rc = rc; sampler_msg = gnn_mlp_last_error(); ready = n_chunks; cv.notify_all(); return; }
            ready = c + 1;
            cv.notify_all();
        }
    });
    int32_t *d_idx = nullptr;
    int rc = GNN_OK;
    if (hipMalloc((void **)&d_idx, idx.size() * sizeof(int32_t)) != hipSuccess) rc = fail(GNN_ERR_HIP, "hipMalloc of the index buffer failed");
    for (int c = 0; c < n_chunks && rc == GNN_OK; c++) {
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return ready > c; });
            if (sampler_rc != GNN_OK) { rc = fail(sampler_rc, sampler_msg); break; }
        }
        const int i0 = bounds[c], i1 = bounds[c + 1];
        // (pageable hipMemcpyAsync returns once the host data has been consumed)
        const hipError_t e = hipMemcpyAsync(d_idx + (size_t)i0 * batch, idx.data() + (size_t)i0 * batch,
                                            (size_t)(i1 - i0) * batch * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
        if (e != hipSuccess) { rc = fail(GNN_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e)); break; }
        for (int i = i0; i < i1 && rc == GNN_OK; i++) {
            if (h->chain && i + 1 < i1) { // the next draw of this chunk is already on the device
                h->have_next = true; h->next_a0 = h->DX; h->next_idx = d_idx + (size_t)(i + 1) * batch; h->next_B = cnt[i + 1];
            }
            rc = step_on_device_indices(h, d_idx + (size_t)i * batch, cnt[i], step, momentum);
        }
    }
    producer.join(); // (on an early exit the sampler still finishes its draws: its state stays well defined)
    (void)hipStreamSynchronize(h->stream); // idx (host) and d_idx are released below
    h->slab_valid = false; h->have_next = false; // (they may name rows through d_idx)
    if (d_idx) (void)hipFree(d_idx);
    return rc;
}

// ---- data-parallel hooks --------------------------------------------------------------------
int gnn_mlp_grad_device_ptr(gnn_mlp_t *h, void **dev_ptr) {
    if (!h || !dev_ptr) return fail(GNN_ERR_BAD_ARG, "null argument");
    *dev_ptr = h->G;
    return GNN_OK;
}

int gnn_mlp_bind_grad_buffer(gnn_mlp_t *h, void *dev_ptr, int64_t n_elems) {
    TRY(check_handle(h));
    if (!dev_ptr) { h->G = h->G_own; return GNN_OK; }
    if (n_elems < h->n_pad) return fail(GNN_ERR_BAD_ARG, "gradient buffer shorter than gnn_mlp_grad_elems()");
    if (reinterpret_cast<uintptr_t>(dev_ptr) % 16) return fail(GNN_ERR_BAD_ARG, "gradient buffer must be 16-byte aligned");
    h->G = static_cast<float *>(dev_ptr);
    return GNN_OK;
}

int gnn_mlp_set_stream(gnn_mlp_t *h, void *hip_stream) {
    TRY(check_handle(h));
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(h->stream, &st) != hipSuccess || st != hipStreamCaptureStatusNone;
    if (!capturing) HIP_TRY(hipStreamSynchronize(h->stream)); // (a capturing stream cannot be waited on)
    else (void)hipGetLastError();
    h->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : h->own_stream;
    return GNN_OK;
}

int gnn_mlp_compute_gradient_range(gnn_mlp_t *h, int64_t first, int B) {
    TRY(check_handle(h));
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    const float *a0 = h->DX + (size_t)first * h->ld[0];
    maybe_specialize(h);
    do_gradient(h, a0, h->DY + (size_t)first * h->ld[h->L - 1], B, false, 0.f, 0.f, true);
    TRY_LAUNCHES(h);
    return GNN_OK;
}

int gnn_mlp_hint_next_range(gnn_mlp_t *h, int64_t first, int B) {
    TRY(check_handle(h));
    TRY(check_batch(h, B));
    TRY(check_range(h, first, B));
    hint_range(h, first, B);
    return GNN_OK;
}

int gnn_mlp_apply_update(gnn_mlp_t *h, int B_global, double step, double momentum) {
    TRY(check_handle(h));
    if (B_global <= 0) return fail(GNN_ERR_BAD_ARG, "B_global must be positive");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    NextBatch nb{};
    if (h->chain && take_next(h, &nb)) {
        // the update by weight tiles, each tile going straight on to the next batch's first-layer slab
        launch_tile_step(h, 2, 2, &nb, nullptr, PAD, (float)(step / (double)B_global), (float)momentum);
        slabs_now_hold(h, nb, nb.idx != nullptr);
    } else {
        const int64_t n4 = h->n_pad / 4;
        launch_timed(h, GNN_K_UPDATE, sgd_momentum_kernel, dim3(grid_for(n4)), dim3(256), 0,
                     SgdParams{reinterpret_cast<float4 *>(h->W), reinterpret_cast<float4 *>(h->V),
                               reinterpret_cast<const float4 *>(h->G), n4, (float)(step / (double)B_global), (float)momentum,
                               reinterpret_cast<sgd_bf16x4 *>(h->Wb)});
        h->slab_valid = false; h->have_next = false;
    }
    h->time++;
    TRY_LAUNCHES(h);
    return GNN_OK;
}

int gnn_mlp_advance_time(gnn_mlp_t *h, int steps) {
    if (!h || h->time + steps < 0) return fail(GNN_ERR_BAD_ARG, "bad argument");
    h->time += steps; // negative: steps that were only CAPTURED (enqueued into a graph, not run)
    return GNN_OK;
}

int gnn_mlp_recover_stream(gnn_mlp_t *h) {
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
}

int gnn_mlp_synchronize(gnn_mlp_t *h) {
    TRY(check_handle(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GNN_OK;
}

// ---- shape specialisation ---------------------------------------------------------------------
int gnn_mlp_specialize(gnn_mlp_t *h) {
    TRY(check_handle(h));
    h->jit_tried = false;
    try_specialize(h);
    return GNN_OK;
}
int gnn_mlp_specialization(const gnn_mlp_t *h) { return h ? h->specialization : -1; }
int gnn_mlp_step_launches(const gnn_mlp_t *h) { return !h ? -1 : h->chain ? 2 : h->mid4 ? 3 : 0; }

// ---- measurement ---------------------------------------------------------------------------
int gnn_mlp_timing_enable(gnn_mlp_t *h, int on) {
    TRY(check_handle(h));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->timing = on != 0;
    for (TimerClass &t : h->timers) t.used = 0;
    return GNN_OK;
}

int gnn_mlp_timing_read(gnn_mlp_t *h, int which, double *mean_us, int64_t *count) {
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
}

} // extern "C"

// =====================================================================================================
// One handle, N device replicas (include/gnn_mlp.h "data parallel inside the library"; csrc/dp_handle.h)
// =====================================================================================================
#include "dp_handle.h"

#include <dlfcn.h>

namespace {

// RCCL is bound at run time: libgnn_mlp_hip.so itself has no link dependency on it
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::mutex mu;
    bool load(std::string *why) {
        std::lock_guard<std::mutex> lock(mu);
        if (lib) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { *why = std::string("RCCL not found: ") + dlerror(); return false; }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !CommDestroy || !AllReduce || !GroupStart || !GroupEnd) { *why = "RCCL lacks the expected entry points"; return false; }
        return true;
    }
};
RcclApi g_rccl;
constexpr int kNcclFloat = 7, kNcclSum = 0; // ncclFloat32 / ncclSum (rccl.h)

} // namespace

struct gnn_mlp_dp {
    int n = 0, reducer = 0, max_batch = 0;
    std::vector<gnn_mlp *> rep;
    std::vector<int> dev;
    std::vector<void *> comm;                     // GNN_REDUCE_RCCL
    std::vector<float *> gbuf[2];                 // GNN_REDUCE_DIRECT: gradient buffers by step parity
    std::vector<hipEvent_t> grad_done[2], red_done[2];
    int64_t steps = 0;
};

namespace {

void dp_shard(int B, int r, int n, int *lo, int *hi) { // contiguous row blocks; the first B % n replicas get one more
    const int base = B / n, extra = B % n;
    *lo = r * base + (r < extra ? r : extra);
    *hi = *lo + base + (r < extra ? 1 : 0);
}

int dp_rccl_fail(int rc, const char *what) {
    return fail(GNN_ERR_HIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}

// after every replica's partial gradient is enqueued: sum across replicas + update
int dp_reduce_and_update(gnn_mlp_dp *d, int B_global, double step, double momentum) {
    const int n = d->n;
    if (d->reducer == GNN_REDUCE_RCCL) {
        if (n > 1) {
            int rc = g_rccl.GroupStart();
            if (rc) return dp_rccl_fail(rc, "ncclGroupStart");
            for (int r = 0; r < n; r++) {
                HIP_TRY(hipSetDevice(d->dev[r]));
                rc = g_rccl.AllReduce(d->rep[r]->G, d->rep[r]->G, (size_t)d->rep[r]->n_pad, kNcclFloat, kNcclSum, d->comm[r], d->rep[r]->stream);
                if (rc) { (void)g_rccl.GroupEnd(); return dp_rccl_fail(rc, "ncclAllReduce"); }
            }
            rc = g_rccl.GroupEnd();
            if (rc) return dp_rccl_fail(rc, "ncclGroupEnd");
        } else {
            const int rc = g_rccl.AllReduce(d->rep[0]->G, d->rep[0]->G, (size_t)d->rep[0]->n_pad, kNcclFloat, kNcclSum, d->comm[0], d->rep[0]->stream);
            if (rc) return dp_rccl_fail(rc, "ncclAllReduce");
        }
        for (int r = 0; r < n; r++) TRY(gnn_mlp_apply_update(d->rep[r], B_global, step, momentum));
        return GNN_OK;
    }
    // direct: events order the devices, one kernel per replica reads every partial gradient
    const int par = (int)(d->steps & 1);
    for (int r = 0; r < n; r++) {
        gnn_mlp *h = d->rep[r];
        HIP_TRY(hipSetDevice(d->dev[r]));
        HIP_TRY(hipEventRecord(d->grad_done[par][r], h->stream));
    }
    for (int r = 0; r < n; r++) {
        gnn_mlp *h = d->rep[r];
        HIP_TRY(hipSetDevice(d->dev[r]));
        for (int j = 0; j < n; j++)
            if (j != r) HIP_TRY(hipStreamWaitEvent(h->stream, d->grad_done[par][j], 0));
        DirectReduceParams p{};
        for (int j = 0; j < n; j++) p.G[j] = reinterpret_cast<const float4 *>(d->gbuf[par][j]);
        p.n = n;
        p.W = reinterpret_cast<float4 *>(h->W); p.V = reinterpret_cast<float4 *>(h->V);
        p.Wb = reinterpret_cast<sgd_bf16x4 *>(h->Wb);
        p.n4 = h->n_pad / 4;
        p.step_over_b = (float)(step / (double)B_global); p.momentum = (float)momentum;
        launch_timed(h, GNN_K_UPDATE, direct_reduce_update_kernel, dim3(grid_for(p.n4)), dim3(256), 0, p);
        HIP_TRY(hipEventRecord(d->red_done[par][r], h->stream));
        h->time++;
        h->slab_valid = false; h->have_next = false;
        TRY_LAUNCHES(h);
    }
    return GNN_OK;
}

// before a replica writes its parity buffer again: every peer has finished reading it (two steps ago)
int dp_direct_begin(gnn_mlp_dp *d, int r) {
    const int par = (int)(d->steps & 1);
    gnn_mlp *h = d->rep[r];
    TRY(gnn_mlp_bind_grad_buffer(h, d->gbuf[par][r], h->n_pad));
    if (d->steps >= 2)
        for (int j = 0; j < d->n; j++)
            if (j != r) HIP_TRY(hipStreamWaitEvent(h->stream, d->red_done[par][j], 0));
    return GNN_OK;
}

int dp_check(const gnn_mlp_dp *d) { return d ? GNN_OK : fail(GNN_ERR_BAD_ARG, "null handle"); }

} // namespace

extern "C" {

int gnn_mlp_dp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act, int loss, int64_t seed,
                      int dtype, const int32_t *devices, int n_dev, int max_batch, int reducer, gnn_mlp_dp_t **out) {
    if (!out) return fail(GNN_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > DP_MAX_REPLICAS) return fail(GNN_ERR_BAD_ARG, "n_dev must be 1..16");
    if (reducer != GNN_REDUCE_RCCL && reducer != GNN_REDUCE_DIRECT) return fail(GNN_ERR_BAD_ARG, "bad reducer");
    if (max_batch <= 0) return fail(GNN_ERR_BAD_ARG, "max_batch must be positive");
    if (reducer == GNN_REDUCE_RCCL) {
        for (int i = 0; i < n_dev; i++)
            for (int j = 0; j < i; j++)
                if (devices[i] == devices[j]) return fail(GNN_ERR_BAD_ARG, "RCCL needs one distinct device per replica (GNN_REDUCE_DIRECT accepts repeats)");
        std::string why;
        if (!g_rccl.load(&why)) return fail(GNN_ERR_UNSUPPORTED, why);
    }
    gnn_mlp_dp *d = new gnn_mlp_dp();
    d->n = n_dev; d->reducer = reducer; d->max_batch = max_batch;
    d->dev.assign(devices, devices + n_dev);
    auto cleanup = [&](int rc) { gnn_mlp_dp_destroy(d); return rc; };
    for (int r = 0; r < n_dev; r++) {
        gnn_mlp *h = nullptr;
        // (sized for the whole batch, not for a shard: propagate / loss / argmax on a replica take any batch the handle takes)
        const int rc = gnn_mlp_create(dims, n_dims, out_kind, inner_act, last_act, loss, seed, dtype, devices[r], max_batch, &h);
        if (rc != GNN_OK) return cleanup(rc);
        d->rep.push_back(h);
    }
    if (reducer == GNN_REDUCE_RCCL) {
        d->comm.assign(n_dev, nullptr);
        std::vector<int> devs(d->dev.begin(), d->dev.end());
        const int rc = g_rccl.CommInitAll(d->comm.data(), n_dev, devs.data());
        if (rc) { d->comm.clear(); return cleanup(dp_rccl_fail(rc, "ncclCommInitAll")); }
    } else {
        for (int r = 0; r < n_dev; r++) {
            if (hipSetDevice(d->dev[r]) != hipSuccess) return cleanup(fail(GNN_ERR_HIP, "hipSetDevice"));
            for (int j = 0; j < n_dev; j++) {
                if (d->dev[j] == d->dev[r]) continue;
                int can = 0;
                if (hipDeviceCanAccessPeer(&can, d->dev[r], d->dev[j]) != hipSuccess || !can)
                    return cleanup(fail(GNN_ERR_UNSUPPORTED, "the devices of a direct reducer must be peer-accessible"));
                const hipError_t e = hipDeviceEnablePeerAccess(d->dev[j], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return cleanup(fail(GNN_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e)));
                (void)hipGetLastError();
            }
            for (int par = 0; par < 2; par++) {
                float *g = nullptr;
                const int rc = dev_alloc(&g, (size_t)d->rep[r]->n_pad, d->rep[r]->stream);
                if (rc != GNN_OK) return cleanup(rc);
                d->gbuf[par].push_back(g);
                hipEvent_t a = nullptr, b = nullptr;
                if (hipEventCreateWithFlags(&a, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&b, hipEventDisableTiming) != hipSuccess)
                    return cleanup(fail(GNN_ERR_HIP, "hipEventCreate"));
                d->grad_done[par].push_back(a); d->red_done[par].push_back(b);
            }
            if (hipStreamSynchronize(d->rep[r]->stream) != hipSuccess) return cleanup(fail(GNN_ERR_HIP, "hipStreamSynchronize"));
        }
    }
    *out = d;
    return GNN_OK;
}

int gnn_mlp_dp_destroy(gnn_mlp_dp_t *d) {
    if (!d) return GNN_OK;
    for (size_t r = 0; r < d->rep.size(); r++) {
        (void)hipSetDevice(d->dev[r]);
        if (d->rep[r]->stream) (void)hipStreamSynchronize(d->rep[r]->stream);
    }
    for (void *c : d->comm) if (c) (void)g_rccl.CommDestroy(c);
    for (int par = 0; par < 2; par++) {
        for (size_t r = 0; r < d->gbuf[par].size(); r++) {
            (void)hipSetDevice(d->dev[r]);
            (void)gnn_mlp_bind_grad_buffer(d->rep[r], nullptr, 0); // back to the replica's own buffer before ours goes away
            (void)hipFree(d->gbuf[par][r]);
        }
        for (hipEvent_t e : d->grad_done[par]) (void)hipEventDestroy(e);
        for (hipEvent_t e : d->red_done[par]) (void)hipEventDestroy(e);
    }
    for (gnn_mlp *h : d->rep) (void)gnn_mlp_destroy(h);
    delete d;
    return GNN_OK;
}

int gnn_mlp_dp_num_replicas(const gnn_mlp_dp_t *d) { return d ? d->n : -1; }

int gnn_mlp_dp_replica(gnn_mlp_dp_t *d, int r, gnn_mlp_t **out) {
    TRY(dp_check(d));
    if (!out || r < 0 || r >= d->n) return fail(GNN_ERR_BAD_ARG, "replica index out of range");
    *out = d->rep[r];
    return GNN_OK;
}

int gnn_mlp_dp_gradient_step(gnn_mlp_dp_t *d, const double *X, const double *Y, int B, double step, double momentum, int noise) {
    TRY(dp_check(d));
    if (!X || !Y) return fail(GNN_ERR_BAD_ARG, "null argument (reference: assert batch != null, SCE:299)");
    if (B <= 0) return fail(GNN_ERR_BAD_ARG, "batch must be non-empty (SCE:300)");
    if (B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B exceeds max_batch given to gnn_mlp_dp_create");
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    const int d0 = d->rep[0]->dims[0], dl = d->rep[0]->dims[d->rep[0]->L - 1];
    for (int r = 0; r < d->n; r++) {
        int lo, hi;
        dp_shard(B, r, d->n, &lo, &hi);
        gnn_mlp *h = d->rep[r];
        TRY(check_handle(h));
        if (d->reducer == GNN_REDUCE_DIRECT) TRY(dp_direct_begin(d, r));
        if (hi > lo) TRY(gnn_mlp_compute_gradient(h, X + (size_t)lo * d0, Y + (size_t)lo * dl, hi - lo));
        else HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * (size_t)h->n_pad, h->stream)); // no rows: a zero partial gradient
    }
    TRY(dp_reduce_and_update(d, B, step, momentum));
    d->steps++;
    return GNN_OK;
}

int gnn_mlp_dp_upload_dataset(gnn_mlp_dp_t *d, const double *X, const double *Y, int64_t N) {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_upload_dataset(d->rep[r], X, Y, N)); // every replica holds every row: any batch can be sharded
    return GNN_OK;
}

static int dp_step_range(gnn_mlp_dp *d, int64_t first, int B, double step, double momentum, int64_t next_first) {
    for (int r = 0; r < d->n; r++) {
        int lo, hi;
        dp_shard(B, r, d->n, &lo, &hi);
        gnn_mlp *h = d->rep[r];
        TRY(check_handle(h));
        if (d->reducer == GNN_REDUCE_DIRECT) TRY(dp_direct_begin(d, r));
        else if (next_first >= 0 && hi > lo) TRY(gnn_mlp_hint_next_range(h, next_first + lo, hi - lo)); // the update kernel then starts the next step
        if (hi > lo) TRY(gnn_mlp_compute_gradient_range(h, first + lo, hi - lo));
        else HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * (size_t)h->n_pad, h->stream));
    }
    TRY(dp_reduce_and_update(d, B, step, momentum));
    d->steps++;
    return GNN_OK;
}

int gnn_mlp_dp_gradient_step_range(gnn_mlp_dp_t *d, int64_t first, int B, double step, double momentum, int noise) {
    TRY(dp_check(d));
    if (B <= 0 || B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B out of range");
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    if (first < 0 || first + B > d->rep[0]->dataset_n) return fail(GNN_ERR_BAD_ARG, "dataset rows out of range");
    return dp_step_range(d, first, B, step, momentum, -1);
}

int gnn_mlp_dp_train_range(gnn_mlp_dp_t *d, int64_t first, int B, int n_steps, double step, double momentum) {
    TRY(dp_check(d));
    if (B <= 0 || B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B out of range");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    if (n_steps <= 0) return fail(GNN_ERR_BAD_ARG, "n_steps must be positive (NNT:62)");
    const int64_t nb = d->rep[0]->dataset_n / B;
    if (nb <= 0 || first < 0 || first % B != 0) return fail(GNN_ERR_BAD_ARG, "first must be a multiple of B inside the dataset");
    for (int s = 0; s < n_steps; s++) {
        const int64_t row0 = ((first / B + s) % nb) * B;
        const int64_t nxt = (s + 1 < n_steps) ? ((first / B + s + 1) % nb) * B : -1;
        TRY(dp_step_range(d, row0, B, step, momentum, nxt));
    }
    return GNN_OK;
}

int gnn_mlp_dp_set_weights(gnn_mlp_dp_t *d, const double *flat) {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_set_weights(d->rep[r], flat));
    return GNN_OK;
}

int gnn_mlp_dp_synchronize(gnn_mlp_dp_t *d) {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_synchronize(d->rep[r]));
    return GNN_OK;
}

int gnn_mlp_dp_replicas_identical(gnn_mlp_dp_t *d, int *identical) {
    TRY(dp_check(d));
    if (!identical) return fail(GNN_ERR_BAD_ARG, "null output");
    std::vector<double> w0((size_t)d->rep[0]->n_params), v0(w0.size()), w(w0.size()), v(w0.size());
    TRY(gnn_mlp_get_weights(d->rep[0], w0.data()));
    TRY(gnn_mlp_get_momentum(d->rep[0], v0.data()));
    *identical = 1;
    for (int r = 1; r < d->n && *identical; r++) {
        TRY(gnn_mlp_get_weights(d->rep[r], w.data()));
        TRY(gnn_mlp_get_momentum(d->rep[r], v.data()));
        if (memcmp(w.data(), w0.data(), w.size() * sizeof(double)) || memcmp(v.data(), v0.data(), v.size() * sizeof(double)) ||
            d->rep[r]->time != d->rep[0]->time) *identical = 0;
    }
    return GNN_OK;
}

} // extern "C"
