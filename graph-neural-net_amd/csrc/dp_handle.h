// dp_handle.h -- ONE handle, N device replicas: the data-parallel gradientStep inside the library.
//
// The reference's caller is one JVM thread calling NeuralNet.gradientStep (NNT:83); it cannot be
// "one process per GPU".  gnn_mlp_dp_* gives that caller the sharded step behind a single call:
// the batch's rows are dealt to the replicas in contiguous blocks (the sum at SCE:305-322 is the only
// cross-sample operation), every replica forms the partial gradient of its rows into its flat f32
// buffer, the buffers are summed across devices, and every replica applies the identical update with
// batchSize = the whole batch (SCE:333) -- so the replicas stay bitwise identical.
//
// Two reducers behind one interface:
//   GNN_REDUCE_RCCL    ncclCommInitAll over the devices, one ncclAllReduce(SUM) per replica per step
//                      inside ncclGroupStart/End (RCCL over xGMI), then the flat momentum update.
//                      RCCL is loaded with dlopen at create: the library itself does not link it.
//   GNN_REDUCE_DIRECT  no collective library: every replica's gradient buffer is peer-mapped
//                      (hipDeviceEnablePeerAccess); after an event of every peer's gradient kernel each
//                      replica runs ONE kernel that reads all N partial gradients in rank order, sums
//                      them and applies the update (direct_reduce_update_kernel): one pass over G less
//                      than all-reduce + update, one launch less, and for a ~1 MB message no ring.
//                      Gradient buffers are double-buffered by step parity so that a replica never
//                      rewrites a buffer a slower peer may still be reading.  Cross-device ordering
//                      uses stream events only (kernel-boundary visibility, no in-kernel flags).
//                      Replicas may share a device (devices = {0, 0}): that is how the reducer is
//                      tested on a one-GPU box.
//                      On a net that takes the two-launch step with the next batch known, the reduction is part of the
//                      tile-owner kernel (tile_step_kernel<GSRC = 3>: sum of the peers' tiles -> update -> next slabs).
//   GNN_REDUCE_DIRECT_RS  the same transport cut in two phases: replica r first reduces ONLY its slice r of the flat
//                      buffer from all peers (reduce-scatter by peer reads), then every replica gathers the reduced
//                      slices from their owners while it updates (tile_step_kernel<GSRC = 4>, or the flat
//                      gather-update kernel).  Every link then carries 2 P 4 / N bytes per step instead of
//                      (N-1) P 4 / ... per peer pair -- the form for wide nets; one more launch and one more event hop.
#pragma once
#include "kernels.h"

namespace gnn {

constexpr int DP_MAX_REPLICAS = 16;

struct DirectReduceParams {
    const float4 *G[DP_MAX_REPLICAS]; // every replica's partial gradient (peer pointers), rank order
    int n;
    float4 *W; float4 *V;             // this replica's masters
    sgd_bf16x4 *Wb;                   // bf16 mode: shadow of W (may be null)
    int64_t n4;
    float step_over_b, momentum;
};

// G = sum_r G_r in rank order (the same order on every replica: identical bits everywhere), then SCE:333-339.
static __global__ __launch_bounds__(256) void direct_reduce_update_kernel(DirectReduceParams p) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n4; i += (int64_t)gridDim.x * 256) {
        float4 g = p.G[0][i];
#pragma unroll
        for (int r = 1; r < DP_MAX_REPLICAS; r++) {
            if (r < p.n) {
                const float4 o = p.G[r][i];
                g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
            }
        }
        float4 v = p.V[i], w = p.W[i];
        v.x = sgd_adj(p.step_over_b, g.x, p.momentum, v.x);
        v.y = sgd_adj(p.step_over_b, g.y, p.momentum, v.y);
        v.z = sgd_adj(p.step_over_b, g.z, p.momentum, v.z);
        v.w = sgd_adj(p.step_over_b, g.w, p.momentum, v.w);
        w.x -= v.x; w.y -= v.y; w.z -= v.z; w.w -= v.w;
        p.V[i] = v;
        p.W[i] = w;
        if (p.Wb) p.Wb[i] = (sgd_bf16x4){(__bf16)w.x, (__bf16)w.y, (__bf16)w.z, (__bf16)w.w};
    }
}

// reduce-scatter by peer reads: this replica's slice [lo4, hi4) of the flat buffer, summed over the replicas in rank order
struct DirectScatterParams {
    const float4 *G[DP_MAX_REPLICAS];
    int n;
    float4 *red;          // this replica's reduced buffer (indexed like G; only its slice is written)
    int64_t lo4, hi4;
};
static __global__ __launch_bounds__(256) void direct_reduce_scatter_kernel(DirectScatterParams p) {
    for (int64_t i = p.lo4 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.hi4; i += (int64_t)gridDim.x * 256) {
        float4 g = p.G[0][i];
#pragma unroll
        for (int r = 1; r < DP_MAX_REPLICAS; r++) {
            if (r < p.n) {
                const float4 o = p.G[r][i];
                g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
            }
        }
        p.red[i] = g;
    }
}
// all-gather by peer reads + SCE:333-339: element i comes from the replica that owns its slice
struct DirectGatherParams {
    const float4 *red[DP_MAX_REPLICAS];
    int n;
    int64_t slice4;       // float4s per owner
    float4 *W; float4 *V;
    sgd_bf16x4 *Wb;
    int64_t n4;
    float step_over_b, momentum;
};
static __global__ __launch_bounds__(256) void direct_gather_update_kernel(DirectGatherParams p) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n4; i += (int64_t)gridDim.x * 256) {
        const int owner = (int)(i / p.slice4);
        const float4 *src = p.red[0];
#pragma unroll
        for (int r = 1; r < DP_MAX_REPLICAS; r++) src = (owner == r) ? p.red[r] : src;
        const float4 g = src[i];
        float4 v = p.V[i], w = p.W[i];
        v.x = sgd_adj(p.step_over_b, g.x, p.momentum, v.x);
        v.y = sgd_adj(p.step_over_b, g.y, p.momentum, v.y);
        v.z = sgd_adj(p.step_over_b, g.z, p.momentum, v.z);
        v.w = sgd_adj(p.step_over_b, g.w, p.momentum, v.w);
        w.x -= v.x; w.y -= v.y; w.z -= v.z; w.w -= v.w;
        p.V[i] = v;
        p.W[i] = w;
        if (p.Wb) p.Wb[i] = (sgd_bf16x4){(__bf16)w.x, (__bf16)w.y, (__bf16)w.z, (__bf16)w.w};
    }
}

} // namespace gnn
