// eval_kernels.h -- the two small kernels that keep the trainer's EVALUATION loops on the device:
//   count_hits_kernel   testOnTrainingData / testOnTestData (MT:159-197): hit when the `>=` argmax of propagate() (the labels the
//                       forward kernels wrote) equals the expected class, the LAST index whose expected value is 1 (MT:186-188);
//   sum_loss_kernel     validate (NNT:102-113): the sum of calculateLoss over the validation rows, into one fp64 slot.
// Integer counts are exact whatever the order; the loss sum has ONE fixed order (a thread's strided partial sums, then a tree
// in LDS), so a run gives the same bits every time.
#pragma once
#include "kernels.h"

namespace gnn {

struct HitsParams {
    const int32_t *label; // [rows] from the forward kernels (-1 for padded rows)
    const float *Y; int ldy; // expected rows (one-hot for MNIST, MT:112-118)
    int n_out, rows;
    unsigned long long *hits;
};
static __global__ __launch_bounds__(256) void count_hits_kernel(HitsParams p) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    bool hit = false;
    if (row < p.rows) {
        const float *y = p.Y + (size_t)row * p.ldy;
        int expected = 0;
        for (int i = 0; i < p.n_out; i++) if (y[i] == 1.f) expected = i; // MT:186-188
        hit = p.label[row] == expected;                                  // MT:195
    }
    const unsigned long long m = __ballot(hit);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(p.hits, (unsigned long long)__popcll(m));
}

struct LossSumParams {
    const float *loss; int rows;
    double *out; int accumulate; // out += sum (accumulate) or out = sum
};
static __global__ __launch_bounds__(256) void sum_loss_kernel(LossSumParams p) {
    __shared__ double part[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < p.rows; i += 256) s += (double)p.loss[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *p.out = (p.accumulate ? *p.out : 0.0) + part[0];
}

// One workgroup per ROW of a [n_rows][stride] matrix of per-sample losses: out[row] = the sum of its first `cols` entries, in fp64,
// one fixed order.  The observed training loop keeps every iteration's validation losses in such a matrix and sums them ONCE at
// the end of the call: a reduction launch per iteration (4.9 us each, profiles/r04/observed_loop_kernel_stats.csv) was 13 % of
// the loop.
struct RowSumParams {
    const float *rows; int64_t stride; int cols;
    double *out;
};
static __global__ __launch_bounds__(256) void sum_rows_kernel(RowSumParams p) {
    __shared__ double part[256];
    const float *r = p.rows + (size_t)blockIdx.x * p.stride;
    double s = 0.0;
    for (int i = threadIdx.x; i < p.cols; i += 256) s += (double)r[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) p.out[blockIdx.x] = part[0];
}

} // namespace gnn
