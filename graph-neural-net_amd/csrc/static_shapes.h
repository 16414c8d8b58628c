// static_shapes.h -- the shapes with PREBUILT compile-time plans (BASELINE.json's two MNIST nets) and the tables of their
// kernel instances.  launch_small.hip instantiates the SoftmaxCrossEntropyNeuralNet forms (OUTK = 0), launch_small_gnn.hip
// the GeneralNeuralNet forms (OUTK = 1: last_act + loss, GNN:215-218, GNN:267-271) -- two translation units so that the two
// families compile side by side (each is ~70 s of hipcc).
#pragma once
#include "handle.h"

namespace gnn {
namespace host {

using ShapeMnistA = StaticShape<784, 300, 100, 10>;
using ShapeMnistB = StaticShape<784, 100, 50, 10>;
using RbMnistA = RbStaticShape<784, 300, 100, 10>;
using RbMnistB = RbStaticShape<784, 100, 50, 10>;

// middle4_kernel table: [shape policy][activation][output kind][variant]
// variant: 0 forward only, 1 forward + backward, 2 forward + backward with A_1 from the K slabs of tile_step_kernel, 3 = 2 in bf16
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

template <class SH, int OUTK, bool BF> const void *rb_fn_static(int act) {
    switch (act) {
    case 0: return reinterpret_cast<const void *>(&rowblock_kernel<SH, 0, OUTK, false, 0, BF>);
    case 1: return reinterpret_cast<const void *>(&rowblock_kernel<SH, 1, OUTK, false, 0, BF>);
    case 2: return reinterpret_cast<const void *>(&rowblock_kernel<SH, 2, OUTK, false, 0, BF>);
    case 3: return reinterpret_cast<const void *>(&rowblock_kernel<SH, 3, OUTK, false, 0, BF>);
    default: return reinterpret_cast<const void *>(&rowblock_kernel<SH, 4, OUTK, false, 0, BF>);
    }
}

// which: 0 = 784-300-100-10, 1 = 784-100-50-10
template <int OUTK> const void *mid4_static_table(int which, int act, int variant) {
    return which == 0 ? mid4_fn_sh<ShapeMnistA, OUTK>(act, variant) : mid4_fn_sh<ShapeMnistB, OUTK>(act, variant);
}
template <int OUTK> const void *rb_static_table(int which, int act, bool bf) {
    if (which == 0) return bf ? rb_fn_static<RbMnistA, OUTK, true>(act) : rb_fn_static<RbMnistA, OUTK, false>(act);
    return bf ? rb_fn_static<RbMnistB, OUTK, true>(act) : rb_fn_static<RbMnistB, OUTK, false>(act);
}

// launch_small_gnn.hip
const void *mid4_static_general(int which, int act, int variant);
const void *rb_static_general(int which, int act, bool bf);

} // namespace host
} // namespace gnn
