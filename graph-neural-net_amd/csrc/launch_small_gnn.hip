// launch_small_gnn.hip -- the PREBUILT GeneralNeuralNet instances (element-wise output: last_act + loss, GNN:215-218,
// GNN:237-239, GNN:267-271) of the small-net kernels for the two MNIST shapes BASELINE.json names: middle4_kernel (inference,
// three-launch step) and rowblock_kernel (the two-launch training step), f32 and bf16.  Until round 4 a GeneralNeuralNet of
// these shapes ran the runtime-shape instances until its 16th step and the hiprtc instance afterwards (jit.h); the static
// row-block kernel is 6.9 us where the runtime-shape form is 10.2.  A translation unit of its own: the
// instances compile beside launch_small.hip's.
#include "static_shapes.h"

namespace gnn {
namespace host {

const void *mid4_static_general(int which, int act, int variant) { return mid4_static_table<1>(which, act, variant); }
const void *rb_static_general(int which, int act, bool bf) { return rb_static_table<1>(which, act, bf); }

} // namespace host
} // namespace gnn
