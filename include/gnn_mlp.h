/*
 * gnn_mlp.h -- C ABI of the MI355X-native (gfx950) MLP mini-batch SGD path.
 *
 * This is the drop-in boundary for the hot path of asheptunov/graph-neural-net: the
 * arithmetic of SoftmaxCrossEntropyNeuralNet / GeneralNeuralNet (the two implementations of
 * the `NeuralNet` operator interface) moves behind these entry points; the Java NeuralNet /
 * NeuralNetTrainer API surface above it stays.  The reference has no FFI of its own; the
 * functions below are what a JNI shim for its `NeuralNet` interface binds (INTEGRATION.md shows
 * the Java `native` declarations and the JNI stub for each).
 *
 * Citations are file:line under /root/reference/src.  NN = NeuralNet.java,
 * SCE = SoftmaxCrossEntropyNeuralNet.java, GNN = GeneralNeuralNet.java,
 * NNT = NeuralNetTrainer.java, MT = MNISTTrainer.java.
 *
 * Conventions
 *  - plain C types only; every function returns a gnn_status (0 = ok) and never throws or
 *    aborts across the ABI; gnn_mlp_last_error() gives the message of the calling thread's
 *    last failure.
 *  - host matrices are dense row-major fp64, exactly the reference's `double[]` rows laid end
 *    to end: X is B x d_0, Y is B x d_{L-1}.  The reference passes one sample (or a
 *    Map<double[],double[]>) per call; the shim flattens the Map in iteration order.
 *  - flat parameter vectors (weights, momentum, gradients) are layer-major, each layer
 *    row-major [in][out] like `weights.get(l)[in][out]` (SCE:44-47), UNPADDED, fp64 on the host.
 *  - the library never keeps a host pointer past the call (the reference aliases the caller's
 *    input array in neurons[0], SCE:167-168; no caller can observe that through NN:16-65).
 *  - one gnn_mlp_t = one GPU = one logical stream of calls (gnn_mlp_dp_t: one handle over N GPUs); calls on one handle must be serialised
 *    by the caller (the reference classes are not re-entrant either: `neurons` is shared scratch).
 *  - there is NO CPU fallback: with no gfx950 device every entry point fails with
 *    GNN_ERR_NO_DEVICE.
 */
#ifndef GNN_MLP_H
#define GNN_MLP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct gnn_mlp gnn_mlp_t;

typedef enum {
    GNN_OK = 0,
    GNN_ERR_BAD_ARG = 1,     /* null pointer, bad dims, B out of range ... (reference: `assert`) */
    GNN_ERR_HIP = 2,         /* a HIP runtime call failed; message holds hipGetErrorString */
    GNN_ERR_UNSUPPORTED = 3, /* e.g. noise=1 (SCE:334-336 is NaN-producing) */
    GNN_ERR_NO_DEVICE = 4,   /* no HIP device visible */
    GNN_ERR_STATE = 5        /* e.g. indexed step before a dataset was uploaded */
} gnn_status;

/* Closed enum standing in for the reference's ActivationFunction / ActivationPrime lambdas
 * (ActivationFunction.java:14, ActivationPrime.java:14).  LEAKY_RELU is the pair the
 * reference ships: a>0 ? a : 0.01a  and  a<=0 ? 0.01 : 1.0  (MT:234-235). */
typedef enum {
    GNN_ACT_LEAKY_RELU = 0,
    GNN_ACT_SIGMOID = 1,
    GNN_ACT_TANH = 2,
    GNN_ACT_RELU = 3,
    GNN_ACT_IDENTITY = 4
} gnn_act;

typedef enum {
    GNN_OUT_SOFTMAX_CE = 0, /* SoftmaxCrossEntropyNeuralNet: softmax + cross entropy (SCE:197,216,250) */
    GNN_OUT_ACT_LOSS = 1    /* GeneralNeuralNet: last_act + loss (GNN:215-218,238,267-271) */
} gnn_out_kind;

typedef enum {
    GNN_LOSS_HALF_SQUARED = 0 /* 0.5*(a-y)^2, derivative (a-y)  (LossFunction.java:16) */
} gnn_loss;

typedef enum {
    GNN_DTYPE_F32 = 0, /* f32 operands, f32 MFMA (v_mfma_f32_16x16x4_f32), f32 accumulate */
    GNN_DTYPE_BF16 = 1 /* bf16 GEMM operands, f32 accumulate, f32 master weights + momentum */
} gnn_dtype;

/* ---- construction (SCE:103-128 / GNN:112-147) ------------------------------------------- */

/* Builds the net on HIP device `device` and draws the initial weights exactly as
 * appendLayer does (SCE:139-156): java.util.Random(seed), layer by layer, row-major,
 * nextDouble() - 0.5; momentum ("previousUpdate") zero; time 0.  The reference hard-codes
 * seed 1 (SCE:111).  `last_act`/`loss` are ignored for GNN_OUT_SOFTMAX_CE.
 * `max_batch` bounds B of every later call (device buffers are sized once, here). */
int gnn_mlp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act,
                   int loss, int64_t seed, int dtype, int device, int max_batch,
                   gnn_mlp_t **out);
int gnn_mlp_destroy(gnn_mlp_t *h);

int gnn_mlp_input_dim(const gnn_mlp_t *h);    /* NN:58 getInputDim  (SCE:383) */
int gnn_mlp_output_dim(const gnn_mlp_t *h);   /* NN:65 getOutputDim (SCE:392) */
int64_t gnn_mlp_num_params(const gnn_mlp_t *h); /* sum_l d_l*d_{l+1} */
int gnn_mlp_time(const gnn_mlp_t *h);         /* `time`: gradient steps taken (SCE:23,343) */
const char *gnn_mlp_last_error(void);

/* ---- the NeuralNet operator interface, batched ------------------------------------------ */

/* NN:16 propagate (SCE:164-198, GNN:183-221) for B samples: out is B x d_{L-1}
 * (softmax probabilities, or last_act(z) for GNN_OUT_ACT_LOSS).  Synchronous. */
int gnn_mlp_propagate(gnn_mlp_t *h, const double *X, int B, double *out);

/* NN:27 calculateLoss (SCE:207-220, GNN:230-242): loss_per_sample has B entries. */
int gnn_mlp_loss(gnn_mlp_t *h, const double *X, const double *Y, int B, double *loss_per_sample);

/* NN:39 calculateWeightGradient (SCE:229-287, GNN:251-307), summed over the B samples
 * (B = 1 reproduces the reference call); flat_grad has gnn_mlp_num_params entries. */
int gnn_mlp_weight_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B,
                            double *flat_grad);

/* NN:51 gradientStep (SCE:297-346, GNN:317-366): G = sum over the batch of the per-sample
 * gradients; adj = step*G/B + momentum*prev; W -= adj; prev = adj; time++.
 * noise != 0 -> GNN_ERR_UNSUPPORTED.  Returns after enqueueing (asynchronous).  On the small-net path the weight UPDATE of a
 * call is enqueued with the NEXT gradient_step call (one kernel then applies it and starts the new batch: three dependent
 * launches per call instead of four), or by whichever other entry point of the handle is called first -- every entry point
 * sees the updated weights, `time` counts the step at the call, results are the same bit for bit (GNN_MLP_DEFER=0 switches
 * it off). */
int gnn_mlp_gradient_step(gnn_mlp_t *h, const double *X, const double *Y, int B, double step,
                          double momentum, int noise);

/* MT:166-168 / MT:191-193 argmax of propagate(): `>=`, so ties go to the HIGHEST index. */
int gnn_mlp_argmax(gnn_mlp_t *h, const double *X, int B, int32_t *labels);

/* ---- parameter access (extension: `weights` is private with no getter, SCE:15) ----------- */
int gnn_mlp_get_weights(gnn_mlp_t *h, double *flat);
int gnn_mlp_set_weights(gnn_mlp_t *h, const double *flat);
int gnn_mlp_get_momentum(gnn_mlp_t *h, double *flat);
int gnn_mlp_set_momentum(gnn_mlp_t *h, const double *flat);

/* Checkpoint (the reference has no persistence at all: a trained net is lost at JVM exit,
 * SURVEY 5).  File, little endian: "GNNMLP2\0", int32 L, int32 dims[L], int32 out_kind, inner_act,
 * last_act, loss, dtype, int32 time, int64 P, fp64 weights[P], fp64 momentum[P] (flat layer-major
 * row-major [in][out] like get_weights), uint64 FNV-1a of all preceding bytes.  Loading requires
 * identical dims AND configuration (a sigmoid / bf16 / GeneralNeuralNet file is refused by a
 * leaky-ReLU / f32 / softmax net of the same dims) and an intact payload (length and checksum). */
int gnn_mlp_save_checkpoint(gnn_mlp_t *h, const char *path);
int gnn_mlp_load_checkpoint(gnn_mlp_t *h, const char *path);

/* ---- device-resident training data (the caller of the path: NNT:28-43, NNT:143-168) ------ */

/* Copies N samples to HBM once (fp64 -> compute dtype, inner activation applied to the input
 * as SCE:183-186 does for layer 0).  Later *_indexed / *_range calls do no host->device copy
 * of sample data. */
int gnn_mlp_upload_dataset(gnn_mlp_t *h, const double *X, const double *Y, int64_t N);
/* Same from raw IDX payloads with the reference's encoding done on the GPU:
 * pixel -> (byte & 0xff)/255.0 (MT:98), label -> one-hot 1.0 (MT:112-118). */
int gnn_mlp_upload_dataset_u8(gnn_mlp_t *h, const uint8_t *pixels, const uint8_t *labels,
                              int64_t N);
int64_t gnn_mlp_dataset_size(const gnn_mlp_t *h);

/* gradientStep on dataset rows idx[0..B) (host int32 indices, e.g. one NNT.sample draw). */
int gnn_mlp_gradient_step_indexed(gnn_mlp_t *h, const int32_t *idx, int B, double step,
                                  double momentum, int noise);
/* gradientStep on the contiguous dataset rows [first, first+B). */
int gnn_mlp_gradient_step_range(gnn_mlp_t *h, int64_t first, int B, double step,
                                double momentum, int noise);
/* n_steps consecutive gradientSteps (the loop NNT:82-85) over rows
 * [first + s*B, first + (s+1)*B) mod the rows that fit, s = 0..n_steps-1, with no host work
 * between steps. */
int gnn_mlp_train_range(gnn_mlp_t *h, int64_t first, int B, int n_steps, double step,
                        double momentum);
int gnn_mlp_loss_range(gnn_mlp_t *h, int64_t first, int B, double *loss_per_sample);
int gnn_mlp_argmax_range(gnn_mlp_t *h, int64_t first, int B, int32_t *labels);
/* testOnTrainingData / testOnTestData (MT:159-197) over the dataset rows [first, first + n), any n: propagate + the `>=`
 * argmax (MT:166-168) per row, a hit when it equals the expected class -- the LAST index whose expected value is 1
 * (MT:186-188; for the one-hot rows of MT:112-118 that is the label).  The rows are walked in blocks of max_batch with no host
 * work in between; ONE count comes back.  The reference returns hits / size (MT:172, 197): the division is the caller's. */
int gnn_mlp_count_hits_range(gnn_mlp_t *h, int64_t first, int64_t n, int64_t *hits);

/* ---- the trainer's sampling loop (NeuralNetTrainer.java) ------------------------------------ */

/* Epoch sampler without replacement, NNT:28-43 + NNT:143-168: java.util.Random(seed) (the
 * reference uses 1, NNT:42), nextInt(remaining) picks the r-th REMAINING sample in master order
 * (ArrayList.get(r) + remove(r), NNT:152-154), the list refills when empty -- also in the middle
 * of a batch (NNT:149-151), in which case a sample drawn twice collapses in the reference's
 * HashMap (NNT:155) and the batch is shorter than requested.  Master order here is
 * dataset row order (the reference's is HashMap iteration order, i.e. unspecified). */
typedef struct gnn_sampler gnn_sampler_t;
int gnn_sampler_create(int32_t master_size, int64_t seed, gnn_sampler_t **out);
int gnn_sampler_destroy(gnn_sampler_t *s);
/* One sample(batchSize) call: writes the DISTINCT rows in first-draw order, *n_out of them. */
int gnn_sampler_sample(gnn_sampler_t *s, int batch, int32_t *out_idx, int *n_out);
/* The loop NNT:82-85 / NNT:88-90: `iterations` times { sample(batch); gradientStep }. All draws
 * are made up front and uploaded once; the steps are then enqueued with no host->device copy. */
int gnn_mlp_train_sampled(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step,
                          double momentum, int noise);
/* The OBSERVED loops NNT:68-72 / NNT:75-79: `iterations` times { sample(batch); gradientStep; validate(validation_size) }.
 * validate (NNT:102-113) is the mean of calculateLoss over the first validation_size samples in master order -- here dataset
 * rows [0, validation_size).  Steps and validation passes are enqueued back to back; the summed losses stay in a device buffer
 * and come back ONCE: val_loss[i] = validate(validation_size) after iteration i (the value the reference prints as "%d,%.2f",
 * NNT:71).  Needs 0 < validation_size <= the dataset's size. */
int gnn_mlp_train_sampled_observed(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step,
                                   double momentum, int noise, int validation_size, double *val_loss);

/* ---- data-parallel hooks (one process per GPU; the exchange is the caller's collective) -- */

/* The reference sums per-sample gradients at SCE:305-322 and divides by batch.size() at
 * SCE:333.  Sharded over ranks that sum becomes: each rank's partial G (its rows only) ->
 * all-reduce(SUM) of the flat device gradient buffer -> identical update on every rank with
 * B_global.  The buffer is fp32, gnn_mlp_grad_elems() long (padded layout, pads are zero). */
int64_t gnn_mlp_grad_elems(const gnn_mlp_t *h);
int gnn_mlp_grad_device_ptr(gnn_mlp_t *h, void **dev_ptr);
/* Use a caller-owned device buffer (e.g. a torch tensor registered with RCCL) instead. */
int gnn_mlp_bind_grad_buffer(gnn_mlp_t *h, void *dev_ptr, int64_t n_elems);
/* Run all kernels on the caller's hipStream_t (e.g. torch's current stream); NULL = own. */
int gnn_mlp_set_stream(gnn_mlp_t *h, void *hip_stream);
/* forward + backward of dataset rows [first, first+B_local) into the gradient buffer. */
int gnn_mlp_compute_gradient_range(gnn_mlp_t *h, int64_t first, int B_local);
int gnn_mlp_compute_gradient(gnn_mlp_t *h, const double *X, const double *Y, int B_local);
/* SCE:324-344 on the (all-reduced) gradient buffer with batchSize = B_global. */
int gnn_mlp_apply_update(gnn_mlp_t *h, int B_global, double step, double momentum);
/* Optional: names the dataset rows [first, first+B) that the NEXT gradient computation will run on
 * (the loop NNT:82-85 knows its next batch).  On the small-net path the kernel that updates the
 * weights then also forms that batch's first-layer sums, tile by tile, from the weights it has just
 * written (one dependent launch less per step).  Good for one update; results do not depend on it. */
int gnn_mlp_hint_next_range(gnn_mlp_t *h, int64_t first, int B);
int gnn_mlp_synchronize(gnn_mlp_t *h);
/* A caller that captured steps into a hipGraph (stream capture on the stream given to
 * gnn_mlp_set_stream) replays device work the host-side `time` counter (SCE:343) does not see;
 * it reports the replayed steps here (and takes back, with a negative count, the steps that
 * were only captured, not executed). */
int gnn_mlp_advance_time(gnn_mlp_t *h, int steps);
/* Drops what the handle remembers about work done AHEAD of the next gradient computation (the next batch's first-layer
 * sums made by the previous step's tile kernel, a pending gnn_mlp_hint_next_range).  The next gradient computation then
 * starts its own chain: results are unchanged (bitwise), one extra launch.  A caller that captures
 * gnn_mlp_compute_gradient_range / gnn_mlp_apply_update into a HIP graph itself MUST call this before the capture begins
 * (so that the captured sequence does not depend on what ran before it), after it ends, and after every replay (the host
 * bookkeeping describes the captured pass, not what the device then holds).  gnn_mlp_train_range's own captured pass
 * (GNN_MLP_GRAPH=1) is a closed chain -- it starts and ends on the same batch -- and keeps that bookkeeping consistent itself. */
int gnn_mlp_forget_lookahead(gnn_mlp_t *h);
/* After a stream capture that FAILED (e.g. a collective that cannot be captured invalidated it):
 * ends a capture still open on the handle's stream, discards its graph and clears the sticky HIP
 * error of the calling thread, so that eager calls work again; if the runtime keeps the stream
 * in the invalidated state the handle returns to its own stream and the caller binds a fresh one
 * with gnn_mlp_set_stream. No reference counterpart. */
int gnn_mlp_recover_stream(gnn_mlp_t *h);

/* ---- one process per GPU, the exchange INSIDE the library's step loop -------------------------------------------------
 * The partitioning of the hooks above (rows sharded over the ranks, ONE sum of the flat gradient per step, the identical
 * update with batchSize = B_global, SCE:305-322 / SCE:333) with the sum done by the library itself: every rank process attaches
 * an RCCL communicator to its handle (ncclCommInitRank; RCCL is loaded at run time) and then runs n_steps steps in ONE call --
 * gradient kernels, ncclAllReduce on the same stream, update kernel, no host work of the caller's in between.
 *   rank 0:      gnn_mlp_rccl_unique_id(id)            -> 128 bytes, handed to every rank by the caller (a torch.distributed
 *                                                         store, MPI, a file: the library does no rendezvous of its own)
 *   every rank:  gnn_mlp_rccl_attach(h, id, n_ranks, rank)   (collective: returns when all ranks have joined)
 *                gnn_mlp_rccl_train_range(h, first, B_local, n_steps, step, momentum)   rows [first + s B_local, ..) of THIS
 *                                                         rank's resident data set per step; batchSize = B_local * n_ranks
 * Replicas stay bitwise identical (the all-reduce leaves the same bits on every rank).  UNVERIFIED beyond a world of one rank:
 * the build boxes have one GPU. */
int gnn_mlp_rccl_unique_id(void *id128);
int gnn_mlp_rccl_attach(gnn_mlp_t *h, const void *id128, int n_ranks, int rank);
int gnn_mlp_rccl_detach(gnn_mlp_t *h);
int gnn_mlp_rccl_train_range(gnn_mlp_t *h, int64_t first, int B_local, int n_steps, double step, double momentum);

/* ---- data parallel INSIDE the library: one handle, N device replicas -------------------------------
 * For a caller that is one thread in one process (the reference: NeuralNetTrainer calls gradientStep from
 * the JVM's main thread, NNT:83).  The batch's rows are dealt to the replicas in contiguous blocks, every
 * replica forms the partial gradient of its rows (the per-sample loop SCE:305-322 restricted to them), the
 * flat gradient buffers are summed across the devices, and every replica applies the identical update with
 * batchSize = B (SCE:333): replicas stay bitwise identical.  `max_batch` bounds the GLOBAL batch.
 *   GNN_REDUCE_RCCL   : ncclCommInitAll + one ncclAllReduce(SUM) per replica per step (RCCL over xGMI);
 *                       RCCL is loaded at run time (dlopen), one distinct device per replica.
 *   GNN_REDUCE_DIRECT : peer-mapped gradient buffers, stream events between the devices, ONE kernel per
 *                       replica that sums all partial gradients in rank order and updates (no collective
 *                       library; replicas may share a device, which is how it is tested on one GPU).
 * Everything that is per net (propagate, loss, argmax, get/set of weights, checkpoint) goes through a
 * replica's own handle, gnn_mlp_dp_replica(h, r, &net) -- borrowed, never destroyed by the caller. */
typedef struct gnn_mlp_dp gnn_mlp_dp_t;
typedef enum { GNN_REDUCE_RCCL = 0, GNN_REDUCE_DIRECT = 1, GNN_REDUCE_DIRECT_RS = 2 } gnn_reducer;
int gnn_mlp_dp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act, int loss,
                      int64_t seed, int dtype, const int32_t *devices, int n_dev, int max_batch, int reducer,
                      gnn_mlp_dp_t **out);
int gnn_mlp_dp_destroy(gnn_mlp_dp_t *h);
int gnn_mlp_dp_num_replicas(const gnn_mlp_dp_t *h);
int gnn_mlp_dp_replica(gnn_mlp_dp_t *h, int r, gnn_mlp_t **out);
/* NN:51 gradientStep, sharded over the replicas. */
int gnn_mlp_dp_gradient_step(gnn_mlp_dp_t *h, const double *X, const double *Y, int B, double step,
                             double momentum, int noise);
/* Every replica keeps the whole training set (NNT:28-43): any batch of it can then be sharded in place. */
int gnn_mlp_dp_upload_dataset(gnn_mlp_dp_t *h, const double *X, const double *Y, int64_t N);
int gnn_mlp_dp_gradient_step_range(gnn_mlp_dp_t *h, int64_t first, int B, double step, double momentum,
                                   int noise);
int gnn_mlp_dp_train_range(gnn_mlp_dp_t *h, int64_t first, int B, int n_steps, double step, double momentum);
int gnn_mlp_dp_set_weights(gnn_mlp_dp_t *h, const double *flat);
int gnn_mlp_dp_synchronize(gnn_mlp_dp_t *h);
/* *identical = 1 when every replica holds the same weights, momentum and step count, bit for bit. */
int gnn_mlp_dp_replicas_identical(gnn_mlp_dp_t *h, int *identical);

/* ---- shape specialisation ---------------------------------------------------------------------
 * The per-row-block kernel of the fused small-net path is a template over the net's shape; with
 * compile-time layer sizes it is ~1.5x faster than with sizes read from kernel arguments.
 * gnn_mlp_specialize instantiates it for THIS net at run time (hiprtc, ~0.4 s once per shape and
 * process); a handle does it by itself at its 16th gradient computation or when a long training
 * call (>= 64 steps) starts; env GNN_MLP_JIT=0 turns it off.  Results are bitwise identical
 * either way.
 * gnn_mlp_specialization: 0 = generic kernels (sizes from arguments, or the net does not take
 * the fused path), 1 = prebuilt instantiation (the two MNIST shapes BASELINE.json names),
 * 2 = instantiated at run time. */
int gnn_mlp_specialize(gnn_mlp_t *h);
int gnn_mlp_specialization(const gnn_mlp_t *h);
/* Kernel launches of one gradientStep inside a training loop on this net: 2 = the two-launch path
 * (row-block kernel + tile-owner kernel, csrc/tile_step_kernel.h), 3 = first layer / row-block kernel /
 * gradient+update, 0 = per-layer GEMMs (the count then depends on the layer count). */
int gnn_mlp_step_launches(const gnn_mlp_t *h);
/* The row-block kernel of the two-launch TRAINING step (csrc/rowblock_kernel.h): 0 = not taken (bf16, a plan that does not
 * fit, GNN_MLP_ROWBLOCK=0: the step then uses middle4_kernel), 1 = runtime-shape instantiation, 2 = prebuilt for the shape,
 * 3 = instantiated at run time for the shape (gnn_mlp_specialize / the 16th step). */
int gnn_mlp_rowblock_state(const gnn_mlp_t *h);
/* Why gnn_mlp_step_launches() is not 2 for this net ("" when it is): the decision gnn_mlp_create took (layer count,
 * LDS budget, slab count, a failed allocation, a GNN_MLP_* development switch).  The string lives as long as the handle. */
const char *gnn_mlp_plan_note(const gnn_mlp_t *h);

/* ---- measurement support (bench.py) -------------------------------------------------------
 * Mean duration in microseconds of the kernel class `which` over the launches since the last
 * reset.  Kernel classes are timed with the dispatch's own begin/end timestamps
 * (hipExtLaunchKernel start/stop events on the handle's stream), i.e. the quantity rocprofv3's
 * kernel trace reports; timing must be enabled first. */
typedef enum {
    GNN_K_FWD_GEMM0 = 0,  /* first forward GEMM  (B x d_0 x d_1), as its own launch (inference; start of a step chain) */
    GNN_K_GRAD_GEMM0 = 1, /* weight-gradient kernel (d_0 x d_1 x B ...): with the fused update, and on the two-launch
                             path with the next batch's first-layer product */
    GNN_K_STEP = 2,       /* one whole gradient step (events recorded around the launches) */
    GNN_K_MIDDLE = 3,     /* fused path only: the per-row-block kernel between A_1 and delta_1 */
    GNN_K_UPDATE = 4      /* data-parallel path: the momentum update after the all-reduce */
} gnn_kernel_class;
int gnn_mlp_timing_enable(gnn_mlp_t *h, int on);
int gnn_mlp_timing_read(gnn_mlp_t *h, int which, double *mean_us, int64_t *count);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GNN_MLP_H */
