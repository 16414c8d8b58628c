// sampler.hip -- the trainer's side of the boundary (NNT:143-168): the exact epoch sampler of NeuralNetTrainer.sample /
// refillSampler on java.util.Random, and gnn_mlp_train_sampled, the train loop of NNT:60-92 on a resident dataset.
#include "handle.h"
#include "java_random.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace gnn;
using namespace gnn::host;

// The k-th set bit (k 0-based, k < popcount) of a 64-bit word: deposit one bit at the k-th set position, count the zeros below it.
#if defined(__x86_64__)
__attribute__((target("bmi2"))) static inline int select_bit_bmi2(uint64_t w, int k) { return (int)__builtin_ctzll(__builtin_ia32_pdep_di(1ull << k, w)); }
static inline bool have_bmi2() { return __builtin_cpu_supports("bmi2"); }
#else
static inline int select_bit_plain(uint64_t w, int k);
static inline int select_bit_bmi2(uint64_t w, int k) { return select_bit_plain(w, k); }
static inline bool have_bmi2() { return false; }
#endif
static inline int select_bit_plain(uint64_t w, int k) {
    for (; k > 0; k--) w &= w - 1;
    return (int)__builtin_ctzll(w);
}

struct gnn_sampler {
    int32_t master = 0, remaining = 0;
    // "row still in dataSampler" as one bit per row, 64 rows per word, and a Fenwick tree over the words' counts: the r-th
    // remaining row in master order is what ArrayList.get(r) returns after removals (NNT:152-154).  A tree over words, not
    // rows: 938 counters for MNIST's 60 000 rows stay in L1 and the walk is 10 levels instead of 16 through L2 -- the sampler
    // feeds a training loop whose step is 13 us, and at two row-level walks per draw it was the slower of the two
    // (14.3 us per batch of 128 against 13.2; tools/bench_trainer.py).
    std::vector<uint64_t> bits;
    std::vector<int32_t> fen; // 1-based over the words
    int32_t words = 0;
    JavaRandom rnd{1};
    int log2w = 0;
    bool bmi2 = false;
    void refill() { // refillSampler NNT:164-168
        bits.assign((size_t)words, ~0ull);
        if (master % 64) bits[(size_t)words - 1] = (1ull << (master % 64)) - 1;
        // (entries behind the last word, up to the next power of two, read as "more than any k": the walk below needs no bounds test)
        fen.assign(((size_t)2 << log2w) + 1, INT32_MAX);
        std::fill(fen.begin(), fen.begin() + words + 1, 0);
        for (int32_t i = 1; i <= words; i++) {
            fen[i] += (int32_t)__builtin_popcountll(bits[(size_t)i - 1]);
            const int32_t j = i + (i & -i);
            if (j <= words) fen[j] += fen[i];
        }
        remaining = master;
    }
    int32_t take(int32_t r) { // remove and return the r-th (0-based) remaining row
        int32_t pos = 0, k = r + 1;
        for (int32_t pw = 1 << log2w; pw > 0; pw >>= 1) { // (selects, not branches: the comparison is a coin toss per level)
            const int32_t f = fen[pos + pw];
            const bool go = f < k;
            pos += go ? pw : 0;
            k -= go ? f : 0;
        }
        // word `pos` (0-based) holds the row: its (k-1)-th set bit
        const uint64_t w = bits[(size_t)pos];
        const int bit = bmi2 ? select_bit_bmi2(w, k - 1) : select_bit_plain(w, k - 1);
        bits[(size_t)pos] = w & ~(1ull << bit);
        for (int32_t i = pos + 1; i <= words; i += i & -i) fen[i] -= 1;
        remaining--;
        return pos * 64 + bit; // 0-based row
    }
};

extern "C" {

int gnn_sampler_create(int32_t master_size, int64_t seed, gnn_sampler_t **out) { return guarded([&]() -> int {
    if (!out || master_size <= 0) return fail(GNN_ERR_BAD_ARG, "bad sampler arguments");
    gnn_sampler *s = new gnn_sampler();
    s->master = master_size;
    s->words = (master_size + 63) / 64;
    s->rnd.set_seed(seed);
    while ((1 << (s->log2w + 1)) <= s->words) s->log2w++;
    s->bmi2 = have_bmi2();
    s->refill();
    *out = s;
    return GNN_OK;
}); }

int gnn_sampler_destroy(gnn_sampler_t *s) { delete s; return GNN_OK; }

int gnn_sampler_sample(gnn_sampler_t *s, int batch, int32_t *out_idx, int *n_out) { return guarded([&]() -> int {
    if (!s || !out_idx || !n_out || batch <= 0) return fail(GNN_ERR_BAD_ARG, "bad sampler arguments");
    int n = 0, before_refill = 0; // rows drawn before a refill inside this batch: only THEY can come again (within an epoch rows are distinct)
    for (int i = 0; i < batch; i++) {
        if (s->remaining == 0) { s->refill(); before_refill = n; } // NNT:149-151
        const int32_t r = s->rnd.next_int(s->remaining);           // NNT:152
        const int32_t row = s->take(r);                            // NNT:153-154
        bool dup = false;                                          // HashMap.put, NNT:155
        for (int k = 0; k < before_refill; k++) if (out_idx[k] == row) { dup = true; break; }
        if (!dup) out_idx[n++] = row;
    }
    *n_out = n;
    return GNN_OK;
}); }

} // extern "C"

namespace gnn {
namespace host {
// validate(validation_size) of NNT:102-113 without its division: calculateLoss over the first n samples in master order
// (dataset rows [0, n)), block by block on the handle's stream, summed into ONE fp64 slot on the device.
int validation_loss_sum(gnn_mlp *h, int n, double *d_out) {
    const int Lm = h->L - 1;
    const int block = eval_block_rows(h, n); // (601 rows at MNIST's size: ONE block through the evaluation workspace instead of five of max_batch)
    int rc_ws = GNN_OK;
    EvalScope scope(h, block, &rc_ws);
    if (rc_ws != GNN_OK) return rc_ws;
    for (int off = 0; off < n; off += block) {
        const int B = std::min(block, n - off);
        do_forward(h, h->DX + (size_t)off * h->ld[0], h->DY + (size_t)off * h->ld[Lm], B, false, true, false);
        hipLaunchKernelGGL(sum_loss_kernel, dim3(1), dim3(256), 0, h->stream, LossSumParams{h->lossv, B, d_out, off > 0 ? 1 : 0});
    }
    return GNN_OK;
}
bool validation_losses_to_row(gnn_mlp *h, int n, float *loss_row, int *rc) {
    *rc = GNN_OK;
    if (eval_block_rows(h, n) < n) return false;
    const int Lm = h->L - 1;
    EvalScope scope(h, n, rc);
    if (*rc != GNN_OK) return true;
    float *keep = h->lossv;
    h->lossv = loss_row; // (the forward kernels write loss[row]: straight into the matrix's row)
    do_forward(h, h->DX, h->DY, n, false, true, false);
    h->lossv = keep;
    return true;
}
} // namespace host
} // namespace gnn

// The loop NNT:60-92 on a resident dataset; d_val != null: the observed variants (NNT:68-72, 75-79) -- after iteration i the
// summed validation loss of rows [0, validation_size) goes to d_val[i] (device).
static int train_sampled_impl(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step, double momentum,
                              int noise, int validation_size, double *d_val, float *d_rows = nullptr, int64_t row_stride = 0) {
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
    // Index storage is a RING of kRing chunk slots on the host and on the device, whatever the run length (NNT:62 runs
    // 500 000 iterations): the sampler waits for a free host slot; a device slot is rewritten by a copy that is enqueued on
    // the handle's stream behind the steps that read it.
    static constexpr int kRing = 4, kChunk = 256;
    auto chunk_begin = [](int c) { // 0, 16, 48, 112, 240, 496, 752, ..
        int b = 0, sz = 16;
        for (int i = 0; i < c; i++) { b += sz; sz = std::min(kChunk, sz * 2); if (sz == kChunk && i + 1 < c) { b += (c - i - 1) * kChunk; break; } }
        return b;
    };
    int n_chunks = 0;
    while (chunk_begin(n_chunks) < iterations) n_chunks++;
    auto chunk_end = [&](int c) { return std::min(iterations, chunk_begin(c + 1)); };
    const size_t slot_elems = (size_t)kChunk * batch;
    // The host ring is PINNED memory and every upload is followed by an event: the copy is then truly asynchronous (from pageable
    // memory hipMemcpyAsync stages the data before it returns -- in this runtime by waiting for the stream, i.e. for every step
    // queued so far: the GPU idled at each chunk boundary, and the slot's safety rested on that behaviour), and a host slot goes
    // back to the sampler only when its event has completed.
    struct PinnedRing {
        int32_t *p = nullptr; hipEvent_t ev[kRing] = {};
        ~PinnedRing() { if (p) (void)hipHostFree(p); for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } ring;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(&ring.p), slot_elems * kRing * sizeof(int32_t), hipHostMallocDefault));
    for (int i = 0; i < kRing; i++) HIP_TRY(hipEventCreateWithFlags(&ring.ev[i], hipEventDisableTiming));
    int32_t *const idx = ring.p;
    std::vector<int> cnt((size_t)kChunk * kRing);
    struct Shared {
        std::mutex mu;
        std::condition_variable cv;
        int ready = 0, consumed = 0, sampler_rc = GNN_OK; // chunks sampled / chunks whose host slot is free again
        bool stop = false;
        std::string sampler_msg;
    } sh;
    std::thread producer([&]() {
        for (int c = 0; c < n_chunks; c++) {
            {
                std::unique_lock<std::mutex> lk(sh.mu);
                sh.cv.wait(lk, [&] { return sh.stop || c < sh.consumed + kRing; });
                if (sh.stop) return;
            }
            int rc = GNN_OK;
            const int i0 = chunk_begin(c), i1 = chunk_end(c);
            int32_t *slot = idx + (size_t)(c % kRing) * slot_elems;
            int *scnt = cnt.data() + (size_t)(c % kRing) * kChunk;
            for (int i = i0; i < i1 && rc == GNN_OK; i++) rc = gnn_sampler_sample(s, batch, slot + (size_t)(i - i0) * batch, &scnt[i - i0]);
            std::lock_guard<std::mutex> lk(sh.mu);
            if (rc != GNN_OK) { sh.sampler_rc = rc; sh.sampler_msg = gnn_mlp_last_error(); sh.ready = n_chunks; sh.cv.notify_all(); return; }
            sh.ready = c + 1;
            sh.cv.notify_all();
        }
    });
    struct Joiner { // the worker is joined on EVERY exit path, an exception included (a joinable std::thread's destructor terminates)
        std::thread &t; Shared &sh;
        ~Joiner() {
            { std::lock_guard<std::mutex> lk(sh.mu); sh.stop = true; }
            sh.cv.notify_all();
            if (t.joinable()) t.join();
        }
    } joiner{producer, sh};
    std::vector<int> dcnt((size_t)kRing * kChunk, 0); // the batch sizes of the chunks on the device (the host ring is the sampler's again by then)
    DevScratch dbuf;
    int rc = dbuf.alloc(slot_elems * kRing * sizeof(int32_t));
    int32_t *d_idx = dbuf.as<int32_t>();
    // Chunk c + 1's draws go to the device ring BEFORE chunk c's steps are enqueued (when the sampler has them, which it has:
    // it runs kRing chunks ahead): the last step of a chunk then knows its successor like every other step, and the chain of
    // two-launch steps runs through the chunk boundary -- an upload in stream order at the boundary left the GPU idle for
    // the copy and restarted the chain with a forward-only launch, ~1 us per step over a run.
    int uploaded = 0; // chunks whose draws are on the device (or in flight in front of every launch that reads them)
    int released = 0; // chunks whose HOST slot is the sampler's again (their upload's event has completed)
    auto upload = [&](int c) -> int {
        const int j0 = chunk_begin(c), j1 = chunk_end(c);
        const size_t so = (size_t)(c % kRing) * slot_elems;
        // (the counts are copied out of the ring at once; the indices stay in the pinned slot until the copy has run)
        std::copy(cnt.begin() + (size_t)(c % kRing) * kChunk, cnt.begin() + (size_t)(c % kRing) * kChunk + (j1 - j0), dcnt.begin() + (size_t)(c % kRing) * kChunk);
        hipError_t e = hipMemcpyAsync(d_idx + so, idx + so, (size_t)(j1 - j0) * batch * sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
        if (e == hipSuccess) e = hipEventRecord(ring.ev[c % kRing], h->stream);
        if (e != hipSuccess) return fail(GNN_ERR_HIP, std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
        uploaded = c + 1;
        return GNN_OK;
    };
    // hands finished uploads' host slots back to the sampler: those whose event has completed -- all of them when `wait`
    // (an upload sits in front of the steps of the chunk BEFORE its own, so by the time the sampler needs the slot,
    //  kRing chunks later, the copy ran long ago: the wait does not stall)
    auto release = [&](bool wait) -> int {
        int r = released;
        while (r < uploaded) {
            const hipError_t q = (wait && r == released) ? hipEventSynchronize(ring.ev[r % kRing]) : hipEventQuery(ring.ev[r % kRing]); // (wait: for the OLDEST one)
            if (q == hipErrorNotReady) break;
            if (q != hipSuccess) return fail(GNN_ERR_HIP, std::string("upload event: ") + hipGetErrorString(q));
            r++;
        }
        if (r != released) {
            released = r;
            { std::lock_guard<std::mutex> lk(sh.mu); sh.consumed = r; }
            sh.cv.notify_all();
        }
        return GNN_OK;
    };
    for (int c = 0; c < n_chunks && rc == GNN_OK; c++) {
        rc = release(false);
        if (rc != GNN_OK) break;
        if (uploaded <= c) {
            std::unique_lock<std::mutex> lk(sh.mu);
            // (the sampler may be waiting for a host slot whose upload is still in flight: hand slots back while waiting for it)
            while (!(sh.ready > c)) {
                lk.unlock();
                rc = release(uploaded - released >= kRing); // every slot taken and none released: wait for the oldest upload
                lk.lock();
                if (rc != GNN_OK || sh.ready > c) break;
                sh.cv.wait_for(lk, std::chrono::microseconds(200));
            }
            if (rc != GNN_OK) break;
            if (sh.sampler_rc != GNN_OK) { rc = fail(sh.sampler_rc, sh.sampler_msg); break; }
            lk.unlock();
            rc = upload(c);
            if (rc != GNN_OK) break;
        }
        if (c + 1 < n_chunks) { // the successor too, if it is drawn already (no waiting for it)
            bool have = false;
            { std::lock_guard<std::mutex> lk(sh.mu); have = sh.ready > c + 1 && sh.sampler_rc == GNN_OK; }
            if (have) { rc = upload(c + 1); if (rc != GNN_OK) break; }
        }
        const int i0 = chunk_begin(c), i1 = chunk_end(c);
        const size_t so = (size_t)(c % kRing) * slot_elems;
        const int *ccnt = dcnt.data() + (size_t)(c % kRing) * kChunk;
        for (int i = i0; i < i1 && rc == GNN_OK; i++) {
            if (h->chain && i + 1 < i1) { // the next draw of this chunk is already on the device
                h->have_next = true; h->next_a0 = h->DX; h->next_idx = d_idx + so + (size_t)(i + 1 - i0) * batch; h->next_B = ccnt[i + 1 - i0];
            } else if (h->chain && uploaded > c + 1) { // ... and so is the first draw of the next chunk
                const size_t sn = (size_t)((c + 1) % kRing) * slot_elems;
                h->have_next = true; h->next_a0 = h->DX; h->next_idx = d_idx + sn; h->next_B = dcnt[(size_t)((c + 1) % kRing) * kChunk];
            }
            rc = step_on_device_indices(h, d_idx + so + (size_t)(i - i0) * batch, ccnt[i - i0], step, momentum);
            // (the validation pass reads the weights the step has just written; it touches neither the slabs the step's tile
            //  kernel made for the next batch nor the staged rows, so the chain of two-launch steps runs on behind it)
            if (rc == GNN_OK && d_val) {
                // the per-sample losses of iteration i stay in row i of d_rows (summed once, behind the loop); a validation set of
                // more than one block is summed block by block as before
                int vrc = GNN_OK;
                if (d_rows && validation_losses_to_row(h, validation_size, d_rows + (size_t)i * row_stride, &vrc)) rc = vrc;
                else rc = validation_loss_sum(h, validation_size, d_val + i);
            }
        }
    }
    // (on an early exit the sampler stops after the chunk it is drawing: its state stays well defined)
    (void)hipStreamSynchronize(h->stream); // the device ring is released below
    h->slab_valid = false; h->have_next = false; // (they may name rows through d_idx)
    return rc;
}

extern "C" {

int gnn_mlp_train_sampled(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step, double momentum,
                          int noise) { return guarded([&]() -> int {
    return train_sampled_impl(h, s, iterations, batch, step, momentum, noise, 0, nullptr);
}); }

int gnn_mlp_train_sampled_observed(gnn_mlp_t *h, gnn_sampler_t *s, int iterations, int batch, double step, double momentum,
                                   int noise, int validation_size, double *val_loss) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!val_loss) return fail(GNN_ERR_BAD_ARG, "null output");
    if (iterations <= 0) return fail(GNN_ERR_BAD_ARG, "iterations must be positive (NNT:62)");
    if (validation_size <= 0 || validation_size > h->dataset_n) return fail(GNN_ERR_BAD_ARG, "validation size outside the dataset (NNT:104)");
    DevScratch dv, drows;
    TRY(dv.alloc(sizeof(double) * (size_t)iterations));
    // one row of per-sample losses per iteration when the validation set is ONE forward block (601 rows at MNIST's size): the rows
    // are summed by one launch behind the loop instead of one per iteration.  (Capped at 1 GiB: longer calls sum per iteration.)
    const int64_t stride = pad_up(validation_size);
    const bool rows_form = eval_block_rows(h, validation_size) >= validation_size && (int64_t)iterations * stride * 4 <= (1ll << 30);
    if (rows_form) TRY(drows.alloc(sizeof(float) * (size_t)iterations * (size_t)stride));
    TRY(train_sampled_impl(h, s, iterations, batch, step, momentum, noise, validation_size, dv.as<double>(), rows_form ? drows.as<float>() : nullptr, stride));
    if (rows_form) {
        hipLaunchKernelGGL(sum_rows_kernel, dim3(iterations), dim3(256), 0, h->stream, RowSumParams{drows.as<float>(), stride, validation_size, dv.as<double>()});
        TRY_LAUNCHES(h);
        HIP_TRY(hipStreamSynchronize(h->stream));
    }
    // (train_sampled_impl has waited for the stream: every sum is in place)
    HIP_TRY(hipMemcpy(val_loss, dv.p, sizeof(double) * (size_t)iterations, hipMemcpyDeviceToHost));
    for (int i = 0; i < iterations; i++) val_loss[i] /= (double)validation_size; // NNT:112
    return GNN_OK;
}); }

} // extern "C"
