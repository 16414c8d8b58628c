// sampler.hip -- the trainer's side of the boundary (NNT:143-168): the exact epoch sampler of NeuralNetTrainer.sample /
// refillSampler on java.util.Random, and gnn_mlp_train_sampled, the train loop of NNT:60-92 on a resident dataset.
#include "handle.h"
#include "java_random.h"

#include <algorithm>
#include <condition_variable>
#include <mutex>
#include <thread>

using namespace gnn;
using namespace gnn::host;

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

} // extern "C"
