// dp.hip -- gnn_mlp_dp_*: ONE handle, N device replicas (csrc/dp_handle.h; include/gnn_mlp.h "data parallel inside the
// library").
#include "handle.h"
#include "dp_handle.h"

#include <algorithm>
#include <cstring>
#include <dlfcn.h>
#include <mutex>

using namespace gnn;
using namespace gnn::host;

namespace {

// RCCL is bound at run time: libgnn_mlp_hip.so itself has no link dependency on it
struct RcclApi {
    void *lib = nullptr;
    bool ok = false; // every entry point below resolved (a library that lacks one is closed again and never used)
    struct UniqueId { char bytes[128]; }; // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128), passed by value
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*ReduceScatter)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::mutex mu;
    bool load(std::string *why) {
        std::lock_guard<std::mutex> lock(mu);
        if (ok) return true;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) {
            const char *e = dlerror();
            *why = std::string("RCCL not found: ") + (e ? e : "dlopen failed");
            return false;
        }
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        ReduceScatter = reinterpret_cast<decltype(ReduceScatter)>(dlsym(lib, "ncclReduceScatter"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!CommInitAll || !GetUniqueId || !CommInitRank || !CommDestroy || !AllReduce || !ReduceScatter || !AllGather || !GroupStart || !GroupEnd) {
            *why = "RCCL lacks the expected entry points";
            (void)dlclose(lib);
            lib = nullptr;
            CommInitAll = nullptr; GetUniqueId = nullptr; CommInitRank = nullptr; CommDestroy = nullptr; AllReduce = nullptr; ReduceScatter = nullptr; AllGather = nullptr;
            GroupStart = nullptr; GroupEnd = nullptr; GetErrorString = nullptr;
            return false;
        }
        ok = true;
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
    std::vector<float *> gbuf[2];                 // GNN_REDUCE_DIRECT / _RS: gradient buffers by step parity
    std::vector<hipEvent_t> grad_done[2], red_done[2];
    std::vector<float *> red;                     // GNN_REDUCE_DIRECT_RS: each replica's reduced slice (indexed like the gradient)
    std::vector<hipEvent_t> scat_done;
    int64_t slice = 0;                            // floats per owner, a multiple of 16
    int64_t steps = 0;
    // a step that failed AFTER some replica's work was enqueued (a refused launch, a failed event call): the replicas may no
    // longer hold the same weights / step count / buffer parity -- every later step is refused instead of training on silently
    bool broken = false;
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
    // direct: events order the devices; the sum, the update and (two-launch nets with the next batch known) the next
    // step's first layer are ONE kernel per replica, reading every partial gradient through peer pointers
    static_assert(TS_MAX_PEERS == DP_MAX_REPLICAS, "tile_step_kernel's peer table and the handle's replica limit");
    const int par = (int)(d->steps & 1);
    const float sob = (float)(step / (double)B_global), mom = (float)momentum;
    for (int r = 0; r < n; r++) {
        gnn_mlp *h = d->rep[r];
        HIP_TRY(hipSetDevice(d->dev[r]));
        HIP_TRY(hipEventRecord(d->grad_done[par][r], h->stream));
    }
    if (d->reducer == GNN_REDUCE_DIRECT_RS) { // phase 1: replica r reduces slice r
        for (int r = 0; r < n; r++) {
            gnn_mlp *h = d->rep[r];
            HIP_TRY(hipSetDevice(d->dev[r]));
            for (int j = 0; j < n; j++)
                if (j != r) HIP_TRY(hipStreamWaitEvent(h->stream, d->grad_done[par][j], 0));
            DirectScatterParams p{};
            for (int j = 0; j < n; j++) p.G[j] = reinterpret_cast<const float4 *>(d->gbuf[par][j]);
            p.n = n;
            p.red = reinterpret_cast<float4 *>(d->red[r]);
            p.lo4 = std::min<int64_t>((int64_t)r * d->slice, h->n_pad) / 4;
            p.hi4 = std::min<int64_t>((int64_t)(r + 1) * d->slice, h->n_pad) / 4;
            if (p.hi4 > p.lo4) launch_timed(h, -1, direct_reduce_scatter_kernel, dim3(grid_for(p.hi4 - p.lo4)), dim3(256), 0, p);
            HIP_TRY(hipEventRecord(d->scat_done[r], h->stream));
        }
    }
    for (int r = 0; r < n; r++) {
        gnn_mlp *h = d->rep[r];
        HIP_TRY(hipSetDevice(d->dev[r]));
        const bool rs = d->reducer == GNN_REDUCE_DIRECT_RS;
        for (int j = 0; j < n; j++)
            if (j != r) HIP_TRY(hipStreamWaitEvent(h->stream, rs ? d->scat_done[j] : d->grad_done[par][j], 0));
        NextBatch nb{};
        if (h->chain && take_next(h, &nb)) {
            // by weight tiles: each tile's sum (or gather) -> update -> the next batch's first-layer slab, one launch
            PeerGradients pg{};
            for (int j = 0; j < n; j++) pg.G[j] = rs ? d->red[j] : d->gbuf[par][j];
            pg.n = n; pg.slice = (unsigned)d->slice;
            launch_tile_step(h, rs ? 4 : 3, 2, &nb, nullptr, PAD, sob, mom, false, &pg);
            slabs_now_hold(h, nb, nb.idx != nullptr);
        } else if (rs) {
            DirectGatherParams p{};
            for (int j = 0; j < n; j++) p.red[j] = reinterpret_cast<const float4 *>(d->red[j]);
            p.n = n; p.slice4 = d->slice / 4;
            p.W = reinterpret_cast<float4 *>(h->W); p.V = reinterpret_cast<float4 *>(h->V);
            p.Wb = reinterpret_cast<sgd_bf16x4 *>(h->Wb);
            p.n4 = h->n_pad / 4;
            p.step_over_b = sob; p.momentum = mom;
            launch_timed(h, GNN_K_UPDATE, direct_gather_update_kernel, dim3(grid_for(p.n4)), dim3(256), 0, p);
            h->slab_valid = false; h->have_next = false;
        } else {
            DirectReduceParams p{};
            for (int j = 0; j < n; j++) p.G[j] = reinterpret_cast<const float4 *>(d->gbuf[par][j]);
            p.n = n;
            p.W = reinterpret_cast<float4 *>(h->W); p.V = reinterpret_cast<float4 *>(h->V);
            p.Wb = reinterpret_cast<sgd_bf16x4 *>(h->Wb);
            p.n4 = h->n_pad / 4;
            p.step_over_b = sob; p.momentum = mom;
            launch_timed(h, GNN_K_UPDATE, direct_reduce_update_kernel, dim3(grid_for(p.n4)), dim3(256), 0, p);
            h->slab_valid = false; h->have_next = false;
        }
        HIP_TRY(hipEventRecord(d->red_done[par][r], h->stream));
    }
    // every replica's update is enqueued: one check over all of them, then the step counts for all or for none
    int rc = GNN_OK;
    for (int r = 0; r < n; r++) { const int e = check_launches(d->rep[r]); if (e != GNN_OK && rc == GNN_OK) rc = e; }
    if (rc != GNN_OK) return rc;
    for (int r = 0; r < n; r++) d->rep[r]->time++;
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

} // namespace

namespace gnn {
namespace host {
void rccl_detach_handle(gnn_mlp *h) {
    if (!h->rccl_comm) return;
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    if (g_rccl.CommDestroy) (void)g_rccl.CommDestroy(h->rccl_comm);
    h->rccl_comm = nullptr; h->rccl_ranks = 0; h->rccl_rank = 0;
}
} // namespace host
} // namespace gnn

namespace {
int dp_check(const gnn_mlp_dp *d) { return d ? GNN_OK : fail(GNN_ERR_BAD_ARG, "null handle"); }
int dp_check_steppable(const gnn_mlp_dp *d) {
    TRY(dp_check(d));
    if (d->broken) return fail(GNN_ERR_STATE, "an earlier data-parallel step failed part-way: the replicas may differ (gnn_mlp_dp_replicas_identical); "
                                              "destroy the handle, or restore every replica from a checkpoint and call gnn_mlp_dp_set_weights");
    return GNN_OK;
}
// runs a step's body; a failure once work may have been enqueued on some replica leaves the handle marked
template <class F> int dp_guard_step(gnn_mlp_dp *d, F &&body) {
    const int rc = body();
    if (rc != GNN_OK) d->broken = true;
    return rc;
}

} // namespace

extern "C" {

// ---- one process per GPU, the exchange inside the library's step loop ------------------------------------------------------
int gnn_mlp_rccl_unique_id(void *id128) { return guarded([&]() -> int {
    if (!id128) return fail(GNN_ERR_BAD_ARG, "null output");
    std::string why;
    if (!g_rccl.load(&why)) return fail(GNN_ERR_UNSUPPORTED, why);
    RcclApi::UniqueId id{};
    const int rc = g_rccl.GetUniqueId(&id);
    if (rc) return dp_rccl_fail(rc, "ncclGetUniqueId");
    std::memcpy(id128, id.bytes, sizeof(id.bytes));
    return GNN_OK;
}); }

int gnn_mlp_rccl_attach(gnn_mlp_t *h, const void *id128, int n_ranks, int rank) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!id128 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(GNN_ERR_BAD_ARG, "bad communicator arguments");
    if (h->rccl_comm) return fail(GNN_ERR_STATE, "the handle already has a communicator (gnn_mlp_rccl_detach first)");
    std::string why;
    if (!g_rccl.load(&why)) return fail(GNN_ERR_UNSUPPORTED, why);
    RcclApi::UniqueId id{};
    std::memcpy(id.bytes, id128, sizeof(id.bytes));
    void *comm = nullptr;
    const int rc = g_rccl.CommInitRank(&comm, n_ranks, id, rank); // (collective: returns when every rank has joined)
    if (rc) return dp_rccl_fail(rc, "ncclCommInitRank");
    h->rccl_comm = comm; h->rccl_ranks = n_ranks; h->rccl_rank = rank;
    return GNN_OK;
}); }

int gnn_mlp_rccl_detach(gnn_mlp_t *h) { return guarded([&]() -> int {
    TRY(check_handle(h));
    rccl_detach_handle(h);
    return GNN_OK;
}); }

int gnn_mlp_rccl_train_range(gnn_mlp_t *h, int64_t first, int B_local, int n_steps, double step, double momentum) { return guarded([&]() -> int {
    TRY(check_handle(h));
    if (!h->rccl_comm) return fail(GNN_ERR_STATE, "no communicator attached (gnn_mlp_rccl_attach)");
    TRY(check_step_args(h, B_local, step, 0));
    if (!h->DX) return fail(GNN_ERR_STATE, "no dataset uploaded");
    if (n_steps <= 0) return fail(GNN_ERR_BAD_ARG, "n_steps must be positive (NNT:62)");
    const int64_t nb = h->dataset_n / B_local;
    if (nb <= 0 || first < 0 || first % B_local != 0) return fail(GNN_ERR_BAD_ARG, "first must be a multiple of B_local inside the dataset");
    if (n_steps >= 64) try_specialize(h);
    const int Lm = h->L - 1;
    const int B_global = B_local * h->rccl_ranks;                 // batchSize of SCE:333: every rank holds B_local rows of the batch
    const float sob = (float)(step / (double)B_global), mom = (float)momentum;
    for (int s = 0; s < n_steps; s++) {
        const int64_t row0 = ((first / B_local + s) % nb) * B_local;
        // (1) this rank's partial gradient (the per-sample loop SCE:305-322 restricted to its rows) into the flat buffer
        hint_range(h, ((first / B_local + s + 1) % nb) * B_local, B_local);
        maybe_specialize(h);
        do_gradient(h, h->DX + (size_t)row0 * h->ld[0], h->DY + (size_t)row0 * h->ld[Lm], B_local, false, 0.f, 0.f, true);
        // (2) ONE sum of the flat buffer over the ranks, on the stream the kernels run on
        const int rc = g_rccl.AllReduce(h->G, h->G, (size_t)h->n_pad, kNcclFloat, kNcclSum, h->rccl_comm, h->stream);
        if (rc) return dp_rccl_fail(rc, "ncclAllReduce");
        // (3) the identical update on every rank (SCE:327-342 with batchSize = B_global), by weight tiles, each tile going on to
        //     the next batch's first-layer slab where the net takes the two-launch path
        NextBatch nb_next{};
        if (h->chain && take_next(h, &nb_next)) {
            launch_tile_step(h, 2, 2, &nb_next, nullptr, PAD, sob, mom);
            slabs_now_hold(h, nb_next, nb_next.idx != nullptr);
        } else {
            launch_flat_update(h, B_global, step, momentum);
            h->slab_valid = false; h->have_next = false;
        }
        h->time++;
        TRY_LAUNCHES(h);
    }
    return GNN_OK;
}); }

int gnn_mlp_dp_create(const int32_t *dims, int n_dims, int out_kind, int inner_act, int last_act, int loss, int64_t seed,
                      int dtype, const int32_t *devices, int n_dev, int max_batch, int reducer, gnn_mlp_dp_t **out) { return guarded([&]() -> int {
    if (!out) return fail(GNN_ERR_BAD_ARG, "out is null");
    *out = nullptr;
    if (!devices || n_dev < 1 || n_dev > DP_MAX_REPLICAS) return fail(GNN_ERR_BAD_ARG, "n_dev must be 1..16");
    if (reducer != GNN_REDUCE_RCCL && reducer != GNN_REDUCE_DIRECT && reducer != GNN_REDUCE_DIRECT_RS) return fail(GNN_ERR_BAD_ARG, "bad reducer");
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
            if (reducer == GNN_REDUCE_DIRECT_RS) {
                float *g = nullptr;
                const int rc = dev_alloc(&g, (size_t)d->rep[r]->n_pad, d->rep[r]->stream);
                if (rc != GNN_OK) return cleanup(rc);
                d->red.push_back(g);
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return cleanup(fail(GNN_ERR_HIP, "hipEventCreate"));
                d->scat_done.push_back(e);
            }
            if (hipStreamSynchronize(d->rep[r]->stream) != hipSuccess) return cleanup(fail(GNN_ERR_HIP, "hipStreamSynchronize"));
        }
    }
    {   // floats per owner: the flat (padded) buffer cut into n_dev pieces on 16-float boundaries
        const int64_t np = d->rep[0]->n_pad;
        d->slice = ((np + n_dev - 1) / n_dev + 15) / 16 * 16;
    }
    *out = d;
    return GNN_OK;
}); }

int gnn_mlp_dp_destroy(gnn_mlp_dp_t *d) { return guarded([&]() -> int {
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
    for (size_t r = 0; r < d->red.size(); r++) { (void)hipSetDevice(d->dev[r]); (void)hipFree(d->red[r]); }
    for (hipEvent_t e : d->scat_done) (void)hipEventDestroy(e);
    for (gnn_mlp *h : d->rep) (void)gnn_mlp_destroy(h);
    delete d;
    return GNN_OK;
}); }

int gnn_mlp_dp_num_replicas(const gnn_mlp_dp_t *d) { return d ? d->n : -1; }

int gnn_mlp_dp_replica(gnn_mlp_dp_t *d, int r, gnn_mlp_t **out) { return guarded([&]() -> int {
    TRY(dp_check(d));
    if (!out || r < 0 || r >= d->n) return fail(GNN_ERR_BAD_ARG, "replica index out of range");
    *out = d->rep[r];
    return GNN_OK;
}); }

int gnn_mlp_dp_gradient_step(gnn_mlp_dp_t *d, const double *X, const double *Y, int B, double step, double momentum, int noise) { return guarded([&]() -> int {
    TRY(dp_check_steppable(d));
    if (!X || !Y) return fail(GNN_ERR_BAD_ARG, "null argument (reference: assert batch != null, SCE:299)");
    if (B <= 0) return fail(GNN_ERR_BAD_ARG, "batch must be non-empty (SCE:300)");
    if (B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B exceeds max_batch given to gnn_mlp_dp_create");
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    const int d0 = d->rep[0]->dims[0], dl = d->rep[0]->dims[d->rep[0]->L - 1];
    return dp_guard_step(d, [&]() -> int {
        for (int r = 0; r < d->n; r++) {
            int lo, hi;
            dp_shard(B, r, d->n, &lo, &hi);
            gnn_mlp *h = d->rep[r];
            TRY(check_handle(h));
            if (d->reducer != GNN_REDUCE_RCCL) TRY(dp_direct_begin(d, r));
            if (hi > lo) TRY(gnn_mlp_compute_gradient(h, X + (size_t)lo * d0, Y + (size_t)lo * dl, hi - lo));
            else HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * (size_t)h->n_pad, h->stream)); // no rows: a zero partial gradient
        }
        TRY(dp_reduce_and_update(d, B, step, momentum));
        d->steps++;
        return GNN_OK;
    });
}); }

int gnn_mlp_dp_upload_dataset(gnn_mlp_dp_t *d, const double *X, const double *Y, int64_t N) { return guarded([&]() -> int {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_upload_dataset(d->rep[r], X, Y, N)); // every replica holds every row: any batch can be sharded
    return GNN_OK;
}); }

static int dp_step_range_body(gnn_mlp_dp *d, int64_t first, int B, double step, double momentum, int64_t next_first);
static int dp_step_range(gnn_mlp_dp *d, int64_t first, int B, double step, double momentum, int64_t next_first) {
    return dp_guard_step(d, [&]() -> int { return dp_step_range_body(d, first, B, step, momentum, next_first); });
}
static int dp_step_range_body(gnn_mlp_dp *d, int64_t first, int B, double step, double momentum, int64_t next_first) {
    for (int r = 0; r < d->n; r++) {
        int lo, hi;
        dp_shard(B, r, d->n, &lo, &hi);
        gnn_mlp *h = d->rep[r];
        TRY(check_handle(h));
        if (d->reducer != GNN_REDUCE_RCCL) TRY(dp_direct_begin(d, r));
        if (next_first >= 0 && hi > lo) TRY(gnn_mlp_hint_next_range(h, next_first + lo, hi - lo)); // the update kernel then starts the next step
        if (hi > lo) TRY(gnn_mlp_compute_gradient_range(h, first + lo, hi - lo));
        else HIP_TRY(hipMemsetAsync(h->G, 0, sizeof(float) * (size_t)h->n_pad, h->stream));
    }
    TRY(dp_reduce_and_update(d, B, step, momentum));
    d->steps++;
    return GNN_OK;
}

int gnn_mlp_dp_gradient_step_range(gnn_mlp_dp_t *d, int64_t first, int B, double step, double momentum, int noise) { return guarded([&]() -> int {
    TRY(dp_check_steppable(d));
    if (B <= 0 || B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B out of range");
    if (noise) return fail(GNN_ERR_UNSUPPORTED, "noise=true is not built on the GPU (SCE:335)");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    if (first < 0 || first + B > d->rep[0]->dataset_n) return fail(GNN_ERR_BAD_ARG, "dataset rows out of range");
    return dp_step_range(d, first, B, step, momentum, -1);
}); }

int gnn_mlp_dp_train_range(gnn_mlp_dp_t *d, int64_t first, int B, int n_steps, double step, double momentum) { return guarded([&]() -> int {
    TRY(dp_check_steppable(d));
    if (B <= 0 || B > d->max_batch) return fail(GNN_ERR_BAD_ARG, "B out of range");
    if (!(step > 0)) return fail(GNN_ERR_BAD_ARG, "step must be positive (SCE:301)");
    if (n_steps <= 0) return fail(GNN_ERR_BAD_ARG, "n_steps must be positive (NNT:62)");
    const int64_t nb = d->rep[0]->dataset_n / B;
    if (nb <= 0 || first < 0 || first % B != 0) return fail(GNN_ERR_BAD_ARG, "first must be a multiple of B inside the dataset");
    for (int s = 0; s < n_steps; s++) {
        const int64_t row0 = ((first / B + s) % nb) * B;
        // (the LAST step names the batch that follows the range in the data set, as gnn_mlp_train_range does: calls that walk the
        //  data set one after the other then continue one chain of two-launch steps)
        const int64_t nxt = ((first / B + s + 1) % nb) * B;
        TRY(dp_step_range(d, row0, B, step, momentum, nxt));
    }
    return GNN_OK;
}); }

int gnn_mlp_dp_set_weights(gnn_mlp_dp_t *d, const double *flat) { return guarded([&]() -> int {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_set_weights(d->rep[r], flat));
    return GNN_OK;
}); }

int gnn_mlp_dp_synchronize(gnn_mlp_dp_t *d) { return guarded([&]() -> int {
    TRY(dp_check(d));
    for (int r = 0; r < d->n; r++) TRY(gnn_mlp_synchronize(d->rep[r]));
    return GNN_OK;
}); }

int gnn_mlp_dp_replicas_identical(gnn_mlp_dp_t *d, int *identical) { return guarded([&]() -> int {
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
}); }

} // extern "C"
