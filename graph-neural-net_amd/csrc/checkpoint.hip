// checkpoint.hip -- gnn_mlp_save_checkpoint / gnn_mlp_load_checkpoint (the reference has no persistence).
#include "handle.h"

#include <cstdio>
#include <cstring>

using namespace gnn;
using namespace gnn::host;

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

extern "C" {

int gnn_mlp_save_checkpoint(gnn_mlp_t *h, const char *path) { return guarded([&]() -> int {
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
}); }

int gnn_mlp_load_checkpoint(gnn_mlp_t *h, const char *path) { return guarded([&]() -> int {
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
}); }

} // extern "C"
