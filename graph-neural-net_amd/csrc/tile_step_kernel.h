// tile_step_kernel.h -- the "tile owner" kernel of the small-net path: TWO launches per gradientStep.
//
// A gradientStep (SCE:297-346) has two all-to-all exchanges: the first layer is tiled over the
// columns of W_0 while the chain A_1 -> delta_1 is per batch row, and the weight gradient
// G_l = A_l^T . delta_{l+1} (SCE:253-258, 279-283 summed over the batch by SCE:305-322) is tiled over
// the weight matrix again.  The hop from one step's update to the next step's first layer is NOT
// an exchange: the workgroup that has just updated a tile of W_0 (SCE:333-339) holds exactly the
// weights the next batch's product A_0 . W_0 (SCE:187-192) needs from that tile.  So this kernel,
// for one 64 x 16 tile of one layer's weight matrix:
//
//   GSRC = 1  G = A_l^T . delta_{l+1} over the batch rows            (MFMA, K = batch)
//   GSRC = 2  G = the tile of the all-reduced gradient buffer        (data parallel)
//   GSRC = 3  G = the rank-ordered SUM of the tile in every replica's gradient buffer (data parallel inside the library,
//             GNN_REDUCE_DIRECT: peer pointers -- the reduction, the update and the next step's first layer in ONE launch)
//   GSRC = 4  G = the tile of the REDUCED gradient, each 16-float piece read from the replica that owns its slice
//             (GNN_REDUCE_DIRECT_RS: a reduce-scatter kernel ran before; this launch is the all-gather + update + first layer)
//   GDST = 1  stores G                                               (data parallel: the all-reduce follows)
//   GDST = 2  adj = (step*G)/B + momentum*prev ; W -= adj ; prev = adj
//   FWD       layer 0 only: Zp[b][n] = sum over the tile's 64 input neurons of A_0'[b][m] . W_0[m][n]
//             for the NEXT batch A_0' with the tile's NEW weights -- one K slab of the next step's
//             first-layer sums.  middle4_kernel adds the ceil(d_0/64) slabs in slab order (fixed:
//             results do not depend on dispatch order) and applies f.
//   GSRC = 0, GDST = 0, FWD: the slabs of a batch from the weights as they are (start of a chain).
//
// One step is then { middle4_kernel ; tile_step_kernel } instead of { fwd_first ; middle4 ;
// grad_update }: one dependent launch (~2.5 us fixed on this chip) and the cold re-read of W_0 less.
// The slab arithmetic is the same whether a slab was made by the previous step's tile kernel or by a
// forward-only launch, so a sequence of steps gives bitwise the same weights however it is cut into
// calls.
//
// All contractions on v_mfma_f32_16x16x4_f32 (exact f32).  8 waves:
//   gradient: waves 0..3 -> one m tile of 16 each, product taken transposed (G^T = delta^T . A) so that a lane's
//             accumulator is four consecutive n of one row of the tile: the 16 B of W / V it updates
//   forward : wave -> 16 batch rows of every 128-row chunk; K = the tile's 64 input neurons; A_0' rows go
//             straight to registers in fragment form and the product is taken transposed, so that the
//             slab is stored 16 B per lane from the accumulators (no LDS image, one barrier in all)
#pragma once
#include "fused_kernels.h"
#include "gemm_bf16.h"
#ifndef __HIPCC_RTC__
#include <algorithm>
#include <vector>
#endif

namespace gnn {

constexpr int TS_TM = 64, TS_TN = 16, TS_THREADS = 512, TS_KC = 128;
constexpr int TS_MAX_SLABS = 16; // middle4_kernel keeps one float4 per slab in registers
constexpr int TS_MAX_PEERS = 16; // replicas of one data-parallel handle (dp_handle.h: DP_MAX_REPLICAS)
constexpr int TS_MAP_ARGS = 640; // workgroup -> tile entries that travel in the kernel arguments

struct TileStepParams {
    GradLayer layer[MAX_LAYERS]; // tiling in 64 x 16 tiles; block_begin per layer
    int n_layers;
    int K, k_true;               // padded / live batch rows of the CURRENT batch (gradient)
    float step_over_b, momentum;
    const int32_t *row_idx;      // optional, layer 0: batch row k of A_0 is dataset row row_idx[k]
    // next batch (FWD), layer 0 only
    const float *An; int ldan;   // A_0' rows
    const int32_t *next_idx;     // optional row indices of the next batch
    int next_rows, next_K;       // live / padded rows of the next batch
    float *slabs;                // [n slabs][slab_rows][ldz]
    int slab_rows, ldz;
    unsigned long long *stamps;  // STAMP builds only (tools/tile_probe.hip): 16 slots per block
    // GNN_DTYPE_BF16 (tile_step_bf16_kernel): the bf16 roundings of the operands, same shapes and leading dimensions
    const __bf16 *Ab[MAX_LAYERS]; const __bf16 *Db[MAX_LAYERS]; __bf16 *Wb[MAX_LAYERS];
    const __bf16 *Anb;
    // A sampled next batch (next_idx != null): the workgroups of tile column 0 also write the rows they fetched to a
    // contiguous copy [next_K][ldan] (f32 or bf16 by kernel), which the NEXT step's gradient product then reads in place of
    // the index-gathered rows -- no dependent index load at the start of that kernel.  Null = no copy.
    float *stage_out; __bf16 *stage_out_b;
    // GSRC = 3 / 4: every replica's gradient buffer (3: partial gradients, 4: reduced slices), rank order; Gself = the base
    // of THIS replica's bound gradient buffer (layer[l].G - Gself is the layer's offset in every peer's buffer); slice =
    // floats per owner (a multiple of 16), GSRC = 4 only
    const float *Gpeer[TS_MAX_PEERS]; int n_peer; const float *Gself; unsigned slice;
    // workgroup -> tile: layer | tile row << 4 | tile column << 18, ~0 = idle (make_tile_map below)
    const uint32_t *tile_map;
    // the same map inside the kernel arguments, 16 bits per workgroup (layer | row << 3 | column << 9, 0xffff = idle), when
    // the grid has at most TS_MAP_ARGS entries: the entry then arrives with the first batch of scalar loads instead of
    // one more round trip to memory in front of everything a workgroup does (~700 cycles in tools/tile_probe's stamps)
    int map_in_args;
    uint32_t map_words[TS_MAP_ARGS / 2];
};

// Which workgroup takes which tile.  The dispatcher deals workgroup i to XCD i % 8 and, inside an XCD, the j-th of them to
// the CU that also gets the (j + 32)-th (tools/tile_probe: HW_ID of every workgroup): with 284 live tiles for 256 CUs, 28 CUs
// run two workgroups, and a pair of full layer-0 tiles took 5.2 us where a workgroup alone takes 4.0 -- the pair, not
// the tile, set the kernel's duration.  The per-layer rectangles of XcdTiling also left two XCDs almost empty (a 13th tile
// row of 16 weights) while the others ran 36-40 tiles on 32 CUs, idle blocks in the grid taking CU slots of live ones.
// So the map is built on the host, tile by tile:
//   * full layer-0 tiles ("heavy": gradient + update + the next batch's slab) go to the XCD of their rectangle
//     (panels shared through that XCD's L2); every other tile (a partial last row, the small layers) to the XCD with the
//     fewest tiles so far;
//   * inside an XCD the lightest tiles take the slots that share a CU -- the first k and the last k when it has
//     cus_per_xcd + k tiles -- and idle entries, if any, come last.
// Placement changes which tile a workgroup computes and nothing else: results do not depend on it.
struct TileMapLayer { int M, N; }; // padded rows / columns of W_l
inline std::vector<uint32_t> make_tile_map(const TileMapLayer *layers, int n_layers, int cus_per_xcd, bool next_batch_slabs) {
    struct T { uint32_t e; int w; };
    std::vector<T> per_xcd[8];
    long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<T> light;
    for (int l = 0; l < n_layers; l++) {
        const int tiles_m = (layers[l].M + TS_TM - 1) / TS_TM, tiles_n = layers[l].N / TS_TN;
        int full_m = 0; // tile rows that count as heavy: layer 0, more than half a tile of weights
        if (l == 0) for (int tm = 0; tm < tiles_m; tm++) if (layers[l].M - tm * TS_TM > TS_TM / 2) full_m = tm + 1;
        const XcdTiling rect = full_m > 0 ? make_xcd_tiling(full_m, tiles_n) : XcdTiling{};
        for (int tm = 0; tm < tiles_m; tm++)
            for (int tn = 0; tn < tiles_n; tn++) {
                const int rows = layers[l].M - tm * TS_TM < TS_TM ? layers[l].M - tm * TS_TM : TS_TM;
                // cost, measured (tools/tile_probe): a layer-0 tile takes ~4.0 us whatever its height (the next batch's slab is
                // formed over all 128 rows either way), a tile of a later layer ~2.5 us
                const T t{(uint32_t)l | (uint32_t)tm << 4 | (uint32_t)tn << 18, (l == 0 && next_batch_slabs) ? 100 + rows / 4 : rows};
                if (l == 0 && tm < full_m) {
                    const int x = (tm / rect.rm) * rect.xn + tn / rect.rn;
                    per_xcd[x].push_back(t);
                    load[x] += t.w;
                } else light.push_back(t);
            }
    }
    std::stable_sort(light.begin(), light.end(), [](const T &a, const T &b) { return a.w > b.w; });
    for (const T &t : light) { // heaviest first, each to the XCD with the fewest tiles (ties: the least work, then the lowest id)
        int best = 0;
        for (int x = 1; x < 8; x++)
            if (per_xcd[x].size() < per_xcd[best].size() || (per_xcd[x].size() == per_xcd[best].size() && load[x] < load[best])) best = x;
        per_xcd[best].push_back(t);
        load[best] += t.w;
    }
    size_t slots = 0;
    for (int x = 0; x < 8; x++) slots = per_xcd[x].size() > slots ? per_xcd[x].size() : slots;
    std::vector<uint32_t> map(slots * 8, ~0u);
    for (int x = 0; x < 8; x++) {
        std::vector<T> &v = per_xcd[x];
        const int n = (int)v.size(), k = n > cus_per_xcd ? (n - cus_per_xcd < n / 2 ? n - cus_per_xcd : n / 2) : 0;
        std::vector<int> order(n);
        for (int i = 0; i < n; i++) order[i] = i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return v[a].w < v[b].w; }); // lightest first
        std::vector<char> is_light(n, 0);
        std::vector<uint32_t> seq;
        for (int i = k; i < 2 * k; i++) { seq.push_back(v[order[i]].e); is_light[order[i]] = 1; } // slots 0 .. k-1
        for (int i = 0; i < k; i++) is_light[order[i]] = 1;
        for (int i = 0; i < n; i++) if (!is_light[i]) seq.push_back(v[i].e);                       // the rest, in rectangle order
        for (int i = 0; i < k; i++) seq.push_back(v[order[i]].e);                                  // the last k: the lightest of all
        for (int j = 0; j < n; j++) map[(size_t)j * 8 + x] = seq[j];
    }
    return map;
}
// the 16-bit form for the kernel arguments; false when the grid or a field does not fit
inline bool pack_tile_map(const std::vector<uint32_t> &map, uint32_t (&words)[TS_MAP_ARGS / 2]) {
    if (map.size() > (size_t)TS_MAP_ARGS) return false;
    for (uint32_t &w : words) w = 0xffffffffu;
    for (size_t i = 0; i < map.size(); i++) {
        uint32_t e16 = 0xffffu;
        if (map[i] != ~0u) {
            const uint32_t li = map[i] & 15u, tm = (map[i] >> 4) & 0x3fffu, tn = map[i] >> 18;
            if (li > 7 || tm > 63 || tn > 126) return false;
            e16 = li | tm << 3 | tn << 9;
        }
        words[i >> 1] = (i & 1) ? ((words[i >> 1] & 0x0000ffffu) | e16 << 16) : ((words[i >> 1] & 0xffff0000u) | e16);
    }
    return true;
}

// 16-B store that is written THROUGH the XCD's L2 (sc1): the line does not stay dirty, so the end of the kernel has nothing
// to write back for it (a kernel boundary costs ~B / 6 TB/s for B dirty bytes, MI355X_MICROARCH.md, row `boundary`), and the
// bytes leave while the kernel still runs.  Inline asm (the compiler has no spelling for it on a 16-B vector); the s_nop
// covers the store-data hazard the compiler cannot see.  WT = false: a plain store.
// (The value must come out of an ordinary vector instruction.  Stored straight from an MFMA's destination registers the
//  asm statement read them before the matrix pipe had written them -- the wait states between an MFMA and a memory
//  instruction that reads its result are the compiler's to insert, and it cannot see into inline asm: the bf16 kernel's
//  slabs came out as garbage that changed from run to run.  ts_mfma_result() supplies them.)
__device__ __forceinline__ void ts_mfma_result(f32x4 &v) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(v)); }
template <bool WT> __device__ __forceinline__ void ts_store16(float *dst, f32x4 v) {
    if constexpr (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(dst), "v"(v) : "memory");
    else *reinterpret_cast<f32x4 *>(dst) = v;
}

// the gradient tile's 16 B of this lane when it does not come from this launch's own product
template <int GSRC> __device__ __forceinline__ float4 ts_gradient_in(const TileStepParams &p, const GradLayer &L, size_t e_off, bool e_ok) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!e_ok) return g;
    if (GSRC == 2) {
        g = *reinterpret_cast<const float4 *>(L.G + e_off);
    } else if (GSRC == 3) {
        const size_t off = (size_t)(L.G - p.Gself) + e_off;
        g = *reinterpret_cast<const float4 *>(p.Gpeer[0] + off);
#pragma unroll
        for (int r = 1; r < TS_MAX_PEERS; r++) {
            if (r < p.n_peer) { // (block-uniform; rank order: the same bits on every replica)
                const float4 o = *reinterpret_cast<const float4 *>(p.Gpeer[r] + off);
                g.x += o.x; g.y += o.y; g.z += o.z; g.w += o.w;
            }
        }
    } else if (GSRC == 4) {
        const size_t off = (size_t)(L.G - p.Gself) + e_off;
        const unsigned owner = (unsigned)off / p.slice; // (a 16-float row piece never straddles two owners)
        const float *src = p.Gpeer[0];
#pragma unroll
        for (int r = 1; r < TS_MAX_PEERS; r++) src = (owner == (unsigned)r) ? p.Gpeer[r] : src;
        g = *reinterpret_cast<const float4 *>(src + off);
    }
    return g;
}

#define GNN_TS_STAMP(i)                                                                                  \
    do {                                                                                                 \
        if (STAMP && threadIdx.x == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#define GNN_TS_STAMP_REAL(i)                                                                             \
    do {                                                                                                 \
        if (STAMP && threadIdx.x == 0) p.stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// NW: waves per workgroup, 8 or 4 (4: half the waves to dispatch for the same tiles -- the launch ramp of ~340 workgroups is
// a measurable share of this kernel -- at the price of two row groups per wave in the forward product)
// WT: 1 = the slabs, 2 = the masters too are stored write-through (ts_store16)
template <int GSRC, int GDST, bool FWD, bool STAMP = false, int NW = 8, int WT = 2>
__global__ __launch_bounds__(NW * 64) void tile_step_kernel(TileStepParams p) {
    constexpr int NT = NW * 64, RPW = 8 / NW; // threads; 16-row groups of a 128-row chunk per wave
    static_assert(NW == 8 || NW == 4, "tile_step_kernel: 8 or 4 waves");
    constexpr int LDA = TS_TM + 16;  // gradient A image [k][m]: row stride = 16 (mod 32) floats
    constexpr int LDD = TS_TN;       // delta image [k][n]: 16 floats (lanes 16-31 land on banks 16-31)
    constexpr int LDW = TS_TN + 4;   // weight tile / partial tiles [m][n]
    __shared__ __attribute__((aligned(16))) float sA[TS_KC * LDA];       // A chunk [128][64]
    __shared__ __attribute__((aligned(16))) float sD[TS_KC * LDD];       // delta chunk [128][16]
    __shared__ __attribute__((aligned(16))) float sW[TS_TM * LDW];       // the tile's (new) weights (53 KB in all)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fq = lane >> 4;

    // by value: one batch of scalar loads, not one round trip per field as it is first used; layer 0's descriptor (nine of
    // ten workgroups) is requested at once, beside the map entry that names the tile, not behind it
    int li, tm, tn;
    GradLayer L = p.layer[0];
    if (p.map_in_args) {
        const uint32_t w = p.map_words[blockIdx.x >> 1], e = (blockIdx.x & 1) ? w >> 16 : w & 0xffffu;
        if (e == 0xffffu) return;
        li = (int)(e & 7u); tm = (int)((e >> 3) & 63u); tn = (int)(e >> 9);
    } else {
        const uint32_t e = p.tile_map[blockIdx.x];
        if (e == ~0u) return;
        li = (int)(e & 15u); tm = (int)((e >> 4) & 0x3fffu); tn = (int)(e >> 18);
    }
    if (li != 0) L = p.layer[li];
    const int m0 = tm * TS_TM, n0 = tn * TS_TN;
    const bool fwd = FWD && li == 0; // block-uniform

    // this thread's 16 B of the weight tile (waves 0..3): row er = 16*wave + fr, columns 4*fq .. 4*fq+3 -- the
    // accumulator layout of the transposed gradient product below, so G never leaves its registers
    const int er = (t >> 6) * 16 + fr, eq = fq;
    const bool e_ok = t < 256 && (m0 + er < L.M);
    const size_t e_off = (size_t)(m0 + er) * L.ldd + n0 + eq * 4;

    GNN_TS_STAMP(0);
    GNN_TS_STAMP_REAL(8);
    if (STAMP && threadIdx.x == 0) { // where this workgroup runs: HW_ID (CU, SE, ..) and the XCC id
        p.stamps[blockIdx.x * 16 + 10] = (unsigned)__builtin_amdgcn_s_getreg(0xF804);
        p.stamps[blockIdx.x * 16 + 11] = (unsigned)__builtin_amdgcn_s_getreg(0xF814);
    }
    // ---- everything this block reads first, all loads in flight together ------------------------
    // gradient operands of the first K chunk
    float4 va[4 * RPW], vd[RPW];
    const int kc0 = (p.K < TS_KC) ? p.K : TS_KC;
    // A tile wholly inside its layer, a whole first chunk, rows in place: no bounds tests, no exec-mask branches -- every wave
    // runs this prologue before the first barrier, and a guarded 16-B load is ~13 instructions (see gemm_f32_kernel)
    const bool interior = (m0 + TS_TM <= L.M) && (p.K >= TS_KC) && !(li == 0 && p.row_idx) &&
                          (unsigned long long)TS_KC * (unsigned)(L.lda > L.ldd ? L.lda : L.ldd) < 0xffffffffull; // 32-bit offsets
    if (GSRC == 1 && interior) {
#pragma unroll
        for (int i = 0; i < 4 * RPW; i++) {
            const int idx = t + i * NT, k = idx >> 4, q = idx & 15;
            va[i] = *reinterpret_cast<const float4 *>(L.A + ((unsigned)k * (unsigned)L.lda + m0 + q * 4));
        }
#pragma unroll
        for (int i = 0; i < RPW; i++) {
            const int idx = t + i * NT;
            vd[i] = *reinterpret_cast<const float4 *>(L.D + ((unsigned)(idx >> 2) * (unsigned)L.ldd + n0 + (idx & 3) * 4));
        }
    } else if (GSRC == 1) {
#pragma unroll
        for (int i = 0; i < 4 * RPW; i++) {
            const int idx = t + i * NT, k = idx >> 4, q = idx & 15;
            va[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kc0 && m0 + q * 4 < L.M) {
                size_t a_row = (size_t)k;
                bool live = true;
                if (li == 0 && p.row_idx) { live = k < p.k_true; a_row = live ? (size_t)p.row_idx[k] : 0; }
                if (live) va[i] = *reinterpret_cast<const float4 *>(L.A + a_row * L.lda + m0 + q * 4);
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; i++) {
            const int idx = t + i * NT, k = idx >> 2, q = idx & 3;
            vd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < kc0) vd[i] = *reinterpret_cast<const float4 *>(L.D + (size_t)k * L.ldd + n0 + q * 4);
        }
    }
    float4 w_old = make_float4(0.f, 0.f, 0.f, 0.f), v_old = w_old, g_in = w_old;
    if (e_ok) {
        if (GDST == 2 || FWD) w_old = *reinterpret_cast<const float4 *>(L.W + e_off);
        if (GDST == 2) v_old = *reinterpret_cast<const float4 *>(L.V + e_off);
    }
    if (GSRC >= 2) g_in = ts_gradient_in<GSRC>(p, L, e_off, e_ok);
    const bool next_plain = interior && fwd && !p.next_idx && p.next_rows >= TS_KC && (unsigned long long)TS_KC * (unsigned)p.ldan < 0xffffffffull; // the first chunk of the next batch: all rows live, in place
    const bool next_gather = (m0 + TS_TM <= L.M) && fwd && p.next_idx && p.next_rows >= TS_KC;
    const int stage_cols = (L.N / TS_TN) < 8 ? (L.N / TS_TN) : 8; // tile columns that share the staging copy's 16-row groups
    // the next batch's rows, already in MFMA fragment form: wave -> 16 batch rows of a 128-row chunk, lane
    // (fr, fq) -> row fr, inputs 16c + 4fq .. +3 of the tile (c = 0..3).  Straight to registers: A_0' is
    // k-contiguous, no wave shares another's rows, and the product below needs no LDS image of it.
    f32x4 vn[RPW][4];
    // a sampled next batch: this lane's row index of the first chunk is fetched NOW, so that the row loads, requested
    // ~3 000 cycles from here, do not start with a dependent round trip
    int next_row0[RPW];
#pragma unroll
    for (int g = 0; g < RPW; g++) {
        next_row0[g] = (wave + g * NW) * 16 + fr;
        if (fwd && p.next_idx && next_row0[g] < p.next_rows) next_row0[g] = p.next_idx[next_row0[g]];
    }
    auto load_next = [&](int b0) {
#pragma unroll
        for (int g = 0; g < RPW; g++) {
            const int b = b0 + (wave + g * NW) * 16 + fr;
            if (b0 == 0 && next_plain) { // (block-uniform)
                const float *src = p.An + ((unsigned)b * (unsigned)p.ldan + m0 + 4 * fq);
#pragma unroll
                for (int c = 0; c < 4; c++) vn[g][c] = *reinterpret_cast<const f32x4 *>(src + c * 16);
                continue;
            }
            if (b0 == 0 && next_gather) { // (block-uniform) a sampled batch, every row live, the tile inside the layer: the same
                                          // four loads from the row the index names (fetched at the top), no bounds tests
                const float *src = p.An + ((size_t)next_row0[g] * p.ldan + m0 + 4 * fq);
#pragma unroll
                for (int c = 0; c < 4; c++) vn[g][c] = *reinterpret_cast<const f32x4 *>(src + c * 16);
                continue;
            }
            const bool live = b < p.next_rows;
            const size_t row = live ? (b0 == 0 ? (size_t)next_row0[g] : p.next_idx ? (size_t)p.next_idx[b] : (size_t)b) : 0;
            const float *src = p.An + row * p.ldan + m0 + 4 * fq;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                vn[g][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (live && m0 + c * 16 + 4 * fq < L.M) vn[g][c] = *reinterpret_cast<const f32x4 *>(src + c * 16);
            }
        }
    };
    // (requested behind the gradient operands, below: it is needed ~2 500 cycles later, and in front of them it
    //  delayed the first barrier by the time its 32 KB take to cross the CU's load path)
    if (fwd && GSRC != 1) load_next(0);

    // ---- gradient tile, transposed: G^T[n][m] = sum_k D[k][n] A[k][m] -----------------------------
    // Waves 0..3 take one 16-wide m tile each over the whole K chunk; lane (fr, fq) ends with G[m = 16 wave + fr]
    // [n = 4 fq .. 4 fq + 3]: the 16 B of W and V it loaded at the top.  No partial tiles, no second barrier.
    // (Waves 4..7 sit on the same four SIMDs: splitting K over them bought no MFMA time and cost an LDS round trip.)
    float4 g = g_in;
    if (GSRC == 1) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < p.K; k0 += TS_KC) {
            const int kc = (p.K - k0 < TS_KC) ? p.K - k0 : TS_KC; // a multiple of 16
            if (k0) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4 * RPW; i++) {
                    const int idx = t + i * NT, k = idx >> 4, q = idx & 15;
                    va[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < kc && m0 + q * 4 < L.M) {
                        size_t a_row = (size_t)(k0 + k);
                        bool live = true;
                        if (li == 0 && p.row_idx) { live = k0 + k < p.k_true; a_row = live ? (size_t)p.row_idx[k0 + k] : 0; }
                        if (live) va[i] = *reinterpret_cast<const float4 *>(L.A + a_row * L.lda + m0 + q * 4);
                    }
                }
#pragma unroll
                for (int i = 0; i < RPW; i++) {
                    const int idx = t + i * NT, k = idx >> 2, q = idx & 3;
                    vd[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (k < kc) vd[i] = *reinterpret_cast<const float4 *>(L.D + (size_t)(k0 + k) * L.ldd + n0 + q * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 4 * RPW; i++) {
                const int idx = t + i * NT, k = idx >> 4, q = idx & 15;
                *reinterpret_cast<float4 *>(&sA[k * LDA + q * 4]) = va[i];
            }
#pragma unroll
            for (int i = 0; i < RPW; i++) {
                const int idx = t + i * NT;
                *reinterpret_cast<float4 *>(&sD[(idx >> 2) * LDD + (idx & 3) * 4]) = vd[i];
            }
            __syncthreads();
            if (k0 == 0) {
                GNN_TS_STAMP(1);
                if (fwd) load_next(0); // lands under the gradient MFMAs and the update
            }
            if (wave < 4) {
                const float *ap = &sA[fq * LDA + wave * 16 + fr];
                const float *dp = &sD[fq * LDD + fr];
                int kk = 0;
                if (kc == TS_KC) {
                    // a whole chunk (every step at B = 128): the operands of trip t + 1 are requested before the MFMAs of trip t --
                    // trip by trip, the first MFMA of every trip waited out an LDS round trip that nothing covered (one wave per
                    // SIMD here).  The same MFMAs on the same accumulators in the same order as the loop below.  (All 64 reads up
                    // front took 148 registers: two workgroups no longer fit a CU, and 28 CUs hold two.)
                    float a[2][8], d[2][8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { a[0][j] = ap[(4 * j) * LDA]; d[0][j] = dp[(4 * j) * LDD]; }
#pragma unroll
                    for (int trip = 0; trip < TS_KC / 32; trip++) {
                        const int cur = trip & 1;
                        if (trip + 1 < TS_KC / 32) {
#pragma unroll
                            for (int j = 0; j < 8; j++) { a[cur ^ 1][j] = ap[(32 * (trip + 1) + 4 * j) * LDA]; d[cur ^ 1][j] = dp[(32 * (trip + 1) + 4 * j) * LDD]; }
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 8; j += 2) {
                            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[cur][j], a[cur][j], acc0, 0, 0, 0);
                            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[cur][j + 1], a[cur][j + 1], acc1, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    kk = TS_KC;
                }
                for (; kk + 32 <= kc; kk += 32) { // 8 MFMAs per trip, the trip's 16 LDS reads issued first
                    float a[8], d[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { a[j] = ap[(kk + 4 * j) * LDA]; d[j] = dp[(kk + 4 * j) * LDD]; }
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j], a[j], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(d[j + 1], a[j + 1], acc1, 0, 0, 0);
                    }
                }
                for (; kk < kc; kk += 4)
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(dp[kk * LDD], ap[kk * LDA], acc0, 0, 0, 0);
            }
        }
        const f32x4 acc = acc0 + acc1; // rows n = 4 fq + r, column m = 16 wave + fr
        GNN_TS_STAMP(2);
        g = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }

    // ---- store G, or the momentum update (SCE:333-339) -------------------------------------------
    float4 w_new = w_old;
    if (GDST == 1) {
        if (e_ok) *reinterpret_cast<float4 *>(L.G + e_off) = g;
    } else if (GDST == 2) {
        float4 adj; // ((step*G)/B) + (momentum*prev)
        adj.x = sgd_adj(p.step_over_b, g.x, p.momentum, v_old.x);
        adj.y = sgd_adj(p.step_over_b, g.y, p.momentum, v_old.y);
        adj.z = sgd_adj(p.step_over_b, g.z, p.momentum, v_old.z);
        adj.w = sgd_adj(p.step_over_b, g.w, p.momentum, v_old.w);
        w_new = make_float4(w_old.x - adj.x, w_old.y - adj.y, w_old.z - adj.z, w_old.w - adj.w);
        if (e_ok) {
            ts_store16<(WT >= 2)>(L.W + e_off, (f32x4){w_new.x, w_new.y, w_new.z, w_new.w});
            ts_store16<(WT >= 2)>(L.V + e_off, (f32x4){adj.x, adj.y, adj.z, adj.w});
        }
    }
    GNN_TS_STAMP(3);
    if (!fwd) { GNN_TS_STAMP_REAL(9); return; }

    // ---- the next batch's first-layer sums over this tile's 64 input neurons ---------------------
    // Computed TRANSPOSED, Zp^T[n][b] = sum_m W[m][n] A'[b][m]: the accumulator then holds four
    // consecutive n of one batch row per lane -- a 16-B store each, no trip through LDS.
    // MFMA j of chunk c: slot q holds k = 16c + 4q + j on both operands.
    if (t < 256) *reinterpret_cast<float4 *>(&sW[er * LDW + eq * 4]) = e_ok ? w_new : make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    GNN_TS_STAMP(4);
    float *slab = p.slabs + (size_t)tm * p.slab_rows * p.ldz;
    const float *wcol = &sW[(4 * fq) * LDW + fr];
    float wv[4][4];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int j = 0; j < 4; j++) wv[c][j] = wcol[(c * 16 + j) * LDW];
    for (int b0 = 0; b0 < p.next_K; b0 += TS_KC) {
        if (b0) load_next(b0);
#pragma unroll
        for (int g = 0; g < RPW; g++) {
            const int wr = (wave + g * NW) * 16; // this wave's row group of the chunk
            // the contiguous copy of a sampled next batch: every tile of this tile row holds the same rows; column tn copies
            // the 16-row groups g16 with g16 % n_tn == tn (all of them in one column made its 13 workgroups the kernel's last)
            if (p.stage_out && b0 + wr < p.next_K && (((b0 + wr) >> 4) % stage_cols) == tn % stage_cols && tn < stage_cols) { // (wave-uniform; rows past the batch and columns past M are zeros in vn)
                float *dst = p.stage_out + (size_t)(b0 + wr + fr) * p.ldan + m0 + 4 * fq;
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (m0 + c * 16 + 4 * fq < L.M) *reinterpret_cast<f32x4 *>(dst + c * 16) = vn[g][c];
            }
            if (b0 + wr < p.next_K) { // wave-uniform
                f32x4 z0 = {0.f, 0.f, 0.f, 0.f}, z1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < 4; c++) {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (c & 1) z1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][j], vn[g][c][j], z1, 0, 0, 0);
                        else z0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[c][j], vn[g][c][j], z0, 0, 0, 0);
                    }
                }
                const f32x4 z = z0 + z1; // rows n = 4*fq + r, column b = fr
                GNN_TS_STAMP(5);
                ts_store16<(WT >= 1)>(slab + (size_t)(b0 + wr + fr) * p.ldz + n0 + 4 * fq, z);
            }
        }
    }
    GNN_TS_STAMP(6);
    GNN_TS_STAMP_REAL(9);
}

// ------------------------------------------------------------------------------------------------
// tile_step_bf16_kernel: the same tile owner with bf16 GEMM operands (GNN_DTYPE_BF16): activations, deltas,
// next-batch rows and the weight tile enter the two products as bf16 (written once by their producers), both
// products run on v_mfma_f32_16x16x32_bf16 with f32 accumulation, the update is on the f32 masters and the
// tile's new weights go back to the bf16 shadow.  128 f32 MFMAs per tile become 16 (gradient) + 16 (forward);
// the operand images are half the bytes.  Both gradient operands are k-major ([batch row][neuron]): staged as
// they are and read with ds_read_b64_tr_b16; the weight tile [m][n] is k-major for the forward product too.
// ------------------------------------------------------------------------------------------------
template <int GSRC, int GDST, bool FWD>
__global__ __launch_bounds__(TS_THREADS) void tile_step_bf16_kernel(TileStepParams p) {
    constexpr int LDA = TS_TM + 16;  // [k][m] bf16 image: 160-B rows (32*odd)
    constexpr int LDD = TS_TN;       // [k][n] bf16 image: 32-B rows
    __shared__ __attribute__((aligned(16))) __bf16 sA[TS_KC * LDA];
    __shared__ __attribute__((aligned(16))) __bf16 sD[TS_KC * LDD];
    __shared__ __attribute__((aligned(16))) __bf16 sW[TS_TM * TS_TN]; // the tile's (new) weights [m][n], 32-B rows
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int fr = lane & 15, fg = lane >> 4;

    // by value: one batch of scalar loads, not one round trip per field as it is first used; layer 0's descriptor (nine of
    // ten workgroups) is requested at once, beside the map entry that names the tile, not behind it
    int li, tm, tn;
    GradLayer L = p.layer[0];
    if (p.map_in_args) {
        const uint32_t w = p.map_words[blockIdx.x >> 1], e = (blockIdx.x & 1) ? w >> 16 : w & 0xffffu;
        if (e == 0xffffu) return;
        li = (int)(e & 7u); tm = (int)((e >> 3) & 63u); tn = (int)(e >> 9);
    } else {
        const uint32_t e = p.tile_map[blockIdx.x];
        if (e == ~0u) return;
        li = (int)(e & 15u); tm = (int)((e >> 4) & 0x3fffu); tn = (int)(e >> 18);
    }
    if (li != 0) L = p.layer[li];
    const int m0 = tm * TS_TM, n0 = tn * TS_TN;
    const bool fwd = FWD && li == 0;
    const __bf16 *Ab = p.Ab[li], *Db = p.Db[li];
    const int stage_cols = (L.N / TS_TN) < 8 ? (L.N / TS_TN) : 8; // tile columns that share the staging copy's 16-row groups

    const int er = (t >> 6) * 16 + fr, eq = fg; // (as in tile_step_kernel: the accumulator layout of the transposed product)
    const bool e_ok = t < 256 && (m0 + er < L.M);
    const size_t e_off = (size_t)(m0 + er) * L.ldd + n0 + eq * 4;

    // ---- loads: gradient operands of the first K chunk, the tile's masters, then the next batch's rows ----
    bf16x8 va[2], vd;
    const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
    auto load_grad = [&](int k0, int kc) {
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int idx = t + i * TS_THREADS, k = idx >> 3, q = idx & 7; // 8 chunks of 8 bf16 per row
            va[i] = zero8;
            if (k < kc && m0 + q * 8 < L.M) {
                size_t a_row = (size_t)(k0 + k);
                bool live = true;
                if (li == 0 && p.row_idx) { live = k0 + k < p.k_true; a_row = live ? (size_t)p.row_idx[k0 + k] : 0; }
                if (live) va[i] = *reinterpret_cast<const bf16x8 *>(Ab + a_row * L.lda + m0 + q * 8);
            }
        }
        vd = zero8;
        if (t < 256) {
            const int k = t >> 1, q = t & 1;
            if (k < kc) vd = *reinterpret_cast<const bf16x8 *>(Db + (size_t)(k0 + k) * L.ldd + n0 + q * 8);
        }
    };
    const int kc0 = (p.K < TS_KC) ? p.K : TS_KC;
    if (GSRC == 1) load_grad(0, kc0);
    float4 w_old = make_float4(0.f, 0.f, 0.f, 0.f), v_old = w_old, g_in = w_old;
    if (e_ok) {
        if (GDST == 2 || FWD) w_old = *reinterpret_cast<const float4 *>(L.W + e_off);
        if (GDST == 2) v_old = *reinterpret_cast<const float4 *>(L.V + e_off);
    }
    if (GSRC >= 2) g_in = ts_gradient_in<GSRC>(p, L, e_off, e_ok);
    // next batch: lane (fr, fg) of wave w -> row 16w + fr; per 32-wide k block the inputs 4fg..4fg+3 and 16+4fg..+3
    s16x4 vn[2][2];
    int next_row0 = wave * 16 + fr; // (as in tile_step_kernel: the index of a sampled next batch's row is fetched ahead)
    if (fwd && p.next_idx && next_row0 < p.next_rows) next_row0 = p.next_idx[next_row0];
    auto load_next = [&](int b0) {
        const int b = b0 + wave * 16 + fr;
        const bool live = b < p.next_rows;
        const size_t row = live ? (b0 == 0 ? (size_t)next_row0 : p.next_idx ? (size_t)p.next_idx[b] : (size_t)b) : 0;
        const __bf16 *src = p.Anb + row * p.ldan + m0 + 4 * fg;
#pragma unroll
        for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                vn[kb][hh] = (s16x4){0, 0, 0, 0};
                if (live && m0 + kb * 32 + hh * 16 + 4 * fg < L.M) vn[kb][hh] = *reinterpret_cast<const s16x4 *>(src + kb * 32 + hh * 16);
            }
    };
    if (fwd) load_next(0);

    // ---- gradient tile ------------------------------------------------------------------------------
    float4 g = g_in;
    if (GSRC == 1) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        for (int k0 = 0; k0 < p.K; k0 += TS_KC) {
            const int kc = (p.K - k0 < TS_KC) ? p.K - k0 : TS_KC;
            if (k0) { __syncthreads(); load_grad(k0, kc); }
#pragma unroll
            for (int i = 0; i < 2; i++) {
                const int idx = t + i * TS_THREADS, k = idx >> 3, q = idx & 7;
                *reinterpret_cast<bf16x8 *>(&sA[k * LDA + q * 8]) = va[i];
            }
            if (t < 256) *reinterpret_cast<bf16x8 *>(&sD[(t >> 1) * LDD + (t & 1) * 8]) = vd;
            __syncthreads();
            // waves 0..3: G^T[n][m] over the chunk's 32-wide k blocks (rows past kc were staged as zeros)
            if (wave < 4) {
                for (int kk = 0; kk < kc; kk += 64) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sD, LDD, 0, kk, lane), tr_frag(sA, LDA, wave * 16, kk, lane), acc0, 0, 0, 0);
                    if (kk + 32 < kc)
                        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag(sD, LDD, 0, kk + 32, lane), tr_frag(sA, LDA, wave * 16, kk + 32, lane), acc1, 0, 0, 0);
                }
            }
        }
        const f32x4 acc = acc0 + acc1; // rows n = 4 fg + r, column m = 16 wave + fr
        g = make_float4(acc[0], acc[1], acc[2], acc[3]);
    }

    float4 w_new = w_old;
    if (GDST == 1) {
        if (e_ok) *reinterpret_cast<float4 *>(L.G + e_off) = g;
    } else if (GDST == 2) {
        float4 adj; // ((step*G)/B) + (momentum*prev), SCE:333, on the f32 masters
        adj.x = sgd_adj(p.step_over_b, g.x, p.momentum, v_old.x);
        adj.y = sgd_adj(p.step_over_b, g.y, p.momentum, v_old.y);
        adj.z = sgd_adj(p.step_over_b, g.z, p.momentum, v_old.z);
        adj.w = sgd_adj(p.step_over_b, g.w, p.momentum, v_old.w);
        w_new = make_float4(w_old.x - adj.x, w_old.y - adj.y, w_old.z - adj.z, w_old.w - adj.w);
        if (e_ok) {
            ts_store16<true>(L.W + e_off, (f32x4){w_new.x, w_new.y, w_new.z, w_new.w}); // (write-through: see ts_store16)
            ts_store16<true>(L.V + e_off, (f32x4){adj.x, adj.y, adj.z, adj.w});
            *reinterpret_cast<bf16x4 *>(p.Wb[li] + e_off) = (bf16x4){(__bf16)w_new.x, (__bf16)w_new.y, (__bf16)w_new.z, (__bf16)w_new.w};
        }
    }
    if (!fwd) return;

    // ---- next batch's first-layer sums over this tile's 64 inputs, transposed: Zp^T[n][b] = sum_m W[m][n] A'[b][m] ----
    if (t < 256) {
        const float4 w = e_ok ? w_new : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<bf16x4 *>(&sW[er * TS_TN + eq * 4]) = (bf16x4){(__bf16)w.x, (__bf16)w.y, (__bf16)w.z, (__bf16)w.w};
    }
    __syncthreads();
    float *slab = p.slabs + (size_t)tm * p.slab_rows * p.ldz;
    const bf16x8 w0 = tr_frag(sW, TS_TN, 0, 0, lane), w1 = tr_frag(sW, TS_TN, 0, 32, lane); // A operand: rows n, k = m
    for (int b0 = 0; b0 < p.next_K; b0 += TS_KC) {
        if (b0) load_next(b0);
        if (p.stage_out_b && tn < stage_cols && (((b0 + wave * 16) >> 4) % stage_cols) == tn && b0 + wave * 16 < p.next_K) { // (as in tile_step_kernel)
            __bf16 *dst = p.stage_out_b + (size_t)(b0 + wave * 16 + fr) * p.ldan + m0 + 4 * fg;
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int hh = 0; hh < 2; hh++)
                    if (m0 + kb * 32 + hh * 16 + 4 * fg < L.M) *reinterpret_cast<s16x4 *>(dst + kb * 32 + hh * 16) = vn[kb][hh];
        }
        if (b0 + wave * 16 < p.next_K) { // wave-uniform
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0, join8(vn[0][0], vn[0][1]), z, 0, 0, 0);
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, join8(vn[1][0], vn[1][1]), z, 0, 0, 0);
            ts_mfma_result(z);
            ts_store16<true>(slab + (size_t)(b0 + wave * 16 + fr) * p.ldz + n0 + 4 * fg, z); // rows n = 4fg + r, column b = fr
        }
    }
}

} // namespace gnn
