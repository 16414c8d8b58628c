// Host-only check of make_tile_map (csrc/tile_step_kernel.h): prints, for a list of layer shapes, the grid and -- per XCD --
// how many tiles it got and which classes sit in the CU-sharing slots.  Compiled and run by tests/test_tile_map.py (no GPU).
#include "../../graph-neural-net_amd/csrc/tile_step_kernel.h"
#include <cstdio>
#include <cstdlib>
#include <set>
using namespace gnn;
int main(int argc, char **argv) {
    // usage: tile_map_check cus_per_xcd d0 d1 .. dL-1   (padded sizes are derived as the library does: multiples of 16)
    if (argc < 4) return 2;
    const int cus = atoi(argv[1]);
    std::vector<int> ld;
    for (int i = 2; i < argc; i++) ld.push_back((atoi(argv[i]) + 15) / 16 * 16);
    std::vector<TileMapLayer> layers;
    for (size_t l = 0; l + 1 < ld.size(); l++) layers.push_back(TileMapLayer{ld[l], ld[l + 1]});
    const std::vector<uint32_t> map = make_tile_map(layers.data(), (int)layers.size(), cus, true);
    std::set<uint32_t> seen;
    size_t live = 0, expected = 0;
    for (const TileMapLayer &t : layers) expected += (size_t)((t.M + TS_TM - 1) / TS_TM) * (size_t)(t.N / TS_TN);
    bool dup = false;
    for (uint32_t e : map) if (e != ~0u) { live++; dup |= !seen.insert(e).second; }
    printf("grid %zu live %zu expected %zu duplicates %d\n", map.size(), live, expected, (int)dup);
    uint32_t words[TS_MAP_ARGS / 2];
    const bool packed = pack_tile_map(map, words);
    bool pack_ok = packed;
    if (packed)
        for (size_t i = 0; i < map.size(); i++) {
            const uint32_t w = words[i >> 1], e = (i & 1) ? w >> 16 : w & 0xffffu;
            const uint32_t back = e == 0xffffu ? ~0u : ((e & 7u) | ((e >> 3) & 63u) << 4 | (e >> 9) << 18);
            pack_ok &= back == map[i];
        }
    printf("packed %d roundtrip %d\n", (int)packed, (int)pack_ok);
    for (int x = 0; x < 8; x++) {
        int n = 0, idle_before_live = 0, shared_heavy_pairs = 0;
        const int slots = (int)map.size() / 8;
        bool seen_idle = false;
        for (int j = 0; j < slots; j++) {
            const uint32_t e = map[(size_t)j * 8 + x];
            if (e == ~0u) seen_idle = true; else { n++; if (seen_idle) idle_before_live++; }
        }
        for (int j = cus; j < slots; j++) { // slot j shares a CU with slot j - cus: count pairs of two full layer-0 tiles
            const uint32_t a = map[(size_t)j * 8 + x], b = map[(size_t)(j - cus) * 8 + x];
            if (a == ~0u || b == ~0u) continue;
            auto heavy = [&](uint32_t e) { return (e & 15u) == 0; };
            if (heavy(a) && heavy(b)) shared_heavy_pairs++;
        }
        printf("xcd %d tiles %d idle_before_live %d layer0_pairs %d\n", x, n, idle_before_live, shared_heavy_pairs);
    }
    return 0;
}
