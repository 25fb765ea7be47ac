// Checks layout.h's TileOrder on the host: for every frame shape, rank count and stripe width tried, item-order tile
// j -> local tile l must be a permutation of the rank's tiles.  Built and run by tests/test_tile_order.py (g++, no GPU).
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../rayrs_amd/csrc/layout.h"
using namespace rayrs;

static long check(uint32_t n_local, uint32_t tiles_x, uint32_t ranks, uint32_t cols) {
    const TileOrder to = make_tile_order(n_local, tiles_x, ranks, cols);
    std::vector<int> seen(n_local, 0);
    long bad = 0;
    for (uint32_t j = 0; j < n_local; j++) {
        const uint32_t l = local_tile_of(to, j);
        if (l >= n_local) bad++;
        else seen[l]++;
    }
    for (uint32_t l = 0; l < n_local; l++) bad += seen[l] != 1;
    return bad;
}

int main() {
    long bad = 0, cases = 0;
    for (uint32_t tx = 1; tx <= 40; tx++)
        for (uint32_t ty = 1; ty <= 12; ty++)
            for (uint32_t ranks = 1; ranks <= 9; ranks++)
                for (uint32_t rank = 0; rank < ranks; rank++) {
                    const uint32_t n_tiles = tx * ty;
                    if (n_tiles <= rank) continue;
                    const uint32_t n_local = (n_tiles - rank + ranks - 1) / ranks;
                    for (uint32_t cols : {0u, 1u, 2u, 3u, 5u, 7u, 8u, 1000u}) bad += check(n_local, tx, ranks, cols), cases++;
                }
    // the headline frame (256 x 256 tiles) alone and as one rank of eight, one stripe per stream
    bad += check(65536, 256, 1, 32) + check(8192, 256, 8, 4) + check(65536, 256, 1, 1) + check(21846, 256, 3, 37);
    const TileOrder to = make_tile_order(65536, 256, 1, 32);
    // a stripe is walked row by row: 32 tiles of the first row, then the second row's; the second stripe starts at column 32
    bad += local_tile_of(to, 31) != 31 || local_tile_of(to, 32) != 256 || local_tile_of(to, 8192) != 32;
    std::printf("cases %ld bad %ld\n", cases + 4, bad);
    return bad != 0;
}
