// Host-side check of the tile-blocked addressing of the per-call layers (groundgrid_amd/csrc/gg_internal.h): every (layer, cell) of
// a slot has its own element, a tile's block is contiguous, K3's three layers come first, the liveness bit of a cell is its half
// column.  Compiled host-only by tests/test_layout_cpu.py; prints "ok" or the first violation.
#include <cstdio>
#include <vector>

#include "gg_internal.h"

int main()
{
    using namespace gg;
    const int layers[9] = {GG_LAYER_POINTS,        GG_LAYER_VARIANCE,     GG_LAYER_MINGROUNDHEIGHT, GG_LAYER_M2,       GG_LAYER_POINTSRAW,
                           GG_LAYER_MEANVARIANCE, GG_LAYER_MAXGROUNDHEIGHT, GG_LAYER_GROUNDCANDIDATES, GG_LAYER_PLANEDIST};
    bool seen[PERCALL_LAYERS] = {};
    for (int k = 0; k < 9; ++k) {
        const int p = percall_position(layers[k]);
        if (p < 0 || p >= PERCALL_LAYERS || seen[p]) return std::printf("position of layer %d\n", layers[k]), 1;
        seen[p] = true;
    }
    if (percall_position(GG_LAYER_GROUND) != -1 || percall_position(GG_LAYER_GROUNDPATCH) != -1) return std::printf("persistent layers\n"), 1;
    if (percall_position(GG_LAYER_POINTS) != 0 || percall_position(GG_LAYER_VARIANCE) != 1 || percall_position(GG_LAYER_MINGROUNDHEIGHT) != 2)
        return std::printf("k_patch's layers first\n"), 1;
    // the three layers GG_FLAG_MINIMAL_LAYERS leaves out are the last three positions
    for (int l : {GG_LAYER_MAXGROUNDHEIGHT, GG_LAYER_GROUNDCANDIDATES, GG_LAYER_PLANEDIST})
        if (percall_position(l) < 6) return std::printf("minimal layers\n"), 1;
    const int T = 23 * 23;
    std::vector<unsigned char> hit((size_t)T * PERCALL_BLOCK, 0);
    for (int rank = 0; rank < T; ++rank)
        for (int p = 0; p < PERCALL_LAYERS; ++p)
            for (int cell = 0; cell < TILE * TILE; ++cell) {
                const size_t at = percall_index(rank, p, cell);
                if (at >= hit.size() || hit[at]) return std::printf("element %d %d %d\n", rank, p, cell), 1;
                if (at / PERCALL_BLOCK != (size_t)rank) return std::printf("block of %d\n", rank), 1;
                hit[at] = 1;
            }
    for (int cell = 0; cell < TILE * TILE; ++cell) {
        const int row = cell % TILE, col = cell / TILE;
        if (live_bit(cell) != col * 2 + row / 8 || live_bit(cell) < 0 || live_bit(cell) >= 32) return std::printf("live bit of %d\n", cell), 1;
    }
    std::printf("ok\n");
    return 0;
}
