// K7 -- the single-cell members of the reference's class, detect_ground_patch<S>(map, i, j) (src/GroundSegmentation.cpp:343-395) and
// interpolate_cell(map, x, y) (:445-465), as stages of their own (gg_run_stage).  Nothing on the path calls them one cell at a time
// -- k_patch and k_sweep are their many-cell forms -- but they are public in include/groundgrid/GroundSegmentation.h:60,62, so the
// drop-in library answers them: one lane, the slot's layers as they stand, the expressions of the reference in its order and
// precision (oracle/gg_oracle.c detect_ground_patch / interpolate_cell are the CPU twins).
#include "gg_device.h"

#include <float.h>

namespace gg {

namespace {

// a per-call layer's value at (row, col): stored in the live half columns, the per-call reset value elsewhere (gg_internal.h tile_live)
GG_DEV float percall_at(const Arena &a, int slot, int layer, int row, int col)
{
    return cell_is_live(a, slot, row, col) ? percall_ptr(a, slot)[percall_index_of(a, percall_position(layer), row, col)] : layer_reset_value(layer);
}

template <int S> GG_DEV float block_sum(const Arena &a, const float *e)
{
    if (S == 3) return tree9(e);
    return a.eigen_reduction == GG_EIGEN_34_SSE ? tree25_eigen34(e) : tree25(e);
}

template <int S> GG_DEV void detect_cell(const Arena &a, int slot, int i, int j)
{
    constexpr int ci = S / 2, SS = S * S; // :352
    const DevConfig &cfg = a.cfg;
    float pts[SS], var[SS], mn[SS], prod[SS];
    for (int s = 0; s < SS; ++s) pts[s] = percall_at(a, slot, GG_LAYER_POINTS, i - ci + s % S, j - ci + s / S); // :355
    const double di = (double)i - (double)a.g.rows / 2.0, dj = (double)j - (double)a.g.cols / 2.0;
    const float sqdist = (float)((di * di + dj * dj) * ((double)a.g.resolution_f * (double)a.g.resolution_f)); // :356
    const float expected = a.expected[(size_t)i + (size_t)j * a.g.rows];                                         // :358
    const float pointsblockSum = block_sum<S>(a, pts);                                                           // :359
    float2 *cell = gp2_ptr(a, slot) + gp_idx(a, i, j);
    const float2 old = *cell;
    const float oldConfidence = old.y, oldGroundheight = old.x; // :360-361
    if ((double)pointsblockSum < std_max(floor(cfg.gpd_min_point_count_threshold * (double)S * (double)expected), 3.0)) return; // :364-365
    for (int s = 0; s < SS; ++s) { // :370-371
        var[s] = percall_at(a, slot, GG_LAYER_VARIANCE, i - ci + s % S, j - ci + s / S);
        mn[s] = percall_at(a, slot, GG_LAYER_MINGROUNDHEIGHT, i - ci + s % S, j - ci + s / S);
    }
    const float varThresholdsq = (float)std_min(std_max((double)sqdist * cfg.distance_factor_sq, cfg.minimum_distance_factor_sq), cfg.minimum_distance_factor_x10_sq); // :369
    float localmin = mn[0]; // :373
    for (int s = 1; s < SS; ++s)
        if (mn[s] < localmin) localmin = mn[s];
    float maxVar; // :374
    if (pts[ci + ci * S] >= (float)cfg.point_count_cell_variance_threshold) {
        maxVar = var[ci + ci * S];
    } else {
        for (int s = 0; s < SS; ++s) prod[s] = pts[s] * var[s];
        maxVar = block_sum<S>(a, prod) / pointsblockSum;
    }
    for (int s = 0; s < SS; ++s) prod[s] = pts[s] * mn[s];
    const float groundlevel = block_sum<S>(a, prod) / pointsblockSum;                                    // :375
    const float groundDiff = std_max((groundlevel - oldGroundheight) * (2.0f * oldConfidence), 1.0f);    // :376
    if ((double)oldConfidence > 0.5 && (double)groundlevel >= (double)oldGroundheight + cfg.outlier_tolerance) return; // :379-380
    if ((double)varThresholdsq > (double)maxVar * (double)maxVar && maxVar > 0.0f &&
        (double)pointsblockSum > (double)((groundDiff * expected) * (float)S) * cfg.gpd_min_point_count_threshold) { // :382
        const float newConfidence = (float)std_min((double)pointsblockSum / cfg.occupied_cells_point_count_factor, 1.0); // :383
        const float G = (groundlevel * newConfidence + (oldConfidence * oldGroundheight) * 2.0f) / (newConfidence + oldConfidence * 2.0f); // :385
        const float Cf = (float)std_min(((double)pointsblockSum / cfg.occupied_cells_point_count_factor_x2 + (double)oldConfidence) / 2.0, 1.0); // :387
        *cell = make_float2(G, Cf);
    } else if (localmin < oldGroundheight) { // :389
        *cell = make_float2(localmin, std_min(oldConfidence + 0.1f, 0.5f)); // :391, :393
    }
}

GG_DEV void interpolate_one(const Arena &a, int slot, int x, int y)
{
    float2 *gp2 = gp2_ptr(a, slot);
    float g[9], w[9], prod[9];
    for (int s = 0; s < 9; ++s) { // :453, :458 block<3,3>(x-1, y-1)
        const float2 v = gp2[gp_idx(a, x - 1 + s % 3, y - 1 + s / 3)];
        g[s] = v.x;
        w[s] = v.y;
    }
    const float height = g[4], occupied = w[4]; // :455-456
    const float gvlSum = tree9(w) + FLT_MIN;    // :457
    for (int s = 0; s < 9; ++s) prod[s] = w[s] * g[s];
    const float avg = tree9(prod) / gvlSum; // :458
    float2 out = make_float2((1.0f - occupied) * avg + occupied * height, occupied); // :460
    const float fx = (float)x - (float)a.g.center, fy = (float)y - (float)a.g.center;
    const double d2 = ((double)fx * (double)fx + (double)fy * (double)fy) * (a.g.resolution * a.g.resolution); // :463
    if (d2 > (double)a.g.min_dist_squared) out.y = (float)std_max((double)occupied - (double)occupied / a.cfg.occupied_cells_decrease_factor, 0.001); // :464
    gp2[gp_idx(a, x, y)] = out;
}

} // namespace

__global__ void k_stage_cell(const Arena a, int slot, int stage, int i, int j)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (stage == GG_STAGE_DETECT_GROUND_PATCH_3) detect_cell<3>(a, slot, i, j);
    else if (stage == GG_STAGE_DETECT_GROUND_PATCH_5) detect_cell<5>(a, slot, i, j);
    else interpolate_one(a, slot, i, j);
}

void launch_stage_cell(const Arena &a, int slot, int stage, int i, int j, hipStream_t s) { hipLaunchKernelGGL(k_stage_cell, dim3(1), dim3(64), 0, s, a, slot, stage, i, j); }

} // namespace gg
