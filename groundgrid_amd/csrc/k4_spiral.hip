// K4 -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465).
//
// The reference sweeps the grid centre-outwards, ring by ring, IN PLACE: every visit reads the 3x3
// neighbourhood of `ground` and `groundpatch` -- inner ring and predecessor already updated, outer ring
// and successor not yet -- and rewrites its own cell.  It is a Gauss-Seidel-like serial chain of
// 130 680 visits (n = 364), not a Jacobi stencil, so an LDS-halo stencil would compute different numbers.
//
// What is exact is any schedule that preserves, per visit, WHICH neighbours are fresh.  At gg_create the
// host replays the serial visit order once and assigns each visit the earliest level that respects
// read-after-write, write-after-read and write-after-write on the 3x3 neighbourhood (gg_context.hip
// build_spiral_schedule); visits of one level are mutually independent.  The critical path is 903 levels
// for n = 364 (3 per ring along the doubly-visited corners + the last ring's edge).  One work-group per
// cloud walks the levels with a barrier between them, updating the layers in place; per-visit
// arithmetic is the reference's, verbatim.
//
// Latency-bound (chain of n_levels dependent 3x3 gathers), not bandwidth-bound: reported as such.
#include "gg_device.h"

#include <float.h>

namespace gg {

constexpr int SPIRAL_THREADS = 320;

__global__ __launch_bounds__(SPIRAL_THREADS) void k_spiral(const Arena a, const CloudParams *__restrict__ params)
{
    const int cloud = blockIdx.x;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows;
    const int center = a.g.center;
    float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
    float *ground = L + GG_LAYER_GROUND * a.layer_stride;
    float *gpatch = L + GG_LAYER_GROUNDPATCH * a.layer_stride;
    float *points = L + GG_LAYER_POINTS * a.layer_stride;
    const double res2 = a.g.resolution * a.g.resolution; // :463 pow(map.getResolution(), 2.0f)
    const double min_dist_sq = (double)a.g.min_dist_squared;
    const double decrease = a.cfg.occupied_cells_decrease_factor;

    if (threadIdx.x == 0) {
        gpatch[center + center * rows] = 1.0f;      // :405
        ground[center + center * rows] = cp.base_z; // :406-411
    }
    // :147 map["points"].setConstant(0.0) -- K3 was the last reader of the KEPT counts; K5 re-counts non-ground points
    for (int k = threadIdx.x; k < a.g.C; k += SPIRAL_THREADS) points[k] = 0.0f;
    __syncthreads();

    const uint32_t *__restrict__ level_start = a.level_start;
    const uint32_t *__restrict__ visits = a.visits;
    uint32_t s = level_start[0];
    for (int lvl = 0; lvl < a.n_levels; ++lvl) {
        const uint32_t e = level_start[lvl + 1];
        for (uint32_t v = s + threadIdx.x; v < e; v += SPIRAL_THREADS) {
            const uint32_t cell = visits[v];
            const int x = (int)(cell % (uint32_t)rows), y = (int)(cell / (uint32_t)rows);
            float w[9], g[9], pr[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) { // :453,458 block<3,3>(x-1, y-1), column-major linear index
                const int idx = (x - 1 + q % 3) + (y - 1 + q / 3) * rows;
                w[q] = gpatch[idx];
                g[q] = ground[idx];
            }
            const float height = g[4], occupied = w[4]; // :455-456
            const float gvlSum = tree9(w) + FLT_MIN;    // :457
#pragma unroll
            for (int q = 0; q < 9; ++q) pr[q] = w[q] * g[q];
            const float avg = tree9(pr) / gvlSum;       // :458
            ground[x + y * rows] = (1.0f - occupied) * avg + occupied * height; // :460
            // :463-464
            const float fx = (float)x - (float)center, fy = (float)y - (float)center;
            const double d2 = ((double)fx * (double)fx + (double)fy * (double)fy) * res2;
            if (d2 > min_dist_sq) gpatch[x + y * rows] = (float)std_max((double)occupied - (double)occupied / decrease, 0.001);
        }
        s = e;
        __syncthreads();
    }
}

void launch_spiral(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    hipLaunchKernelGGL(k_spiral, dim3(n_clouds), dim3(SPIRAL_THREADS), 0, s, a, d_params);
}

} // namespace gg
