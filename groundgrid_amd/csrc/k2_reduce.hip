// K2 -- insert_cloud, per-cell part (src/GroundSegmentation.cpp:282-309) fused with the per-call layer
// reset (:61-75) and the variance layer (:323).
//
// One work-group per (cloud, tile); one thread per cell of the 16x16 tile.  The tile's records arrive
// in cloud order (stable tile sort).  They are staged through LDS in chunks of CH records: every record
// sets bit `position` in a per-cell bitmask with an LDS atomic OR (order-free), then each cell thread
// walks its own mask from the lowest bit upwards -- i.e. in cloud order -- and runs the reference's
// float32 recurrence (count, groundCandidates, running mean, planeDist, m2, min, max) in registers.
// Running state stays in registers across chunks; the 9 per-call layers are written exactly once, which
// also performs the reset of cells that received no point (points = 0, min = FLT_MAX, max = FLT_MIN...).
//
// Algorithmic bytes: 8 per in-map record read; 9 (full) or 4 (minimal) layers x 4 B per cell written.
#include "gg_device.h"

#include <float.h>

namespace gg {

constexpr int CH = 1024; // records staged per pass

template <bool FULL>
__global__ __launch_bounds__(256) void k_reduce(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ uint32_t mask[CH / 32][TILE_CELLS]; // 32 KiB: bit p of word [p/32][cell] <=> staged record p is a KEPT point of cell
    __shared__ float zs[CH];                        // 4 KiB
    __shared__ uint32_t raw_cnt[TILE_CELLS];        // pointsRaw (:234): every in-map point of the cell

    const int cloud = blockIdx.y;
    const CloudParams cp = params[cloud];
    const int rank = blockIdx.x;
    const int tid = threadIdx.x;
    const int tile = a.rank_tile[rank];
    const int tr = tile % a.g.tiles_r, tc = tile / a.g.tiles_r;

    const uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;
    const uint32_t start = tile_start[rank], end = tile_start[rank + 1];
    const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    const float oz = cp.oz;

    // per-cell running state == the layer values after :61-75
    float c = 0.0f;        // points
    float gc = 0.0f;       // groundCandidates
    float mean = 0.0f;     // meanVariance
    float pdm = 0.0f;      // planeDist
    float m2 = 0.0f;       // m2
    float mx = FLT_MIN;    // maxGroundHeight  (numeric_limits<float>::min(), sic, :73)
    float mn = FLT_MAX;    // minGroundHeight  (:72)

    raw_cnt[tid] = 0u;
#pragma unroll
    for (int w = 0; w < CH / 32; ++w) mask[w][tid] = 0u;
    __syncthreads();

    for (uint32_t base = start; base < end; base += CH) {
        const int cnt = (int)min((uint32_t)CH, end - base);
        for (int k = tid; k < cnt; k += 256) {
            const uint2 r = sorted[base + k];
            zs[k] = __uint_as_float(r.x);
            const uint32_t cit = r.y & 255u;
            atomicAdd(&raw_cnt[cit], 1u);
            if (((r.y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT) atomicOr(&mask[k >> 5][cit], 1u << (k & 31));
        }
        __syncthreads();
        const int nw = (cnt + 31) >> 5;
        for (int w = 0; w < nw; ++w) {
            uint32_t m = mask[w][tid];
            if (m) {
                mask[w][tid] = 0u;
                do {
                    const int b = __ffs((int)m) - 1;
                    m &= m - 1u;
                    const float z = zs[(w << 5) + b];
                    // ---- src/GroundSegmentation.cpp:295-309, one KEPT point, `c` = points before it ----
                    const float planeDist = z - oz; // :295
                    if (FULL) gc = (float)((double)(z + c * gc) / ((double)c + 1.0)); // :296
                    if ((double)mean == 0.0) mean = planeDist;                          // :298-299
                    if (!isnan(planeDist)) {                                            // :300
                        const float delta = planeDist - mean;                           // :301
                        mean += delta / (c + 1.0f);                                     // :302
                        if (FULL) pdm = (float)((double)(planeDist + c * pdm) / ((double)c + 1.0)); // :303
                        m2 += delta * (planeDist - mean);                               // :304
                    }
                    if (FULL) mx = std_max(mx, z);   // :307
                    mn = std_min(mn, z - 0.0001f);   // :308
                    c = (float)((double)c + 1.0);    // :309
                } while (m);
            }
        }
        __syncthreads();
    }

    const int row = tr * TILE + (tid & 15), col = tc * TILE + (tid >> 4);
    if (row < a.g.rows && col < a.g.cols) {
        const size_t idx = (size_t)row + (size_t)col * a.g.rows;
        float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
        const size_t ls = a.layer_stride;
        L[GG_LAYER_POINTS * ls + idx] = c;
        L[GG_LAYER_MINGROUNDHEIGHT * ls + idx] = mn;
        L[GG_LAYER_M2 * ls + idx] = m2;
        L[GG_LAYER_VARIANCE * ls + idx] = m2 / (c + FLT_MIN); // :323
        L[GG_LAYER_POINTSRAW * ls + idx] = (float)raw_cnt[tid];
        L[GG_LAYER_MEANVARIANCE * ls + idx] = mean;
        L[GG_LAYER_MAXGROUNDHEIGHT * ls + idx] = mx;
        L[GG_LAYER_GROUNDCANDIDATES * ls + idx] = gc;
        L[GG_LAYER_PLANEDIST * ls + idx] = pdm;
    }
}

void launch_reduce(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    dim3 grid(a.g.T, n_clouds);
    if (a.flags & GG_FLAG_MINIMAL_LAYERS)
        hipLaunchKernelGGL(k_reduce<false>, grid, dim3(256), 0, s, a, d_params);
    else
        hipLaunchKernelGGL(k_reduce<true>, grid, dim3(256), 0, s, a, d_params);
}

} // namespace gg
