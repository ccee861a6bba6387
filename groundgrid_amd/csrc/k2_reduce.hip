// K2 -- insert_cloud, per-cell part (src/GroundSegmentation.cpp:282-309) fused with the per-call layer
// reset (:61-75) and the variance layer (:323).
//
// One work-group per (cloud, tile), 256 threads.  The tile's records arrive in cloud order (stable tile sort).  The float32
// recurrence of a cell (:295-309) is order dependent, so the only parallelism is ACROSS cells -- and point counts per cell
// are very uneven (most cells hold a handful of points, a few next to the sensor hundreds).  The kernel therefore separates
// "bring every cell's heights together, in cloud order" from "run the recurrences", so that the second part can hand out
// cells to lanes by point count:
//   1. count      each wave takes a contiguous quarter of the tile's records and counts, per cell, its in-map records
//                 (pointsRaw, :234) and its KEPT records (one 64-bit LDS atomic per record; order free);
//   2. scan       thread = cell: the cell's total, its segment in the tile's region of `zcell`, the start of every wave's
//                 share inside that segment, and the cell's slot in an order of DESCENDING point count (counting sort on a
//                 clipped count);
//   3. place      each wave walks its quarter again, 64 records at a time, in cloud order.  The records of one cell inside
//                 a 64-record window rank themselves through a 64-bit lane mask in LDS (ds_or, read back, mbcnt) -- no
//                 loop over the window's cells, no barrier -- and drop their z into the cell's segment: a stable counting
//                 sort by cell, so every segment lists the cell's KEPT heights in cloud order;
//   4. recur      thread = slot of the count order: wave 0 holds the tile's 64 fullest cells, wave 3 the emptiest, and
//                 the lanes of a wave have similar trip counts.  Each lane streams its segment (16-byte loads, two
//                 batches in flight) through the reference's recurrence in registers;
//   5. write      results go back to "thread = cell" through LDS and the 9 per-call layers are written exactly once,
//                 coalesced, which also performs the reset of cells that received no point (points = 0, min = FLT_MAX ...).
// All cells start from the per-call reset state (:61-75), so iteration i of every lane sees the same count c = i.
//
// Double rounding: the reference computes groundCandidates and planeDist as (float)((double)num / ((double)c + 1.0))
// with num and c + 1 exactly representable floats (:296, :303).  For binary32 operands a quotient rounded to
// binary64 (53 >= 2*24 + 2 bits) and then to binary32 equals the correctly rounded binary32 quotient (Figueroa,
// "When is double rounding innocuous?", 1995), so the IEEE float division below is bit-identical and keeps the
// 150-cycle f64 divide off the per-point dependency chain (tests/test_oracle_cpu.py::test_double_rounding_identity).
//
// Algorithmic bytes: 8 per in-map record read; 9 (full) or 5 (minimal) layers x 4 B per cell written.  (The records are read
// twice, the second time from L2; each KEPT height takes a 4-byte round trip through `zcell`, L2 resident.)
#include "gg_device.h"

#include <float.h>

namespace gg {

constexpr int NBIN = 64; // count classes of step 2: 0..31 exact, then steps of 16 up to 527, then "more"

struct __attribute__((packed, aligned(4))) zquad {
    float v[4];
};

GG_DEV void lds_order() { __asm__ volatile("" ::: "memory"); } // LDS operations of one wave execute in program order

template <bool FULL>
__global__ __launch_bounds__(256) void k_reduce(const Arena a, const CloudParams *__restrict__ params)
{
    // [wave][cell]: low 32 bits = in-map records, high 32 bits = KEPT records of the wave's quarter; later the result exchange
    __shared__ unsigned long long cnt64[4][TILE_CELLS];  // 8 KiB
    __shared__ unsigned long long wmask[4][TILE_CELLS];  // 8 KiB  lane mask of the window's KEPT records per cell (step 3)
    __shared__ uint32_t woffs[4][TILE_CELLS];            // 4 KiB  next free position of (wave, cell) in the tile's zcell region
    __shared__ uint32_t cseg[TILE_CELLS], ctot[TILE_CELLS], craw[TILE_CELLS];
    __shared__ uint16_t perm[TILE_CELLS];                // slot of the count order -> cell
    __shared__ uint32_t bin_cnt[NBIN], bin_start[NBIN];
    __shared__ uint32_t wave_tot[4];

    // (cloud, tile rank) from the dispatch order, XCD-aware (gg_device.h): the tiles of one cloud are reduced on one XCD,
    // in Morton order, so vertically adjacent tiles complete each other's 128-byte layer lines in the same L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x);
    const CloudParams cp = params[cloud];
    const int rank = (int)(item % gridDim.x);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tile = a.rank_tile[rank];
    const int tr = tile % a.g.tiles_r, tc = tile / a.g.tiles_r;

    const uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;
    const uint32_t start = tile_start[rank], end = tile_start[rank + 1];
    // More than half of the tiles of a sensor cloud receive no point at all, and a tile that received none in the previous
    // cloud either already holds the per-call reset values (:61-75: they were written when it last went empty): nothing to
    // do.  tile_live[rank] = "the tile's per-call layers may hold something else" (set by the cloud that put points there, by
    // gg_reset_map and by host writes).  Exact: every layer in HBM holds at all times what the reference's would.
    uint8_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
    if (start == end && !tile_live[rank]) return; // (uniform)
    const float oz = cp.oz;

    // per-cell running state == the layer values after :61-75
    float c = 0.0f;     // points
    float gc = 0.0f;    // groundCandidates
    float mean = 0.0f;  // meanVariance
    float pdm = 0.0f;   // planeDist
    float m2 = 0.0f;    // m2
    float mx = FLT_MIN; // maxGroundHeight  (numeric_limits<float>::min(), sic, :73)
    float mn = FLT_MAX; // minGroundHeight  (:72)
    float raw = 0.0f;   // pointsRaw
    int my_cell = tid;

    if (start != end) { // (uniform) tiles without any point only write the reset values below
        const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
        // the tile's region of zcell starts on its own 64-byte line (a work-group reads back only lines it wrote itself)
        float *zc = a.zcell + (size_t)cp.slot * a.zcell_stride + (size_t)((start + 15u) & ~15u) + (size_t)rank * 32u;
        const uint32_t n = end - start;
        const uint32_t Q = ((n + 255u) >> 8) << 6; // records per wave: a multiple of the window
        const uint32_t qs = start + (uint32_t)wave * Q, qe = min(qs + Q, end);

#pragma unroll
        for (int k = 0; k < 4; ++k) {
            cnt64[wave][lane + 64 * k] = 0ull;
            wmask[wave][lane + 64 * k] = 0ull;
        }
        if (tid < NBIN) bin_cnt[tid] = 0u;
        lds_order();
        // ---- 1. count ----
        for (uint32_t p0 = qs; p0 < qe; p0 += 64u) {
            const uint32_t p = p0 + (uint32_t)lane;
            if (p < qe) {
                const uint32_t key = sorted[p].y;
                const unsigned long long kept = ((key >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT ? 1ull : 0ull;
                atomicAdd(&cnt64[wave][key & 255u], 1ull | (kept << 32));
            }
        }
        __syncthreads();
        // ---- 2. thread = cell: totals, segment, the waves' shares, count class ----
        uint32_t kw[4], tot = 0u, rawc = 0u;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const unsigned long long v = cnt64[w][tid];
            kw[w] = (uint32_t)(v >> 32);
            tot += kw[w];
            rawc += (uint32_t)v;
        }
        const uint32_t bin = tot < 32u ? tot : min((uint32_t)NBIN - 1u, 32u + ((tot - 32u) >> 4));
        const uint32_t in_bin = atomicAdd(&bin_cnt[bin], 1u);
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        if (wave == 3) { // start of every count class in the descending order
            const uint32_t h = bin_cnt[NBIN - 1 - lane];
            uint32_t hs = h;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(hs, d, 64);
                if (lane >= d) hs += o;
            }
            bin_start[NBIN - 1 - lane] = hs - h;
        }
        uint32_t seg = inc - tot;
#pragma unroll
        for (int w = 0; w < 3; ++w)
            if (w < wave) seg += wave_tot[w];
        cseg[tid] = seg;
        ctot[tid] = tot;
        craw[tid] = rawc;
        {
            uint32_t run = seg;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                woffs[w][tid] = run;
                run += kw[w];
            }
        }
        __syncthreads();
        perm[bin_start[bin] + in_bin] = (uint16_t)tid;
        // ---- 3. place: stable counting sort of the KEPT heights by cell ----
        for (uint32_t p0 = qs; p0 < qe; p0 += 64u) {
            const uint32_t p = p0 + (uint32_t)lane;
            uint2 r = make_uint2(0u, KEY_OUTSIDE);
            if (p < qe) r = sorted[p];
            const bool kept = p < qe && ((r.y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT;
            const uint32_t cit = r.y & 255u;
            volatile unsigned long long *wm = &wmask[wave][cit];
            volatile uint32_t *wo = &woffs[wave][cit];
            if (kept) atomicOr(&wmask[wave][cit], 1ull << lane);
            lds_order();
            if (kept) {
                const unsigned long long mm = *wm; // the window's records of this cell
                const uint32_t base = *wo;
                zc[base + (uint32_t)rank_below(mm)] = __uint_as_float(r.x);
                if ((mm >> lane) == 1ull) { // the cell's last record of the window advances the cell for the next one
                    *wo = base + (uint32_t)__popcll(mm);
                    *wm = 0ull;
                }
            }
            lds_order();
        }
        __syncthreads(); // (the heights written above are read by other waves of this work-group below)
        // ---- 4. thread = slot of the count order ----
        my_cell = (int)perm[tid];
        const uint32_t np = ctot[my_cell];
        raw = (float)craw[my_cell];
        const zquad *zq = reinterpret_cast<const zquad *>(zc + cseg[my_cell]);
        zquad q0 = {{0.0f, 0.0f, 0.0f, 0.0f}}, q1 = q0;
        if (np > 0u) q0 = zq[0];
        if (np > 4u) q1 = zq[1];
        for (uint32_t i = 0; i < np; i += 4u) {
            const zquad cur = q0;
            q0 = q1;
            if (i + 8u < np) q1 = zq[(i >> 2) + 2u];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (i + (uint32_t)k < np) {
                    const float z = cur.v[k];
                    // ---- src/GroundSegmentation.cpp:295-309, one KEPT point, `c` = points before it ----
                    const float planeDist = z - oz;                                  // :295
                    if (FULL) gc = (z + c * gc) / (c + 1.0f);                        // :296 (see note on double rounding above)
                    if ((double)mean == 0.0) mean = planeDist;                       // :298-299
                    if (!isnan(planeDist)) {                                         // :300
                        const float delta = planeDist - mean;                        // :301
                        mean += delta / (c + 1.0f);                                  // :302
                        if (FULL) pdm = (planeDist + c * pdm) / (c + 1.0f);          // :303
                        m2 += delta * (planeDist - mean);                            // :304
                    }
                    if (FULL) mx = std_max(mx, z);  // :307
                    mn = std_min(mn, z - 0.0001f);  // :308
                    c = (float)((double)c + 1.0);   // :309
                }
            }
        }
    }

    // ---- 5. back to thread = cell, write the per-call layers ----
    float *ex = reinterpret_cast<float *>(&cnt64[0][0]); // (8 KiB; the counters were last read in step 2, two barriers ago)
    ex[0 * TILE_CELLS + my_cell] = c;
    ex[1 * TILE_CELLS + my_cell] = mn;
    ex[2 * TILE_CELLS + my_cell] = m2;
    ex[3 * TILE_CELLS + my_cell] = raw;
    ex[4 * TILE_CELLS + my_cell] = mean;
    ex[5 * TILE_CELLS + my_cell] = mx; // (minimal layers: these three keep their reset values)
    ex[6 * TILE_CELLS + my_cell] = gc;
    ex[7 * TILE_CELLS + my_cell] = pdm;
    __syncthreads();
    if (tid == 0) tile_live[rank] = start != end;
    const int row = tr * TILE + (tid & 15), col = tc * TILE + (tid >> 4);
    if (row < a.g.rows && col < a.g.cols) {
        const size_t idx = (size_t)row + (size_t)col * a.g.rows;
        float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
        const size_t ls = a.layer_stride;
        const float cc = ex[0 * TILE_CELLS + tid], m2c = ex[2 * TILE_CELLS + tid];
        L[GG_LAYER_POINTS * ls + idx] = cc;
        L[GG_LAYER_MINGROUNDHEIGHT * ls + idx] = ex[1 * TILE_CELLS + tid];
        L[GG_LAYER_M2 * ls + idx] = m2c;
        L[GG_LAYER_VARIANCE * ls + idx] = m2c / (cc + FLT_MIN); // :323
        L[GG_LAYER_POINTSRAW * ls + idx] = ex[3 * TILE_CELLS + tid];
        L[GG_LAYER_MEANVARIANCE * ls + idx] = ex[4 * TILE_CELLS + tid];
        L[GG_LAYER_MAXGROUNDHEIGHT * ls + idx] = ex[5 * TILE_CELLS + tid];
        L[GG_LAYER_GROUNDCANDIDATES * ls + idx] = ex[6 * TILE_CELLS + tid];
        L[GG_LAYER_PLANEDIST * ls + idx] = ex[7 * TILE_CELLS + tid];
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
