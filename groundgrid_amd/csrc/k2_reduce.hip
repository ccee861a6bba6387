// K2 -- insert_cloud, per-cell part (src/GroundSegmentation.cpp:282-309) fused with the per-call layer
// reset (:61-75) and the variance layer (:323).
//
// One work-group per (cloud, tile); one thread per cell of the 16x16 tile.  The tile's records arrive in
// cloud order (stable tile sort).  They are staged through LDS in chunks of CH records and counting-sorted
// by cell INSIDE LDS, stably:
//   1. every record sets bit `position` in its cell's bitmask (LDS atomic OR: order-free),
//   2. each cell thread turns its 32 mask words into per-word exclusive popcount prefixes and its total,
//      a block scan of the 256 totals gives each cell a contiguous segment,
//   3. every KEPT record computes its rank = prefix[word] + popc(mask[word] & bits below) -- the number of
//      earlier records of the same cell, i.e. cloud order -- and drops its z into the cell's segment,
//   4. each cell thread walks its own segment front to back and runs the reference's float32 recurrence
//      (count, groundCandidates, running mean, planeDist, m2, min, max) in registers -- except for "heavy" cells
//      (more than LIGHT_MAX points in the pass), whose recurrences are delegated to the lanes of one wave so that
//      the other three waves do not idle behind one full cell each.
// Running state stays in registers across chunks (delegated cells: parked in LDS for the pass); the 9 per-call layers are written exactly once, which also
// performs the reset of cells that received no point (points = 0, min = FLT_MAX, max = FLT_MIN ...).
//
// Double rounding: the reference computes groundCandidates and planeDist as (float)((double)num / ((double)c + 1.0))
// with num and c + 1 exactly representable floats (:296, :303).  For binary32 operands a quotient rounded to
// binary64 (53 >= 2*24 + 2 bits) and then to binary32 equals the correctly rounded binary32 quotient (Figueroa,
// "When is double rounding innocuous?", 1995), so the IEEE float division below is bit-identical and keeps the
// 150-cycle f64 divide off the per-point dependency chain (tests/test_oracle_cpu.py::test_double_rounding_identity).
//
// Algorithmic bytes: 8 per in-map record read; 9 (full) or 5 (minimal) layers x 4 B per cell written.
#include "gg_device.h"

#include <float.h>

namespace gg {

constexpr int CH = 512;           // records staged per pass (27 KiB of LDS per work-group -> 5 work-groups per CU)
constexpr int NW = CH / 32;       // mask words per cell
constexpr int RPT = CH / TILE_CELLS; // records per thread per pass
constexpr int LIGHT_MAX = 2;      // a cell with more KEPT points in a pass is "heavy": its recurrence is delegated (step 4)
constexpr int HMAX = 64;          // heavy cells delegated per pass (one wavefront of runners)

template <bool FULL>
__global__ __launch_bounds__(256) void k_reduce(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ uint32_t mask[NW][TILE_CELLS];   // 16 KiB  bit p%32 of [p/32][cell] <=> staged record p is a KEPT point of cell
    __shared__ uint16_t wprefix[NW][TILE_CELLS]; // 8 KiB   segment start of the cell + its KEPT records in words < w
    __shared__ float zsorted[CH];                // 2 KiB   z, grouped by cell, cloud order inside a cell
    __shared__ uint32_t raw_cnt[TILE_CELLS];     // pointsRaw (:234): every in-map point of the cell
    __shared__ uint32_t wave_tot[4], wave_heavy[4];
    __shared__ float hst[8][HMAX];               // running state of the delegated heavy cells (step 4); row 7 = count after the pass
    __shared__ uint16_t hseg[2][HMAX];           // their segment start / length in zsorted

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
    const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    const float oz = cp.oz;

    // per-cell running state == the layer values after :61-75
    float c = 0.0f;     // points
    float gc = 0.0f;    // groundCandidates
    float mean = 0.0f;  // meanVariance
    float pdm = 0.0f;   // planeDist
    float m2 = 0.0f;    // m2
    float mx = FLT_MIN; // maxGroundHeight  (numeric_limits<float>::min(), sic, :73)
    float mn = FLT_MAX; // minGroundHeight  (:72)

    raw_cnt[tid] = 0u;
    if (start != end) { // (uniform) tiles without any point only write the reset values below
#pragma unroll
        for (int w = 0; w < NW; ++w) mask[w][tid] = 0u;
    }
    __syncthreads();

    // the records of the NEXT chunk are loaded into registers before the current chunk is processed
    uint2 nxt[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
        const uint32_t k = start + (uint32_t)(j * TILE_CELLS + tid);
        nxt[j] = make_uint2(0u, KEY_OUTSIDE);
        if (k < end) nxt[j] = sorted[k];
    }
    for (uint32_t base = start; base < end; base += CH) {
        const int cnt = (int)min((uint32_t)CH, end - base);
        // ---- 1. stage: bitmask per cell ----
        uint2 r[RPT];
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            r[j] = nxt[j];
            const uint32_t kn = base + CH + (uint32_t)(j * TILE_CELLS + tid);
            nxt[j] = make_uint2(0u, KEY_OUTSIDE);
            if (kn < end) nxt[j] = sorted[kn];
        }
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int k = j * TILE_CELLS + tid;
            if (k >= cnt) r[j].y = KEY_OUTSIDE;
            if (k < cnt) {
                const uint32_t cit = r[j].y & 255u;
                atomicAdd(&raw_cnt[cit], 1u);
                if (((r[j].y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT)
                    atomicOr(&mask[k >> 5][cit], 1u << (k & 31));
                else
                    r[j].y = KEY_OUTSIDE; // not a KEPT record: nothing to place
            }
        }
        __syncthreads();
        // ---- 2. per-cell word prefixes + block scan of the cell totals ----
        uint32_t tot = 0;
        uint32_t pc[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            pc[w] = (uint32_t)__popc(mask[w][tid]);
            tot += pc[w];
        }
        uint32_t inc = tot;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wave_tot[wave] = inc;
        const bool heavy = tot > (uint32_t)LIGHT_MAX;
        const unsigned long long hb = __ballot(heavy);
        if (lane == 0) wave_heavy[wave] = (uint32_t)__popcll(hb);
        __syncthreads();
        uint32_t wbase = 0, hpos = (uint32_t)__popcll(hb & ((1ull << lane) - 1ull)), n_heavy = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) wbase += wave_tot[w];
            if (w < wave) hpos += wave_heavy[w];
            n_heavy += wave_heavy[w];
        }
        n_heavy = min(n_heavy, (uint32_t)HMAX);
        const uint32_t my_start = wbase + inc - tot;
        const bool delegated = heavy && hpos < (uint32_t)HMAX;
        if (delegated) { // hand the cell's state to the runner (read after the next barrier)
            hst[0][hpos] = c;
            hst[1][hpos] = mean;
            hst[2][hpos] = m2;
            hst[3][hpos] = mn;
            if (FULL) {
                hst[4][hpos] = gc;
                hst[5][hpos] = pdm;
                hst[6][hpos] = mx;
            }
            hseg[0][hpos] = (uint16_t)my_start;
            hseg[1][hpos] = (uint16_t)tot;
        }
        {
            uint32_t run = my_start;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                wprefix[w][tid] = (uint16_t)run;
                run += pc[w];
            }
        }
        __syncthreads();
        // ---- 3. stable placement ----
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            if (r[j].y != KEY_OUTSIDE) {
                const int k = j * TILE_CELLS + tid;
                const uint32_t cit = r[j].y & 255u;
                const uint32_t m = mask[k >> 5][cit];
                const uint32_t rk = (uint32_t)wprefix[k >> 5][cit] + (uint32_t)__popc(m & ((1u << (k & 31)) - 1u));
                zsorted[rk] = __uint_as_float(r[j].x);
            }
        }
        __syncthreads();
        // ---- 4. ordered per-cell recurrence; clear this cell's masks for the next pass ----
#pragma unroll
        for (int w = 0; w < NW; ++w) mask[w][tid] = 0u;
        // Load balance.  A wavefront is busy for as long as its fullest cell, and point counts per cell are very uneven
        // (a few cells next to the sensor hold hundreds of points, most a handful): with a strict "thread = cell" every
        // wave of the tile waited for one of the full cells.  So a cell thread runs its own recurrence only when the
        // cell is light (<= LIGHT_MAX points in this pass); up to HMAX heavy cells per pass were compacted in step 2 and
        // are run from state parked in LDS by dedicated waves (below), so no wave idles behind one full cell.  The order INSIDE a cell is untouched.
        auto recur = [&](uint32_t s0, uint32_t n, float &c_, float &gc_, float &mean_, float &pdm_, float &m2_, float &mx_, float &mn_) {
            float znext = n ? zsorted[s0] : 0.0f;
            for (uint32_t i = 0; i < n; ++i) {
                const float z = znext;
                if (i + 1 < n) znext = zsorted[s0 + i + 1];
                // ---- src/GroundSegmentation.cpp:295-309, one KEPT point, `c_` = points before it ----
                const float planeDist = z - oz;                                       // :295
                if (FULL) gc_ = (z + c_ * gc_) / (c_ + 1.0f);                         // :296 (see note on double rounding above)
                if ((double)mean_ == 0.0) mean_ = planeDist;                          // :298-299
                if (!isnan(planeDist)) {                                              // :300
                    const float delta = planeDist - mean_;                            // :301
                    mean_ += delta / (c_ + 1.0f);                                     // :302
                    if (FULL) pdm_ = (planeDist + c_ * pdm_) / (c_ + 1.0f);           // :303
                    m2_ += delta * (planeDist - mean_);                               // :304
                }
                if (FULL) mx_ = std_max(mx_, z);  // :307
                mn_ = std_min(mn_, z - 0.0001f);  // :308
                c_ = (float)((double)c_ + 1.0);   // :309
            }
        };
        if (!delegated) recur(my_start, tot, c, gc, mean, pdm, m2, mx, mn);
        // Delegated cells: the recurrence of one point consists of three chains that only share the point count --
        // (mean, m2, min), (groundCandidates, max), (planeDist) -- so three waves each run ONE chain for all heavy cells
        // (lane = heavy cell): a third of the dependent instructions per point on the tile's critical path.  The
        // roles rotate with the tile so that concurrent work-groups of a CU keep different SIMDs busy.
        const int role = (wave - rank) & 3; // wave-uniform
        if ((uint32_t)lane < n_heavy && role < (FULL ? 3 : 1)) {
            const uint32_t s0 = (uint32_t)hseg[0][lane], n = (uint32_t)hseg[1][lane];
            float hc = hst[0][lane]; // points before this pass
            float znext = zsorted[s0];
            // (`hc` is an integer-valued float < 2^24: hc + 1.0f is the reference's (float)((double)hc + 1.0), :309)
            if (role == 0) {
                float hmean = hst[1][lane], hm2 = hst[2][lane], hmn0 = hst[3][lane]; // (min: here only without the other chains)
                for (uint32_t i = 0; i < n; ++i) {
                    const float z = znext;
                    if (i + 1 < n) znext = zsorted[s0 + i + 1];
                    const float planeDist = z - oz;                   // :295
                    if (hmean == 0.0f) hmean = planeDist;             // :298-299
                    if (!isnan(planeDist)) {                          // :300
                        const float delta = planeDist - hmean;        // :301
                        hmean += delta / (hc + 1.0f);                 // :302
                        hm2 += delta * (planeDist - hmean);           // :304
                    }
                    if (!FULL) hmn0 = std_min(hmn0, z - 0.0001f);     // :308
                    hc += 1.0f;                                       // :309
                }
                if (!FULL) hst[3][lane] = hmn0;
                hst[1][lane] = hmean;
                hst[2][lane] = hm2;
                hst[7][lane] = hc; // points after this pass (row 0 is still being read by the other two chains)
            } else if (role == 1) {
                float hgc = hst[4][lane], hmx = hst[6][lane];
                for (uint32_t i = 0; i < n; ++i) {
                    const float z = znext;
                    if (i + 1 < n) znext = zsorted[s0 + i + 1];
                    hgc = (z + hc * hgc) / (hc + 1.0f);               // :296
                    hmx = std_max(hmx, z);                            // :307
                    hc += 1.0f;
                }
                hst[4][lane] = hgc;
                hst[6][lane] = hmx;
            } else {
                float hpdm = hst[5][lane], hmn = hst[3][lane];
                for (uint32_t i = 0; i < n; ++i) {
                    const float z = znext;
                    if (i + 1 < n) znext = zsorted[s0 + i + 1];
                    const float planeDist = z - oz;
                    if (!isnan(planeDist)) hpdm = (planeDist + hc * hpdm) / (hc + 1.0f); // :300, :303
                    hmn = std_min(hmn, z - 0.0001f);                  // :308
                    hc += 1.0f;
                }
                hst[5][lane] = hpdm;
                hst[3][lane] = hmn;
            }
        }
        __syncthreads();
        if (delegated) { // take the state back
            c = hst[7][hpos];
            mean = hst[1][hpos];
            m2 = hst[2][hpos];
            mn = hst[3][hpos];
            if (FULL) {
                gc = hst[4][hpos];
                pdm = hst[5][hpos];
                mx = hst[6][hpos];
            }
        }
    }

    if (tid == 0) tile_live[rank] = start != end;
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
