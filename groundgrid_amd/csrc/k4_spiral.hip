// K4 -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465).
//
// The reference sweeps the grid centre-outwards, ring by ring, IN PLACE: every visit reads the 3x3
// neighbourhood of `ground` and `groundpatch` -- inner ring and predecessor already updated, outer ring
// and successor not yet -- and rewrites its own cell.  It is a Gauss-Seidel-like serial chain of
// 130 680 visits (n = 364), not a Jacobi stencil, so an LDS-halo stencil would compute different numbers.
//
// What is exact is any schedule that preserves, per visit, WHICH neighbours are fresh.  At gg_create the
// host replays the serial visit order once (gg_context.hip build_spiral_schedule) and emits, per visit, a
// 32-byte descriptor: its level (visits of one level are independent; 903 levels for n = 364: 3 per ring
// along the doubly-visited corners + the last ring's edge), and for each of the 9 cells it reads either
// the LDS slot that holds the value an earlier visit produced, or "still the pre-sweep value".
//
// One work-group per cloud walks the levels:
//   * fresh values travel through a small LDS window of (ground, confidence) pairs (slot lifetime <= 8
//     levels, 1723 slots for n = 364) -- the only data on the level-to-level critical path;
//   * pre-sweep values are read from the layers one level AHEAD into registers (their addresses depend only
//     on the descriptor, which is fetched two levels ahead), so global-memory latency is off the chain;
//   * results are also stored to the layers (last visit of a cell only), fire-and-forget: nothing in this
//     kernel reads them back, so the per-level barrier only has to order LDS (s_waitcnt lgkmcnt + s_barrier),
//     not global memory.
// Per-visit arithmetic is the reference's, verbatim (Eigen tree order, float/double promotions).
//
// Latency-bound (a chain of n_levels dependent LDS round trips), not bandwidth-bound: reported as such.
#include "gg_device.h"

#include <float.h>
#include <stdlib.h>

#include <algorithm>

namespace gg {

// two consecutive (ground, confidence) cells; only 8-byte alignment may be assumed
struct __attribute__((aligned(8))) Pair2 {
    float x, y, z, w;
};

struct VisitRegs {
    uint4 lo, hi; // the 32-byte SpiralVisit
};

GG_DEV uint32_t visit_src(const VisitRegs &d, int q)
{
    // SpiralVisit layout: cell u32 | wslot u16 flags u16 | src[0..8] u16 | pad
    // dwords: lo.x = cell, lo.y = wslot | flags << 16, lo.z = src0|src1<<16, lo.w = src2|src3, hi.x = src4|src5, hi.y = src6|src7, hi.z = src8|pad
    const uint32_t w = (q < 2) ? d.lo.z : (q < 4) ? d.lo.w : (q < 6) ? d.hi.x : (q < 8) ? d.hi.y : d.hi.z;
    return (q & 1) ? (w >> 16) : (w & 0xFFFFu);
}

__global__ __launch_bounds__(1024) void k_spiral(const Arena a, const SpiralSched sc, const CloudParams *__restrict__ params)
{
    // LDS: [slots] fresh values | [threads][6] private staging of the cells this thread loaded | [n_levels + 2] level starts
    extern __shared__ float2 fresh[];
    const int nthreads = blockDim.x;
    float2 *stage = fresh + sc.slots + (size_t)threadIdx.x * 6;
    uint32_t *lstart = reinterpret_cast<uint32_t *>(fresh + sc.slots + (size_t)nthreads * 6); // [n_levels + 2]

    const int cloud = blockIdx.x;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows;
    const int center = a.g.center;
    float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
    float2 *gp2 = gp2_ptr(a, cp.slot);
    float *points = L + GG_LAYER_POINTS * a.layer_stride;
    const double decrease = a.cfg.occupied_cells_decrease_factor;
    const int n_levels = sc.n_levels;
    const uint32_t stage_base = (uint32_t)sc.slots + threadIdx.x * 6u;

    if (threadIdx.x == 0) gp2[center + center * rows] = make_float2(cp.base_z, 1.0f); // :405, :406-411
    // :147 map["points"].setConstant(0.0) -- K3 was the last reader of the KEPT counts; K5 re-counts non-ground points
    for (int k = threadIdx.x; k < a.g.C; k += nthreads) points[k] = 0.0f;
    for (int k = threadIdx.x; k <= n_levels; k += nthreads) lstart[k] = sc.level_start[k];
    if (threadIdx.x == 0) lstart[n_levels + 1] = sc.level_start[n_levels];
    __syncthreads(); // full barrier: the centre cell's new values are read from the layer by ring 1

    const uint4 *__restrict__ V = reinterpret_cast<const uint4 *>(sc.visits);

    // Every thread issues the SAME number of vector-memory operations per level, active or not (idle lanes re-read the
    // level's first descriptor and store to a dummy line): with a fixed count the compiler can wait for "the loads
    // issued one level ago" with s_waitcnt vmcnt(N > 0) and leave this level's prefetches in flight; a conditional load
    // or store would make N unknowable and degrade every wait to vmcnt(0).
    const uint32_t n_visits = lstart[n_levels];
    auto load_desc = [&](int lvl, VisitRegs &d, bool &active) {
        const int l = min(lvl, n_levels - 1);
        const uint32_t s0 = lstart[l], e0 = lstart[l + 1];
        const uint32_t v = s0 + threadIdx.x;
        active = lvl < n_levels && v < e0;
        const uint32_t vi = active ? v : min(s0, n_visits - 1);
        d.lo = V[(size_t)vi * 2];
        d.hi = V[(size_t)vi * 2 + 1];
    };
    // The memory path retires about one lane-request per clock whatever the width, so the sweep is paced by the NUMBER of
    // requests: the load plan fetches the not-yet-visited cells of the 3x3 block with three 16-byte requests (two
    // vertically adjacent interleaved cells each) instead of 18 scalar ones.
    auto load_pairs = [&](const VisitRegs &d, Pair2 (&P)[3]) {
        const uint32_t cell = d.lo.x;
        const uint32_t plan = d.lo.y >> 20;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const uint32_t pr = (plan >> (3 * p)) & 7u; // block column * 2 + row offset
            const uint32_t idx = cell - 1u + (pr & 1u) + (uint32_t)(((int)(pr >> 1) - 1) * rows);
            P[p] = *reinterpret_cast<const Pair2 *>(gp2 + idx);
        }
    };

    // Software pipeline: descriptor two levels ahead, pre-sweep values one level ahead.  The register sets rotate by
    // NAME (the loop is unrolled by 6 = lcm(3 descriptor sets, 2 value sets)), never by copying: a copy would read
    // the destination registers of loads issued in the same iteration and force a full vmcnt(0) wait per level.
    VisitRegs D[3];
    bool act[3];
    Pair2 P[2][3];
    load_desc(0, D[0], act[0]);
    load_desc(1, D[1], act[1]);
    load_pairs(D[0], P[0]);
    float2 *const dummy = reinterpret_cast<float2 *>(a.spiral_dummy) + threadIdx.x;

    for (int base = 0; base < n_levels; base += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int lvl = base + u;
            if (lvl < n_levels) { // uniform
                VisitRegs &d0 = D[u % 3];
                load_desc(lvl + 2, D[(u + 2) % 3], act[(u + 2) % 3]);
                load_pairs(D[(u + 1) % 3], P[(u + 1) % 2]);

                const uint32_t cell = d0.lo.x;
                float2 *dst = dummy;
                float2 result = make_float2(0.0f, 0.0f);
                if (act[u % 3]) {
                    // park the six cells of this level's load plan in the thread's private LDS staging: from here on
                    // every input, fresh or pre-sweep, is "an LDS address" and needs no per-input select logic
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        stage[2 * p] = make_float2(P[u % 2][p].x, P[u % 2][p].y);
                        stage[2 * p + 1] = make_float2(P[u % 2][p].z, P[u % 2][p].w);
                    }
                    const uint32_t flags = d0.lo.y >> 16, wslot = d0.lo.y & 0xFFFFu;
                    if (flags & SPIRAL_HELPER) {
                        // a helper only forwards its pair to the slots a later visit will read
                        const uint32_t s0 = visit_src(d0, 0), s1 = visit_src(d0, 1);
                        if (s0 != (uint32_t)SPIRAL_NONE) fresh[s0] = stage[0];
                        if (s1 != (uint32_t)SPIRAL_NONE) fresh[s1] = stage[1];
                    } else {
                        float w[9], g[9], pr[9];
#pragma unroll
                        for (int q = 0; q < 9; ++q) { // :453,458 block<3,3>(x-1, y-1), column-major linear index
                            const uint32_t s = visit_src(d0, q);
                            const float2 f = fresh[s >= (uint32_t)SPIRAL_STAGED ? stage_base + (s - (uint32_t)SPIRAL_STAGED) : s];
                            g[q] = f.x;
                            w[q] = f.y;
                        }
                        const float height = g[4], occupied = w[4]; // :455-456
                        const float gvlSum = tree9(w) + FLT_MIN;    // :457
#pragma unroll
                        for (int q = 0; q < 9; ++q) pr[q] = w[q] * g[q];
                        const float avg = tree9(pr) / gvlSum;                            // :458
                        const float new_g = (1.0f - occupied) * avg + occupied * height; // :460
                        float new_w = occupied;
                        if (flags & SPIRAL_DECAY) new_w = (float)std_max((double)occupied - (double)occupied / decrease, 0.001); // :463-464
                        result = make_float2(new_g, new_w);
                        if (wslot != (uint32_t)SPIRAL_NONE) fresh[wslot] = result;
                        if (flags & SPIRAL_STORE) dst = gp2 + cell;
                    }
                }
                *dst = result;
                // order LDS only: global stores are never read back inside this kernel
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
}

void configure_kernels()
{
    // dynamic LDS requests above the 64 KiB default need an explicit opt-in (large grids: more fresh-value slots)
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_spiral), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void launch_spiral(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    // Few clouds in flight: the widest levels (shortest dependent chain, lowest latency).  Very large batches: levels of
    // one wavefront (no idle waves, no work-group barrier) move ~8 % less through the memory path; measured equal at
    // 256 clouds, ahead at 512.  GG_FLAG_SPIRAL_NARROW forces the narrow schedule (tests).
    const int v = ((a.flags & GG_FLAG_SPIRAL_NARROW) || n_clouds >= 384) ? 1 : 0;
    const SpiralSched &sc = a.sched[v];
    int threads = (sc.max_level_width + 63) / 64 * 64;
    if (threads < 64) threads = 64;
    // gg_create guarantees max_level_width <= 1024 (one visit per thread per level)
    const size_t lds = ((size_t)sc.slots + (size_t)threads * 6) * sizeof(float2) + ((size_t)sc.n_levels + 2) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_spiral, dim3(n_clouds), dim3(threads), lds, s, a, sc, d_params);
}

} // namespace gg
