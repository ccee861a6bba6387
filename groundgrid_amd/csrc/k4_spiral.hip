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
    extern __shared__ float2 fresh[];                                                // [spiral_slots] (ground, confidence)
    uint32_t *lstart = reinterpret_cast<uint32_t *>(fresh + sc.slots);         // [n_levels + 2]

    const int cloud = blockIdx.x;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows;
    const int center = a.g.center;
    const int nthreads = blockDim.x;
    float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
    float2 *gp2 = gp2_ptr(a, cp.slot);
    float *points = L + GG_LAYER_POINTS * a.layer_stride;
    const double decrease = a.cfg.occupied_cells_decrease_factor;
    const int n_levels = sc.n_levels;

    if (threadIdx.x == 0) {
        gp2[center + center * rows] = make_float2(cp.base_z, 1.0f); // :405, :406-411
    }
    // :147 map["points"].setConstant(0.0) -- K3 was the last reader of the KEPT counts; K5 re-counts non-ground points
    for (int k = threadIdx.x; k < a.g.C; k += nthreads) points[k] = 0.0f;
    for (int k = threadIdx.x; k <= n_levels; k += nthreads) lstart[k] = sc.level_start[k];
    if (threadIdx.x == 0) lstart[n_levels + 1] = sc.level_start[n_levels];
    __syncthreads(); // full barrier: the centre cell's new values are read from the layers by ring 1

    const uint4 *__restrict__ V = reinterpret_cast<const uint4 *>(sc.visits);

    // Every thread issues the SAME number of vector-memory operations per level, active or not (idle lanes re-read
    // the level's first descriptor and store to a dummy line): with a fixed count the compiler can wait for "the
    // loads issued one level ago" with s_waitcnt vmcnt(N > 0) and leave this level's prefetches in flight; a
    // conditional load or store would make N unknowable and degrade every wait to vmcnt(0).
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
    // The lanes of a wave sit on different rings, so every load instruction touches 64 different cache lines and the
    // per-CU vector-memory pipeline (one line per clock), not arithmetic, paces a level.  The three cells of one
    // block column are contiguous (column-major layers): one 12-byte load per column and layer = 6 instead of 18.
    auto load_old = [&](const VisitRegs &d, float (&gw)[9], float (&gg_)[9]) {
        const uint32_t cell = d.lo.x;
#pragma unroll
        for (int col = 0; col < 3; ++col) { // :453,458 block<3,3>(x-1, y-1), column-major linear index
            const uint32_t idx = cell - 1u + (uint32_t)((col - 1) * rows);
            const Pair2 c01 = *reinterpret_cast<const Pair2 *>(gp2 + idx); // rows x-1, x in one 16-byte request (8-byte aligned)
            const float2 c2 = gp2[idx + 2];                                  // row  x+1
            gg_[col * 3 + 0] = c01.x;
            gw[col * 3 + 0] = c01.y;
            gg_[col * 3 + 1] = c01.z;
            gw[col * 3 + 1] = c01.w;
            gg_[col * 3 + 2] = c2.x;
            gw[col * 3 + 2] = c2.y;
        }
    };

    // Software pipeline: descriptor two levels ahead, pre-sweep values one level ahead.  The register sets rotate by
    // NAME (the loop is unrolled by 6 = lcm(3 descriptor sets, 2 value sets)), never by copying: a copy would read
    // the destination registers of loads issued in the same iteration and force a full vmcnt(0) wait per level.
    VisitRegs D[3];
    bool act[3];
    float W[2][9], G[2][9];
    load_desc(0, D[0], act[0]);
    load_desc(1, D[1], act[1]);
    load_old(D[0], W[0], G[0]);
    float *const dummy = a.spiral_dummy + 2 * threadIdx.x;

    for (int base = 0; base < n_levels; base += 6) {
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int lvl = base + u;
            if (lvl < n_levels) { // uniform
                VisitRegs &d0 = D[u % 3];
                load_desc(lvl + 2, D[(u + 2) % 3], act[(u + 2) % 3]);
                load_old(D[(u + 1) % 3], W[(u + 1) % 2], G[(u + 1) % 2]);

                const uint32_t cell = d0.lo.x;
                float2 *dst = reinterpret_cast<float2 *>(dummy);
                float new_g = 0.0f, new_w = 0.0f;
                if (act[u % 3]) {
                    float w[9], g[9], pr[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const uint32_t s = visit_src(d0, q);
                        const bool is_fresh = s != (uint32_t)SPIRAL_NONE;
                        const float2 f = fresh[is_fresh ? s : 0u]; // unconditional read + select: no branch per input
                        g[q] = is_fresh ? f.x : G[u % 2][q];
                        w[q] = is_fresh ? f.y : W[u % 2][q];
                    }
                    const float height = g[4], occupied = w[4]; // :455-456
                    const float gvlSum = tree9(w) + FLT_MIN;    // :457
#pragma unroll
                    for (int q = 0; q < 9; ++q) pr[q] = w[q] * g[q];
                    const float avg = tree9(pr) / gvlSum;                            // :458
                    new_g = (1.0f - occupied) * avg + occupied * height; // :460
                    const uint32_t flags = d0.lo.y >> 16, wslot = d0.lo.y & 0xFFFFu;
                    new_w = occupied;
                    if (flags & SPIRAL_DECAY) new_w = (float)std_max((double)occupied - (double)occupied / decrease, 0.001); // :463-464
                    if (wslot != (uint32_t)SPIRAL_NONE) fresh[wslot] = make_float2(new_g, new_w);
                    if (flags & SPIRAL_STORE) {
                        dst = gp2 + cell;
                    }
                }
                *dst = make_float2(new_g, new_w);
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
    const size_t lds = (size_t)sc.slots * sizeof(float2) + ((size_t)sc.n_levels + 2) * sizeof(uint32_t);
    hipLaunchKernelGGL(k_spiral, dim3(n_clouds), dim3(threads), lds, s, a, sc, d_params);
}

} // namespace gg
