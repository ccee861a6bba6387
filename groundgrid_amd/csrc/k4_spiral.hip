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

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct VisitRegs {
    uint4 lo, hi; // the 32-byte SpiralVisit
};

GG_DEV uint32_t visit_src(const VisitRegs &d, int q)
{
    // SpiralVisit layout: cell_flags u32 | wslot u16 stage u16 | src[0..8] u16 | pair[0..2] i16
    // dwords: lo.x = cell | flags << 24, lo.y = wslot | stage << 16, lo.z = src0|src1<<16, lo.w = src2|src3, hi.x = src4|src5, hi.y = src6|src7,
    //         hi.z = src8 | pair0 << 16, hi.w = pair1 | pair2 << 16
    const uint32_t w = (q < 2) ? d.lo.z : (q < 4) ? d.lo.w : (q < 6) ? d.hi.x : (q < 8) ? d.hi.y : d.hi.z;
    return (q & 1) ? (w >> 16) : (w & 0xFFFFu);
}

__global__ __launch_bounds__(1024) void k_spiral(const Arena a, const SpiralSched sc, const CloudParams *__restrict__ params)
{
    // LDS: [slots] results of visits and blocks of fetched cells, both recycled by the host-side allocator
    extern __shared__ float2 fresh[];
    const int nthreads = blockDim.x;
    // Two wave sets take turns: while one set computes level L (LDS reads, arithmetic, LDS write -- the dependent chain of
    // the sweep), the other set issues ALL of its vector-memory work (descriptor / pre-sweep prefetches for its coming
    // levels, the store of its previous result).  A wave that issues scattered 16-byte loads is held at the texture
    // addresser for ~1 clock per lane-request; with one set doing both, that issue time (~900 clocks per level) sat in
    // series with the chain (~500 clocks).  Set s owns levels s, s + 2, s + 4, ...
    const int W = nthreads >> 1;                 // lanes per set = widest level, a multiple of 64: sets are whole waves
    const int set = __builtin_amdgcn_readfirstlane((int)threadIdx.x >= W ? 1 : 0); // wave-uniform: level bounds stay in SGPRs
    const uint32_t lane = threadIdx.x - (uint32_t)(set * W); // position inside the level

    const int cloud = blockIdx.x;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows;
    const int center = a.g.center;
    float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
    float2 *gp2 = gp2_ptr(a, cp.slot);
    float *points = L + GG_LAYER_POINTS * a.layer_stride;
    const double decrease = a.cfg.occupied_cells_decrease_factor;
    // The confidence decay (:463-464) is (float)max(x - x / decrease, 0.001) in double.  The f64 divide is the longest
    // dependent stretch of a visit, so it is replaced by a multiply with 1/decrease whenever that provably rounds alike:
    // for decrease >= 1.25 the product form differs from the quotient form by < 2^-49 relative, both the max and the
    // float conversion are monotonic, so if the two ends of the +-2^-48 interval convert to the same float the exact
    // value does too; otherwise (about 1 visit in 10^7, NaN, or an unusual config) the divide decides.
    const double inv_decrease = 1.0 / decrease;
    const bool decay_fast = decrease >= 1.25 && decrease < 1e300;
    const int n_levels = sc.n_levels;

    if (threadIdx.x == 0) gp2[center + center * rows] = make_float2(cp.base_z, 1.0f); // :405, :406-411
    // :147 map["points"].setConstant(0.0) -- K3 was the last reader of the KEPT counts; K5 re-counts non-ground points
    for (int k = threadIdx.x; k < a.g.C; k += nthreads) points[k] = 0.0f;
    __syncthreads(); // full barrier: the centre cell's new values are read from the layer by ring 1


    // Every thread issues the SAME number of vector-memory operations per level, active or not: with a fixed count the
    // compiler can wait for "the loads issued N levels ago" with s_waitcnt vmcnt(N > 0) and leave the younger prefetches
    // in flight; a conditional load or store would make N unknowable and degrade every wait to vmcnt(0).
    // Level bounds are wave-uniform: scalar loads (SGPRs), requested one level before the descriptor request that
    // needs them so that no wait for them sits behind the barrier.
    // (constant address space: the table is never written while kernels run, which lets the loads be scalar)
    typedef const __attribute__((address_space(4))) uint32_t *ConstU32Ptr;
    const ConstU32Ptr LS = (ConstU32Ptr)(uintptr_t)sc.level_start;
    const uint32_t n_visits = LS[n_levels];
    uint32_t nb0 = 0, nb1 = 0; // [start, end) of the next level to request descriptors for
    auto level_bounds = [&](int lvl) {
        const int l = min(lvl, n_levels - 1);
        nb0 = LS[l];
        nb1 = LS[l + 1];
    };
    // All vector-memory traffic of the level loop goes through BUFFER instructions with hardware range checking: an idle
    // lane passes an out-of-range offset, which returns zeros / drops the store without touching L1 -- the instruction
    // count per level stays uniform (see above) but only active lanes cost bandwidth.  Measured: with plain global
    // loads the 1024 lanes of the work-group moved ~90 KB through the CU's 64 B/clk L1 path per level for ~13 KB of
    // useful data, and that, not latency or arithmetic, set the time per level.
    constexpr uint32_t OOR = 0x80000000u; // beyond any buffer here, also after the +16 / +8 immediate offsets
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<SpiralVisit *>(sc.visits), 0, (int)(n_visits * 32u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(gp2, 0, a.g.C * 8, 0x00020000);
    auto load_desc = [&](int lvl, VisitRegs &d, bool &active) { // nb0 / nb1 hold the bounds of `lvl`
        const uint32_t v = nb0 + lane;
        active = lvl < n_levels && v < nb1;
        const uint32_t off = active ? v * 16u : OOR; // halves are stored as two arrays (gg_context.hip)
        d.lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, off, 0, 0));
        d.hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_v, off, n_visits * 16u, 0));
    };
    // The memory path retires about one lane-request per clock whatever the width, so the sweep is paced by the NUMBER of
    // requests: the load plan fetches the not-yet-visited cells of the 3x3 block with three 16-byte requests (two
    // vertically adjacent interleaved cells each) instead of 18 scalar ones.
    auto load_pairs = [&](const VisitRegs &d, bool active, Pair2 (&P)[3]) {
        const int cell = (int)(d.lo.x & 0xFFFFFFu);
        const int delta[3] = {(int)d.hi.z >> 16, (int)(d.hi.w << 16) >> 16, (int)d.hi.w >> 16}; // SpiralVisit::pair, int16
#pragma unroll
        for (int p = 0; p < 3; ++p) // unused plan entries (most visits fetch one pair) and idle lanes: out of range, no traffic
            P[p] = __builtin_bit_cast(Pair2, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rsrc_g, (active && delta[p] != (int)SPIRAL_NO_PAIR) ? (uint32_t)(cell + delta[p]) * 8u : OOR, 0, 0));
    };

    // Software pipeline, in units of a set's OWN levels (every second level): descriptors are requested DESC_AHEAD own
    // levels early and pre-sweep pairs PAIR_AHEAD own levels early (pre-sweep cells are by definition not rewritten before
    // their visit, so any lead is legal).  The register sets rotate by NAME (the loop is unrolled by lcm(ND, NP)), never by
    // copying: a copy would read the destination registers of loads issued in the same iteration and force a full
    // vmcnt(0) wait per level.
    constexpr int DESC_AHEAD = 5, PAIR_AHEAD = 2, ND = DESC_AHEAD + 1, NP = PAIR_AHEAD + 1, UNROLL = 6;
    static_assert(UNROLL % ND == 0 && UNROLL % NP == 0, "register sets must rotate back after one unrolled body");
    VisitRegs D[ND];
    bool act[ND];
    Pair2 P[NP][3];
#pragma unroll
    for (int k = 0; k < DESC_AHEAD; ++k) {
        level_bounds(2 * k + set);
        load_desc(2 * k + set, D[k], act[k]);
    }
    level_bounds(2 * DESC_AHEAD + set);
#pragma unroll
    for (int k = 0; k < PAIR_AHEAD; ++k) load_pairs(D[k], act[k], P[k]);

    uint32_t dst = OOR; // byte offset of the cell the pending result goes to; out of range = no store
    float2 result = make_float2(0.0f, 0.0f);
    // barrier interval k = level k: set 0 computes in the even intervals and talks to memory in the odd ones, set 1 the
    // other way round (shifted by one barrier)
    if (set) asm volatile("s_barrier" ::: "memory");
    const int own_levels = (n_levels + 1) / 2;
    for (int base = 0; base < own_levels; base += UNROLL) {
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int lvl = 2 * (base + u) + set; // levels past n_levels are empty (load_desc): no branch, no control-flow join
            {
                // ---- memory phase (the other set computes) ----
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, result), rsrc_g, dst, 0, 0); // previous own level
                load_desc(lvl + 2 * DESC_AHEAD, D[(u + DESC_AHEAD) % ND], act[(u + DESC_AHEAD) % ND]);
                level_bounds(lvl + 2 * DESC_AHEAD + 2);
                load_pairs(D[(u + PAIR_AHEAD) % ND], act[(u + PAIR_AHEAD) % ND], P[(u + PAIR_AHEAD) % NP]);
                asm volatile("s_barrier" ::: "memory");

                // ---- compute phase: level `lvl` ----
                VisitRegs &d0 = D[u % ND];
                const Pair2 (&p0)[3] = P[u % NP];
                const bool active = act[u % ND];
                const uint32_t cell = d0.lo.x & 0xFFFFFFu;
                dst = OOR;
                if (active) {
                    const uint32_t flags = d0.lo.x >> 24, wslot = d0.lo.y & 0xFFFFu, stage_slot = d0.lo.y >> 16;
                    // park the cells of this entry's load plan in its LDS block: this visit reads them from there (every
                    // input, fresh or pre-sweep, is "an LDS address"), and so do visits of LATER levels that need the
                    // same pre-sweep cells
                    if (stage_slot != (uint32_t)SPIRAL_NONE) {
                        float2 *stage = fresh + stage_slot;
                        const int used[3] = {(int)d0.hi.z >> 16, (int)(d0.hi.w << 16) >> 16, (int)d0.hi.w >> 16};
#pragma unroll
                        for (int p = 0; p < 3; ++p)
                            if (used[p] != (int)SPIRAL_NO_PAIR) { // the block holds 2 slots per USED pair (pairs are packed from 0)
                                stage[2 * p] = make_float2(p0[p].x, p0[p].y);
                                stage[2 * p + 1] = make_float2(p0[p].z, p0[p].w);
                            }
                    }
                    if (!(flags & SPIRAL_HELPER)) { // (a helper only fetches)
                        float w[9], g[9], pr[9];
#pragma unroll
                        for (int q = 0; q < 9; ++q) { // :453,458 block<3,3>(x-1, y-1), column-major linear index
                            const float2 f = fresh[visit_src(d0, q)]; // fresh-value slot or own staging slot, resolved by the host
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
                        if (flags & SPIRAL_DECAY) { // :463-464
                            const double x = (double)occupied;
                            const double t = x - x * inv_decrease;
                            const float lo = (float)std_max(t * (1.0 - 0x1p-48), 0.001);
                            const float hi = (float)std_max(t * (1.0 + 0x1p-48), 0.001);
                            new_w = lo;
                            if (!(decay_fast && lo == hi)) new_w = (float)std_max(x - x / decrease, 0.001);
                        }
                        result = make_float2(new_g, new_w);
                        if (wslot != (uint32_t)SPIRAL_NONE) fresh[wslot] = result;
                        if (flags & SPIRAL_STORE) dst = cell * 8u;
                    }
                }
                // order LDS only: global stores are never read back inside this kernel
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, result), rsrc_g, dst, 0, 0);
}

void configure_kernels()
{
    // dynamic LDS requests above the 64 KiB default need an explicit opt-in (large grids: more fresh-value slots)
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_spiral), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

void launch_spiral(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    // Schedule 0 (widest levels, shortest dependent chain) is the default at every batch size: since the sweep hands all
    // values through LDS and fetches each cell once, the one-wavefront schedule (levels capped at 64 visits, 2.8x as many
    // levels) is slower even at 1024 clouds per launch (measured 1.10 vs 0.95 ms per 256 clouds).  It stays available
    // through GG_FLAG_SPIRAL_NARROW as an independent second exact schedule (tests).
    const int v = (a.flags & GG_FLAG_SPIRAL_NARROW) ? 1 : 0;
    const SpiralSched &sc = a.sched[v];
    int width = (sc.max_level_width + 63) / 64 * 64;
    if (width < 64) width = 64;
    const int threads = 2 * width; // two wave sets (see k_spiral)
    // gg_create guarantees max_level_width <= 1024 (one visit per thread per level)
    const size_t lds = (size_t)sc.slots * sizeof(float2);
    hipLaunchKernelGGL(k_spiral, dim3(n_clouds), dim3(threads), lds, s, a, sc, d_params);
}

} // namespace gg
