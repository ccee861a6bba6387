// K4 -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465).
//
// The reference sweeps the grid centre-outwards, ring by ring, IN PLACE: every visit reads the 3x3
// neighbourhood of `ground` and `groundpatch` -- inner ring and predecessor already updated, outer ring
// and successor not yet -- and rewrites its own cell.  It is a Gauss-Seidel-like serial chain of
// 130 680 visits (n = 364), not a Jacobi stencil, so an LDS-halo stencil would compute different numbers.
//
// What is exact is any schedule that preserves, per visit, WHICH neighbours are fresh.  At gg_create the host replays the
// serial visit order once (gg_context.hip build_spiral_schedule) and emits, per visit, a 32-byte descriptor: its level
// (visits of one level are independent; 904 levels for n = 364), and for each of the 9 cells it reads the LDS slot that
// will hold the value -- the result of an earlier visit, or the pre-sweep value some entry of an earlier level (or the
// visit itself) fetched from the layer.  LDS slots are allocated on the host with exact lifetimes; the device never
// decides anything from timing.  tests/test_spiral_schedule_cpu.py executes such schedules on the host against the
// oracle's serial sweep.
//
// One work-group per cloud walks the levels, one barrier interval per level (see k_spiral below for the roles of its
// wavefronts).  Per-visit arithmetic is the reference's, verbatim (Eigen tree order, float/double promotions).  Results are
// stored to the layer fire-and-forget (last visit of a cell only): nothing in this kernel reads them back, so the barriers
// only have to order LDS (s_waitcnt lgkmcnt + s_barrier), not global memory.
//
// Bound by the dependent chain of 904 levels (LDS read -> ~100 VALU -> LDS write -> barrier), not by bandwidth: reported as
// such.
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

// Work-group = two "compute" wave sets of W lanes + one loader wave:
//   * the LOADER streams the visit descriptors (the largest memory item of the sweep: 32 B per visit, 4 MB per cloud) into
//     an LDS ring, one level per barrier interval, LEAD levels ahead, with LDS-direct buffer loads (no registers, nothing
//     the compute waves wait for).  Alone on a CU the sweep used to be paced by this stream: per-lane descriptor loads
//     kept only ~40 KB in flight against ~2 us of Infinity-Cache latency;
//   * the two compute sets take turns (set s owns levels s, s + 2, ...): while one runs the dependent chain of level L
//     (LDS reads, ~100 VALU, LDS write), the other does its memory phase -- store of its previous result, the pre-sweep
//     pair loads of its next level but one, and reading its next descriptor from the ring into registers.
// Barrier #j closes interval j - 1; level L is computed in interval L + 1, its memory phase runs in interval L.
template <int NPI, int LEAD>
__global__ __launch_bounds__(1024) void k_spiral(const Arena a, const SpiralSched sc, const CloudParams *__restrict__ params)
{
    constexpr int R = LEAD + 1; // ring depth: level x is overwritten by level x + R, requested in interval x + 1 at the earliest
    // LDS: [slots] results of visits and blocks of fetched cells (host-allocated) | [R][Wd] descriptors of the coming levels
    extern __shared__ float2 fresh[];
    const int nthreads = blockDim.x;
    const int W = (nthreads - 64) >> 1;          // lanes per compute set = widest level rounded to 64: sets are whole waves
    const int Wd = NPI * 32;                     // descriptor capacity of one ring level (NPI loader instructions x 64 lanes x 16 B)
    uint4 *const ring = reinterpret_cast<uint4 *>(fresh + sc.slots);
    const int role = __builtin_amdgcn_readfirstlane((int)threadIdx.x >= 2 * W ? 2 : (int)threadIdx.x >= W ? 1 : 0); // wave-uniform

    const int cloud = blockIdx.x;
    const CloudParams &cp = params[cloud];
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

    // level bounds are wave-uniform (constant address space: scalar loads)
    typedef const __attribute__((address_space(4))) uint32_t *ConstU32Ptr;
    const ConstU32Ptr LS = (ConstU32Ptr)(uintptr_t)sc.level_start;
    const uint32_t n_visits = LS[n_levels];
    constexpr uint32_t OOR = 0x80000000u; // buffer offset beyond any buffer here: reads return 0, writes are dropped, no traffic
    const __amdgpu_buffer_rsrc_t rsrc_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<SpiralVisit *>(sc.visits), 0, (int)(n_visits * 32u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(gp2, 0, a.g.C * 8, 0x00020000);
    const int own_levels = (n_levels + 1) / 2; // iterations of a compute set; barriers: 2 * own_levels (+ 1 for set 1 and the loader)

    if (role == 2) {
        // ---------------------------------------------------------------- loader wave
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t nb0 = LS[0], nb1 = LS[1]; // bounds of the next level to request (scalar loads issued one request early)
        auto request = [&](int lvl) { // descriptors of level `lvl` -> ring[lvl % R], 16-byte pieces, lane-consecutive in LDS
            const uint32_t s0 = nb0, e0 = nb1;
            {
                const int ln = min(lvl + 1, n_levels - 1);
                nb0 = LS[ln];
                nb1 = LS[ln + 1];
            }
            const uint32_t pieces = lvl < n_levels ? 2u * (e0 - s0) : 0u;
            uint4 *base = ring + (size_t)(lvl % R) * (size_t)Wd * 2;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const uint32_t piece = (uint32_t)i * 64u + lane;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_v, (__attribute__((address_space(3))) void *)(base + i * 64), 16,
                                                         piece < pieces ? s0 * 32u + piece * 16u : OOR, 0, 0, 0);
            }
        };
        for (int lvl = 0; lvl < LEAD; ++lvl) request(lvl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // levels 0 .. LEAD - 1 are in the ring
        for (int j = 1; j <= 2 * own_levels + 1; ++j) { // interval j - 1, closed by barrier #j
            request(j - 1 + LEAD);
            // before barrier #j the descriptors of level j + 2 must have landed (the memory phase of interval j reads them):
            // requested LEAD - 3 intervals ago, i.e. all but the youngest (LEAD - 3) * NPI loads are complete
            static_assert((LEAD - 3) * NPI <= 63 && LEAD >= 4, "vmcnt range");
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((LEAD - 3) * NPI) : "memory");
        }
        return;
    }

    // -------------------------------------------------------------------- compute sets
    const int set = role;
    const uint32_t lane = threadIdx.x - (uint32_t)(set * W); // position inside the level
    // Pre-sweep pairs: three range-checked 16-byte buffer loads per entry (unused plan entries and idle lanes pass an
    // out-of-range offset: no traffic, but every lane issues the same number of memory instructions, which lets the
    // compiler wait with vmcnt(N > 0) for "the loads of two phases ago" while younger ones stay in flight).
    auto width_of = [&](int lvl) -> uint32_t {
        const int l = min(lvl, n_levels - 1);
        return lvl < n_levels ? LS[l + 1] - LS[l] : 0u;
    };
    auto load_pairs = [&](int lvl, uint32_t width, Pair2 (&P)[3]) { // reads cell and load plan of this lane's entry of `lvl` from the ring
        const bool active = lane < width;
        const uint4 *e = ring + ((size_t)(lvl % R) * (size_t)Wd + lane) * 2;
        const int cell = (int)(reinterpret_cast<const uint32_t *>(e)[0] & 0xFFFFFFu);
        const uint2 pw = reinterpret_cast<const uint2 *>(e)[3]; // dwords 6, 7: src8 | pair0 << 16, pair1 | pair2 << 16
        const int delta[3] = {(int)pw.x >> 16, (int)(pw.y << 16) >> 16, (int)pw.y >> 16}; // SpiralVisit::pair, int16
#pragma unroll
        for (int p = 0; p < 3; ++p)
            P[p] = __builtin_bit_cast(Pair2, __builtin_amdgcn_raw_buffer_load_b128(
                                                 rsrc_g, (active && delta[p] != (int)SPIRAL_NO_PAIR) ? (uint32_t)(cell + delta[p]) * 8u : OOR, 0, 0));
    };

    __syncthreads(); // the loader's first LEAD levels are in the ring; the centre cell's new value is in the layer
    // A wavefront whose lanes lie beyond the level width has nothing to do, yet every instruction it issues takes an issue
    // slot of the SIMD it shares with a busy wavefront (11 waves on 4 SIMDs).  Level widths grow with the ring radius, so
    // wavefront k of a set only joins the level loop shortly before the first level wider than 64 * k (until then it just
    // keeps the barrier count) and retires after the last such level.
    const int wk = __builtin_amdgcn_readfirstlane((int)(lane >> 6));
    const int first_needed = sc.first_wide[wk] - 2, last_needed = sc.last_wide[wk]; // (pairs are requested one own level early)
    int base0 = 0;
    if (set) asm volatile("s_barrier" ::: "memory"); // set 1 runs one interval behind set 0
    while (2 * (base0 + 2) + set <= first_needed && base0 + 2 < own_levels) { // two own levels = four barriers, nothing else
        asm volatile("s_barrier\n\ts_barrier\n\ts_barrier\n\ts_barrier" ::: "memory");
        base0 += 2;
    }
    Pair2 P[2][3];
    // widths of the own levels lvl, lvl + 2 and (requested one memory phase early, so that no scalar-load round trip sits in
    // a phase) lvl + 4
    const int lvl0 = 2 * base0 + set;
    uint32_t w_cur = width_of(lvl0), w_p2 = width_of(lvl0 + 2), w_p4 = width_of(lvl0 + 4);
    load_pairs(lvl0, w_cur, P[0]);
    uint32_t dst = OOR; // byte offset of the cell the pending result goes to; out of range = no store
    float2 result = make_float2(0.0f, 0.0f);
    for (int base = base0; base < own_levels && 2 * base + set <= last_needed; base += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int lvl = 2 * (base + u) + set; // own level; levels past n_levels are empty
            if (base + u < own_levels) {          // (uniform; own_levels may be odd)
                // ---- memory phase, interval lvl (the other set computes level lvl - 1) ----
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, result), rsrc_g, dst, 0, 0); // previous own level
                load_pairs(lvl + 2, w_p2, P[(u + 1) & 1]);
                // ... and everything of the coming visit that does not depend on other visits: decode the descriptor (read
                // from the ring) into LDS addresses, and park the cells of the visit's own load plan (requested two memory
                // phases ago) in its LDS block -- the host frees a block one level late, so writing it one interval before the
                // visit's level cannot hit a slot a visit of the previous level is still reading.  What is left for the
                // compute phase is the dependent chain proper: nine LDS reads, the arithmetic, one LDS write.
                VisitRegs d0;
                {
                    const uint4 *e = ring + ((size_t)(lvl % R) * (size_t)Wd + lane) * 2;
                    d0.lo = e[0];
                    d0.hi = e[1];
                }
                const bool active = lane < w_cur;
                w_cur = w_p2;
                w_p2 = w_p4;
                w_p4 = width_of(lvl + 6);
                const uint32_t flags = d0.lo.x >> 24, cell = d0.lo.x & 0xFFFFFFu;
                const uint32_t wslot = d0.lo.y & 0xFFFFu, stage_slot = d0.lo.y >> 16;
                const bool visit = active && !(flags & SPIRAL_HELPER); // (a helper only fetches)
                const Pair2 (&p0)[3] = P[u & 1];
                if (active && stage_slot != (uint32_t)SPIRAL_NONE) {
                    float2 *stage = fresh + stage_slot;
                    const int used[3] = {(int)d0.hi.z >> 16, (int)(d0.hi.w << 16) >> 16, (int)d0.hi.w >> 16};
#pragma unroll
                    for (int p = 0; p < 3; ++p)
                        if (used[p] != (int)SPIRAL_NO_PAIR) { // the block holds 2 slots per USED pair (pairs are packed from 0)
                            stage[2 * p] = make_float2(p0[p].x, p0[p].y);
                            stage[2 * p + 1] = make_float2(p0[p].z, p0[p].w);
                        }
                }
                uint32_t src[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) src[q] = visit_src(d0, q);
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

                // ---- compute phase: level lvl, interval lvl + 1 ----
                dst = OOR;
                if (visit) {
                    float w[9], g[9], pr[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) { // :453,458 block<3,3>(x-1, y-1), column-major linear index
                        const float2 f = fresh[src[q]]; // a visit's result slot or an element of a loader's block, resolved by the host
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
                // order LDS only: global stores are never read back inside this kernel
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
        }
    }
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, result), rsrc_g, dst, 0, 0);
}

// template instances: NPI = loader instructions per level (ring level capacity NPI * 32 descriptors), LEAD = request lead
#define GG_SPIRAL_INSTANCES(X) X(2, 8) X(4, 8) X(6, 8) X(8, 8) X(10, 8) X(12, 8) X(14, 6)

int spiral_npi(int max_level_width)
{
    const int need = (std::max(max_level_width, 1) + 31) / 32;
#define X(NPI, LEAD) if (need <= NPI) return NPI;
    GG_SPIRAL_INSTANCES(X)
#undef X
    return -1;
}
static int spiral_lead(int npi) { return npi <= 12 ? 8 : 6; }
size_t spiral_lds_bytes(int slots, int max_level_width)
{
    const int npi = spiral_npi(max_level_width);
    if (npi < 0) return (size_t)1 << 30;
    return (size_t)slots * sizeof(float2) + (size_t)(spiral_lead(npi) + 1) * (size_t)npi * 32 * sizeof(SpiralVisit);
}

void configure_kernels()
{
    // dynamic LDS requests above the 64 KiB default need an explicit opt-in
#define X(NPI, LEAD) hipFuncSetAttribute(reinterpret_cast<const void *>(k_spiral<NPI, LEAD>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    GG_SPIRAL_INSTANCES(X)
#undef X
}

void launch_spiral(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    // Schedule 0 (widest levels, shortest dependent chain) is the default at every batch size: since the sweep hands all
    // values through LDS and fetches each cell once, the one-wavefront schedule (levels capped at 64 visits, 2.8x as many
    // levels) is slower even at 1024 clouds per launch.  It stays available through GG_FLAG_SPIRAL_NARROW as an
    // independent second exact schedule (tests).
    const int v = (a.flags & GG_FLAG_SPIRAL_NARROW) ? 1 : 0;
    const SpiralSched &sc = a.sched[v];
    int width = (sc.max_level_width + 63) / 64 * 64;
    if (width < 64) width = 64;
    const int threads = 2 * width + 64; // two compute sets + the loader wave (see k_spiral)
    const int npi = spiral_npi(sc.max_level_width);
    const size_t lds = spiral_lds_bytes(sc.slots, sc.max_level_width);
#define X(NPI, LEAD)                                                                                         \
    if (npi == NPI) {                                                                                        \
        hipLaunchKernelGGL((k_spiral<NPI, LEAD>), dim3(n_clouds), dim3(threads), lds, s, a, sc, d_params);  \
        return;                                                                                              \
    }
    GG_SPIRAL_INSTANCES(X)
#undef X
}

} // namespace gg
