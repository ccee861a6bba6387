// K2 -- insert_cloud, per-cell part (src/GroundSegmentation.cpp:282-309) fused with the per-call layer
// reset (:61-75) and the variance layer (:323).
//
// Tiles of 16x16 cells; a tile's records arrive in cloud order (stable tile sort).  Tiles with at most K2_LIGHT_MAX records (most
// of them) are reduced by ONE wavefront each, entirely in registers and LDS (reduce_light_tile below); the others by one
// work-group of 256 threads per tile, as follows.  The float32
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
//                 batches in flight) through the reference's recurrence in registers.  In a tile with full cells the
//                 three chains of the 64 fullest cells (which only share the point count) run in three waves;
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

#include <algorithm>
#include <cstdlib>

namespace gg {

constexpr int NBIN = 64;      // count classes of step 2: 0..31 exact, then steps of 16 up to 527, then "more"
constexpr int RCAP = 4080;    // reciprocal table: 1 / (i + 1) in binary64 for the first RCAP points of a cell (a multiple of 12 and 24)
constexpr int WB = 8;         // 64-record windows a wave keeps in flight in step 3 (and the whole light tile)
constexpr int WBC = 16;       // ... in step 1 (keys only)
constexpr int SPLIT_MIN = 24; // a tile with a cell of at least this many points runs its 64 fullest cells one chain per wave

struct __attribute__((packed, aligned(4))) zquad {
    float v[4];
};

struct RecipTable {
    double v[RCAP];
};
constexpr RecipTable make_recip_table()
{
    RecipTable t{};
    for (int i = 0; i < RCAP; ++i) t.v[i] = 1.0 / (double)(i + 1); // (IEEE binary64 division in the compiler's constant evaluator)
    return t;
}
__constant__ const RecipTable recip_table = make_recip_table();

// The measurement switches of this file (env GG_K2_DEBUG, gg_internal.h k2_debug: early returns, per-phase cycle counters, work-group traces)
// are compiled in only with -DGG_INSTRUMENT (gg_device.h GG_DEBUG_SWITCH; tools/k2_phases.py, k2_trace.py need such a library): in the
// production kernel they were a dozen scalar tests per tile, values kept alive across the whole kernel and its only scratch slot.
// measurement only (GG_K2_DEBUG=9): counters of this work-group, k2_dbg[work-group][32]; few writers per slot
GG_DEV void dbg_add(const Arena &a, int slot, unsigned long long v)
{
    const size_t wg = ((size_t)blockIdx.x + (size_t)blockIdx.y * gridDim.x) % (size_t)K2_DBG_WGS; // (more work-groups than slots: shared)
    atomicAdd(&a.k2_dbg[wg * 32 + (size_t)slot], v);
}

GG_DEV void lds_order() { __asm__ volatile("" ::: "memory"); } // LDS operations of one wave execute in program order

// The quotients of the recurrence, a / (c + 1) with c + 1 = b an integer in [1, 2^24] (:296, :302, :303).
// q = (float)((double)a * r), r = RN64(1 / b), is the correctly rounded binary32 quotient whenever |q| >= 2^-100:
//   * (double)a * r is off from a / b by a relative 2^-53 (r) + 2^-53 (product) < 2^-51.9;
//   * a / b is never closer than a relative 2^-49 to a rounding boundary M of binary32 (the midpoint of two adjacent
//     floats, a 25-bit odd integer Mi times a power of two): a - b M = A 2^ea - b Mi 2^em is a non-zero multiple of
//     2^min(ea, em) -- non-zero because the odd part of b Mi has more than 24 bits while A has at most 24 -- and
//     when the two terms are comparable ea >= em, so |a / b - M| >= 2^em / b >= 2^-49 |M|;
//   hence the binary64 product and a / b round to the same float (the argument of gg_device.h index_of).
// Below 2^-100 (results that are or may become denormal, where the boundaries are coarser and exact ties exist, and
// q == 0) the caller falls back to the IEEE division; NaN and infinities propagate through the product as through
// the division.  Bit-identity with the division: tests/test_fast_quotient_cpu.py (the same expression in C over
// random, boundary and tie operands) and the GPU parity tests.
GG_DEV float quot(float a, double r) { return (float)((double)a * r); }

// single-instruction forms where the compiler emits two or three (compare + select for std::min / std::max, a canonicalising
// v_max before every fminf operand, a separate instruction per |x|); NaN operands are ignored by all of them as by the
// expressions they replace
GG_DEV float v_max(float a, float b)
{
    float r;
    __asm__("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
GG_DEV float v_min(float a, float b)
{
    float r;
    __asm__("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
GG_DEV float min_abs(float a, float m) // min(|a|, m)
{
    float r;
    __asm__("v_min_f32 %0, |%1|, %2" : "=v"(r) : "v"(a), "v"(m));
    return r;
}
GG_DEV float min3_abs(float a, float b, float m) // min(|a|, |b|, m)
{
    float r;
    __asm__("v_min3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(m));
    return r;
}

// What a launch of the kernel maintains: all nine per-call layers; the six that the path itself reads (GG_FLAG_MINIMAL_LAYERS);
// or -- afterwards, on demand, from the records the call left behind -- the other three (gg_context.hip ensure_lazy_layers).
enum : int { K2_MINIMAL = 0, K2_FULL = 1, K2_LAZY3 = 2 };

enum : int { R_MEAN = 1 /* meanVariance, m2 */, R_GC = 2 /* groundCandidates, maxGroundHeight */, R_PDM = 4 /* planeDist */, R_MN = 8 /* minGroundHeight */ };

struct CellState {
    float gc, mean, pdm, m2, mx, mn;
};
template <int MODE> constexpr int chains_of()
{
    return MODE == K2_FULL ? (R_MEAN | R_GC | R_PDM | R_MN) : MODE == K2_MINIMAL ? (R_MEAN | R_MN) : (R_GC | R_PDM);
}

// The reference's recurrence (:295-309) of the lane's cell over its `np` heights at `zseg` (cloud order), restricted to
// the chains in R.  The chains only share the point count, and every cell starts the call at count 0 (:61-75), so point i
// of every lane has c = i: wave-uniform, and 1 / (c + 1) comes from a table through the scalar cache.
// One point of the recurrence (:295-309) with table quotients; `c` = points before it, r = 1 / (c + 1).  It is the COMMON case
// only, so that nothing but the arithmetic sits on the dependency chain of a cell (sub, cvt, mul, cvt, add for the mean):
//   * every quotient's magnitude is folded into `smallest` (except the first point's: b = 1, the product is exact);
//   * a point other than the cell's first takes `mean != 0` (:298) for granted and folds |mean| into `smallest` as well;
//   * `planeDist` is taken to be a number (:300); the caller looks at the four heights.
// The caller checks once per four points and, when a quotient came out below 2^-100, the mean was zero or a height was NaN,
// repeats those points from the saved state with one_point_exact.
template <int R, bool FIRST>
GG_DEV void one_point_fast(float z, float c, double r, float oz, CellState &s, float &smallest)
{
    const float planeDist = z - oz; // :295
    float q_gc = 1.0f, q_pdm = 1.0f, q_mean = 1.0f, delta = 0.0f, mean_base = 0.0f, mean_was = 1.0f;
    if (R & R_GC) q_gc = quot(z + c * s.gc, r); // :296
    if (R & R_MEAN) {
        if (FIRST) { // :298-299 the cell's first point finds mean == 0 (:66): mean = planeDist, delta = planeDist - planeDist
            mean_base = planeDist;
            delta = planeDist - planeDist;
            q_mean = delta; // (r = 1)
        } else {
            mean_was = s.mean;
            mean_base = s.mean;
            delta = planeDist - s.mean; // :301
            q_mean = quot(delta, r);
        }
    }
    if (R & R_PDM) q_pdm = quot(planeDist + c * s.pdm, r); // :303
    if (!FIRST) { // (one instruction per two magnitudes; the chains a wavefront does not run contribute nothing)
        if ((R & R_GC) && (R & R_MEAN)) {
            smallest = min3_abs(q_gc, mean_was, smallest);
            smallest = (R & R_PDM) ? min3_abs(q_mean, q_pdm, smallest) : min_abs(q_mean, smallest);
        } else if (R & R_MEAN) {
            smallest = min3_abs(q_mean, mean_was, smallest);
            if (R & R_PDM) smallest = min_abs(q_pdm, smallest);
        } else {
            if (R & R_GC) smallest = min_abs(q_gc, smallest);
            if (R & R_PDM) smallest = min_abs(q_pdm, smallest);
        }
    }
    if (R & R_GC) {
        s.gc = q_gc;            // :296
        s.mx = v_max(s.mx, z);  // :307 (std::max(mx, z): mx > 0 always, so no signed-zero case; a NaN z leaves mx alone in both)
    }
    if (R & R_MEAN) {
        const float mean_new = mean_base + q_mean;          // :302
        s.m2 = s.m2 + delta * (planeDist - mean_new);       // :304
        s.mean = mean_new;
    }
    if (R & R_PDM) s.pdm = q_pdm; // :303
    if (R & R_MN) s.mn = v_min(s.mn, z - 0.0001f); // :308 (std::min: a difference is never -0, a NaN leaves mn alone in both)
}

// the same point with the reference's expressions as they stand (IEEE divisions)
template <int R>
GG_DEV void one_point_exact(float z, float c, float oz, CellState &s)
{
    const float c1 = c + 1.0f;
    const float planeDist = z - oz;                                  // :295
    if (R & R_GC) s.gc = (z + c * s.gc) / c1;                        // :296 (see the note on double rounding above)
    if (R & R_MEAN) {
        if ((double)s.mean == 0.0) s.mean = planeDist;               // :298-299
    }
    if (!isnan(planeDist)) {                                         // :300
        if (R & R_MEAN) {
            const float delta = planeDist - s.mean;                  // :301
            s.mean += delta / c1;                                    // :302
            s.m2 += delta * (planeDist - s.mean);                    // :304
        }
        if (R & R_PDM) s.pdm = (planeDist + c * s.pdm) / c1;         // :303
    }
    if (R & R_GC) s.mx = std_max(s.mx, z);            // :307
    if (R & R_MN) s.mn = std_min(s.mn, z - 0.0001f);  // :308
}

// points i .. i+3 of the lane's cell (z[0..3]); i is wave-uniform, np the lane's point count
template <int R>
GG_DEV void four_points(const float (&z)[4], uint32_t i, uint32_t np, const double (&rr)[4], float oz, CellState &s)
{
    const CellState saved = s;
    const float c0 = (float)i; // (:309: (float)((double)c + 1.0) == c + 1.0f for integers below 2^24)
    float smallest = 1.0f;
    if (np >= i + 4u) { // all four points: no per-point predicate
        if (i == 0u) // (uniform)
            one_point_fast<R, true>(z[0], c0, rr[0], oz, s, smallest);
        else
            one_point_fast<R, false>(z[0], c0, rr[0], oz, s, smallest);
#pragma unroll
        for (int k = 1; k < 4; ++k) one_point_fast<R, false>(z[k], c0 + (float)k, rr[k], oz, s, smallest);
    } else if (np > i) {
        if (i == 0u)
            one_point_fast<R, true>(z[0], c0, rr[0], oz, s, smallest);
        else
            one_point_fast<R, false>(z[0], c0, rr[0], oz, s, smallest);
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (i + (uint32_t)k < np) one_point_fast<R, false>(z[k], c0 + (float)k, rr[k], oz, s, smallest);
    }
    // a NaN among the heights (:300 skips the mean and planeDist updates): the sum of the four is NaN then (and for inf - inf,
    // which only costs the detour); heights beyond np are other cells' or padding.  The maximum's chain takes the detour as well:
    // std::max(mx, z) of :307 leaves mx alone for ANY NaN, the single-instruction v_max_f32 only for a quiet one -- for a
    // SIGNALLING NaN (K1 stores the sensor's bits as they come) it returns a quieted NaN (IEEE mode); one_point_exact compares
    // and selects like the reference (test_edge_cases feeds both kinds).
    if (R & (R_GC | R_MEAN | R_PDM)) {
        const float chk = ((z[0] + z[1]) + (z[2] + z[3])) - oz;
        smallest = (chk != chk) ? 0.0f : smallest;
    }
    if ((R & (R_GC | R_MEAN | R_PDM)) && __any(smallest < 0x1p-100f)) { // (rare; uniform branch)
        __asm__ volatile("; IEEE quotients" ::: "memory"); // (keeps this a branch: if-converted, the divisions would run for every point)
        s = saved;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + (uint32_t)k < np) one_point_exact<R>(z[k], c0 + (float)k, oz, s);
    }
}

// The reference's recurrence (:295-309) of the lane's cell over its `np` heights at `zseg` (cloud order), restricted to
// the chains in R.  The chains only share the point count, and every cell starts the call at count 0 (:61-75), so point i
// of every lane has c = i: wave-uniform, and 1 / (c + 1) comes from a table through the scalar cache.
template <int R, int NB>
GG_DEV void run_cells(const float *zseg, uint32_t np, float oz, CellState &s, bool dbg_one_line = false, const float *dbg_coalesced = nullptr)
{
    // (measurement only, results void: GG_K2_DEBUG=4 every lane re-reads the first 16 bytes of its own segment -- 64 lines per load
    // instruction, all cache hits; GG_K2_DEBUG=7 the lanes read 64 CONSECUTIVE 16-byte pieces -- 8 lines per instruction)
    const zquad *zq = reinterpret_cast<const zquad *>(dbg_coalesced ? dbg_coalesced + 4 * (threadIdx.x & 63) : zseg);
    if (dbg_coalesced) dbg_one_line = true;
    uint32_t nmax = np;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) nmax = max(nmax, (uint32_t)__shfl_xor((int)nmax, d, 64));
    nmax = (uint32_t)__builtin_amdgcn_readfirstlane((int)nmax);
    // NB 16-byte batches per lane in flight, each refilled right after its four points (no register rotation: a copy of a
    // load's destination would wait for the load; unconditional loads at a clamped index: no branch around them).  The
    // heights come back from L2 (this work-group wrote them a moment ago).
    const uint32_t qlast = (np && !dbg_one_line) ? (np - 1u) >> 2 : 0u; // (an empty cell reads 16 bytes of the tile's padded region)
    zquad q[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) q[b] = zq[min((uint32_t)b, qlast)];
    const uint32_t ntab = min(nmax, (uint32_t)RCAP);
    for (uint32_t i = 0; i < ntab; i += 4u * NB) { // (RCAP is a multiple of 4 NB)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t ib = i + 4u * (uint32_t)b;
            if (b > 0 && ib >= ntab) break;
            double rr[4]; // (one 32-byte scalar load; RCAP is a multiple of 4)
#pragma unroll
            for (int k = 0; k < 4; ++k) rr[k] = recip_table.v[ib + (uint32_t)k];
            four_points<R>(q[b].v, ib, np, rr, oz, s);
            q[b] = zq[min((i >> 2) + (uint32_t)(NB + b), qlast)];
        }
    }
    for (uint32_t i = (uint32_t)RCAP; i < nmax; i += 4u) { // cells with more than RCAP points: the reference's expressions
        if (i < np) {
            const zquad cur = zq[i >> 2];
            for (int k = 0; k < 4; ++k)
                if (i + (uint32_t)k < np) one_point_exact<R>(cur.v[k], (float)(i + (uint32_t)k), oz, s);
        }
    }
}

// Shared memory of one work-group, carved by hand: the dense path uses it as one 256-cell tile, the light path as
// four independent wavefront-sized tiles.
struct DenseLds {
    unsigned long long cnt64[4][TILE_CELLS];  // 8 KiB  [wave][cell]: low 32 bits = in-map records, high 32 = KEPT records of the
                                              //        wave's quarter; later the result exchange [layer value][cell]
    union {
        unsigned long long wmask[4][TILE_CELLS]; // 8 KiB  lane mask of the window's KEPT records per cell (step 3)
        struct {                                 //        after step 3: what step 4 needs to know about every cell
            uint32_t cseg[TILE_CELLS], ctot[TILE_CELLS], craw[TILE_CELLS];
            uint16_t perm[TILE_CELLS];           //        slot of the count order -> cell
        };
    };
    uint32_t woffs[4][TILE_CELLS];            // 4 KiB  next free position of (wave, cell) in the tile's zcell region
    uint32_t bin_cnt[NBIN], bin_start[NBIN];
    uint32_t wave_tot[4], wave_full[4];
};
struct LightLds {                             // per wavefront
    unsigned long long cnt64[TILE_CELLS];     // 2 KiB  counters of step 1; then the heights grouped by cell (float[512])
    unsigned long long wmask[TILE_CELLS];     // 2 KiB
    uint32_t woffs[TILE_CELLS];               // 1 KiB
};
union ReduceLds {
    DenseLds dense;
    LightLds light[4];
};

// the 9 per-call layers of one cell (minimal layers: the three that nothing in the path reads are not maintained); B = the tile's
// block of the per-call layers (gg_internal.h percall_index: [layer position][cell]), `cell` = row in tile + 16 * column in tile.
// (Cells of a border tile beyond the map's last row / column exist in the block and are written like the others: nobody reads them.)
template <int MODE>
GG_DEV void write_cell(float *B, int cell, float c, float raw, const CellState &st)
{
    constexpr int P = TILE * TILE;
    // Streaming (non-temporal) stores: a tile's block is written once and read by OTHER kernels (k_patch, k_label, the getters) from
    // wherever it lands; written through the write-back path it competes in the L2 with the records and heights this kernel is still
    // reading.  k_reduce 1.224 -> 1.194 ms per 1024 clouds, k_patch unchanged (profiles/r05a/k2_nontemporal_ab.log).
    if (MODE != K2_LAZY3) { // (K5 has counted the non-ground points into `points` since: a later launch must leave it alone)
        __builtin_nontemporal_store(c, &B[PL_POINTS * P + cell]);
        __builtin_nontemporal_store(st.mn, &B[PL_MINGROUNDHEIGHT * P + cell]);
        __builtin_nontemporal_store(st.m2, &B[PL_M2 * P + cell]);
        __builtin_nontemporal_store(st.m2 / (c + FLT_MIN), &B[PL_VARIANCE * P + cell]); // :323
        __builtin_nontemporal_store(raw, &B[PL_POINTSRAW * P + cell]);
        __builtin_nontemporal_store(st.mean, &B[PL_MEANVARIANCE * P + cell]);
    }
    if (MODE != K2_MINIMAL) {
        __builtin_nontemporal_store(st.mx, &B[PL_MAXGROUNDHEIGHT * P + cell]);
        __builtin_nontemporal_store(st.gc, &B[PL_GROUNDCANDIDATES * P + cell]);
        __builtin_nontemporal_store(st.pdm, &B[PL_PLANEDIST * P + cell]);
    }
}

// Which of a tile's 32 half columns hold records: `held` = ballot of "my cell holds an in-map record" over a wavefront whose lanes
// 8 j .. 8 j + 7 are the cells of one half column (eight half columns per wavefront / per k).  Returns the eight bits.
GG_DEV uint32_t half_column_bits(unsigned long long held)
{
    uint32_t b = 0u;
#pragma unroll
    for (int j = 0; j < 8; ++j) b |= ((held >> (8 * j)) & 0xFFull) ? (1u << j) : 0u;
    return b;
}

// ---- light tiles: one wavefront per tile, no barrier, nothing leaves LDS but the layers --------------------------------------
// At most K2_LIGHT_MAX = 8 x 64 records: the wave holds them all in registers.  Lane l owns cells l, l + 64, l + 128, l + 192
// (4 columns x 16 rows = 256 contiguous bytes of the tile's block per layer and store instruction, as in the dense path).  A wave walks its share of the
// cloud's light list and keeps the next tile's loads in flight: its rank two tiles ahead, its record range one tile ahead,
// its records while the current tile's recurrences run.
GG_DEV void load_light_records(uint2 (&rw)[WB], const uint2 *sorted, uint32_t start, uint32_t end, int lane)
{
#pragma unroll
    for (int j = 0; j < WB; ++j) {
        const uint32_t p = start + 64u * (uint32_t)j + (uint32_t)lane;
        rw[j] = make_uint2(0u, KEY_OUTSIDE);
        if (p < end) rw[j] = sorted[p];
    }
}

template <int MODE>
GG_DEV void reduce_light_tiles(const Arena &a, const CloudParams &cp, const uint4 *tile_list, int n_light, int first, int stride, LightLds &lds)
{
    if (first >= n_light) return;
    const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    float *L = percall_ptr(a, cp.slot);
    uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
    const CellState reset = {0.0f, 0.0f, 0.0f, 0.0f, FLT_MIN, FLT_MAX};
    const float oz = cp.oz;
    const bool timing = GG_DEBUG_SWITCH(a, k2_debug) == 9;
    constexpr int RL = chains_of<MODE>();

    // the work list's entries say everything about a tile (gg_internal.h); the entry of the tile after this one is requested a
    // tile ahead (unconditionally, at a clamped index), its records while this tile's recurrences run
    uint4 ent = tile_list[first];
    uint4 ent_next = tile_list[min(first + stride, n_light - 1)];
    uint32_t start = ent.y, end = ent.z;
    uint2 rw[WB];
    load_light_records(rw, sorted, start, end, (int)(threadIdx.x & 63));
    for (int j = first; j < n_light; j += stride) {
        int lane = threadIdx.x & 63;
        __asm__ volatile("" : "+v"(lane)); // (per tile: keeps the lane's address arithmetic out of long-lived registers)
        const unsigned long long t_begin = timing ? __builtin_readcyclecounter() : 0ull;
        const bool has_next = j + stride < n_light;
        const uint32_t next_start = has_next ? ent_next.y : 0u, next_end = has_next ? ent_next.z : 0u;
        const uint4 ent_after = tile_list[min(j + 2 * stride, n_light - 1)];
        const int rank = (int)(ent.x & 0xFFFFu);
        // only the half columns (8 cells, a 32-byte sector per layer) that hold a record now are written and marked live: the
        // per-call layers are sparse, gg_internal.h tile_live -- a light tile has records in ~40 % of its half columns
        uint32_t cols_now = 0u;
        uint32_t lane_base = 0u;
        if (start != end) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lds.cnt64[lane + 64 * k] = 0ull;
                lds.wmask[lane + 64 * k] = 0ull;
            }
            // 1. count  (only the windows that hold records: a light tile has ~110 on average, two windows of the eight)
            const int n_win = (int)((end - start + 63u) >> 6);
#pragma unroll
            for (int w = 0; w < WB; ++w) {
                if (w >= n_win) break; // (uniform)
                const bool in = rw[w].y != KEY_OUTSIDE;
                const unsigned long long km = __ballot(in && ((rw[w].y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT);
                const LaneRun run = lane_run(in ? (rw[w].y & 255u) : 256u, lane);
                if (in && run.head)
                    __hip_atomic_fetch_add(&lds.cnt64[rw[w].y & 255u],
                                           (unsigned long long)__popcll(run.mask) | ((unsigned long long)__popcll(run.mask & km) << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // 2. the lane's four cells are consecutive segments of the wave's 512 heights
            uint32_t tot[4], rawc[4], t4 = 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long v = __hip_atomic_load(&lds.cnt64[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                tot[k] = (uint32_t)(v >> 32);
                rawc[k] = (uint32_t)v;
                t4 += tot[k];
            }
            uint32_t inc = t4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(inc, d, 64);
                if (lane >= d) inc += o;
            }
            lane_base = inc - t4;
            {
                // woffs[cell] = next free position (low 16 bits, at most 512) | pointsRaw of the cell << 16
                uint32_t run = lane_base;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    lds.woffs[lane + 64 * k] = run | (rawc[k] << 16);
                    run += tot[k];
                }
            }
            // 3. place (the heights take over the counters' memory: every lane has read its counters above, and the LDS
            //    operations of one wave execute in order)
            float *zs = reinterpret_cast<float *>(&lds.cnt64[0]);
            lds_order();
#pragma unroll
            for (int w = 0; w < WB; ++w) {
                if (w >= n_win) break; // (uniform)
                const uint2 r = rw[w];
                const bool kept = r.y != KEY_OUTSIDE && ((r.y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT;
                const LaneRun run = lane_run(kept ? (r.y & 255u) : 256u, lane);
                if (kept) {
                    unsigned long long *wm = &lds.wmask[r.y & 255u];
                    uint32_t *wo = &lds.woffs[r.y & 255u];
                    if (run.head) __hip_atomic_fetch_or(wm, run.mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const unsigned long long mm = __hip_atomic_load(wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    const uint32_t base = __hip_atomic_load(wo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    zs[(base & 0xFFFFu) + (uint32_t)rank_below(mm)] = __uint_as_float(r.x);
                    if ((mm >> lane) == 1ull) {
                        __hip_atomic_store(wo, base + (uint32_t)__popcll(mm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(wm, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
            lds_order();
        }
        const unsigned long long t_p1 = timing ? __builtin_readcyclecounter() : 0ull;
        if (timing && lane == 0 && start != end) dbg_add(a, 29, t_p1 - t_begin); // prologue + wait for the records + count + place
        // the next tile's records travel while this tile's recurrences run
        const bool had_points = start != end;
        load_light_records(rw, sorted, next_start, next_end, lane);
        if (had_points) {
            // 4. + 5. the four cells of the lane, one after the other; point i of every lane's cell has c = i
            const float *zs = reinterpret_cast<const float *>(&lds.cnt64[0]);
            uint32_t seg = lane_base; // (the lane's cells are consecutive segments; after step 3 woffs holds each segment's end)
#pragma unroll 1
            for (int k = 0; k < 4; ++k) {
                CellState st = reset;
                const uint32_t wv = __hip_atomic_load(&lds.woffs[lane + 64 * k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const uint32_t seg_end = wv & 0xFFFFu;
                const uint32_t np = seg_end - seg;
                for (uint32_t i = 0; __any(i < np); i += 4u) { // (uniform; at most K2_LIGHT_MAX < RCAP points)
                    double rr[4];
                    float zz[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        rr[q] = recip_table.v[i + (uint32_t)q];
                        zz[q] = zs[min(seg + i + (uint32_t)q, (uint32_t)K2_LIGHT_MAX - 1u)];
                    }
                    four_points<RL>(zz, i, np, rr, oz, st);
                }
                const int cell = lane + 64 * k;
                const uint32_t cb = half_column_bits(__ballot((wv >> 16) != 0u)); // (pointsRaw > 0: the cell holds an in-map record)
                cols_now |= cb << (8 * k);
                if ((cb >> (lane >> 3)) & 1u)
                    write_cell<MODE>(L + percall_index(rank, 0, 0), cell, (float)np, (float)(wv >> 16), st);
                seg = seg_end;
            }
            lds_order(); // (the next tile reuses the memory)
            if (timing && lane == 0) dbg_add(a, 28, __builtin_readcyclecounter() - t_p1); // chains + writes
        }
        if (MODE != K2_LAZY3 && lane == 0) tile_live[rank] = cols_now; // (K2_LAZY3: the same half columns as when the call wrote them)
        if (timing && lane == 0) {
            dbg_add(a, had_points ? 8 : 12, 1ull);
            dbg_add(a, had_points ? 9 : 14, (unsigned long long)(end - start));
            dbg_add(a, had_points ? 10 : 13, __builtin_readcyclecounter() - t_begin);
        }
        ent = ent_next;
        ent_next = ent_after;
        start = next_start;
        end = next_end;
    }
}

// ---- dense tiles: one work-group per tile -----------------------------------------------------------------------------------
template <int MODE>
GG_DEV void reduce_dense_tile(const Arena &a, const CloudParams &cp, const uint4 ent, DenseLds &lds, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
    const int rank = (int)(ent.x & 0xFFFFu);
    const uint32_t start = ent.y, end = ent.z;
    const float oz = cp.oz;
    float *ex = reinterpret_cast<float *>(&lds.cnt64[0][0]); // result exchange, [layer value][cell] (8 KiB over the counters of step 1)
    const CellState reset = {0.0f, 0.0f, 0.0f, 0.0f, FLT_MIN, FLT_MAX}; // the layer values after :61-75 (max: numeric_limits<float>::min(), sic, :73)

    const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    // the tile's region of zcell starts on its own 64-byte line (a work-group reads back only lines it wrote itself)
    float *zc = a.zcell + (size_t)cp.slot * a.zcell_stride + (size_t)((start + 15u) & ~15u) + (size_t)rank * 32u;
    const uint32_t n = end - start;
    const bool timing = GG_DEBUG_SWITCH(a, k2_debug) == 9;
    unsigned long long tmark[6] = {0, 0, 0, 0, 0, 0};
    if (timing) tmark[0] = __builtin_readcyclecounter();
    const uint32_t Q = ((n + 255u) >> 8) << 6; // records per wave: a multiple of the window
    const uint32_t qs = start + (uint32_t)wave * Q, qe = min(qs + Q, end);

#pragma unroll
    for (int k = 0; k < 4; ++k) {
        lds.cnt64[wave][lane + 64 * k] = 0ull;
        lds.wmask[wave][lane + 64 * k] = 0ull;
    }
    if (tid < NBIN) lds.bin_cnt[tid] = 0u;
    lds_order();
    // ---- 1. count ----  (WB windows of records in flight per wave: the loop is otherwise one memory latency per window)
    for (uint32_t p0 = qs; p0 < qe; p0 += 64u * WBC) {
        uint32_t key[WBC];
#pragma unroll
        for (int j = 0; j < WBC; ++j) {
            const uint32_t p = p0 + 64u * (uint32_t)j + (uint32_t)lane;
            key[j] = KEY_OUTSIDE;
            if (p < qe) key[j] = sorted[p].y;
        }
#pragma unroll
        for (int j = 0; j < WBC; ++j) {
            const bool in = key[j] != KEY_OUTSIDE;
            const unsigned long long km = __ballot(in && ((key[j] >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT);
            const LaneRun run = lane_run(in ? (key[j] & 255u) : 256u, lane);
            if (in && run.head)
                atomicAdd(&lds.cnt64[wave][key[j] & 255u],
                          (unsigned long long)__popcll(run.mask) | ((unsigned long long)__popcll(run.mask & km) << 32));
        }
    }
    __syncthreads();
    if (timing) tmark[1] = __builtin_readcyclecounter();
    if (GG_DEBUG_SWITCH(a, k2_debug) == 2) return;
    // ---- 2. thread = cell: totals, segment, the waves' shares, count class ----
    uint32_t kw[4], tot = 0u, rawc = 0u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const unsigned long long v = lds.cnt64[w][tid];
        kw[w] = (uint32_t)(v >> 32);
        tot += kw[w];
        rawc += (uint32_t)v;
    }
    const uint32_t bin = tot < 32u ? tot : min((uint32_t)NBIN - 1u, 32u + ((tot - 32u) >> 4));
    const uint32_t in_bin = atomicAdd(&lds.bin_cnt[bin], 1u);
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds.wave_tot[wave] = inc;
    const bool any_full = __any(tot >= (uint32_t)SPLIT_MIN);
    if (lane == 0) lds.wave_full[wave] = any_full ? 1u : 0u;
    __syncthreads();
    if (wave == 3) { // start of every count class in the descending order
        const uint32_t h = lds.bin_cnt[NBIN - 1 - lane];
        uint32_t hs = h;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(hs, d, 64);
            if (lane >= d) hs += o;
        }
        lds.bin_start[NBIN - 1 - lane] = hs - h;
    }
    uint32_t seg = inc - tot;
#pragma unroll
    for (int w = 0; w < 3; ++w)
        if (w < wave) seg += lds.wave_tot[w];
    {
        uint32_t run = seg;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            lds.woffs[w][tid] = run;
            run += kw[w];
        }
    }
    __syncthreads();
    if (timing) tmark[2] = __builtin_readcyclecounter();
    const uint32_t slot_of_cell = lds.bin_start[bin] + in_bin;
    // ---- 3. place: stable counting sort of the KEPT heights by cell ----
    for (uint32_t p0 = qs; p0 < qe; p0 += 64u * WB) {
        uint2 rw[WB];
#pragma unroll
        for (int j = 0; j < WB; ++j) {
            const uint32_t p = p0 + 64u * (uint32_t)j + (uint32_t)lane;
            rw[j] = make_uint2(0u, KEY_OUTSIDE);
            if (p < qe) rw[j] = sorted[p];
        }
#pragma unroll
        for (int j = 0; j < WB; ++j) { // window by window, in cloud order
            const uint2 r = rw[j];
            const bool kept = r.y != KEY_OUTSIDE && ((r.y >> KEY_CLASS_SHIFT) & 3u) == (uint32_t)GG_CLASS_KEPT;
            const uint32_t cit = r.y & 255u;
            // (relaxed work-group scope atomics on LDS: plain ds_or / ds_read / ds_write, kept in program order per
            // address; `volatile` would make the compiler drain the vector memory queue after every access)
            unsigned long long *wm = &lds.wmask[wave][cit];
            uint32_t *wo = &lds.woffs[wave][cit];
            const LaneRun run = lane_run(kept ? cit : 256u, lane);
            if (kept) {
                if (run.head) __hip_atomic_fetch_or(wm, run.mask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned long long mm = __hip_atomic_load(wm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // the window's records of this cell
                const uint32_t base = __hip_atomic_load(wo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                zc[base + (uint32_t)rank_below(mm)] = __uint_as_float(r.x);
                if ((mm >> lane) == 1ull) { // the cell's last record of the window advances the cell for the next one
                    __hip_atomic_store(wo, base + (uint32_t)__popcll(mm), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_store(wm, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads(); // (the heights written above are read by other waves of this work-group below)
    if (GG_DEBUG_SWITCH(a, k2_debug) == 3) return;
    lds.cseg[tid] = seg; // (over the lane masks, which are all zero again)
    lds.ctot[tid] = tot;
    lds.craw[tid] = rawc;
    lds.perm[slot_of_cell] = (uint16_t)tid;
    __syncthreads();
    if (timing) tmark[3] = __builtin_readcyclecounter();
    // ---- 4. recurrences, cells handed out in the count order ----
    auto put_shared = [&](int cell) { // points, pointsRaw
        ex[0 * TILE_CELLS + cell] = (float)lds.ctot[cell];
        ex[3 * TILE_CELLS + cell] = (float)lds.craw[cell];
    };
    const bool split = MODE == K2_FULL && (lds.wave_full[0] | lds.wave_full[1] | lds.wave_full[2] | lds.wave_full[3]) != 0u; // (uniform)
    if (!split) {
        const int cell = (int)lds.perm[tid];
        CellState st = reset;
        run_cells<chains_of<MODE>(), 3>(zc + lds.cseg[cell], lds.ctot[cell], oz, st, GG_DEBUG_SWITCH(a, k2_debug) == 4, GG_DEBUG_SWITCH(a, k2_debug) == 7 ? zc : nullptr);
        put_shared(cell);
        ex[1 * TILE_CELLS + cell] = st.mn;
        ex[2 * TILE_CELLS + cell] = st.m2;
        ex[4 * TILE_CELLS + cell] = st.mean;
        ex[5 * TILE_CELLS + cell] = st.mx; // (minimal layers: these three keep their reset values)
        ex[6 * TILE_CELLS + cell] = st.gc;
        ex[7 * TILE_CELLS + cell] = st.pdm;
    } else if (wave < 3) {
        // the 64 fullest cells: one chain per wave, so that the tile's longest cell costs a third of the dependent
        // instructions per point; the fourth wave runs the other 192 (much emptier) cells meanwhile
        const int cell = (int)lds.perm[lane];
        CellState st = reset;
        const float *zseg = zc + lds.cseg[cell];
        const uint32_t np = GG_DEBUG_SWITCH(a, k2_debug) == 8 ? 0u : lds.ctot[cell]; // (GG_K2_DEBUG=8, timing only: what the tile's 64 fullest cells cost)
        if (wave == 0) {
            run_cells<R_MEAN, 6>(zseg, np, oz, st, GG_DEBUG_SWITCH(a, k2_debug) == 4, GG_DEBUG_SWITCH(a, k2_debug) == 7 ? zc : nullptr);
            put_shared(cell);
            ex[2 * TILE_CELLS + cell] = st.m2;
            ex[4 * TILE_CELLS + cell] = st.mean;
        } else if (wave == 1) {
            run_cells<R_GC, 6>(zseg, np, oz, st, GG_DEBUG_SWITCH(a, k2_debug) == 4, GG_DEBUG_SWITCH(a, k2_debug) == 7 ? zc : nullptr);
            ex[5 * TILE_CELLS + cell] = st.mx;
            ex[6 * TILE_CELLS + cell] = st.gc;
        } else {
            run_cells<R_PDM | R_MN, 6>(zseg, np, oz, st, GG_DEBUG_SWITCH(a, k2_debug) == 4, GG_DEBUG_SWITCH(a, k2_debug) == 7 ? zc : nullptr);
            ex[1 * TILE_CELLS + cell] = st.mn;
            ex[7 * TILE_CELLS + cell] = st.pdm;
        }
    } else {
        for (int g = 1; g < 4; ++g) {
            const int cell = (int)lds.perm[g * 64 + lane];
            CellState st = reset;
            run_cells<R_MEAN | R_GC | R_PDM | R_MN, 3>(zc + lds.cseg[cell], lds.ctot[cell], oz, st, GG_DEBUG_SWITCH(a, k2_debug) == 4, GG_DEBUG_SWITCH(a, k2_debug) == 7 ? zc : nullptr);
            put_shared(cell);
            ex[1 * TILE_CELLS + cell] = st.mn;
            ex[2 * TILE_CELLS + cell] = st.m2;
            ex[4 * TILE_CELLS + cell] = st.mean;
            ex[5 * TILE_CELLS + cell] = st.mx;
            ex[6 * TILE_CELLS + cell] = st.gc;
            ex[7 * TILE_CELLS + cell] = st.pdm;
        }
    }
    // ---- 5. back to thread = cell, write the per-call layers ----
    if (timing && lane == 0) { // this wave's own chain time (before the barrier)
        dbg_add(a, 16 + wave, __builtin_readcyclecounter() - tmark[3]);
        if (split) dbg_add(a, 20 + wave, 1ull);
    }
    __syncthreads();
    if (timing) tmark[4] = __builtin_readcyclecounter();
    // (half columns: see reduce_light_tiles; thread = cell, a wavefront holds four columns)
    const uint32_t cb = half_column_bits(__ballot(ex[3 * TILE_CELLS + tid] != 0.0f));
    if (lane == 0) lds.wave_full[wave] = cb; // (the split flags were read before the recurrences; combined after the caller's barrier)
    const bool half_column_written = ((cb >> (lane >> 3)) & 1u) != 0u;
    CellState st;
    st.mn = ex[1 * TILE_CELLS + tid];
    st.m2 = ex[2 * TILE_CELLS + tid];
    st.mean = ex[4 * TILE_CELLS + tid];
    st.mx = ex[5 * TILE_CELLS + tid];
    st.gc = ex[6 * TILE_CELLS + tid];
    st.pdm = ex[7 * TILE_CELLS + tid];
    if (half_column_written)
        write_cell<MODE>(percall_ptr(a, cp.slot) + percall_index(rank, 0, 0), tid, ex[0 * TILE_CELLS + tid], ex[3 * TILE_CELLS + tid], st);
    if (timing && tid == 0) {
        tmark[5] = __builtin_readcyclecounter();
        dbg_add(a, 0, 1ull);                       // dense tiles
        dbg_add(a, 1, (unsigned long long)n);      // their records
        for (int k = 0; k < 5; ++k) dbg_add(a, 2 + k, tmark[k + 1] - tmark[k]); // count, scan, place, chains, write
    }
}

// One share of one cloud's work: group < n_dense_groups walks the cloud's dense list (one tile per work-group at a time), the
// other groups the light list (one tile per wavefront at a time).  k_scan wrote both lists: the tiles that hold records of this
// cloud.  More than half of the tiles of a sensor cloud receive no point at all and are visited by nobody: the per-call layers
// are stored sparsely -- tile_live[rank] says which half columns of a tile physically hold values (the ones with in-map records of
// this cloud), every other cell logically holds the per-call reset values (:61-75), and the readers substitute them
// (gg_internal.h tile_live).  Exact: gg_get_layer returns at all times what the reference's layers would hold.
template <int MODE>
GG_DEV void reduce_share(const Arena &a, const CloudParams *__restrict__ params, ReduceLds &lds, int cloud, int group, int n_groups, int n_dense_groups)
{
    const CloudParams cp = params[cloud];
    const uint4 *tile_list = a.tile_list + (size_t)cp.slot * a.tile_list_stride;
    const uint32_t *list_cnt = a.tile_list_cnt + (size_t)cp.slot * 2;
    if (group < n_dense_groups) {
        const int n_dense = a.k2_skip == 2 ? 0 : (int)list_cnt[1];
        // The dispatcher deals work-groups to an XCD's four shader engines in strict rotation and in order (tools/k2_trace.py:
        // every engine receives exactly a quarter of the work-groups).  Tile j of every cloud holds about the same number of
        // records (same sensor), so which group takes which tile changes with the cloud -- otherwise one engine gets the fullest
        // tile of every cloud.
        const int first = (group + cloud * 5) % n_dense_groups;
        for (int j = first; j < n_dense; j += n_dense_groups) {
            // (the thread index is made opaque per tile: otherwise every address derived from it is computed once, before
            // the loop, and kept in registers across the whole tile -- 30 VGPRs more)
            int tid = threadIdx.x;
            __asm__ volatile("" : "+v"(tid));
            const uint4 ent = tile_list[a.g.T - 1 - j];
            const int rank = (int)(ent.x & 0xFFFFu);
            reduce_dense_tile<MODE>(a, cp, ent, lds.dense, tid);
            __syncthreads(); // (the next tile reuses the shared memory)
            if (MODE != K2_LAZY3 && tid == 0) // the tile's half columns that hold records now (every wavefront left its eight bits)
                (a.tile_live + (size_t)cp.slot * a.tile_live_stride)[rank] =
                    lds.dense.wave_full[0] | (lds.dense.wave_full[1] << 8) | (lds.dense.wave_full[2] << 16) | (lds.dense.wave_full[3] << 24);
        }
    } else {
        const int n_light = a.k2_skip == 1 ? 0 : (int)list_cnt[0];
        const int n_waves = (n_groups - n_dense_groups) * 4;
        const int wave = threadIdx.x >> 6;
        reduce_light_tiles<MODE>(a, cp, tile_list, n_light, (group - n_dense_groups) * 4 + wave, n_waves, lds.light[wave]);
    }
}

// grid = (GD + GL, clouds): one work-group per share.
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_reduce(const Arena a, const CloudParams *__restrict__ params, int n_dense_groups)
{
    __shared__ ReduceLds lds;
    // (cloud, group) from the dispatch order, XCD-aware (gg_device.h): the tiles of one cloud are reduced on one XCD, in
    // Morton order, so vertically adjacent tiles complete each other's 128-byte layer lines in the same L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x);
    const int group = (int)(item % gridDim.x);
    if (GG_DEBUG_SWITCH(a, k2_debug) == 1) return;
    const unsigned long long t_wg = (GG_DEBUG_SWITCH(a, k2_debug) == 9 || GG_DEBUG_SWITCH(a, k2_debug) == 5) ? __builtin_readcyclecounter() : GG_DEBUG_SWITCH(a, k2_debug) == 6 ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long *census = nullptr; // (GG_K2_DEBUG=5, tools/k2_census.py: which CU runs how many work-groups at a time)
    if (GG_DEBUG_SWITCH(a, k2_debug) == 5 && threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20); // HW_ID, XCC_ID
        census = a.k2_dbg + (size_t)(((xcc & 7u) << 7) | ((hw >> 8) & 0x7Fu)) * 8;
        atomicAdd(&census[0], 1ull);
        const unsigned long long r = atomicAdd(&census[1], 1ull) + 1ull;
        atomicMax(&census[2], r);
        atomicMin(&census[4], t_wg);
    }
    reduce_share<MODE>(a, params, lds, cloud, group, (int)gridDim.x, n_dense_groups);
    if (GG_DEBUG_SWITCH(a, k2_debug) == 6 && threadIdx.x == 0 && item < 65536u) { // (tools/k2_trace.py: one record per work-group)
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        unsigned long long *rec = a.k2_dbg + (size_t)item * 4;
        rec[0] = t_wg;
        rec[1] = __builtin_amdgcn_s_memrealtime(); // (100 MHz, the same counter on every XCD)
        rec[2] = ((xcc & 7u) << 7) | ((hw >> 8) & 0x7Fu);
        rec[3] = (unsigned long long)(group < n_dense_groups);
    }
    if (census) {
        const unsigned long long t_end = __builtin_readcyclecounter();
        atomicAdd(&census[1], ~0ull); // (-1)
        atomicAdd(&census[3], t_end - t_wg);
        atomicMax(&census[5], t_end);
    }
    if (GG_DEBUG_SWITCH(a, k2_debug) == 9 && threadIdx.x == 0) {
        const int k = group < n_dense_groups ? 24 : 26;
        dbg_add(a, k, 1ull);
        dbg_add(a, k + 1, __builtin_readcyclecounter() - t_wg);
    }
}

void launch_reduce(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    // at least 64 work-groups per cloud (48 on the dense list, 16 on the light one: measured 1.62 ms per 1024 clouds against 1.72
    // with 16; finer shares balance better than fewer, longer walks), about 4096 per launch when there are few clouds
    const int min_per_cloud = a.tune_k2_per_cloud > 0 ? a.tune_k2_per_cloud : K2_MIN_GROUPS_PER_CLOUD;
    const int per_cloud = std::min(std::max(4096 / n_clouds, min_per_cloud), 2 * a.g.T);
    const int dense_share = a.tune_k2_dense_share > 0 ? a.tune_k2_dense_share : 12; // sixteenths of the groups
    const int gd = std::max(1, per_cloud * dense_share / 16), gl = std::max(1, per_cloud - gd);
    dim3 grid(gd + gl, n_clouds);
    if (a.flags & GG_FLAG_MINIMAL_LAYERS)
        hipLaunchKernelGGL(k_reduce<K2_MINIMAL>, grid, dim3(256), 0, s, a, d_params, gd);
    else
        hipLaunchKernelGGL(k_reduce<K2_FULL>, grid, dim3(256), 0, s, a, d_params, gd);
}

// The three layers nothing in the path reads (maxGroundHeight, groundCandidates, planeDist), for ONE slot whose last cloud was
// processed with GG_FLAG_MINIMAL_LAYERS: the same walk over the tile lists and the tile-sorted records that call left in the
// slot's buffers, running only those layers' recurrences (they share nothing with the others but the point count) and
// storing only those three planes.  `cp` = the parameters of that call (by value: there is no batch).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void k_reduce_lazy(const Arena a, const CloudParams cp, int n_dense_groups)
{
    __shared__ ReduceLds lds;
    reduce_share<K2_LAZY3>(a, &cp, lds, 0, (int)blockIdx.x, (int)gridDim.x, n_dense_groups);
}

// GroundSegmentation::insert_cloud as a member of its own (gg_insert_cloud; include/groundgrid/GroundSegmentation.h:55,
// src/GroundSegmentation.cpp:282-309 and the count of :234): the recurrences CONTINUE from what the layers hold -- no reset (:61-75 is
// filter_cloud's), a cell's count starts wherever an earlier range left it -- so none of k_reduce's shortcuts apply (a wave-uniform
// count, quotients from a table, the sparse half columns).  One work-group per tile that received records, one thread per cell: it
// walks the tile's records in cloud order (k_scatter's stable order) and applies the reference's expressions as they stand to its own.
// A stage entry, not the hot path: the layers of the slot are dense when this runs (launch_materialise_layers).
__global__ __launch_bounds__(256) void k_stage_insert(const Arena a, const CloudParams *__restrict__ params)
{
    const CloudParams &cp = params[0];
    const int rank = (int)blockIdx.x;
    const uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;
    const uint32_t first = tile_start[rank], end = tile_start[rank + 1];
    if (first == end) return;
    const uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    float *blk = percall_ptr(a, cp.slot) + percall_index(rank, 0, 0);
    const uint32_t cell = threadIdx.x; // row in tile + 16 * column in tile
    const uint32_t c0 = a.rank_cell0[rank];
    const bool inside = (int)(c0 & 0xFFFFu) + (int)(cell & 15u) < a.g.rows && (int)(c0 >> 16) + (int)(cell >> 4) < a.g.cols;
    CellState st;
    float count = 0.f, raw = 0.f;
    if (inside) {
        count = blk[PL_POINTS * TILE_CELLS + cell];
        raw = blk[PL_POINTSRAW * TILE_CELLS + cell];
        st.gc = blk[PL_GROUNDCANDIDATES * TILE_CELLS + cell];
        st.mean = blk[PL_MEANVARIANCE * TILE_CELLS + cell];
        st.pdm = blk[PL_PLANEDIST * TILE_CELLS + cell];
        st.m2 = blk[PL_M2 * TILE_CELLS + cell];
        st.mx = blk[PL_MAXGROUNDHEIGHT * TILE_CELLS + cell];
        st.mn = blk[PL_MINGROUNDHEIGHT * TILE_CELLS + cell];
    }
    for (uint32_t i = first; i < end; ++i) {
        const uint2 r = sorted[i]; // (every thread reads the same record: one broadcast load)
        if (!inside || (r.y & 0xFFu) != cell) continue;
        raw += 1.0f; // :234
        if (((r.y >> KEY_CLASS_SHIFT) & 3u) != (uint32_t)GG_CLASS_KEPT) continue;
        // (the expressions of :295-309 letter by letter, binary64 divisions included: a count that gg_set_layer put there need not be an integer)
        const float z = __uint_as_float(r.x);
        const float planeDist = z - cp.oz;                                                       // :295
        st.gc = (float)((double)(z + count * st.gc) / ((double)count + 1.0));                    // :296
        if ((double)st.mean == 0.0) st.mean = planeDist;                                         // :298-299
        if (!isnan(planeDist)) {                                                                 // :300
            const float delta = planeDist - st.mean;                                             // :301
            st.mean += delta / (count + 1);                                                      // :302
            st.pdm = (float)((double)(planeDist + count * st.pdm) / ((double)count + 1.0));      // :303
            st.m2 += delta * (planeDist - st.mean);                                              // :304
        }
        st.mx = std_max(st.mx, z);                                                               // :307
        st.mn = std_min(st.mn, z - 0.0001f);                                                     // :308
        count = (float)((double)count + 1.0);                                                    // :309
    }
    if (inside) {
        blk[PL_POINTS * TILE_CELLS + cell] = count;
        blk[PL_POINTSRAW * TILE_CELLS + cell] = raw;
        blk[PL_GROUNDCANDIDATES * TILE_CELLS + cell] = st.gc;
        blk[PL_MEANVARIANCE * TILE_CELLS + cell] = st.mean;
        blk[PL_PLANEDIST * TILE_CELLS + cell] = st.pdm;
        blk[PL_M2 * TILE_CELLS + cell] = st.m2;
        blk[PL_MAXGROUNDHEIGHT * TILE_CELLS + cell] = st.mx;
        blk[PL_MINGROUNDHEIGHT * TILE_CELLS + cell] = st.mn;
    }
}

void launch_stage_insert(const Arena &a, const CloudParams *d_params, hipStream_t s)
{
    hipLaunchKernelGGL(k_stage_insert, dim3(a.g.T), dim3(256), 0, s, a, d_params);
}

void launch_reduce_lazy(const Arena &a, const CloudParams &cp, hipStream_t s)
{
    const int per_cloud = std::min(4096, 2 * a.g.T);
    const int gd = std::max(1, per_cloud * 12 / 16), gl = std::max(1, per_cloud - gd);
    hipLaunchKernelGGL(k_reduce_lazy, dim3(gd + gl), dim3(256), 0, s, a, cp, gd);
}

} // namespace gg
