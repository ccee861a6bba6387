// sweep_core.h -- the terrain sweep (spiral_ground_interpolation / interpolate_cell, src/GroundSegmentation.cpp:398-465)
// as a ring-per-lane dataflow.  Written once, in per-lane scalar form, and compiled twice: for gfx950 (k4_sweep.hip, one
// lane = one thread of a wavefront) and for the host (sweep_emul.hip: a lock-step emulation of the same wavefronts with
// randomised wave scheduling, compared with the oracle's serial sweep by tests/test_sweep_emul_cpu.py -- no GPU needed).
//
// The serial sweep.  c = n/2 - 1.  For ring r = 1 .. c-1 (rp = c - r, R = c + r) the reference visits, in this order,
//     A_k = (rp, rp+k)  k = 0..2r-1      B_k = (rp+k, rp)  k = 0..2r-1      (B_0 revisits the corner (rp, rp))
//     C_k = (R, R-k)    k = 0..2r        D_k = (R-k, R)    k = 0..2r        (D_0 revisits the corner (R, R))
// and every visit rewrites its cell from the 3x3 block around it, in place: neighbours visited earlier are read NEW, the
// others OLD (= their value before the sweep: every cell is rewritten by its own visit only).
//
// Dependences (Appendix F of SURVEY.md, re-derived here).  Along a side every visit needs its predecessor (a first-order
// chain); from the ring inside it needs three NEW cells; the two doubly visited corners chain three visits per ring
// (A_0 -> A_1 -> B_0, C_0 -> C_1 -> D_0) and read nothing new from anybody else.  What limits a schedule is where two sides
// meet: the last visit of B's chain of ring r is the JOIN of C's chain of ring r (needed by C's last-but-one visit), and C's
// last visit is the join of B's chain of ring r + 1 -- likewise A and D.  So the ends of the sides advance by 4 visits per
// ring (two of B, two of C, alternating), ~4 (c - 1) visits in all; everything else keeps up with that when ring r + 1 follows
// ring r by SKEW = 1 step (a chain of ring r + 1 is 2 visits longer: SKEW + 2 = 3 steps of work per ring and wavefront):
//   * two CORNER wavefronts walk the corner chains ring by ring (3 visits per ring) and publish A_1, B_0 / C_1, D_0;
//   * every (side, ring) is a CHAIN owned by one lane: A_2..A_{2r-1}, B_1..B_{2r-1}, C_2..C_{2r}, D_1..D_{2r};
//     lanes of one wavefront = 64 consecutive rings of one side, ring r+1 running SKEW steps behind ring r.
// With k0 = first chain index (2, 1, 2, 1) and len = chain length (2r-2, 2r-1, 2r-1, 2r), step s of a chain needs the
// stream S[s], S[s+1], S[s+2] from the inner side, where -- for all four sides alike --
//     S[0], S[1]            corner values (A: B_0(r-1), A_1(r-1)   B: A_1(r), B_0(r-1)   C: D_0(r-1), C_1(r-1)   D: C_1(r), D_0(r-1))
//     S[m], 2 <= m < len    result of step m-2 of the SAME side's chain of ring r-1   (lane l-1, SKEW steps ago)
//     S[len]                the JOIN: last value of another side (A: D_last(r-1)  B: C_last(r-1)  C: B_last(r)  D: A_last(r))
//     S[len+1]              an OLD cell (it belongs to a chain of ring r that has not got there yet)
// Everything else in a window is OLD and streams from the layer, prefetched: one cell of the own line and one of the outer
// line per step.  NEW values never travel through memory: lane to lane inside a wavefront (the previous step's result read
// with a wave shift), through LDS between wavefronts (corner values, joins, the chain of the last ring of a 64-ring group), each
// LDS hand-over guarded by a monotonic progress counter that the consumer polls.  No barriers, no descriptors, no
// per-visit tables: the schedule is arithmetic on (side, ring, step).
//
// Exactness: per visit the float operations are the reference's, in its order (Eigen's unrolled tree over the 3x3 block
// in column-major order).  A neighbour's product w * g is formed once, where the neighbour is produced or loaded -- the
// same two floats the reference multiplies.
#pragma once

#include <float.h>
#include <stdint.h>
#include <string.h>

#include "gp_layout.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SW_HD __host__ __device__ __forceinline__
#else
#define SW_HD inline
#endif

namespace gg {
namespace sweep {

enum { SIDE_A = 0, SIDE_B = 1, SIDE_C = 2, SIDE_D = 3 };
// rings per group, steps between neighbouring rings, prefetch distance of the layer streams, steps per trip of the device loop.
// TRIP = 6 = lcm of the periods of everything that rotates per step -- the load queue, the 3-deep window lines and history, the
// 2-deep own line: a loop body of TRIP steps can carry every value in a fixed register.  The sweep is bound by instruction
// issue, not by memory (dropping every load and store changes nothing): PF = 3 steps (~4000 cycles) is ample and keeps the
// queue at 12 registers.
enum { LANES = 64, SKEW = GG_SWEEP_SKEW, PF = 3, TRIP = 6 };
static_assert(SKEW >= 1 && SKEW <= 3 && TRIP % SKEW == 0 && TRIP % PF == 0 && TRIP % 2 == 0, "the step code names t mod SKEW residues");

struct WP {
    float w, p; // confidence, confidence * ground
};
struct Cell {
    float g, w; // ground, confidence (the interleaved layer's element)
};
// what a preparing wavefront leaves for its chain wavefront per lane and wave-step ("Split steps" below)
struct PrepRec {
    float w_new;               // the visited cell's new confidence (decayed_confidence of its old one)
    float own_g, own_w, own_p; // the arriving own-line cell (the next successor): height, confidence, product
    float out_w, out_p;        // the arriving outer-line cell: confidence, product
};

// uniform parameters of one sweep
struct Params {
    int n;      // rows = cols
    int c;      // centre index n/2 - 1
    int rings;  // c - 1
    int groups; // ceil(rings / LANES)
    int waves_per_side;
    int gpw;    // ring groups per work-group ("part"); >= groups: one work-group sweeps the whole map (see "Parts" below)
    int split_steps; // 1: every chain wavefront has a PREPARING wavefront next to it (see "Split steps"; needs gpw == 1)
    int fresh_cell; // FRESH maps (gg_internal.h Arena::gp_bits): the layer element that holds the reset's (ground, confidence)
    int r2min;  // the confidence decay (:463-464) applies to cell (x, y) iff (x-c)^2 + (y-c)^2 >= r2min (host-computed, exact)
    double decrease, inv_decrease;
    int decay_fast;
    int poll_cap;    // polls after which a wait gives up (k4_sweep.hip "Bounded waits"; emulator: unused)
    int debug_fault; // tests only: 1 = the parts of a cloud take their tickets in REVERSE order (consumers start before their producers),
                     // 2 = the exporter withholds the joins of its last ring (the next part's wait must run out, not hang)
    int keep_points; // 1: the sweep as the stand-alone stage spiral_ground_interpolation (gg_run_stage): filter_cloud's reset of `points`
                     // (:147, folded into this kernel's prologue) is not part of it
    GpLayout gl; // where cell (row, col) of the layer lives (gp_layout.h): a wavefront's 64 cells of a step are contiguous
};

// ---------------------------------------------------------------------------------------------------------------------
// geometry of a side (compile-time SIDE)
// ---------------------------------------------------------------------------------------------------------------------
template <int SIDE> SW_HD int chain_len(int r) { return SIDE == SIDE_A ? 2 * r - 2 : SIDE == SIDE_D ? 2 * r : 2 * r - 1; }
template <int SIDE> SW_HD int chain_k0() { return (SIDE == SIDE_A || SIDE == SIDE_C) ? 2 : 1; }

// cell of `line` (-1 inner, 0 own, +1 outer) at along-position j of ring r, as its element index in the layer
template <int SIDE> SW_HD int side_cell(const Params &P, int r, int line, int j)
{
    const int rp = P.c - r, R = P.c + r;
    int x, y;
    if (SIDE == SIDE_A) {
        x = rp - line;
        y = rp + j;
    } else if (SIDE == SIDE_B) {
        x = rp + j;
        y = rp - line;
    } else if (SIDE == SIDE_C) {
        x = R + line;
        y = R - j;
    } else {
        x = R - j;
        y = R + line;
    }
    return gp_index(P.gl, x, y);
}

// (x - c)^2 + (y - c)^2 of the own-line cell at along-position j
template <int SIDE> SW_HD int side_r2(int r, int j)
{
    const int a = (SIDE == SIDE_A || SIDE == SIDE_B) ? j - r : r - j; // offset along the side from the centre column / row
    return r * r + a * a;
}

// position of window element (line, pos) in the 3x3 block's column-major linear order q = drow + 3 * dcol
//   line: 0 inner, 1 own, 2 outer        pos: 0 predecessor side, 1 self, 2 successor side
template <int SIDE> SW_HD constexpr int tree_pos(int line, int pos)
{
    return SIDE == SIDE_A   ? (2 - line) + 3 * pos        // rows: outer, own, inner    cols: pred, self, succ
           : SIDE == SIDE_B ? pos + 3 * (2 - line)        // rows: pred, self, succ     cols: outer, own, inner
           : SIDE == SIDE_C ? line + 3 * (2 - pos)        // rows: inner, own, outer    cols: succ, self, pred
                            : (2 - pos) + 3 * line;       // rows: succ, self, pred     cols: inner, own, outer
}

// ---------------------------------------------------------------------------------------------------------------------
// one visit: interpolate_cell (:445-465) on a prepared window
// ---------------------------------------------------------------------------------------------------------------------
SW_HD float sw_tree9(const float *e) { return ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + (e[7] + e[8]))); }
SW_HD double sw_max(double a, double b) { return (a < b) ? b : a; } // libstdc++ std::max

// :463-464  occupied = (float)max(x - x / decrease, 0.001) in double, for cells outside the decay radius.  It depends on the
// cell's OLD confidence only, so it is computed ahead of the height chain.  Multiply form when it provably rounds alike:
// for decrease >= 1.25 the product form differs from the quotient form by < 2^-49 relative, the max and the float conversion
// are monotonic, so if the two ends of the +-2^-48 interval convert to the same float the exact value does too; otherwise
// (about 1 visit in 10^7, NaN, or an unusual config) the divide decides.
SW_HD float sw_maxf(float a, float b) { return (a < b) ? b : a; }
// "does any lane of the wavefront need this?" -- the device takes rare paths as wave-uniform branches (a divergent branch costs
// half a dozen scalar instructions of exec-mask bookkeeping in EVERY step); on the host a lane is its own wavefront
#if defined(__HIP_DEVICE_COMPILE__)
SW_HD bool sw_any(bool b) { return __any(b) != 0; }
#define SW_KEEP_BRANCH() __asm__ volatile("; rare path" ::: "memory")
#else
SW_HD bool sw_any(bool b) { return b; }
#define SW_KEEP_BRANCH() ((void)0)
#endif

SW_HD float decayed_confidence(float occupied, bool decay, const Params &P)
{
    // (float)std::max(v, 0.001) == max((float)v, (float)0.001) with the same NaN behaviour: the conversion is monotonic
    const float floor_f = (float)0.001;
    const double x = (double)occupied;
    const double t = x - x * P.inv_decrease;
    // t is within a relative 2^-49 of the reference's x - x / decrease (decrease >= 1.25), i.e. within 32 of its own ulps, so the
    // two convert to the same float unless t's significand lies that close to a binary32 rounding boundary -- the midpoint of two
    // floats: low 29 bits of the significand = 2^28.  Integer test on the low word (64 ulps of margin) instead of converting both
    // ends of the interval: the sweep is bound by instruction issue, and this is three integer instructions for two binary64
    // multiplies, a conversion and a max.  (Results below the floor or not finite need no care: 0.001, inf and NaN come out of
    // either form alike.)
    uint64_t bits;
#if defined(__HIP_DEVICE_COMPILE__)
    bits = (uint64_t)__double_as_longlong(t);
#else
    memcpy(&bits, &t, 8);
#endif
    const uint32_t low = (uint32_t)bits & 0x1FFFFFFFu;
    const bool near_boundary = (low - (0x10000000u - 64u)) <= 128u;
    float d = sw_maxf((float)t, floor_f);
    const bool exact = (!P.decay_fast || near_boundary) && decay;
    if (sw_any(exact)) {
        SW_KEEP_BRANCH();
        const float de = (float)sw_max(x - x / P.decrease, 0.001);
        d = exact ? de : d;
    }
    return decay ? d : occupied;
}

// :457-460 the new height from the prepared window (w = confidences, p = confidence * height, column-major block order)
SW_HD float interpolated_height(const float (&w)[9], const float (&p)[9], float height, float occupied)
{
    const float gvlSum = sw_tree9(w) + FLT_MIN;          // :457
    const float avg = sw_tree9(p) / gvlSum;              // :458
    return (1.0f - occupied) * avg + occupied * height;  // :460
}

// the same with the window as (w, p) pairs: both trees have one shape, so every addition is one packed instruction on the device
#if defined(__HIP_DEVICE_COMPILE__)
typedef float sw_f2 __attribute__((ext_vector_type(2)));
SW_HD float interpolated_height2(const WP (&e)[9], float height, float occupied)
{
    sw_f2 v[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) v[k] = sw_f2{e[k].w, e[k].p};
    const sw_f2 sum = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + (v[7] + v[8])));
    const float gvlSum = sum.x + FLT_MIN;                // :457
    const float avg = sum.y / gvlSum;                    // :458
    return (1.0f - occupied) * avg + occupied * height;  // :460
}
#else
SW_HD float interpolated_height2(const WP (&e)[9], float height, float occupied)
{
    float w[9], p[9];
    for (int k = 0; k < 9; ++k) {
        w[k] = e[k].w;
        p[k] = e[k].p;
    }
    return interpolated_height(w, p, height, occupied);
}
#endif

SW_HD Cell visit(const float (&w)[9], const float (&p)[9], float height, float occupied, bool decay, const Params &P)
{
    Cell out;
    out.w = decayed_confidence(occupied, decay, P);
    out.g = interpolated_height(w, p, height, occupied);
    return out;
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS image of one sweep: progress counters + the values that cross wavefronts.  Offsets in 4-byte words.
// ---------------------------------------------------------------------------------------------------------------------
struct LdsMap {
    int corner_done; // [2]   rings finished by the AB / CD corner lane
    int join_done;   // [4]   highest ring whose chain of that side has published its last value
    int bnd_done;    // [4][groups]   steps published by the last lane of a group (consumed by lane 0 of the next group)
    int prep_done;   // [4]   "split steps" (below): wave-steps the side's preparing wavefront has put into the ring
    int take_done;   // [4]   ... and wave-steps the side's chain wavefront has taken out of it
    int corner;      // WP[2][c][2]   AB: (A_1, B_0), CD: (C_1, D_0) per ring; ring 0 = the centre cell
    int join;        // WP[4][c]      last chain value per side and ring
    int bnd;         // WP[4][bnd_words / 2]   full chains of the group-boundary rings
    int bnd_stride;  // WP entries per side
    int bnd_base;    // bnd_offset() of the first boundary this work-group's table holds (a part keeps only its own boundaries)
    int scratch;     // per-lane dummy targets of conditional publishes (2 data words + 1 counter word per lane, 8-byte aligned)
    int prep;        // PrepRec[4][PREP_DEPTH][LANES]   ring of prepared wave-steps per side (split steps only; else 0 words)
    int words;       // total size
};

SW_HD int bnd_offset(int b) { return 64 * b * (b + 1) + 2 * b; } // boundary b = ring 64 (b + 1): chains of <= 128 (b + 1) + 2 steps before it

// ---------------------------------------------------------------------------------------------------------------------
// Parts.  A map with more ring groups than a work-group has wavefronts for (n = 1000: 8 groups per side, 16 wavefronts per
// work-group) is swept by several work-groups: part p owns the consecutive groups [p gpw, (p + 1) gpw).  Everything a part
// needs from the part inside it is FEED-FORWARD -- the chains of its innermost ring read the boundary chains, corner values
// and two joins of the ring before, nothing flows back -- so the hand-over may take its time: the producing work-group's chain
// and corner wavefronts publish into their LDS tables as if the next group lived next door, an EXPORTER wavefront copies what
// appears there, tagged with the launch's sequence number, into an exchange region in global memory (the chain wavefronts
// themselves must not: loads and stores share one in-order counter, and a store that has to reach memory before it is
// acknowledged would hold up every later wait of the wavefront -- measured: 2x slower), and an IMPORTER wavefront of the
// consuming work-group polls the region and feeds the values and their progress counters into its own work-group's LDS tables,
// where the chain and corner wavefronts find them as if a wavefront next door had published them.  Which work-group sweeps which
// part is decided by a ticket taken when the work-group starts running (k4_sweep.hip), parts of a cloud in increasing order: a
// consumer never waits for a producer that has not started, in whatever order work-groups are dispatched.
//
// Exchange region of one cloud, in entries of one WP: per boundary gb = 1 .. groups - 1 (between ring 64 gb and 64 gb + 1)
//   4 sides x xchg_len(gb) chain values of ring 64 gb,  then  AB x1, AB y0, CD x1, CD y0,  then  C_last, D_last of ring 64 gb.
// ---------------------------------------------------------------------------------------------------------------------
SW_HD int xchg_len(int gb) { return 2 * LANES * gb + 2; }
SW_HD int xchg_base(int gb)
{
    int o = 0;
    for (int g = 1; g < gb; ++g) o += 4 * xchg_len(g) + 6;
    return o;
}
SW_HD int xchg_entries(int groups) { return xchg_base(groups > 1 ? groups : 1); }
enum { X_CORNER = 0, X_JOIN_C = 4, X_JOIN_D = 5 };
SW_HD int xchg_chain(int gb, int side, int e) { return xchg_base(gb) + side * xchg_len(gb) + e; }
SW_HD int xchg_misc(int gb, int k) { return xchg_base(gb) + 4 * xchg_len(gb) + k; }

// g0, g1: the groups of the work-group (default: all of them)
enum { PREP_DEPTH = TRIP, PREP_WORDS = 6 }; // ring slots per side (= TRIP: the slot of a step is a constant of the unrolled loop); words of one PrepRec
SW_HD LdsMap lds_layout(int c, int groups, int g0 = 0, int g1 = -1, bool split_steps = false)
{
    if (g1 < 0) g1 = groups;
    LdsMap m;
    int o = 0;
    m.corner_done = o;
    o += 2;
    m.join_done = o;
    o += 4;
    m.bnd_done = o;
    o += 4 * groups;
    m.prep_done = o;
    o += 4;
    m.take_done = o;
    o += 4;
    o = (o + 1) & ~1;
    m.corner = o;
    o += 2 * c * 2 * 2;
    m.join = o;
    o += 4 * c * 2;
    // boundaries b = g0 - 1 (imported from the part inside) .. g1 - 1 (the last one is read by the exporter wavefront when another
    // part follows; the last ring of the map has no boundary chain)
    const int b_first = g0 > 0 ? g0 - 1 : 0, b_end = g1 < groups ? g1 : (g1 > 0 ? g1 - 1 : 0);
    m.bnd_base = bnd_offset(b_first);
    m.bnd_stride = bnd_offset(b_end > b_first ? b_end : b_first) - m.bnd_base;
    m.bnd = o;
    o += 4 * m.bnd_stride * 2;
    m.scratch = o; // [64][3]: where the lanes that have nothing to publish write (publish_if, device)
    o += LANES * 3 + 1;
    o = (o + 1) & ~1;
    m.prep = o;
    if (split_steps) o += 4 * PREP_DEPTH * LANES * PREP_WORDS;
    m.words = o;
    return m;
}

// ---------------------------------------------------------------------------------------------------------------------
// Memory back end.  The device one (k4_sweep.hip) issues real loads / LDS operations; the host one can resolve a queued
// layer load as late as its use (worst case for write-after-read hazards) and counts what it is asked to do.
//   Cell  load_issue(bool valid, int cell)    start a load of the interleaved layer element (invalid: no traffic, value 0)
//   Cell  load_value(const Cell &queued, bool valid, int cell)   the value at use time (device: `queued` itself)
//   Cell  fresh(Cell v)                   v in registers of its own (host: identity)
//   void  mark(int k)                     timing instrumentation point (no-op unless a tool asks for it)
//   void  store(bool valid, int cell, Cell v)
//   int   counter(int word)               read a progress counter (LDS)
//   void  counters3(w0, w1, w2, &v0, &v1, &v2)   three of them in one round trip
//   void  publish(int data_word, WP v, int counter_word, int value)    LDS data, then counter -- in this order
//   void  publish_if(bool c, int lane, ...)   the same for the lanes with c; branch-free on the device (the others write to
//                                             their scratch words: a divergent branch is ~7 scalar instructions of exec bookkeeping)
//   void  put(int data_word, WP v)  /  WP get(int data_word)  /  put_if(bool c, int lane, L, word, v)
//   WP    bcast(WP v, int lane)           lane's value in every lane (device: v_readlane)
// ---------------------------------------------------------------------------------------------------------------------

// ---------------------------------------------------------------------------------------------------------------------
// What a chain wavefront waits for.  Everything that crosses wavefronts sits behind three monotonic counters: the corner
// lane's ring count, the partner side's "last value published up to ring", the previous group's boundary chain length.
// What a group needs of each is a non-decreasing step function of the wave-step t in closed form (a lone wavefront issues ~1
// instruction per 4 cycles of ANY kind: measured, the first version of this kernel spent more time on its scalar bookkeeping
// than on the float arithmetic):
//   corner  ring of the latest lane whose first step is <= t                  (lanes start every SKEW steps)
//   join    join ring of the latest lane whose join step (s = len - 2) is <= t  (every SKEW + 2 steps)
//   bnd     t + 1 while lane 0 still reads the previous group's chain
// The counters are cached; LDS is polled only when a cached value is too small.
// ---------------------------------------------------------------------------------------------------------------------
template <int SIDE> struct ChainSync {
    enum { NEVER = 0x7fffffff, ALWAYS = 0x3fffffff };
    int t_;                                     // the wave-step in question (advance)
    int safe_a, safe_b;                         // the last wave-step whose step_a / step_b the cached counters cover (cover())
    int have_corner, have_join, have_bnd;       // cached counter values (lower bounds)
    int start_t0, start_r0, n_start;            // first lane start: time, ring; number of lanes with a chain
    int join_t0, join_r0;                       // first join: time, join ring (one per started lane, every SKEW + 2 steps)
    int bnd_end;                                // lane 0 reads the previous group's chain for 0 <= t < bnd_end
    int w_corner, w_join, w_bnd;                // LDS words of the three counters

    SW_HD void init(int r0, int nl, int group, const Params &P, const LdsMap &L)
    {
        const int b = SIDE == SIDE_A ? -2 : SIDE == SIDE_D ? 0 : -1; // len(r) = 2 r + b
        const int side_from = SIDE == SIDE_A ? SIDE_D : SIDE == SIDE_B ? SIDE_C : SIDE == SIDE_C ? SIDE_B : SIDE_A;
        w_corner = L.corner_done + ((SIDE == SIDE_A || SIDE == SIDE_B) ? 0 : 1);
        w_join = L.join_done + side_from;
        w_bnd = L.bnd_done + SIDE * P.groups + (group > 0 ? group - 1 : 0);
        have_corner = have_join = have_bnd = 0;
        // lanes with a chain (len >= 1) start at t = SKEW * l and join (s = len - 2) at t = (SKEW + 2) l + 2 r0 + b - 2
        int l1 = (1 - b + 1) / 2 - r0; // smallest l with 2 (r0 + l) + b >= 1
        if (l1 < 0) l1 = 0;
        n_start = l1 < nl ? nl - l1 : 0;
        start_t0 = l1 < nl ? SKEW * l1 : (int)NEVER;
        start_r0 = r0 + l1;
        join_t0 = l1 < nl ? (SKEW + 2) * l1 + 2 * r0 + b - 2 : (int)NEVER;
        join_r0 = (SIDE == SIDE_A || SIDE == SIDE_B) ? r0 + l1 - 1 : r0 + l1;
        bnd_end = group > 0 ? chain_len<SIDE>(r0) - 2 : 0;
        t_ = -2 - (int)PF; // group_first_step()
        cover();
    }
    // requirements of wave-step t: non-decreasing step functions of t in closed form
    SW_HD int need_corner_at(int t) const
    {
        const int ds = t - start_t0; // (NEVER: negative for every t)
        const int ks = (int)((unsigned)ds / (unsigned)SKEW), last = n_start - 1; // (ks only used when non-negative)
        return ds < 0 ? 0 : start_r0 + (ks < last ? ks : last);
    }
    SW_HD int need_join_at(int t) const
    {
        const int dj = t - join_t0;
        const int kj = (int)((unsigned)dj / ((unsigned)SKEW + 2u)), last = n_start - 1;
        return dj < 0 ? 0 : join_r0 + (kj < last ? kj : last);
    }
    SW_HD int need_bnd_at(int t) const
    {
        const int tb = t + 1 < bnd_end ? t + 1 : bnd_end;
        return tb > 0 ? tb : 0;
    }
    // ... and their inverses: the last step each cached counter is good for.  The per-step test is then ONE scalar compare per half
    // step; the closed forms above run only when a wavefront has to poll (they took 22 scalar instructions of every step, a sixth of
    // it, and a lone wavefront pays for a scalar instruction what it pays for a vector one).
    SW_HD void cover()
    {
        const int last = n_start - 1;
        int sc, kc = have_corner - start_r0;
        if (start_t0 == (int)NEVER || kc >= last) sc = ALWAYS;
        else if (kc < 0) sc = start_t0 - 1;
        else sc = start_t0 + (int)SKEW * kc + (int)SKEW - 1;
        const int sb = have_bnd >= bnd_end ? (int)ALWAYS : have_bnd - 1;
        safe_a = sc < sb ? sc : sb;
        cover_b();
    }
    SW_HD void cover_b()
    {
        const int kj = have_join - join_r0;
        if (join_t0 == (int)NEVER || kj >= n_start - 1) safe_b = ALWAYS;
        else if (kj < 0) safe_b = join_t0 - 1;
        else safe_b = join_t0 + ((int)SKEW + 2) * kj + (int)SKEW + 1;
    }
    SW_HD void advance(int t) { t_ = t; }
    SW_HD bool ok_a() const { return t_ <= safe_a; } // what step_a reads is there as far as the cached counters know
    SW_HD bool ok_b() const { return t_ <= safe_b; } // what step_b reads
    SW_HD bool slow_ok_a() const { return have_corner >= need_corner_at(t_) && have_bnd >= need_bnd_at(t_); } // the same from the closed forms
    SW_HD bool slow_ok_b() const { return have_join >= need_join_at(t_); }
    template <class Mem> SW_HD void poll(Mem &mem) { mem.counters3(w_corner, w_join, w_bnd, have_corner, have_join, have_bnd); }
    template <class Mem> SW_HD void refresh(Mem &mem)
    {
        poll(mem);
        cover();
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// one lane of a chain wavefront
// ---------------------------------------------------------------------------------------------------------------------
// The step is branch-free on purpose: every lane issues the same memory and LDS operations every step (addresses of lanes
// that have nothing to fetch are out of range: no traffic) and the rare events -- first step, join, last step -- are
// selects on per-lane compares of precomputed constants with the (scalar) step number.  A conditional load would make the
// compiler drain the whole prefetch queue (s_waitcnt vmcnt(0)).
template <int SIDE> struct ChainLane {
    // per-lane constants
    int l, r, len;      // lane in the group, ring, chain length (0: idle lane)
    int l3, lend;       // SKEW * l (step s = t - l3),  l3 + len
    int lim;            // len > 0 ? len + 2 : 0: a column (along-position k0 + s + 1) exists iff (unsigned)(s + 2) < lim
    int ownA, outA;     // layer element of the own / outer line at step s = ownA / outA + 64 * (t + 1)  [own: column; - 64: the visited cell]
    int xold_cell, own_end; // layer elements of S[len + 1] and of the own line's far end k0 + len (it belongs to another side)
    int a_s0, a_s1, a_pred, a_join, a_bnd, a_pub; // LDS words: S[0], S[1], predecessor, join, previous group's chain, own join slot
    int r2c, r2r;       // (x-c)^2 + (y-c)^2 of the visited cell = r2r + (t + r2c)^2
    int pb_word, pb_step, pb_cnt, pb_idle; // the boundary chain the group's last lane leaves for the next group: LDS word of step t = pb_word + pb_step t and
                                           // its counter (the other lanes: their scratch words pb_idle, pb_idle + 2, step 0)
    // wave-uniform constants of the group (scalar registers on the device): which steps can contain the rare per-lane events
    int u_start_last;   // lanes take their corner values at steps 0, SKEW, .. <= u_start_last
    int u_join_first, u_join_last; // a lane reads its join at t = lend - 2: only for t in this range
    int u_len0;         // chain length of lane 0 (it reads the previous group's boundary chain while t < u_len0)
    int u_l3_last, u_lend_last; // the last lane's first step and end (its results are the next group's boundary chain)
    // window: inner and outer lines as (w, p); own line: predecessor (w, p) new, self and successor old (g, w, p)
    WP I[3], U[3], OP;
    float Sg, Sw, Sp, Ng, Nw, Np;
    WP xold;            // S[len + 1]
    WP h1, h2, h3;      // results of the last three steps (lane l + 1 reads the one of SKEW steps ago: handed_over())
    Cell q_own[PF], q_out[PF]; // layer loads in flight, one slot per wave-step mod PF
    // handed from the first half of a step to the second
    WP xa, cs0, cs1, cpred;
    float w_new_;
    bool xold_bit = true, end_bit = true; // FRESH maps: are the two cells off the lines (xold_cell, own_end) in memory

    SW_HD void init(int lane, int r0, int nl, int group, const Params &P, const LdsMap &L)
    {
        l = lane;
        const bool live = lane < nl;
        r = live ? r0 + lane : r0; // (idle lanes keep a valid ring for their never-used addresses)
        len = live ? chain_len<SIDE>(r) : 0;
        const int k0 = chain_k0<SIDE>();
        l3 = SKEW * l;
        lend = l3 + len;
        lim = len > 0 ? len + 2 : 0;
        const int own1 = side_cell<SIDE>(P, r, 0, 1), out1 = side_cell<SIDE>(P, r, 1, 1); // along-position 1; stride 64 per position
        // column of step s = position k0 + s + 1 = element own1 + 64 (k0 + s) = own1 + 64 (k0 - 1 - l3) + 64 (t + 1)
        ownA = own1 + 64 * (k0 - 1 - l3);
        outA = out1 + 64 * (k0 - 1 - l3);
        xold_cell = side_cell<SIDE>(P, r, -1, k0 + len);
        own_end = side_cell<SIDE>(P, r, 0, k0 + len);
        const int side = (SIDE == SIDE_A || SIDE == SIDE_B) ? 0 : 1;
        const int own_first = L.corner + 2 * ((side * P.c + r) * 2), own_second = own_first + 2;
        const int in_first = L.corner + 2 * ((side * P.c + r - 1) * 2), in_second = in_first + 2;
        const int side_from = SIDE == SIDE_A ? SIDE_D : SIDE == SIDE_B ? SIDE_C : SIDE == SIDE_C ? SIDE_B : SIDE_A;
        a_join = L.join + 2 * (side_from * P.c + ((SIDE == SIDE_A || SIDE == SIDE_B) ? r - 1 : r));
        a_pub = L.join + 2 * (SIDE * P.c + r);
        if (SIDE == SIDE_A) { // S[0] = B_0(r-1), S[1] = A_1(r-1), predecessor A_1(r)
            a_s0 = in_second;
            a_s1 = in_first;
            a_pred = own_first;
        } else if (SIDE == SIDE_B) { // A_1(r), B_0(r-1), predecessor B_0(r)
            a_s0 = own_first;
            a_s1 = in_second;
            a_pred = own_second;
        } else if (SIDE == SIDE_C) { // D_0(r-1), C_1(r-1) -- for ring 1 that cell is B_last(1), i.e. the join --, predecessor C_1(r)
            a_s0 = in_second;
            a_s1 = r == 1 ? a_join : in_first;
            a_pred = own_first;
        } else { // C_1(r), D_0(r-1), predecessor D_0(r)
            a_s0 = own_first;
            a_s1 = in_second;
            a_pred = own_second;
        }
        a_bnd = group > 0 ? L.bnd + 2 * (SIDE * L.bnd_stride + bnd_offset(group - 1) - L.bnd_base) : L.bnd;
        // decay test of the visited cell (along-position k0 + s, s = t - l3): offset from the centre line = (k0 + s) - r (sides
        // A, B) or r - (k0 + s) (C, D); its square is the same
        r2c = k0 - r - l3;
        r2r = r * r;
        pb_idle = (L.scratch + 1 + 3 * l) & ~1;
        pb_word = l == LANES - 1 ? L.bnd + 2 * ((SIDE * L.bnd_stride) + bnd_offset(group) - L.bnd_base - l3) : pb_idle;
        pb_step = l == LANES - 1 ? 2 : 0;
        pb_cnt = l == LANES - 1 ? L.bnd_done + SIDE * P.groups + group : pb_idle + 2;
        u_start_last = SKEW * (nl - 1);
        u_len0 = chain_len<SIDE>(r0);
        u_join_first = u_len0 - 2;
        u_l3_last = SKEW * (LANES - 1);
        u_lend_last = SKEW * (nl - 1) + chain_len<SIDE>(r0 + nl - 1); // (lends grow with the lane)
        u_join_last = u_lend_last - 2;
        I[0] = I[1] = I[2] = U[0] = U[1] = U[2] = OP = xold = h1 = h2 = h3 = WP{0.f, 0.f};
        Sg = Sw = Sp = Ng = Nw = Np = 0.f;
        for (int k = 0; k < PF; ++k) q_own[k] = q_out[k] = Cell{0.f, 0.f};
    }

    SW_HD WP handed_over() const { return SKEW == 1 ? h1 : SKEW == 2 ? h2 : h3; }

    // One wave-step.  `slot` = wave-step mod PF and tmod = ((t mod SKEW) + SKEW) mod SKEW, both compile-time constants in the
    // device's unrolled loop; x_in = (lane l - 1).handed_over() as it was BEFORE this step (lane 0: anything); lane0 = (l == 0).
    //
    // The step comes in two halves.  step_a is everything that does not need the JOIN (the partner side's last value of the
    // ring inside): the new confidence, the layer loads, the own and outer window lines, the corner and boundary values.
    // step_b takes the join, completes the inner line and makes the visit.  A wavefront waits for the join between the two:
    // the sides of a ring hand their ends to each other ring after ring (B -> C -> B ..., A -> D -> A ...), that cycle is the
    // sweep's critical path, and this way only the second half of a step sits on it.
    // What a step can contain is decided by wave-uniform ranges of t (u_*).  The caller may know more than the step -- a whole trip
    // of the unrolled loop inside or outside a range -- and says so (the device's run_chain; sweep_emul.hip runs the defaults):
    //   STARTS = false   t > u_start_last: no lane takes its corner values any more
    //   BND    0 / 2     no step / every step reads the previous group's boundary chain (2: has_prev_group and 0 <= t < u_len0)
    //   JOIN   0 / 2     no step / every step can be a lane's join step (2: t >= u_join_first; past u_join_last the read is unused)
    //   BND, JOIN = 3    the caller tested the ranges once for the trip: has_prev_group / trip_join / has_next_group ARE the answers (a
    //                    step outside a range that its trip touches reads a value no lane selects, and publishes from no active lane)
    // (1 = test per step).  A range test is seven scalar instructions and a taken branch, the blocks meet the step at control-flow
    // joins that cost register copies, and a lone wavefront pays ~5 cycles for an instruction of any kind.
    // FRESH (device only; gg_internal.h Arena::gp_bits): the map's layer holds the reset's pair by definition, only the cells marked in the
    // bit map are in memory -- own_bit / out_bit say so for the cells this step REQUESTS on the own and the outer line; the others are read
    // from P.fresh_cell, the one element that holds the reset's pair
    template <bool STARTS = true, int BND = 1, bool FRESH = false, class Mem>
    SW_HD void step_a(int t, int slot, int tmod, WP x_in, const Params &P, const LdsMap &L, bool has_prev_group, Mem &mem, bool own_bit = true, bool out_bit = true)
    {
        (void)L;
        // ---- the visited cell's new confidence depends on its old one only (its rare exact path is the step's only branch
        //      besides the publishes at the end: what follows is one basic block for the instruction scheduler)
        const int ao = t + r2c;
        w_new_ = decayed_confidence(Nw, r2r + ao * ao >= P.r2min, P); // (the successor of the last step becomes "self" now)
        // ---- LDS: what this lane could need (garbage until published; selected only when it is) -- but only in the
        //      (wave-uniform) ranges of steps in which some lane can be at that event
        WP c_bnd{0.f, 0.f};
        if (BND == 2 || (BND == 3 && has_prev_group) || (BND == 1 && has_prev_group && t >= 0 && t < u_len0)) c_bnd = mem.get(a_bnd + (l == 0 ? 2 * t : 0));
        if (STARTS && tmod == 0 && t >= 0 && t <= u_start_last) { // a lane's first step (t = SKEW l): the corner values
            cs0 = mem.get(a_s0);
            cs1 = mem.get(a_s1);
            cpred = mem.get(a_pred);
        }
        // ---- the column that arrives now (along-position k0 + s + 1), requested PF steps ago into this slot; the own-line
        //      request of step -2 (the predecessor's cell, which is never read from the layer) carries the old cell S[len + 1].
        //      (mem.fresh: on the device a real register copy -- the arriving values stay live for two or three more steps
        //      as window elements; copied out, the slot's registers are free for the request below and the compiler need not
        //      copy freshly requested registers at the loop's back edge, which would be a wait for loads just issued)
        const unsigned ua = (unsigned)(t + 2 - l3); // s + 2
        const bool col = ua < (unsigned)lim;
        const int own_col = ownA + 64 * (t + 1);
        int own_now = (int)ua == len + 1 ? own_end : own_col;
        own_now = ua == 0u ? xold_cell : own_now;
        const Cell own = mem.fresh(mem.load_value(q_own[slot], col, own_now));
        const Cell out = mem.fresh(mem.load_value(q_out[slot], col, outA + 64 * (t + 1)));
        // ---- and the one to request for step s + PF
        const unsigned uq = ua + (unsigned)PF;
        const bool colq = uq < (unsigned)lim;
        const int own_colq = ownA + 64 * (t + 1 + PF);
        int own_req = (int)uq == len + 1 ? own_end : own_colq; // (two selects on ready values: no branch)
        own_req = uq == 0u ? xold_cell : own_req;
        int out_req = outA + 64 * (t + 1 + PF);
        if (FRESH) {
            bool ob = (int)uq == len + 1 ? end_bit : own_bit;
            ob = uq == 0u ? xold_bit : ob;
            own_req = ob ? own_req : P.fresh_cell;
            out_req = out_bit ? out_req : P.fresh_cell;
        }
        q_own[slot] = mem.load_issue(colq, own_req);
        q_out[slot] = mem.load_issue(colq, out_req);
        // ---- advance the window
        Sg = Ng;
        Sw = Nw;
        Sp = Np;
        Ng = own.g;
        Nw = own.w;
        Np = own.w * own.g;
        xold = (tmod == (2 * SKEW - 2) % SKEW && ua == 0u) ? WP{own.w, Np} : xold; // step -2: t + 2 = SKEW l, i.e. only when t = -2 (mod SKEW)
        U[0] = U[1];
        U[1] = U[2];
        U[2] = WP{out.w, out.w * out.g};
        // stream element S[s + 2]: the inner lane's step s (SKEW wave-steps ago; lane 0: the previous group's boundary chain),
        // at the far end the old cell; the join comes in step_b
        xa = (BND != 0 && l == 0) ? c_bnd : x_in; // (BND = 0: lane 0's chain is over, what it takes in is not used)
        xa = t + 1 == lend ? xold : xa; // s + 2 == len + 1
    }

    // step_a when a preparing wavefront did the layer half (split steps): `rec` is its record of this wave-step
    template <bool STARTS = true, int BND = 1, class Mem>
    SW_HD void take(int t, int tmod, const PrepRec &rec, WP x_in, bool has_prev_group, Mem &mem)
    {
        w_new_ = rec.w_new;
        WP c_bnd{0.f, 0.f};
        if (BND == 2 || (BND == 3 && has_prev_group) || (BND == 1 && has_prev_group && t >= 0 && t < u_len0)) c_bnd = mem.get(a_bnd + (l == 0 ? 2 * t : 0));
        if (STARTS && tmod == 0 && t >= 0 && t <= u_start_last) {
            cs0 = mem.get(a_s0);
            cs1 = mem.get(a_s1);
            cpred = mem.get(a_pred);
        }
        const unsigned ua = (unsigned)(t + 2 - l3); // s + 2
        Sg = Ng;
        Sw = Nw;
        Sp = Np;
        Ng = rec.own_g;
        Nw = rec.own_w;
        Np = rec.own_p;
        xold = (tmod == (2 * SKEW - 2) % SKEW && ua == 0u) ? WP{rec.own_w, rec.own_p} : xold;
        U[0] = U[1];
        U[1] = U[2];
        U[2] = WP{rec.out_w, rec.out_p};
        xa = (BND != 0 && l == 0) ? c_bnd : x_in; // (BND = 0: lane 0's chain is over, what it takes in is not used)
        xa = t + 1 == lend ? xold : xa;
    }

    // join_turn: (wave-uniform) some lane of the group can end its chain at this step -- lane l ends at t + 1 = 3 l + lend of
    // lane 0 (SKEW = 1), i.e. in every third step only; see join_turn_of
    template <bool STARTS = true, int JOIN = 1, class Mem>
    SW_HD void step_b(int t, int tmod, const Params &P, const LdsMap &L, bool has_next_group, int group, Mem &mem, bool join_turn = true, WP join_read = WP{0.f, 0.f},
                      bool trip_join = true)
    {
        // (JOIN = 2: the caller has read this lane's join slot -- early, so that the LDS round trip is over when the step gets here)
        WP x = xa;
        if (JOIN == 2 || (JOIN == 3 && trip_join) || (JOIN == 1 && t >= u_join_first && t <= u_join_last)) {
            const WP c_join = JOIN == 2 ? join_read : mem.get(a_join);
            x = t + 2 == lend ? c_join : x; // s + 2 == len
        }
        I[0] = I[1];
        I[1] = I[2];
        I[2] = x;
        if (STARTS && tmod == 0 && t >= 0 && t <= u_start_last) {
            const bool first = t == l3;
            I[0] = first ? cs0 : I[0];
            I[1] = first ? cs1 : I[1];
            OP = first ? cpred : OP;
        }
        // ---- the visit
        const bool active = (unsigned)(t - l3) < (unsigned)len;
        WP win[9];
        win[tree_pos<SIDE>(0, 0)] = I[0];
        win[tree_pos<SIDE>(0, 1)] = I[1];
        win[tree_pos<SIDE>(0, 2)] = I[2];
        win[tree_pos<SIDE>(1, 0)] = OP;
        win[tree_pos<SIDE>(1, 1)] = WP{Sw, Sp};
        win[tree_pos<SIDE>(1, 2)] = WP{Nw, Np};
        win[tree_pos<SIDE>(2, 0)] = U[0];
        win[tree_pos<SIDE>(2, 1)] = U[1];
        win[tree_pos<SIDE>(2, 2)] = U[2];
        Cell v;
        v.w = w_new_;
        v.g = interpolated_height2(win, Sg, Sw);
        mem.store(active, ownA + 64 * t, v);
        const WP res = WP{v.w, v.w * v.g};
        OP = active ? res : OP;
        h3 = h2;
        h2 = h1;
        h1 = res;
        // ---- publish what other wavefronts wait for (data first, then the counter)
        if (join_turn) mem.publish_if(t + 1 == lend && len > 0, l, L, a_pub, res, L.join_done + SIDE, r);
        // (uniform: only while the last lane runs.  Without STARTS t > u_start_last = u_l3_last when there is a next group, and past
        // u_lend_last no lane is active)
        if (has_next_group && (!STARTS || JOIN == 3 || (t >= u_l3_last && t < u_lend_last)))
            mem.publish(active ? pb_word + pb_step * t : pb_idle, res, active ? pb_cnt : pb_idle + 2, t - l3 + 1);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// Split steps (latency launches: one ring group per work-group).  A lone wavefront issues one instruction per 4-8 cycles
// whatever its kind, and the sweep of one cloud is as long as its ~540 dependent wave-steps: whatever a chain wavefront does
// per step that does NOT depend on the values handed to it -- the layer loads, the decay of the visited cell's confidence, the
// products of the arriving own-line and outer-line cells -- is done by a second, PREPARING wavefront of the same side, which
// runs a few steps ahead and leaves one PrepRec per lane and step in an LDS ring (PREP_DEPTH steps).  The chain wavefront
// takes the record instead of executing step_a's first half: about half of its instructions per step are gone.  Two
// monotonic counters per side guard the ring (prepared / taken steps, counted from the group's first step).
// ---------------------------------------------------------------------------------------------------------------------
template <int SIDE> struct PrepLane {
    int l3, len, lim, ownA, outA, xold_cell, own_end, r2c, r2r;
    float Nw;                 // confidence of the cell that arrived last (= the cell visited in the next step)
    Cell q_own[PF], q_out[PF];
    SW_HD void init(int lane, int r0, int nl, const Params &P)
    {
        const bool live = lane < nl;
        const int r = live ? r0 + lane : r0;
        len = live ? chain_len<SIDE>(r) : 0;
        const int k0 = chain_k0<SIDE>();
        l3 = SKEW * lane;
        lim = len > 0 ? len + 2 : 0;
        ownA = side_cell<SIDE>(P, r, 0, 1) + 64 * (k0 - 1 - l3);
        outA = side_cell<SIDE>(P, r, 1, 1) + 64 * (k0 - 1 - l3);
        xold_cell = side_cell<SIDE>(P, r, -1, k0 + len);
        own_end = side_cell<SIDE>(P, r, 0, k0 + len);
        r2c = k0 - r - l3;
        r2r = r * r;
        Nw = 0.f;
        for (int k = 0; k < PF; ++k) q_own[k] = q_out[k] = Cell{0.f, 0.f};
    }
    // the layer half of ChainLane::step_a, for wave-step t (slot = wave-step mod PF)
    template <class Mem> SW_HD PrepRec step(int t, int slot, const Params &P, Mem &mem)
    {
        PrepRec rec;
        const int ao = t + r2c;
        rec.w_new = decayed_confidence(Nw, r2r + ao * ao >= P.r2min, P);
        const unsigned ua = (unsigned)(t + 2 - l3);
        const bool col = ua < (unsigned)lim;
        int own_now = (int)ua == len + 1 ? own_end : ownA + 64 * (t + 1);
        own_now = ua == 0u ? xold_cell : own_now;
        const Cell own = mem.fresh(mem.load_value(q_own[slot], col, own_now));
        const Cell out = mem.fresh(mem.load_value(q_out[slot], col, outA + 64 * (t + 1)));
        const unsigned uq = ua + (unsigned)PF;
        const bool colq = uq < (unsigned)lim;
        int own_req = (int)uq == len + 1 ? own_end : ownA + 64 * (t + 1 + PF);
        own_req = uq == 0u ? xold_cell : own_req;
        q_own[slot] = mem.load_issue(colq, own_req);
        q_out[slot] = mem.load_issue(colq, outA + 64 * (t + 1 + PF));
        Nw = own.w;
        rec.own_g = own.g;
        rec.own_w = own.w;
        rec.own_p = own.w * own.g;
        rec.out_w = out.w;
        rec.out_p = out.w * out.g;
        return rec;
    }
};

// Lane l of a group ends its chain at wave-step t with t + 1 = lend_l = (SKEW + 2) l + lend_0: with SKEW = 1 only every third step
// has a lane that publishes its last value.  Residue of those steps modulo 3 (wave-uniform per group); the device's unrolled loop
// compares it with a compile-time constant and skips the publish's address selects and LDS writes in the other two thirds.
template <int SIDE> SW_HD int join_turn_residue(int r0) { return (((chain_len<SIDE>(r0) - 1) % 3) + 3) % 3; } // t = lend_0 - 1 (mod 3)
SW_HD bool join_turn_of(int t, int residue) { return SKEW != 1 || ((t % 3) + 3) % 3 == residue; }

// first / last wave-step of a group (lane 0 starts its warm-up columns at step -2, loads are requested PF steps earlier)
SW_HD int group_first_step() { return -2 - PF; }
template <int SIDE> SW_HD int group_last_step(int r0, int nl) { return SKEW * (nl - 1) + chain_len<SIDE>(r0 + nl - 1) - 1; }

// ---------------------------------------------------------------------------------------------------------------------
// the corner wavefronts: AB walks (rp, rp), CD walks (R, R); three visits per ring (see the header comment).
//
// The three visits of a ring chain on each other and on the previous ring's (height(X_0) -> height(X_1) -> height(Y_0) ->
// next ring), but only through FOUR window elements; everything else in the three windows is OLD and known before the
// sweep starts, and so are the three new confidences (they depend on old confidences only).  So a corner wavefront works
// in groups of 64 rings, lane = ring:
//   prepare   every lane loads the 10 old cells of its ring, forms their products w * g and the ring's three confidences
//             -- 64 rings at once, off the critical path (old cells are only overwritten by visits that come after the
//             ring's corner visits, so they can be fetched arbitrarily early);
//   recur     ring after ring, the three dependent heights; the wavefront executes them for all lanes but only lane
//             (ring - r0) holds meaningful operands: its X_1 and Y_0 results are broadcast (v_readlane) as the next
//             ring's inner corner and inner X_1, stored and published.  ~1/5 of the instructions per ring of a lone lane
//             doing everything.
// ---------------------------------------------------------------------------------------------------------------------
template <int CD> struct CornerRing {
    // CD = 0: corner z = c - r, "outward" o = -1;  CD = 1: z = c + r, o = +1.  Cell (z + o a, z + o b): a, b = -1 inner, 0, +1 outer.
    // Old cells of ring r: rows a = -1..1, columns b = -2..1 without (-1, -1) [Y_0 of ring r-1] and (-1, -2) [X_1 of ring r-1].
    int r;              // this lane's ring
    bool live;          // r <= rings
    Cell q[3][4];       // loads in flight [a + 1][b + 2]
    int e00, e0m1;      // elements of (0, 0) and (0, -1): where this ring's results go
    float g00, g0m1;    // old heights of (0, 0) and (0, -1)
    float ow[3][4], op[3][4]; // old confidences and products w * g
    float x0w, x1w, y0w; // the three new confidences

    SW_HD static int cell_at(const Params &P, int r, int a, int b)
    {
        const int o = CD ? 1 : -1, z = P.c + o * r;
        if (r < 3) return gp_index(P.gl, z + o * a, z + o * b);
        // gp_index specialised (gp_layout.h): from ring 3 on these cells stay in the corner's quadrant, where the side follows
        // from (a, b) alone -- the first side (A / C) if a >= b, else the second (B / D) -- and the ring is r + max(a, b)
        const int ring = r + (a >= b ? a : b);
        const int g = (ring - 1) >> 6, l = (ring - 1) & 63;
        const int side = CD ? (a >= b ? 2 : 3) : (a >= b ? 0 : 1);
        const int along = a >= b ? b : a; // the coordinate that runs along the side, in outward units
        const int v = CD ? P.n - 1 - (z + along) : z - along;
        return 1 + ((side * P.gl.G + g) * P.gl.VS + v + GP_SHEAR * l) * 64 + l;
    }
    SW_HD static bool is_old(int r, int a, int b)
    {
        // ring 1 of AB: (-1, -2) is D_1(1), still old (ring 0 has no X_1)
        return !(a == -1 && b == -1) && !(a == -1 && b == -2 && !(r == 1 && !CD));
    }
    // block index of cell (z + o a, z + o b) in the 3x3 block centred at (z + o ca, z + o cb)
    SW_HD static int q9(int a, int b, int ca, int cb)
    {
        const int o = CD ? 1 : -1;
        return (o * (a - ca) + 1) + 3 * (o * (b - cb) + 1);
    }

    // ---- prepare, part 1: request the ring's old cells
    template <bool FRESH = false, class Mem> SW_HD void issue(int ring, const Params &P, Mem &mem)
    {
        live = ring <= P.rings;
        r = live ? ring : P.rings; // (idle lanes keep valid addresses)
        for (int a = -1; a <= 1; ++a)
            for (int b = -2; b <= 1; ++b) {
                int cell = cell_at(P, r, a, b);
                if constexpr (FRESH) cell = mem.bit_of(cell) ? cell : P.fresh_cell; // (a FRESH map: step_a)
                q[a + 1][b + 2] = mem.load_issue(live && is_old(r, a, b), cell);
            }
        e00 = cell_at(P, r, 0, 0);
        e0m1 = cell_at(P, r, 0, -1);
    }
    // ---- prepare, part 2: products and the three confidences
    template <class Mem> SW_HD void finish(const Params &P, Mem &mem)
    {
        for (int a = -1; a <= 1; ++a)
            for (int b = -2; b <= 1; ++b) {
                const Cell v = mem.load_value(q[a + 1][b + 2], live && is_old(r, a, b), cell_at(P, r, a, b)); // (the element only matters to the host's late loads)
                ow[a + 1][b + 2] = v.w;
                op[a + 1][b + 2] = v.w * v.g;
                if (a == 0 && b == 0) g00 = v.g;
                if (a == 0 && b == -1) g0m1 = v.g;
            }
        const bool decay0 = 2 * r * r >= P.r2min, decay1 = r * r + (r - 1) * (r - 1) >= P.r2min;
        x0w = decayed_confidence(ow[1][2], decay0, P); // X_0: (0, 0)
        x1w = decayed_confidence(ow[1][1], decay1, P); // X_1: (0, -1)
        y0w = decayed_confidence(x0w, decay0, P);      // Y_0: (0, 0) again
    }
    // ---- recur: ring r's first-side visits X_0 = (z, z), X_1 = (z, z - o) (A_1 / C_1), then the revisit Y_0 = (z, z) (B_0 / D_0),
    //      given in_corner = Y_0 of ring r - 1 at (-1, -1) (ring 0: the centre) and in_x1 = X_1 of ring r - 1 at (-1, -2).
    //      `mine`: this lane's ring is the one being computed (its results are stored and published).
    template <class Mem>
    SW_HD void recur(bool mine, int lane, WP in_corner, WP in_x1_ring, const Params &P, const LdsMap &L, Mem &mem, WP &x1_out, WP &y0_out)
    {
        // Ring 1 has no predecessor ring: for AB the cell (-1, -2) is D_1(1), still OLD; for CD it is B_1(1) = B_last(1),
        // already NEW (sides A and B of a ring come before C and D) -- the caller passes the join in that case.
        const WP in_x1 = (r == 1 && !CD) ? WP{ow[0][0], op[0][0]} : in_x1_ring;
        WP win[9];
        // ---- X_0 at (0, 0): everything old except the inner corner (-1, -1)
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b) win[q9(a, b, 0, 0)] = (a == -1 && b == -1) ? in_corner : WP{ow[a + 1][b + 2], op[a + 1][b + 2]};
        const float x0g = interpolated_height2(win, g00, ow[1][2]);
        const WP x0{x0w, x0w * x0g};
        // ---- X_1 at (0, -1): new = X_0 at (0, 0), inner corner (-1, -1), X_1(r-1) at (-1, -2)
        for (int a = -1; a <= 1; ++a)
            for (int b = -2; b <= 0; ++b)
                win[q9(a, b, 0, -1)] = (a == 0 && b == 0) ? x0 : (a == -1 && b == -1) ? in_corner : (a == -1 && b == -2) ? in_x1 : WP{ow[a + 1][b + 2], op[a + 1][b + 2]};
        const float x1g = interpolated_height2(win, g0m1, ow[1][1]);
        const WP x1{x1w, x1w * x1g};
        // ---- Y_0 at (0, 0) again: new = itself (X_0), X_1 at (0, -1), inner corner
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b)
                win[q9(a, b, 0, 0)] = (a == 0 && b == 0) ? x0 : (a == 0 && b == -1) ? x1 : (a == -1 && b == -1) ? in_corner : WP{ow[a + 1][b + 2], op[a + 1][b + 2]};
        const float y0g = interpolated_height2(win, x0g, x0w);
        const WP y0{y0w, y0w * y0g};
        mem.store(mine, e00, Cell{y0g, y0w});
        mem.store(mine, e0m1, Cell{x1g, x1w});
        const int base = L.corner + 2 * ((CD * P.c + r) * 2);
        mem.put_if(mine, lane, L, base, x1);
        mem.publish_if(mine, lane, L, base + 2, y0, L.corner_done + CD, r);
        if (!CD) // A_last(1) = A_1(1): side A of ring 1 has no chain
            mem.publish_if(mine && r == 1, lane, L, L.join + 2 * (SIDE_A * P.c + 1), x1, L.join_done + SIDE_A, 1);
        x1_out = x1;
        y0_out = y0;
    }
    // CD's first ring reads B_last(1)
    template <class Mem> SW_HD static bool ready(int ring, const LdsMap &L, Mem &mem)
    {
        return !(CD && ring == 1) || mem.counter(L.join_done + SIDE_B) >= 1;
    }
};

} // namespace sweep
} // namespace gg
