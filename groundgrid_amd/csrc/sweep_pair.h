// sweep_pair.h -- the terrain sweep (spiral_ground_interpolation / interpolate_cell, src/GroundSegmentation.cpp:398-465) for LATENCY
// launches (one or a few clouds): the two sides of a ring hand-over in ONE wavefront.
//
// sweep_core.h gives every side of the spiral its own wavefronts; the ends of the sides hand their last values to each other ring after
// ring (B_last(r) -> C's join -> C_last(r) -> B's join of ring r + 1, likewise A -> D -> A), so two wavefronts find each other through LDS
// counters twice per ring, and for one cloud that poll IS the sweep's time (~40 % of a chain wavefront's cycles).  Here a wavefront owns a
// PAIR of sides for 32 consecutive rings: lanes 0..31 walk side X (A or B), lanes 32..63 side Y (D or C) of the same rings.
//     X(r) ends at wave-step E_X(r) = 4 r + const, Y(r) two steps later, X(r + 1) two steps after that:
//     every join is the partner half's last result of the previous wave-step -- a lane exchange (v_permlane32_swap + a wave shift), no
//     publish, no poll.  Ring r + 1 follows ring r by 2 steps (its chain is 2 visits longer: 4 steps of work per ring and wavefront --
//     the four visits per ring that chain on each other are the floor of ANY schedule).
// What is left between wavefronts is feed-forward only: the 32-ring group outside reads the last ring of the group inside (its boundary
// chain, whose last value is the join) and every chain takes three corner values at its first step -- through LDS, tagged entries and
// two counters; a consumer that has caught up with its producer stays behind it, nobody waits in a cycle.
//
// Second change: nothing on the chain reads the layer.  Everything a visit needs that does not depend on NEW heights is known before the
// sweep starts -- all nine confidences of its window (a cell's new confidence is a function of its old one and its position: :463-464),
// hence gvlSum (:457), the factors (1 - occupied) and occupied * height of :460, and the products w * g of the five OLD window cells.
// A preparation kernel (k_sweep_records: one thread per visit, no dependences) leaves one 32-byte VisitRec per wave-step and lane in the
// order the chain wavefronts walk them, 64 lanes = 64 consecutive records (the OLD products arrive one window column per step, like a
// stream; the window itself is five registers); the chain adds the four NEW products to the tree (:458), divides, blends and
// multiplies: per visit 8 additions, an IEEE division, 2 multiplications and an addition -- the reference's float operations in the
// reference's order (the trees are Eigen's unrolled 3x3 reduction in the side's own column-major window order, the new confidence is
// decayed_confidence() of sweep_core.h).  The snapshot also removes every write-after-read hazard on the in-place layer: the sweep kernel
// only WRITES it.
//
// The two corners of a ring (visited twice: A_0, A_1, B_0 / C_0, C_1, D_0) stay two wavefronts of their own, ring after ring, from their
// own records (CornerRec); B_1 of ring 1 -- the one chain visit the CD corner needs -- is evaluated by the AB corner wavefront as well,
// so that the pair A/D and the pair B/C of one cloud can run in two work-groups on two CUs without a word between them (both run both
// corner wavefronts; what they store twice is bit-identical).
//
// Written once in per-lane scalar form and compiled twice, like sweep_core.h: gfx950 (k4p_sweep_pair.hip) and the host emulation
// (sweep_emul.hip gg_debug_emulate_pair_sweep, tests/test_sweep_pair_emul_cpu.py: adversarial wave interleaving against the oracle).
#pragma once

#include "sweep_core.h"

#if defined(__HIP_DEVICE_COMPILE__)
#define SW_UNROLL _Pragma("unroll")
#else
#define SW_UNROLL
#endif

namespace gg {
namespace sweep {
namespace pair {

// PTRIP: a group's wave-steps are rounded up to whole trips of the device's unrolled loop; PPF: how many wave-steps ahead the device requests
// a step's record (its queue slot is a constant of the unrolled loop: PPF divides PTRIP; smaller queues -- 6, 4, 3 -- divide it too)
enum { HALF = 32, PAIR_AD = 0, PAIR_BC = 1, MAX_GROUPS = 72 /* n <= 4600 */, PTRIP = 12, PPF = 12 };
static_assert(PTRIP % PPF == 0, "a record slot is a constant of the unrolled loop");

SW_HD constexpr int len_of(int side, int r) { return side == SIDE_A ? 2 * r - 2 : side == SIDE_D ? 2 * r : 2 * r - 1; }
SW_HD constexpr int k0_of(int side) { return (side == SIDE_A || side == SIDE_C) ? 2 : 1; }
SW_HD constexpr int side_x(int pair) { return pair == PAIR_AD ? (int)SIDE_A : (int)SIDE_B; }
SW_HD constexpr int side_y(int pair) { return pair == PAIR_AD ? (int)SIDE_D : (int)SIDE_C; }
// first wave-step of lane l of a half: 2 l + start0.  X: 0.  Y: chosen so that Y(r) ends two steps after X(r).
SW_HD constexpr int start0(int pair, bool is_x) { return (!is_x && pair == PAIR_BC) ? 2 : 0; }

// cell (row, col) of `line` (-1 inner, 0 own, +1 outer) at along-position j of ring r of a side (sweep_core.h side_cell without the layout)
SW_HD void side_xy(int side, int c, int r, int line, int j, int &x, int &y)
{
    const int rp = c - r, R = c + r;
    if (side == SIDE_A) {
        x = rp - line;
        y = rp + j;
    } else if (side == SIDE_B) {
        x = rp + j;
        y = rp - line;
    } else if (side == SIDE_C) {
        x = R + line;
        y = R - j;
    } else {
        x = R - j;
        y = R + line;
    }
}
SW_HD int tree_pos_of(int side, int line, int pos)
{
    return side == SIDE_A ? tree_pos<SIDE_A>(line, pos) : side == SIDE_B ? tree_pos<SIDE_B>(line, pos) : side == SIDE_C ? tree_pos<SIDE_C>(line, pos) : tree_pos<SIDE_D>(line, pos);
}

// ---------------------------------------------------------------------------------------------------------------------
// Where the window elements of a visit sit in Eigen's tree, per pair.  Roles: I0 I1 I2 = inner line (NEW: the ring inside), OP = the
// predecessor on the own line (NEW), S N = self and successor (OLD), U0 U1 U2 = outer line (OLD).
//   A: [U0 OP I0 U1 S I1 U2 N I2]   D: [I2 I1 I0 N S OP U2 U1 U0]      B: [U0 U1 U2 OP S N I0 I1 I2]   C: [I2 N U2 I1 S U1 I0 OP U0]
// Four tree positions hold an OLD element on both sides of a pair (COMMON), I0 sits at one place, position 0 is OLD for X and I2 for Y,
// position 8 the other way round (FLEX), and OP / I1 swap places: a lane of either half fills the tree with four selects.
// ---------------------------------------------------------------------------------------------------------------------
template <int PAIR> struct Slots {
    static constexpr int c0 = PAIR == PAIR_AD ? 3 : 1, c1 = PAIR == PAIR_AD ? 4 : 2, c2 = PAIR == PAIR_AD ? 6 : 4, c3 = PAIR == PAIR_AD ? 7 : 5;
    static constexpr int i0 = PAIR == PAIR_AD ? 2 : 6;
    static constexpr int op_x = PAIR == PAIR_AD ? 1 : 3; // OP for X, I1 for Y
    static constexpr int i1_x = PAIR == PAIR_AD ? 5 : 7; // I1 for X, OP for Y
};
static_assert(tree_pos<SIDE_A>(0, 0) == 2 && tree_pos<SIDE_D>(0, 0) == 2 && tree_pos<SIDE_B>(0, 0) == 6 && tree_pos<SIDE_C>(0, 0) == 6, "I0");
static_assert(tree_pos<SIDE_A>(1, 0) == 1 && tree_pos<SIDE_D>(0, 1) == 1 && tree_pos<SIDE_B>(1, 0) == 3 && tree_pos<SIDE_C>(0, 1) == 3, "OP of X = I1 of Y");
static_assert(tree_pos<SIDE_A>(0, 1) == 5 && tree_pos<SIDE_D>(1, 0) == 5 && tree_pos<SIDE_B>(0, 1) == 7 && tree_pos<SIDE_C>(1, 0) == 7, "I1 of X = OP of Y");
static_assert(tree_pos<SIDE_A>(0, 2) == 8 && tree_pos<SIDE_D>(0, 2) == 0 && tree_pos<SIDE_B>(0, 2) == 8 && tree_pos<SIDE_C>(0, 2) == 0, "I2: 8 for X, 0 for Y");

// What the preparation leaves per lane and wave-step (20 bytes; four steps of a lane are packed as five 16-byte words, the chain loads
// five words per four steps).  The OLD part of a window is a stream: the own line's products are the `b` of the steps themselves (self =
// this step's, successor = the next step's), the outer line's arrive one per step (nU = the cell beside the successor) and are kept for
// three steps; steps s = -2, -1 of a chain carry the two outer cells its first visit finds already there, and step s = len -- one past
// the chain -- carries what the LAST visit still needs: the own line's cell behind it (b) and, in `gvl`, the product of the inner line's
// successor cell, which is still OLD there (S[len + 1]).
struct VisitRec {
    float gvl;        // :457 gvlSum = (window confidences).sum() + FLT_MIN, every neighbour visited earlier with its NEW confidence
    float a, b;       // :460 (1 - occupied), occupied * height of the visited cell (both OLD)
    float wn;         // :464 the visited cell's new confidence
    float nU;         // product w * g of the arriving OLD cell of the outer line
};
enum { WARMUP = 2, QUAD = 4, QUAD_FLOATS = 20, QUAD_BLOCK_FLOATS = 64 * QUAD_FLOATS }; // a quad block: [chunk 0..3 = step (gvl a b wn)][chunk 4 = nU x 4] x 64 lanes x 16 bytes
SW_HD int quad_word(int step, int lane, int chunk) { return (step >> 2) * QUAD_BLOCK_FLOATS + chunk * 256 + lane * 4; } // (float index of a 16-byte word)
SW_HD VisitRec rec_at(const float *stream, int step, int lane)
{
    const float *q = stream + quad_word(step, lane, step & 3);
    VisitRec R;
    R.gvl = q[0], R.a = q[1], R.b = q[2], R.wn = q[3];
    R.nU = stream[quad_word(step, lane, 4) + (step & 3)];
    return R;
}

// the OLD products of a window (own line: self, successor; outer line: predecessor side, middle, successor side)
struct OldWindow {
    float S, N, U0, U1, U2;
};

// :458 (products).sum() in Eigen's order for a lane of either half.  COMMON positions: X (U1 . . N), Y (N . . U1) with (S, U2) between
// them for the pair A/D and (U2, S) for B/C; FLEX = U0.
template <int PAIR> SW_HD float window_sum(bool is_x, const OldWindow &O, float I0, float I1, float I2, float OP)
{
    float e[9];
    e[Slots<PAIR>::c0] = is_x ? O.U1 : O.N;
    e[Slots<PAIR>::c1] = PAIR == PAIR_AD ? O.S : O.U2;
    e[Slots<PAIR>::c2] = PAIR == PAIR_AD ? O.U2 : O.S;
    e[Slots<PAIR>::c3] = is_x ? O.N : O.U1;
    e[Slots<PAIR>::i0] = I0;
    e[Slots<PAIR>::op_x] = is_x ? OP : I1;
    e[Slots<PAIR>::i1_x] = is_x ? I1 : OP;
    e[0] = is_x ? O.U0 : I2;
    e[8] = is_x ? I2 : O.U0;
    return sw_tree9(e);
}
static_assert(tree_pos<SIDE_A>(2, 1) == Slots<PAIR_AD>::c0 && tree_pos<SIDE_A>(1, 1) == Slots<PAIR_AD>::c1 && tree_pos<SIDE_A>(2, 2) == Slots<PAIR_AD>::c2 &&
                  tree_pos<SIDE_A>(1, 2) == Slots<PAIR_AD>::c3 && tree_pos<SIDE_A>(2, 0) == 0, "A: U1 S U2 N, U0 at 0");
static_assert(tree_pos<SIDE_D>(1, 2) == Slots<PAIR_AD>::c0 && tree_pos<SIDE_D>(1, 1) == Slots<PAIR_AD>::c1 && tree_pos<SIDE_D>(2, 2) == Slots<PAIR_AD>::c2 &&
                  tree_pos<SIDE_D>(2, 1) == Slots<PAIR_AD>::c3 && tree_pos<SIDE_D>(2, 0) == 8, "D: N S U2 U1, U0 at 8");
static_assert(tree_pos<SIDE_B>(2, 1) == Slots<PAIR_BC>::c0 && tree_pos<SIDE_B>(2, 2) == Slots<PAIR_BC>::c1 && tree_pos<SIDE_B>(1, 1) == Slots<PAIR_BC>::c2 &&
                  tree_pos<SIDE_B>(1, 2) == Slots<PAIR_BC>::c3 && tree_pos<SIDE_B>(2, 0) == 0, "B: U1 U2 S N, U0 at 0");
static_assert(tree_pos<SIDE_C>(1, 2) == Slots<PAIR_BC>::c0 && tree_pos<SIDE_C>(2, 2) == Slots<PAIR_BC>::c1 && tree_pos<SIDE_C>(1, 1) == Slots<PAIR_BC>::c2 &&
                  tree_pos<SIDE_C>(2, 1) == Slots<PAIR_BC>::c3 && tree_pos<SIDE_C>(2, 0) == 8, "C: N U2 S U1, U0 at 8");
// :458 / :460 from the tree sum
SW_HD float height_of(float gvl, float a, float b, float sum)
{
#ifdef GG_PAIR_X_NODIV
    const float avg = sum * gvl; // (timing experiment: k4p_sweep_pair.hip)
#else
    const float avg = sum / gvl;
#endif
    return a * avg + b;
}

// ---------------------------------------------------------------------------------------------------------------------
// geometry of the groups and of the record stream
// ---------------------------------------------------------------------------------------------------------------------
struct Group {
    int r0, nl;          // first ring, rings (<= 32)
    int t_first, t_last; // wave-steps of the group (t_first = -2: the first lane's window warms up two steps before its first visit)
    int steps;           // t_last - t_first + 1 rounded up to whole trips
};
SW_HD Group group_of(int pair, int g, int rings)
{
    Group G;
    G.r0 = HALF * g + 1;
    G.nl = rings - (G.r0 - 1) < (int)HALF ? rings - (G.r0 - 1) : (int)HALF;
    G.t_first = -(int)WARMUP;
    G.t_last = 2 * (G.nl - 1) + start0(pair, false) + len_of(side_y(pair), G.r0 + G.nl - 1) - 1; // Y of the last ring ends last
    const int n = G.t_last - G.t_first + 2; // (+ 1: the record behind the last chain's end)
    G.steps = (n + PTRIP - 1) / PTRIP * PTRIP;
    return G;
}
SW_HD int n_groups(int rings) { return rings > 0 ? (rings + HALF - 1) / HALF : 0; }
// host-computed once per geometry, passed to the kernels by value
struct Plan {
    int groups;                      // 32-ring groups
    int base[2][MAX_GROUPS + 1];     // first wave-step record of (pair, group), in wave-steps (x 64 records)
    int total_steps;                 // wave-step records of one cloud
    int bnd_off[MAX_GROUPS + 1];     // LDS: first entry of boundary b (between group b and b + 1) inside a half's table; [groups - 1] = entries per half
};
inline Plan make_plan(int rings)
{
    Plan pl;
    pl.groups = n_groups(rings);
    if (pl.groups > (int)MAX_GROUPS) pl.groups = 0; // (the launcher falls back to sweep_core.h)
    int o = 0;
    for (int p = 0; p < 2; ++p) {
        for (int g = 0; g < pl.groups; ++g) {
            pl.base[p][g] = o;
            o += group_of(p, g, rings).steps;
        }
        pl.base[p][pl.groups] = o;
    }
    pl.total_steps = o;
    int e = 0;
    for (int b = 0; b + 1 < pl.groups; ++b) { // boundary ring 32 (b + 1): a chain of at most 2 r values
        pl.bnd_off[b] = e;
        e += 2 * HALF * (b + 1);
    }
    pl.bnd_off[pl.groups > 0 ? pl.groups - 1 : 0] = e;
    return pl;
}

// Where the height of (wave-step record `step`, lane) sits in the result stream: four consecutive steps of a lane are 16 contiguous
// bytes (the device stores them with one instruction), 64 lanes x 16 bytes a row.
SW_HD int out_slot(int step, int lane) { return (((step >> 2) * 64 + lane) << 2) + (step & 3); }

// LDS image of one work-group (4-byte words).  Boundary entries are (value, tag) pairs: tag != 0 = published.
struct Lds {
    int cnt_corner; // [2] rings finished by the AB / CD corner wavefront
    int b1;         // [2] B_1 of ring 1 (value, tag)
    int corner;     // float [2][c + 1][2]: AB (A_1, B_0), CD (C_1, D_0) products per ring; ring 0 second entry = the centre cell
    int scratch;    // [64][2] dummy entries: where lanes that publish nothing write, and what lanes that import nothing read (tag preset)
    int bnd;        // [pairs of the work-group][2 halves][entries][2]
    int bnd_half;   // entries per half
    int both_pairs; // the work-group runs both pairs (else one: two work-groups per cloud)
    int words;
};
SW_HD Lds lds_of(int c, const Plan &pl, bool both_pairs)
{
    Lds L;
    int o = 0;
    L.cnt_corner = o;
    o += 2;
    L.b1 = o;
    o += 2;
    L.corner = o;
    o += 2 * (c + 1) * 2;
    L.scratch = o;
    o += 64 * 2;
    L.bnd = o;
    L.bnd_half = pl.groups > 0 ? pl.bnd_off[pl.groups - 1] : 0;
    L.both_pairs = both_pairs ? 1 : 0;
    o += (both_pairs ? 2 : 1) * 2 * L.bnd_half * 2;
    L.words = o;
    return L;
}
SW_HD int corner_word(const Lds &L, int c, int cd, int ring, int k) { return L.corner + ((cd * (c + 1) + ring) * 2 + k); }

// ---------------------------------------------------------------------------------------------------------------------
// Preparation (no dependences: one thread per record).  `load(x, y)` = the layer's OLD (ground, confidence) of cell (row x, col y).
// ---------------------------------------------------------------------------------------------------------------------
// the confidence cell (x, y) has once the sweep has passed it: the centre is set to 1 (:405), the two diagonal corners of a ring are
// visited twice (A_0 and B_0, C_0 and D_0), every other cell of rings 1 .. c - 1 once
SW_HD float final_confidence(const Params &P, int x, int y, float w_old)
{
    const int dx = x - P.c, dy = y - P.c;
    if (dx == 0 && dy == 0) return 1.0f;
    const bool decay = dx * dx + dy * dy >= P.r2min;
    float w = decayed_confidence(w_old, decay, P);
    if (dx == dy) w = decayed_confidence(w, decay, P);
    return w;
}

// the records of steps s0 .. s0 + 3 of chain (side, ring r) as one quad (20 floats: four x (gvl a b wn), then four nU); steps outside
// [-WARMUP, len] stay zero.  The four windows share their cells: six columns of three are loaded once.
template <int SIDE, class Load> SW_HD void make_visit_quad_of(const Params &P, int r, int s0, Load load, float (&out)[QUAD_FLOATS])
{
    constexpr int side = SIDE;
    const int len = len_of(side, r), k0 = k0_of(side);
    SW_UNROLL
    for (int i = 0; i < (int)QUAD_FLOATS; ++i) out[i] = 0.f;
    // column j = along-position k0 + s0 - 1 + j of the inner / own / outer line (every index below is a constant once the loops are unrolled:
    // the arrays are registers)
    float in_w[6], in_p[6], in_f[6], own_w[6], own_p[6], own_f[6], out_w[6], out_p[6];
    SW_UNROLL
    for (int j = 0; j < 6; ++j) {
        const int pos = k0 + s0 - 1 + j;
    SW_UNROLL
        for (int line = -1; line <= 1; ++line) {
            int x, y;
            side_xy(side, P.c, r, line, pos, x, y);
            x = x < 0 ? 0 : x >= P.n ? P.n - 1 : x; // (columns beyond the chain's ends are never used; keep the loads inside the map)
            y = y < 0 ? 0 : y >= P.n ? P.n - 1 : y;
            const Cell v = load(x, y);
            const float p = v.w * v.g;
            if (line == -1) in_w[j] = v.w, in_p[j] = p, in_f[j] = final_confidence(P, x, y, v.w);
            else if (line == 0) own_w[j] = v.w, own_p[j] = p, own_f[j] = final_confidence(P, x, y, v.w);
            else out_w[j] = v.w, out_p[j] = p;
        }
    }
    SW_UNROLL
    for (int i = 0; i < (int)QUAD; ++i) {
        const int s = s0 + i; // its visited cell is column i + 1
        const bool in_range = s >= -(int)WARMUP && s <= len, visit = s >= 0 && s < len, past = s == len;
        float w[9];
        w[tree_pos<SIDE>(0, 0)] = in_f[i];
        w[tree_pos<SIDE>(0, 1)] = in_f[i + 1];
        w[tree_pos<SIDE>(0, 2)] = s == len - 1 ? in_w[i + 2] : in_f[i + 2]; // S[len + 1] belongs to a chain that has not got there yet
        w[tree_pos<SIDE>(1, 0)] = own_f[i];
        w[tree_pos<SIDE>(1, 1)] = own_w[i + 1];
        w[tree_pos<SIDE>(1, 2)] = own_w[i + 2];
        w[tree_pos<SIDE>(2, 0)] = out_w[i];
        w[tree_pos<SIDE>(2, 1)] = out_w[i + 1];
        w[tree_pos<SIDE>(2, 2)] = out_w[i + 2];
        const float gvl = sw_tree9(w) + FLT_MIN; // :457
        // (unconditional stores of selected values: the quad stays in registers on the device)
        out[4 * i + 0] = visit ? gvl : past ? in_p[i + 1] : 0.f; // ... one past the chain: what the last visit needs of its successor column
        out[4 * i + 1] = visit ? 1.0f - own_w[i + 1] : 0.f;       // :460
        out[4 * i + 2] = (visit || past) ? own_p[i + 1] : 0.f;
        out[4 * i + 3] = visit ? own_f[i + 1] : 0.f;               // :463-464 (a chain cell is visited once: its final confidence)
        out[16 + i] = in_range ? out_p[i + 2] : 0.f;               // nU: the outer line's cell beside the successor
    }
}
template <class Load> SW_HD void make_visit_quad(const Params &P, int pair, bool is_x, int r, int s0, Load load, float (&out)[QUAD_FLOATS])
{
    if (pair == PAIR_AD) {
        if (is_x) make_visit_quad_of<SIDE_A>(P, r, s0, load, out);
        else make_visit_quad_of<SIDE_D>(P, r, s0, load, out);
    } else {
        if (is_x) make_visit_quad_of<SIDE_B>(P, r, s0, load, out);
        else make_visit_quad_of<SIDE_C>(P, r, s0, load, out);
    }
}

// The three corner visits of ring r: X_0 = (z, z), X_1 = (z, z - o), Y_0 = (z, z) again (z = c - r, o = -1 for AB; z = c + r, o = +1 for
// CD).  Cell (a, b) below = (z + o a, z + o b).  NEW in their windows: the inner corner (-1, -1) [Y_0 of ring r - 1], X_1 of ring r - 1
// at (-1, -2), and the ring's own earlier visits.
struct CornerRec {
    float gvl[3], a[3], b[2], wn[3]; // per visit (b of the revisit is the first visit's product: not known here)
    float o0[8], o1[6], o2[6];       // OLD products in increasing tree position
    float oin;                       // the product of (-1, -2) as an OLD cell (ring 1 of AB, where it is D_1(1))
};
enum { CORNER_REC_FLOATS = 32 };
static_assert(sizeof(CornerRec) == 4 * CORNER_REC_FLOATS, "one float plane per field");
template <int CD> struct CornerSlots {
    static constexpr int q9(int a, int b, int ca, int cb) { return CD ? (a - ca + 1) + 3 * (b - cb + 1) : (ca - a + 1) + 3 * (cb - b + 1); }
    static constexpr int x0_in = q9(-1, -1, 0, 0);                                                      // visit X_0: the inner corner
    static constexpr int x1_x0 = q9(0, 0, 0, -1), x1_in = q9(-1, -1, 0, -1), x1_x1 = q9(-1, -2, 0, -1); // visit X_1: X_0, inner corner, inner X_1
    static constexpr int y0_x0 = q9(0, 0, 0, 0), y0_x1 = q9(0, -1, 0, 0), y0_in = q9(-1, -1, 0, 0);     // visit Y_0: itself (X_0), X_1, inner corner
};
template <int CD, class Load> SW_HD CornerRec make_corner_rec(const Params &P, int r, Load load)
{
    const int o = CD ? 1 : -1, z = P.c + o * r;
    CornerRec R;
    R.oin = 0.f;
    const bool decay0 = 2 * r * r >= P.r2min;
    const Cell c00 = load(z, z);
    const float x0w = decayed_confidence(c00.w, decay0, P);
    SW_UNROLL
    for (int v = 0; v < 3; ++v) {
        const int ca = 0, cb = v == 1 ? -1 : 0;
        float w[9];
        int n_old = 0;
        SW_UNROLL
        for (int q = 0; q < 9; ++q) { // increasing tree position: (a, b) of position q around (ca, cb)
            const int da = q % 3 - 1, db = q / 3 - 1;
            const int a = CD ? ca + da : ca - da, b = CD ? cb + db : cb - db;
            const int x = z + o * a, y = z + o * b;
            const Cell cv = load(x, y);
            bool from_chain; // the element's product reaches the visit through the recurrence, not through the record
            float wq;
            if (a == 0 && b == 0) { // the corner cell itself: OLD for X_0, once decayed for X_1 and Y_0
                from_chain = v != 0;
                wq = v == 0 ? cv.w : x0w;
            } else if (a == -1 && b == -1) { // the corner of the ring inside (the centre for ring 1)
                from_chain = true;
                wq = final_confidence(P, x, y, cv.w);
            } else if (a == -1 && b == -2) { // X_1 of the ring inside (X_1's window only); ring 1: AB finds D_1(1) there, still OLD -- CD
                from_chain = true;           // finds B_1(1), NEW
                const bool is_old = r == 1 && !CD;
                wq = is_old ? cv.w : final_confidence(P, x, y, cv.w);
                R.oin = cv.w * cv.g;
            } else if (a == 0 && b == -1 && v == 2) { // X_1 of this ring in the revisit's window
                from_chain = true;
                wq = final_confidence(P, x, y, cv.w);
            } else {
                from_chain = false;
                wq = cv.w;
            }
            w[q] = wq;
            if (!from_chain) {
                const float p = cv.w * cv.g;
                if (v == 0) R.o0[n_old] = p;
                else if (v == 1) R.o1[n_old] = p;
                else R.o2[n_old] = p;
                ++n_old;
            }
        }
        R.gvl[v] = sw_tree9(w) + FLT_MIN;
    }
    const int x1x = z, x1y = z - o;
    const Cell c0m1 = load(x1x, x1y);
    const int dx1 = x1x - P.c, dy1 = x1y - P.c;
    R.a[0] = 1.0f - c00.w;
    R.b[0] = c00.w * c00.g;
    R.wn[0] = x0w;
    R.a[1] = 1.0f - c0m1.w;
    R.b[1] = c0m1.w * c0m1.g;
    R.wn[1] = decayed_confidence(c0m1.w, dx1 * dx1 + dy1 * dy1 >= P.r2min, P);
    R.a[2] = 1.0f - x0w; // the revisit: occupied = the confidence the first visit left, height = the height it left
    R.wn[2] = decayed_confidence(x0w, decay0, P);
    return R;
}

// ---------------------------------------------------------------------------------------------------------------------
// Memory back end of the chain / corner code (device: k4p_sweep_pair.hip PairMem; host: sweep_emul.hip PairHostMem)
//   float lds_f(int word)                 read a float another wavefront may have written
//   void  lds_put(int word, float v)  /  lds_put2(int word, float v0, float v1)   (two consecutive words, 8-byte aligned, one write)
//   void  lds_entry(int word, float v)    (v, tag 1) as ONE 8-byte write
//   int   lds_i(int word)  /  void lds_set(int word, int v)
//   void  store(bool valid, int cell, Cell v)     the layer (fire and forget; the corner wavefronts only)
//   void  emit(int slot, float g)                 the chain's height of wave-step t, lane: slot = (record block of the step) * 64 + lane
// ---------------------------------------------------------------------------------------------------------------------

// one lane of a pair wavefront
template <int PAIR> struct PairLane {
    // constants
    bool is_x, live, jl_lane; // half; the lane has a ring; X lane 0: its join comes from LDS (the group inside, or the centre for ring 1)
    int l, r, len, start;
    int st_base;            // layer element of the cell visited at wave-step t = st_base + 64 t (the emulation holds finish_cell's inverse map to it)
    int out_step0, lane_;   // the height of wave-step t goes to out_slot(out_step0 + t, lane_) of the result stream
    int a_s0, a_s1, a_pred; // LDS words of S[0], S[1] and the first predecessor (corner table)
    int a_bnd;              // lane 0 of a half, group > 0: (value, tag) of wave-step t = a_bnd + 2 t while the step imports (others: never)
    int a_jl;               // X lane 0: LDS entry (value, tag) of the join
    int pb;                 // last lane of a half when a group follows: wave-step t's result goes to entry pb + 2 t (others: -1)
    int scr;                // the lane's scratch entry (tag preset)
    int cd;                 // which corner wavefront the lane's first step waits for
    // state: the inner line S[s], S[s+1], S[s+2]; the last two results.  h1 is the predecessor's product of the next visit, the join the
    // partner lane takes one step after this chain's end, and -- a step later, as h2 -- what lane l + 1 reads as its S[s + 2]
    float I0, I1, I2, h1, h2;
    float U1, U2; // the outer line's products of the last two steps (the window's OLD part: VisitRec)

    SW_HD void init(int lane, int group, const Group &G, const Params &P, const Plan &pl, const Lds &L)
    {
        is_x = lane < (int)HALF;
        l = lane & (HALF - 1);
        const int side = is_x ? side_x(PAIR) : side_y(PAIR);
        live = l < G.nl;
        r = live ? G.r0 + l : G.r0;
        len = live ? len_of(side, r) : 0;
        start = 2 * l + start0(PAIR, is_x);
        jl_lane = is_x && l == 0;
        int x, y;
        side_xy(side, P.c, r, 0, 1, x, y);
        st_base = gp_index(P.gl, x, y) + 64 * (k0_of(side) - 1 - start);
        out_step0 = pl.base[PAIR][group] - G.t_first;
        lane_ = lane;
        cd = (side == SIDE_A || side == SIDE_B) ? 0 : 1;
        if (side == SIDE_A) { // S[0] = B_0(r-1), S[1] = A_1(r-1), predecessor A_1(r)
            a_s0 = corner_word(L, P.c, cd, r - 1, 1);
            a_s1 = corner_word(L, P.c, cd, r - 1, 0);
            a_pred = corner_word(L, P.c, cd, r, 0);
        } else if (side == SIDE_B) { // A_1(r), B_0(r-1), predecessor B_0(r)
            a_s0 = corner_word(L, P.c, cd, r, 0);
            a_s1 = corner_word(L, P.c, cd, r - 1, 1);
            a_pred = corner_word(L, P.c, cd, r, 1);
        } else if (side == SIDE_C) { // D_0(r-1), C_1(r-1), predecessor C_1(r)
            a_s0 = corner_word(L, P.c, cd, r - 1, 1);
            a_s1 = corner_word(L, P.c, cd, r - 1, 0);
            a_pred = corner_word(L, P.c, cd, r, 0);
        } else { // C_1(r), D_0(r-1), predecessor D_0(r)
            a_s0 = corner_word(L, P.c, cd, r, 0);
            a_s1 = corner_word(L, P.c, cd, r - 1, 1);
            a_pred = corner_word(L, P.c, cd, r, 1);
        }
        scr = L.scratch + 2 * lane;
        const int pair_base = L.bnd + (L.both_pairs ? PAIR * 4 * L.bnd_half : 0), half_base = pair_base + (is_x ? 0 : 2 * L.bnd_half);
        // import: lane 0 at step s = t - start reads entry s of the boundary inside
        a_bnd = (l == 0 && group > 0) ? half_base + 2 * (pl.bnd_off[group - 1] - start) : scr;
        // X lane 0: the join = the last value of Y of the ring inside (group 0: ring 0 = the centre cell, which sits in the corner table)
        a_jl = scr;
        if (jl_lane && group > 0) a_jl = pair_base + 2 * L.bnd_half + 2 * (pl.bnd_off[group - 1] + len_of(side_y(PAIR), G.r0 - 1) - 1);
        // export: the last lane of a half, when a group follows
        pb = (l == (int)HALF - 1 && group + 1 < pl.groups) ? half_base + 2 * (pl.bnd_off[group] - start) : -1;
        I0 = I1 = I2 = h1 = h2 = 0.f;
        U1 = U2 = 0.f;
    }
    // what wave-step t of this lane reads from other wavefronts (the wavefront may run the step once all of it is there)
    SW_HD bool first_at(int t) const { return live && t == start; } // (ring 1 of side A has no chain, but its "first step" still takes A_1(1): D's join)
    SW_HD bool imports_at(int t, int group) const { return l == 0 && group > 0 && t - start >= 0 && t - start + 2 < len; }
    SW_HD bool join_from_lds_at(int t, int group) const { return jl_lane && group > 0 && t - start + 2 == len; }
    SW_HD int import_entry(int t) const { return a_bnd + 2 * t; }

    // 1. the first step takes the predecessor from the corner table -- BEFORE the wavefront exchanges h1 (see first_at)
    template <class Mem> SW_HD void pre(int t, Mem &mem)
    {
        if (first_at(t)) h1 = mem.lds_f(a_pred);
    }
    // 2. the visit.  R, Rn = the records of this and of the next wave-step; x_prev = h2 of lane - 1, j_perm = h1 of the partner lane (X l <- Y
    //    l - 1, Y l <- X l), both as they were after pre(); centre_p = the centre cell's product (the join of ring 1 of side B)
    template <class Mem> SW_HD void step(int t, int group, const VisitRec &R, const VisitRec &Rn, float x_prev, float j_perm, float centre_p, Mem &mem)
    {
        const int s = t - start;
        float x = x_prev;                                             // S[s + 2]: step s of the ring inside (lane - 1, two steps ago) ...
        if (imports_at(t, group)) x = mem.lds_f(import_entry(t));     // ... which for lane 0 is the last lane of the group inside
        float j = j_perm;
        if (jl_lane) j = group > 0 ? mem.lds_f(a_jl) : centre_p;
        x = s + 2 == len ? j : x;    // the join
        x = s + 1 == len ? Rn.gvl : x; // an OLD cell at the far end (it travels in the record behind the chain's last)
        I0 = I1;
        I1 = I2;
        I2 = x;
        if (first_at(t)) {
            I0 = mem.lds_f(a_s0);
            if (len != 1) I1 = mem.lds_f(a_s1); // (a chain of one visit: S[1] is the join, taken a step ago)
        }
        const bool active = (unsigned)s < (unsigned)len;
        const OldWindow O{R.b, Rn.b, U1, U2, R.nU};
        U1 = U2;
        U2 = R.nU;
        const float g = height_of(R.gvl, R.a, R.b, window_sum<PAIR>(is_x, O, I0, I1, I2, h1));
        const float res = R.wn * g;
#if !defined(__HIP_DEVICE_COMPILE__)
        if (getenv("GG_PAIR_DBG") && active && r == atoi(getenv("GG_PAIR_DBG")) && s < 3) fprintf(stderr, "pair %d x %d r %d s %d t %d: I %g %g %g pred %g | gvl %g a %g b %g wn %g o %g %g %g %g %g xo %g -> g %g\n", PAIR, (int)is_x, r, s, t, I0, I1, I2, h1, R.gvl, R.a, R.b, R.wn, O.S, O.N, O.U0, O.U1, O.U2, Rn.gvl, g);
#endif
        // the height goes to the result stream, 64 lanes = 64 consecutive floats (the lanes' cells lie in 64 different lines of the layer: a
        // scattered store costs the CU's memory front end more than the rest of the step); finish_cell() puts it into the layer
        if (active) mem.emit(out_slot(out_step0 + t, lane_), g);
        h2 = h1;
        h1 = res;
        if (pb >= 0 && active) mem.lds_entry(pb + 2 * t, res);
    }
};

// The chain visit that rewrites cell (x, y), if one does (the centre, the cells of the corner wavefronts -- the two diagonal corners and the
// second cell of sides A and C -- and the three border lines are not chain cells): where its height sits in the result stream.
SW_HD bool chain_slot_of_cell(const Params &P, const Plan &pl, int x, int y, int &slot)
{
    const int dx = x - P.c, dy = y - P.c;
    const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy, r = ax > ay ? ax : ay;
    if (r == 0 || r > P.rings) return false;
    const int rp = P.c - r, R = P.c + r;
    int side, k;
    if (dx == -r && dy < r) {
        side = SIDE_A;
        k = y - rp;
    } else if (dx == r) {
        side = SIDE_C;
        k = R - y;
    } else if (dy == -r) {
        side = SIDE_B;
        k = x - rp;
    } else {
        side = SIDE_D;
        k = R - x;
    }
    const int s = k - k0_of(side);
    if (s < 0 || s >= len_of(side, r)) return false;
    const int pair = (side == SIDE_A || side == SIDE_D) ? (int)PAIR_AD : (int)PAIR_BC;
    const bool is_x = side == SIDE_A || side == SIDE_B;
    const int g = (r - 1) / HALF, l = (r - 1) % HALF;
    const int t = s + 2 * l + start0(pair, is_x);
    slot = out_slot(pl.base[pair][g] + t + (int)WARMUP, l + (is_x ? 0 : (int)HALF)); // (t_first = -WARMUP)
    return true;
}
// ... and what the cell holds after the sweep: the streamed height, its own decayed confidence (:463-464; a chain cell is visited once)
SW_HD Cell finished_cell(const Params &P, int x, int y, float w_old, float g_new)
{
    const int dx = x - P.c, dy = y - P.c;
    return Cell{g_new, decayed_confidence(w_old, dx * dx + dy * dy >= P.r2min, P)};
}

// one lane of a corner wavefront: lane = ring within a batch of 64, the three dependent visits run ring after ring (every lane executes
// them with its own record, the lane whose ring it is holds the meaningful operands)
template <int CD> struct CornerLane {
    int r;
    bool live;
    CornerRec R;
    int e00, e0m1; // layer elements of (0, 0) and (0, -1): where this ring's results go
    SW_HD void init(int ring, const Params &P, const CornerRec &rec)
    {
        live = ring <= P.rings;
        r = live ? ring : P.rings;
        R = rec;
        const int o = CD ? 1 : -1, z = P.c + o * r;
        e00 = gp_index(P.gl, z, z);
        e0m1 = gp_index(P.gl, z, z - o);
    }
    template <int N_OLD> SW_HD static float tree_with(const float (&old)[N_OLD], int q0, float v0, int q1, float v1, int q2, float v2)
    {
        float e[9];
        int k = 0;
        for (int q = 0; q < 9; ++q) {
            if (q == q0) e[q] = v0;
            else if (q == q1) e[q] = v1;
            else if (q == q2) e[q] = v2;
            else e[q] = old[k++];
        }
        return sw_tree9(e);
    }
    // ring r's three visits, given in_corner = Y_0 of ring r - 1 (ring 0: the centre) and in_x1 = X_1 of ring r - 1 (ring 1 of CD: B_1(1)):
    // heights and products of X_1 and of the revisit Y_0
    SW_HD void visits(float in_corner, float in_x1_ring, float &x1g, float &x1p, float &y0g, float &y0p) const
    {
        using S = CornerSlots<CD>;
        const float in_x1 = (r == 1 && !CD) ? R.oin : in_x1_ring;
        const float x0g = height_of(R.gvl[0], R.a[0], R.b[0], tree_with<8>(R.o0, S::x0_in, in_corner, -1, 0.f, -1, 0.f));
        const float x0p = R.wn[0] * x0g;
        x1g = height_of(R.gvl[1], R.a[1], R.b[1], tree_with<6>(R.o1, S::x1_x0, x0p, S::x1_in, in_corner, S::x1_x1, in_x1));
        x1p = R.wn[1] * x1g;
        // the revisit: occupied * height = the first visit's confidence times the height it left = its product, the same two floats
        y0g = height_of(R.gvl[2], R.a[2], x0p, tree_with<6>(R.o2, S::y0_x0, x0p, S::y0_x1, x1p, S::y0_in, in_corner));
        y0p = R.wn[2] * y0g;
    }
    // the two cells of the ring: kept by the lane whose ring it is, stored once per batch of rings (nothing reads the layer meanwhile)
    float kx1g = 0.f, ky0g = 0.f;
    SW_HD void keep(bool mine, float x1g, float y0g)
    {
        kx1g = mine ? x1g : kx1g;
        ky0g = mine ? y0g : ky0g;
    }
    template <class Mem> SW_HD void flush(bool done, Mem &mem) const
    {
        mem.store(live && done, e00, Cell{ky0g, R.wn[2]});
        mem.store(live && done, e0m1, Cell{kx1g, R.wn[1]});
    }
    // what the chains of ring r (and r + 1) take at their first steps: the two products, then the ring count (one wavefront's LDS
    // operations execute in order)
    template <class Mem> SW_HD static void publish(int ring, float x1p, float y0p, const Params &P, const Lds &L, Mem &mem)
    {
        mem.lds_put2(corner_word(L, P.c, CD, ring, 0), x1p, y0p);
        mem.lds_set(L.cnt_corner + CD, ring);
    }
};
// B_1 of ring 1 from the AB corner's results (A_1(1), B_0(1)) and the record of that visit (pair B/C, group 0, lane 0, wave-step 0)
// (W2, W1, R, Rn: that lane's records of wave-steps -2 .. 1)
SW_HD float b1_of_ring1(const VisitRec &W2, const VisitRec &W1, const VisitRec &R, const VisitRec &Rn, float a1p, float b0p, float centre_p)
{
    const OldWindow O{R.b, Rn.b, W2.nU, W1.nU, R.nU};
    const float g = height_of(R.gvl, R.a, R.b, window_sum<PAIR_BC>(true, O, a1p, centre_p, Rn.gvl, b0p)); // (a chain of one visit: S[2] is the OLD far end)
    return R.wn * g;
}

} // namespace pair
} // namespace sweep
} // namespace gg
