// sweep_pairb.h -- the pair sweep (sweep_pair.h: both sides of a ring hand-over in one wavefront, lanes 0..31 side X, lanes 32..63 side Y of
// 32 consecutive rings, ring r + 1 two steps behind ring r, every join a lane exchange) for THROUGHPUT launches: hundreds of clouds per
// launch, several work-groups per CU.
//
// The latency launches take everything a visit needs of OLD cells from records a preparation kernel wrote (sweep_pair.h): 5 MB per cloud
// through HBM and back -- nothing for one cloud, 10 GB for 1024.  Here a lane prepares its own visits, like sweep_core.h's: per wave-step
// one cell of the own line and one of the outer line arrive from the layer (requested PFB steps ahead, 512 contiguous bytes per half: the
// layer's shear is the pair sweep's, gp_layout.h), the visited cell's new confidence is decayed in place, the window of OLD values lives in
// registers, and what travels between lanes, wavefronts and through LDS is the (confidence, confidence x height) PAIR of a visit: the
// window's two trees (:457, :458) are one sequence of packed additions.  What the throughput launch gains over k_sweep is what the
// latency launch gains: no half steps, no closed-form wait tests, no polls on the joins -- a step is ~80 instructions where k_sweep's is
// ~300 (vector + scalar), and a launch that fills the chip is bound by instruction issue.
//
// Same schedule, same LDS protocol (tagged entries, two ring counters), same corner wavefronts (CornerLane on a CornerRec the lane builds
// from the layer itself) as sweep_pair.h; entries carry pairs.  Per-lane code for gfx950 (k4b_sweep_pair_batch.hip) and the host emulation
// (sweep_emul.hip gg_debug_emulate_pair_sweep, mode 1: layer loads can resolve as late as their use -- the worst case for the in-place
// layer's write-after-read hazards).
#pragma once

#include "sweep_pair.h"

namespace gg {
namespace sweep {
namespace pair {

#ifndef GG_PAIRB_PF
#define GG_PAIRB_PF 2
#endif
enum { PFB = GG_PAIRB_PF }; // wave-steps a layer cell is requested ahead (the queue is 4 registers per step)

// LDS image of one work-group (4-byte words): entries are (w, tag, p, tag) = 16 bytes -- each half one 8-byte unit that is written and read
// whole, so a reader that finds both tags set has both values however the 16 bytes travel --, the corner table holds plain pairs
struct LdsB {
    int cnt_corner; // [2]
    int b1;         // [4] B_1 of ring 1, an entry
    int corner;     // WP [2][c + 1][2]: AB (A_1, B_0), CD (C_1, D_0); ring 0 second entry = the centre cell
    int scratch;    // [64][4]
    int bnd;        // [pairs][2 halves][entries][4]
    int bnd_half;
    int both_pairs;
    int words;
};
SW_HD LdsB ldsb_of(int c, const Plan &pl, bool both_pairs)
{
    LdsB L;
    int o = 0;
    L.cnt_corner = o;
    o += 4;
    L.b1 = o;
    o += 4;
    L.corner = o;
    o += 2 * (c + 1) * 2 * 2;
    o = (o + 3) & ~3;
    L.scratch = o;
    o += 64 * 4;
    L.bnd = o;
    L.bnd_half = pl.groups > 0 ? pl.bnd_off[pl.groups - 1] : 0;
    L.both_pairs = both_pairs ? 1 : 0;
    o += (both_pairs ? 2 : 1) * 2 * L.bnd_half * 4;
    L.words = o;
    return L;
}
SW_HD int cornerb_word(const LdsB &L, int c, int cd, int ring, int k) { return L.corner + 2 * ((cd * (c + 1) + ring) * 2 + k); }

// :457-460 on a window given by roles (sweep_pair.h Slots): the pair's common positions, I0, the OP / I1 swap, the FLEX ends
template <int PAIR> SW_HD float window_height(bool is_x, WP S, WP N, WP U0, WP U1, WP U2, WP I0, WP I1, WP I2, WP OP, float height, float occupied)
{
    WP e[9];
    e[Slots<PAIR>::c0] = is_x ? U1 : N;
    e[Slots<PAIR>::c1] = PAIR == PAIR_AD ? S : U2;
    e[Slots<PAIR>::c2] = PAIR == PAIR_AD ? U2 : S;
    e[Slots<PAIR>::c3] = is_x ? N : U1;
    e[Slots<PAIR>::i0] = I0;
    e[Slots<PAIR>::op_x] = is_x ? OP : I1;
    e[Slots<PAIR>::i1_x] = is_x ? I1 : OP;
    e[0] = is_x ? U0 : I2;
    e[8] = is_x ? I2 : U0;
    return interpolated_height2(e, height, occupied);
}

// Memory back end (device: k4b_sweep_pair_batch.hip PairBMem; host: sweep_emul.hip PairBHostMem)
//   Cell load_issue(bool valid, int cell) / Cell load_value(const Cell &queued, bool valid, int cell)     the layer, as sweep_core.h
//   void store(bool valid, int cell, Cell v)
// the LDS traffic is the wavefront's business (waits included): PairLaneB only names the words

template <int PAIR> struct PairLaneB {
    // constants (sweep_pair.h PairLane, entries of 4 words)
    bool is_x, live, jl_lane;
    int l, r, len, lim, start;
    int ownA, outA;         // layer element of the own / outer line's cell that ARRIVES at wave-step t (along-position k0 + s + 1) = ownA / outA + 64 t
    int xold_cell, own_end; // elements of S[len + 1] (inner line, position k0 + len) and of the own line's far end (position k0 + len: another side's cell)
    int st_base;            // element of the cell visited at wave-step t = st_base + 64 t
    int a_s0, a_s1, a_pred, a_bnd, a_jl, pb, scr, cd;
    int r2c, r2r;           // (x - c)^2 + (y - c)^2 of the visited cell = r2r + (t + r2c)^2
    // state
    WP I0, I1, I2, h1, h2, U0, U1, U2, xold;
    float Sg, Sw, Sp, Ng, Nw, Np;
    Cell q_own[PFB], q_out[PFB];

    SW_HD void init(int lane, int group, const Group &G, const Params &P, const Plan &pl, const LdsB &L)
    {
        is_x = lane < (int)HALF;
        l = lane & (HALF - 1);
        const int side = is_x ? side_x(PAIR) : side_y(PAIR);
        live = l < G.nl;
        r = live ? G.r0 + l : G.r0;
        len = live ? len_of(side, r) : 0;
        lim = len > 0 ? len + 2 : 0;
        start = 2 * l + start0(PAIR, is_x);
        jl_lane = is_x && l == 0;
        const int k0 = k0_of(side);
        int x, y;
        side_xy(side, P.c, r, 0, 1, x, y);
        const int own1 = gp_index(P.gl, x, y); // own line, along-position 1 (positions of a side are 64 elements apart)
        side_xy(side, P.c, r, 1, 1, x, y);
        const int out1 = gp_index(P.gl, x, y);
        // the column of step s = t - start is along-position k0 + s + 1: element own1 + 64 (k0 + s) = own1 + 64 (k0 - start) + 64 t
        ownA = own1 + 64 * (k0 - start);
        outA = out1 + 64 * (k0 - start);
        st_base = own1 + 64 * (k0 - 1 - start);
        side_xy(side, P.c, r, -1, k0 + len, x, y);
        xold_cell = gp_index(P.gl, x < 0 ? 0 : x, y < 0 ? 0 : y);
        side_xy(side, P.c, r, 0, k0 + len, x, y);
        own_end = gp_index(P.gl, x < 0 ? 0 : x, y < 0 ? 0 : y);
        cd = (side == SIDE_A || side == SIDE_B) ? 0 : 1;
        if (side == SIDE_A) {
            a_s0 = cornerb_word(L, P.c, cd, r - 1, 1);
            a_s1 = cornerb_word(L, P.c, cd, r - 1, 0);
            a_pred = cornerb_word(L, P.c, cd, r, 0);
        } else if (side == SIDE_B) {
            a_s0 = cornerb_word(L, P.c, cd, r, 0);
            a_s1 = cornerb_word(L, P.c, cd, r - 1, 1);
            a_pred = cornerb_word(L, P.c, cd, r, 1);
        } else if (side == SIDE_C) {
            a_s0 = cornerb_word(L, P.c, cd, r - 1, 1);
            a_s1 = cornerb_word(L, P.c, cd, r - 1, 0);
            a_pred = cornerb_word(L, P.c, cd, r, 0);
        } else {
            a_s0 = cornerb_word(L, P.c, cd, r, 0);
            a_s1 = cornerb_word(L, P.c, cd, r - 1, 1);
            a_pred = cornerb_word(L, P.c, cd, r, 1);
        }
        scr = L.scratch + 4 * lane;
        const int pair_base = L.bnd + (L.both_pairs ? PAIR * 8 * L.bnd_half : 0), half_base = pair_base + (is_x ? 0 : 4 * L.bnd_half);
        a_bnd = (l == 0 && group > 0) ? half_base + 4 * (pl.bnd_off[group - 1] - start) : scr;
        a_jl = scr;
        if (jl_lane && group > 0) a_jl = pair_base + 4 * L.bnd_half + 4 * (pl.bnd_off[group - 1] + len_of(side_y(PAIR), G.r0 - 1) - 1);
        pb = (l == (int)HALF - 1 && group + 1 < pl.groups) ? half_base + 4 * (pl.bnd_off[group] - start) : -1;
        r2c = k0 - r - start; // offset of the visited cell (position k0 + s) from the side's middle: (k0 + t - start) - r
        r2r = r * r;
        I0 = I1 = I2 = h1 = h2 = U0 = U1 = U2 = xold = WP{0.f, 0.f};
        Sg = Sw = Sp = Ng = Nw = Np = 0.f;
        for (int k = 0; k < (int)PFB; ++k) q_own[k] = q_out[k] = Cell{0.f, 0.f};
    }
    SW_HD bool first_at(int t) const { return live && t == start; }
    SW_HD bool imports_at(int t, int group) const { return l == 0 && group > 0 && t - start >= 0 && t - start + 2 < len; }
    SW_HD bool join_from_lds_at(int t, int group) const { return jl_lane && group > 0 && t - start + 2 == len; }
    SW_HD int import_entry(int t) const { return a_bnd + 4 * t; }
    // which cell the own-line request of column ua = s + 2 names (the warm-up column's own cell is never used: its slot fetches S[len + 1])
    SW_HD int own_cell_of(int t, unsigned ua) const
    {
        int c = (int)ua == len + 1 ? own_end : ownA + 64 * t;
        return ua == 0u ? xold_cell : c;
    }
    // before the group's first step: the requests of the first PFB steps (t_first .. t_first + PFB - 1)
    template <class Mem> SW_HD void prime(int t_first, Mem &mem)
    {
        for (int k = 0; k < (int)PFB; ++k) {
            const int t = t_first + k;
            const unsigned ua = (unsigned)(t - start + 2);
            const bool col = ua < (unsigned)lim;
            const int slot = (((t - t_first) % (int)PFB) + (int)PFB) % (int)PFB;
            q_own[slot] = mem.load_issue(col, own_cell_of(t, ua));
            q_out[slot] = mem.load_issue(col, outA + 64 * t);
        }
    }
    // before the halves exchange their last results: a lane at its first step takes its predecessor (the corner wavefront's X_1 / Y_0)
    SW_HD void pre(bool first, WP c_pred) { h1 = first ? c_pred : h1; }
    // One wave-step, after pre() and the exchanges.  The caller (the wavefront: k4b_sweep_pair_batch.hip / sweep_emul.hip) has waited for
    // and read whatever comes from other wavefronts:
    //   slot   (t - t_first) mod PFB
    //   x_in   S[s + 2] as it reaches the lane: h2 of lane - 1, or -- first lanes of the halves, imports_at() -- the entry of the group inside
    //   j_in   the join: h1 of the partner half's lane (X lane l <- Y lane l - 1, Y lane l <- X lane l); X lane 0: the ring inside's
    //          entry (join_from_lds_at()), in group 0 the centre cell
    //   first  the lane's first step; c0, c1 = the corner table's values at a_s0, a_s1
    // RES = (t - t_first) & 3 names which of a lane's rare events the step can hold (sweep_pair.h Residue): t = RES - 2 (mod 4), side X
    // joins at t = b, ends at b + 1, side Y joins at b + 2, ends at b + 3 (b = 2 for pair A/D, 3 for B/C), lanes start at even t.
    // Returns the visit's (confidence, product): the next step's h1, and what the last lanes publish for the group outside.
    template <int RES, class Mem> SW_HD WP step(int t, int slot, WP x_in, WP j_in, bool first, WP c0, WP c1, const Params &P, Mem &mem)
    {
        constexpr int res4 = (RES + 2) & 3, x_join = PAIR == PAIR_AD ? 2 : 3, y_join = (x_join + 2) & 3;
        constexpr bool join_step = res4 == x_join || res4 == y_join, start_step = (RES & 1) == 0;
        const int s = t - start;
        const unsigned ua = (unsigned)(s + 2);
        // ---- the visited cell's new confidence (it depends on its OLD one only: the successor of the last step)
        const int ao = t + r2c;
        const float w_new = decayed_confidence(Nw, r2r + ao * ao >= P.r2min, P);
        // ---- the column that arrives now, the request for step t + PFB into its place
        const bool col = ua < (unsigned)lim;
        const Cell own = mem.load_value(q_own[slot], col, own_cell_of(t, ua));
        const Cell out = mem.load_value(q_out[slot], col, outA + 64 * t);
        const unsigned uq = ua + (unsigned)PFB;
        const bool colq = uq < (unsigned)lim;
        q_own[slot] = mem.load_issue(colq, own_cell_of(t + (int)PFB, uq));
        q_out[slot] = mem.load_issue(colq, outA + 64 * (t + (int)PFB));
        // ---- the window of OLD values moves on
        Sg = Ng, Sw = Nw, Sp = Np;
        Ng = own.g, Nw = own.w, Np = own.w * own.g;
        xold = ua == 0u ? WP{own.w, Np} : xold;
        U0 = U1;
        U1 = U2;
        U2 = WP{out.w, out.w * out.g};
        // ---- S[s + 2]
        WP x = x_in;
        if (join_step) x = s + 2 == len ? j_in : x;
        else x = s + 1 == len ? xold : x;
        I0 = I1;
        I1 = I2;
        I2 = x;
        if (start_step) {
            I0 = first ? c0 : I0;
            I1 = (first && len != 1) ? c1 : I1;
        }
        // ---- the visit
        const bool active = (unsigned)s < (unsigned)len;
        const float g = window_height<PAIR>(is_x, WP{Sw, Sp}, WP{Nw, Np}, U0, U1, U2, I0, I1, I2, h1, Sg, Sw);
        const WP res{w_new, w_new * g};
        mem.store(active, st_base + 64 * t, Cell{g, w_new});
        h2 = h1;
        h1 = res;
        return res;
    }
    SW_HD bool exports_at(int t) const { return pb >= 0 && (unsigned)(t - start) < (unsigned)len; }
    SW_HD int export_entry(int t) const { return pb + 4 * t; }
};

// A corner wavefront of a throughput launch: the lane builds the ring's CornerRec from the layer (the 64 rings of a batch at once: old
// cells are only rewritten by visits that come after the ring's corner visits), then the recurrences and the publishing of sweep_pair.h's
// CornerLane -- with the confidences beside the products.  The two cells a ring's corner visits rewrite (the corner, X_1) are read from
// the layer by nobody but the corner wavefront itself: the first ring of a batch takes the OLD confidences of the ring inside from it
// (its window sums apply the decay themselves), so a batch's cells are stored only after the NEXT batch's loads (CornerHeld).
struct CornerHeld {
    int e00, e0m1;
    float y0g, wn2, x1g, wn1;
    bool live;
    template <class Mem> SW_HD void flush(Mem &mem) const
    {
        mem.store(live, e00, Cell{y0g, wn2});
        mem.store(live, e0m1, Cell{x1g, wn1});
    }
};
template <int CD> struct CornerLaneB {
    CornerLane<CD> c;
    template <class Mem> SW_HD void init(int ring, const Params &P, Mem &mem)
    {
        const bool live = ring <= P.rings;
        const int rr = live ? ring : P.rings;
        auto load = [&](int x, int y) {
            const int cell = gp_index(P.gl, x, y);
            return mem.load_value(mem.load_issue(true, cell), true, cell);
        };
        c.init(ring, P, make_corner_rec<CD>(P, rr, load));
    }
    SW_HD CornerHeld hold(bool done) const { return CornerHeld{c.e00, c.e0m1, c.ky0g, c.R.wn[2], c.kx1g, c.R.wn[1], c.live && done}; }
    template <class Mem> SW_HD static void publish(int ring, WP x1, WP y0, const Params &P, const LdsB &L, Mem &mem)
    {
        mem.lds_put_wp(cornerb_word(L, P.c, CD, ring, 0), x1);
        mem.lds_put_wp(cornerb_word(L, P.c, CD, ring, 1), y0);
        mem.lds_set(L.cnt_corner + CD, ring);
    }
};

} // namespace pair
} // namespace sweep
} // namespace gg
