// sweep_emul.hip -- host-side lock-step emulation of the ring sweep (sweep_core.h): the SAME per-lane code the gfx950
// kernel runs, executed wavefront by wavefront on the CPU with an arbitrary (seeded, adversarial) interleaving of the
// wavefronts and, optionally, every layer load resolved as late as its use.  tests/test_sweep_emul_cpu.py compares the
// result with the oracle's serial sweep: the dataflow (who hands what to whom, when it is ready) is proven without a GPU.
// Host code only; compiled with the library's flags (-ffp-contract=off), so the floats are the device's.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>
#include <cstdlib>
#include <algorithm>

#include "groundgrid_hip.h"
#include "sweep_core.h"

namespace {

using namespace gg::sweep;

struct HostMem { // one per emulated work-group ("part"): its own LDS, the layer and the exchange region shared with the others
    Cell *gp2;
    const std::vector<uint64_t> *bits = nullptr; // FRESH maps (gg_internal.h Arena::gp_bits): bit e - 1 = element e is in memory; null: a written map
    bool bit_of(int cell) const { return cell <= 0 || (((*bits)[(size_t)(cell - 1) >> 6] >> ((cell - 1) & 63)) & 1ull) != 0ull; }
    std::vector<int32_t> lds;
    bool late;
    long loads = 0, stores = 0, lds_ops = 0;
    std::vector<uint64_t> *xchg = nullptr; // sweep_core.h "Parts": tagged values between work-groups
    uint32_t seq = 1;
    void export_wp_if(bool c, int entry, WP v)
    {
        if (!c) return;
        uint32_t w, p;
        memcpy(&w, &v.w, 4);
        memcpy(&p, &v.p, 4);
        (*xchg)[2 * (size_t)entry] = ((uint64_t)seq << 32) | w;
        (*xchg)[2 * (size_t)entry + 1] = ((uint64_t)seq << 32) | p;
    }
    bool import_wp(int entry, WP &v)
    {
        const uint64_t a = (*xchg)[2 * (size_t)entry], b = (*xchg)[2 * (size_t)entry + 1];
        const uint32_t w = (uint32_t)a, p = (uint32_t)b;
        memcpy(&v.w, &w, 4);
        memcpy(&v.p, &p, 4);
        return (uint32_t)(a >> 32) == seq && (uint32_t)(b >> 32) == seq;
    }
    void set_counter(int word, int value) { lds[(size_t)word] = value; }
    void ring_put(int word, const PrepRec &r)
    {
        ++lds_ops;
        memcpy(&lds[(size_t)word], &r, sizeof r);
    }
    PrepRec ring_get(int word)
    {
        ++lds_ops;
        PrepRec r;
        memcpy(&r, &lds[(size_t)word], sizeof r);
        return r;
    }
    Cell load_issue(bool valid, int cell)
    {
        if (!valid) return Cell{0.f, 0.f};
        ++loads;
        return gp2[cell];
    }
    Cell load_value(const Cell &queued, bool valid, int cell) { return (late && valid) ? gp2[cell] : queued; }
    Cell fresh(const Cell &v) { return v; }
    void mark(int) {}
    void store(bool valid, int cell, Cell v)
    {
        if (!valid) return;
        ++stores;
        gp2[cell] = v;
    }
    int counter(int word) { return lds[(size_t)word]; }
    void counters3(int w0, int w1, int w2, int &v0, int &v1, int &v2)
    {
        v0 = counter(w0);
        v1 = counter(w1);
        v2 = counter(w2);
    }
    void put(int word, WP v)
    {
        ++lds_ops;
        memcpy(&lds[(size_t)word], &v, sizeof v);
    }
    WP get(int word)
    {
        ++lds_ops;
        WP v;
        memcpy(&v, &lds[(size_t)word], sizeof v);
        return v;
    }
    void publish(int data_word, WP v, int counter_word, int value)
    {
        put(data_word, v);
        lds[(size_t)counter_word] = value;
    }
    void put_if(bool c, int, const LdsMap &, int word, WP v)
    {
        if (c) put(word, v);
    }
    void publish_if(bool c, int, const LdsMap &, int data_word, WP v, int counter_word, int value)
    {
        if (c) publish(data_word, v, counter_word, value);
    }
};

struct WaveBase {
    virtual ~WaveBase() {}
    virtual bool bad() const { return false; }
    virtual bool done() const = 0;
    virtual bool try_step() = 0; // false: stalled on another wavefront (or done)
};

template <int SIDE> struct ChainWave : WaveBase {
    const Params &P;
    const LdsMap &L;
    HostMem &mem;
    int g1; // the part's groups end here
    int wave, group, t, t_last, r0, nl;
    ChainLane<SIDE> lane[LANES];
    ChainSync<SIDE> sync;
    bool advanced = false;
    long steps = 0;
    bool plan_mismatch = false;
    ChainWave(const Params &p, const LdsMap &l, HostMem &m, int w, int g0_, int g1_) : P(p), L(l), mem(m), g1(g1_), wave(w), group(g0_ + w - p.waves_per_side) { next_group(); }
    void next_group()
    {
        group += P.waves_per_side;
        if (group >= g1) return;
        r0 = LANES * group + 1;
        nl = P.rings - (r0 - 1) < LANES ? P.rings - (r0 - 1) : LANES;
        for (int l = 0; l < LANES; ++l) {
            lane[l].init(l, r0, nl, group, P, L);
            // the stride-64 addressing of a lane's two lines is the layout's promise: hold it to gp_index() cell by cell
            const ChainLane<SIDE> &c = lane[l];
            const int k0 = chain_k0<SIDE>();
            for (int j = k0; j <= k0 + c.len - 1; ++j) // visited cell of step s = j - k0 is element ownA + 64 t, t = l3 + s
                if (c.ownA + 64 * (c.l3 + j - k0) != side_cell<SIDE>(P, c.r, 0, j)) plan_mismatch = true;
            for (int j = k0 - 1; c.len > 0 && j <= k0 + c.len; ++j) // column of step s = j - k0 - 1 is element outA + 64 (t + 1)
                if (c.outA + 64 * (c.l3 + j - k0) != side_cell<SIDE>(P, c.r, 1, j)) plan_mismatch = true;
            if (l > 0 && l < nl && c.len > 0 && lane[l - 1].len > 0 && c.ownA != lane[l - 1].ownA + 1) plan_mismatch = true; // one wave-step = 64 consecutive elements
        }
        if (mem.bits)
            for (int l = 0; l < LANES; ++l) {
                lane[l].xold_bit = mem.bit_of(lane[l].xold_cell);
                lane[l].end_bit = mem.bit_of(lane[l].own_end);
            }
        t = group_first_step();
        t_last = group_last_step<SIDE>(r0, nl);
        t_last += (TRIP - (t_last - t + 1) % TRIP) % TRIP; // the device runs whole trips of TRIP steps
        sync.init(r0, nl, group, P, L);
        advanced = false;
    }
    bool done() const override { return group >= g1; }
    bool bad() const override { return plan_mismatch; }
    bool half = false; // step_a of wave-step t is done, step_b waits for the join
    bool try_step() override
    {
        if (done()) return false;
        const int tmod = ((t % SKEW) + SKEW) % SKEW;
        if (!half) {
            if (!advanced) { // requirements of step t, once
                sync.advance(t);
                advanced = true;
            }
            if (sync.ok_a() != sync.slow_ok_a()) plan_mismatch = true; // (the one-compare test is the closed forms' inverse)
            if (!sync.ok_a()) {
                sync.refresh(mem);
                if (sync.ok_a() != sync.slow_ok_a()) plan_mismatch = true;
                if (!sync.ok_a()) return false;
            }
            // the cached counters are lower bounds of what the step reads: hold the lazily polled protocol to the exact needs
            for (int l = 0; l < nl; ++l) {
                const ChainLane<SIDE> &c = lane[l];
                if (c.len > 0 && t - c.l3 == 0 && mem.counter(sync.w_corner) < c.r) plan_mismatch = true;
            }
            if (group > 0 && t >= 0 && t + 2 < lane[0].len && mem.counter(sync.w_bnd) < t + 1) plan_mismatch = true;
            const int step_no = t - group_first_step(); // (split steps: position in the ring's step count)
            if (P.split_steps && mem.counter(L.prep_done + SIDE) <= step_no) return false; // the preparing wavefront has not got there yet
            WP x_in[LANES];
            for (int l = 0; l < LANES; ++l) x_in[l] = l ? lane[l - 1].handed_over() : WP{0.f, 0.f}; // wave shift right by one, before anybody moves
            const int slot = ((t % PF) + PF) % PF;
            if (P.split_steps) {
                for (int l = 0; l < LANES; ++l)
                    lane[l].take(t, tmod, mem.ring_get(L.prep + ((SIDE * PREP_DEPTH + (step_no % PREP_DEPTH)) * LANES + l) * PREP_WORDS), x_in[l], group > 0, mem);
                mem.set_counter(L.take_done + SIDE, step_no + 1);
            } else if (mem.bits) { // a FRESH map (k4_sweep.hip run_chain<FRESH>): the bits of the two cells this step REQUESTS on its lines
                for (int l = 0; l < LANES; ++l) {
                    const int eo = lane[l].ownA + 64 * (t + 1 + (int)PF), eu = lane[l].outA + 64 * (t + 1 + (int)PF);
                    const bool in_o = eo >= 1 && eo < P.gl.elems, in_u = eu >= 1 && eu < P.gl.elems; // (outside: requests of idle steps)
                    lane[l].template step_a<true, 1, true>(t, slot, tmod, x_in[l], P, L, group > 0, mem, in_o && mem.bit_of(eo), in_u && mem.bit_of(eu));
                }
            } else {
                for (int l = 0; l < LANES; ++l) lane[l].step_a(t, slot, tmod, x_in[l], P, L, group > 0, mem);
            }
            half = true;
        }
        if (sync.ok_b() != sync.slow_ok_b()) plan_mismatch = true;
        if (!sync.ok_b()) {
            sync.refresh(mem);
            if (sync.ok_b() != sync.slow_ok_b()) plan_mismatch = true;
            if (!sync.ok_b()) return false; // (other wavefronts run between the two halves)
        }
        for (int l = 0; l < nl; ++l) {
            const ChainLane<SIDE> &c = lane[l];
            const int s_ = t - c.l3;
            if (c.len > 0 && s_ == c.len - 2 && s_ >= -1 && mem.counter(sync.w_join) < ((SIDE == SIDE_A || SIDE == SIDE_B) ? c.r - 1 : c.r)) plan_mismatch = true;
        }
        for (int l = 0; l < LANES; ++l) lane[l].step_b(t, tmod, P, L, group + 1 < P.groups, group, mem, join_turn_of(t, join_turn_residue<SIDE>(r0)));
        half = false;
        ++steps;
        advanced = false;
        if (++t > t_last) next_group();
        return true;
    }
};

template <int CD> struct CornerWave : WaveBase {
    const Params &P;
    const LdsMap &L;
    HostMem &mem;
    int g0, g1;
    int r;                 // next ring of the recurrence
    int r_end;             // last ring of the part
    bool prepared = false; // the current group's lanes have their old cells
    CornerRing<CD> lane[LANES];
    WP in_corner{0.f, 0.f}, in_x1{0.f, 0.f};
    CornerWave(const Params &p, const LdsMap &l, HostMem &m, int g0_, int g1_)
        : P(p), L(l), mem(m), g0(g0_), g1(g1_), r(LANES * g0_ + 1), r_end(LANES * g1_ < p.rings ? LANES * g1_ : p.rings) {}
    bool done() const override { return r > r_end; }
    bool try_step() override
    {
        if (done() || !CornerRing<CD>::ready(r, L, mem)) return false;
        const int l = (r - 1) % LANES, r0 = r - l;
        if (!prepared) { // like the device: the whole group's old cells, requested and consumed at the start of the group
            if (g0 > 0 && r0 == LANES * g0 + 1 && mem.counter(L.corner_done + CD) < r0 - 1) return false; // the importer has not delivered the ring before the part yet
            for (int k = 0; k < LANES; ++k) {
                if (mem.bits) lane[k].template issue<true>(r0 + k, P, mem);
                else lane[k].issue(r0 + k, P, mem);
            }
            for (int k = 0; k < LANES; ++k) lane[k].finish(P, mem);
            const int prev = L.corner + 2 * ((CD * P.c + r0 - 1) * 2);
            in_corner = mem.get(prev + 2);
            in_x1 = r0 > 1 ? mem.get(prev) : WP{0.f, 0.f};
            prepared = true;
        }
        if (CD && r == 1) in_x1 = mem.get(L.join + 2 * (SIDE_B * P.c + 1));
        WP x1, y0;
        lane[l].recur(true, l, in_corner, in_x1, P, L, mem, x1, y0);
        in_corner = y0;
        in_x1 = x1;
        ++r;
        if (l == LANES - 1) prepared = false;
        return true;
    }
};

// the preparing wavefront of one side (split steps, k4_sweep.hip run_prep): one try = one wave-step, if the ring has room
template <int SIDE> struct PrepWave : WaveBase {
    const Params &P;
    const LdsMap &L;
    HostMem &mem;
    int group, r0, nl, t, t_last;
    PrepLane<SIDE> lane[LANES];
    PrepWave(const Params &p, const LdsMap &l, HostMem &m, int g) : P(p), L(l), mem(m), group(g)
    {
        r0 = LANES * group + 1;
        nl = P.rings - (r0 - 1) < LANES ? P.rings - (r0 - 1) : LANES;
        for (int k = 0; k < LANES; ++k) lane[k].init(k, r0, nl, P);
        t = group_first_step();
        t_last = group_last_step<SIDE>(r0, nl);
        t_last += (TRIP - (t_last - t + 1) % TRIP) % TRIP; // (whole trips, like the chain wavefront)
    }
    bool done() const override { return t > t_last; }
    bool try_step() override
    {
        if (done()) return false;
        const int step_no = t - group_first_step();
        if (step_no - mem.counter(L.take_done + SIDE) >= PREP_DEPTH) return false; // the ring is full
        const int slot = ((t % PF) + PF) % PF;
        for (int k = 0; k < LANES; ++k)
            mem.ring_put(L.prep + ((SIDE * PREP_DEPTH + (step_no % PREP_DEPTH)) * LANES + k) * PREP_WORDS, lane[k].step(t, slot, P, mem));
        mem.set_counter(L.prep_done + SIDE, step_no + 1);
        ++t;
        return true;
    }
};

// the exporter wavefront of a part with a successor (k4_sweep.hip run_export): one try = whatever has been published since
struct ExportWave : WaveBase {
    const Params &P;
    const LdsMap &L;
    HostMem &mem;
    int g1;
    bool head = false, corners = false;
    int sent[4] = {0, 0, 0, 0}, len[4];
    ExportWave(const Params &p, const LdsMap &l, HostMem &m, int g1_) : P(p), L(l), mem(m), g1(g1_)
    {
        const int rb = LANES * g1;
        len[0] = chain_len<SIDE_A>(rb), len[1] = chain_len<SIDE_B>(rb), len[2] = chain_len<SIDE_C>(rb), len[3] = chain_len<SIDE_D>(rb);
    }
    bool done() const override { return head && corners && sent[0] >= len[0] && sent[1] >= len[1] && sent[2] >= len[2] && sent[3] >= len[3]; }
    bool try_step() override
    {
        const int rb = LANES * g1;
        bool progress = false;
        for (int side = 0; side < 4; ++side) {
            const int avail = mem.counter(L.bnd_done + side * P.groups + (g1 - 1));
            for (; sent[side] < avail; ++sent[side], progress = true)
                mem.export_wp_if(true, xchg_chain(g1, side, sent[side]), mem.get(L.bnd + 2 * (side * L.bnd_stride + bnd_offset(g1 - 1) - L.bnd_base + sent[side])));
        }
        if (!corners && mem.counter(L.corner_done + 0) >= rb && mem.counter(L.corner_done + 1) >= rb) {
            for (int k = 0; k < 4; ++k) mem.export_wp_if(true, xchg_misc(g1, k), mem.get(L.corner + 2 * (((k >> 1) * P.c + rb) * 2) + 2 * (k & 1)));
            corners = true;
            progress = true;
        }
        if (!head && mem.counter(L.join_done + SIDE_C) >= rb && mem.counter(L.join_done + SIDE_D) >= rb) {
            for (int k = 0; k < 2; ++k) mem.export_wp_if(true, xchg_misc(g1, X_JOIN_C + k), mem.get(L.join + 2 * ((k == 0 ? (int)SIDE_C : (int)SIDE_D) * P.c + rb)));
            head = true;
            progress = true;
        }
        return progress;
    }
};

// the importer wavefront of a part p > 0 (k4_sweep.hip run_import): one try = whatever has arrived
struct ImportWave : WaveBase {
    const Params &P;
    const LdsMap &L;
    HostMem &mem;
    int g0;
    bool head = false, joins = false;
    int have[4] = {0, 0, 0, 0}, len[4];
    ImportWave(const Params &p, const LdsMap &l, HostMem &m, int g0_) : P(p), L(l), mem(m), g0(g0_)
    {
        const int rb = LANES * g0;
        len[0] = chain_len<SIDE_A>(rb), len[1] = chain_len<SIDE_B>(rb), len[2] = chain_len<SIDE_C>(rb), len[3] = chain_len<SIDE_D>(rb);
    }
    bool done() const override { return head && joins && have[0] >= len[0] && have[1] >= len[1] && have[2] >= len[2] && have[3] >= len[3]; }
    bool try_step() override
    {
        const int rb = LANES * g0;
        if (!head) { // the corner values first: nothing of this part can start without them
            WP v[4];
            for (int k = 0; k < 4; ++k)
                if (!mem.import_wp(xchg_misc(g0, k), v[k])) return false;
            for (int k = 0; k < 4; ++k) mem.put(L.corner + 2 * (((k >> 1) * P.c + rb) * 2) + 2 * (k & 1), v[k]);
            mem.set_counter(L.corner_done + 0, rb);
            mem.set_counter(L.corner_done + 1, rb);
            head = true;
            return true;
        }
        bool progress = false;
        if (!joins) {
            WP c, d;
            if (mem.import_wp(xchg_misc(g0, X_JOIN_C), c) && mem.import_wp(xchg_misc(g0, X_JOIN_D), d)) {
                mem.put(L.join + 2 * (SIDE_C * P.c + rb), c);
                mem.put(L.join + 2 * (SIDE_D * P.c + rb), d);
                mem.set_counter(L.join_done + SIDE_C, rb);
                mem.set_counter(L.join_done + SIDE_D, rb);
                joins = true;
                progress = true;
            }
        }
        for (int side = 0; side < 4; ++side) {
            int nv = 0;
            WP v;
            while (nv < LANES && have[side] + nv < len[side] && mem.import_wp(xchg_chain(g0, side, have[side] + nv), v)) {
                mem.put(L.bnd + 2 * (side * L.bnd_stride + bnd_offset(g0 - 1) - L.bnd_base + have[side] + nv), v);
                ++nv;
            }
            if (nv) {
                have[side] += nv;
                mem.set_counter(L.bnd_done + side * P.groups + (g0 - 1), have[side]);
                progress = true;
            }
        }
        return progress;
    }
};

} // namespace

namespace gg {
namespace sweep {

Params make_params(int n, double resolution, float min_dist_squared, double decrease)
{
    Params P;
    P.n = n;
    P.c = n / 2 - 1;
    P.rings = P.c - 1 > 0 ? P.c - 1 : 0;
    P.groups = (P.rings + LANES - 1) / LANES;
    // Throughput setting (many clouds per launch): few wavefronts per cloud, so that two clouds share a CU.  launch_sweep raises it
    // to one wavefront per group (at most 3 per side: 14 wavefronts) when the launch has fewer clouds than the chip has CUs:
    // with SKEW = 1 group g + 1 starts only 64 steps after group g, and a wavefront that still works on group g - 1 delays it.
    P.waves_per_side = P.groups <= 1 ? 1 : P.groups <= 3 ? 2 : 3;
    P.split_steps = 0;
    P.poll_cap = 1 << 22;
    P.debug_fault = 0;
    P.keep_points = 0;
    P.fresh_cell = 0;
    P.gpw = P.groups > 1 ? P.groups : 1; // one work-group unless the launcher (or the emulation's GG_SWEEP_GPW) cuts the map into parts
    if (getenv("GG_SWEEP_WAVES")) P.waves_per_side = std::max(1, std::min(std::min(P.groups, 3), atoi(getenv("GG_SWEEP_WAVES")))); // (the host emulation: tests/test_sweep_emul_cpu.py)
    // :463 (pow((float)x - center, 2.0) + pow((float)y - center, 2.0)) * pow(resolution, 2.0f) > minDistSquared: the left side is a
    // non-decreasing function of the integer (x-c)^2 + (y-c)^2, so the test is an integer threshold
    int r2 = 0;
    while (r2 < 2 * n * n && !((double)r2 * (resolution * resolution) > (double)min_dist_squared)) ++r2;
    P.r2min = r2;
    P.decrease = decrease;
    P.inv_decrease = 1.0 / decrease;
    P.decay_fast = decrease >= 1.25 && decrease < 1e300;
    P.gl = make_gp_layout(n);
    return P;
}

} // namespace sweep
} // namespace gg

namespace {
// ChainSync's one-compare wait test against the closed forms it inverts (sweep_core.h cover()): every side and ring group of an n x n
// map, every value each of the three counters can have (the other two complete), every wave-step of the group and a trip beyond
template <int SIDE> long sync_mismatches(const Params &P)
{
    long bad = 0;
    const LdsMap L = lds_layout(P.c, P.groups);
    for (int group = 0; group < std::max(P.groups, 1); ++group) {
        const int r0 = LANES * group + 1, nl = std::min(P.rings - (r0 - 1), (int)LANES);
        if (nl <= 0) break;
        const int t_last = group_last_step<SIDE>(r0, nl) + TRIP, top = r0 + nl + 2, bnd_top = chain_len<SIDE>(r0) + 2;
        for (int which = 0; which < 3; ++which)
            for (int have = 0; have <= (which == 2 ? bnd_top : top); ++have) {
                ChainSync<SIDE> sy;
                sy.init(r0, nl, group, P, L);
                sy.have_corner = which == 0 ? have : top;
                sy.have_join = which == 1 ? have : top;
                sy.have_bnd = which == 2 ? have : bnd_top;
                sy.cover();
                for (int t = group_first_step(); t <= t_last; ++t) {
                    sy.advance(t);
                    bad += sy.ok_a() != sy.slow_ok_a();
                    bad += sy.ok_b() != sy.slow_ok_b();
                }
            }
    }
    return bad;
}
} // namespace

extern "C" long gg_debug_sweep_sync_selftest(int n)
{
    if (n < 8) return -1;
    const Params P = gg::sweep::make_params(n, 0.33, 12.0f, 10.0);
    return sync_mismatches<SIDE_A>(P) + sync_mismatches<SIDE_B>(P) + sync_mismatches<SIDE_C>(P) + sync_mismatches<SIDE_D>(P);
}

// patched (nullable): a FRESH map (gg_internal.h Arena::gp_bits) -- n x n bytes, column-major like gp2: which cells are in memory (what k_patch
// wrote); every other cell holds (fresh_ground, 1e-7) by definition, and its memory is POISON (NaN): a read of it spoils the result
static int emulate_ring_sweep(int n, double resolution, float min_dist_squared, float *gp2, float base_z, double decrease, unsigned seed, int late_loads, long *stats,
                              const unsigned char *patched, float fresh_ground)
{
    if (n < 8 || !gp2) return GG_ERR_INVALID;
    Params P = gg::sweep::make_params(n, resolution, min_dist_squared, decrease);
    P.fresh_cell = 1 + (P.gl.VS - 1) * 64; // (gg_create: a padding element)
    if (getenv("GG_SWEEP_GPW")) { // emulate the multi-work-group sweep (sweep_core.h "Parts")
        P.gpw = std::max(1, std::min(atoi(getenv("GG_SWEEP_GPW")), std::max(P.groups, 1)));
        P.waves_per_side = std::max(1, std::min(P.waves_per_side, P.gpw));
        if (getenv("GG_SWEEP_SPLIT") && atoi(getenv("GG_SWEEP_SPLIT")) && P.gpw == 1) P.split_steps = 1; // + a preparing wavefront per side
    }
    const int n_groups = std::max(P.groups, 1), n_parts = (n_groups + P.gpw - 1) / P.gpw;
    // the layer in the device's sheared element order (gp_layout.h); gp2 is Eigen-style column-major on both ends
    std::vector<Cell> sheared((size_t)P.gl.elems, Cell{0.f, 0.f});
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) sheared[(size_t)gp_index(P.gl, row, col)] = Cell{gp2[2 * ((size_t)row + (size_t)col * n)], gp2[2 * ((size_t)row + (size_t)col * n) + 1]};
    std::vector<uint64_t> bits;
    if (patched) {
        if (late_loads) return GG_ERR_INVALID; // (the late-load check re-reads a request's CELL, not the element the fresh step asked for)
        bits.assign((size_t)P.gl.elems / 64 + 2, 0ull);
        const float poison = __builtin_nanf("");
        for (int col = 0; col < n; ++col)
            for (int row = 0; row < n; ++row) {
                const int e = gp_index(P.gl, row, col);
                if (patched[(size_t)row + (size_t)col * n] && e > 0) bits[(size_t)(e - 1) >> 6] |= 1ull << ((e - 1) & 63);
                else sheared[(size_t)e] = Cell{poison, poison};
            }
        sheared[(size_t)P.fresh_cell] = Cell{fresh_ground, (float)0.0000001};
        // k_sweep<FRESH>'s first lines: the cells no sweep visits take the reset's pair unless they are marked
        for (int col = 0; col < n; ++col)
            for (int row = 0; row < n; ++row)
                if (std::max(std::abs(row - P.c), std::abs(col - P.c)) >= P.c && !patched[(size_t)row + (size_t)col * n])
                    sheared[(size_t)gp_index(P.gl, row, col)] = sheared[(size_t)P.fresh_cell];
    }
    std::vector<uint64_t> xchg(2 * (size_t)std::max(xchg_entries(P.groups), 1), 0ull);
    std::vector<LdsMap> maps((size_t)n_parts);
    std::vector<HostMem> mems((size_t)n_parts);
    // :405-411 centre cell; ring 0 of every hand-over table is the centre
    sheared[(size_t)gp_index(P.gl, P.c, P.c)] = Cell{base_z, 1.0f};
    const WP centre{1.0f, 1.0f * base_z};
    std::vector<WaveBase *> waves;
    size_t lds_words = 0;
    for (int part = 0; part < n_parts; ++part) {
        const int g0 = part * P.gpw, g1 = std::min(g0 + P.gpw, P.groups);
        LdsMap &L = maps[(size_t)part];
        HostMem &mem = mems[(size_t)part];
        L = lds_layout(P.c, P.groups, g0, g1, P.split_steps != 0);
        lds_words = std::max(lds_words, (size_t)L.words);
        mem.gp2 = sheared.data();
        mem.bits = patched ? &bits : nullptr;
        mem.lds.assign((size_t)L.words, 0);
        mem.late = late_loads != 0;
        mem.xchg = &xchg;
        mem.seq = 7;
        for (int side = 0; side < 2; ++side) mem.put(L.corner + 2 * ((side * P.c + 0) * 2) + 2, centre);
        mem.put(L.join + 2 * (SIDE_C * P.c + 0), centre);
        mem.put(L.join + 2 * (SIDE_D * P.c + 0), centre);
        for (int w = 0; w < P.waves_per_side; ++w) waves.push_back(new ChainWave<SIDE_A>(P, L, mem, w, g0, g1));
        for (int w = 0; w < P.waves_per_side; ++w) waves.push_back(new ChainWave<SIDE_B>(P, L, mem, w, g0, g1));
        for (int w = 0; w < P.waves_per_side; ++w) waves.push_back(new ChainWave<SIDE_C>(P, L, mem, w, g0, g1));
        for (int w = 0; w < P.waves_per_side; ++w) waves.push_back(new ChainWave<SIDE_D>(P, L, mem, w, g0, g1));
        if (P.split_steps) {
            waves.push_back(new PrepWave<SIDE_A>(P, L, mem, g0));
            waves.push_back(new PrepWave<SIDE_B>(P, L, mem, g0));
            waves.push_back(new PrepWave<SIDE_C>(P, L, mem, g0));
            waves.push_back(new PrepWave<SIDE_D>(P, L, mem, g0));
        }
        waves.push_back(new CornerWave<0>(P, L, mem, g0, g1));
        waves.push_back(new CornerWave<1>(P, L, mem, g0, g1));
        if (part > 0) waves.push_back(new ImportWave(P, L, mem, g0));
        if (g1 < P.groups) waves.push_back(new ExportWave(P, L, mem, g1));
    }

    // wave scheduling: seed 0 = round robin; otherwise a seeded random walk in which a chosen wavefront runs a random burst
    // (up to "as far as it can") before the next one is chosen: wavefronts drift apart as far as the dataflow allows
    uint32_t rng = seed * 2654435761u + 12345u;
    auto rnd = [&]() { return rng = rng * 1664525u + 1013904223u; };
    long total_steps = 0, stalls = 0, rounds = 0;
    int rc = GG_OK;
    for (;;) {
        bool all_done = true, progress = false;
        for (auto *w : waves) all_done &= w->done();
        if (all_done) break;
        ++rounds;
        if (seed == 0) {
            for (auto *w : waves)
                if (w->try_step()) {
                    progress = true;
                    ++total_steps;
                } else if (!w->done())
                    ++stalls;
        } else {
            for (int tries = 0; tries < 4 * (int)waves.size() && !progress; ++tries) {
                WaveBase *w = waves[(rnd() >> 8) % waves.size()];
                const uint32_t mode = (rnd() >> 8) % 8;
                int burst = mode == 0 ? 1 << 30 : mode < 4 ? 1 + (int)((rnd() >> 8) % 64) : 1;
                while (burst-- > 0 && w->try_step()) {
                    progress = true;
                    ++total_steps;
                }
                if (!progress && !w->done()) ++stalls;
            }
            if (!progress) // the random picks were all stalled: look at everybody before calling it a deadlock
                for (auto *w : waves)
                    if (w->try_step()) {
                        progress = true;
                        ++total_steps;
                        break;
                    }
        }
        if (!progress) {
            rc = -10; // deadlock: every unfinished wavefront waits for another one
            break;
        }
    }
    if (getenv("GG_EMUL_VERBOSE")) fprintf(stderr, "ring sweep emulation: n %d, %ld scheduler rounds, %ld wave-steps, %ld stalls\n", n, rounds, total_steps, stalls);
    if (stats) {
        stats[0] = total_steps;
        stats[1] = stalls;
        stats[2] = stats[3] = stats[4] = 0;
        for (const HostMem &m : mems) {
            stats[2] += m.loads;
            stats[3] += m.stores;
            stats[4] += m.lds_ops;
        }
        stats[5] = (long)lds_words * 4;
        stats[6] = (P.waves_per_side * 4 + 2) * n_parts + 2 * (n_parts - 1);
        stats[7] = P.r2min;
    }
    for (auto *w : waves) {
        if (w->bad() && rc == GG_OK) rc = -11; // the stepped plan disagrees with the closed form
        delete w;
    }
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) {
            const Cell v = sheared[(size_t)gp_index(P.gl, row, col)];
            gp2[2 * ((size_t)row + (size_t)col * n)] = v.g;
            gp2[2 * ((size_t)row + (size_t)col * n) + 1] = v.w;
        }
    return rc;
}

extern "C" int gg_debug_emulate_ring_sweep(int n, double resolution, float min_dist_squared, float *gp2, float base_z, double decrease,
                                           unsigned seed, int late_loads, long *stats)
{
    return emulate_ring_sweep(n, resolution, min_dist_squared, gp2, base_z, decrease, seed, late_loads, stats, nullptr, 0.0f);
}
// the same on a FRESH map: gp2 holds the values of the cells marked in `patched` (anything elsewhere: those cells are (fresh_ground, 1e-7) by
// definition and their memory is poisoned); on return every cell is real
extern "C" int gg_debug_emulate_ring_sweep_fresh(int n, double resolution, float min_dist_squared, float *gp2, const unsigned char *patched, float fresh_ground, float base_z,
                                                 double decrease, unsigned seed, long *stats)
{
    if (!patched) return GG_ERR_INVALID;
    return emulate_ring_sweep(n, resolution, min_dist_squared, gp2, base_z, decrease, seed, 0, stats, patched, fresh_ground);
}

// =====================================================================================================================
// The pair sweep (sweep_pair.h) under the same kind of lock-step emulation: the preparation (records), then pair wavefronts
// (lanes 0..31 side X, 32..63 side Y of 32 rings) and the two corner wavefronts of one or two work-groups, interleaved adversarially.
// =====================================================================================================================
#include "sweep_pair.h"

namespace {
namespace gp = gg::sweep::pair;

struct PairHostMem { // one per emulated work-group
    Cell *layer;
    std::vector<float> *out; // the chains' result stream (shared by the work-groups of a cloud)
    void emit(int slot, float g)
    {
        ++stores;
        (*out)[(size_t)slot] = g;
    }
    std::vector<int32_t> lds;
    long stores = 0, lds_ops = 0;
    bool fault = false; // a read of an entry nobody has published
    float lds_f(int word)
    {
        ++lds_ops;
        float v;
        memcpy(&v, &lds[(size_t)word], 4);
        return v;
    }
    void lds_put(int word, float v)
    {
        ++lds_ops;
        memcpy(&lds[(size_t)word], &v, 4);
    }
    void lds_put2(int word, float v0, float v1)
    {
        if (word & 1) fault = true; // (one 8-byte write on the device)
        lds_put(word, v0);
        lds_put(word + 1, v1);
    }
    void lds_entry(int word, float v)
    {
        lds_put(word, v);
        lds[(size_t)word + 1] = 1;
    }
    int lds_i(int word) { return lds[(size_t)word]; }
    void lds_set(int word, int v) { lds[(size_t)word] = v; }
    void store(bool valid, int cell, Cell v)
    {
        if (!valid) return;
        ++stores;
        layer[cell] = v;
    }
};

struct Records {
    std::vector<float> visit;          // quad blocks (sweep_pair.h quad_word), total_steps / 4 + 1 of them
    std::vector<gp::CornerRec> corner; // [2][rings + 1]
};

template <int PAIR> struct PairWave : WaveBase {
    const Params &P;
    const gp::Plan &pl;
    const gp::Lds &L;
    PairHostMem &mem;
    const Records &rec;
    float centre_p;
    int W, group, t, t_end;
    gp::Group G;
    gp::PairLane<PAIR> lane[64];
    bool plan_mismatch = false;
    PairWave(const Params &p, const gp::Plan &pl_, const gp::Lds &l, PairHostMem &m, const Records &r, float cp, int w, int W_)
        : P(p), pl(pl_), L(l), mem(m), rec(r), centre_p(cp), W(W_), group(w - W_)
    {
        next_group();
    }
    void next_group()
    {
        group += W;
        if (group >= pl.groups) return;
        G = gp::group_of(PAIR, group, P.rings);
        for (int k = 0; k < 64; ++k) {
            lane[k].init(k, group, G, P, pl, L);
            // the stride-64 store addressing against gp_index(), visit by visit
            const gp::PairLane<PAIR> &c = lane[k];
            const int side = c.is_x ? gp::side_x(PAIR) : gp::side_y(PAIR);
            for (int s = 0; s < c.len; ++s) {
                int x, y;
                gp::side_xy(side, P.c, c.r, 0, gp::k0_of(side) + s, x, y);
                if (c.st_base + 64 * (c.start + s) != gp_index(P.gl, x, y)) plan_mismatch = true;
                int slot = -1;
                if (!gp::chain_slot_of_cell(P, pl, x, y, slot) || slot != gp::out_slot(c.out_step0 + c.start + s, c.lane_)) plan_mismatch = true; // the finish finds this visit's height
            }
        }
        t = G.t_first;
        t_end = G.t_first + G.steps; // whole trips, like the device
    }
    bool done() const override { return group >= pl.groups; }
    bool bad() const override { return plan_mismatch || mem.fault; }
    bool try_step() override
    {
        if (done()) return false;
        // what the step reads from other wavefronts must be there
        for (int k = 0; k < 64; ++k) {
            const gp::PairLane<PAIR> &c = lane[k];
            if (c.first_at(t) && mem.lds_i(L.cnt_corner + c.cd) < c.r) return false;
            if (c.imports_at(t, group) && mem.lds_i(c.import_entry(t) + 1) == 0) return false;
            if (c.join_from_lds_at(t, group) && mem.lds_i(c.a_jl + 1) == 0) return false;
        }
        for (int k = 0; k < 64; ++k) lane[k].pre(t, mem);
        float x_prev[64], j_perm[64];
        for (int k = 0; k < 64; ++k) {
            x_prev[k] = k ? lane[k - 1].h2 : 0.f;                                  // wave shift right by one (lane 32 gets side X's last lane: never used)
            j_perm[k] = k < 32 ? (k ? lane[32 + k - 1].h1 : 0.f) : lane[k - 32].h1; // X l <- Y l - 1, Y l <- X l
        }
        const int step = pl.base[PAIR][group] + (t - G.t_first);
        for (int k = 0; k < 64; ++k) lane[k].step(t, group, gp::rec_at(rec.visit.data(), step, k), gp::rec_at(rec.visit.data(), step + 1, k), x_prev[k], j_perm[k], centre_p, mem);
        if (++t >= t_end) next_group();
        return true;
    }
};

template <int CD> struct PairCornerWave : WaveBase {
    const Params &P;
    const gp::Plan &pl;
    const gp::Lds &L;
    PairHostMem &mem;
    const Records &rec;
    float centre_p;
    int r = 1;
    gp::CornerLane<CD> lane[64];
    float in_corner, in_x1 = 0.f;
    PairCornerWave(const Params &p, const gp::Plan &pl_, const gp::Lds &l, PairHostMem &m, const Records &rc, float cp) : P(p), pl(pl_), L(l), mem(m), rec(rc), centre_p(cp), in_corner(cp) {}
    bool done() const override { return r > P.rings; }
    bool try_step() override
    {
        if (done()) return false;
        if (CD && r == 1) {
            if (mem.lds_i(L.b1 + 1) == 0) return false; // B_1(1) from the AB corner wavefront
            in_x1 = mem.lds_f(L.b1);
        }
        const int k = (r - 1) % 64;
        if (k == 0)
            for (int j = 0; j < 64; ++j) {
                const int ring = r + j;
                lane[j].init(ring, P, rec.corner[(size_t)CD * (P.rings + 1) + (ring <= P.rings ? ring : P.rings)]);
            }
        float x1g, x1, y0g, y0;
        lane[k].visits(in_corner, in_x1, x1g, x1, y0g, y0);
        lane[k].keep(true, x1g, y0g);
        lane[k].flush(true, mem);
        gp::CornerLane<CD>::publish(r, x1, y0, P, L, mem);
        if (!CD && r == 1) { // the one chain visit the other corner needs
            const int st0 = pl.base[gp::PAIR_BC][0] + (0 - gp::group_of(gp::PAIR_BC, 0, P.rings).t_first); // lane 0's wave-step 0
            const float *v = rec.visit.data();
            mem.lds_entry(L.b1, gp::b1_of_ring1(gp::rec_at(v, st0 - 2, 0), gp::rec_at(v, st0 - 1, 0), gp::rec_at(v, st0, 0), gp::rec_at(v, st0 + 1, 0), x1, y0, centre_p));
        }
        in_corner = y0;
        in_x1 = x1;
        ++r;
        return true;
    }
};

} // namespace

extern "C" int gg_debug_emulate_pair_sweep(int n, double resolution, float min_dist_squared, float *gp2, float base_z, double decrease, unsigned seed, int n_wgs,
                                           int waves_per_pair, long *stats)
{
    if (n < 8 || !gp2) return GG_ERR_INVALID;
    const Params P = gg::sweep::make_params(n, resolution, min_dist_squared, decrease);
    const gp::Plan pl = gp::make_plan(P.rings);
    if (pl.groups <= 0) return GG_ERR_INVALID;
    std::vector<Cell> sheared((size_t)P.gl.elems, Cell{0.f, 0.f});
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) sheared[(size_t)gp_index(P.gl, row, col)] = Cell{gp2[2 * ((size_t)row + (size_t)col * n)], gp2[2 * ((size_t)row + (size_t)col * n) + 1]};
    // ---- the preparation: every record from the OLD layer
    Records rec;
    const float poison = __builtin_nanf("");
    rec.visit.assign(((size_t)pl.total_steps / 4 + 1) * gp::QUAD_BLOCK_FLOATS, poison);
    auto load = [&](int x, int y) { return sheared[(size_t)gp_index(P.gl, x, y)]; };
    long n_visits = 0;
    for (int p = 0; p < 2; ++p)
        for (int g = 0; g < pl.groups; ++g) {
            const gp::Group G = gp::group_of(p, g, P.rings);
            for (int t0 = G.t_first; t0 < G.t_first + G.steps; t0 += gp::QUAD)
                for (int lane = 0; lane < 64; ++lane) {
                    const bool is_x = lane < 32;
                    const int l = lane & 31, side = is_x ? gp::side_x(p) : gp::side_y(p);
                    if (l >= G.nl) continue; // (a lane without a ring: its records stay poison)
                    const int r = G.r0 + l, len = gp::len_of(side, r), s0 = t0 - (2 * l + gp::start0(p, is_x));
                    if (s0 + 3 < -(int)gp::WARMUP || s0 > len) continue;
                    float q[gp::QUAD_FLOATS];
                    gp::make_visit_quad(P, p, is_x, r, s0, load, q);
                    const int step = pl.base[p][g] + (t0 - G.t_first);
                    for (int i = 0; i < 4; ++i) {
                        const int s = s0 + i;
                        if (s < -(int)gp::WARMUP || s > len) continue; // (steps outside the chain: poison, like the device's untouched memory)
                        float *w = rec.visit.data() + gp::quad_word(step + i, lane, i);
                        w[0] = q[4 * i], w[1] = q[4 * i + 1], w[2] = q[4 * i + 2], w[3] = q[4 * i + 3];
                        rec.visit[(size_t)gp::quad_word(step + i, lane, 4) + i] = q[16 + i];
                        n_visits += s >= 0 && s < len;
                    }
                }
        }
    rec.corner.resize(2 * (size_t)(P.rings + 1));
    for (int r = 1; r <= P.rings; ++r) {
        rec.corner[(size_t)r] = gp::make_corner_rec<0>(P, r, load);
        rec.corner[(size_t)(P.rings + 1) + r] = gp::make_corner_rec<1>(P, r, load);
    }
    // ---- the sweep
    sheared[(size_t)gp_index(P.gl, P.c, P.c)] = Cell{base_z, 1.0f}; // :405-411
    const float centre_p = 1.0f * base_z;
    std::vector<float> out((size_t)pl.total_steps * 64, poison);
    if (n_wgs != 2) n_wgs = 1;
    const gp::Lds L = gp::lds_of(P.c, pl, n_wgs == 1);
    const int W = std::max(1, std::min(waves_per_pair > 0 ? waves_per_pair : pl.groups, pl.groups));
    std::vector<PairHostMem> mems((size_t)n_wgs);
    std::vector<WaveBase *> waves;
    for (int wg = 0; wg < n_wgs; ++wg) {
        PairHostMem &mem = mems[(size_t)wg];
        mem.layer = sheared.data();
        mem.out = &out;
        mem.lds.assign((size_t)L.words, 0);
        for (int k = 0; k < 64; ++k) mem.lds[(size_t)L.scratch + 2 * k + 1] = 1;
        for (int cd = 0; cd < 2; ++cd) mem.lds_put(gp::corner_word(L, P.c, cd, 0, 1), centre_p);
        for (int w = 0; w < W; ++w) {
            if (n_wgs == 1 || wg == 0) waves.push_back(new PairWave<gp::PAIR_AD>(P, pl, L, mem, rec, centre_p, w, W));
            if (n_wgs == 1 || wg == 1) waves.push_back(new PairWave<gp::PAIR_BC>(P, pl, L, mem, rec, centre_p, w, W));
        }
        waves.push_back(new PairCornerWave<0>(P, pl, L, mem, rec, centre_p));
        waves.push_back(new PairCornerWave<1>(P, pl, L, mem, rec, centre_p));
    }
    uint32_t rng = seed * 2654435761u + 12345u;
    auto rnd = [&]() { return rng = rng * 1664525u + 1013904223u; };
    long total_steps = 0, stalls = 0;
    int rc = GG_OK;
    for (;;) {
        bool all_done = true, progress = false;
        for (auto *w : waves) all_done &= w->done();
        if (all_done) break;
        if (seed == 0) {
            for (auto *w : waves)
                if (w->try_step()) {
                    progress = true;
                    ++total_steps;
                } else if (!w->done())
                    ++stalls;
        } else {
            for (int tries = 0; tries < 4 * (int)waves.size() && !progress; ++tries) {
                WaveBase *w = waves[(rnd() >> 8) % waves.size()];
                const uint32_t mode = (rnd() >> 8) % 8;
                int burst = mode == 0 ? 1 << 30 : mode < 4 ? 1 + (int)((rnd() >> 8) % 64) : 1;
                while (burst-- > 0 && w->try_step()) {
                    progress = true;
                    ++total_steps;
                }
                if (!progress && !w->done()) ++stalls;
            }
            if (!progress)
                for (auto *w : waves)
                    if (w->try_step()) {
                        progress = true;
                        ++total_steps;
                        break;
                    }
        }
        if (!progress) {
            rc = -10; // deadlock
            break;
        }
    }
    // ---- the finish: the streamed heights into the layer, with the cells' own new confidences (element by element, like the device:
    //      the layout's inverse map must name every cell exactly once)
    long finished = 0, cells_seen = 0;
    for (int e = 0; e < P.gl.elems; ++e) {
        int x, y, slot;
        if (!gg::gp_cell_of(P.gl, e, x, y)) continue;
        ++cells_seen;
        if (!gp::chain_slot_of_cell(P, pl, x, y, slot)) continue;
        sheared[(size_t)e] = gp::finished_cell(P, x, y, sheared[(size_t)e].w, out[(size_t)slot]);
        ++finished;
    }
    if ((finished != n_visits || cells_seen != (long)n * n) && rc == GG_OK) rc = -12;
    if (stats) {
        stats[0] = total_steps;
        stats[1] = stalls;
        stats[2] = n_visits;
        stats[3] = 0;
        for (const PairHostMem &m : mems) stats[3] += m.stores;
        stats[4] = (long)L.words * 4;
        stats[5] = (long)waves.size();
        stats[6] = pl.total_steps;
        stats[7] = pl.groups;
    }
    for (auto *w : waves) {
        if (w->bad() && rc == GG_OK) rc = -11;
        delete w;
    }
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) {
            const Cell v = sheared[(size_t)gp_index(P.gl, row, col)];
            gp2[2 * ((size_t)row + (size_t)col * n)] = v.g;
            gp2[2 * ((size_t)row + (size_t)col * n) + 1] = v.w;
        }
    return rc;
}

// =====================================================================================================================
// The throughput pair sweep (sweep_pairb.h): the lanes load their own cells from the in-place layer -- `late` resolves every load at its
// use, the worst case for write-after-read hazards -- one work-group with both pairs and the two corner wavefronts.
// =====================================================================================================================
#include "sweep_pairb.h"

namespace {

struct PairBHostMem {
    Cell *layer;
    std::vector<int32_t> lds;
    bool late = false, fault = false;
    long loads = 0, stores = 0;
    Cell load_issue(bool valid, int cell)
    {
        if (!valid) return Cell{0.f, 0.f};
        ++loads;
        return layer[cell];
    }
    Cell load_value(const Cell &queued, bool valid, int cell) { return (late && valid) ? layer[cell] : queued; }
    void store(bool valid, int cell, Cell v)
    {
        if (!valid) return;
        ++stores;
        layer[cell] = v;
    }
    WP lds_wp(int word)
    {
        WP v;
        memcpy(&v, &lds[(size_t)word], 8);
        return v;
    }
    void lds_put_wp(int word, WP v) { memcpy(&lds[(size_t)word], &v, 8); }
    void entry_write(int word, WP v) // (w, tag, p, tag): one 16-byte write on the device
    {
        if (word & 3) fault = true;
        memcpy(&lds[(size_t)word], &v.w, 4);
        memcpy(&lds[(size_t)word + 2], &v.p, 4);
        lds[(size_t)word + 1] = lds[(size_t)word + 3] = 1;
    }
    bool entry_read(int word, WP &v)
    {
        if (word & 3) fault = true;
        memcpy(&v.w, &lds[(size_t)word], 4);
        memcpy(&v.p, &lds[(size_t)word + 2], 4);
        return lds[(size_t)word + 1] != 0 && lds[(size_t)word + 3] != 0;
    }
    int lds_i(int word) { return lds[(size_t)word]; }
    void lds_set(int word, int v) { lds[(size_t)word] = v; }
};

template <int PAIR> struct PairWaveB : WaveBase {
    const Params &P;
    const gp::Plan &pl;
    const gp::LdsB &L;
    PairBHostMem &mem;
    WP centre;
    int W, group, t, t_end;
    gp::Group G;
    gp::PairLaneB<PAIR> lane[64];
    bool plan_mismatch = false;
    PairWaveB(const Params &p, const gp::Plan &pl_, const gp::LdsB &l, PairBHostMem &m, WP c, int w, int W_) : P(p), pl(pl_), L(l), mem(m), centre(c), W(W_), group(w - W_) { next_group(); }
    void next_group()
    {
        group += W;
        if (group >= pl.groups) return;
        G = gp::group_of(PAIR, group, P.rings);
        for (int k = 0; k < 64; ++k) {
            lane[k].init(k, group, G, P, pl, L);
            const gp::PairLaneB<PAIR> &c = lane[k];
            const int side = c.is_x ? gp::side_x(PAIR) : gp::side_y(PAIR), k0 = gp::k0_of(side);
            for (int s = 0; s < c.len; ++s) { // the stride-64 addressing of the three lines against gp_index(), visit by visit
                int x, y;
                gp::side_xy(side, P.c, c.r, 0, k0 + s, x, y);
                if (c.st_base + 64 * (c.start + s) != gp_index(P.gl, x, y)) plan_mismatch = true;
                gp::side_xy(side, P.c, c.r, 1, k0 + s + 1, x, y);
                if (c.outA + 64 * (c.start + s) != gp_index(P.gl, x, y)) plan_mismatch = true;
                if (s + 1 < c.len) {
                    gp::side_xy(side, P.c, c.r, 0, k0 + s + 1, x, y);
                    if (c.ownA + 64 * (c.start + s) != gp_index(P.gl, x, y)) plan_mismatch = true;
                }
            }
        }
        t = G.t_first;
        t_end = G.t_first + G.steps;
        for (int k = 0; k < 64; ++k) lane[k].prime(G.t_first, mem);
    }
    bool done() const override { return group >= pl.groups; }
    bool bad() const override { return plan_mismatch || mem.fault; }
    template <int RES> void steps(int slot, const bool (&first)[64], const WP (&x_in)[64], const WP (&j_in)[64], const WP (&c0)[64], const WP (&c1)[64], WP (&res)[64])
    {
        for (int k = 0; k < 64; ++k) res[k] = lane[k].template step<RES>(t, slot, x_in[k], j_in[k], first[k], c0[k], c1[k], P, mem);
    }
    bool try_step() override
    {
        if (done()) return false;
        WP imp[64], jl[64];
        for (int k = 0; k < 64; ++k) { // everything the step takes from other wavefronts must be there
            const gp::PairLaneB<PAIR> &c = lane[k];
            if (c.first_at(t) && mem.lds_i(L.cnt_corner + c.cd) < c.r) return false;
            if (c.imports_at(t, group) && !mem.entry_read(c.import_entry(t), imp[k])) return false;
            if (c.join_from_lds_at(t, group) && !mem.entry_read(c.a_jl, jl[k])) return false;
        }
        bool first[64];
        WP c0[64], c1[64];
        for (int k = 0; k < 64; ++k) {
            first[k] = lane[k].first_at(t);
            c0[k] = mem.lds_wp(lane[k].a_s0);
            c1[k] = mem.lds_wp(lane[k].a_s1);
            lane[k].pre(first[k], mem.lds_wp(lane[k].a_pred));
        }
        WP x_in[64], j_in[64], res[64];
        for (int k = 0; k < 64; ++k) {
            const gp::PairLaneB<PAIR> &c = lane[k];
            x_in[k] = c.imports_at(t, group) ? imp[k] : (k ? lane[k - 1].h2 : WP{0.f, 0.f});
            j_in[k] = k < 32 ? (k ? lane[32 + k - 1].h1 : WP{0.f, 0.f}) : lane[k - 32].h1;
            if (c.jl_lane) j_in[k] = group > 0 ? (c.join_from_lds_at(t, group) ? jl[k] : WP{0.f, 0.f}) : centre;
        }
        const int slot = (t - G.t_first) % (int)gp::PFB;
        switch ((t - G.t_first) & 3) {
        case 0: steps<0>(slot, first, x_in, j_in, c0, c1, res); break;
        case 1: steps<1>(slot, first, x_in, j_in, c0, c1, res); break;
        case 2: steps<2>(slot, first, x_in, j_in, c0, c1, res); break;
        default: steps<3>(slot, first, x_in, j_in, c0, c1, res); break;
        }
        for (int k = 0; k < 64; ++k)
            if (lane[k].exports_at(t)) mem.entry_write(lane[k].export_entry(t), res[k]);
        // B_1 of ring 1 for the CD corner wavefront (the B chain of ring 1 is one visit, at wave-step 0 of group 0)
        if (PAIR == gp::PAIR_BC && group == 0 && t == 0) mem.entry_write(L.b1, lane[0].h1);
        if (++t >= t_end) next_group();
        return true;
    }
};

template <int CD> struct PairCornerWaveB : WaveBase {
    const Params &P;
    const gp::LdsB &L;
    PairBHostMem &mem;
    int r = 1;
    gp::CornerLaneB<CD> lane[64];
    gp::CornerHeld held[64]; // the batch before: stored after this batch's loads
    bool have_held = false;
    float in_corner, in_x1 = 0.f;
    PairCornerWaveB(const Params &p, const gp::LdsB &l, PairBHostMem &m, float centre_p) : P(p), L(l), mem(m), in_corner(centre_p) {}
    bool done() const override { return r > P.rings; }
    bool try_step() override
    {
        if (done()) return false;
        if (CD && r == 1) {
            WP b1;
            if (!mem.entry_read(L.b1, b1)) return false;
            in_x1 = b1.p;
        }
        const int k = (r - 1) % 64;
        if (k == 0) {
            for (int j = 0; j < 64; ++j) lane[j].init(r + j, P, mem);
            if (have_held)
                for (int j = 0; j < 64; ++j) held[j].flush(mem);
            have_held = false;
        }
        float x1g, x1, y0g, y0;
        lane[k].c.visits(in_corner, in_x1, x1g, x1, y0g, y0);
        lane[k].c.keep(true, x1g, y0g);
        gp::CornerLaneB<CD>::publish(r, WP{lane[k].c.R.wn[1], x1}, WP{lane[k].c.R.wn[2], y0}, P, L, mem);
        in_corner = y0;
        in_x1 = x1;
        if (k == 63 || r == P.rings) { // the batch is done
            for (int j = 0; j < 64; ++j) held[j] = lane[j].hold(j <= k);
            have_held = true;
            if (r == P.rings)
                for (int j = 0; j < 64; ++j) held[j].flush(mem);
        }
        ++r;
        return true;
    }
};

} // namespace

extern "C" int gg_debug_emulate_pair_sweep_batch(int n, double resolution, float min_dist_squared, float *gp2, float base_z, double decrease, unsigned seed, int late_loads,
                                                 int waves_per_pair, long *stats)
{
    if (n < 8 || !gp2) return GG_ERR_INVALID;
    const Params P = gg::sweep::make_params(n, resolution, min_dist_squared, decrease);
    const gp::Plan pl = gp::make_plan(P.rings);
    if (pl.groups <= 0) return GG_ERR_INVALID;
    std::vector<Cell> sheared((size_t)P.gl.elems, Cell{0.f, 0.f});
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) sheared[(size_t)gp_index(P.gl, row, col)] = Cell{gp2[2 * ((size_t)row + (size_t)col * n)], gp2[2 * ((size_t)row + (size_t)col * n) + 1]};
    sheared[(size_t)gp_index(P.gl, P.c, P.c)] = Cell{base_z, 1.0f}; // :405-411
    const WP centre{1.0f, 1.0f * base_z};
    const gp::LdsB L = gp::ldsb_of(P.c, pl, true);
    const int W = std::max(1, std::min(waves_per_pair > 0 ? waves_per_pair : pl.groups, pl.groups));
    PairBHostMem mem;
    mem.layer = sheared.data();
    mem.late = late_loads != 0;
    mem.lds.assign((size_t)L.words, 0);
    for (int k = 0; k < 64; ++k) mem.lds[(size_t)L.scratch + 4 * k + 1] = mem.lds[(size_t)L.scratch + 4 * k + 3] = 1;
    for (int cd = 0; cd < 2; ++cd) mem.lds_put_wp(gp::cornerb_word(L, P.c, cd, 0, 1), centre);
    std::vector<WaveBase *> waves;
    for (int w = 0; w < W; ++w) {
        waves.push_back(new PairWaveB<gp::PAIR_AD>(P, pl, L, mem, centre, w, W));
        waves.push_back(new PairWaveB<gp::PAIR_BC>(P, pl, L, mem, centre, w, W));
    }
    waves.push_back(new PairCornerWaveB<0>(P, L, mem, centre.p));
    waves.push_back(new PairCornerWaveB<1>(P, L, mem, centre.p));
    uint32_t rng = seed * 2654435761u + 12345u;
    auto rnd = [&]() { return rng = rng * 1664525u + 1013904223u; };
    long total_steps = 0, stalls = 0;
    int rc = GG_OK;
    for (;;) {
        bool all_done = true, progress = false;
        for (auto *w : waves) all_done &= w->done();
        if (all_done) break;
        if (seed == 0) {
            for (auto *w : waves)
                if (w->try_step()) {
                    progress = true;
                    ++total_steps;
                } else if (!w->done())
                    ++stalls;
        } else {
            for (int tries = 0; tries < 4 * (int)waves.size() && !progress; ++tries) {
                WaveBase *w = waves[(rnd() >> 8) % waves.size()];
                const uint32_t mode = (rnd() >> 8) % 8;
                int burst = mode == 0 ? 1 << 30 : mode < 4 ? 1 + (int)((rnd() >> 8) % 64) : 1;
                while (burst-- > 0 && w->try_step()) {
                    progress = true;
                    ++total_steps;
                }
                if (!progress && !w->done()) ++stalls;
            }
            if (!progress)
                for (auto *w : waves)
                    if (w->try_step()) {
                        progress = true;
                        ++total_steps;
                        break;
                    }
        }
        if (!progress) {
            rc = -10;
            break;
        }
    }
    if (stats) {
        stats[0] = total_steps;
        stats[1] = stalls;
        stats[2] = mem.loads;
        stats[3] = mem.stores;
        stats[4] = (long)L.words * 4;
        stats[5] = (long)waves.size();
        stats[6] = pl.total_steps;
        stats[7] = pl.groups;
    }
    for (auto *w : waves) {
        if (w->bad() && rc == GG_OK) rc = -11;
        delete w;
    }
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) {
            const Cell v = sheared[(size_t)gp_index(P.gl, row, col)];
            gp2[2 * ((size_t)row + (size_t)col * n)] = v.g;
            gp2[2 * ((size_t)row + (size_t)col * n) + 1] = v.w;
        }
    return rc;
}
