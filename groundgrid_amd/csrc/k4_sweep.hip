// K4 -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465), second generation: the
// ring-per-lane dataflow of sweep_core.h on gfx950.
//
// One work-group per cloud: 4 x W chain wavefronts (sides A, B, C, D of the rings; a wavefront owns groups of 64
// consecutive rings, lane = ring, ring r + 1 SKEW = 1 step behind ring r) + two corner wavefronts (lane = ring as well: 64 rings
// are prepared at once, the three dependent corner visits then run ring after ring, sweep_core.h CornerRing).
// There is no barrier after start-up and no table: every wavefront free-runs through its steps; what a step needs from
// other wavefronts arrives through LDS behind monotonic progress counters that the consumer polls (LDS operations of a
// wavefront are executed in issue order, so "data, then counter" needs no fence), what it needs from the inner ring of the
// same side arrives by a wave shift (DPP wave_shr:1) of the previous step's result.  A step comes in two halves with the wait
// for the partner side's join between them (ChainLane::step_a / step_b).
// The dependent chain of a step is: product with the predecessor's new height -> 3 adds of the Eigen tree -> IEEE divide
// -> blend; everything else (all confidence sums, the other 8 products, the decay of the confidence) is independent of it.
//
// Memory: per step and lane one 8-byte element of the own line and one of the outer line, requested PF steps ahead with
// range-checked buffer loads (lanes with nothing to fetch pass an out-of-range offset: no traffic) so that the step stays
// branch-free; results are written fire-and-forget.  LDS: 24 KB for n = 364 (progress counters, corner values, joins, the
// chains of the group-boundary rings), so several clouds share a CU and one cloud's waits are another's compute.
//
// Instruction count IS the time of a chain wavefront (a lone wavefront issues one instruction of any kind per ~5 cycles), so the step
// loop is written around it (run_chain): whether a half step may run is one scalar compare against the last step the counters read
// last are good for (ChainSync::cover inverts the closed-form needs where a counter is re-read); the trips of a group are compiled per
// phase (TripKind: lanes still starting / lane 0 still reading the previous group's chain / joins only), the two long phases without
// range tests; everything rare sits out of line (__builtin_expect), its polls as rolled loops; the wait for the join -- the sweep's
// critical path, ring after ring -- reads one counter and goes on (wait_b).  DESIGN.md 4 "The chain wavefront's step" has the numbers.
//
// The same per-lane code runs on the host under a lock-step emulation (sweep_emul.hip, tests/test_sweep_emul_cpu.py).
#include "gg_device.h"
#include "sweep_core.h"

#include <algorithm>
#include <cstdlib>

#include <stdlib.h>

namespace gg {

using namespace sweep;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

// The kernel's only static LDS, 16 bytes and 16-byte aligned, so that the dynamic region behind it (64-bit accesses all over) keeps
// its alignment: [0] ticket, [1] epoch (k_sweep), [2] a wavefront gave up a bounded wait (DevMemT::give_up).
__shared__ __attribute__((aligned(16))) uint32_t sweep_static[4];
#define sweep_wait_failed sweep_static[2]

template <bool DBG> struct DevMemT {
    __amdgpu_buffer_rsrc_t rsrc; // the interleaved (ground, confidence) layer of this cloud
    lds_int *lds;
    int dbg_mode_rt = 0; // timing experiments only (GG_SWEEP_DEBUG): 1 = drop the result stores, 2 = drop the layer loads, 4 = phase marks
    unsigned long long *marks = nullptr, last_mark = 0, acc[6] = {0, 0, 0, 0, 0, 0};
    GG_DEV int dbg_mode() const { return DBG ? dbg_mode_rt : 0; } // (the production kernel is compiled without any of this)
    GG_DEV void mark(int k)
    {
        if (!(dbg_mode() & 4)) return;
        const unsigned long long now = __builtin_readcyclecounter();
        if (k > 0) acc[k] += now - last_mark;
        last_mark = now;
    }
    GG_DEV void flush_marks()
    {
        if (dbg_mode() & 4)
            for (int k = 0; k < 6; ++k) marks[k] = acc[k];
    }
    // Bounded waits.  A wavefront of this kernel waits either for another wavefront of its OWN work-group (LDS counters) or -- the
    // importer wavefront only -- for values of ANOTHER work-group in the exchange region.  Only the second kind can wait for
    // something that never comes (a producer that is not running, a corrupted region), and everything else waits, directly or
    // not, for what the importer republishes.  So the importer counts its polls: when a wait outlasts Params::poll_cap polls it
    // gives up, hands on what it has as if it were complete (its consumers -- and through the exporter the work-groups further out
    // -- run through their steps instead of waiting in turn) and raises a flag that the work-group's first thread passes to the
    // context's error word when the kernel ends: the next gg_* call that synchronises returns GG_ERR_HIP, the outputs of the batch
    // are void, and nothing hangs.
    GG_DEV bool give_up(int polls, const Params &P) const
    {
        if (polls < P.poll_cap) return false;
        __hip_atomic_store(&sweep_wait_failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return true;
    }
    static constexpr uint32_t OOR = 0x80000000u; // beyond the buffer: loads return 0, stores are dropped, no traffic
    // exchange region of this cloud (sweep_core.h "Parts"): one WP = two 64-bit words (value | launch sequence number << 32), each
    // written and read as ONE relaxed agent-scope atomic: a reader that finds both tags current has the value, whatever the
    // caches and whichever XCD the two work-groups run on -- no fence, no separate flag, nothing to reset between launches
    unsigned long long *xchg = nullptr;
    uint32_t seq = 0;
    GG_DEV void export_wp_if(bool c, int entry, WP v) const
    {
        if (c) {
            const unsigned long long tag = (unsigned long long)seq << 32;
            __hip_atomic_store(xchg + 2 * (size_t)entry, tag | __float_as_uint(v.w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(xchg + 2 * (size_t)entry + 1, tag | __float_as_uint(v.p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    GG_DEV bool import_wp(int entry, WP &v) const
    {
        const unsigned long long a = __hip_atomic_load(xchg + 2 * (size_t)entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(xchg + 2 * (size_t)entry + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = WP{__uint_as_float((uint32_t)a), __uint_as_float((uint32_t)b)};
        return (uint32_t)(a >> 32) == seq && (uint32_t)(b >> 32) == seq;
    }
    // one PrepRec = three 64-bit LDS words (split steps); written by the preparing wavefront, read by the chain wavefront
    GG_DEV void ring_put(int word, const PrepRec &r) const
    {
        lds_u64 *p = (lds_u64 *)(lds + word);
        __hip_atomic_store(p + 0, (uint64_t)__float_as_uint(r.w_new) | ((uint64_t)__float_as_uint(r.own_g) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(p + 1, (uint64_t)__float_as_uint(r.own_w) | ((uint64_t)__float_as_uint(r.own_p) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_store(p + 2, (uint64_t)__float_as_uint(r.out_w) | ((uint64_t)__float_as_uint(r.out_p) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV PrepRec ring_get(int word) const
    {
        lds_u64 *p = (lds_u64 *)(lds + word);
        const uint64_t a = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t b = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint64_t c = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PrepRec r;
        r.w_new = __uint_as_float((uint32_t)a), r.own_g = __uint_as_float((uint32_t)(a >> 32));
        r.own_w = __uint_as_float((uint32_t)b), r.own_p = __uint_as_float((uint32_t)(b >> 32));
        r.out_w = __uint_as_float((uint32_t)c), r.out_p = __uint_as_float((uint32_t)(c >> 32));
        return r;
    }
    GG_DEV void set_counter(int word, int value) const
    {
        __hip_atomic_store(lds + word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }

    GG_DEV Cell load_issue(bool valid, int cell) const
    {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (valid && !(dbg_mode() & 2)) ? (uint32_t)cell * 8u : OOR, 0, 0);
        return Cell{__uint_as_float(v.x), __uint_as_float(v.y)};
    }
    GG_DEV Cell load_value(const Cell &queued, bool, int) const { return queued; }
    // FRESH maps (gg_internal.h Arena::gp_bits): is the cell's element in memory?  The slot's bit words lie behind its layer, inside the
    // same buffer descriptor, from byte bits_byte0.  (Per-lane gather: the corner lanes' cells and the two cells of a chain that lie off its
    // lines; the cells ON the lines are looked up a wave-step at a time, run_chain)
    uint32_t bits_byte0 = 0u;
    GG_DEV uint32_t bits_dword(uint32_t voffset, uint32_t soffset) const { return __builtin_amdgcn_raw_buffer_load_b32(rsrc, voffset, soffset, 0); }
    GG_DEV bool bit_of(int cell) const
    {
        const unsigned e = (unsigned)cell - 1u;
        const uint32_t w = bits_dword(bits_byte0 + (e >> 5) * 4u, 0u);
        return cell <= 0 || ((w >> (e & 31u)) & 1u) != 0u; // (element 0, the centre cell, is written when the kernel starts)
    }
    GG_DEV Cell fresh(const Cell &v) const
    {
        Cell o;
        asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(o.g), "=&v"(o.w) : "v"(v.g), "v"(v.w));
        return o;
    }
    GG_DEV void store(bool valid, int cell, Cell v) const
    {
        u32x2 d;
        d.x = __float_as_uint(v.g);
        d.y = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, (valid && !(dbg_mode() & 1)) ? (uint32_t)cell * 8u : OOR, 0, 0);
    }
    // LDS.  Other wavefronts write what is read here: every access is an atomic (relaxed, work-group scope) so that the
    // compiler neither caches nor hoists it; ordering comes from the hardware (in-order DS queue per wavefront).
    GG_DEV WP get(int word) const
    {
        const uint64_t u = __hip_atomic_load((lds_u64 *)(lds + word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return WP{__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32))};
    }
    GG_DEV void put(int word, WP v) const
    {
        const uint64_t u = (uint64_t)__float_as_uint(v.w) | ((uint64_t)__float_as_uint(v.p) << 32);
        __hip_atomic_store((lds_u64 *)(lds + word), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV void publish(int data_word, WP v, int counter_word, int value) const
    {
        // data, then counter, as two DS writes of one instruction group: the DS queue keeps them in this order
        const uint32_t da = (uint32_t)(data_word * 4) + lds_base(), ca = (uint32_t)(counter_word * 4) + lds_base();
        u32x2 d;
        d.x = __float_as_uint(v.w);
        d.y = __float_as_uint(v.p);
        asm volatile("ds_write_b64 %0, %1\n\tds_write_b32 %2, %3" ::"v"(da), "v"(d), "v"(ca), "v"(value) : "memory");
    }
    GG_DEV void publish_if(bool c, int lane, const LdsMap &L, int data_word, WP v, int counter_word, int value) const
    {
        const int sw = (L.scratch + 1 + 3 * lane) & ~1; // (8-byte aligned data slot, the counter slot behind it)
        publish(c ? data_word : sw, v, c ? counter_word : sw + 2, value);
    }
    GG_DEV void put_if(bool c, int lane, const LdsMap &L, int word, WP v) const { put(c ? word : (L.scratch + 1 + 3 * lane) & ~1, v); }
    GG_DEV WP bcast(WP v, int lane) const
    {
        return WP{__int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.w), lane)), __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.p), lane))};
    }
    GG_DEV int counter(int word) const
    {
        int v;
        const uint32_t ca = (uint32_t)(word * 4) + lds_base();
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ca) : "memory");
        return __builtin_amdgcn_readfirstlane(v);
    }
    // a counter and the (per-lane) data word behind it in one round trip: the DS queue is in order, a counter that says "published"
    // came back before the data was read
    GG_DEV int counter_and_get(int counter_word, int data_word, WP &v) const
    {
        int c;
        u32x2 d;
        const uint32_t ca = (uint32_t)(counter_word * 4) + lds_base(), da = (uint32_t)(data_word * 4) + lds_base();
        asm volatile("ds_read_b32 %0, %2\n\tds_read_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)" : "=&v"(c), "=&v"(d) : "v"(ca), "v"(da) : "memory");
        v = WP{__uint_as_float(d.x), __uint_as_float(d.y)};
        return __builtin_amdgcn_readfirstlane(c);
    }
    GG_DEV void counters3(int w0, int w1, int w2, int &v0, int &v1, int &v2) const
    {
        int a, b, c;
        const uint32_t a0 = (uint32_t)(w0 * 4) + lds_base(), a1 = (uint32_t)(w1 * 4) + lds_base(), a2 = (uint32_t)(w2 * 4) + lds_base();
        asm volatile("ds_read_b32 %0, %3\n\tds_read_b32 %1, %4\n\tds_read_b32 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(a), "=&v"(b), "=&v"(c)
                     : "v"(a0), "v"(a1), "v"(a2)
                     : "memory");
        v0 = __builtin_amdgcn_readfirstlane(a);
        v1 = __builtin_amdgcn_readfirstlane(b);
        v2 = __builtin_amdgcn_readfirstlane(c);
    }
    GG_DEV uint32_t lds_base() const { return (uint32_t)(uintptr_t)lds; } // LDS byte address of word 0 (an address-space-3 pointer IS the offset)
};

// a wave-uniform integer as the compiler must take it where it stands (a scalar register): a loop-invariant condition tested through
// this is a scalar compare and branch in the step, not a lane mask kept through the loop and negated with vector instructions
GG_DEV int scalar_here(int v)
{
    v = __builtin_amdgcn_readfirstlane(v);
    asm volatile("" : "+s"(v));
    return v;
}

GG_DEV float wave_shr1(float v)
{
    // DPP wave_shr:1 -- lane l reads lane l - 1 (lane 0 keeps its own value)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false));
}

// optional instrumentation (gg_debug_sweep_timing): per wavefront {start, end, cycles spent polling, polls that had to wait}
template <bool DBG> struct WaveClockT {
    unsigned long long *out_rt; // nullptr: off
    GG_DEV unsigned long long *out_ptr() const { return DBG ? out_rt : nullptr; }
    unsigned long long t0 = 0, polling = 0, waits = 0;
    GG_DEV void begin() { t0 = out_ptr() ? __builtin_readcyclecounter() : 0ull; }
    GG_DEV void end(int wave, int lane)
    {
        unsigned long long *out = out_ptr();
        if (out && lane == 0) {
            out[wave * 4 + 0] = t0;
            out[wave * 4 + 1] = __builtin_readcyclecounter();
            out[wave * 4 + 2] = polling;
            out[wave * 4 + 3] = waits;
        }
    }
};

#ifndef GG_SWEEP_SLEEP_LONG
#define GG_SWEEP_SLEEP_LONG 6
#endif
constexpr int SLEEP_LONG = GG_SWEEP_SLEEP_LONG;

// A/B switches of run_chain (the defaults are what was measured best, profiles/r05e/sweep_ab_*.log)
#ifndef GG_SWEEP_REC_AHEAD
#define GG_SWEEP_REC_AHEAD 1 // split steps: the chain wavefront reads its preparing wavefront's record a step early
#endif
#ifndef GG_SWEEP_BATCH_RANGES
#define GG_SWEEP_BATCH_RANGES 3 // throughput launches: 1 = the ranges of a step's rare blocks tested in every step, 3 = once per trip
#endif
#ifndef GG_SWEEP_BATCH_KINDS
#define GG_SWEEP_BATCH_KINDS 2 // trip variants of the throughput launches: 1 = one loop, 2 = the first-step block peeled off, 3 = the phases of the latency launches (2.5x slower: instruction cache)
#endif
// what the six steps of a trip are compiled for (sweep_core.h ChainLane::step_a: STARTS, BND, JOIN)
template <bool S, int B, int J> struct TripKind {
    static constexpr bool starts = S;
    static constexpr int bnd = B, join = J;
};

// SPLIT: a preparing wavefront (run_prep) does the layer half of every step (sweep_core.h "Split steps")
template <int SIDE, bool DBG, bool SPLIT, bool FRESH = false>
GG_DEV void run_chain(const Params &P, const LdsMap &L, DevMemT<DBG> &mem, int wave_of_side, int lane, WaveClockT<DBG> &clk, int g0, int g1)
{
    static_assert(!(FRESH && SPLIT), "fresh maps: the plain chain only");
    ChainLane<SIDE> st;
    int have_prep = 0; // (split steps) cached count of prepared wave-steps
    // a counter only lane 0 writes: the other lanes write to their scratch words instead of being masked off (an exec-mask round trip
    // is five scalar instructions of every step)
    const int w_take = lane == 0 ? L.take_done + SIDE : ((L.scratch + 1 + 3 * lane) & ~1) + 2;
    for (int group = g0 + wave_of_side; group < g1; group += P.waves_per_side) {
        const int r0 = LANES * group + 1;
        const int nl = min(P.rings - (r0 - 1), (int)LANES);
        st.init(lane, r0, nl, group, P, L);
        const int t_first = group_first_step(), t_last = group_last_step<SIDE>(r0, nl);
        const int has_next = __builtin_amdgcn_readfirstlane(group + 1 < P.groups ? 1 : 0);
        // FRESH: the requests of wave-step t name elements ownA + 64 (t + 1 + PF) and outA + 64 (t + 1 + PF): bit (X - 1) & 63 of word
        // ((X - 1) >> 6) + t + 1 + PF of the slot's bit map (a wave-step of the own line is one word, gp_layout.h; of the outer line bits 1 .. 63
        // of a word and, for lane 63, bit 0 of a word of the next storage group).  Each lane fetches the 32-bit half that holds its bit, two
        // steps ahead and BEFORE that step's cell requests: loads return in order, and by then only cells the step needs anyway are older.
        // (Measured per 1024 clouds, k_sweep 0.88 ms on written maps: this 0.98 ms; the words as scalar loads a step ahead, or 64 steps' words
        // in one coalesced load and v_readlane per step -- lane masks in scalar registers either way -- 1.06 ms: scalar-register spills.)
        uint32_t fo_own = 0u, fo_out = 0u, sh_own = 0u, sh_out = 0u, mq_own[2] = {0u, 0u}, mq_out[2] = {0u, 0u};
        if constexpr (FRESH) {
            const unsigned eo = (unsigned)st.ownA - 1u, eu = (unsigned)st.outA - 1u;
            fo_own = mem.bits_byte0 + 8u * (unsigned)((int)(eo >> 6) + t_first + 1 + (int)PF) + 4u * ((eo & 63u) >> 5);
            fo_out = mem.bits_byte0 + 8u * (unsigned)((int)(eu >> 6) + t_first + 1 + (int)PF) + 4u * ((eu & 63u) >> 5);
            sh_own = eo & 31u;
            sh_out = eu & 31u;
            st.xold_bit = mem.bit_of(st.xold_cell);
            st.end_bit = mem.bit_of(st.own_end);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                mq_own[k] = mem.bits_dword(fo_own, 8u * (unsigned)k);
                mq_out[k] = mem.bits_dword(fo_out, 8u * (unsigned)k);
            }
        }
        // TRIP steps per trip, no per-step condition: a step past t_last finds every lane idle (no loads, no stores, nothing
        // to wait for), and without a conditional around it the queue registers of a slot never meet a control-flow join --
        // a join makes the compiler copy freshly loaded registers, i.e. wait for the loads it has just issued
        ChainSync<SIDE> sync;
        sync.init(r0, nl, group, P, L);
        // which third of the steps can end a chain: t = tb + u with tb = t_first (mod TRIP), TRIP = 0 (mod 3)
        const int turn = __builtin_amdgcn_readfirstlane((((join_turn_residue<SIDE>(r0) - t_first) % 3) + 3) % 3); // u = turn (mod 3)
        // the trips in three loops (sweep_core.h ChainLane::step_a on STARTS, BND, JOIN): the general one while lanes still start and
        // where two ranges meet, one for the trips in which lane 0 still reads the previous group's chain and no lane is at its join,
        // one for the trips after that -- the last two are most of a group's steps and run without the range tests
        auto trip = [&](const int tb, auto kind) __attribute__((always_inline)) {
            using K = decltype(kind);
            constexpr bool STARTS = K::starts;
            // (kinds 3: the wave-uniform ranges of a step's rare blocks tested once per trip, not in every step)
            const bool trip_bnd = K::bnd == 3 && group > 0 && tb + (int)TRIP - 1 >= 0 && tb < st.u_len0;
            const bool trip_join = K::join == 3 && tb + (int)TRIP - 1 >= st.u_join_first && tb <= st.u_join_last;
            const bool trip_pub = K::join == 3 && has_next != 0 && tb + (int)TRIP - 1 >= st.u_l3_last && tb < st.u_lend_last;
            PrepRec rec_ahead{};
#pragma unroll
            for (int u = 0; u < TRIP; ++u) {
                const int t = tb + u;
                sync.advance(t);
                // (rare) something the next half step reads has not been published yet as far as the counters read last say.  A waiting
                // wavefront's polls take issue slots from the working wavefronts of its SIMD: back off after the first few (the hand-over
                // it waits for is then several steps away).  Rolled loops: this is cold code in the middle of every step of the trip
                auto wait_a = [&]() {
                    const unsigned long long w0 = clk.out_ptr() ? __builtin_readcyclecounter() : 0ull;
                    sync.poll(mem);
                    int spins = 0;
#pragma nounroll
                    for (; spins < 4 && !sync.slow_ok_a(); ++spins) {
                        __builtin_amdgcn_s_sleep(1);
                        sync.poll(mem);
                    }
#pragma nounroll
                    while (!sync.slow_ok_a()) {
                        __builtin_amdgcn_s_sleep(SLEEP_LONG);
                        sync.poll(mem);
                        ++spins;
                    }
                    sync.cover(); // how far the counters read last are good for
                    if (clk.out_ptr()) {
                        clk.polling += __builtin_readcyclecounter() - w0;
                        clk.waits += spins ? 1 : 0;
                    }
                };
                // the join: the wait on the sweep's critical path (the sides of a ring hand their ends to each other ring after ring).
                // One counter, and as little as possible between seeing it and going on.  (s_sleep 1 between polls is the measured optimum
                // for a lone cloud: s_nop gaps or none +2..5 %, s_sleep 2 / 3 +1.5 / +3.5 %)
                auto wait_b = [&](WP &join_read) {
                    const unsigned long long w0 = clk.out_ptr() ? __builtin_readcyclecounter() : 0ull;
                    const int need = sync.need_join_at(t);
                    sync.have_join = K::join == 2 ? mem.counter_and_get(sync.w_join, st.a_join, join_read) : mem.counter(sync.w_join);
                    int spins = 0;
#pragma nounroll
                    for (; spins < 4 && sync.have_join < need; ++spins) {
                        __builtin_amdgcn_s_sleep(1);
                        sync.have_join = K::join == 2 ? mem.counter_and_get(sync.w_join, st.a_join, join_read) : mem.counter(sync.w_join);
                    }
#pragma nounroll
                    while (sync.have_join < need) {
                        __builtin_amdgcn_s_sleep(SLEEP_LONG);
                        sync.have_join = K::join == 2 ? mem.counter_and_get(sync.w_join, st.a_join, join_read) : mem.counter(sync.w_join);
                        ++spins;
                    }
                    sync.cover_b();
                    if (clk.out_ptr()) {
                        clk.polling += __builtin_readcyclecounter() - w0;
                        clk.waits += spins ? 1 : 0;
                    }
                };
                if (__builtin_expect(!sync.ok_a(), 0)) wait_a();
                const WP ho = st.handed_over();
                const WP x_in{wave_shr1(ho.w), wave_shr1(ho.p)};
                // the lane's join slot, read a half step early (valid if the counters read last cover step_b of this step: checked below)
                WP join_read{0.f, 0.f};
                if (K::join == 2) join_read = mem.get(st.a_join);
                constexpr int t_first_mod = ((-2 - (int)PF) % (int)SKEW + (int)SKEW) % (int)SKEW; // tb = t_first (mod TRIP), TRIP = 0 (mod SKEW)
                const int tmod = (t_first_mod + u) % (int)SKEW;
                if (SPLIT) {
                    const int step_no = t - t_first;
                    // the record of this step was read a step ago (the first of a trip: now), the next step's is read now: the LDS
                    // round trip of a record is over when its step begins.  GG_SWEEP_REC_AHEAD=0: every step reads its own
                    const int AHEAD = (GG_SWEEP_REC_AHEAD && u + 1 < (int)TRIP) ? 1 : 0;
                    if (__builtin_expect(have_prep <= step_no + AHEAD, 0)) { // (rare) the preparing wavefront is not that far ahead
                        have_prep = mem.counter(L.prep_done + SIDE);
                        while (have_prep <= step_no + AHEAD) {
                            __builtin_amdgcn_s_sleep(1);
                            have_prep = mem.counter(L.prep_done + SIDE);
                        }
                    }
                    const int rec_word = L.prep + (SIDE * (int)PREP_DEPTH * (int)LANES + lane) * (int)PREP_WORDS; // + slot * LANES * PREP_WORDS, slot = step_no mod PREP_DEPTH = u
                    PrepRec rec;
                    if (GG_SWEEP_REC_AHEAD && u > 0) rec = rec_ahead;
                    else rec = mem.ring_get(rec_word + u * (int)LANES * (int)PREP_WORDS);
                    if (AHEAD) rec_ahead = mem.ring_get(rec_word + (u + 1) * (int)LANES * (int)PREP_WORDS);
                    st.template take<STARTS, K::bnd>(t, tmod, rec, x_in, group > 0, mem);
                    mem.set_counter(w_take, step_no + 1 + AHEAD); // (after the reads above: the DS queue is in order)
                } else if constexpr (FRESH) {
                    static_assert((int)TRIP % 2 == 0, "the bit queue's slot is a constant of the unrolled trip");
                    const uint32_t w_own = mq_own[u & 1], w_out = mq_out[u & 1]; // (requested two steps ago)
                    const uint32_t ahead = 8u * (unsigned)(t + 2 - t_first);
                    mq_own[u & 1] = mem.bits_dword(fo_own, ahead);
                    mq_out[u & 1] = mem.bits_dword(fo_out, ahead);
                    const bool own_bit = ((w_own >> sh_own) & 1u) != 0u, out_bit = ((w_out >> sh_out) & 1u) != 0u;
                    st.template step_a<STARTS, K::bnd, true>(t, u % (int)PF, tmod, x_in, P, L, K::bnd == 3 ? trip_bnd : group > 0, mem, own_bit, out_bit);
                } else {
                    st.template step_a<STARTS, K::bnd>(t, u % (int)PF, tmod, x_in, P, L, K::bnd == 3 ? trip_bnd : group > 0, mem);
                }
                if (__builtin_expect(!sync.ok_b(), 0)) wait_b(join_read);
                st.template step_b<STARTS, K::join>(t, tmod, P, L, K::join == 3 ? trip_pub : (SPLIT ? scalar_here(has_next) : has_next) != 0, group, mem,
                                                    SKEW != 1 || (SPLIT ? scalar_here(turn) : turn) == (u % 3), join_read, trip_join); // (scalar_here: worth it for a lone wavefront only)
            }
        };
        if (SPLIT || GG_SWEEP_BATCH_KINDS >= 3) {
            const int bnd_until = min(st.u_len0, st.u_join_first);
            for (int tb = t_first; tb <= t_last;) {
                if (tb > st.u_start_last && tb >= st.u_join_first && (group == 0 || tb >= st.u_len0)) {
                    for (; tb <= t_last; tb += TRIP) trip(tb, TripKind<false, 0, 2>{});
                } else if (group > 0 && tb > st.u_start_last && tb + (int)TRIP - 1 < bnd_until) {
                    for (; tb <= t_last && tb + (int)TRIP - 1 < bnd_until; tb += TRIP) trip(tb, TripKind<false, 2, 0>{});
                } else {
                    trip(tb, TripKind<true, 1, 1>{});
                    tb += TRIP;
                }
            }
        } else {
            // (throughput launches: the work-groups of many clouds share a compute unit's instruction cache, every group at another
            // place of its loops -- a third copy of the trip made the launch 2.5 times slower)
            int tb = t_first;
#if GG_SWEEP_BATCH_KINDS >= 2
            for (; tb <= t_last && tb <= st.u_start_last; tb += TRIP) trip(tb, TripKind<true, GG_SWEEP_BATCH_RANGES, GG_SWEEP_BATCH_RANGES>{});
            for (; tb <= t_last; tb += TRIP) trip(tb, TripKind<false, GG_SWEEP_BATCH_RANGES, GG_SWEEP_BATCH_RANGES>{});
#else
            for (; tb <= t_last; tb += TRIP) trip(tb, TripKind<true, GG_SWEEP_BATCH_RANGES, GG_SWEEP_BATCH_RANGES>{});
#endif
        }
    }
}

// The preparing wavefront of one side (split steps): the layer half of every wave-step of the work-group's ring group, a few
// steps ahead of the chain wavefront, into the side's LDS ring.
template <int SIDE, bool DBG> GG_DEV void run_prep(const Params &P, const LdsMap &L, DevMemT<DBG> &mem, int lane, int group)
{
    PrepLane<SIDE> st;
    const int r0 = LANES * group + 1;
    const int nl = min(P.rings - (r0 - 1), (int)LANES);
    st.init(lane, r0, nl, P);
    static_assert((int)PREP_DEPTH == (int)TRIP, "a step's ring slot is its position in the trip");
    const int t_first = group_first_step(), t_last = group_last_step<SIDE>(r0, nl);
    int have_taken = 0;
    const int w_prep = lane == 0 ? L.prep_done + SIDE : ((L.scratch + 1 + 3 * lane) & ~1) + 2; // (lane 0's counter; see run_chain)
    for (int tb = t_first; tb <= t_last; tb += TRIP) {
#pragma unroll
        for (int u = 0; u < TRIP; ++u) {
            const int t = tb + u, step_no = t - t_first;
            if (step_no - have_taken >= (int)PREP_DEPTH) { // the ring is full: wait for the chain wavefront
                have_taken = mem.counter(L.take_done + SIDE);
                while (step_no - have_taken >= (int)PREP_DEPTH) {
                    __builtin_amdgcn_s_sleep(2);
                    have_taken = mem.counter(L.take_done + SIDE);
                }
            }
            const PrepRec rec = st.step(t, u % (int)PF, P, mem);
            mem.ring_put(L.prep + ((SIDE * (int)PREP_DEPTH + u) * (int)LANES + lane) * (int)PREP_WORDS, rec);
            mem.set_counter(w_prep, step_no + 1); // (the records first: in-order DS queue)
        }
    }
}

template <int CD, bool DBG, bool FRESH = false> GG_DEV void run_corner(const Params &P, const LdsMap &L, DevMemT<DBG> &mem, int lane, WaveClockT<DBG> &clk, int g0, int g1)
{
    (void)clk;
    CornerRing<CD> st;
    for (int group = g0; group < g1; ++group) {
        const int r0 = LANES * group + 1;
        const int nl = min(P.rings - (r0 - 1), (int)LANES);
        // prepare: 64 rings at once
        st.template issue<FRESH>(r0 + lane, P, mem);
        st.finish(P, mem);
        // a part that does not start at the centre: the ring before its first one comes from another work-group (the importer
        // wavefront sets the counter once the two corner values of that ring are in this work-group's table)
        if (group == g0 && g0 > 0)
            while (mem.counter(L.corner_done + CD) < r0 - 1) __builtin_amdgcn_s_sleep(SLEEP_LONG);
        const int prev = L.corner + 2 * ((CD * P.c + r0 - 1) * 2);
        WP in_corner = mem.get(prev + 2); // Y_0 of the ring before the group (ring 0: the centre) -- written by this wavefront
        WP in_x1 = r0 > 1 ? mem.get(prev) : WP{0.f, 0.f};
        // recur: ring after ring
        for (int l = 0; l < nl; ++l) {
            if (CD && r0 + l == 1) {
                while (!CornerRing<CD>::ready(1, L, mem)) __builtin_amdgcn_s_sleep(1);
                in_x1 = mem.get(L.join + 2 * (SIDE_B * P.c + 1));
            }
            WP x1, y0;
            st.recur(lane == l, lane, in_corner, in_x1, P, L, mem, x1, y0);
            in_corner = mem.bcast(y0, l);
            in_x1 = mem.bcast(x1, l);
        }
    }
    mem.flush_marks();
}

// The importer wavefront of a part p > 0 (sweep_core.h "Parts"): polls the exchange region for what the part inside publishes about
// its last ring (ring 64 g0) -- the two corner values of either diagonal, C_last and D_last, and the four boundary chains as they
// grow -- and republishes it in this work-group's LDS tables, data first, then the progress counter the consumers poll.
template <bool DBG> GG_DEV void run_import(const Params &P, const LdsMap &L, DevMemT<DBG> &mem, int lane, int g0)
{
    const int gb = g0, rb = LANES * g0; // boundary index in the exchange region, the ring it is about
    {   // the four corner values of ring rb (lanes 0..3): the first thing this part's wavefronts need, and the first thing the
        // part inside produces about that ring (its corner wavefronts run ahead of the chains)
        WP v{0.f, 0.f};
        bool ok = lane >= 4;
        for (int polls = 0;; ++polls) {
            if (!ok) ok = mem.import_wp(xchg_misc(gb, lane), v);
            if (__all(ok) || mem.give_up(polls, P)) break;
            __builtin_amdgcn_s_sleep(SLEEP_LONG);
        }
        if (lane < 4) mem.put(L.corner + 2 * (((lane >> 1) * P.c + rb) * 2) + 2 * (lane & 1), v); // AB / CD: x1, then y0
        // (the LDS operations of one wavefront execute in issue order: values first, then the counters)
        if (lane == 0) {
            mem.set_counter(L.corner_done + 0, rb);
            mem.set_counter(L.corner_done + 1, rb);
        }
    }
    bool joins = false; // C_last / D_last of ring rb: the LAST values the part inside produces, needed by the ends of A / B of ring rb + 1
    // the boundary chains of ring rb: side s has chain_len<s>(rb) values; 64 consecutive entries per poll
    int done[4] = {0, 0, 0, 0};
    const int len[4] = {chain_len<SIDE_A>(rb), chain_len<SIDE_B>(rb), chain_len<SIDE_C>(rb), chain_len<SIDE_D>(rb)};
    for (int idle = 0;;) {
        bool all_done = joins, progress = false;
        if (!joins) {
            WP v{0.f, 0.f};
            const bool ok = lane >= 2 || mem.import_wp(xchg_misc(gb, X_JOIN_C + lane), v);
            if (__all(ok)) {
                if (lane < 2) mem.put(L.join + 2 * ((lane == 0 ? (int)SIDE_C : (int)SIDE_D) * P.c + rb), v);
                if (lane == 0) {
                    mem.set_counter(L.join_done + SIDE_C, rb);
                    mem.set_counter(L.join_done + SIDE_D, rb);
                }
                joins = true;
                progress = true;
            }
        }
#pragma unroll
        for (int side = 0; side < 4; ++side) {
            if (done[side] >= len[side]) continue; // (uniform)
            all_done = false;
            const int e = done[side] + lane;
            WP v{0.f, 0.f};
            const bool ok = e < len[side] && mem.import_wp(xchg_chain(gb, side, e), v);
            const unsigned long long m = __ballot(ok);
            const int nv = (~m == 0ull) ? 64 : __builtin_ctzll(~m); // the values arrive in order: take the leading run
            if (nv == 0) continue;
            if (lane < nv) mem.put(L.bnd + 2 * (side * L.bnd_stride + bnd_offset(g0 - 1) - L.bnd_base + e), v);
            done[side] += nv;
            if (lane == 0) mem.set_counter(L.bnd_done + side * P.groups + (g0 - 1), done[side]);
            progress = true;
        }
        if (all_done) break;
        idle = progress ? 0 : idle + 1;
        if (!progress) {
            if (mem.give_up(idle, P)) { // (what never arrived is handed on as it is: the consumers must not wait for it either)
                if (lane == 0) {
                    mem.set_counter(L.join_done + SIDE_C, rb);
                    mem.set_counter(L.join_done + SIDE_D, rb);
                    for (int side = 0; side < 4; ++side) mem.set_counter(L.bnd_done + side * P.groups + (g0 - 1), len[side]);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(SLEEP_LONG);
        }
    }
}

// The exporter wavefront of a part that has a successor: copies what the chain and corner wavefronts of this work-group publish
// about the part's last ring (ring 64 g1) from LDS into the exchange region, as it appears.
template <bool DBG> GG_DEV void run_export(const Params &P, const LdsMap &L, DevMemT<DBG> &mem, int lane, int g1)
{
    const int gb = g1, rb = LANES * g1;
    int sent[4] = {0, 0, 0, 0};
    const int len[4] = {chain_len<SIDE_A>(rb), chain_len<SIDE_B>(rb), chain_len<SIDE_C>(rb), chain_len<SIDE_D>(rb)};
    bool head = false, corners = false;
    for (;;) {
        bool all_done = head && corners, progress = false;
#pragma unroll
        for (int side = 0; side < 4; ++side) {
            if (sent[side] >= len[side]) continue; // (uniform)
            all_done = false;
            const int avail = mem.counter(L.bnd_done + side * P.groups + (g1 - 1));
            while (sent[side] < avail) { // (uniform)
                const int e = sent[side] + lane;
                if (e < avail) mem.export_wp_if(true, xchg_chain(gb, side, e), mem.get(L.bnd + 2 * (side * L.bnd_stride + bnd_offset(g1 - 1) - L.bnd_base + e)));
                sent[side] = min(sent[side] + (int)LANES, avail);
                progress = true;
            }
        }
        if (!corners) {
            int c0, c1, unused;
            mem.counters3(L.corner_done + 0, L.corner_done + 1, L.join_done + SIDE_C, c0, c1, unused);
            if (c0 >= rb && c1 >= rb) { // (uniform) AB x1, AB y0, CD x1, CD y0 of ring rb: lanes 0..3
                if (lane < 4) mem.export_wp_if(true, xchg_misc(gb, lane), mem.get(L.corner + 2 * (((lane >> 1) * P.c + rb) * 2) + 2 * (lane & 1)));
                corners = true;
                progress = true;
            }
        }
        if (!head) {
            const int jc = mem.counter(L.join_done + SIDE_C), jd = mem.counter(L.join_done + SIDE_D);
            if (jc >= rb && jd >= rb) { // (uniform) C_last, D_last of ring rb: lanes 0..1
                if (lane < 2 && P.debug_fault != 2) mem.export_wp_if(true, xchg_misc(gb, X_JOIN_C + lane), mem.get(L.join + 2 * ((lane == 0 ? (int)SIDE_C : (int)SIDE_D) * P.c + rb)));
                head = true;
                progress = true;
            }
        }
        if (all_done) break;
        if (!progress) __builtin_amdgcn_s_sleep(SLEEP_LONG);
    }
}

// 5 waves per SIMD (<= 96 registers, nothing spilled): two clouds share a CU.  (Measured: forcing 64 registers for three clouds
// per CU spills and is 1.6x slower at 1024 clouds per launch.)
// PARTS: the launch cuts every cloud into several work-groups (tickets, importer / exporter wavefronts, the exchange region); the
// throughput launches -- one work-group per cloud -- are compiled without any of that.
// FRESH: every map of the launch is fresh (gg_internal.h Arena::gp_bits; no split steps): a variant of its own --
// the work-groups of a throughput launch share the instruction cache, a kernel that carried both step codes was the slower one for both
template <bool DBG, bool PARTS, bool FRESH = false>
__global__ __launch_bounds__(1024, 5) void k_sweep(const Arena a, const Params P, const CloudParams *__restrict__ params, int n_clouds, int n_parts_rt,
                                                   unsigned long long *dbg)
{
    const int n_parts = PARTS ? n_parts_rt : 1;
    extern __shared__ __attribute__((aligned(16))) int lds[];
    uint32_t *s_sync = sweep_static;
    // work-group -> (cloud, part).  A part waits for values of the part inside it (feed-forward only), so a consumer must never
    // hold a place its producer needs: the work-groups of a launch with several parts per cloud take a TICKET when they start
    // running (one atomic counter) and the ticket, not blockIdx, names (cloud, part) -- the parts of a cloud follow each other in
    // ticket order, so a producer has always started before its consumer, in whatever order the dispatcher starts work-groups
    // (MI355X_MICROARCH.md Contract [G]).  In bundles of eight clouds the parts of a cloud differ by multiples of 8 in the id:
    // when work-groups do start in index order they go to the 8 XCDs round-robin and the parts of a cloud meet in ONE L2 (speed
    // only: the hand-over is correct on any placement, agent-scope atomics).  The values in the exchange region are tagged with an
    // EPOCH that lives in device memory: the last work-group of a launch to finish advances it (and re-arms the ticket counter),
    // so nothing depends on a per-launch kernel argument (a replayed graph would freeze one).
    uint32_t *sync_words = a.sweep_sync;
    uint32_t id = blockIdx.x, epoch = 0u;
    if (PARTS) {
        if (threadIdx.x == 0) {
            s_sync[0] = __hip_atomic_fetch_add(sync_words + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_sync[1] = __hip_atomic_load(sync_words + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        id = s_sync[0];
        epoch = s_sync[1];
        if (P.debug_fault == 1) id = gridDim.x - 1u - id;
    }
    int cloud, part;
    {
        const int full = n_clouds & ~7, tail = n_clouds - full;
        if ((int)id < full * n_parts) {
            const int rem = (int)id % (8 * n_parts);
            part = rem >> 3;
            cloud = ((int)id / (8 * n_parts)) * 8 + (rem & 7);
        } else {
            const int rem = (int)id - full * n_parts;
            part = rem / tail;
            cloud = full + rem % tail;
        }
    }
    const int g0 = part * P.gpw, g1 = min(g0 + P.gpw, P.groups);
    const LdsMap L = lds_layout(P.c, P.groups, g0, g1, P.split_steps != 0);
    const CloudParams &cp = params[cloud];
    float2 *gp2 = gp2_ptr(a, cp.slot);
    float *percall = percall_ptr(a, cp.slot);
    const int nthreads = blockDim.x;

    // hand-over tables start empty: counters 0 = "ring 0 done", and ring 0 of every table is the centre cell
    for (int k = threadIdx.x; k < L.corner; k += nthreads) lds[k] = 0;
    if (PARTS && threadIdx.x == 0) sweep_wait_failed = 0u;
    const WP centre{1.0f, 1.0f * cp.base_z}; // :405 groundpatch(centre) = 1, :406-411 ground(centre) = translation.z
    if (threadIdx.x == 0) {
        if (part == 0) gp2[gp_index(P.gl, P.c, P.c)] = make_float2(cp.base_z, 1.0f);
        float *f = reinterpret_cast<float *>(lds);
        for (int side = 0; side < 2; ++side) {
            f[L.corner + 2 * ((side * P.c + 0) * 2) + 2] = centre.w;
            f[L.corner + 2 * ((side * P.c + 0) * 2) + 3] = centre.p;
        }
        for (int side = SIDE_C; side <= SIDE_D; ++side) {
            f[L.join + 2 * (side * P.c + 0)] = centre.w;
            f[L.join + 2 * (side * P.c + 0) + 1] = centre.p;
        }
    }
    // FRESH maps: the cells no sweep visits (ring >= c) were not written by the reset either: unless k_patch has written them (maps with an
    // odd number of rows: its quadrant loops reach one line into ring c, :325-328) they take the reset's pair now, from the padding element
    // that holds it.  (Nobody reads these before the launch ends: where the bit is 0 the sweep reads the padding element.)
    if (FRESH) {
        const float2 init = gp2[a.gp_fresh_cell];
        const unsigned long long *bw = a.gp_bits + (size_t)cp.slot * a.gp_bits_stride;
        for (int k = (int)threadIdx.x + part * nthreads; k < a.gp_border_n; k += nthreads * n_parts) {
            const int e = a.gp_border[k];
            const unsigned q = (unsigned)e - 1u;
            if (((bw[q >> 6] >> (q & 63u)) & 1ull) == 0ull) gp2[e] = init;
        }
    }
    // :147 map["points"].setConstant(0.0) -- K3 was the last reader of the KEPT counts; K5 re-counts non-ground points.  Only the
    // half columns of tiles that received records hold anything but 0 (K2's tile_live invariant): one wavefront per such tile, 4 cells
    // per lane
    if (!P.keep_points) {
        const uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
        const int lane_ = threadIdx.x & 63, wave_ = (int)(threadIdx.x >> 6) + part * (nthreads >> 6), nwaves = (nthreads >> 6) * n_parts; // (shared by the parts)
        for (int rank = wave_; rank < a.g.T; rank += nwaves) {
            const uint32_t cols_live = tile_live[rank];
            if (!cols_live) continue; // (uniform)
            float *points = percall + percall_index(rank, PL_POINTS, 0); // (the tile's 256 counts are contiguous)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = lane_ + 64 * k;
                if ((cols_live >> live_bit(cell)) & 1u) points[cell] = 0.0f;
            }
        }
    }
    __syncthreads(); // the only barrier of the sweep

    DevMemT<DBG> mem;
    mem.rsrc = __builtin_amdgcn_make_buffer_rsrc(gp2, 0, FRESH ? (a.gp_bits_off + a.gp_bits_words) * 8 : P.gl.elems * 8, 0x00020000); // (FRESH: the layer and its bit words)
    mem.lds = (lds_int *)lds;
    mem.xchg = a.sweep_xchg + (size_t)cp.slot * a.sweep_xchg_stride;
    mem.seq = epoch;
    if (FRESH) mem.bits_byte0 = (uint32_t)a.gp_bits_off * 8u;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    const int W = P.waves_per_side;
    WaveClockT<DBG> clk;
    clk.out_rt = (DBG && dbg && blockIdx.x == 0) ? dbg : nullptr;
    mem.dbg_mode_rt = (DBG && dbg) ? (int)dbg[63] : 0;
    mem.marks = (DBG && dbg) ? dbg + 48 + ((threadIdx.x >> 6) & 1) * 6 : nullptr; // (the two corner wavefronts have consecutive wave ids)
    clk.begin();
    // wavefront id -> (side, wavefront of the side): the four sides of one ring group work at the same time and the wavefronts of
    // a work-group go to the CU's four SIMDs round-robin, so ids 4 w + side put them on four different SIMDs (with the sides'
    // wavefronts numbered consecutively, A_w and C_w shared one SIMD and B_w and D_w another while two SIMDs idled)
    const int side = wave & 3, w_of_side = wave >> 2;
    if (!FRESH && P.split_steps) { // one ring group per work-group: wavefronts 0..3 chains, 4..7 their preparing wavefronts, 8 / 9 corners, 10 importer, 11 exporter
        if (wave < 4) {
            if (side == SIDE_A) run_chain<SIDE_A, DBG, true>(P, L, mem, 0, lane, clk, g0, g1);
            else if (side == SIDE_B) run_chain<SIDE_B, DBG, true>(P, L, mem, 0, lane, clk, g0, g1);
            else if (side == SIDE_C) run_chain<SIDE_C, DBG, true>(P, L, mem, 0, lane, clk, g0, g1);
            else run_chain<SIDE_D, DBG, true>(P, L, mem, 0, lane, clk, g0, g1);
        } else if (wave < 8) {
            if (side == SIDE_A) run_prep<SIDE_A, DBG>(P, L, mem, lane, g0);
            else if (side == SIDE_B) run_prep<SIDE_B, DBG>(P, L, mem, lane, g0);
            else if (side == SIDE_C) run_prep<SIDE_C, DBG>(P, L, mem, lane, g0);
            else run_prep<SIDE_D, DBG>(P, L, mem, lane, g0);
        } else if (wave == 8)
            run_corner<0, DBG>(P, L, mem, lane, clk, g0, g1);
        else if (wave == 9)
            run_corner<1, DBG>(P, L, mem, lane, clk, g0, g1);
        else if (PARTS && wave == 10) {
            if (part > 0) run_import<DBG>(P, L, mem, lane, g0);
        } else if (PARTS && g1 < P.groups)
            run_export<DBG>(P, L, mem, lane, g1);
    } else if (wave < 4 * W && side == SIDE_A)
        run_chain<SIDE_A, DBG, false, FRESH>(P, L, mem, w_of_side, lane, clk, g0, g1);
    else if (wave < 4 * W && side == SIDE_B)
        run_chain<SIDE_B, DBG, false, FRESH>(P, L, mem, w_of_side, lane, clk, g0, g1);
    else if (wave < 4 * W && side == SIDE_C)
        run_chain<SIDE_C, DBG, false, FRESH>(P, L, mem, w_of_side, lane, clk, g0, g1);
    else if (wave < 4 * W)
        run_chain<SIDE_D, DBG, false, FRESH>(P, L, mem, w_of_side, lane, clk, g0, g1);
    else if (wave == 4 * W)
        run_corner<0, DBG, FRESH>(P, L, mem, lane, clk, g0, g1);
    else if (wave == 4 * W + 1)
        run_corner<1, DBG, FRESH>(P, L, mem, lane, clk, g0, g1);
    else if (PARTS && wave == 4 * W + 2) {
        if (part > 0) run_import<DBG>(P, L, mem, lane, g0);
    } else if (PARTS && g1 < P.groups)
        run_export<DBG>(P, L, mem, lane, g1);
    clk.end(wave, lane);
    if (PARTS) { // the last work-group of the launch re-arms the tickets and advances the epoch (never 0: the arena starts zeroed)
        __syncthreads();
        if (threadIdx.x == 0 && sweep_wait_failed) __hip_atomic_store(a.dev_error, (uint32_t)GG_DEVERR_SWEEP_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0 && __hip_atomic_fetch_add(sync_words + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) {
            __hip_atomic_store(sync_words + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync_words + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (bit 31 names the region the epoch belongs to -- a context has two sets of these words, gg_internal.h sweep_sync2, and a map's
            // exchange region may be written under either: the epochs of the two never meet; an epoch is never 0 in its low 31 bits)
            uint32_t next = ((epoch + 1u) & 0x7FFFFFFFu) | (epoch & 0x80000000u);
            if ((next & 0x7FFFFFFFu) == 0u) next |= 1u;
            __hip_atomic_store(sync_words + 2, next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// the largest LDS table any part of a sweep with `gpw` groups per work-group needs
static size_t parts_lds_bytes(const Params &P, int gpw, bool split = false)
{
    size_t words = 0;
    for (int g0 = 0; g0 < std::max(P.groups, 1); g0 += gpw)
        words = std::max(words, (size_t)lds_layout(P.c, P.groups, g0, std::min(g0 + gpw, P.groups), split).words);
    return words * 4;
}
// what gg_create checks: the sweep must fit in LDS at least when every work-group takes a single ring group per side
size_t sweep_lds_bytes(const Params &P) { return parts_lds_bytes(P, 1); }
size_t sweep_xchg_entries(const Params &P) { return (size_t)xchg_entries(P.groups); }

// Would launch_sweep give this launch the plain k_sweep -- one work-group per cloud, no split steps, nothing forced?  (What a launch of FRESH
// maps needs, gg_internal.h Arena::gp_bits: decided before k_patch runs.)
bool sweep_takes_fresh(const Arena &a, const Params &P_in, int n_clouds)
{
    if (P_in.rings <= 0 || n_clouds <= SWEEP_PAIR_MAX_CLOUDS || a.tune_sweep_gpw || a.tune_sweep_split == 1 || a.tune_sweep_fault || a.tune_sweep_pair == 4) return false;
    // (launch_sweep's shape, below)
    Params P = P_in;
    const int n_groups = std::max(P.groups, 1);
    int n_parts = std::max(1, std::min(n_groups, SWEEP_LATENCY_MAX_CLOUDS / std::max(n_clouds, 1)));
    P.gpw = (n_groups + n_parts - 1) / n_parts;
    while (P.gpw > 1 && parts_lds_bytes(P, P.gpw) > 158 * 1024) --P.gpw;
    n_parts = (n_groups + P.gpw - 1) / P.gpw;
    const bool split = P.gpw == 1 && a.tune_sweep_split != 2 && n_clouds * n_parts <= SWEEP_LATENCY_MAX_CLOUDS && parts_lds_bytes(P, 1, true) <= 158 * 1024;
    return !split; // (one work-group per cloud or several: the chains are the same)
}

void launch_sweep(const Arena &a, const Params &P_in, const CloudParams *d_params, int n_clouds, hipStream_t s, unsigned long long *dbg)
{
    if (n_clouds == 0 || P_in.rings <= 0) return;
    // latency launches: the pair sweep (sweep_pair.h) -- both sides of a ring hand-over in one wavefront, old values from records
    // (a context whose k_sweep shape is forced -- wavefronts per side, groups per work-group, split steps, a fault to inject -- means k_sweep)
    const bool k_sweep_forced = a.tune_sweep_waves || a.tune_sweep_gpw || a.tune_sweep_split || a.tune_sweep_fault;
    // (the pair sweep on the layer in place, sweep_pairb.h / k4b_sweep_pair_batch.hip, takes a launch only when asked to -- tuning
    // sweep_pair = 4: correct on every geometry, but at 1024 clouds per launch 1.12 ms where k_sweep takes 0.89 ms)
    if (!dbg && a.tune_sweep_pair != 2 && !k_sweep_forced) {
        if (a.tune_sweep_pair == 4 && launch_sweep_pair_batch(a, P_in, d_params, n_clouds, s)) return;
        if (a.tune_sweep_pair != 4 && launch_sweep_pair(a, P_in, d_params, n_clouds, s)) return;
    }
    Params P = P_in;
    // Work-groups ("parts", sweep_core.h) per cloud.  The sweep of one cloud is a dependency chain, so a launch that leaves CUs idle
    // spreads every cloud over as many work-groups as there are CUs to take them (measured, k_sweep per launch: n = 1000, 1 / 8
    // clouds 1.33 -> 0.87 ms with one 64-ring group per work-group; 128 clouds 1.37 -> 1.19 ms with two work-groups per cloud;
    // n = 364, 1 .. 64 clouds 0.335 -> 0.32 ms); a launch that fills the chip anyway keeps one work-group per cloud -- more only add
    // importer / exporter wavefronts.  A work-group never takes more than three groups per side and wavefront set (16 wavefronts).
    const int n_groups = std::max(P.groups, 1);
    int n_parts = std::max(1, std::min(n_groups, SWEEP_LATENCY_MAX_CLOUDS / std::max(n_clouds, 1)));
    P.gpw = (n_groups + n_parts - 1) / n_parts;
    if (a.tune_sweep_gpw > 0) P.gpw = std::min(a.tune_sweep_gpw, n_groups);
    while (P.gpw > 1 && parts_lds_bytes(P, P.gpw) > 158 * 1024) --P.gpw; // (big maps: the hand-over tables of many groups do not fit in one LDS)
    n_parts = (n_groups + P.gpw - 1) / P.gpw;
    const int groups_per_part = std::min(P.gpw, n_groups);
    // Latency setting: a launch of at most one work-group per CU gives every 64-ring group of a side its own wavefront (up to 3)
    if (a.tune_sweep_waves > 0)
        P.waves_per_side = std::max(1, std::min(std::min(groups_per_part, 3), a.tune_sweep_waves));
    else if (n_clouds * n_parts <= SWEEP_LATENCY_MAX_CLOUDS)
        P.waves_per_side = std::max(1, std::min(groups_per_part, 3));
    else
        P.waves_per_side = std::min(P.waves_per_side, std::max(1, std::min(groups_per_part, 3)));
    // Split steps (sweep_core.h): with one ring group per work-group there are wavefront slots left; a preparing wavefront per side
    // takes the layer half of every step off the chain wavefront (tune_sweep_split: 0 auto, 1 on where possible, 2 off)
    P.split_steps = (P.gpw == 1 && a.tune_sweep_split != 2 && (a.tune_sweep_split == 1 || n_clouds * n_parts <= SWEEP_LATENCY_MAX_CLOUDS) &&
                     parts_lds_bytes(P, 1, true) <= 158 * 1024)
                        ? 1
                        : 0;
    if (P.split_steps) P.waves_per_side = 1;
    const size_t lds = parts_lds_bytes(P, P.gpw, P.split_steps != 0);
    static PerDeviceOnce big_lds;
    if (lds > 64 * 1024)
        big_lds.run([] {
            // (dynamic + the kernel's 16 static bytes must stay within the CU's 160 KiB, or the attribute is refused)
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_sweep<false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
        });
    P.fresh_cell = a.gp_fresh_cell;
    P.poll_cap = a.tune_sweep_poll_cap > 0 ? a.tune_sweep_poll_cap : 1 << 22; // (x ~0.2 us: about a second)
    P.debug_fault = a.tune_sweep_fault;
    const int threads = P.split_steps ? 12 * 64 : (4 * P.waves_per_side + 2 + (n_parts > 1 ? 2 : 0)) * 64; // (+ importer and exporter)
    if (dbg) // (GG_SWEEP_TIMING: the instrumented twin)
        hipLaunchKernelGGL((k_sweep<true, true>), dim3(n_clouds * n_parts), dim3(threads), lds, s, a, P, d_params, n_clouds, n_parts, dbg);
    else if (a.fresh_launch && !P.split_steps && n_parts == 1)
        hipLaunchKernelGGL((k_sweep<false, false, true>), dim3(n_clouds), dim3(threads), lds, s, a, P, d_params, n_clouds, n_parts, dbg);
    else if (a.fresh_launch && !P.split_steps)
        hipLaunchKernelGGL((k_sweep<false, true, true>), dim3(n_clouds * n_parts), dim3(threads), lds, s, a, P, d_params, n_clouds, n_parts, dbg);
    else if (n_parts > 1)
        hipLaunchKernelGGL((k_sweep<false, true>), dim3(n_clouds * n_parts), dim3(threads), lds, s, a, P, d_params, n_clouds, n_parts, dbg);
    else
        hipLaunchKernelGGL((k_sweep<false, false>), dim3(n_clouds * n_parts), dim3(threads), lds, s, a, P, d_params, n_clouds, n_parts, dbg);
}

} // namespace gg
