// K4 for latency launches -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465) as the pair sweep of
// sweep_pair.h on gfx950: three kernels.
//
//   k_sweep_records   one thread per chain visit (and per ring corner): everything the visit needs that is known before the sweep
//                     starts -- gvlSum, the two factors of :460, the new confidence, the products of the OLD window cells in tree
//                     position -- from the layer as k_patch left it, written in the order the chain wavefronts read it (a wave-step =
//                     64 consecutive records: three coalesced loads per lane and step).  No dependences, the whole chip at once.
//   k_sweep_pair      per cloud one work-group (both pairs) or two (pair A/D and pair B/C on two CUs, nothing shared): per pair one
//                     wavefront per 32-ring group -- lanes 0..31 side X, lanes 32..63 side Y, every join a lane exchange -- plus the two
//                     corner wavefronts.  A wave-step is: three record loads (queued PPF steps ahead), a wave shift and a half swap of
//                     the last results, four selects, the 8 additions of Eigen's tree, one IEEE division, the blend, one product, one
//                     4-byte store into the result stream.  The only waits are feed-forward (corner values at a chain's first step, the
//                     last ring of the group inside): a wavefront that has caught up with its producer stays behind it.
//   k_sweep_finish    one thread per layer element: the streamed height and the cell's own new confidence into the layer, coalesced (the
//                     64 cells of a wave-step lie in 64 different lines of the layer -- a scattered store costs the CU's memory front end
//                     ~165 cycles, more than the rest of a step, profiles/r04c/ubench_ta_lines.log; a stream store 11); also :147 and the
//                     centre cell (:405-411).
//
// k_sweep_pair never reads the layer (sweep_pair.h): no write-after-read hazard between its wavefronts or work-groups.
#include "gg_device.h"
#include "sweep_pair.h"

#include <algorithm>

namespace gg {

using namespace sweep;
namespace sp = sweep::pair;

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

// records of one cloud inside the scratch region (floats): quad blocks of 1280 floats (sweep_pair.h quad_word: four wave-steps x 64 lanes
// as five 16-byte words per lane) for total_steps / 4 + 4 quads (the queue reads three quads ahead), then 32 planes [2][ring_pad] of the
// corner records, then the result stream
struct RecLayout {
    size_t corner, ring_pad, out, floats; // (out: the chains' result stream, one float per record)
};
static __host__ __device__ RecLayout rec_layout(const sp::Plan &pl, int rings)
{
    RecLayout R;
    R.corner = ((size_t)pl.total_steps / 4 + 4) * sp::QUAD_BLOCK_FLOATS;
    R.ring_pad = ((size_t)rings + 1 + 63) / 64 * 64 + 64;
    R.out = R.corner + (size_t)sp::CORNER_REC_FLOATS * 2 * R.ring_pad;
    R.floats = R.out + ((size_t)pl.total_steps + 4) * 64;
    return R;
}
size_t sweep_pair_rec_floats(const Params &P)
{
    const sp::Plan pl = sp::make_plan(P.rings);
    if (pl.groups <= 0) return 0;
    return (rec_layout(pl, P.rings).floats + 63) / 64 * 64;
}

// ---------------------------------------------------------------------------------------------------------------------
// the preparation
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sweep_records(const Arena a, const Params P, const sp::Plan pl, const CloudParams *__restrict__ params)
{
    const CloudParams &cp = params[blockIdx.y];
    const float2 *gp2 = gp2_ptr(a, cp.slot);
    float *rec = a.sweep_rec + (size_t)blockIdx.y * a.sweep_rec_stride;
    const RecLayout RL = rec_layout(pl, P.rings);
    auto load = [&](int x, int y) {
        const float2 v = gp2[gp_index(P.gl, x, y)];
        return Cell{v.x, v.y};
    };
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n_visit = (size_t)pl.total_steps / 4 * 64; // one thread per (quad of wave-steps, lane): the four windows share their cells
    if (e < n_visit) {
        // (a wavefront = one half -- one side of the spiral: no divergence on it -- of two consecutive quads)
        const int half = (int)(e >> 6) & 1, lane = half * (int)sp::HALF + (int)(e & 31);
        const int step = ((int)(e >> 7) * 2 + ((int)(e >> 5) & 1)) * 4;
        int p = step >= pl.base[1][0] ? 1 : 0, g = 0;
        while (g + 1 < pl.groups && step >= pl.base[p][g + 1]) ++g; // (wave-uniform)
        const sp::Group G = sp::group_of(p, g, P.rings);
        const int t0 = G.t_first + (step - pl.base[p][g]);
        const bool is_x = lane < (int)sp::HALF;
        const int l = lane & (sp::HALF - 1), side = is_x ? sp::side_x(p) : sp::side_y(p);
        const int r = G.r0 + l, len = sp::len_of(side, r), s0 = t0 - (2 * l + sp::start0(p, is_x));
        if (l >= G.nl || s0 + 3 < -(int)sp::WARMUP || s0 > len) return; // (the chain lane is idle at these steps and uses nothing of the records)
        float q[sp::QUAD_FLOATS];
        sp::make_visit_quad(P, p, is_x, r, s0, load, q);
        // (steps of the quad outside [-WARMUP, len] come out as zeros: nobody reads them)
#pragma unroll
        for (int c = 0; c < 5; ++c) reinterpret_cast<float4 *>(rec + sp::quad_word(step, lane, c))[0] = make_float4(q[4 * c], q[4 * c + 1], q[4 * c + 2], q[4 * c + 3]);
        return;
    }
    const size_t k = e - n_visit;
    if (k >= 2 * (size_t)P.rings) return;
    const int cd = (int)(k / (size_t)P.rings), r = 1 + (int)(k % (size_t)P.rings);
    const sp::CornerRec R = cd ? sp::make_corner_rec<1>(P, r, load) : sp::make_corner_rec<0>(P, r, load);
    const float *f = reinterpret_cast<const float *>(&R);
#pragma unroll
    for (int i = 0; i < (int)sp::CORNER_REC_FLOATS; ++i) rec[RL.corner + ((size_t)i * 2 + cd) * RL.ring_pad + r] = f[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// the sweep
// ---------------------------------------------------------------------------------------------------------------------
struct PairMem {
    __amdgpu_buffer_rsrc_t layer; // the interleaved (ground, confidence) layer of this cloud: the corner wavefronts' stores only
    lds_int *lds;
    static constexpr uint32_t OOR = 0x80000000u;
    GG_DEV uint32_t lds_base() const { return (uint32_t)(uintptr_t)lds; }
    GG_DEV float lds_f(int word) const { return __int_as_float(__hip_atomic_load(lds + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)); }
    GG_DEV void lds_put(int word, float v) const { __hip_atomic_store(lds + word, __float_as_int(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    GG_DEV void lds_put2(int word, float v0, float v1) const
    {
        const uint64_t u = (uint64_t)__float_as_uint(v0) | ((uint64_t)__float_as_uint(v1) << 32);
        __hip_atomic_store((lds_u64 *)(lds + word), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV void lds_entry(int word, float v) const
    {
        const uint64_t u = (uint64_t)__float_as_uint(v) | (1ull << 32);
        __hip_atomic_store((lds_u64 *)(lds + word), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV uint64_t lds_entry_get(int word) const { return __hip_atomic_load((lds_u64 *)(lds + word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    // the same by LDS byte address (per-lane base + stride x step: one multiply-add)
    GG_DEV uint64_t lds_entry_get_at(uint32_t addr) const { return __hip_atomic_load((lds_u64 *)(uintptr_t)addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    GG_DEV void lds_entry_at(uint32_t addr, float v) const
    {
        const uint64_t u = (uint64_t)__float_as_uint(v) | (1ull << 32);
        __hip_atomic_store((lds_u64 *)(uintptr_t)addr, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV int lds_i(int word) const { return __hip_atomic_load(lds + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    GG_DEV void lds_set(int word, int v) const { __hip_atomic_store(lds + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    GG_DEV void store(bool valid, int cell, Cell v) const
    {
        u32x2 d;
        d.x = __float_as_uint(v.g);
        d.y = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b64(d, layer, valid ? (uint32_t)cell * 8u : OOR, 0, 0);
    }
};

GG_DEV float pair_wave_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false)); }
// The partner half's last result: X lane l <- Y lane l - 1 (lane 32 + l - 1), Y lane l <- X lane l.  v_permlane32_swap exchanges the upper
// half of its first operand with the lower half of its second: with both = h1, the first comes back as [X | X] and the second as [Y | Y].
GG_DEV float partner_for_y(float h1) // (meaningful in the upper half)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1), __float_as_uint(h1), false, false);
    return __uint_as_float(sw[0]);
}
GG_DEV float partner_for_x(float h1) // (meaningful in the lower half; lane 0 takes its join from LDS)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1), __float_as_uint(h1), false, false);
    return pair_wave_shr1(__uint_as_float(sw[1]));
}
GG_DEV float partner_both(float h1)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h1), __float_as_uint(h1), false, false);
    return __int_as_float(__builtin_amdgcn_update_dpp((int)sw[0], (int)sw[1], 0x138 /* wave_shr:1 */, 0x3 /* rows 0, 1 */, 0xF, false));
}

// Timing experiments only (tools/build_pair_variant.sh; results void): what a step costs without one of its parts
//   -DGG_PAIR_X_NODIV    a multiplication instead of the IEEE division       -DGG_PAIR_X_NOSTORE  no store into the result stream
//   -DGG_PAIR_X_NOLDS    no import / export / join through LDS (nor waits)    -DGG_PAIR_X_NOLOAD   the record queue is never refilled
//   -DGG_PAIR_X_NOPERM   no exchange of the halves' last results
// three quads of records in flight per lane (sweep_pair.h quad_word): slot j of a trip holds the trip's j-th quad.  A slot is refilled
// while the NEXT quad runs, one 16-byte word per step (two in the first: a burst of five loads behind one step stalled the wavefront at
// the memory pipeline's door) -- every word at least 7 steps before its first use
struct RecQueue {
    u32x4 c[3][5];
};
GG_DEV void quad_request(RecQueue &Q, int slot, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t quad_soff)
{
#pragma unroll
    for (int k = 0; k < 5; ++k) Q.c[slot][k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024u * (uint32_t)k, quad_soff, 0);
}
GG_DEV void word_request(RecQueue &Q, int slot, int k, __amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t quad_soff)
{
    Q.c[slot][k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024u * (uint32_t)k, quad_soff, 0);
}
// the record of step u of the trip (u = 12: the next trip's first, already in slot 0)
GG_DEV sp::VisitRec quad_rec(const RecQueue &Q, int u)
{
    const int slot = (u >> 2) % 3, i = u & 3;
    sp::VisitRec R;
    R.gvl = __uint_as_float(Q.c[slot][i].x), R.a = __uint_as_float(Q.c[slot][i].y), R.b = __uint_as_float(Q.c[slot][i].z), R.wn = __uint_as_float(Q.c[slot][i].w);
    R.nU = __uint_as_float(Q.c[slot][4][i]);
    return R;
}

// Which of a lane's rare events a wave-step can hold follows from t mod 4 alone: X(r) ends at 4 l + 2 r0 + b - 1 (r0 = 1 mod 32), so the
// join step of side X (the step before its last) falls on t = b (mod 4), its last step one later, side Y's two steps later.  A trip of the
// unrolled loop is 12 steps from t = -2 (mod 12): the residue is a constant of every step.
template <int PAIR> struct Residue {
    static constexpr int x_join = PAIR == sp::PAIR_AD ? 2 : 3, x_last = (x_join + 1) & 3, y_join = (x_join + 2) & 3, y_last = (x_join + 3) & 3;
};
// what the twelve steps of a trip are compiled for.
//   STARTS  0 no lane takes its first step; 1 tested per step, the corner table read where a lane starts (the generic trip); 2 the corner
//           values of every lane that starts inside the trip are read ONCE before it (the corner wavefronts run far ahead: the trip waits
//           until they cover its last ring) and the first step is three selects
//   IMP     the first lanes of the halves read the last ring of the group inside: 0 never, 1 tested per step with the LDS round trip in
//           the step (the generic trip: also the step in which X lane 0 takes its join from LDS), 2 every step of the trip, both halves,
//           3 by per-lane range (trips at the edges of the import range); 2 and 3 request the entry a step ahead
//   EXP     the last lanes publish for the group outside: 0 / 1 / 2 / 3 alike (where no group follows, 2 writes scratch words)
template <int S, int I, int E> struct PairKind {
    static constexpr int starts = S, imp = I, exp = E;
};
enum { PAIR_TRIP = 12 };

// a tagged LDS entry requested early and awaited late: the round trip runs behind a step's arithmetic.  (The compiler keeps relaxed
// atomics in program order and sank a plain prefetch to just before the next LDS operation; between `issue` and `wait` the destination
// is touched by nothing.)
GG_DEV void entry_issue(uint32_t addr, u32x2 &dst) { asm volatile("ds_read_b64 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); }
// (NEWER: LDS operations the wavefront has issued since -- the step's publish; the queue returns in order)
template <int NEWER> GG_DEV void entry_wait(u32x2 &dst)
{
    if (NEWER == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(dst)::"memory");
    else asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(dst)::"memory");
}

template <int PAIR, int PF>
GG_DEV void run_pair(const Params &P, const sp::Plan &pl, const sp::Lds &L, PairMem &mem, __amdgpu_buffer_rsrc_t rec, __amdgpu_buffer_rsrc_t out, int w, int W, int lane, float centre_p,
                     unsigned long long *dbg)
{
    unsigned long long wait_corner = 0, wait_import = 0, n_wait = 0; // (tools: GG_PAIR_TIMING)
    int gi = 0;
    static_assert((int)sp::PTRIP == (int)PAIR_TRIP && PAIR_TRIP == 3 * (int)sp::QUAD, "a trip is three quads of records; a step's residue and queue slot are constants of the unrolled loop");
    (void)PF;
    sp::PairLane<PAIR> st;
    const bool is_x = lane < (int)sp::HALF;
    for (int group_ = w; group_ < pl.groups; group_ += W) {
        const int group = __builtin_amdgcn_readfirstlane(group_); // (wave-uniform, and the compiler must know: it becomes scalar offsets)
        const sp::Group G = sp::group_of(PAIR, group, P.rings);
        st.init(lane, group, G, P, pl, L);
        const int step0 = __builtin_amdgcn_readfirstlane(pl.base[PAIR][group]); // first wave-step record of the group
        const uint32_t voff = (uint32_t)lane * 16u;
        constexpr uint32_t QUAD_BYTES = (uint32_t)sp::QUAD_BLOCK_FLOATS * 4u;
        RecQueue Q;
#pragma unroll
        for (int k = 0; k < 2; ++k) quad_request(Q, k, rec, voff, (uint32_t)(step0 / 4 + k) * QUAD_BYTES); // (the third slot fills during the first quad)
        Q.c[2][0] = Q.c[2][1] = Q.c[2][2] = Q.c[2][3] = Q.c[2][4] = u32x4{0u, 0u, 0u, 0u};
        // ---- per-lane constants of the fast trips
        const uint32_t lds0 = mem.lds_base();
        const bool l0 = st.l == 0;
        const uint32_t imp_a = lds0 + 4u * (uint32_t)((l0 && group > 0) ? st.a_bnd : st.scr), imp_k = (l0 && group > 0) ? 8u : 0u; // entry of step t at imp_a + imp_k t
        const uint32_t exp_a = lds0 + 4u * (uint32_t)(st.pb >= 0 ? st.pb : st.scr), exp_k = st.pb >= 0 ? 8u : 0u;
        const int endm = st.start + st.len; // the lane's join step is t + 2 == endm, its last t + 1 == endm
        // ---- wave-uniform ranges
        const int sy = sp::start0(PAIR, false);
        const int len_x0 = sp::len_of(sp::side_x(PAIR), G.r0), len_y0 = sp::len_of(sp::side_y(PAIR), G.r0);
        const int t_start_last = 2 * (G.nl - 1) + sy; // lanes take their corner values up to here
        const bool has_prev = group > 0, has_next = group + 1 < pl.groups;
        const int imp_all_lo = sy, imp_all_hi = min(len_x0 - 2, sy + len_y0 - 2), imp_any_hi = has_prev ? max(len_x0 - 2, sy + len_y0 - 2) : -1000;
        const int t_jl = has_prev ? len_x0 - 2 : -1000; // X lane 0 takes its join from LDS here
        const int e0 = 2 * ((int)sp::HALF - 1);        // the last lane of X starts here
        const int len_x31 = sp::len_of(sp::side_x(PAIR), G.r0 + (int)sp::HALF - 1), len_y31 = sp::len_of(sp::side_y(PAIR), G.r0 + (int)sp::HALF - 1);
        const int exp_all_lo = e0 + sy, exp_all_hi = min(e0 + len_x31, e0 + sy + len_y31), exp_any_lo = has_next ? e0 : 1 << 30, exp_any_hi = has_next ? max(e0 + len_x31, e0 + sy + len_y31) : -1000;
        int have_ab = 0, have_cd = 0;
        const int t_end = G.t_first + G.steps;
        if (dbg && lane == 0 && gi < 4) dbg[2 + gi * 6 + 0] = __builtin_readcyclecounter();
        wait_corner = wait_import = n_wait = 0;

        // per-lane import / export ranges (ranged kinds): the lane imports at wave-steps [imp_lo, imp_lo + imp_n), publishes at [exp_lo, exp_lo + exp_n)
        const int imp_lo = st.start, imp_n = (l0 && has_prev && st.len > 2) ? st.len - 2 : 0;
        const int exp_lo = st.start, exp_n = st.pb >= 0 ? st.len : 0;
        const uint32_t scr_a = lds0 + 4u * (uint32_t)st.scr;
        auto trip = [&](const int tb, auto kind) __attribute__((always_inline)) {
            using K = decltype(kind);
#ifdef GG_PAIR_X_NOLDS
            constexpr int kimp = 0, kexp = 0;
#else
            constexpr int kimp = K::imp, kexp = K::exp;
#endif
            u32x2 ent_q{0u, 0u};
            uint32_t imp_cur = imp_a + imp_k * (uint32_t)tb, exp_cur = exp_a + exp_k * (uint32_t)tb;
            auto imp_addr = [&](int t) { return (kimp == 2 || (unsigned)(t - imp_lo) < (unsigned)imp_n) ? imp_cur : scr_a; };
            if (kimp >= 2) entry_issue(imp_addr(tb), ent_q);
            // scalar byte offsets of the quad block two quads ahead (the refill of the slot behind the running quad) and of the step's row of
            // the result stream
            uint32_t rec_soff = (uint32_t)((step0 + (tb - G.t_first)) / 4 + 2) * QUAD_BYTES;
            uint32_t out_soff = (uint32_t)(step0 + (tb - G.t_first)) * 256u; // (a row of the result stream = 4 steps x 64 lanes x 4 bytes)
            u32x4 g4{0u, 0u, 0u, 0u};
            float cs0 = 0.f, cs1 = 0.f, cpred = 0.f;
            if (K::starts == 2) { // the corner values of the lanes that start in [tb, tb + 12): once, up front
                const int te = tb + (int)PAIR_TRIP;
                const int lx = min(G.nl - 1, (te - 1) >> 1), ly = min(G.nl - 1, (te - 1 - sy) >> 1);
                const int need_ab = G.r0 + lx, need_cd = ly >= 0 ? G.r0 + ly : 0;
                if (__builtin_expect(have_ab < need_ab || have_cd < need_cd, 0)) {
                    const unsigned long long w0 = dbg ? __builtin_readcyclecounter() : 0ull;
                    for (;;) {
                        have_ab = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 0));
                        have_cd = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 1));
                        if (have_ab >= need_ab && have_cd >= need_cd) break;
                        __builtin_amdgcn_s_sleep(2);
                    }
                    if (dbg) wait_corner += __builtin_readcyclecounter() - w0;
                }
                cpred = mem.lds_f(st.a_pred);
                cs0 = mem.lds_f(st.a_s0);
                cs1 = mem.lds_f(st.a_s1);
            }
#pragma unroll
            for (int u = 0; u < (int)PAIR_TRIP; ++u) {
                const int t = tb + u;
                const int res4 = (u + 2) & 3;     // t = -2 + u (mod 4)
                const bool t_even = (u & 1) == 0; // (lanes start at even t only)
                // ---- the records of this step and of the next (requested at least 8 steps ago)
                const sp::VisitRec R = quad_rec(Q, u), Rn = quad_rec(Q, u + 1);
#ifndef GG_PAIR_X_NOLOAD
                {   // refill the slot of the quad that ended before this one: words 0 and 4 first (the next quad's first step needs them), then 1, 2, 3
                    const int fs = ((u >> 2) + 2) % 3, fi = u & 3;
                    if (fi == 0) word_request(Q, fs, 4, rec, voff, rec_soff);
                    word_request(Q, fs, fi, rec, voff, rec_soff);
                    if (fi == 3) rec_soff += QUAD_BYTES;
                }
#endif
                // ---- first steps: the corner values
                bool first = false;
                if (K::starts == 1 && t >= 0 && t <= t_start_last) { // (uniform) the generic trip: read where a lane starts, poll if the cached counters do not cover it
                    first = st.first_at(t);
                    const int have = st.cd ? have_cd : have_ab;
                    if (__builtin_expect(__any(first && have < st.r), 0)) {
                        const unsigned long long w0 = dbg ? __builtin_readcyclecounter() : 0ull;
                        for (;;) {
                            have_ab = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 0));
                            have_cd = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 1));
                            if (!__any(first && (st.cd ? have_cd : have_ab) < st.r)) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (dbg) wait_corner += __builtin_readcyclecounter() - w0;
                    }
                    cpred = mem.lds_f(st.a_pred);
                    cs0 = mem.lds_f(st.a_s0);
                    cs1 = mem.lds_f(st.a_s1);
                    st.h1 = first ? cpred : st.h1;
                } else if (K::starts == 2 && t_even) {
                    first = st.first_at(t);
                    st.h1 = first ? cpred : st.h1;
                }
                // ---- S[s + 2]: lane - 1's result of two steps ago; the first lane of a half reads the group inside
                float x = pair_wave_shr1(st.h2);
                if (kimp >= 2) {
                    if (u >= 1 && kexp >= 2) entry_wait<1>(ent_q); // (requested a step ago; the one LDS operation since is that step's publish)
                    else entry_wait<0>(ent_q);
                    u32x2 ent = ent_q;
                    if (__builtin_expect(__any(ent.y == 0u), 0)) { // (rare) the group inside is less than a step ahead
                        const unsigned long long w0 = dbg ? __builtin_readcyclecounter() : 0ull;
                        const uint32_t addr = imp_addr(t);
                        do {
                            __builtin_amdgcn_s_sleep(1);
                            const uint64_t e64 = mem.lds_entry_get_at(addr);
                            ent.x = (uint32_t)e64;
                            ent.y = (uint32_t)(e64 >> 32);
                        } while (__any(ent.y == 0u));
                        if (dbg) wait_import += __builtin_readcyclecounter() - w0, ++n_wait;
                    }
                    const bool mine = kimp == 2 ? l0 : (unsigned)(t - imp_lo) < (unsigned)imp_n;
                    imp_cur += imp_k;
                    if (u + 1 < (int)PAIR_TRIP) entry_issue(imp_addr(t + 1), ent_q);
                    x = mine ? __uint_as_float(ent.x) : x;
                } else if (kimp == 1 && t < imp_any_hi) { // (uniform)
                    const bool imp = st.imports_at(t, group);
                    const int word = imp ? st.import_entry(t) : st.scr;
                    uint64_t ent = mem.lds_entry_get(word);
                    while (__builtin_expect(__any((uint32_t)(ent >> 32) == 0u), 0)) {
                        __builtin_amdgcn_s_sleep(1);
                        ent = mem.lds_entry_get(word);
                    }
                    x = imp ? __uint_as_float((uint32_t)ent) : x;
                }
                // ---- the join / the far end: one kind of event per residue of t
                if (res4 == Residue<PAIR>::x_join || res4 == Residue<PAIR>::y_join) {
#ifdef GG_PAIR_X_NOPERM
                    float j = st.h2;
#else
                    float j = res4 == Residue<PAIR>::x_join ? partner_for_x(st.h1) : partner_for_y(st.h1);
#endif
                    if ((kimp == 1 || kimp == 3) && res4 == Residue<PAIR>::x_join) {
                        if (kimp == 1 && l0 && is_x) j = centre_p; // (group 0: the join of ring 1 of side B is the centre cell)
                        if (t == t_jl) {                           // (uniform, once per group) X lane 0: Y's last value of the ring inside
                            const int word = st.jl_lane ? st.a_jl : st.scr;
                            uint64_t ent = mem.lds_entry_get(word);
                            while (__builtin_expect(__any((uint32_t)(ent >> 32) == 0u), 0)) {
                                __builtin_amdgcn_s_sleep(1);
                                ent = mem.lds_entry_get(word);
                            }
                            j = st.jl_lane ? __uint_as_float((uint32_t)ent) : j;
                        }
                    }
                    x = t + 2 == endm ? j : x;
                } else {
                    x = t + 1 == endm ? Rn.gvl : x; // (an OLD cell at the far end: it travels in the record behind the chain's last)
                }
                st.I0 = st.I1;
                st.I1 = st.I2;
                st.I2 = x;
                if (K::starts == 1 && t >= 0 && t <= t_start_last) {
                    st.I0 = first ? cs0 : st.I0;
                    st.I1 = (first && st.len != 1) ? cs1 : st.I1;
                } else if (K::starts == 2 && t_even) {
                    st.I0 = first ? cs0 : st.I0;
                    st.I1 = first ? cs1 : st.I1; // (chains of one visit -- ring 1 -- start in the generic trip)
                }
                // ---- the visit; its height goes to the result stream (idle lanes write slots nobody reads)
                const sp::OldWindow O{R.b, Rn.b, st.U1, st.U2, R.nU};
                st.U1 = st.U2;
                st.U2 = R.nU;
                const float g = sp::height_of(R.gvl, R.a, R.b, sp::window_sum<PAIR>(is_x, O, st.I0, st.I1, st.I2, st.h1));
                const float res = R.wn * g;
                g4[u & 3] = __float_as_uint(g);
                if ((u & 3) == 3) { // four heights of a lane = 16 contiguous bytes, one store
#ifndef GG_PAIR_X_NOSTORE
#ifdef GG_PAIR_X_STORE4
                    for (int k4 = 0; k4 < 4; ++k4) __builtin_amdgcn_raw_buffer_store_b32(g4[k4], out, (uint32_t)lane * 16u + 4u * k4, out_soff, 0);
#else
                    // (the row's offset goes into the VECTOR offset on purpose.  A 16-byte buffer store whose offset sits in a scalar register
                    // is still reading its data registers when the next instruction issues -- a select that reused g4's first register
                    // put LDS addresses into the stream, on some boxes, in some lanes -- and the compiler only keeps writers away from the
                    // data of stores WITHOUT a scalar offset register: tools/pair_race.py found it)
                    __builtin_amdgcn_raw_buffer_store_b128(g4, out, (uint32_t)lane * 16u + out_soff, 0, 0);
#endif
#endif
                    out_soff += 1024u;
                }
                st.h2 = st.h1;
                st.h1 = res;
                if (kexp == 2) {
                    mem.lds_entry_at(exp_cur, res);
                } else if (kexp == 3) {
                    mem.lds_entry_at((unsigned)(t - exp_lo) < (unsigned)exp_n ? exp_cur : scr_a, res);
                } else if (kexp == 1 && t >= exp_any_lo && t < exp_any_hi) { // (uniform)
                    const bool active = (unsigned)(t - st.start) < (unsigned)st.len;
                    mem.lds_entry((st.pb >= 0 && active) ? st.pb + 2 * t : st.scr, res);
                }
                exp_cur += exp_k;
            }
        };
        for (int tb = G.t_first; tb < t_end; tb += PAIR_TRIP) {
            const int te = tb + (int)PAIR_TRIP; // the trip is [tb, te)
            const bool no_start = tb > t_start_last, has_jl = t_jl >= tb && t_jl < te;
            const bool imp2 = !has_prev || (tb >= imp_all_lo && te <= imp_all_hi), imp0 = tb >= imp_any_hi && tb > t_jl; // (no group inside: every lane reads its scratch word)
            const bool exp2 = !has_next || (tb >= exp_all_lo && te <= exp_all_hi);
            static_assert((int)sp::WARMUP == 2, "the residues and the parity of the first steps count from t = -2");
            (void)has_jl; // (the ranged kinds take X lane 0's join from LDS where it falls; a trip of all-importing steps ends before it)
            if (group == 0 && tb == G.t_first) trip(tb, PairKind<1, 1, 1>{}); // ring 1: chains of one visit, the centre as a join
            else if (!no_start) trip(tb, PairKind<2, 3, 3>{});
            else if (imp0 && exp2) trip(tb, PairKind<0, 0, 2>{});
            else if (imp2 && exp2) trip(tb, PairKind<0, 2, 2>{});
            else trip(tb, PairKind<0, 3, 3>{});
            if (dbg && lane == 0 && gi < 4 && tb <= t_start_last && tb + (int)PAIR_TRIP > t_start_last) dbg[2 + gi * 6 + 1] = __builtin_readcyclecounter();
        }
        if (dbg && lane == 0 && gi < 4) {
            dbg[2 + gi * 6 + 2] = __builtin_readcyclecounter();
            dbg[2 + gi * 6 + 3] = wait_corner;
            dbg[2 + gi * 6 + 4] = wait_import;
            dbg[2 + gi * 6 + 5] = n_wait;
        }
        ++gi;
    }
}

template <int CD>
GG_DEV void run_pair_corner(const Params &P, const sp::Plan &pl, const sp::Lds &L, PairMem &mem, const float *__restrict__ rec, const RecLayout &RL, int lane, float centre_p, unsigned long long *dbg)
{
    sp::CornerLane<CD> st;
    float in_corner = centre_p, in_x1 = 0.f;
    for (int r0 = 1; r0 <= P.rings; r0 += 64) {
        const int nl = min(P.rings - (r0 - 1), 64);
        sp::CornerRec R;
        float *f = reinterpret_cast<float *>(&R);
        const int ring = min(r0 + lane, P.rings);
#pragma unroll
        for (int i = 0; i < (int)sp::CORNER_REC_FLOATS; ++i) f[i] = rec[RL.corner + ((size_t)i * 2 + CD) * RL.ring_pad + ring];
        st.init(r0 + lane, P, R);
        if (dbg && lane == 0 && r0 / 64 < 8) dbg[2 + 2 * (r0 / 64)] = __builtin_readcyclecounter(); // (the batch's records are here ...)
        for (int l = 0; l < nl; ++l) {
            if (CD && r0 + l == 1) { // B_1 of ring 1, from the AB corner wavefront
                uint64_t ent = mem.lds_entry_get(L.b1);
                while ((uint32_t)(ent >> 32) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    ent = mem.lds_entry_get(L.b1);
                }
                in_x1 = __uint_as_float((uint32_t)ent);
            }
            float x1g, x1, y0g, y0;
            st.visits(in_corner, in_x1, x1g, x1, y0g, y0);
            st.keep(lane == l, x1g, y0g);
#ifdef GG_PAIR_X_FLUSHEACH
            st.flush(lane == l, mem);
#endif
            in_corner = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y0), l));
            in_x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), l));
            if (lane == 0) sp::CornerLane<CD>::publish(r0 + l, in_x1, in_corner, P, L, mem); // (uniform values: one lane writes)
            if (!CD && r0 + l == 1) { // the one chain visit the CD corner needs: B_1(1) = pair B/C, group 0, lane 0, wave-step 0
                const int st0 = pl.base[sp::PAIR_BC][0] + (0 - sp::group_of(sp::PAIR_BC, 0, P.rings).t_first); // lane 0's wave-step 0
                const float b1 = sp::b1_of_ring1(sp::rec_at(rec, st0 - 2, 0), sp::rec_at(rec, st0 - 1, 0), sp::rec_at(rec, st0, 0), sp::rec_at(rec, st0 + 1, 0), in_x1, in_corner, centre_p);
                if (lane == 0) mem.lds_entry(L.b1, b1);
            }
        }
        st.flush(lane < nl, mem); // the batch's cells, one store per lane
        if (dbg && lane == 0 && r0 / 64 < 8) dbg[3 + 2 * (r0 / 64)] = __builtin_readcyclecounter(); // (... and its rings are done)
    }
}

// BOTH: one work-group per cloud runs both pairs; else two work-groups per cloud (blockIdx & 1 = the pair)
// PF: wave-steps a record is requested ahead (10 registers each); THREADS: the launch bound that leaves the registers for it
template <bool BOTH, int PF, int THREADS>
__global__ __launch_bounds__(THREADS) void k_sweep_pair(const Arena a, const Params P, const sp::Plan pl, const CloudParams *__restrict__ params, int W)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int cloud = BOTH ? (int)blockIdx.x : (int)(blockIdx.x >> 1), which = BOTH ? 0 : (int)(blockIdx.x & 1u);
    const sp::Lds L = sp::lds_of(P.c, pl, BOTH);
    const CloudParams &cp = params[cloud];
    float2 *gp2 = gp2_ptr(a, cp.slot);
    const int nthreads = blockDim.x;
    const float centre_p = 1.0f * cp.base_z; // :405 groundpatch(centre) = 1, :406-411 ground(centre) = translation.z
    for (int k = threadIdx.x; k < L.words; k += nthreads) lds[k] = (k >= L.scratch && k < L.bnd && ((k - L.scratch) & 1)) ? 1 : 0; // (scratch entries: tag preset)
    __syncthreads();
    if (threadIdx.x == 0) {
        float *f = reinterpret_cast<float *>(lds);
        f[sp::corner_word(L, P.c, 0, 0, 1)] = centre_p;
        f[sp::corner_word(L, P.c, 1, 0, 1)] = centre_p;
    }
    __syncthreads();

    PairMem mem;
    mem.layer = __builtin_amdgcn_make_buffer_rsrc(gp2, 0, P.gl.elems * 8, 0x00020000);
    mem.lds = (lds_int *)lds;
    const float *rec = a.sweep_rec + (size_t)cloud * a.sweep_rec_stride;
    const RecLayout RL = rec_layout(pl, P.rings);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    const int n_chain = BOTH ? 2 * W : W;
    unsigned long long *dbg = (a.pair_dbg && cloud == 0) ? a.pair_dbg + ((size_t)which * 16 + wave) * 32 : nullptr;
    if (dbg && lane == 0) dbg[0] = __builtin_readcyclecounter();
    if (wave < n_chain) {
        // wavefront ids go to the CU's four SIMDs round-robin.  The corner wavefronts (ids n_chain, n_chain + 1) work while the inner groups
        // do; the two chain wavefronts that share their SIMDs take the LAST groups, which start when the corners are nearly done
        int p = BOTH ? (wave & 1) : which, w = BOTH ? (wave >> 1) : wave;
#ifndef GG_PAIR_X_NOREMAP
        if (!BOTH && W >= 4 && ((n_chain & 3) == 2)) { // ids 2, 3 share SIMDs with the corners (n_chain = 2 mod 4): swap them with the last two
            if (w == 2 || w == 3) w = W - 4 + w;         // ... wavefronts of the pair
            else if (w >= W - 2) w = w - (W - 4);
        }
#endif
        const __amdgpu_buffer_rsrc_t rrec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec), 0, (int)(RL.corner * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(rec) + RL.out, 0, (int)((RL.floats - RL.out) * 4), 0x00020000);
        if (p == 0) run_pair<sp::PAIR_AD, PF>(P, pl, L, mem, rrec, rout, w, W, lane, centre_p, dbg);
        else run_pair<sp::PAIR_BC, PF>(P, pl, L, mem, rrec, rout, w, W, lane, centre_p, dbg);
    } else if (wave == n_chain)
        run_pair_corner<0>(P, pl, L, mem, rec, RL, lane, centre_p, dbg);
    else if (wave == n_chain + 1)
        run_pair_corner<1>(P, pl, L, mem, rec, RL, lane, centre_p, dbg);
    if (dbg && lane == 0) dbg[1] = __builtin_readcyclecounter();
}

// the streamed heights into the layer: one thread per layer element (coalesced; padding elements hold no cell), plus :147 and the centre
__global__ __launch_bounds__(256) void k_sweep_finish(const Arena a, const Params P, const sp::Plan pl, const CloudParams *__restrict__ params)
{
    const CloudParams &cp = params[blockIdx.y];
    float2 *gp2 = gp2_ptr(a, cp.slot);
    const float *rec = a.sweep_rec + (size_t)blockIdx.y * a.sweep_rec_stride;
    const RecLayout RL = rec_layout(pl, P.rings);
    const int e = (int)(blockIdx.x * 256u + threadIdx.x);
    if (e < P.gl.elems && ((a.gp_valid[e >> 5] >> (e & 31)) & 1u)) {
        int x, y, slot;
        if (e == 0) gp2[0] = make_float2(cp.base_z, 1.0f); // :405 groundpatch(centre) = 1, :406-411 ground(centre) = translation.z
        else if (gp_cell_of(P.gl, e, x, y) && sp::chain_slot_of_cell(P, pl, x, y, slot)) {
            const float2 old = gp2[e];
            const Cell v = sp::finished_cell(P, x, y, old.y, rec[RL.out + (size_t)slot]);
            gp2[e] = make_float2(v.g, v.w);
        }
    }
    // :147 map["points"].setConstant(0.0) (k4_sweep.hip: only the live half columns hold anything but 0): one wavefront per tile
    if (!P.keep_points) {
        const uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
        float *percall = percall_ptr(a, cp.slot);
        const int lane_ = threadIdx.x & 63, wave_ = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), nwaves = (int)gridDim.x * 4;
        for (int rank = wave_; rank < a.g.T; rank += nwaves) {
            const uint32_t cols_live = tile_live[rank];
            if (!cols_live) continue;
            float *points = percall + percall_index(rank, PL_POINTS, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = lane_ + 64 * k;
                if ((cols_live >> live_bit(cell)) & 1u) points[cell] = 0.0f;
            }
        }
    }
}

// returns false when the launch cannot take the pair sweep (the caller falls back to k_sweep)
bool launch_sweep_pair(const Arena &a, const Params &P, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (!a.sweep_rec || n_clouds > a.sweep_rec_clouds || P.rings <= 0) return false;
    static thread_local sp::Plan pl;
    static thread_local int pl_rings = -1;
    if (pl_rings != P.rings) {
        pl = sp::make_plan(P.rings);
        pl_rings = P.rings;
    }
    if (pl.groups <= 0) return false;
    const bool both = a.tune_sweep_pair_wgs == 1;
    const sp::Lds L = sp::lds_of(P.c, pl, both);
    const size_t lds = (size_t)L.words * 4;
    if (lds > 158 * 1024) return false;
    // at most 8 wavefronts per work-group: 256 registers per lane hold the record queue and a trip's five variants without spilling (the
    // 16-wavefront shape spilled 250 registers and was slower than fewer wavefronts taking several groups in turn: n = 1000, one cloud,
    // 14 wavefronts per pair 0.54 ms, 6 wavefronts 0.43 ms)
    const int max_w = both ? 3 : 6;
    int W = std::min(pl.groups, max_w);
    if (a.tune_sweep_pair_waves > 0) W = std::max(1, std::min(W, a.tune_sweep_pair_waves));
    const RecLayout RL = rec_layout(pl, P.rings);
    if (RL.floats > a.sweep_rec_stride) return false;
    const int waves = (both ? 2 * W : W) + 2;
    const void *fn = both ? (const void *)k_sweep_pair<true, 12, 512> : (const void *)k_sweep_pair<false, 12, 512>;
    if (lds > 64 * 1024) hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); // (idempotent; big maps only)
    const size_t n_rec = (size_t)pl.total_steps / 4 * 64 + 2 * (size_t)P.rings;
    hipLaunchKernelGGL(k_sweep_records, dim3((unsigned)((n_rec + 255) / 256), (unsigned)n_clouds), dim3(256), 0, s, a, P, pl, d_params);
    const dim3 grid(both ? n_clouds : 2 * n_clouds), block(waves * 64);
    if (both) hipLaunchKernelGGL((k_sweep_pair<true, 12, 512>), grid, block, lds, s, a, P, pl, d_params, W);
    else hipLaunchKernelGGL((k_sweep_pair<false, 12, 512>), grid, block, lds, s, a, P, pl, d_params, W);
    hipLaunchKernelGGL(k_sweep_finish, dim3((unsigned)((P.gl.elems + 255) / 256), (unsigned)n_clouds), dim3(256), 0, s, a, P, pl, d_params);
    return true;
}

} // namespace gg
