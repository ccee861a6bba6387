// K4 for throughput launches -- spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465) as the pair sweep of
// sweep_pairb.h on gfx950: one kernel, one work-group per cloud.
//
//   k_sweep_pair_batch   per cloud: per pair of sides (A/D, B/C) up to three wavefronts that take the 32-ring groups in turn -- lanes 0..31
//                        side X, lanes 32..63 side Y of the same rings, every join a lane exchange -- plus the two corner wavefronts.
//                        A wave-step is: two layer loads (the arriving cells of the own and of the outer line, requested PFB steps
//                        ahead; the layer's shear makes each half's 32 cells one 256-byte run), the decay of the visited cell's
//                        confidence, a wave shift of the (confidence, product) pairs, at a join step a half swap, the 8 packed additions
//                        of Eigen's two trees, one IEEE division, the blend, one product, one 8-byte store of the finished cell in place.
//                        The waits are feed-forward only (the corner values at a chain's first step, the last ring of the group inside).
//
// What this launch is bound by is instruction issue -- a few work-groups per CU, every wavefront a dependent chain of ~500 steps --, so
// the step is written for instruction count: no half steps, no closed-form wait tests (k_sweep: ~300 instructions per step and side; here
// one step serves two sides).
#include "gg_device.h"
#include "sweep_pairb.h"

#include <algorithm>

namespace gg {

using namespace sweep;
namespace sp = sweep::pair;

namespace {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(3))) uint64_t lds_u64;

struct PairBMem {
    __amdgpu_buffer_rsrc_t layer; // the interleaved (ground, confidence) layer of this cloud, read and rewritten in place
    lds_int *lds;
    static constexpr uint32_t OOR = 0x80000000u; // beyond the buffer: loads return 0, stores are dropped, no traffic
    GG_DEV uint32_t lds_addr(int word) const { return (uint32_t)(uintptr_t)lds + 4u * (uint32_t)word; }
    GG_DEV Cell load_issue(bool valid, int cell) const
    {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(layer, valid ? (uint32_t)cell * 8u : OOR, 0, 0);
        return Cell{__uint_as_float(v.x), __uint_as_float(v.y)};
    }
    GG_DEV Cell load_value(const Cell &queued, bool, int) const { return queued; }
    GG_DEV void store(bool valid, int cell, Cell v) const
    {
        u32x2 d;
        d.x = __float_as_uint(v.g);
        d.y = __float_as_uint(v.w);
        __builtin_amdgcn_raw_buffer_store_b64(d, layer, valid ? (uint32_t)cell * 8u : OOR, 0, 0);
    }
    // LDS.  Other wavefronts write what is read here: every access is an atomic (relaxed, work-group scope) or a volatile instruction, so
    // that the compiler neither caches nor moves it; ordering comes from the hardware (one wavefront's LDS operations execute in order)
    GG_DEV WP lds_wp(int word) const
    {
        const uint64_t u = __hip_atomic_load((lds_u64 *)(lds + word), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return WP{__uint_as_float((uint32_t)u), __uint_as_float((uint32_t)(u >> 32))};
    }
    GG_DEV void lds_put_wp(int word, WP v) const
    {
        const uint64_t u = (uint64_t)__float_as_uint(v.w) | ((uint64_t)__float_as_uint(v.p) << 32);
        __hip_atomic_store((lds_u64 *)(lds + word), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    GG_DEV int lds_i(int word) const { return __hip_atomic_load(lds + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    GG_DEV void lds_set(int word, int v) const { __hip_atomic_store(lds + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    // an entry (w, tag, p, tag) by LDS byte address: one 16-byte instruction each way (sweep_pairb.h LdsB: each 8-byte half carries its
    // own tag, so it does not matter whether the 16 bytes of a lane travel at once)
    GG_DEV void entry_write_at(uint32_t addr, WP v) const
    {
        const u32x4 d{__float_as_uint(v.w), 1u, __float_as_uint(v.p), 1u};
        asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(d) : "memory");
    }
    GG_DEV u32x4 entry_read_at(uint32_t addr) const
    {
        u32x4 d;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(addr) : "memory");
        return d;
    }
    // a lane's entry, waited for: all lanes read (those that take nothing read their scratch entry, tags preset)
    GG_DEV WP entry_await(uint32_t addr) const
    {
        u32x4 d = entry_read_at(addr);
        while (__builtin_expect(__any((d.y & d.w) == 0u), 0)) { // the producer is less than a step ahead
            __builtin_amdgcn_s_sleep(1);
            d = entry_read_at(addr);
        }
        return WP{__uint_as_float(d.x), __uint_as_float(d.z)};
    }
};

GG_DEV float wave_shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), 0x138, 0xF, 0xF, false)); }
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
    return wave_shr1(__uint_as_float(sw[1]));
}

template <int PAIR> GG_DEV void run_pairb(const Params &P, const sp::Plan &pl, const sp::LdsB &L, PairBMem &mem, int w, int W, int lane, WP centre)
{
    static_assert((int)sp::WARMUP == 2 && (int)sp::PTRIP % 4 == 0 && 4 % (int)sp::PFB == 0, "a group's steps are whole trips of four from t = -2: residue and queue slot are constants of the unrolled loop");
    constexpr int x_join = PAIR == sp::PAIR_AD ? 2 : 3, y_join = (x_join + 2) & 3;
    sp::PairLaneB<PAIR> st;
    for (int group_ = w; group_ < pl.groups; group_ += W) {
        const int group = __builtin_amdgcn_readfirstlane(group_);
        const sp::Group G = sp::group_of(PAIR, group, P.rings);
        st.init(lane, group, G, P, pl, L);
        // ---- per-lane constants: LDS byte addresses of the entry of step t = base + 16 t (lanes that take / publish nothing: their scratch entry)
        const bool l0 = st.l == 0, has_prev = group > 0, has_next = group + 1 < pl.groups;
        const uint32_t scr_a = mem.lds_addr(st.scr);
        const uint32_t imp_a = mem.lds_addr(st.a_bnd), exp_a = mem.lds_addr(st.pb >= 0 ? st.pb : st.scr);
        const uint32_t jl_a = mem.lds_addr(st.a_jl);
        const int imp_lo = st.start, imp_n = (l0 && has_prev && st.len > 2) ? st.len - 2 : 0; // the lane imports at wave-steps [imp_lo, imp_lo + imp_n)
        const int exp_lo = st.start, exp_n = st.pb >= 0 ? st.len : 0;                          // ... publishes at [exp_lo, exp_lo + exp_n)
        // ---- wave-uniform ranges of the events
        const int sy = sp::start0(PAIR, false);
        const int len_x0 = sp::len_of(sp::side_x(PAIR), G.r0), len_y0 = sp::len_of(sp::side_y(PAIR), G.r0);
        const int t_start_last = 2 * (G.nl - 1) + sy; // lanes take their corner values up to here
        const int imp_any_hi = has_prev ? max(len_x0 - 2, sy + len_y0 - 2) : -1000;
        const int t_jl = has_prev ? len_x0 - 2 : -1000; // X lane 0 takes its join from LDS here
        const int e0 = 2 * ((int)sp::HALF - 1);        // the last lane of X starts here
        const int len_x31 = sp::len_of(sp::side_x(PAIR), G.r0 + (int)sp::HALF - 1), len_y31 = sp::len_of(sp::side_y(PAIR), G.r0 + (int)sp::HALF - 1);
        const int exp_any_lo = has_next ? e0 : 1 << 30, exp_any_hi = has_next ? max(e0 + len_x31, e0 + sy + len_y31) : -1000;
        int have_ab = 0, have_cd = 0;
        const int t_end = G.t_first + G.steps;
        st.prime(G.t_first, mem);
        for (int tb = G.t_first; tb < t_end; tb += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = tb + u;
                constexpr int dummy = 0;
                (void)dummy;
                const int res4 = (u + 2) & 3; // t = u - 2 (mod 4)
                // ---- first steps (even t): the corner wavefronts' values, once they cover the lanes that start now
                bool first = false;
                WP c0{0.f, 0.f}, c1{0.f, 0.f};
                if ((u & 1) == 0 && t >= 0 && t <= t_start_last) { // (uniform)
                    const int lx = t >> 1, ly = (t - sy) >> 1;
                    const int need_ab = lx < G.nl ? G.r0 + lx : 0, need_cd = (ly >= 0 && ly < G.nl) ? G.r0 + ly : 0;
                    if (__builtin_expect(have_ab < need_ab || have_cd < need_cd, 0)) {
                        for (;;) {
                            have_ab = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 0));
                            have_cd = __builtin_amdgcn_readfirstlane(mem.lds_i(L.cnt_corner + 1));
                            if (have_ab >= need_ab && have_cd >= need_cd) break;
                            __builtin_amdgcn_s_sleep(2);
                        }
                    }
                    first = st.first_at(t);
                    const WP cpred = mem.lds_wp(st.a_pred);
                    c0 = mem.lds_wp(st.a_s0);
                    c1 = mem.lds_wp(st.a_s1);
                    st.pre(first, cpred);
                }
                // ---- S[s + 2]: lane - 1's result of two steps ago; the first lane of a half reads the group inside
                WP x{wave_shr1(st.h2.w), wave_shr1(st.h2.p)};
                if (t >= 0 && t < imp_any_hi) { // (uniform)
                    const bool mine = (unsigned)(t - imp_lo) < (unsigned)imp_n;
                    const WP e = mem.entry_await(mine ? imp_a + 16u * (uint32_t)t : scr_a);
                    x = mine ? e : x;
                }
                // ---- the join: the partner half's last result
                WP j{0.f, 0.f};
                if (res4 == x_join) {
                    j = WP{partner_for_x(st.h1.w), partner_for_x(st.h1.p)};
                    if (!has_prev) j = st.jl_lane ? centre : j; // (group 0: the join of ring 1 of side B is the centre cell)
                    if (t == t_jl) {                            // (uniform, once per group) X lane 0: Y's last value of the ring inside
                        const WP e = mem.entry_await(st.jl_lane ? jl_a : scr_a);
                        j = st.jl_lane ? e : j;
                    }
                } else if (res4 == y_join) {
                    j = WP{partner_for_y(st.h1.w), partner_for_y(st.h1.p)};
                }
                const int slot = u % (int)sp::PFB;
                WP res;
                if (u == 0) res = st.template step<0>(t, slot, x, j, first, c0, c1, P, mem);
                else if (u == 1) res = st.template step<1>(t, slot, x, j, first, c0, c1, P, mem);
                else if (u == 2) res = st.template step<2>(t, slot, x, j, first, c0, c1, P, mem);
                else res = st.template step<3>(t, slot, x, j, first, c0, c1, P, mem);
                // ---- the last lanes publish for the group outside
                if (t >= exp_any_lo && t < exp_any_hi) // (uniform)
                    mem.entry_write_at((unsigned)(t - exp_lo) < (unsigned)exp_n ? exp_a + 16u * (uint32_t)t : scr_a, res);
                // B_1 of ring 1 for the CD corner wavefront (the B chain of ring 1 is one visit, at wave-step 0 of group 0)
                if (PAIR == sp::PAIR_BC && !has_prev && t == 0) mem.entry_write_at(lane == 0 ? mem.lds_addr(L.b1) : scr_a, res);
            }
        }
    }
}

template <int CD> GG_DEV void run_pairb_corner(const Params &P, const sp::LdsB &L, PairBMem &mem, int lane, WP centre)
{
    sp::CornerLaneB<CD> st;
    sp::CornerHeld held{0, 0, 0.f, 0.f, 0.f, 0.f, false};
    float in_corner = centre.p, in_x1 = 0.f;
    for (int r0 = 1; r0 <= P.rings; r0 += 64) {
        const int nl = min(P.rings - (r0 - 1), 64);
        st.init(r0 + lane, P, mem); // the batch's old cells, all at once ...
        held.flush(mem);            // ... and only then the cells of the batch before (sweep_pairb.h CornerHeld)
        for (int l = 0; l < nl; ++l) {
            if (CD && r0 + l == 1) { // B_1 of ring 1, from the B/C pair's first wavefront
                u32x4 e = mem.entry_read_at(mem.lds_addr(L.b1));
                while ((e.y & e.w) == 0u) {
                    __builtin_amdgcn_s_sleep(1);
                    e = mem.entry_read_at(mem.lds_addr(L.b1));
                }
                in_x1 = __uint_as_float(__builtin_amdgcn_readfirstlane(e.z));
            }
            float x1g, x1, y0g, y0;
            st.c.visits(in_corner, in_x1, x1g, x1, y0g, y0);
            st.c.keep(lane == l, x1g, y0g);
            in_corner = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y0), l));
            in_x1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x1), l));
            const float wn1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.c.R.wn[1]), l));
            const float wn2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st.c.R.wn[2]), l));
            if (lane == 0) sp::CornerLaneB<CD>::publish(r0 + l, WP{wn1, in_x1}, WP{wn2, in_corner}, P, L, mem); // (uniform values: one lane writes)
        }
        held = st.hold(lane < nl);
    }
    held.flush(mem);
}

// wavefronts: 2 W chain wavefronts (id & 1 = the pair, id >> 1 = the wavefront of the pair), then the two corner wavefronts
__global__ __launch_bounds__(512) void k_sweep_pair_batch(const Arena a, const Params P, const sp::Plan pl, const CloudParams *__restrict__ params, int W)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    const int cloud = (int)blockIdx.x;
    const sp::LdsB L = sp::ldsb_of(P.c, pl, true);
    const CloudParams &cp = params[cloud];
    float2 *gp2 = gp2_ptr(a, cp.slot);
    float *percall = percall_ptr(a, cp.slot);
    const int nthreads = blockDim.x;
    const WP centre{1.0f, 1.0f * cp.base_z}; // :405 groundpatch(centre) = 1, :406-411 ground(centre) = translation.z
    // hand-over tables start empty (tags 0, counters 0 = "ring 0 done"); the scratch entries read as published
    for (int k = threadIdx.x; k < L.words; k += nthreads) lds[k] = (k >= L.scratch && k < L.bnd && ((k - L.scratch) & 1)) ? 1 : 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        gp2[gp_index(P.gl, P.c, P.c)] = make_float2(cp.base_z, 1.0f);
        float *f = reinterpret_cast<float *>(lds);
        for (int cd = 0; cd < 2; ++cd) { // ring 0 of the corner tables = the centre cell
            f[sp::cornerb_word(L, P.c, cd, 0, 1)] = centre.w;
            f[sp::cornerb_word(L, P.c, cd, 0, 1) + 1] = centre.p;
        }
    }
    // :147 map["points"].setConstant(0.0) -- as k_sweep: only the half columns of tiles that received records hold anything but 0
    if (!P.keep_points) {
        const uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
        const int lane_ = threadIdx.x & 63, wave_ = (int)(threadIdx.x >> 6), nwaves = nthreads >> 6;
        for (int rank = wave_; rank < a.g.T; rank += nwaves) {
            const uint32_t cols_live = tile_live[rank];
            if (!cols_live) continue; // (uniform)
            float *points = percall + percall_index(rank, PL_POINTS, 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int cell = lane_ + 64 * k;
                if ((cols_live >> live_bit(cell)) & 1u) points[cell] = 0.0f;
            }
        }
    }
    __syncthreads(); // the only barrier of the sweep

    PairBMem mem;
    mem.layer = __builtin_amdgcn_make_buffer_rsrc(gp2, 0, P.gl.elems * 8, 0x00020000);
    mem.lds = (lds_int *)lds;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63u);
    if (wave < 2 * W) {
        if ((wave & 1) == 0) run_pairb<sp::PAIR_AD>(P, pl, L, mem, wave >> 1, W, lane, centre);
        else run_pairb<sp::PAIR_BC>(P, pl, L, mem, wave >> 1, W, lane, centre);
    } else if (wave == 2 * W)
        run_pairb_corner<0>(P, L, mem, lane, centre);
    else if (wave == 2 * W + 1)
        run_pairb_corner<1>(P, L, mem, lane, centre);
}

} // namespace

// returns false when the launch cannot take the throughput pair sweep (the caller falls back to k_sweep)
bool launch_sweep_pair_batch(const Arena &a, const Params &P, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (P.rings <= 0) return false;
    static thread_local sp::Plan pl;
    static thread_local int pl_rings = -1;
    if (pl_rings != P.rings) {
        pl = sp::make_plan(P.rings);
        pl_rings = P.rings;
    }
    if (pl.groups <= 0) return false;
    const sp::LdsB L = sp::ldsb_of(P.c, pl, true);
    const size_t lds = (size_t)L.words * 4;
    if (lds > 158 * 1024) return false;
    int W = std::min(pl.groups, 3);
    if (a.tune_sweep_pair_waves > 0) W = std::max(1, std::min(W, a.tune_sweep_pair_waves));
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)k_sweep_pair_batch, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256); // (idempotent; big maps only)
    hipLaunchKernelGGL(k_sweep_pair_batch, dim3(n_clouds), dim3((2 * W + 2) * 64), lds, s, a, P, pl, d_params, W);
    return true;
}

} // namespace gg
