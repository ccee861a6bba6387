// sort_core.h -- device-side pieces of the stable tile sort between K1 and K2, shared by the stand-alone kernels (k_sort.hip)
// and the fused front end (k1_classify.hip, Arena::tune_front):
//
//   scan_cloud     one work-group: hist[chunk][tile] (one row per wave-chunk, written by K1) -> exclusive offsets in
//                  (tile-major, chunk-minor) order, tile_start[], K2's two work lists (light / dense tiles), the cleared
//                  liveness masks of tiles without records, and the exclusive prefixes of the per-chunk emission counters
//                  (kept / ignored / outliers) that give every point its position in the returned cloud (K5).
//   scatter_chunk  one wavefront: re-walks its chunk in cloud order and places record p at offset[tile] + (number of earlier
//                  points of the chunk in that tile).  Ranks inside a 64-point window come from ballots, the running offset
//                  of a tile from ONE returning LDS add per distinct tile and window (issued by the tile's first lane, handed
//                  to the others with a lane permute) -- the LDS unit executes a wavefront's operations in order, so the adds
//                  of several windows are in flight together and the sort is STABLE: inside a tile, and therefore inside every
//                  cell, records stay in cloud order, which is what makes the float32 Welford recurrence of K2
//                  bit-reproducible (src/GroundSegmentation.cpp:296-305 is order dependent).
//
// Rows of `hist` cross work-groups INSIDE a launch when the scan is fused into K1 (the last work-group of a cloud to finish
// scans for all of them) and again when the scatter is fused as well (everybody waits for that scan): every access to them is
// a 16-byte agent-scope (sc1) buffer access -- written through to memory, read past the L1 -- so no fence is needed and nothing
// depends on which XCD a work-group runs on (MI355X_MICROARCH.md "Inter-workgroup visibility", recipe R1).
#pragma once

#include "gg_device.h"

namespace gg {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int AUX_SC1 = 16; // cache-policy bits of the raw buffer builtins on gfx940+: sc0 = 1, nt = 2, sc1 = 16

GG_DEV __amdgpu_buffer_rsrc_t words_rsrc(const uint32_t *base, size_t words)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, (int)(words * 4), 0x00020000);
}
GG_DEV u32x4 load16_agent(__amdgpu_buffer_rsrc_t r, uint32_t word) { return __builtin_amdgcn_raw_buffer_load_b128(r, word * 4u, 0, AUX_SC1); }
GG_DEV void store16_agent(__amdgpu_buffer_rsrc_t r, uint32_t word, u32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(v, r, word * 4u, 0, AUX_SC1); }
GG_DEV void drain_vector_memory() { __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory"); } // (inline asm: the compiler cannot drop it)

// wave-level inclusive scan (64 lanes)
GG_DEV uint32_t wave_inclusive_scan(uint32_t v, int lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(v, d, 64);
        if (lane >= d) v += o;
    }
    return v;
}

// block-level exclusive scan of one value per thread (blockDim.x = 64 NW), returns the exclusive prefix; total in `total`
template <int NW>
GG_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /*[NW + 1]*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    __syncthreads(); // protect lds reuse across calls
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t w = (lane < NW) ? lds[lane] : 0u;
        const uint32_t winc = wave_inclusive_scan(w, lane);
        if (lane < NW) lds[lane] = winc - w;
        if (lane == NW - 1) lds[NW] = winc;
    }
    __syncthreads();
    total = lds[NW];
    return lds[wave] + inc - v;
}

// The scan of one cloud by one work-group of 64 NW threads.  A thread owns FOUR consecutive tiles (one 16-byte column segment
// of every chunk's row); `nch` = the cloud's chunks.  AGENT: the rows were written by other work-groups of THIS launch (fused
// front end): agent-scope accesses; as a launch of its own the scan uses plain ones (the rows are still in the L2).
// `part` != nullptr (64 NW entries): when there are fewer tile groups than threads, the threads of a tile group share the cloud's
// chunks among them -- a thread's loads are one dependent memory round trip per unrolled batch, and the scan of ONE cloud
// (latency launches) is as long as that chain.
template <bool AGENT> GG_DEV u32x4 load16_row(__amdgpu_buffer_rsrc_t r, uint32_t word)
{
    return AGENT ? load16_agent(r, word) : __builtin_amdgcn_raw_buffer_load_b128(r, word * 4u, 0, 0);
}
template <bool AGENT> GG_DEV void store16_row(__amdgpu_buffer_rsrc_t r, uint32_t word, u32x4 v)
{
    if (AGENT) store16_agent(r, word, v);
    else __builtin_amdgcn_raw_buffer_store_b128(v, r, word * 4u, 0, 0);
}

// PARTS (k_scan only, maps with many tiles and few clouds per launch): the cloud's tile groups are cut into `n_parts` consecutive
// ranges, one work-group each (`part` = a ticket taken when the work-group started: whoever waits, waits for work-groups that are
// already running, in any dispatch order).  A part sums its own columns, leaves (records, light tiles, dense tiles) of its range
// in ONE 64-bit word of `sync` (valid bit 63: no second flag, nothing to order), waits -- bounded -- for the words of the parts
// before it, and goes on from their sum.  The hand-over costs every part one agent-scope round trip in a kernel that is two
// passes over a 4 MB histogram (n = 1000).  `sync` == nullptr: one work-group does it all.
constexpr uint32_t SCAN_WAIT_POLLS = 1u << 24; // x s_sleep(16): several seconds.  A part only ever waits for parts that are running or done (tickets), so the
                                               // bound is a backstop against a lost work-group, not a scheduling assumption: a work-group that is merely
                                               // preempted or time-sliced on a shared GPU must not trip it.  Tests shorten it (Arena::tune_scan_poll_cap).
// the hand-over word of a part: bits 0..31 records, 32..46 light tiles, 47..61 dense tiles, 63 valid.  A part covers at most one
// round of the work-group = 1024 tile groups of 4 tiles (k_sort.hip scan_parts), so both tile counts stay below 2^15.
constexpr int SCAN_PART_MAX_GROUPS = 1024;
static_assert(4 * SCAN_PART_MAX_GROUPS < (1 << 15), "the light / dense tile counts of a scan part must fit their 15-bit fields");
template <int NW, bool AGENT>
GG_DEV void scan_cloud(const Arena &a, const CloudParams &cp, int nch, uint32_t *lds /*[NW + 1]*/, u32x4 *part = nullptr /*[64 NW]*/,
                       int my_part = 0, int n_parts = 1, unsigned long long *sync = nullptr /*[n_parts]*/)
{
    constexpr int NT = 64 * NW;
    const int tid = threadIdx.x;
    const int T = a.g.T, TP = a.hist_pitch, G = TP / 4;
    const __amdgpu_buffer_rsrc_t hist = words_rsrc(a.hist + (size_t)cp.slot * a.hist_stride, a.hist_stride);
    uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;
    uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
    uint4 *tile_list = a.tile_list + (size_t)cp.slot * a.tile_list_stride;
    // this work-group's tile groups [G_lo, G_hi) (a part's range fits one round: launch_scan)
    const int Gp = (G + n_parts - 1) / n_parts;
    const int G_lo = min(G, my_part * Gp), G_hi = min(G, G_lo + Gp);
    // GS tile groups per round, each shared by Q threads (chunk ranges); thread -> (share q, group gi)
    const int GS = part ? min(NT, (max(G_hi - G_lo, 1) + 63) & ~63) : NT, Q = part ? NT / GS : 1;
    const int gi = tid % GS, q = tid / GS;
    const int c_lo = q < Q ? (int)((long long)nch * q / Q) : 0, c_hi = q < Q ? (int)((long long)nch * (q + 1) / Q) : 0;

    uint32_t carry = 0, lcarry = 0, dcarry = 0;
    for (int g0 = G_lo; g0 < G_hi || (sync && g0 == G_lo); g0 += GS) { // (a part without tile groups still says so)
        const int g = g0 + gi;
        const bool have = g < G_hi && q < Q;
        u32x4 s = {0u, 0u, 0u, 0u};
        if (have) {
#pragma unroll 16
            for (int c = c_lo; c < c_hi; ++c) s += load16_row<AGENT>(hist, (uint32_t)c * (uint32_t)TP + 4u * (uint32_t)g);
        }
        u32x4 before = {0u, 0u, 0u, 0u}; // the records of this group in the chunks before this thread's share
        if (part && Q > 1) {
            __syncthreads();
            part[tid] = s;
            __syncthreads();
            u32x4 all = {0u, 0u, 0u, 0u};
            for (int k = 0; k < Q; ++k) {
                const u32x4 v = part[k * GS + gi];
                if (k < q) before += v;
                all += v;
            }
            s = all; // (every thread of the group now holds the group's totals)
        }
        const bool lead = have && q == 0; // one thread per group speaks for it in the scans and writes the lists
        uint32_t total;
        const uint32_t excl_lead = block_exclusive_scan<NW>(lead ? s.x + s.y + s.z + s.w : 0u, lds, total);
        uint32_t excl = excl_lead;
        if (part && Q > 1) { // the group's prefix reaches its other threads through the same scratch
            __syncthreads();
            if (lead) part[gi].x = excl_lead;
            __syncthreads();
            excl = part[gi].x;
        }
        {
            // K2's work lists (k2_reduce.hip): tiles with more than K2_LIGHT_MAX records from the back of tile_list, the
            // other tiles with records from the front, both in rank order.  A tile without records is on neither list: its half
            // columns simply stop being live (the per-call layers are sparse, gg_internal.h tile_live) -- nothing is cleaned.
            uint32_t nl = 0, nd = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool in = lead && 4 * g + k < T;
                nd += (in && s[k] > (uint32_t)K2_LIGHT_MAX) ? 1u : 0u;
                nl += (in && s[k] > 0u && s[k] <= (uint32_t)K2_LIGHT_MAX) ? 1u : 0u;
            }
            uint32_t ltotal;
            const uint32_t lexcl = block_exclusive_scan<NW>(nl | (nd << 16), lds, ltotal);
            if (sync) { // (one round per part)
                if (tid == 0 && !(a.tune_scan_fault && my_part == 0)) // (tests: the first part withholds its word, the others' waits must run out)
                    __hip_atomic_store(&sync[my_part], (unsigned long long)total | ((unsigned long long)(ltotal & 0x7FFFu) << 32) |
                                       ((unsigned long long)((ltotal >> 16) & 0x7FFFu) << 47) | (1ull << 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned long long w = 1ull << 63;
                if (tid < my_part) {
                    uint32_t polls = 0;
                    const uint32_t poll_cap = a.tune_scan_poll_cap > 0 ? (uint32_t)a.tune_scan_poll_cap : SCAN_WAIT_POLLS;
                    while (((w = __hip_atomic_load(&sync[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) >> 63) == 0ull && polls < poll_cap) {
                        __builtin_amdgcn_s_sleep(16);
                        ++polls;
                    }
                    if ((w >> 63) == 0ull) __hip_atomic_store(a.dev_error, (uint32_t)GG_DEVERR_SCAN_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                const bool mine = tid < my_part;
                uint32_t before_t, before_l;
                block_exclusive_scan<NW>(mine ? (uint32_t)w : 0u, lds, before_t);
                block_exclusive_scan<NW>(mine ? (uint32_t)((w >> 32) & 0x7FFFu) | ((uint32_t)((w >> 47) & 0x7FFFu) << 16) : 0u, lds, before_l);
                carry = before_t;
                lcarry = before_l & 0xFFFFu;
                dcarry = before_l >> 16;
            }
            uint32_t li = lcarry + (lexcl & 0xFFFFu), di = dcarry + (lexcl >> 16), start = carry + excl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * g + k;
                if (lead && t < T) {
                    // (rank, the tile's records, its first cell: K2 starts on a tile after ONE lookup)
                    const uint4 entry = make_uint4((uint32_t)t, start, start + s[k], a.rank_cell0[t]);
                    if (s[k] == 0u) tile_live[t] = 0u;
                    else if (s[k] > (uint32_t)K2_LIGHT_MAX) tile_list[(uint32_t)T - 1u - di++] = entry;
                    else tile_list[li++] = entry;
                    tile_start[t] = start;
                    start += s[k];
                }
            }
            lcarry += ltotal & 0xFFFFu;
            dcarry += ltotal >> 16;
        }
        if (have) { // every chunk's count becomes its first position in `sorted`
            u32x4 run;
            run.x = carry + excl;
            run.y = run.x + s.x;
            run.z = run.y + s.y;
            run.w = run.z + s.z;
            run += before;
#pragma unroll 16
            for (int c = c_lo; c < c_hi; ++c) {
                const uint32_t w = (uint32_t)c * (uint32_t)TP + 4u * (uint32_t)g;
                const u32x4 h = load16_row<AGENT>(hist, w);
                store16_row<AGENT>(hist, w, run);
                run += h;
            }
        }
        carry += total;
    }
    if (tid == 0 && my_part == n_parts - 1) { // (the last part knows the sums over all of them)
        tile_start[T] = carry;
        uint32_t *lc = a.tile_list_cnt + (size_t)cp.slot * 2;
        lc[0] = lcarry;
        lc[1] = dcarry;
    }
    if (my_part != 0) return; // (the part that waits for nobody also does the emission counters)

    // emission counters: exclusive prefix over chunks for each of the 4 categories (read by K5, the next kernel but two)
    const __amdgpu_buffer_rsrc_t ce = words_rsrc(a.chunk_emit + (size_t)cp.slot * a.emit_stride, a.emit_stride);
    uint32_t *ce_out = a.chunk_emit + (size_t)cp.slot * a.emit_stride;
    u32x4 kcarry = {0u, 0u, 0u, 0u};
    for (int c0 = 0; c0 < nch; c0 += NT) {
        const int c = c0 + tid;
        u32x4 v = {0u, 0u, 0u, 0u}, e;
        if (c < nch) v = load16_row<AGENT>(ce, 4u * (uint32_t)c);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint32_t total;
            e[k] = kcarry[k] + block_exclusive_scan<NW>(v[k], lds, total);
            kcarry[k] += total;
        }
        if (c < nch) *reinterpret_cast<u32x4 *>(ce_out + 4 * (size_t)c) = e;
    }
    if (tid == 0) *reinterpret_cast<u32x4 *>(a.totals + (size_t)cp.slot * 4) = kcarry;
}

// The stable scatter of one wave-chunk [base, end) of cloud records.  `offs` = the wavefront's LDS row (hist_pitch words, or
// half as many when PACKED): unpacked it holds the next free position of every tile (the caller loaded the chunk's row of
// `hist`), packed (maps of more than PACKED_TILE_COUNTERS_MIN_T tiles) only the number of records already placed per tile, two
// 16-bit counters per word, zeroed by the caller -- the chunk's first position per tile then stays in its row of `hist` and every
// lane fetches its own record's.  AGENT: the offsets were written by another work-group of this launch (fused front end).
template <bool PACKED, bool AGENT>
GG_DEV void scatter_chunk(uint32_t *offs, __amdgpu_buffer_rsrc_t hist, uint32_t row_word, const uint2 *__restrict__ rec, uint2 *__restrict__ sorted,
                          int base, int end, int lane)
{
    constexpr int SI = 4; // windows per batch
    if (base >= end) return;
    // The records of the NEXT batch are requested before this batch is placed (two register sets that take turns: a copy at the
    // loop's back edge would wait for the load where it is issued; unconditional loads at clamped indices: a load under a branch
    // makes every later wait a vmcnt(0)): a batch is a chain -- records, ballots, LDS add, permute, store -- and the first link
    // of the next one now travels during the others.
    auto request = [&](uint2 (&r)[SI], int p0) {
#pragma unroll
        // (streaming loads: the scatter reads a record once, and k_label, the other reader, comes five kernels later -- kept out of the
        // L2 they leave it to the `sorted` records k_reduce is about to read: k_scatter 0.443 -> 0.425 ms per 1024 clouds)
        for (int j = 0; j < SI; ++j) {
            typedef uint32_t u2_native __attribute__((ext_vector_type(2)));
            const u2_native v = __builtin_nontemporal_load(reinterpret_cast<const u2_native *>(rec) + min(p0 + 64 * j + lane, end - 1));
            r[j] = make_uint2(v.x, v.y);
        }
    };
    auto place = [&](uint2 (&r)[SI], int p0) {
        uint32_t dst[SI];
#pragma unroll
        for (int j = 0; j < SI; ++j) {
            if (p0 + 64 * j + lane >= end) r[j].y = KEY_OUTSIDE; // (the clamped lanes of the chunk's last batch)
            const bool inmap = r[j].y != KEY_OUTSIDE;
            const uint32_t t = r[j].y >> KEY_TILE_SHIFT;
            uint32_t first_pos = 0u;
            if (PACKED) first_pos = __builtin_amdgcn_raw_buffer_load_b32(hist, inmap ? (row_word + t) * 4u : 0xFFFFFFFFu, 0, AGENT ? AUX_SC1 : 0); // (out of range: 0)
            // rank among the window's records of the same tile, their number, and the tile's first lane: ballots only
            uint32_t rank = 0u, cnt = 0u;
            int first = lane;
            for (unsigned long long todo = __ballot(inmap); todo != 0ull;) { // one iteration per distinct tile (wave-uniform)
                const int leader = __builtin_ctzll(todo);
                const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);
                const bool mine = inmap && t == t0;
                const unsigned long long same = __ballot(mine);
                if (mine) {
                    rank = (uint32_t)rank_below(same);
                    cnt = (uint32_t)__popcll(same);
                    first = leader;
                }
                todo &= ~same;
            }
            // ... one returning LDS add per distinct tile (no two lanes of one instruction meet on an address); the adds of the
            // SI windows queue up behind each other in the LDS unit, in program order
            uint32_t old = 0u;
            if (inmap && lane == first) {
                if (PACKED) {
                    const uint32_t sh = (t & 1u) * 16u;
                    old = (__hip_atomic_fetch_add(&offs[t >> 1], cnt << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xFFFFu;
                } else {
                    old = __hip_atomic_fetch_add(&offs[t], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            dst[j] = first_pos + (uint32_t)__shfl((int)old, first, 64) + rank;
        }
#pragma unroll
        for (int j = 0; j < SI; ++j)
            if (r[j].y != KEY_OUTSIDE) sorted[dst[j]] = r[j];
    };
    uint2 ra[SI], rb[SI];
    request(ra, base);
    for (int p0 = base; p0 < end; p0 += 2 * 64 * SI) {
        request(rb, min(p0 + 64 * SI, end - 1));
        place(ra, p0);
        if (p0 + 64 * SI >= end) break;
        request(ra, min(p0 + 2 * 64 * SI, end - 1));
        place(rb, p0 + 64 * SI);
    }
}

} // namespace gg
