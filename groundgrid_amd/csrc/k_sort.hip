// Tile sort, between K1 and K2: a stable counting sort of the in-map points by Morton tile rank.
//
//   k_scan    : one work-group per cloud.  hist[chunk][tile] (written by K1, one row per wave-chunk)
//               -> exclusive offsets in (tile-major, chunk-minor) order, tile_start[], and the
//               exclusive prefixes of the per-chunk emission counters (kept / ignored / outliers)
//               that give every point its position in the returned cloud (K5); and K2's two work lists (light / dense tiles).
//   k_scatter : same wave <-> chunk mapping as K1.  Each wave re-walks its chunk in cloud order and
//               places record p at offset[tile] + (number of earlier points of the chunk in that tile):
//               ranks inside a 64-point window come from ballots (deterministic, no atomics), so the
//               sort is STABLE -- inside a tile, and therefore inside every cell, records stay in cloud
//               order, which is what makes the float32 Welford recurrence of K2 bit-reproducible
//               (src/GroundSegmentation.cpp:296-305 is order dependent).
#include "gg_device.h"

namespace gg {

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

// block-level exclusive scan of one value per thread (blockDim.x = 1024), returns exclusive prefix; total in `total`
GG_DEV uint32_t block_exclusive_scan(uint32_t v, uint32_t *lds /*[17]*/, uint32_t &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t inc = wave_inclusive_scan(v, lane);
    __syncthreads(); // protect lds reuse across calls
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const uint32_t w = (lane < 16) ? lds[lane] : 0u;
        const uint32_t winc = wave_inclusive_scan(w, lane);
        if (lane < 16) lds[lane] = winc - w;
        if (lane == 15) lds[16] = winc;
    }
    __syncthreads();
    total = lds[16];
    return lds[wave] + inc - v;
}

__global__ __launch_bounds__(1024) void k_scan(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ uint32_t lds[17];
    const int cloud = blockIdx.x;
    const CloudParams cp = params[cloud];
    const int T = a.g.T;
    const int nch = (cp.n_points + a.PW - 1) / a.PW;
    uint32_t *hist = a.hist + (size_t)cp.slot * a.hist_stride;
    uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;

    uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
    uint4 *tile_list = a.tile_list + (size_t)cp.slot * a.tile_list_stride;
    uint32_t carry = 0, lcarry = 0, dcarry = 0;
    for (int t0 = 0; t0 < T; t0 += 1024) {
        const int t = t0 + (int)threadIdx.x;
        uint32_t s = 0;
        if (t < T) {
#pragma unroll 16
            for (int c = 0; c < nch; ++c) s += hist[(size_t)c * T + t];
        }
        uint32_t total;
        const uint32_t excl = block_exclusive_scan(s, lds, total);
        {
            // K2's work lists (k2_reduce.hip): tiles with more than K2_LIGHT_MAX records from the back of tile_list, the
            // other tiles with records from the front.  A tile without records is on neither list: its half columns simply stop
            // being live (the per-call layers are sparse, gg_internal.h tile_live) -- nothing is cleaned.
            const bool dense = t < T && s > (uint32_t)K2_LIGHT_MAX;
            const bool light = t < T && !dense && s > 0u;
            if (t < T && s == 0u) tile_live[t] = 0u;
            uint32_t ltotal;
            const uint32_t lexcl = block_exclusive_scan((light ? 1u : 0u) | (dense ? 0x10000u : 0u), lds, ltotal);
            // (rank, the tile's records, its first cell: K2 starts on a tile after ONE lookup)
            const uint4 entry = make_uint4((uint32_t)t, carry + excl, carry + excl + s, t < T ? a.rank_cell0[t] : 0u);
            if (light) tile_list[lcarry + (lexcl & 0xFFFFu)] = entry;
            if (dense) tile_list[(uint32_t)T - 1u - (dcarry + (lexcl >> 16))] = entry;
            lcarry += ltotal & 0xFFFFu;
            dcarry += ltotal >> 16;
        }
        if (t < T) {
            uint32_t running = carry + excl;
            tile_start[t] = running;
#pragma unroll 16
            for (int c = 0; c < nch; ++c) {
                const uint32_t h = hist[(size_t)c * T + t];
                hist[(size_t)c * T + t] = running;
                running += h;
            }
        }
        carry += total;
    }
    if (threadIdx.x == 0) {
        tile_start[T] = carry;
        uint32_t *lc = a.tile_list_cnt + (size_t)cp.slot * 2;
        lc[0] = lcarry;
        lc[1] = dcarry;
    }

    // emission counters: exclusive prefix over chunks for each of the 4 categories
    uint32_t *ce = a.chunk_emit + (size_t)cp.slot * a.emit_stride;
    uint32_t *totals = a.totals + (size_t)cp.slot * 4;
    for (int k = 0; k < 4; ++k) {
        uint32_t kcarry = 0;
        for (int c0 = 0; c0 < nch; c0 += 1024) {
            const int c = c0 + (int)threadIdx.x;
            const uint32_t v = (c < nch) ? ce[(size_t)c * 4 + k] : 0u;
            uint32_t total;
            const uint32_t excl = block_exclusive_scan(v, lds, total);
            if (c < nch) ce[(size_t)c * 4 + k] = kcarry + excl;
            kcarry += total;
        }
        if (threadIdx.x == 0) totals[k] = kcarry;
    }
}

void launch_scan(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    hipLaunchKernelGGL(k_scan, dim3(n_clouds), dim3(1024), 0, s, a, d_params);
}

__global__ __launch_bounds__(256) void k_scatter(const Arena a, const CloudParams *__restrict__ params)
{
    extern __shared__ uint32_t lds_offs[]; // [4][T]
    // XCD-aware (gg_device.h): the chunks of one cloud run on one XCD, so its layers / records are cached in ONE L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x), bx = (int)(item % gridDim.x);
    const CloudParams cp = params[cloud];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = bx * 4 + wave;
    const int n = cp.n_points;
    const int nch = (n + a.PW - 1) / a.PW;
    if (chunk >= nch) return;

    const int T = a.g.T;
    // Small maps: the wavefront's running offset per tile, absolute, in LDS.  Big maps (PACKED_TILE_COUNTERS_MIN_T): only the
    // number of records already placed per tile, two 16-bit counters per word; the chunk's start per tile stays in its row of
    // `hist`, and every lane fetches its own record's (a gather, once per window, outside the serial loop over the window's tiles).
    const bool packed = T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const int words = packed ? (T + 1) / 2 : T;
    uint32_t *offs = lds_offs + wave * words;
    const uint32_t *ghist = a.hist + (size_t)cp.slot * a.hist_stride + (size_t)chunk * T;
    if (packed)
        for (int t = lane; t < words; t += 64) offs[t] = 0u;
    else
        for (int t = lane; t < T; t += 64) offs[t] = ghist[t];

    const uint2 *rec = a.rec + (size_t)cp.slot * a.point_stride;
    uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;

    const int base = chunk * a.PW;
    const int end = min(base + a.PW, n);
    for (int p0 = base; p0 < end; p0 += 64) {
        const int p = p0 + lane;
        uint2 r = make_uint2(0u, KEY_OUTSIDE);
        if (p < end) r = rec[p];
        const bool inmap = r.y != KEY_OUTSIDE;
        const uint32_t t = r.y >> KEY_TILE_SHIFT;
        unsigned long long todo = __ballot(inmap);
        uint32_t dst = (packed && inmap) ? ghist[t] : 0u; // (packed: the chunk's first position in the record's tile)
        while (todo) { // one iteration per distinct tile in the window (wave-uniform loop)
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)t, leader);
            const bool mine = inmap && t == t0;
            const unsigned long long same = __ballot(mine);
            if (packed) { // (uniform)
                const uint32_t sh = (t0 & 1u) * 16u, w = offs[t0 >> 1];
                if (mine) dst += ((w >> sh) & 0xFFFFu) + (uint32_t)rank_below(same);
                if (lane == leader) offs[t0 >> 1] = w + ((uint32_t)__popcll(same) << sh);
            } else {
                const uint32_t b = offs[t0];
                if (mine) dst = b + (uint32_t)rank_below(same);
                if (lane == leader) offs[t0] = b + (uint32_t)__popcll(same);
            }
            todo &= ~same;
        }
        if (inmap) sorted[dst] = r;
    }
}

void launch_scatter(const Arena &a, const CloudParams *d_params, int n_clouds, int max_n, hipStream_t s)
{
    const int nch = (max_n + a.PW - 1) / a.PW;
    if (nch == 0 || n_clouds == 0) return;
    dim3 grid((nch + 3) / 4, n_clouds);
    const bool packed = a.g.T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const size_t lds = (size_t)4 * (packed ? (a.g.T + 1) / 2 : a.g.T) * sizeof(uint32_t);
    static std::atomic<uint64_t> big_lds_devices{0}; // (see launch_classify)
    if (lds > 64 * 1024 && first_use_on_this_device(big_lds_devices))
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(k_scatter, grid, dim3(256), lds, s, a, d_params);
}

} // namespace gg
