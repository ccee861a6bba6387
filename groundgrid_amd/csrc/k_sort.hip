// Tile sort, between K1 and K2, as launches of their own: a stable counting sort of the in-map points by Morton tile rank.
// (The front end can also run the scan, or the scan and the scatter, inside k_classify: k1_classify.hip, gg_internal.h FRONT_*.
// The work itself is sort_core.h in every shape.)
//
//   k_scan    : one work-group per cloud.  hist[chunk][tile] (written by K1, one row per wave-chunk)
//               -> exclusive offsets in (tile-major, chunk-minor) order, tile_start[], and the
//               exclusive prefixes of the per-chunk emission counters (kept / ignored / outliers)
//               that give every point its position in the returned cloud (K5); and K2's two work lists (light / dense tiles).
//   k_scatter : same wave <-> chunk mapping as K1.  Each wave re-walks its chunk in cloud order and
//               places record p at offset[tile] + (number of earlier points of the chunk in that tile); STABLE
//               (sort_core.h scatter_chunk), which is what makes the float32 Welford recurrence of K2 bit-reproducible
//               (src/GroundSegmentation.cpp:296-305 is order dependent).
#include "gg_device.h"
#include "sort_core.h"

#include <algorithm>

namespace gg {

__global__ __launch_bounds__(1024) void k_scan(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ uint32_t lds[32];
    __shared__ u32x4 part[1024];
    const CloudParams cp = params[blockIdx.x];
    scan_cloud<16, false>(a, cp, (cp.n_points + a.PW - 1) / a.PW, lds, part);
}

// ... of one cloud by gridDim.x work-groups (sort_core.h scan_cloud "PARTS"): few clouds on a map with thousands of tiles
// (configs[3]: 3969 tiles x 257 chunk rows = 4 MB of counters per cloud, and one work-group per cloud would leave half the CUs idle).
__global__ __launch_bounds__(1024) void k_scan_parts(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ uint32_t lds[32];
    __shared__ u32x4 part[1024];
    const CloudParams cp = params[blockIdx.y];
    unsigned long long *sync = a.scan_sync + (size_t)blockIdx.y * SCAN_SYNC_WORDS;
    if (threadIdx.x == 0) lds[0] = __hip_atomic_fetch_add(reinterpret_cast<uint32_t *>(sync), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int my_part = (int)lds[0]; // (a ticket, not blockIdx: the parts before this one are running or done)
    __syncthreads();
    scan_cloud<16, false>(a, cp, (cp.n_points + a.PW - 1) / a.PW, lds, part, my_part, (int)gridDim.x, sync + 1);
}

// parts per cloud: only where a part still has >= 64 tile groups and the launch would not fill the chip by itself; a part's
// range must fit one round of the work-group (1024 tile groups)
int scan_parts(const Arena &a, int n_clouds)
{
    const int G = a.hist_pitch / 4;
    if (a.tune_scan_parts > 0) return std::max((G + SCAN_PART_MAX_GROUPS - 1) / SCAN_PART_MAX_GROUPS, std::min(a.tune_scan_parts, SCAN_MAX_PARTS));
    if (G < 512 || n_clouds >= 512) return 1;
    return std::max((G + SCAN_PART_MAX_GROUPS - 1) / SCAN_PART_MAX_GROUPS, std::min(std::min(SCAN_MAX_PARTS, G / 64), 512 / n_clouds));
}

void launch_scan(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    const int parts = scan_parts(a, n_clouds);
    if (parts <= 1) { // (one work-group walks any number of rounds)
        hipLaunchKernelGGL(k_scan, dim3(n_clouds), dim3(1024), 0, s, a, d_params);
        return;
    }
    hipMemsetAsync(a.scan_sync, 0, (size_t)n_clouds * SCAN_SYNC_WORDS * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(k_scan_parts, dim3(parts, n_clouds), dim3(1024), 0, s, a, d_params);
}

__global__ __launch_bounds__(256) void k_scatter(const Arena a, const CloudParams *__restrict__ params)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_offs[]; // [4][words]
    // XCD-aware (gg_device.h): the chunks of one cloud run on one XCD, so its layers / records are cached in ONE L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x), bx = (int)(item % gridDim.x);
    const CloudParams cp = params[cloud];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = bx * 4 + wave;
    const int n = cp.n_points;
    const int nch = (n + a.PW - 1) / a.PW;
    if (chunk >= nch) return;

    const int T = a.g.T, TP = a.hist_pitch;
    // Small maps: the wavefront's running offset per tile, absolute, in LDS.  Big maps (PACKED_TILE_COUNTERS_MIN_T): only the
    // number of records already placed per tile, two 16-bit counters per word; the chunk's start per tile stays in its row of
    // `hist`, and every lane fetches its own record's (a gather, once per window).
    const bool packed = T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const int words = packed ? TP / 2 : TP;
    uint32_t *offs = lds_offs + wave * words;
    const __amdgpu_buffer_rsrc_t ghist = words_rsrc(a.hist + (size_t)cp.slot * a.hist_stride, a.hist_stride);
    const uint32_t row = (uint32_t)chunk * (uint32_t)TP;
    const uint2 *rec = a.rec + (size_t)cp.slot * a.point_stride;
    uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    const int base = chunk * a.PW;
    const int end = min(base + a.PW, n);
    if (packed) {
        for (int t = lane; t < words; t += 64) offs[t] = 0u;
        scatter_chunk<true, false>(offs, ghist, row, rec, sorted, base, end, lane);
    } else {
        for (int g = lane; g < TP / 4; g += 64) *reinterpret_cast<u32x4 *>(offs + 4 * g) = load16_row<false>(ghist, row + 4u * (uint32_t)g);
        scatter_chunk<false, false>(offs, ghist, row, rec, sorted, base, end, lane);
    }
}

void launch_scatter(const Arena &a, const CloudParams *d_params, int n_clouds, int max_n, hipStream_t s)
{
    const int nch = (max_n + a.PW - 1) / a.PW;
    if (nch == 0 || n_clouds == 0) return;
    dim3 grid((nch + 3) / 4, n_clouds);
    const bool packed = a.g.T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const size_t lds = (size_t)4 * (packed ? a.hist_pitch / 2 : a.hist_pitch) * sizeof(uint32_t);
    static PerDeviceOnce big_lds; // (see launch_classify)
    if (lds > 64 * 1024)
        big_lds.run([] { hipFuncSetAttribute(reinterpret_cast<const void *>(k_scatter), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL(k_scatter, grid, dim3(256), lds, s, a, d_params);
}

} // namespace gg
