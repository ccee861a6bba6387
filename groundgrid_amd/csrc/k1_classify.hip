// K1 -- insert_cloud, per-point part (src/GroundSegmentation.cpp:219-279):
//   map index + inside test (:222-231), ignore test (:237-240), line-of-sight outlier test (:243-275),
//   and the Morton-tile key the ordered per-cell reduction (K2) is sorted by.
//
// One wavefront owns one contiguous chunk of PW points and walks it 64 points at a time with
// coalesced 16-B loads (packed gg_point16) or 16+4-B loads (PointXYZIR).  Per wavefront it keeps a
// tile histogram in LDS (ds_add, order-free counts) that is written out once per chunk; the stable
// scatter (k_scatter) turns the scanned histograms into destinations.  No global atomics.
//
// HBM-bound: algorithmic bytes per point = 16 (point) + 4 (old ground gather) read, 8 (z,key) written.
#include "gg_device.h"

#include <algorithm>

namespace gg {

constexpr int WALK_COOPERATIVE_MAX = 12; // candidates per 64-point window up to which their rays are walked cooperatively
// The reference walks a candidate's ray one metre per step until it reaches the point (:258) with no bound of its own: a
// corrupt z of -1e9 costs it seconds, and past 2^31 steps its int counter overflows (UB).  A library that takes raw
// sensor buffers must not hang the GPU on such a record, so steps >= WALK_MAX_STEP are not evaluated (documented
// deviation, mirrored by the oracle): it only matters for points more than 65 km away from the sensor.
constexpr int WALK_MAX_STEP = 1 << 16;

struct PointIn {
    float x, y, z;
    int ring;
};

template <int FMT>
GG_DEV PointIn load_point(const void *base, size_t idx)
{
    PointIn p;
    if (FMT == GG_POINT16) {
        const uint4 v = reinterpret_cast<const uint4 *>(base)[idx];
        p.x = __uint_as_float(v.x);
        p.y = __uint_as_float(v.y);
        p.z = __uint_as_float(v.z);
        p.ring = (int)(v.w & 0xFFFFu);
    } else {
        const uint4 *q = reinterpret_cast<const uint4 *>(base) + idx * 2;
        const uint4 v = q[0];
        const uint32_t w = reinterpret_cast<const uint32_t *>(q + 1)[1]; // byte offset 20: ring (u16) + pad
        p.x = __uint_as_float(v.x);
        p.y = __uint_as_float(v.y);
        p.z = __uint_as_float(v.z);
        p.ring = (int)(w & 0xFFFFu);
    }
    return p;
}

// :222-231 -- map index + inside test (branch-free part, so that several points per lane can be in flight)
GG_DEV bool locate_point(const Arena &a, const CloudParams &cp, const PointIn &pt, int &gi0, int &gi1)
{
    const Geometry &g = a.g;
    const double posx = (double)pt.x, posy = (double)pt.y;
    const bool inside = position_inside(g, cp.pos_x, cp.pos_y, posx, posy);
    index_from_position(g, cp.pos_x, cp.pos_y, posx, posy, gi0, gi1);
    // an index outside the grid while isInside is true is UB in the reference; treated as outside
    return inside && gi0 >= 0 && gi1 >= 0 && gi0 < g.rows && gi1 < g.cols;
}

// :237-244 -- ignore test and the entry condition of the line-of-sight test.  `oldgroundheight` = ground(gi) before this cloud.
GG_DEV int classify_point(const Arena &a, const CloudParams &cp, const PointIn &pt, float oldgroundheight, bool &walk)
{
    const float dx = pt.x - cp.ox, dy = pt.y - cp.oy;
    const float sqdist = (float)((double)dx * (double)dx + (double)dy * (double)dy); // :223
    walk = false;
    if (pt.ring > a.cfg.max_ring || sqdist < a.g.min_dist_squared) return GG_CLASS_IGNORED; // :237
    // :243-244 Outlier detection test.  (A map that holds no confidence above 0.01 anywhere -- the first cloud after
    // gg_reset_map -- cannot produce an outlier, :269: the walk is skipped instead of marching every ground return's ray
    // to its end.)
    walk = !cp.no_confidence && (double)pt.z < (double)oldgroundheight - 0.2;
    return GG_CLASS_KEPT;
}

// :246-275 -- the line-of-sight walk of ONE point, run by a whole wavefront: lane l evaluates the steps 3 + l, 67 + l, ...
// The reference walks `step` = 3, 4, ... while |step * v|^2 < len^2 and stops at the first cell whose stored ground lies
// above the ray.  The steps do not depend on each other and the continuation test is monotonic in `step` (a product with
// a fixed float factor is monotonic), so "some step before the end of the ray hits" is the same predicate -- evaluated
// 64 steps at a time instead of one lane idling its 63 neighbours through up to a hundred serial steps.
GG_DEV bool ray_walk_hits(const Arena &a, const CloudParams &cp, const float2 *__restrict__ gp2, float px, float py, float pz, int lane)
{
    const Geometry &g = a.g;
    const int rows = g.rows, cols = g.cols;
    float vx = px - cp.ox, vy = py - cp.oy, vz = pz - cp.oz;        // :248-250
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);           // :252
    vx /= len;                                                      // :253-255
    vy /= len;
    vz /= len;
    const double len2 = (double)len * (double)len;
    for (int base = 3; base < WALK_MAX_STEP; base += 64) { // :258
        const int step = base + lane;
        const float sx = (float)step * vx, sy = (float)step * vy, sz = (float)step * vz;
        const double d2 = (double)sx * (double)sx + (double)sy * (double)sy + (double)sz * (double)sz;
        const bool on_ray = d2 < len2 && vz < -0.01f && step < WALK_MAX_STEP;
        bool hit = false;
        if (on_ray) {
            const float ipx = sx + cp.ox, ipy = sy + cp.oy; // :260
            int I0, I1;
            index_from_position(g, cp.pos_x, cp.pos_y, (double)ipx, (double)ipy, I0, I1); // :261
            if (!(I0 <= 0 || I1 <= 0 || I0 >= rows - 1 || I1 >= cols - 1)) {              // :264-265
                const int r0 = max(I0 - 1, 2), c0 = max(I1 - 1, 2);                       // :268
                float e[9];
#pragma unroll
                for (int s = 0; s < 9; ++s) e[s] = gp2[gp_idx(a, r0 + s % 3, c0 + s / 3)].y;
                const float bsum = tree9(e);
                const float2 gI = gp2[gp_idx(a, I0, I1)];
                hit = (double)bsum > a.cfg.min_outlier_detection_ground_confidence && gI.y > 0.01f &&
                      (double)gI.x >= (double)(sz + cp.oz) + a.cfg.outlier_tolerance; // :269
            }
        }
        if (__ballot(hit) != 0ull) return true;
        if (__ballot(on_ray) != ~0ull) return false; // the ray ended inside this group of steps
    }
    return false;
}

// The same walk, one point per lane, serially (the reference's loop as written).  Used when most lanes of a window are
// candidates -- a freshly initialised map has ground = 0, so every return from the road surface is "below ground" -- where
// 64 lanes walking their own rays in parallel beat 64 cooperative walks one after the other.
GG_DEV bool ray_walk_hits_lane(const Arena &a, const CloudParams &cp, const float2 *__restrict__ gp2, float px, float py, float pz)
{
    const Geometry &g = a.g;
    const int rows = g.rows, cols = g.cols;
    float vx = px - cp.ox, vy = py - cp.oy, vz = pz - cp.oz;        // :248-250
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);           // :252
    vx /= len;                                                      // :253-255
    vy /= len;
    vz /= len;
    const double len2 = (double)len * (double)len;
    for (int step = 3; step < WALK_MAX_STEP; ++step) { // :258
        const float sx = (float)step * vx, sy = (float)step * vy, sz = (float)step * vz;
        const double d2 = (double)sx * (double)sx + (double)sy * (double)sy + (double)sz * (double)sz;
        if (!(d2 < len2 && vz < -0.01f)) return false;
        const float ipx = sx + cp.ox, ipy = sy + cp.oy; // :260
        int I0, I1;
        index_from_position(g, cp.pos_x, cp.pos_y, (double)ipx, (double)ipy, I0, I1); // :261
        if (I0 <= 0 || I1 <= 0 || I0 >= rows - 1 || I1 >= cols - 1) continue;         // :264-265
        const int r0 = max(I0 - 1, 2), c0 = max(I1 - 1, 2);                            // :268
        float e[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) e[s] = gp2[gp_idx(a, r0 + s % 3, c0 + s / 3)].y;
        const float bsum = tree9(e);
        const float2 gI = gp2[gp_idx(a, I0, I1)];
        if ((double)bsum > a.cfg.min_outlier_detection_ground_confidence && gI.y > 0.01f &&
            (double)gI.x >= (double)(sz + cp.oz) + a.cfg.outlier_tolerance) // :269
            return true;
    }
    return false;
}

GG_DEV uint32_t make_key(const Arena &a, const uint16_t *tile_rank, int gi0, int gi1, int cls)
{
    const Geometry &g = a.g;
    const uint32_t emit = (g.rows <= gi0 + 3 || g.cols <= gi1 + 3) ? 0u : KEY_EMIT_BIT; // :167-168 (decided by the cell only)
    const uint32_t tile = (uint32_t)tile_rank[(gi0 / TILE) + (gi1 / TILE) * g.tiles_r];
    return (tile << KEY_TILE_SHIFT) | emit | ((uint32_t)cls << KEY_CLASS_SHIFT) | (uint32_t)(gi0 % TILE) |
           ((uint32_t)(gi1 % TILE) << 4);
}

template <int FMT>
__global__ __launch_bounds__(256, 6) void k_classify(const Arena a, const CloudParams *__restrict__ params, const BatchIO io)
{
    extern __shared__ uint32_t lds_hist[]; // [4][T]
    // XCD-aware (gg_device.h): the chunks of one cloud run on one XCD, so its layers / records are cached in ONE L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x), bx = (int)(item % gridDim.x);
    const CloudParams &cp = params[cloud]; // (a reference: the 12 transform doubles stay in memory unless has_tf)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = bx * 4 + wave;
    const int n = cp.n_points;
    const int nch = (n + a.PW - 1) / a.PW;
    const int T = a.g.T;
    // tile -> Morton rank, staged once per work-group (behind the four histograms): a point's key waits for this lookup
    // (big maps: two 16-bit counters per word -- a chunk has fewer than 65536 points -- so that four work-groups fit a CU's LDS
    // where two did: this kernel lives on wavefronts in flight)
    const bool packed = T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const int words = packed ? (T + 1) / 2 : T; // per wavefront
    uint16_t *lds_tile_rank = reinterpret_cast<uint16_t *>(lds_hist + 4 * words);
    for (int t = threadIdx.x; t < T; t += 256) lds_tile_rank[t] = a.tile_rank[t];
    __syncthreads();
    if (chunk >= nch) return;

    uint32_t *hist = lds_hist + wave * words;
    for (int t = lane; t < words; t += 64) hist[t] = 0u;

    const float2 *gp2 = gp2_ptr(a, cp.slot);
    const char *pts = reinterpret_cast<const char *>(io.d_points) +
                      (size_t)cloud * io.cloud_stride * (FMT == GG_POINT16 ? 16 : 32);
    uint2 *rec = a.rec + (size_t)cp.slot * a.point_stride;

    uint32_t n_kept = 0, n_ign = 0, n_outl = 0, n_inmap = 0;
    const int base = chunk * a.PW;
    const int end = min(base + a.PW, n);
    // ITEMS independent 64-point windows per trip: all point loads are issued before the first classification, so
    // several HBM / L2 round trips (point record, then the old-ground gather) are in flight per lane.
    constexpr int ITEMS = 4;
    for (int p0 = base; p0 < end; p0 += 64 * ITEMS) {
        PointIn pt[ITEMS];
        bool valid[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int p = p0 + j * 64 + lane;
            valid[j] = p < end;
            pt[j] = load_point<FMT>(pts, (size_t)(valid[j] ? p : base));
        }
        if (cp.has_tf) { // N2: cloud still in the sensor frame (uniform branch)
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) transform_point(cp.tf, pt[j].x, pt[j].y, pt[j].z);
        }
        int gi0[ITEMS], gi1[ITEMS];
        bool inmap_[ITEMS];
        float og[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { // index math for all windows, then all old-ground gathers in flight together
            inmap_[j] = locate_point(a, cp, pt[j], gi0[j], gi1[j]) && valid[j];
            og[j] = gp2[inmap_[j] ? gp_idx(a, gi0[j], gi1[j]) : 0].x; // :243
        }
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int p = p0 + j * 64 + lane;
            int cls = GG_CLASS_KEPT;
            bool walk = false;
            if (inmap_[j]) cls = classify_point(a, cp, pt[j], og[j], walk);
            unsigned long long todo = __ballot(walk);
            if (__popcll(todo) > WALK_COOPERATIVE_MAX) { // (uniform) most lanes are candidates: everybody walks its own ray
                if (walk && ray_walk_hits_lane(a, cp, gp2, pt[j].x, pt[j].y, pt[j].z)) cls = GG_CLASS_OUTLIER;
                todo = 0ull;
            }
            for (; todo != 0ull; todo &= todo - 1ull) { // (uniform loop, rarely entered)
                const int src = __builtin_ctzll(todo);
                const bool hit = ray_walk_hits(a, cp, gp2, __shfl(pt[j].x, src, 64), __shfl(pt[j].y, src, 64), __shfl(pt[j].z, src, 64), lane);
                if (lane == src && hit) cls = GG_CLASS_OUTLIER;
            }
            uint32_t key = KEY_OUTSIDE;
            if (inmap_[j]) key = make_key(a, lds_tile_rank, gi0[j], gi1[j], cls);
            if (valid[j]) rec[p] = make_uint2(__float_as_uint(pt[j].z), key);
            const bool inmap = key != KEY_OUTSIDE;
            if (inmap) {
                const uint32_t tr = key >> KEY_TILE_SHIFT;
                if (packed) // (uniform)
                    atomicAdd(&hist[tr >> 1], 1u << ((tr & 1u) * 16u));
                else
                    atomicAdd(&hist[tr], 1u);
            }
            const bool emit = inmap && (key & KEY_EMIT_BIT);
            n_inmap += (uint32_t)__popcll(__ballot(inmap));
            n_kept += (uint32_t)__popcll(__ballot(emit && cls == GG_CLASS_KEPT));
            n_ign += (uint32_t)__popcll(__ballot(emit && cls == GG_CLASS_IGNORED));
            n_outl += (uint32_t)__popcll(__ballot(inmap && cls == GG_CLASS_OUTLIER));
        }
    }

    uint32_t *ghist = a.hist + (size_t)cp.slot * a.hist_stride + (size_t)chunk * T;
    for (int t = lane; t < T; t += 64) ghist[t] = packed ? (hist[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu : hist[t];
    if (lane == 0) {
        uint32_t *ce = a.chunk_emit + (size_t)cp.slot * a.emit_stride + (size_t)chunk * 4;
        ce[0] = n_kept;
        ce[1] = n_ign;
        ce[2] = n_outl;
        ce[3] = n_inmap;
    }
}

void launch_classify(const Arena &a, const CloudParams *d_params, const BatchIO &io, int n_clouds, int max_n, hipStream_t s)
{
    const int nch = (max_n + a.PW - 1) / a.PW;
    if (nch == 0 || n_clouds == 0) return;
    dim3 grid((nch + 3) / 4, n_clouds);
    const bool packed = a.g.T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const size_t lds = (size_t)4 * (packed ? (a.g.T + 1) / 2 : a.g.T) * sizeof(uint32_t) + (((size_t)a.g.T * 2 + 3) & ~(size_t)3);
    // per-wave tile histograms beyond the 64 KiB default (grids above ~1024 cells per side) need the explicit opt-in
    static std::atomic<uint64_t> big_lds_devices{0};
    if (lds > 64 * 1024 && first_use_on_this_device(big_lds_devices)) {
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (io.point_format == GG_POINT16)
        hipLaunchKernelGGL(k_classify<GG_POINT16>, grid, dim3(256), lds, s, a, d_params, io);
    else
        hipLaunchKernelGGL(k_classify<GG_POINT32>, grid, dim3(256), lds, s, a, d_params, io);
}

// ---- small utility kernels ----------------------------------------------------------------

__global__ void k_fill(float *dst, size_t n, float v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}

__global__ void k_fill2(float2 *dst, size_t n, float x, float y)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float2(x, y);
}
void launch_fill2(float2 *dst, size_t n, float x, float y, hipStream_t s)
{
    if (n == 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(k_fill2, dim3(blocks), dim3(256), 0, s, dst, n, x, y);
}

// one plane of the interleaved (ground, confidence) pair, device element order (gp_layout.h) <-> a plain column-major
// float layer (host boundary, K6)
__global__ void k_plane_extract(const Arena a, int slot, int comp, float *__restrict__ dst)
{
    const float2 *src = gp2_ptr(a, slot);
    const int rows = a.g.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x) {
        const float2 v = src[gp_idx(a, i % rows, i / rows)];
        dst[i] = comp ? v.y : v.x;
    }
}
__global__ void k_plane_insert(const Arena a, int slot, int comp, const float *__restrict__ src)
{
    float2 *dst = gp2_ptr(a, slot);
    const int rows = a.g.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x) {
        const int k = gp_idx(a, i % rows, i / rows);
        if (comp) dst[k].y = src[i];
        else dst[k].x = src[i];
    }
}
void launch_plane_extract(const Arena &a, int slot, int comp, float *dst, hipStream_t s)
{
    const int blocks = std::min((a.g.C + 255) / 256, 2048);
    hipLaunchKernelGGL(k_plane_extract, dim3(blocks), dim3(256), 0, s, a, slot, comp, dst);
}
void launch_plane_insert(const Arena &a, int slot, int comp, const float *src, hipStream_t s)
{
    const int blocks = std::min((a.g.C + 255) / 256, 2048);
    hipLaunchKernelGGL(k_plane_insert, dim3(blocks), dim3(256), 0, s, a, slot, comp, src);
}

__global__ void k_fill_strided(float *dst, size_t n, size_t stride, float v)
{
    float *d = dst + (size_t)blockIdx.y * stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = v;
}
void launch_fill_strided(float *dst, size_t n, size_t stride, int count, float v, hipStream_t s)
{
    if (n == 0 || count <= 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)256);
    hipLaunchKernelGGL(k_fill_strided, dim3(blocks, count), dim3(256), 0, s, dst, n, stride, v);
}
// `valid`: one bit per element of the sheared layer (gp_layout.h) -- only a third of the elements are cells, the rest is the
// padding that makes a sweep wavefront's 64 cells contiguous and is never read
__global__ __launch_bounds__(256) void k_fill2_strided(float2 *dst, size_t n, size_t stride, float x, float y, const uint32_t *__restrict__ valid)
{
    // every work-group fills one contiguous chunk of the slot; its part of the mask is read once, coalesced (per-element mask
    // loads in front of every store made the fill latency-bound: 1.1 GB in 0.45 ms)
    constexpr int MAXW = 1024; // mask words per chunk
    __shared__ uint32_t m[MAXW];
    float2 *d = dst + (size_t)blockIdx.y * stride; // (stride and the slot base are multiples of 32 elements: 16-byte aligned pairs)
    const size_t words = (n + 31) / 32, per_block = (words + gridDim.x - 1) / gridDim.x;
    const float4 both = make_float4(x, y, x, y);
    for (size_t w0 = (size_t)blockIdx.x * per_block; w0 < min(words, ((size_t)blockIdx.x + 1) * per_block); w0 += MAXW) {
        const size_t nw = min((size_t)MAXW, min(words, ((size_t)blockIdx.x + 1) * per_block) - w0);
        __syncthreads();
        for (size_t k = threadIdx.x; k < nw; k += blockDim.x) m[k] = valid[w0 + k];
        __syncthreads();
        for (size_t p = threadIdx.x; p < nw * 16; p += blockDim.x) { // pairs of elements
            const size_t i = (w0 << 5) + 2 * p;
            const uint32_t b = (m[p >> 4] >> ((2 * p) & 31)) & 3u; // (bits beyond n are 0)
            if (b == 3u)
                *reinterpret_cast<float4 *>(d + i) = both;
            else if (b == 1u)
                d[i] = make_float2(x, y);
            else if (b == 2u)
                d[i + 1] = make_float2(x, y);
        }
    }
}
void launch_fill2_strided(float2 *dst, size_t n, size_t stride, int count, float x, float y, const uint32_t *valid, hipStream_t s)
{
    if (n == 0 || count <= 0) return;
    // (many slots: few, fat work-groups per slot -- 256 x 1024 work-groups cost more to launch than their stores take)
    const int blocks = (int)std::min<size_t>((n + 255) / 256, count >= 64 ? (size_t)32 : (size_t)256);
    hipLaunchKernelGGL(k_fill2_strided, dim3(blocks, count), dim3(256), 0, s, dst, n, stride, x, y, valid);
}

__global__ void k_fill_bytes(uint8_t *dst, size_t n, uint8_t v)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
void launch_fill_bytes(uint8_t *dst, size_t n, uint8_t v, hipStream_t s)
{
    if (n == 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)256);
    hipLaunchKernelGGL(k_fill_bytes, dim3(blocks), dim3(256), 0, s, dst, n, v);
}

void launch_fill(float *dst, size_t n, float v, hipStream_t s)
{
    if (n == 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, s, dst, n, v);
}

// PointXYZIR (32 B) -> packed 16-B record, for callers that upload reference-layout clouds
__global__ void k_pack16(const gg_point32 *__restrict__ src, gg_point16 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const PointIn p = load_point<GG_POINT32>(src, i);
        uint4 o;
        o.x = __float_as_uint(p.x);
        o.y = __float_as_uint(p.y);
        o.z = __float_as_uint(p.z);
        o.w = (uint32_t)p.ring;
        reinterpret_cast<uint4 *>(dst)[i] = o;
    }
}

void launch_pack16(const gg_point32 *src, gg_point16 *dst, size_t n, hipStream_t s)
{
    if (n == 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(k_pack16, dim3(blocks), dim3(256), 0, s, src, dst, n);
}

// class + cell of every input point, decoded from the keys K1 wrote (gg_get_point_classes)
__global__ void k_decode_classes(const Arena a, int slot, size_t n, uint8_t *d_class, int32_t *d_cell)
{
    const uint2 *rec = a.rec + (size_t)slot * a.point_stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t key = rec[i].y;
        if (key == KEY_OUTSIDE) {
            d_class[i] = GG_CLASS_OUTSIDE;
            d_cell[i] = -1;
        } else {
            int row, col;
            key_to_cell(a, key, row, col);
            d_class[i] = (uint8_t)((key >> KEY_CLASS_SHIFT) & 3u);
            d_cell[i] = row + col * a.g.rows;
        }
    }
}

void launch_decode_classes(const Arena &a, int slot, size_t n, uint8_t *d_class, int32_t *d_cell, hipStream_t s)
{
    if (n == 0) return;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)2048);
    hipLaunchKernelGGL(k_decode_classes, dim3(blocks), dim3(256), 0, s, a, slot, n, d_class, d_cell);
}

} // namespace gg
