// K5 -- filter_cloud's label loop and epilogue (src/GroundSegmentation.cpp:147-189).
//
// The reference returns a NEW cloud: kept points (cloud order), then ignored points (cloud order), then
// outliers (cloud order), intensity overwritten with 99 (non-ground) / 49 (ground); points outside the
// map or in the last 3 rows / cols are dropped.  This kernel produces, per INPUT point, the label and the
// position in that returned cloud (and optionally the returned cloud itself): same wave <-> chunk
// mapping as K1, per-chunk exclusive emission prefixes from k_scan, ranks inside a 64-point window from
// ballots -> deterministic, identical order.  Non-ground points are counted back into `points` (:176)
// with float atomics (exact: integer counts < 2^24, order-free).
//
// Algorithmic bytes per point: 8 (z,key) + 8 (ground, variance gathers) read (+ 16 for the ~2 % of the points whose
// tolerance needs the exact distance); 1 + 4 written.
#include "gg_device.h"

namespace gg {

// x, y of input point p in the map frame (N2: same arithmetic as K1 -> the same values; z travels in rec)
template <int FMT>
GG_DEV void load_xy(const char *pts, int p, const CloudParams &cp, float &x, float &y)
{
    const uint4 v = *reinterpret_cast<const uint4 *>(pts + (size_t)p * (FMT == GG_POINT16 ? 16 : 32));
    x = __uint_as_float(v.x);
    y = __uint_as_float(v.y);
    if (cp.has_tf) {
        const double dx = (double)x, dy = (double)y, dz = (double)__uint_as_float(v.z);
        x = (float)(((cp.tf[0] * dx + cp.tf[1] * dy) + cp.tf[2] * dz) + cp.tf[3]);
        y = (float)(((cp.tf[4] * dx + cp.tf[5] * dy) + cp.tf[6] * dz) + cp.tf[7]);
    }
}

// One record of the returned cloud in the 18-byte sensor_msgs/PointCloud2 layout of scripts/kitti_data_publisher.py:139-150 -- x@0 y@4
// z@8 intensity@12 (float32) ring@16 (uint16), point_step 18 -- at position idx: 18 idx is a multiple of 4 for even idx and 2 more
// for odd ones, so a record is four aligned words and one half word, in that order or the other.
GG_DEV void store_pc2_record(uint8_t *out, int32_t idx, uint32_t x, uint32_t y, uint32_t z, uint32_t intensity, uint32_t ring)
{
    uint8_t *p = out + (size_t)idx * GG_PC2_POINT_STEP;
    if (idx & 1) {
        *reinterpret_cast<uint16_t *>(p) = (uint16_t)x;
        uint32_t *w = reinterpret_cast<uint32_t *>(p + 2);
        w[0] = (x >> 16) | (y << 16);
        w[1] = (y >> 16) | (z << 16);
        w[2] = (z >> 16) | (intensity << 16);
        w[3] = (intensity >> 16) | (ring << 16);
    } else {
        uint32_t *w = reinterpret_cast<uint32_t *>(p);
        w[0] = x;
        w[1] = y;
        w[2] = z;
        w[3] = intensity;
        *reinterpret_cast<uint16_t *>(p + 16) = (uint16_t)ring;
    }
}

// PC2: the instantiation that also emits the returned cloud as 18-byte PointCloud2 records (gg_batch.d_out_pc2) -- a twin, so that the
// throughput kernel keeps its registers
template <int FMT, bool PC2>
__global__ __launch_bounds__(256) void k_label(const Arena a, const CloudParams *__restrict__ params, const BatchIO io)
{
    // XCD-aware (gg_device.h): the chunks of one cloud run on one XCD, so its layers / records are cached in ONE L2
    const uint32_t item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x), bx = (int)(item % gridDim.x);
    const CloudParams cp = params[cloud];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunk = bx * 4 + wave;
    const int n = cp.n_points;
    const int nch = (n + a.PW - 1) / a.PW;
    // tile rank -> first cell of the tile, staged once per work-group: the lookup sits between a point's record and its two
    // gathers, and from LDS it is not a third round trip through the caches
    extern __shared__ uint32_t lds_cell0[]; // [T]
    for (int t = threadIdx.x; t < a.g.T; t += 256) lds_cell0[t] = a.rank_cell0[t];
    __syncthreads();

    const uint32_t *totals = a.totals + (size_t)cp.slot * 4;
    if (io.d_out_counts && bx == 0 && threadIdx.x == 0) {
        int32_t *oc = io.d_out_counts + (size_t)cp.io_index * 4;
        oc[0] = (int32_t)(totals[0] + totals[1] + totals[2]);
        oc[1] = (int32_t)totals[0];
        oc[2] = (int32_t)totals[1];
        oc[3] = (int32_t)totals[2];
    }
    if (chunk >= nch) return;

    const float *L = percall_ptr(a, cp.slot);
    const float2 *gp2 = gp2_ptr(a, cp.slot);
    // (a record's key names its tile and its cell in the tile: the per-call layers are addressed without the tile's origin)
    const float *variance = L + percall_index(0, PL_VARIANCE, 0);
    float *points = const_cast<float *>(L) + percall_index(0, PL_POINTS, 0);
    const uint2 *rec = a.rec + (size_t)cp.slot * a.point_stride;
    const char *pts = reinterpret_cast<const char *>(io.d_points) + (size_t)cp.io_index * io.cloud_stride * (FMT == GG_POINT16 ? 16 : 32);
    uint8_t *labels = io.d_labels ? io.d_labels + (size_t)cp.io_index * io.cloud_stride + cp.label_shift : nullptr;
    uint8_t *masks = io.d_label_masks ? io.d_label_masks + (size_t)cp.io_index * ((io.cloud_stride + 3) / 4) : nullptr;
    int32_t *out_index = io.d_out_index ? io.d_out_index + (size_t)cp.io_index * io.cloud_stride : nullptr;
    gg_point32 *out_cloud = (FMT == GG_POINT32 && io.d_out_clouds) ? io.d_out_clouds + (size_t)cp.io_index * io.cloud_stride : nullptr;
    uint8_t *out_pc2 = PC2 ? io.d_out_pc2 + (size_t)cp.io_index * io.cloud_stride * GG_PC2_POINT_STEP : nullptr;

    const uint32_t *ce = a.chunk_emit + (size_t)cp.slot * a.emit_stride + (size_t)chunk * 4;
    uint32_t kept_base = ce[0];
    uint32_t ign_base = totals[0] + ce[1];
    uint32_t outl_base = totals[0] + totals[1] + ce[2];

    const DevConfig &cfg = a.cfg;
    const int base = chunk * a.PW;
    const int end = min(base + a.PW, n);
    // :171 tolerance = max(min(t, thres), obs), t = (5 mdf * dist) / variance * thres.  On real data t is far above
    // thres (variance ~1e-4 m^2), so the clamp decides; a float estimate of t with a 1e-3 safety band selects the clamp
    // without the f64 sqrt and divide, and only estimates inside the band (or NaN / odd configs) take the exact path.
    const double thres = cfg.min_point_height_thres, obs = cfg.min_point_height_obs_thres;
    const bool normal_cfg = obs <= thres && thres > 0.0 && obs > 0.0 && cfg.min_dist_fac > 0.0;
    const double tol_hi = std_max(thres, obs); // value of the expression when t > thres
    const float thres_f = (float)thres, obs_f = (float)obs, fac_f = (float)cfg.min_dist_fac;
    // ... and the estimate does not need the point either: its cell (from the key) bounds its distance from the origin to
    // dc -+ 0.7072 res (dc = the cell centre's distance; 0.75 res covers the float rounding of all of this), t is monotonic
    // in the distance, and the clamp is decided when both ends of that interval say the same.  ~2 % of the points of a
    // street scene (cells with variance > 0.0025 dist, i.e. obstacles) are in between and read their x, y for the exact
    // expression; the other 98 % never touch the input cloud here.
    const float res_f = (float)a.g.resolution, cell_reach = 0.75f * res_f;
    const float cell0_x = (float)((cp.pos_x + a.g.half0) - (double)cp.ox), cell0_y = (float)((cp.pos_y + a.g.half1) - (double)cp.oy);

    constexpr int ITEMS = 4;
    // The records of the NEXT 256 points are requested while this iteration's gathers are in flight (two buffers that take turns:
    // a copy at the loop's back edge would wait for the load): an iteration is two dependent round trips -- records, then the
    // cell's ground and variance -- and the first of them now overlaps the previous iteration's second.
    auto request_records = [&](uint2 (&r)[ITEMS], int p0) { // (unconditional loads at clamped indices)
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) r[j] = rec[min(p0 + j * 64 + lane, end - 1)];
    };
    auto iteration = [&](uint2 (&r)[ITEMS], uint2 (&r_next)[ITEMS], int p0) {
        bool valid[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            valid[j] = p0 + j * 64 + lane < end;
            if (!valid[j]) r[j].y = KEY_OUTSIDE;
        }
        uint32_t cidx[ITEMS]; // (C < 2^32)
        float gh[ITEMS], var[ITEMS], dc[ITEMS];
        bool lab[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) { // all gathers of all windows in flight together
            const uint32_t key = r[j].y;
            const bool inmap = key != KEY_OUTSIDE;
            const int cls = (int)((key >> KEY_CLASS_SHIFT) & 3u);
            lab[j] = inmap && (key & KEY_EMIT_BIT) && cls != GG_CLASS_OUTLIER; // kept or ignored, not on the border (:167)
            int row = 0, col = 0;
            if (lab[j]) {
                const uint32_t c0 = lds_cell0[key >> KEY_TILE_SHIFT];
                row = (int)(c0 & 0xFFFFu) + (int)(key & 15u);
                col = (int)(c0 >> 16) + (int)((key >> 4) & 15u);
            }
            cidx[j] = lab[j] ? (key >> KEY_TILE_SHIFT) * (uint32_t)PERCALL_BLOCK + (key & 255u) : 0u;
            if (GG_DEBUG_SWITCH(a, k5_debug) & 1) cidx[j] = 0u, row = col = 0; // (measurement)
            gh[j] = gp2[gp_idx(a, row, col)].x; // :162
            var[j] = variance[cidx[j]]; // :165
            const float cx = cell0_x - ((float)row + 0.5f) * res_f, cy = cell0_y - ((float)col + 0.5f) * res_f;
            dc[j] = __builtin_amdgcn_sqrtf(cx * cx + cy * cy); // distance of the cell's centre from the origin (1 ulp: it is a bound)
        }
        request_records(r_next, min(p0 + 64 * ITEMS, end - 1));
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) {
            const int p = p0 + j * 64 + lane;
            const uint32_t key = r[j].y;
            const bool inmap = key != KEY_OUTSIDE;
            const int cls = (int)((key >> KEY_CLASS_SHIFT) & 3u);
            const bool emit = inmap && (key & KEY_EMIT_BIT);
            const bool is_kept = emit && cls == GG_CLASS_KEPT;
            const bool is_ign = emit && cls == GG_CLASS_IGNORED;
            const bool is_outl = inmap && cls == GG_CLASS_OUTLIER;

            const unsigned long long mk = __ballot(is_kept), mi = __ballot(is_ign), mo = __ballot(is_outl);
            int32_t idx = -1;
            uint8_t label = GG_LABEL_DROPPED;
            if (is_outl) { // :185-189
                idx = (int32_t)(outl_base + (uint32_t)rank_below(mo));
                label = GG_LABEL_GROUND;
            } else if (lab[j]) { // :158-182
                const float z = __uint_as_float(r[j].x);
                double tolerance;
                const float s_v = fac_f * __builtin_amdgcn_rcpf(var[j]) * thres_f; // (v_rcp_f32, 1 ulp; inf / NaN / negative fall through below)
                const float t_lo = fmaxf(dc[j] - cell_reach, 0.0f) * s_v, t_hi = (dc[j] + cell_reach) * s_v;
                if (normal_cfg && t_lo > thres_f * 1.001f) {
                    tolerance = tol_hi;
                } else if (normal_cfg && t_hi >= 0.0f && t_hi < obs_f * 0.999f) {
                    tolerance = obs;
                } else {
                    asm volatile("; exact tolerance" ::: "memory"); // (keeps the point's load inside the branch)
                    float x, y;
                    load_xy<FMT>(pts, p, cp, x, y);
                    const float dist = ref_hypotf(x - cp.ox, y - cp.oy); // :170
                    tolerance = std_max(std_min((cfg.min_dist_fac * (double)dist) / (double)var[j] * thres, thres), obs); // :171
                }
                label = (tolerance + (double)gh[j] < (double)z) ? GG_LABEL_NONGROUND : GG_LABEL_GROUND; // :173
                idx = is_kept ? (int32_t)(kept_base + (uint32_t)rank_below(mk)) : (int32_t)(ign_base + (uint32_t)rank_below(mi));
            }
            // :176 points(gi) += 1 per non-ground point.  Consecutive points of a scan line mostly fall into the same cell,
            // and same-address atomics serialise in L2: every RUN of consecutive lanes with the same cell issues one
            // hardware float atomic carrying the run length (exact: integer-valued floats < 2^24, order-free) -- one
            // atomic instruction per window, no loop over cells.
            {
                const bool ng = label == GG_LABEL_NONGROUND;
                const uint32_t c = cidx[j];
                const uint32_t c_prev = (uint32_t)__shfl_up((int)c, 1, 64);
                const unsigned long long ngm = __ballot(ng);
                const bool joins = ng && lane > 0 && ((ngm >> (lane - 1)) & 1ull) && c == c_prev; // continues the previous lane's run
                const unsigned long long jm = __ballot(joins);
                if (ng && !joins && !(GG_DEBUG_SWITCH(a, k5_debug) & 4)) {
                    const unsigned long long after = lane == 63 ? 0ull : ~(jm >> (lane + 1)); // first 1 bit = first lane that does not join
                    const int run = 1 + (after ? __builtin_ctzll(after) : 63 - lane);
                    unsafeAtomicAdd(&points[c], (float)run);
                }
            }
            if (masks) { // 2-bit label mask, four points per byte: lanes 4k .. 4k+3 -> lane 4k (windows start at multiples of 64)
                const uint32_t code = !valid[j] ? 0u : label == GG_LABEL_GROUND ? 1u : label == GG_LABEL_NONGROUND ? 2u : 0u;
                const uint32_t b4 = code | ((uint32_t)__shfl_down((int)code, 1, 64) << 2) | ((uint32_t)__shfl_down((int)code, 2, 64) << 4) |
                                    ((uint32_t)__shfl_down((int)code, 3, 64) << 6);
                if ((lane & 3) == 0 && p < (int)io.cloud_stride) masks[p >> 2] = (uint8_t)b4;
            }
            if (valid[j] && !(GG_DEBUG_SWITCH(a, k5_debug) & 2)) {
                if (labels) labels[p] = label;
                if (out_index) out_index[p] = idx;
                if (FMT == GG_POINT32 && out_cloud && idx >= 0) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(pts) + (size_t)p * 2;
                    uint4 lo = src[0], hi = src[1];
                    if (cp.has_tf) { // the returned cloud is in the map frame
                        float x, y;
                        load_xy<FMT>(pts, p, cp, x, y);
                        lo.x = __float_as_uint(x);
                        lo.y = __float_as_uint(y);
                        lo.z = r[j].x;
                    }
                    hi.x = __float_as_uint((float)label); // intensity := 49 / 99
                    uint4 *dst = reinterpret_cast<uint4 *>(out_cloud + idx);
                    dst[0] = lo;
                    dst[1] = hi;
                }
                if (PC2 && idx >= 0) { // (src/GroundGridNodelet.cpp:196-200: the returned cloud goes out as a PointCloud2)
                    const uint4 v = *reinterpret_cast<const uint4 *>(pts + (size_t)p * (FMT == GG_POINT16 ? 16 : 32));
                    uint32_t x = v.x, y = v.y, ring;
                    if (FMT == GG_POINT16)
                        ring = v.w & 0xFFFFu;
                    else
                        ring = *reinterpret_cast<const uint16_t *>(pts + (size_t)p * 32 + 20);
                    if (cp.has_tf) { // the returned cloud is in the map frame
                        float fx, fy;
                        load_xy<FMT>(pts, p, cp, fx, fy);
                        x = __float_as_uint(fx);
                        y = __float_as_uint(fy);
                    }
                    store_pc2_record(out_pc2, idx, x, y, r[j].x, __float_as_uint((float)label), ring);
                }
            }
            kept_base += (uint32_t)__popcll(mk);
            ign_base += (uint32_t)__popcll(mi);
            outl_base += (uint32_t)__popcll(mo);
        }
    };
    if (base >= end) return;
    uint2 ra[ITEMS], rb[ITEMS];
    request_records(ra, base);
    for (int p0 = base; p0 < end; p0 += 2 * 64 * ITEMS) {
        iteration(ra, rb, p0);
        if (p0 + 64 * ITEMS >= end) break;
        iteration(rb, ra, p0 + 64 * ITEMS);
    }
}

void launch_label(const Arena &a, const CloudParams *d_params, const BatchIO &io, int n_clouds, int max_n, hipStream_t s)
{
    if (n_clouds == 0) return;
    int nch = (max_n + a.PW - 1) / a.PW;
    if (nch == 0) nch = 1; // still publish the (all-zero) counts
    dim3 grid((nch + 3) / 4, n_clouds);
    const size_t lds = (size_t)a.g.T * sizeof(uint32_t);
    if (io.point_format == GG_POINT16) {
        if (io.d_out_pc2)
            hipLaunchKernelGGL((k_label<GG_POINT16, true>), grid, dim3(256), lds, s, a, d_params, io);
        else
            hipLaunchKernelGGL((k_label<GG_POINT16, false>), grid, dim3(256), lds, s, a, d_params, io);
    } else {
        if (io.d_out_pc2)
            hipLaunchKernelGGL((k_label<GG_POINT32, true>), grid, dim3(256), lds, s, a, d_params, io);
        else
            hipLaunchKernelGGL((k_label<GG_POINT32, false>), grid, dim3(256), lds, s, a, d_params, io);
    }
}

} // namespace gg
