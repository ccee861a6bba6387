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
#include "sort_core.h"

#include <algorithm>

namespace gg {

// The reference walks a candidate's ray one metre per step until it reaches the point (:258) with no bound of its own: a
// corrupt z of -1e9 costs it seconds, and past 2^31 steps its int counter overflows (UB).  A library that takes raw
// sensor buffers must not hang the GPU on such a record, so steps >= WALK_MAX_STEP are not evaluated (documented
// deviation, mirrored by the oracle): it only matters for points more than 65 km away from the sensor.
constexpr int WALK_MAX_STEP = 1 << 16;
constexpr int WALK_PACKED_MAX = 12; // candidates per 64-point window up to which their steps are dealt to the lanes (walk_packed); above, a ray per lane

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

// :246-275 -- the line-of-sight walks of a 64-point window, PACKED: the reference walks `step` = 3, 4, ... along a candidate's ray while
// |step * v|^2 < len^2 and stops at the first cell whose stored ground lies above the ray.  The steps do not depend on each other and
// the continuation test is monotonic in `step` (a product with a fixed float factor is monotonic), so "some step before the end of
// the ray hits" is the same predicate, and the (candidate, step) pairs of a window are independent work items.  A ray of length len
// has at most floor(len) + 2 - 3 steps (|v| is 1 within a few 2^-24, so |step v|^2 < len^2 implies step <= floor(len) + 1).  The
// wavefront takes the items 64 at a time: the candidates are consumed in lane order by a SCALAR cursor (candidate, next step, steps
// left) that deals consecutive lanes to consecutive steps and moves on to the next candidate when a ray is used up -- a pass holds the
// end of one ray and the beginning of the next (or a dozen short rays) -- with the ray's parameters broadcast from its lane
// (v_readlane); a candidate that has hit gives up the rest of its steps.  Round 4 walked one candidate per pass (64 steps, mostly
// beyond the end of a 20-step ray) or, above a dozen candidates, every lane its own ray serially.  Returns the lanes whose ray hits.
GG_DEV float lane_value(float v, int src) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); }
GG_DEV unsigned long long walk_packed(const Arena &a, const CloudParams &cp, const float2 *__restrict__ gp2, float px, float py, float pz, bool walk, int lane)
{
    const Geometry &g = a.g;
    const int rows = g.rows, cols = g.cols;
    float vx = px - cp.ox, vy = py - cp.oy, vz = pz - cp.oz;        // :248-250
    const float len = sqrtf(vx * vx + vy * vy + vz * vz);           // :252
    vx /= len;                                                      // :253-255
    vy /= len;
    vz /= len;
    const float len2_f = len; // (the ray's owner squares it in double: one float travels instead of a double)
    // steps 3 .. smax can be on the ray (:258: needs vec.z < -0.01f; a NaN anywhere compares false: no step)
    const int smax = min((int)fminf(len, 70000.0f) + 1, WALK_MAX_STEP - 1);
    const int count = (walk && vz < -0.01f && smax >= 3) ? smax - 2 : 0;
    unsigned long long todo = __ballot(count > 0), hits = 0ull;
    int cur = 0, cur_left = 0, cur_step = 3; // (scalar) the candidate being consumed
    float cvx = 0.f, cvy = 0.f, cvz = 0.f, clen = 0.f;
    while (todo != 0ull || cur_left > 0) { // one pass of up to 64 items (uniform)
        float ovx = 0.f, ovy = 0.f, ovz = 0.f, olen = 0.f;
        int step = 0, owner = 0;
        bool item = false;
        int fill = 0;
        while (fill < 64 && (cur_left > 0 || todo != 0ull)) { // (uniform) deal lanes [fill, fill + take) to the cursor's candidate
            if (cur_left == 0) {
                cur = __builtin_ctzll(todo);
                todo &= todo - 1ull;
                cvx = lane_value(vx, cur), cvy = lane_value(vy, cur), cvz = lane_value(vz, cur), clen = lane_value(len2_f, cur);
                cur_left = __builtin_amdgcn_readlane(count, cur);
                cur_step = 3;
            }
            const int take = min(cur_left, 64 - fill);
            const bool mine = (unsigned)(lane - fill) < (unsigned)take;
            ovx = mine ? cvx : ovx, ovy = mine ? cvy : ovy, ovz = mine ? cvz : ovz, olen = mine ? clen : olen;
            step = mine ? cur_step + (lane - fill) : step;
            owner = mine ? cur : owner;
            item = item || mine;
            fill += take;
            cur_left -= take;
            cur_step += take;
        }
        const float sx = (float)step * ovx, sy = (float)step * ovy, sz = (float)step * ovz;
        const double d2 = (double)sx * (double)sx + (double)sy * (double)sy + (double)sz * (double)sz;
        const bool on_ray = item && d2 < (double)olen * (double)olen; // (:258; vec.z < -0.01f is in `count`)
        bool hit = false;
        int r0 = 2, c0 = 2;
        if (on_ray) {
            const float ipx = sx + cp.ox, ipy = sy + cp.oy; // :260
            int I0, I1;
            index_from_position(g, cp.pos_x, cp.pos_y, (double)ipx, (double)ipy, I0, I1); // :261
            if (!(I0 <= 0 || I1 <= 0 || I0 >= rows - 1 || I1 >= cols - 1)) {              // :264-265
                // :269 is a conjunction of three pure reads: the two conditions on the cell itself first (one gather) -- along most of a
                // ray the stored ground lies BELOW the ray, so they rule the step out -- and the 3 x 3 confidence sum (nine gathers in the
                // sheared layer, the walk's traffic) only for the steps they let through
                const float2 gI = gp2[gp_idx(a, I0, I1)];
                hit = gI.y > 0.01f && (double)gI.x >= (double)(sz + cp.oz) + a.cfg.outlier_tolerance;
                r0 = max(I0 - 1, 2), c0 = max(I1 - 1, 2); // :268
            }
        }
        unsigned long long hm = __ballot(hit);
        if (hm != 0ull) { // (uniform; rare)
            if (hit) {
                float e[9];
#pragma unroll
                for (int s = 0; s < 9; ++s) e[s] = gp2[gp_idx(a, r0 + s % 3, c0 + s / 3)].y;
                hit = (double)tree9(e) > a.cfg.min_outlier_detection_ground_confidence;
            }
            for (hm = __ballot(hit); hm != 0ull; hm &= hm - 1ull) hits |= 1ull << __builtin_amdgcn_readlane(owner, __builtin_ctzll(hm));
            if ((hits >> cur) & 1ull) cur_left = 0; // the ray in hand has its answer: its remaining steps are not needed
        }
    }
    return hits;
}

// The walk of one point per lane, serially (the reference's loop as written).  Used when most lanes of a window are candidates --
// a map that meets an unrelated scene: the near rings' returns all lie under the stored terrain, 64 candidates with rays of five to
// thirty steps -- where 64 lanes walking their own rays side by side, each leaving at its first hit, beat any dealing of items
// (measured on that stress case: 3.2 ms per 1024 clouds against 6.4 for walk_packed, profiles/r05a/walk_ab.log).
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
    if (!(vz < -0.01f)) return false; // :258 (no step at all)
    // WALK_UNROLL steps per trip, their cells' gathers in flight together: the steps are independent, and a trip of the serial loop
    // is one dependent round trip to the layer per step
    constexpr int WALK_UNROLL = 2;
    for (int base = 3; base < WALK_MAX_STEP; base += WALK_UNROLL) { // :258
        bool on[WALK_UNROLL], inside[WALK_UNROLL];
        int I0[WALK_UNROLL], I1[WALK_UNROLL];
        float rz[WALK_UNROLL];
        float2 gI[WALK_UNROLL];
#pragma unroll
        for (int k = 0; k < WALK_UNROLL; ++k) {
            const int step = base + k;
            const float sx = (float)step * vx, sy = (float)step * vy, sz = (float)step * vz;
            const double d2 = (double)sx * (double)sx + (double)sy * (double)sy + (double)sz * (double)sz;
            on[k] = d2 < len2 && step < WALK_MAX_STEP; // (monotonic in step: once false, false for every later step; the bound as in walk_packed and the oracle)
            const float ipx = sx + cp.ox, ipy = sy + cp.oy; // :260
            index_from_position(g, cp.pos_x, cp.pos_y, (double)ipx, (double)ipy, I0[k], I1[k]); // :261
            inside[k] = on[k] && !(I0[k] <= 0 || I1[k] <= 0 || I0[k] >= rows - 1 || I1[k] >= cols - 1); // :264-265
            rz[k] = sz + cp.oz;
            gI[k] = gp2[inside[k] ? gp_idx(a, I0[k], I1[k]) : 0]; // (unconditional load at a clamped index)
        }
        if (!on[0]) return false;
#pragma unroll
        for (int k = 0; k < WALK_UNROLL; ++k) {
            // :269, the cell's own two conditions first (most steps of a ray run above the stored ground: they end here)
            if (!(inside[k] && gI[k].y > 0.01f && (double)gI[k].x >= (double)rz[k] + a.cfg.outlier_tolerance)) continue;
            const int r0 = max(I0[k] - 1, 2), c0 = max(I1[k] - 1, 2); // :268
            float e[9];
#pragma unroll
            for (int s = 0; s < 9; ++s) e[s] = gp2[gp_idx(a, r0 + s % 3, c0 + s / 3)].y;
            if ((double)tree9(e) > a.cfg.min_outlier_detection_ground_confidence) return true;
        }
        if (!on[WALK_UNROLL - 1]) return false;
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

// The front end in one, two or three launches (gg_internal.h FRONT_*):
//   FRONT_THREE_LAUNCHES    this kernel classifies and writes the chunk histograms; k_scan and k_scatter (k_sort.hip) follow.
//   FRONT_SCAN_IN_CLASSIFY  the work-groups of a cloud count their arrivals; the LAST one to finish scans the cloud's histograms
//                           (sort_core.h scan_cloud) while the chip classifies other clouds.  Nobody waits for anybody: no
//                           assumption about dispatch order or residency.  k_scatter follows.
//   FRONT_ONE_LAUNCH        ... and the others wait for that scan (one lane polls one word, bounded) and scatter their own
//                           chunks, whose records they wrote a moment ago.  A waiting work-group needs the other work-groups of
//                           its cloud to START: work is handed out by TICKET (an atomic counter per XCD, taken when a
//                           work-group starts running, cloud-major), so at most one cloud per counter is ever partly started,
//                           and the launcher only picks this shape when the device holds more work-groups than that
//                           (launch_classify) -- whatever order the dispatcher starts work-groups in (MI355X_MICROARCH.md
//                           Contract [G]).  A wait that still runs out leaves GG_DEVERR_FRONT_WAIT instead of hanging.
// What crosses work-groups inside the launch (histogram rows, emission counters, the scanned offsets) is written and read with
// 16-byte agent-scope accesses (sort_core.h), the arrival counter and the flag with agent-scope atomics; the words are zeroed by
// a memset node in front of every launch.
constexpr uint32_t FRONT_WAIT_POLLS = 1u << 22; // x ~0.3 us: gives up after about a second

template <int FMT, int SHAPE>
__global__ __launch_bounds__(256, 5) void k_classify(const Arena a, const CloudParams *__restrict__ params, const BatchIO io, int n_counters)
{
    // dynamic LDS only (a static variable would shift its base off 16 bytes): [4][words] histograms, the tile ranks, 16 scratch words
    extern __shared__ __attribute__((aligned(16))) uint32_t lds_hist[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int T = a.g.T, TP = a.hist_pitch;
    // tile -> Morton rank, staged once per work-group (behind the four histograms): a point's key waits for this lookup
    // (big maps: two 16-bit counters per word -- a chunk has fewer than 65536 points -- so that four work-groups fit a CU's LDS
    // where two did: this kernel lives on wavefronts in flight)
    const bool packed = T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    const int words = packed ? TP / 2 : TP; // per wavefront
    uint16_t *lds_tile_rank = reinterpret_cast<uint16_t *>(lds_hist + 4 * words);
    uint32_t *s_scan = lds_hist + 4 * words + (((T * 2 + 15) & ~15) >> 2);
    uint32_t &s_word = s_scan[8];
    const uint32_t n_items = gridDim.x * gridDim.y;
    uint32_t item;
    if (SHAPE == FRONT_ONE_LAUNCH) {
        // a ticket from this XCD's counter (its share of the cloud-major work list, as xcd_contiguous_item cuts it), from the next
        // XCD's when that one is used up: there are as many tickets as work-groups
        if (threadIdx.x == 0) {
            uint32_t *tickets = a.front_sync + 2 * (size_t)a.n_slots;
            const uint32_t nc = (uint32_t)n_counters, q = n_items / nc, r = n_items % nc;
            const uint32_t home = (__builtin_amdgcn_s_getreg((31 << 11) | 20) & 7u) % nc; // XCC_ID
            uint32_t got = 0xFFFFFFFFu;
            for (uint32_t k = 0; k < nc && got == 0xFFFFFFFFu; ++k) {
                const uint32_t x = (home + k) % nc, size = q + (x < r ? 1u : 0u);
                const uint32_t t = __hip_atomic_fetch_add(&tickets[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t < size) got = x * q + (x < r ? x : r) + t;
            }
            s_word = got;
        }
        __syncthreads();
        item = s_word;
        if (item == 0xFFFFFFFFu) return; // (cannot happen: as many tickets as work-groups)
    } else {
        // XCD-aware (gg_device.h): the chunks of one cloud run on one XCD, so its layers / records are cached in ONE L2
        item = xcd_contiguous_item(blockIdx.x + blockIdx.y * gridDim.x, n_items);
    }
    const int cloud = (int)(item / gridDim.x), bx = (int)(item % gridDim.x);
    const CloudParams &cp = params[cloud]; // (a reference: the 12 transform doubles stay in memory unless has_tf)
    const int chunk = bx * 4 + wave;
    const int n = cp.n_points;
    const int nch = (n + a.PW - 1) / a.PW;
    const int groups_of_cloud = max(1, (nch + 3) / 4); // work-groups that hold chunks of this cloud (an empty cloud: one, which scans)
    if (bx >= groups_of_cloud) return; // (uniform; the grid is sized for the batch's largest cloud)
    for (int t = threadIdx.x; t < T; t += 256) lds_tile_rank[t] = a.tile_rank[t];
    uint32_t *hist = lds_hist + wave * words;
    for (int t = lane; t < words; t += 64) hist[t] = 0u;
    __syncthreads();
    const bool active = chunk < nch;

    const float2 *gp2 = gp2_ptr(a, cp.slot);
    const char *pts = reinterpret_cast<const char *>(io.d_points) +
                      (size_t)cp.io_index * io.cloud_stride * (FMT == GG_POINT16 ? 16 : 32);
    uint2 *rec = a.rec + (size_t)cp.slot * a.point_stride;
    const __amdgpu_buffer_rsrc_t ghist = words_rsrc(a.hist + (size_t)cp.slot * a.hist_stride, a.hist_stride);
    const uint32_t row = (uint32_t)chunk * (uint32_t)TP; // this chunk's row of `hist`, in words

    const int base = chunk * a.PW;
    const int end = min(base + a.PW, n);
    if (active) {
        uint32_t n_kept = 0, n_ign = 0, n_outl = 0, n_inmap = 0;
        // ITEMS independent 64-point windows per trip: all point loads are issued before the first classification, so
        // several HBM / L2 round trips (point record, then the old-ground gather) are in flight per lane.
        constexpr int ITEMS = 4;
        // ... and the points of the NEXT trip are requested before this one is classified (unconditional loads at clamped indices; the
        // registers change hands at the loop's back edge, where the wait for them is the wait the next trip would start with)
        PointIn pt[ITEMS];
#pragma unroll
        for (int j = 0; j < ITEMS; ++j) pt[j] = load_point<FMT>(pts, (size_t)min(base + j * 64 + lane, end - 1));
        for (int p0 = base; p0 < end; p0 += 64 * ITEMS) {
            PointIn pt_next[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) pt_next[j] = load_point<FMT>(pts, (size_t)min(p0 + 64 * ITEMS + j * 64 + lane, end - 1));
            bool valid[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) valid[j] = p0 + j * 64 + lane < end;
            if (cp.has_tf) { // N2: cloud still in the sensor frame (uniform branch)
#pragma unroll
                for (int j = 0; j < ITEMS; ++j) transform_point(cp.tf, pt[j].x, pt[j].y, pt[j].z);
            }
            int gi0[ITEMS], gi1[ITEMS];
            bool inmap_[ITEMS];
            float og[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) inmap_[j] = locate_point(a, cp, pt[j], gi0[j], gi1[j]) && valid[j];
            // index math for all windows, then all old-ground gathers in flight together.  The old ground only feeds the entry test
            // of the line-of-sight walk (:243-244), and a map without any confidence above 0.01 -- every map of a cold step -- cannot
            // produce an outlier (:269, classify_point): no walk, so nothing to gather either (a quarter of this kernel's reads).  The
            // load stays unconditional (element 0, one broadcast line, for the lanes with nothing to fetch): a load under a branch
            // costs the warm case its schedule
            const bool need_ground = !cp.no_confidence;
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) og[j] = gp2[(inmap_[j] && need_ground) ? gp_idx(a, gi0[j], gi1[j]) : 0].x; // :243
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) {
                const int p = p0 + j * 64 + lane;
                int cls = GG_CLASS_KEPT;
                bool walk = false;
                if (inmap_[j]) cls = classify_point(a, cp, pt[j], og[j], walk);
                const int n_walk = __popcll(__ballot(walk));
                if (n_walk > WALK_PACKED_MAX) { // (uniform) most lanes are candidates: every lane walks its own ray
                    if (walk && ray_walk_hits_lane(a, cp, gp2, pt[j].x, pt[j].y, pt[j].z)) cls = GG_CLASS_OUTLIER;
                } else if (n_walk > 0) { // (uniform) a few candidates: their steps dealt to the 64 lanes
                    const unsigned long long hits = walk_packed(a, cp, gp2, pt[j].x, pt[j].y, pt[j].z, walk, lane);
                    if ((hits >> lane) & 1ull) cls = GG_CLASS_OUTLIER;
                }
                uint32_t key = KEY_OUTSIDE;
                if (inmap_[j]) key = make_key(a, lds_tile_rank, gi0[j], gi1[j], cls);
                if (valid[j]) rec[p] = make_uint2(__float_as_uint(pt[j].z), key);
                const bool inmap = key != KEY_OUTSIDE;
                // the tile histogram: consecutive returns of a scan line fall into one tile, and same-address LDS atomics execute a
                // lane at a time -- the first lane of every run of equal tiles adds the run's length (gg_device.h lane_run)
                const uint32_t tr = key >> KEY_TILE_SHIFT;
                const LaneRun run = lane_run(inmap ? tr : 0x100000u, lane);
                if (inmap && run.head) {
                    const uint32_t len = (uint32_t)__popcll(run.mask);
                    if (packed) // (uniform)
                        atomicAdd(&hist[tr >> 1], len << ((tr & 1u) * 16u));
                    else
                        atomicAdd(&hist[tr], len);
                }
                const bool emit = inmap && (key & KEY_EMIT_BIT);
                n_inmap += (uint32_t)__popcll(__ballot(inmap));
                n_kept += (uint32_t)__popcll(__ballot(emit && cls == GG_CLASS_KEPT));
                n_ign += (uint32_t)__popcll(__ballot(emit && cls == GG_CLASS_IGNORED));
                n_outl += (uint32_t)__popcll(__ballot(inmap && cls == GG_CLASS_OUTLIER));
            }
#pragma unroll
            for (int j = 0; j < ITEMS; ++j) pt[j] = pt_next[j];
        }

        // the chunk's row of `hist` (zeros in its padding) and its emission counters, 16 bytes at a time
        for (int g = lane; g < TP / 4; g += 64) {
            u32x4 v;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 4 * g + k;
                v[k] = packed ? (hist[t >> 1] >> ((t & 1) * 16)) & 0xFFFFu : hist[t];
            }
            store16_row<SHAPE != FRONT_THREE_LAUNCHES>(ghist, row + 4u * (uint32_t)g, v); // (agent scope when the scan runs in this launch)
        }
        if (lane == 0) {
            const u32x4 v = {n_kept, n_ign, n_outl, n_inmap};
            store16_row<SHAPE != FRONT_THREE_LAUNCHES>(words_rsrc(a.chunk_emit + (size_t)cp.slot * a.emit_stride, a.emit_stride), 4u * (uint32_t)chunk, v);
        }
    }
    if (SHAPE == FRONT_THREE_LAUNCHES) return;

    // ---- the cloud's last work-group scans ----
    drain_vector_memory(); // (every storing wavefront, before the arrival is counted)
    __syncthreads();
    uint32_t *arrivals = a.front_sync + (size_t)cp.slot, *scanned = a.front_sync + (size_t)a.n_slots + (size_t)cp.slot;
    if (threadIdx.x == 0) s_word = __hip_atomic_fetch_add(arrivals, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = s_word == (uint32_t)groups_of_cloud - 1u;
    if (last) {
        scan_cloud<4, true>(a, cp, nch, s_scan);
        if (SHAPE == FRONT_ONE_LAUNCH) {
            drain_vector_memory();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(scanned, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (SHAPE != FRONT_ONE_LAUNCH) return;

    // ---- everybody scatters its own chunks ----
    if (!last) {
        __syncthreads(); // (s_word is reused)
        if (threadIdx.x == 0) {
            uint32_t polls = 0;
            while (__hip_atomic_load(scanned, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && polls < FRONT_WAIT_POLLS) {
                __builtin_amdgcn_s_sleep(16);
                ++polls;
            }
            s_word = polls < FRONT_WAIT_POLLS ? 1u : 0u;
            if (polls >= FRONT_WAIT_POLLS) __hip_atomic_store(a.dev_error, (uint32_t)GG_DEVERR_FRONT_WAIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __syncthreads();
        if (s_word == 0u) return; // (the error word says so; the batch's outputs are void)
    }
    if (!active) return;
    uint2 *sorted = a.sorted + (size_t)cp.slot * a.point_stride;
    if (packed) {
        for (int t = lane; t < words; t += 64) hist[t] = 0u;
        scatter_chunk<true, true>(hist, ghist, row, rec, sorted, base, end, lane);
    } else {
        for (int g = lane; g < TP / 4; g += 64) *reinterpret_cast<u32x4 *>(hist + 4 * g) = load16_agent(ghist, row + 4u * (uint32_t)g);
        scatter_chunk<false, true>(hist, ghist, row, rec, sorted, base, end, lane);
    }
}

size_t classify_lds_bytes(const Arena &a)
{
    const bool packed = a.g.T > PACKED_TILE_COUNTERS_MIN_T && a.PW < 65536;
    return (size_t)4 * (packed ? a.hist_pitch / 2 : a.hist_pitch) * sizeof(uint32_t) + (((size_t)a.g.T * 2 + 15) & ~(size_t)15) + 64;
}

template <int FMT, int SHAPE>
static void launch_classify_as(const Arena &a, const CloudParams *d_params, const BatchIO &io, dim3 grid, size_t lds, int n_counters, hipStream_t s)
{
    hipLaunchKernelGGL((k_classify<FMT, SHAPE>), grid, dim3(256), lds, s, a, d_params, io, n_counters);
}

// work-groups of k_classify<.., FRONT_ONE_LAUNCH> the current device holds at a time (0: unknown)
static int front_resident_groups(size_t lds)
{
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(k_classify<GG_POINT16, FRONT_ONE_LAUNCH>), 256, lds) != hipSuccess) return 0;
    // (the API can overstate what the hardware admits by one work-group per CU, MI355X_MICROARCH.md "Residency": count one less)
    return std::max(0, per_cu - 1) * cus;
}

int launch_classify(const Arena &a, const CloudParams *d_params, const BatchIO &io, int n_clouds, int max_n, hipStream_t s)
{
    const int nch = (max_n + a.PW - 1) / a.PW;
    if (n_clouds == 0) return FRONT_ONE_LAUNCH; // (nothing to do for anybody)
    const int groups = std::max(1, (nch + 3) / 4); // (an empty batch still scans: tile lists, counts)
    dim3 grid(groups, n_clouds);
    const size_t lds = classify_lds_bytes(a);
    // per-wave tile histograms beyond the 64 KiB default (grids above ~1024 cells per side) need the explicit opt-in
    static PerDeviceOnce big_lds;
    if (lds > 64 * 1024)
        big_lds.run([] {
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT16, FRONT_THREE_LAUNCHES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT32, FRONT_THREE_LAUNCHES>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT16, FRONT_SCAN_IN_CLASSIFY>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT32, FRONT_SCAN_IN_CLASSIFY>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT16, FRONT_ONE_LAUNCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            hipFuncSetAttribute(reinterpret_cast<const void *>(k_classify<GG_POINT32, FRONT_ONE_LAUNCH>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        });
    int shape = a.tune_front != FRONT_AUTO ? a.tune_front : FRONT_DEFAULT_SHAPE;
    int n_counters = 8;
    if (shape == FRONT_ONE_LAUNCH) {
        // a waiting work-group needs the rest of its cloud to start: with one ticket counter per XCD at most 8 clouds are partly
        // started, with a single counter one -- the device must hold more work-groups than those can have waiting
        const int resident = front_resident_groups(lds);
        if (resident >= 8 * groups + 8) n_counters = 8;
        else if (resident >= groups + 1) n_counters = 1;
        else shape = FRONT_SCAN_IN_CLASSIFY;
    }
    if (shape != FRONT_THREE_LAUNCHES) hipMemsetAsync(a.front_sync, 0, ((size_t)2 * a.n_slots + 16) * sizeof(uint32_t), s);
    const bool p16 = io.point_format == GG_POINT16;
    if (shape == FRONT_THREE_LAUNCHES)
        p16 ? launch_classify_as<GG_POINT16, FRONT_THREE_LAUNCHES>(a, d_params, io, grid, lds, n_counters, s) : launch_classify_as<GG_POINT32, FRONT_THREE_LAUNCHES>(a, d_params, io, grid, lds, n_counters, s);
    else if (shape == FRONT_SCAN_IN_CLASSIFY)
        p16 ? launch_classify_as<GG_POINT16, FRONT_SCAN_IN_CLASSIFY>(a, d_params, io, grid, lds, n_counters, s) : launch_classify_as<GG_POINT32, FRONT_SCAN_IN_CLASSIFY>(a, d_params, io, grid, lds, n_counters, s);
    else
        p16 ? launch_classify_as<GG_POINT16, FRONT_ONE_LAUNCH>(a, d_params, io, grid, lds, n_counters, s) : launch_classify_as<GG_POINT32, FRONT_ONE_LAUNCH>(a, d_params, io, grid, lds, n_counters, s);
    return shape;
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
    for (size_t w0 = (size_t)blockIdx.x * per_block; w0 < min(words, ((size_t)blockIdx.x + 1) * per_block); w0 += MAXW) {
        const size_t nw = min((size_t)MAXW, min(words, ((size_t)blockIdx.x + 1) * per_block) - w0);
        __syncthreads();
        for (size_t k = threadIdx.x; k < nw; k += blockDim.x) m[k] = valid[w0 + k];
        __syncthreads();
        for (size_t p = threadIdx.x; p < nw * 16; p += blockDim.x) { // pairs of elements
            const size_t i = (w0 << 5) + 2 * p;
            const uint32_t b = (m[p >> 4] >> ((2 * p) & 31)) & 3u; // (bits beyond n are 0)
            if (b == 3u) {
                // streaming (non-temporal) stores: the 1.5 GB a re-initialisation of 1024 maps writes is not read again before the whole insert
                // has run, and written normally it evicts what k_classify is about to stream (k_classify 0.63 -> 0.58 ms per 1024 clouds
                // behind the fill, the step -0.04 ms: profiles/r05a/fill_nontemporal_ab.log)
                typedef float f4_native __attribute__((ext_vector_type(4)));
                const f4_native v4 = {x, y, x, y};
                __builtin_nontemporal_store(v4, reinterpret_cast<f4_native *>(d + i));
            }
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

// gg_reset_maps for maps that stay FRESH (gg_internal.h Arena::gp_bits): the (ground, confidence) layer is not written at all -- the next
// batch's sweep rewrites every cell of rings 1 .. c - 1, reads the reset's pair from ONE padding element instead, and writes the cells no
// sweep visits (ring >= c) itself while it runs (k_sweep<FRESH>, Arena::gp_border).  What is written here: that padding element and the
// slot's written-cell bits, all 0.  (Writing the 2.2 k border cells of every map here -- each in a 128-byte line of its own -- took
// 0.04 of this kernel's 0.05 ms per 1024 maps.)
__global__ __launch_bounds__(256) void k_reset_fresh(const Arena a, int first_slot, float x, float y, int all_ones)
{
    const int slot = first_slot + (int)blockIdx.y;
    float2 *gp2 = gp2_ptr(a, slot);
    // (the slot's bit words are 16-byte aligned and a multiple of 16 bytes long: gg_create)
    uint4 *bits = reinterpret_cast<uint4 *>(a.gp_bits + (size_t)slot * a.gp_bits_stride);
    const int tid = (int)(blockIdx.x * blockDim.x + threadIdx.x), nt = (int)(gridDim.x * blockDim.x);
    const uint32_t v = all_ones ? ~0u : 0u; // (all_ones: measurements, GG_FRESH_MAPS=2 -- every cell written by a fill AND marked)
    for (int w = tid; w < a.gp_bits_words / 2; w += nt) bits[w] = make_uint4(v, v, v, v);
    if (tid == 0) gp2[a.gp_fresh_cell] = make_float2(x, y);
}
void launch_reset_fresh(const Arena &a, int first_slot, int count, float x, float y, hipStream_t s, int all_ones)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_reset_fresh, dim3(count >= 64 ? 2 : 12, count), dim3(256), 0, s, a, first_slot, x, y, all_ones);
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
