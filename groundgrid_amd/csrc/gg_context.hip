// gg_context.hip -- host side of libgroundgrid_hip.so: the C ABI of include/groundgrid_hip.h.
//
// One gg_context = one device arena (single hipMalloc, sized for n_slots independent map states and
// max_points per cloud), one HIP stream, and the tables GroundSegmentation::init precomputes.
// No CPU fallback: every failure is reported through gg_status / gg_last_error.
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstdio>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include <chrono>

#include <dlfcn.h>
#include <sched.h>
#include <unistd.h>

#include "gg_internal.h"
#include "host_helper.h"
#include "sweep_core.h"

using namespace gg;

namespace {

constexpr int PARAM_RING = 4;

// one timed kernel of a profiled launch sequence (GG_FLAG_PROFILE).  Consecutive kernels of a sequence SHARE the event between them -- the
// stop of one is the start of the next (owns_start = false): eight events per batch instead of fourteen (the markers cost the 1024-cloud
// step 0.05 ms) -- so a kernel's time includes the few microseconds the stream idles in front of it.
struct EventPair {
    hipEvent_t start, stop;
    int kernel;
    bool owns_start;
};

} // namespace

struct gg_context {
    int device = 0;
    HostHelper helper;
    hipStream_t stream = nullptr;
    Arena arena{};
    gg_config cfg{};
    gg_geometry geom{};
    int n_slots = 0;
    size_t max_points = 0;
    unsigned flags = 0;
    std::string last_error;

    void *d_arena = nullptr;
    size_t arena_bytes = 0;
    std::vector<float> h_expected;
    std::vector<double> pos_x, pos_y; // per slot map position
    std::vector<char> slot_seen;      // scratch of gg_filter_batch's check of gg_batch.slots
    std::vector<char> no_confidence;  // per slot: groundpatch <= 0.01 everywhere for sure (set by gg_reset_map, cleared by any writer)
    // per slot: the map is FRESH -- gg_reset_maps left the interior of its (ground, confidence) layer unwritten (gg_internal.h Arena::gp_bits);
    // a batch of at least FRESH_MIN_CLOUDS fresh maps sweeps them as they are, anything else that touches the layer fills it first (make_real)
    std::vector<char> fresh;
    std::vector<float> fresh_z;       // ... and the ground height its cells hold by definition
    bool fresh_enabled = true;        // env GG_FRESH_MAPS=0 / tuning "fresh_maps": gg_reset_maps writes every cell, as before
    // GG_FLAG_MINIMAL_LAYERS: the slot's last cloud left maxGroundHeight / groundCandidates / planeDist unwritten; a reader of one of
    // them has them computed first, from what that call left in the slot's buffers and with its parameters (ensure_lazy_layers)
    bool probe_unordered_streams = false; // measurement only: batches and resets on different caller streams are NOT ordered by the library
    std::vector<char> lazy_pending;
    std::vector<gg::CloudParams> lazy_params;

    gg_conventions conv{};
    gg::sweep::Params sweep_params{};

    // cross-stream ordering (include/groundgrid_hip.h, gg_filter_batch): `map_event` is recorded on ctx->stream after every
    // map mutation enqueued there, `batch_event` on the launch stream after every batch
    hipEvent_t map_event = nullptr, batch_event = nullptr;
    hipEvent_t gather_event = nullptr;       // recorded behind the last gg_allgather_label_masks
    hipStream_t gather_stream = nullptr;
    const uint8_t *gather_lo = nullptr, *gather_hi = nullptr; // ... and the send buffer it reads (null: none pending)
    bool map_event_pending = false;          // a mutation was enqueued on ctx->stream since the last batch waited for it
    hipStream_t last_batch_stream = nullptr; // stream batch_event was last recorded on
    bool have_batch_event = false;
    // GG_FLAG_CONCURRENT_HALVES: the clouds (and map re-initialisations) of the upper half of the slots run on half_stream; half_done is
    // recorded there behind each of them, half_fork on the caller's stream in front (the side stream sees what the caller enqueued before)
    hipStream_t half_stream = nullptr;
    bool layer_copy_failed = false; // a download of the fused filter + layers call could not be enqueued (reported by the call)
    hipEvent_t half_fork = nullptr, half_done = nullptr;
    bool have_half_event = false;
    // Output rows are indexed by a cloud's position in the batch, its half by its slot: two divided batches that are not joined (same caller
    // stream) may write the same row of the same buffer from different streams.  The last few divided batches' output ranges and
    // row-to-half maps are remembered; a batch that shares a range with one of them under another map joins first (enqueue_batch).
    struct HalvesRecord {
        const uint8_t *lo[6] = {}, *hi[6] = {};
        uint64_t map_hash = 0;
    };
    HalvesRecord halves_hist[4];
    int halves_hist_n = 0;
    int halves_min_clouds = 256;
    // gg_filter_cloud / _async / _layers: k_label writes counts, index and labels straight into the pinned host block (posted writes over
    // the link while the kernel runs) instead of into HBM with a copy behind the kernel: one transfer and one stream hop less per call
    int results_direct = 1;
    int upload_pieces = 1; // the input cloud is packed and uploaded in this many pieces (the copy of one travels while the next is packed:
                           // measured 1 / 2 / 3 / 4 pieces: 0.472 / 0.478 / 0.488 / 0.494 ms per synchronous call -- a piece's split and copy
                           // call cost more than its overlap gives)
    bool probe_no_fork = false;
    hipEvent_t ring_done2[4]{};
    bool ring_used2[4]{};

    // pipelined host entry point (gg_filter_cloud_async / _wait): GG_ASYNC_DEPTH staging sets + a copy stream each way
    struct AsyncSlot {
        gg_point16 *h_pts = nullptr, *d_pts = nullptr;
        uint8_t *h_labels = nullptr, *d_labels = nullptr;
        int32_t *h_index = nullptr, *d_index = nullptr;
        int32_t *h_counts = nullptr, *d_counts = nullptr;
        int32_t *hd_counts = nullptr; // h_counts as the device addresses it (results_direct)
        hipEvent_t uploaded = nullptr, computed = nullptr, downloaded = nullptr;
        const gg_point32 *cloud = nullptr;
        size_t n = 0;
        bool has_tf = false;
        double tf[12]{};
        int ticket = -1;
    } async_slot[GG_ASYNC_DEPTH];
    hipStream_t h2d_stream = nullptr, d2h_stream = nullptr;
    int next_ticket = 0, oldest_ticket = 0;
    int slot_of_ticket[GG_ASYNC_DEPTH] = {}; // staging set of an outstanding ticket (a synchronous call always takes set 0)
    int next_label_shift = 0;                // enqueue_ticket -> enqueue_batch: CloudParams::label_shift of the one cloud
    double host_t[4] = {0, 0, 0, 0}; // GG_HOST_TIMING: pack+upload, enqueue, wait, copy-out
    long host_calls = 0;

    // per-call parameter ring (pinned host + device)
    CloudParams *h_params = nullptr; // [PARAM_RING][n_slots] pinned
    CloudParams *d_params = nullptr; // [PARAM_RING][n_slots]
    hipEvent_t ring_done[PARAM_RING]{};
    bool ring_used[PARAM_RING]{};
    int ring_next = 0;

    // staging for the host-buffer entry point
    gg_point16 *h_stage_pts = nullptr; // pinned [max_points]
    uint8_t *h_stage_labels = nullptr; // pinned
    int32_t *h_stage_index = nullptr;  // pinned
    int32_t *h_stage_counts = nullptr; // pinned [4]
    gg_point16 *d_stage_pts = nullptr;
    uint8_t *d_stage_labels = nullptr;
    int32_t *d_stage_index = nullptr;
    int32_t *d_stage_counts = nullptr;
    uint8_t *d_stage_class = nullptr;
    int32_t *d_stage_cell = nullptr;
    float *d_scroll_scratch = nullptr; // 2 layers
    float *d_image = nullptr;          // 3 * C floats (wire-format images)
    float *d_planes = nullptr;         // GG_NUM_LAYERS * Cpad floats: dense planes of gg_get_layers (allocated on first use)
    float *h_planes = nullptr;         // ... and their pinned landing zone on the host (one download for all requested layers)
    float *d_bounds = nullptr;         // 2 floats
    uint8_t *d_pc2 = nullptr, *h_pc2 = nullptr; // gg_filter_cloud_pc2_out: [64 B counts][max_points * 18 B records], device and pinned host (allocated on first use)
    volatile uint32_t *h_dev_error = nullptr;  // host view of Arena::dev_error (mapped pinned memory)
    unsigned long long *d_sweep_dbg = nullptr; // GG_SWEEP_TIMING=1: cycle counters of the sweep's wavefronts (cloud 0 of a batch)

    // One cloud per call -- the reference's own call shape (src/GroundGridNodelet.cpp:196) -- replays a captured HIP graph: the seven
    // launches (and, for the fused filter + layers call, the layer extraction and downloads on a side branch) are handed to the device
    // as ONE submission.  A captured sequence bakes in everything but the cloud's parameter record, which every replay reads from the
    // same device address (d_gparams, refreshed by a copy in front of the graph): key = what else the launchers were given.
    struct GraphKey {
        BatchIO io;
        int max_n, slot;
        unsigned flags;
        int eigen;
        unsigned long long generation; // bumped by whatever changes the kernels' by-value arguments (configuration, tuning, conventions)
        unsigned layer_mask;           // fused call: layers extracted and downloaded inside the graph ...
        void *layer_dst[GG_NUM_LAYERS]; // ... and where to (pinned staging or the caller's registered planes)
        hipStream_t stream;
    };
    struct GraphEntry {
        GraphKey key;
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        int uses = 0;
        unsigned long long last_use = 0;
    };
    std::vector<GraphEntry> graphs;
    unsigned long long graph_generation = 1, graph_clock = 0;
    bool graphs_enabled = false; // opt-in (GG_GRAPH=1, gg_debug_set_tuning("graphs", 1)): measured on MI355X / ROCm 7.0.2 a replay is no faster than the
                                 // seven eager launches -- one cloud, device-resident input: 0.420 against 0.416 ms, the synchronous host call 0.589
                                 // against 0.571 (profiles/r05a/host_call_probe*.json): the launches of a call already queue up behind the first
                                 // kernel, there are no gaps left for a graph to close
    long graph_replays = 0, graph_captures = 0;
    CloudParams *d_gparams = nullptr;
    hipEvent_t fork_event = nullptr, join_event = nullptr; // the side branch of the fused filter + layers call
    hipEvent_t layers_event = nullptr;                     // ... and its late layers (behind the results on the context's stream)
    std::vector<std::pair<char *, size_t>> registered;     // gg_host_register: host ranges downloads may land in directly

    // profiling
    std::vector<EventPair> pending;
    std::vector<hipEvent_t> free_single_events;
    double k_ms[GG_NUM_KERNELS]{};
    int64_t k_launches[GG_NUM_KERNELS]{};
};

namespace {

int fail(gg_context *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        ctx->last_error = buf;
    }
    return code;
}

#define HIPCHK(ctx, call)                                                      \
    do {                                                                       \
        hipError_t e__ = (call);                                               \
        if (e__ != hipSuccess) return fail((ctx), GG_ERR_HIP, #call, e__);     \
    } while (0)

// After a synchronisation: did a kernel give up a bounded wait (Arena::dev_error)?  The reference has no error path at all
// (SURVEY 5); a library that spins inside kernels must at least say so instead of hanging or returning garbage silently.
// The word is read AND cleared here: the report is about the batches enqueued since the last report -- their outputs are void, the
// map states they touched should be re-initialised (gg_reset_map) -- and the context goes on working (a sticky word turned one
// spurious timeout into GG_ERR_HIP from every later call, gg_reset_map + gg_synchronize included).
int device_error(gg_context *ctx)
{
    const uint32_t code = ctx->h_dev_error ? *ctx->h_dev_error : 0u;
    if (code == GG_DEVERR_NONE) return GG_OK;
    *ctx->h_dev_error = GG_DEVERR_NONE;
    return fail(ctx, GG_ERR_HIP, code == GG_DEVERR_FRONT_WAIT ? "k_classify: a cloud's tile scan never completed (bounded wait ran out); the outputs of that batch are void"
                                 : code == GG_DEVERR_SCAN_WAIT ? "k_scan: the sums of an earlier part of a cloud never arrived (bounded wait ran out); the outputs of that batch are void"
                                 : code == GG_DEVERR_SWEEP_WAIT ? "k_sweep: a hand-over between work-groups never arrived (bounded wait ran out); the outputs of that batch are void"
                                                                : "a kernel reported an unknown device-side error");
}
#define SYNCCHK(ctx, call)                                   \
    do {                                                     \
        HIPCHK(ctx, call);                                   \
        const int rc__ = device_error(ctx);                  \
        if (rc__ != GG_OK) return rc__;                      \
    } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

uint32_t morton2(uint32_t x, uint32_t y)
{
    auto spread = [](uint32_t v) {
        v &= 0xFFFFu;
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

void make_dev_config(const gg_config &c, DevConfig &d)
{
    d.point_count_cell_variance_threshold = c.point_count_cell_variance_threshold;
    d.max_ring = c.max_ring;
    d.outlier_tolerance = c.outlier_tolerance;
    d.gpd_min_point_count_threshold = c.ground_patch_detection_minimum_point_count_threshold;
    d.patch_size_change_distance_sq = c.patch_size_change_distance * c.patch_size_change_distance; // pow(x, 2.0)
    d.occupied_cells_decrease_factor = c.occupied_cells_decrease_factor;
    d.occupied_cells_point_count_factor = c.occupied_cells_point_count_factor;
    d.occupied_cells_point_count_factor_x2 = c.occupied_cells_point_count_factor * (double)2.0f; // :387
    d.min_outlier_detection_ground_confidence = c.min_outlier_detection_ground_confidence;
    d.distance_factor_sq = c.distance_factor * c.distance_factor;                                 // :369
    d.minimum_distance_factor_sq = c.minimum_distance_factor * c.minimum_distance_factor;
    const double m10 = c.minimum_distance_factor * 10;
    d.minimum_distance_factor_x10_sq = m10 * m10;
    d.min_dist_fac = c.minimum_distance_factor * 5; // :154
    d.min_point_height_thres = c.miminum_point_height_threshold;
    d.min_point_height_obs_thres = c.minimum_point_height_obstacle_threshold;
}

hipStream_t pick_stream(gg_context *ctx, void *stream)
{
    if (stream == GG_STREAM_DEFAULT) return (hipStream_t) nullptr; // the legacy default stream
    return stream ? (hipStream_t)stream : ctx->stream;
}

// `st` follows the half of every earlier batch that ran on the library's side stream (GG_FLAG_CONCURRENT_HALVES)
int stream_waits_for_second_half(gg_context *ctx, hipStream_t st)
{
    if (ctx->have_half_event && st != ctx->half_stream) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->half_done, 0));
    return GG_OK;
}
// Entry points that read or write map state on ctx->stream call this first: the context's stream waits for the last batch
// that ran on another stream (gg_get_layer after a batch on a caller stream read stale layers).
// FRESH maps (gg_context::fresh) among the slots [first, first + n) become real on stream `st`: the fill gg_reset_maps left out (:71-75).
// The caller has ordered `st` behind the reset.
int make_real(gg_context *ctx, int first, int n, hipStream_t st)
{
    const Arena &a = ctx->arena;
    for (int s = first; s < first + n;) {
        if (!ctx->fresh[s]) {
            ++s;
            continue;
        }
        int e = s + 1;
        while (e < first + n && ctx->fresh[e] && ctx->fresh_z[e] == ctx->fresh_z[s]) ++e; // one strided fill per run of equal heights
        launch_fill2_strided(gp2_ptr(a, s), (size_t)a.gpl.elems, a.gp2_stride, e - s, ctx->fresh_z[s], (float)0.0000001, a.gp_valid, st);
        for (int k = s; k < e; ++k) ctx->fresh[k] = 0;
        s = e;
    }
    HIPCHK(ctx, hipGetLastError());
    return GG_OK;
}

// (maps_real = false: the caller touches no layer -- a re-initialisation leaves the other maps as they are)
int own_stream_waits_for_batches(gg_context *ctx, bool maps_real = true)
{
    if (ctx->have_batch_event && ctx->last_batch_stream != ctx->stream)
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->batch_event, 0));
    if (const int rc = stream_waits_for_second_half(ctx, ctx->stream)) return rc;
    if (!maps_real) return GG_OK;
    // whatever the context's own stream does next with map state (getters, setters, scrolls, stages, single clouds) finds real layers
    bool any = false;
    for (int s = 0; s < ctx->n_slots && !any; ++s) any = ctx->fresh[s] != 0;
    if (!any) return GG_OK;
    if (const int rc = make_real(ctx, 0, ctx->n_slots, ctx->stream)) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->map_event, ctx->stream)); // (own_stream_mutated_map: a batch on another stream waits for the fill)
    ctx->map_event_pending = true;
    return GG_OK;
}
// ... and this after enqueueing a mutation of map state on ctx->stream (reset, move, set_layer): the next batch on any
// other stream waits for it.
int own_stream_mutated_map(gg_context *ctx)
{
    HIPCHK(ctx, hipEventRecord(ctx->map_event, ctx->stream));
    ctx->map_event_pending = true;
    return GG_OK;
}

// rocTX ranges named after the reference's four stage timers (src/GroundSegmentation.cpp:124 "rasterization", :138 "patch
// detection", :144 "interpolation", :194 "segmentation") around the launch groups of a batch, so that a rocprofv3
// --marker-trace reads in the reference's terms.  The library is looked up at run time (no link-time dependency, nothing happens
// when no profiler is attached); GG_ROCTX=0 turns it off.
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx()
    {
        const char *e = getenv("GG_ROCTX");
        if (e && atoi(e) == 0) return;
        for (const char *name : {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"}) {
            void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (!h) continue;
            push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
            pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
            if (push && pop) return;
            push = nullptr;
            pop = nullptr;
        }
    }
};
struct StageRange {
    static Roctx &lib()
    {
        static Roctx r;
        return r;
    }
    bool on;
    explicit StageRange(const char *stage) : on(lib().push != nullptr)
    {
        if (on) lib().push(stage);
    }
    ~StageRange()
    {
        if (on) lib().pop();
    }
    StageRange(const StageRange &) = delete;
};

struct Profiler {
    gg_context *ctx;
    hipStream_t s;
    bool on;
    EventPair cur{};
    hipEvent_t prev = nullptr; // the stop event of the kernel before, recorded on `s` by this sequence
    hipEvent_t take()
    {
        hipEvent_t e = nullptr;
        if (!ctx->free_single_events.empty()) {
            e = ctx->free_single_events.back();
            ctx->free_single_events.pop_back();
        } else
            hipEventCreate(&e);
        return e;
    }
    void begin(int k)
    {
        if (!on) return;
        cur.kernel = k;
        cur.owns_start = prev == nullptr;
        if (prev)
            cur.start = prev;
        else {
            cur.start = take();
            hipEventRecord(cur.start, s);
        }
    }
    void end()
    {
        if (!on) return;
        cur.stop = take();
        hipEventRecord(cur.stop, s);
        ctx->pending.push_back(cur);
        prev = cur.stop;
    }
};

int drain_profile(gg_context *ctx)
{
    for (auto &p : ctx->pending) {
        HIPCHK(ctx, hipEventSynchronize(p.stop));
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, p.start, p.stop));
        ctx->k_ms[p.kernel] += ms;
        ctx->k_launches[p.kernel] += 1;
    }
    for (auto &p : ctx->pending) { // (a shared event is the stop of exactly one entry)
        if (p.owns_start) ctx->free_single_events.push_back(p.start);
        ctx->free_single_events.push_back(p.stop);
    }
    ctx->pending.clear();
    return GG_OK;
}

// layers the fused filter + layers call (gg_filter_cloud_layers) extracts and downloads inside the launch sequence
struct LayerPlan {
    unsigned mask = 0u;                 // bit per gg_layer
    void *dst[GG_NUM_LAYERS] = {};      // where layer l lands: the caller's registered plane, or its plane of the pinned staging block
};
// the eight per-call layers that are final once k_reduce has run (K3 / K4 / K5 only read them; `points` is rewritten by K4 / K5,
// ground / groundpatch by K3 / K4): their extraction and download overlap the terrain sweep on a side branch
constexpr unsigned EARLY_LAYERS = (1u << GG_LAYER_MINGROUNDHEIGHT) | (1u << GG_LAYER_MAXGROUNDHEIGHT) | (1u << GG_LAYER_GROUNDCANDIDATES) | (1u << GG_LAYER_PLANEDIST) |
                                  (1u << GG_LAYER_M2) | (1u << GG_LAYER_MEANVARIANCE) | (1u << GG_LAYER_POINTSRAW) | (1u << GG_LAYER_VARIANCE);

// extraction kernel + one download per layer of `mask` on stream `st`.  The early group owns planes 0..7 of d_planes, the late group
// (ground, groundpatch, points) planes 8..10: the two never share one.
void enqueue_layer_downloads(gg_context *ctx, const Arena &a, int slot, unsigned mask, const LayerPlan &plan, hipStream_t st)
{
    if (!mask) return;
    const size_t plane = align_up((size_t)a.g.C * 4, 256) / 4;
    float *base = ctx->d_planes + (size_t)((mask & EARLY_LAYERS) ? 0 : 8) * plane;
    launch_layers_extract(a, slot, mask, base, plane, st); // (plane k of `base` = the k-th layer of `mask`)
    int k = 0;
    for (int l = 0; l < GG_NUM_LAYERS; ++l) {
        if (!((mask >> l) & 1u)) continue;
        if (hipMemcpyAsync(plan.dst[l], base + (size_t)k * plane, (size_t)a.g.C * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) ctx->layer_copy_failed = true;
        ++k;
    }
}

// the kernels of one batched filter_cloud call (and the layer branch of the fused call), on stream s
void launch_sequence(gg_context *ctx, const Arena &a, const CloudParams *dp, const BatchIO &io, int nb, int max_n, hipStream_t s, const LayerPlan *plan, int slot0,
                     bool profile)
{
    Profiler prof{ctx, s, profile};
    {
        StageRange stage("rasterization"); // insert_cloud, :98-124
        // the front end in one, two or three launches (k1_classify.hip): what launch_classify did not do itself follows here
        prof.begin(GG_K_CLASSIFY);
        const int front = launch_classify(a, dp, io, nb, max_n, s);
        prof.end();
        if (front == FRONT_THREE_LAUNCHES) {
            prof.begin(GG_K_SCAN);
            launch_scan(a, dp, nb, s);
            prof.end();
        }
        if (front != FRONT_ONE_LAUNCH) {
            prof.begin(GG_K_SCATTER);
            launch_scatter(a, dp, nb, max_n, s);
            prof.end();
        }
        prof.begin(GG_K_REDUCE);
        launch_reduce(a, dp, nb, s);
        prof.end();
    }
    const bool side = plan && (plan->mask & EARLY_LAYERS);
    if (side) { // fork: the layers k_reduce has just finished travel while the stencil and the sweep run
        hipEventRecord(ctx->fork_event, s);
        hipStreamWaitEvent(ctx->d2h_stream, ctx->fork_event, 0);
        enqueue_layer_downloads(ctx, a, slot0, plan->mask & EARLY_LAYERS, *plan, ctx->d2h_stream);
        hipEventRecord(ctx->join_event, ctx->d2h_stream);
    }
    {
        StageRange stage("patch detection"); // detect_ground_patches, :126-138
        prof.begin(GG_K_PATCH);
        launch_patch(a, dp, nb, s);
        prof.end();
    }
    prof.begin(GG_K_SPIRAL);
    {
        StageRange stage("interpolation"); // spiral_ground_interpolation, :141-144
        sweep::Params sp = ctx->sweep_params;
        sp.decrease = ctx->cfg.occupied_cells_decrease_factor;
        sp.inv_decrease = 1.0 / sp.decrease;
        sp.decay_fast = sp.decrease >= 1.25 && sp.decrease < 1e300;
        launch_sweep(a, sp, dp, nb, s, ctx->d_sweep_dbg);
    }
    prof.end();
    {
        StageRange stage("segmentation"); // the label loop, :146-194
        prof.begin(GG_K_LABEL);
        launch_label(a, dp, io, nb, max_n, s);
        prof.end();
    }
    if (side) hipStreamWaitEvent(s, ctx->join_event, 0); // join (the late layers -- ground, groundpatch, points -- follow the results: enqueue_ticket)
}

void drop_graphs(gg_context *ctx)
{
    for (auto &e : ctx->graphs) {
        if (e.exec) hipGraphExecDestroy(e.exec);
        if (e.graph) hipGraphDestroy(e.graph);
    }
    ctx->graphs.clear();
}

// One cloud per call replays a captured graph (gg_context::GraphKey); everything else, and the first call of a kind, launches eagerly.
int launch_or_replay(gg_context *ctx, const Arena &a, const CloudParams *hp, CloudParams *dp, const BatchIO &io, int nb, int max_n, hipStream_t s, const LayerPlan *plan)
{
    const bool profile = (ctx->flags & GG_FLAG_PROFILE) != 0;
    // (the fused filter + layers call stays eager: its side branch runs on a second stream beside the sweep, and a captured graph
    // executed the branch's nodes one after the other -- 0.77 against 0.72 ms per call, profiles/r05a)
    const bool eligible = nb == 1 && !plan && ctx->graphs_enabled && !profile && !ctx->d_sweep_dbg && s != nullptr && !a.k2_debug && !a.k3_debug && !a.k5_debug && !a.k2_skip;
    if (!eligible) {
        HIPCHK(ctx, hipMemcpyAsync(dp, hp, sizeof(CloudParams) * nb, hipMemcpyHostToDevice, s));
        launch_sequence(ctx, a, dp, io, nb, max_n, s, plan, hp[0].slot, profile);
        return GG_OK;
    }
    gg_context::GraphKey key;
    memset(&key, 0, sizeof key); // (padding included: keys are compared bytewise)
    key.io = io;
    // the grids of a captured sequence cover the largest cloud the buffers can hold: work-groups beyond this cloud's chunks leave at once
    key.max_n = (int)std::min(ctx->max_points, io.cloud_stride);
    key.slot = hp[0].slot;
    key.flags = a.flags;
    key.eigen = a.eigen_reduction;
    key.generation = ctx->graph_generation;
    key.stream = s;
    if (plan) {
        key.layer_mask = plan->mask & EARLY_LAYERS; // (what the captured sequence itself downloads)
        for (int l = 0; l < GG_NUM_LAYERS; ++l) key.layer_dst[l] = ((key.layer_mask >> l) & 1u) ? plan->dst[l] : nullptr;
    }
    gg_context::GraphEntry *entry = nullptr;
    for (auto &e : ctx->graphs)
        if (memcmp(&e.key, &key, sizeof key) == 0) entry = &e;
    if (!entry) {
        if (ctx->graphs.size() >= 8) { // evict the least recently used
            size_t victim = 0;
            for (size_t k = 1; k < ctx->graphs.size(); ++k)
                if (ctx->graphs[k].last_use < ctx->graphs[victim].last_use) victim = k;
            if (ctx->graphs[victim].exec) hipGraphExecDestroy(ctx->graphs[victim].exec);
            if (ctx->graphs[victim].graph) hipGraphDestroy(ctx->graphs[victim].graph);
            ctx->graphs.erase(ctx->graphs.begin() + (long)victim);
        }
        ctx->graphs.emplace_back();
        entry = &ctx->graphs.back();
        entry->key = key;
    }
    entry->last_use = ++ctx->graph_clock;
    // the parameter record of THIS cloud, at the address every capture reads it from (stream order: behind the previous replay)
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_gparams, hp, sizeof(CloudParams), hipMemcpyHostToDevice, s));
    if (entry->exec) {
        HIPCHK(ctx, hipGraphLaunch(entry->exec, s));
        ++ctx->graph_replays;
        return GG_OK;
    }
    if (entry->uses++ == 0) { // first call of a kind: eager (per-device one-time settings of the launchers happen here, outside any capture)
        launch_sequence(ctx, a, ctx->d_gparams, io, 1, key.max_n, s, plan, key.slot, false);
        return GG_OK;
    }
    HIPCHK(ctx, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    launch_sequence(ctx, a, ctx->d_gparams, io, 1, key.max_n, s, plan, key.slot, false);
    hipGraph_t graph = nullptr;
    const hipError_t e_end = hipStreamEndCapture(s, &graph);
    if (e_end != hipSuccess || !graph) {
        (void)hipGetLastError();
        ctx->graphs_enabled = false; // (a runtime that cannot capture this sequence: stay eager)
        launch_sequence(ctx, a, ctx->d_gparams, io, 1, key.max_n, s, plan, key.slot, false);
        return GG_OK;
    }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess || !exec) {
        (void)hipGetLastError();
        hipGraphDestroy(graph);
        ctx->graphs_enabled = false;
        launch_sequence(ctx, a, ctx->d_gparams, io, 1, key.max_n, s, plan, key.slot, false);
        return GG_OK;
    }
    entry->graph = graph;
    entry->exec = exec;
    ++ctx->graph_captures;
    HIPCHK(ctx, hipGraphLaunch(exec, s));
    ++ctx->graph_replays;
    return GG_OK;
}

// which half of a context's slots a map belongs to (GG_FLAG_CONCURRENT_HALVES): the upper half runs on the library's side stream
bool second_half_slot(const gg_context *ctx, int slot) { return slot >= (ctx->n_slots + 1) / 2; }
bool halves_enabled(const gg_context *ctx)
{
    return (ctx->flags & GG_FLAG_CONCURRENT_HALVES) && !(ctx->flags & GG_FLAG_PROFILE) && !ctx->d_sweep_dbg && !ctx->arena.k2_debug && ctx->half_stream;
}

// enqueue the seven kernels of one batched filter_cloud call
// batch_entry: the call is gg_filter_batch (device-resident clouds, no layer asked for): the three published-only layers are left to
// their first reader unless GG_FLAG_EAGER_LAYERS says otherwise
int enqueue_batch(gg_context *ctx, const gg_batch *b, hipStream_t s, const LayerPlan *plan = nullptr, bool batch_entry = false)
{
    const bool lazy = (ctx->flags & GG_FLAG_MINIMAL_LAYERS) || (batch_entry && !(ctx->flags & GG_FLAG_EAGER_LAYERS));
    const unsigned eff_flags = lazy ? (ctx->flags | GG_FLAG_MINIMAL_LAYERS) : (ctx->flags & ~(unsigned)GG_FLAG_MINIMAL_LAYERS);
    const int nb = b->n_clouds;
    if (nb == 0) return GG_OK;
    // parameter ring slot
    const int g = ctx->ring_next;
    ctx->ring_next = (g + 1) % PARAM_RING;
    if (ctx->ring_used[g]) HIPCHK(ctx, hipEventSynchronize(ctx->ring_done[g]));
    if (ctx->ring_used2[g]) HIPCHK(ctx, hipEventSynchronize(ctx->ring_done2[g]));
    ctx->ring_used2[g] = false;
    CloudParams *hp = ctx->h_params + (size_t)g * ctx->n_slots;
    CloudParams *dp = ctx->d_params + (size_t)g * ctx->n_slots;
    // Two halves side by side?  The clouds of the lower slots first in the parameter array, then the others: each half is a batch of its own
    int n_first = nb;
    // (not on the legacy default stream: its implicit synchronisation with every other stream makes the two halves slower than one
    // sequence -- 5.75 against 5.29 ms per 1024 clouds, profiles/r05a/halves_probe_default_stream.json)
    const bool split_wanted = halves_enabled(ctx) && !plan && nb >= ctx->halves_min_clouds && s != ctx->half_stream && s != nullptr;
    if (split_wanted) {
        n_first = 0;
        for (int i = 0; i < nb; ++i) n_first += second_half_slot(ctx, b->slots ? b->slots[i] : b->first_slot + i) ? 0 : 1;
    }
    const bool split = split_wanted && n_first > 0 && n_first < nb;
    if (!split) n_first = nb;
    int max_n[2] = {0, 0}, at[2] = {0, n_first};
    for (int i = 0; i < nb; ++i) {
        const int slot = b->slots ? b->slots[i] : b->first_slot + i;
        const int half = split && second_half_slot(ctx, slot) ? 1 : 0;
        CloudParams &p = hp[at[half]++];
        p.slot = slot;
        p.n_points = b->n_points[i];
        p.ox = b->origins[i * 3 + 0];
        p.oy = b->origins[i * 3 + 1];
        p.oz = b->origins[i * 3 + 2];
        p.base_z = (float)b->base_z[i];
        p.pos_x = ctx->pos_x[slot];
        p.pos_y = ctx->pos_y[slot];
        p.has_tf = b->transforms ? 1 : 0;
        p.no_confidence = ctx->no_confidence[slot] ? 1 : 0;
        p.fresh = 0;
        ctx->no_confidence[slot] = 0; // the sweep of this call writes confidences
        for (int k = 0; k < 12; ++k) p.tf[k] = b->transforms ? b->transforms[(size_t)i * 12 + k] : 0.0;
        p.label_shift = nb == 1 ? ctx->next_label_shift : 0;
        p.io_index = i;
        max_n[half] = std::max(max_n[half], p.n_points);
        ctx->lazy_pending[slot] = lazy ? 1 : 0;
        if (lazy) ctx->lazy_params[slot] = p;
    }
    // FRESH maps (gg_context::fresh): a (half) launch whose maps are all fresh, and large enough for the plain k_sweep, takes them as they
    // are -- k_patch marks what it writes, the sweep reads nothing else of the layer --; in any other launch they are filled first
    bool fresh_half[2] = {false, false};
    for (int h = 0; h < (split ? 2 : 1); ++h) {
        const int lo = h ? n_first : 0, hi = h ? nb : n_first;
        bool all = ctx->fresh_enabled && !plan && !ctx->d_sweep_dbg && sweep_takes_fresh(ctx->arena, ctx->sweep_params, hi - lo);
        for (int i = lo; i < hi && all; ++i) all = ctx->fresh[hp[i].slot] != 0;
        fresh_half[h] = all;
        if (all)
            for (int i = lo; i < hi; ++i) hp[i].fresh = 1, ctx->fresh[hp[i].slot] = 0;
    }
    auto fill_fresh = [&](int lo, int hi, hipStream_t on) -> int { // (the stream is ordered behind the reset by now)
        for (int i = lo; i < hi;) { // one fill per run of consecutive slots (make_real cuts a run where the heights differ)
            int e = i + 1;
            while (e < hi && hp[e].slot == hp[e - 1].slot + 1) ++e;
            if (const int rc = make_real(ctx, hp[i].slot, e - i, on)) return rc;
            i = e;
        }
        return GG_OK;
    };
    // order this batch after everything that touched map state on the context's stream, and after an earlier batch that ran
    // on another stream
    if (s != ctx->stream && ctx->map_event_pending) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->map_event, 0));
    if (ctx->have_batch_event && ctx->last_batch_stream != s && !ctx->probe_unordered_streams) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->batch_event, 0));
    // ... and after the second half of earlier batches -- unless this one is divided the same way: a slot is only ever touched from its
    // half's stream, so the halves of consecutive batches on one caller stream need not meet
    bool joined = false;
    if (!(split && ctx->last_batch_stream == s)) {
        if (const int rc = stream_waits_for_second_half(ctx, s)) return rc;
        joined = true;
    }
    if (split) {
        // ... but the halves of consecutive batches DO meet in the output buffers when a row changes its half (rows follow the batch
        // position, halves the slot): join the two streams once, in both directions, before such a batch
        gg_context::HalvesRecord now;
        const size_t per = b->cloud_stride;
        const uint8_t *p6[6] = {(const uint8_t *)b->d_labels, (const uint8_t *)b->d_out_index, (const uint8_t *)b->d_out_clouds, (const uint8_t *)b->d_out_counts,
                                (const uint8_t *)b->d_label_masks, (const uint8_t *)b->d_out_pc2};
        const size_t bytes6[6] = {(size_t)nb * per, (size_t)nb * per * 4, (size_t)nb * per * sizeof(gg_point32), (size_t)nb * 64, (size_t)nb * ((per + 3) / 4), (size_t)nb * per * 18};
        for (int k = 0; k < 6; ++k) now.lo[k] = p6[k], now.hi[k] = p6[k] ? p6[k] + bytes6[k] : nullptr;
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < nb; ++i) h = (h ^ (uint64_t)(second_half_slot(ctx, b->slots ? b->slots[i] : b->first_slot + i) ? 2 * i + 1 : 2 * i)) * 1099511628211ull;
        now.map_hash = h ^ (uint64_t)nb;
        bool clash = false;
        for (int r = 0; r < ctx->halves_hist_n && !joined && !clash; ++r) {
            const gg_context::HalvesRecord &o = ctx->halves_hist[r];
            if (o.map_hash == now.map_hash) continue; // every row on the stream it was on: ordered by the streams themselves
            for (int x = 0; x < 6 && !clash; ++x)
                for (int y = 0; y < 6 && !clash; ++y) clash = now.lo[x] && o.lo[y] && now.lo[x] < o.hi[y] && o.lo[y] < now.hi[x];
        }
        if (clash) {
            if (const int rc = stream_waits_for_second_half(ctx, s)) return rc;
            if (ctx->have_batch_event) HIPCHK(ctx, hipStreamWaitEvent(ctx->half_stream, ctx->batch_event, 0));
            joined = true;
        }
        if (joined) ctx->halves_hist_n = 0; // (everything before is ordered on both streams now)
        if (ctx->halves_hist_n == 4) {
            for (int r = 1; r < 4; ++r) ctx->halves_hist[r - 1] = ctx->halves_hist[r];
            ctx->halves_hist_n = 3;
        }
        ctx->halves_hist[ctx->halves_hist_n++] = now;
    } else if (joined)
        ctx->halves_hist_n = 0;
    if (ctx->gather_lo) { // an all-gather still reads label masks: a batch on another stream that rewrites them waits for it
        auto overlaps = [&](const uint8_t *p, size_t bytes) { return p && p < ctx->gather_hi && p + bytes > ctx->gather_lo; };
        const size_t per_cloud = b->cloud_stride;
        const bool hit = overlaps(b->d_label_masks, (size_t)nb * ((per_cloud + 3) / 4)) || overlaps(b->d_labels, (size_t)nb * per_cloud);
        if (ctx->gather_stream != s && hit) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->gather_event, 0));
        if (split && hit) HIPCHK(ctx, hipStreamWaitEvent(ctx->half_stream, ctx->gather_event, 0));
        if (ctx->gather_stream == s || hit) ctx->gather_lo = ctx->gather_hi = nullptr; // (ordered now, by the stream or by the event)
    }

    BatchIO io;
    io.d_points = b->d_points;
    io.cloud_stride = b->cloud_stride;
    io.point_format = b->point_format;
    io.d_labels = b->d_labels;
    io.d_out_index = b->d_out_index;
    io.d_out_clouds = b->d_out_clouds;
    io.d_out_counts = b->d_out_counts;
    io.d_label_masks = b->d_label_masks;
    io.d_out_pc2 = b->d_out_pc2;

    Arena a = ctx->arena;
    a.flags = eff_flags;
    a.eigen_reduction = ctx->conv.eigen_reduction;
    a.fresh_launch = 0;
    if (split) {
        // the side stream sees what the caller's stream holds up to here (its inputs, a re-initialisation of the maps on that stream)
        if (!ctx->probe_no_fork) {
            HIPCHK(ctx, hipEventRecord(ctx->half_fork, s));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->half_stream, ctx->half_fork, 0));
        }
        Arena a2 = a; // the words kernels of one launch synchronise through: the second half has its own
        a2.front_sync = a.front_sync2;
        a2.sweep_sync = a.sweep_sync2;
        a2.sweep_rec_clouds = 0; // (one scratch region: the side stream's half keeps k_sweep)
        a2.scan_sync = a.scan_sync2; // (its own region: the halves of consecutive divided batches need not be the same size)
        a.fresh_launch = fresh_half[0] ? 1 : 0;
        a2.fresh_launch = fresh_half[1] ? 1 : 0;
        if (const int rc = fill_fresh(0, n_first, s)) return rc;
        if (const int rc = fill_fresh(n_first, nb, ctx->half_stream)) return rc;
        HIPCHK(ctx, hipMemcpyAsync(dp, hp, sizeof(CloudParams) * n_first, hipMemcpyHostToDevice, s));
        HIPCHK(ctx, hipMemcpyAsync(dp + n_first, hp + n_first, sizeof(CloudParams) * (nb - n_first), hipMemcpyHostToDevice, ctx->half_stream));
        launch_sequence(ctx, a, dp, io, n_first, max_n[0], s, nullptr, hp[0].slot, false);
        launch_sequence(ctx, a2, dp + n_first, io, nb - n_first, max_n[1], ctx->half_stream, nullptr, hp[n_first].slot, false);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(ctx->ring_done2[g], ctx->half_stream));
        ctx->ring_used2[g] = true;
        HIPCHK(ctx, hipEventRecord(ctx->half_done, ctx->half_stream));
        ctx->have_half_event = true;
    } else {
        a.fresh_launch = fresh_half[0] ? 1 : 0;
        if (const int rc = fill_fresh(0, nb, s)) return rc;
        if (const int rc = launch_or_replay(ctx, a, hp, dp, io, nb, max_n[0], s, plan)) return rc;
        HIPCHK(ctx, hipGetLastError());
    }
    HIPCHK(ctx, hipEventRecord(ctx->ring_done[g], s));
    ctx->ring_used[g] = true;
    HIPCHK(ctx, hipEventRecord(ctx->batch_event, s));
    ctx->last_batch_stream = s;
    ctx->have_batch_event = true;
    if (s != ctx->stream) ctx->map_event_pending = false; // (this stream has waited; later batches anywhere follow batch_event)
    return GG_OK;
}

bool slot_ok(const gg_context *ctx, int slot) { return ctx && slot >= 0 && slot < ctx->n_slots; }

// Before anything reads (or densifies) one of the three layers GG_FLAG_MINIMAL_LAYERS leaves out: compute them for this slot, once,
// on the context's stream (the callers have ordered it behind the batches).  SURVEY Appendix E: the published-only layers are
// materialised when somebody asks, from the retained tile-sorted records; gg_get_layer returns at all times what the reference holds.
int ensure_lazy_layers(gg_context *ctx, int slot, bool wanted)
{
    if (!wanted || !ctx->lazy_pending[slot]) return GG_OK;
    launch_reduce_lazy(ctx->arena, ctx->lazy_params[slot], ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    ctx->lazy_pending[slot] = 0;
    return GG_OK;
}
bool lazy_layer(int layer) { return layer == GG_LAYER_MAXGROUNDHEIGHT || layer == GG_LAYER_GROUNDCANDIDATES || layer == GG_LAYER_PLANEDIST; }

} // namespace

static int rebuild_patch_table(gg_context *ctx, const DevConfig &cfg);

extern "C" {

int gg_abi_version(void) { return GG_ABI_VERSION; }

const char *gg_kernel_name(int k)
{
    static const char *names[GG_NUM_KERNELS] = {"k_classify", "k_scan", "k_scatter", "k_reduce", "k_patch", "k_sweep", "k_label"};
    return (k >= 0 && k < GG_NUM_KERNELS) ? names[k] : "?";
}

void gg_default_config(gg_config *c)
{
    if (!c) return;
    // cfg/GroundGrid.cfg:8-21
    c->point_count_cell_variance_threshold = 10;
    c->max_ring = 1024;
    c->groundpatch_detection_minimum_threshold = 0.01;
    c->distance_factor = 0.0001;
    c->minimum_distance_factor = 0.0005;
    c->miminum_point_height_threshold = 0.3;
    c->minimum_point_height_obstacle_threshold = 0.1;
    c->outlier_tolerance = 0.1;
    c->ground_patch_detection_minimum_point_count_threshold = 0.25;
    c->patch_size_change_distance = 20;
    c->occupied_cells_decrease_factor = 5.0;
    c->occupied_cells_point_count_factor = 20;
    c->min_outlier_detection_ground_confidence = 1.25;
    c->thread_count = 8;
}

void gg_default_geometry(gg_geometry *g)
{
    if (!g) return;
    g->length = 120.0f;                                      // GroundGrid.h:71
    g->resolution = .33f;                                    // GroundGrid.h:70
    g->vertical_point_ang_dist = (float)(0.00174532925 * 2); // GroundSegmentation.h:69
    g->min_dist_squared = 12.0f;                             // GroundSegmentation.h:70
}

int gg_create(const gg_geometry *geom_in, int n_slots, size_t max_points, int device, gg_context **out)
{
    if (!out) return GG_ERR_INVALID;
    *out = nullptr;
    if (n_slots <= 0 || max_points == 0 || max_points > (size_t)1 << 30) return GG_ERR_INVALID;
    gg_geometry geom;
    gg_default_geometry(&geom);
    if (geom_in) {
        if (geom_in->length > 0.f) geom.length = geom_in->length;
        if (geom_in->resolution > 0.f) geom.resolution = geom_in->resolution;
        if (geom_in->vertical_point_ang_dist > 0.f) geom.vertical_point_ang_dist = geom_in->vertical_point_ang_dist;
        if (geom_in->min_dist_squared > 0.f) geom.min_dist_squared = geom_in->min_dist_squared;
    }

    // grid_map::GridMap::setGeometry (called at src/GroundGrid.cpp:58)
    const double res = (double)geom.resolution;
    const int n = (int)round((double)geom.length / res);
    // GroundSegmentation::init (src/GroundSegmentation.cpp:38), dimension passed as size_t (Nodelet.cpp:95)
    const size_t dimension = (size_t)geom.length;
    const size_t cellCount = (size_t)roundf((float)dimension / geom.resolution);
    if (n < 8 || (size_t)n != cellCount) return GG_ERR_GEOMETRY;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return GG_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return GG_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return GG_ERR_NO_DEVICE; // this library carries gfx950 code only

    gg_context *ctx = new (std::nothrow) gg_context();
    if (!ctx) return GG_ERR_NOMEM;
    ctx->device = device;
    ctx->geom = geom;
    ctx->n_slots = n_slots;
    ctx->max_points = max_points;
    gg_default_config(&ctx->cfg);
    ctx->pos_x.assign(n_slots, 0.0);
    ctx->no_confidence.assign(n_slots, 0); // layers start as zeros, but only gg_reset_map makes a slot usable
    ctx->fresh.assign(n_slots, 0);
    ctx->fresh_z.assign(n_slots, 0.0f);
    ctx->fresh_enabled = !(getenv("GG_FRESH_MAPS") && atoi(getenv("GG_FRESH_MAPS")) == 0);
    ctx->lazy_pending.assign(n_slots, 0);
    ctx->lazy_params.assign(n_slots, CloudParams{});
    ctx->pos_y.assign(n_slots, 0.0);

#define CREATE_CHK(call)                                             \
    do {                                                             \
        hipError_t e__ = (call);                                     \
        if (e__ != hipSuccess) {                                     \
            fprintf(stderr, "groundgrid_hip: %s failed: %s\n", #call, hipGetErrorString(e__)); \
            gg_destroy(ctx);                                         \
            return e__ == hipErrorOutOfMemory ? GG_ERR_NOMEM : GG_ERR_HIP; \
        }                                                            \
    } while (0)

    CREATE_CHK(hipSetDevice(device));
    CREATE_CHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CREATE_CHK(hipStreamCreateWithFlags(&ctx->h2d_stream, hipStreamNonBlocking));
    CREATE_CHK(hipStreamCreateWithFlags(&ctx->d2h_stream, hipStreamNonBlocking));
    CREATE_CHK(hipStreamCreateWithFlags(&ctx->half_stream, hipStreamNonBlocking));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->half_fork, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->half_done, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->map_event, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->batch_event, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->gather_event, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->join_event, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&ctx->layers_event, hipEventDisableTiming));
    if (const char *e = getenv("GG_GRAPH")) ctx->graphs_enabled = atoi(e) != 0; // (1 = one cloud per call replays a captured graph)

    Arena &a = ctx->arena;
    Geometry &g = a.g;
    g.rows = g.cols = n;
    g.C = n * n;
    g.tiles_r = (n + TILE - 1) / TILE;
    g.tiles_c = (n + TILE - 1) / TILE;
    g.T = g.tiles_r * g.tiles_c;
    g.resolution = res;
    g.inv_resolution = 1.0 / res;
    g.length0 = g.length1 = (double)n * res;
    g.half0 = 0.5 * g.length0;
    g.half1 = 0.5 * g.length1;
    g.resolution_f = (float)res;
    g.min_dist_squared = geom.min_dist_squared;
    g.center = n / 2 - 1;
    a.hist_pitch = (g.T + 3) & ~3;
    a.n_slots = n_slots;
    if (g.T > 65535 || (size_t)4 * a.hist_pitch * sizeof(uint32_t) > 150 * 1024) {
        gg_destroy(ctx);
        return GG_ERR_GEOMETRY; // per-wave LDS tile histograms no longer fit
    }
    // points per wavefront chunk of K1 / scatter / K5: the chunk histograms (K1 -> k_scan -> k_scatter) shrink with it, the
    // number of wavefronts per cloud too -- contexts for big batches take 2048 (k_scan 0.18 -> 0.11 ms per 1024 clouds), small
    // ones 1024 (one cloud: 0.56 instead of 0.60 ms)
    a.PW = g.T <= 1024 ? (n_slots >= PW_BIG_CONTEXT_SLOTS ? 2048 : 1024) : 8192;
    if (getenv("GG_PW")) a.PW = atoi(getenv("GG_PW")); // (tools, tests: points per wavefront chunk of K1 / scatter / K5)
    if (a.PW < 64 || a.PW % 64 != 0) {
        gg_destroy(ctx);
        return GG_ERR_INVALID;
    }
    a.tune_sweep_waves = getenv("GG_SWEEP_WAVES") ? atoi(getenv("GG_SWEEP_WAVES")) : 0;
    a.tune_sweep_gpw = getenv("GG_SWEEP_GPW") ? atoi(getenv("GG_SWEEP_GPW")) : 0;
    a.tune_sweep_split = getenv("GG_SWEEP_SPLIT") ? atoi(getenv("GG_SWEEP_SPLIT")) : 0;
    a.tune_sweep_pair = getenv("GG_SWEEP_PAIR") ? atoi(getenv("GG_SWEEP_PAIR")) : 0;
    a.tune_sweep_pair_wgs = getenv("GG_SWEEP_PAIR_WGS") ? atoi(getenv("GG_SWEEP_PAIR_WGS")) : 0;
    a.tune_sweep_pair_waves = getenv("GG_SWEEP_PAIR_WAVES") ? atoi(getenv("GG_SWEEP_PAIR_WAVES")) : 0;
    a.tune_front = getenv("GG_FRONT") ? atoi(getenv("GG_FRONT")) : 0;
    a.tune_k2_per_cloud = getenv("GG_K2_PER_CLOUD") ? atoi(getenv("GG_K2_PER_CLOUD")) : 0;
    a.tune_k2_dense_share = getenv("GG_K2_DENSE_SHARE") ? atoi(getenv("GG_K2_DENSE_SHARE")) : 0;
    a.k2_skip = getenv("GG_K2_SKIP") ? atoi(getenv("GG_K2_SKIP")) : 0;
    a.NCH = (int)((max_points + a.PW - 1) / a.PW);
    make_dev_config(ctx->cfg, a.cfg);

    // R1: expectedPoints (src/GroundSegmentation.cpp:40-46) -- host libm atanf, as in the reference
    ctx->h_expected.resize((size_t)g.C);
    for (size_t i = 0; i < cellCount; ++i)
        for (size_t j = 0; j < cellCount; ++j) {
            const float dist = (float)hypot((double)i - (double)cellCount / 2.0, (double)j - (double)cellCount / 2.0);
            ctx->h_expected[i + j * cellCount] = atanf(1 / dist) / geom.vertical_point_ang_dist;
        }

    ctx->sweep_params = gg::sweep::make_params(n, res, geom.min_dist_squared, ctx->cfg.occupied_cells_decrease_factor);
    if (gg::sweep_lds_bytes(ctx->sweep_params) > 158 * 1024) {
        gg_destroy(ctx);
        return GG_ERR_GEOMETRY; // the sweep's hand-over tables no longer fit in LDS even with one ring group per work-group (n > ~1500)
    }
    std::vector<uint16_t> tile_rank(g.T), rank_tile(g.T);
    {
        std::vector<std::pair<uint32_t, int>> order(g.T);
        for (int tc = 0; tc < g.tiles_c; ++tc)
            for (int tr = 0; tr < g.tiles_r; ++tr) order[tr + tc * g.tiles_r] = {morton2((uint32_t)tr, (uint32_t)tc), tr + tc * g.tiles_r};
        std::sort(order.begin(), order.end());
        for (int r = 0; r < g.T; ++r) {
            rank_tile[r] = (uint16_t)order[r].second;
            tile_rank[order[r].second] = (uint16_t)r;
        }
    }

    // ---- carve the arena -----------------------------------------------------------------
    const size_t A = 256;
    size_t off = 0;
    auto carve = [&](size_t bytes) {
        const size_t o = off;
        off = align_up(off + bytes, A);
        return o;
    };
    const size_t C = (size_t)g.C;
    const size_t Cpad = align_up(C * 4, A) / 4;
    const size_t Npad = align_up(max_points * 8, A) / 8;
    const size_t o_expected = carve(C * 4);
    const size_t o_ptable = carve(C * 16);
    const size_t o_trank = carve((size_t)g.T * 2);
    const size_t o_rtile = carve((size_t)g.T * 2);
    const size_t o_rcell0 = carve((size_t)g.T * 4);
    const size_t percall_slot_floats = align_up((size_t)g.T * PERCALL_BLOCK * 4, A) / 4;
    const size_t o_layers = carve((size_t)n_slots * percall_slot_floats * 4);
    a.gpl = make_gp_layout(n);
    // (a slot's written-cell bits -- FRESH maps, gg_internal.h -- live behind its layer: one buffer descriptor reaches both)
    a.gp_bits_off = (int)align_up((size_t)a.gpl.elems, 16);
    a.gp_bits_words = (a.gpl.elems / 64 + 2 + 1) & ~1; // (8-byte words; even: k_reset_fresh copies 16 bytes at a time)
    a.gp2_stride = align_up(((size_t)a.gp_bits_off + (size_t)a.gp_bits_words) * 8, A) / 8;
    a.gp_bits_stride = a.gp2_stride;
    const size_t o_gp2 = carve((size_t)n_slots * a.gp2_stride * 8);
    const size_t o_rec = carve((size_t)n_slots * Npad * 8);
    const size_t o_sorted = carve((size_t)n_slots * Npad * 8);
    a.zcell_stride = align_up((Npad + (size_t)32 * g.T + 64) * 4, A) / 4;
    const size_t o_zcell = carve((size_t)n_slots * a.zcell_stride * 4);
    a.hist_stride = align_up((size_t)a.NCH * a.hist_pitch * 4, A) / 4;
    const size_t o_hist = carve((size_t)n_slots * a.hist_stride * 4);
    a.emit_stride = align_up((size_t)a.NCH * 4 * 4, A) / 4;
    const size_t o_emit = carve((size_t)n_slots * a.emit_stride * 4);
    const size_t o_totals = carve((size_t)n_slots * 4 * 4);
    a.tile_start_stride = align_up((size_t)(g.T + 1) * 4, A) / 4;
    const size_t o_tstart = carve((size_t)n_slots * a.tile_start_stride * 4);
    a.tile_live_stride = align_up((size_t)g.T, A);
    const size_t o_tlive = carve((size_t)n_slots * a.tile_live_stride * 4);
    a.tile_list_stride = align_up((size_t)g.T * 16, A) / 16;
    const size_t o_tlist = carve((size_t)n_slots * a.tile_list_stride * 16);
    const size_t o_tlcnt = carve((size_t)n_slots * 2 * 4);
    const size_t o_fsync = carve(((size_t)2 * n_slots + 16) * 4);
    const size_t o_ssync = carve(64);
    const size_t o_fsync2 = carve(((size_t)2 * n_slots + 16) * 4);
    const size_t o_ssync2 = carve(64);
    const size_t o_scansync = carve((size_t)n_slots * SCAN_SYNC_WORDS * 8);
    const size_t o_scansync2 = carve((size_t)n_slots * SCAN_SYNC_WORDS * 8);
    a.sweep_xchg_stride = align_up(std::max<size_t>(gg::sweep_xchg_entries(ctx->sweep_params), 1) * 16, A) / 8;
    const size_t o_xchg = carve((size_t)n_slots * a.sweep_xchg_stride * 8);
    a.sweep_rec_stride = gg::sweep_pair_rec_floats(ctx->sweep_params);
    a.sweep_rec_clouds = a.sweep_rec_stride ? std::min(n_slots, SWEEP_PAIR_MAX_CLOUDS) : 0;
    const size_t o_srec = carve(std::max<size_t>((size_t)a.sweep_rec_clouds * a.sweep_rec_stride * 4, 64));
    const size_t o_params = carve((size_t)PARAM_RING * n_slots * sizeof(CloudParams));
    const size_t o_gparams = carve(sizeof(CloudParams));
    const size_t o_spts = carve(max_points * sizeof(gg_point16));
    const size_t o_slab = carve(max_points);
    const size_t o_sidx = carve(max_points * 4);
    const size_t o_scnt = carve(64);
    const size_t o_scls = carve(max_points);
    const size_t o_scell = carve(max_points * 4);
    size_t o_apts[GG_ASYNC_DEPTH], o_alab[GG_ASYNC_DEPTH], o_aidx[GG_ASYNC_DEPTH], o_acnt[GG_ASYNC_DEPTH];
    for (int k = 0; k < GG_ASYNC_DEPTH; ++k) {
        o_apts[k] = carve(max_points * sizeof(gg_point16));
        // results of one ticket as ONE block -- counts (64 B), then the index (4 n B), then the labels (n B) -- so that they come back
        // with a single copy; where index and labels start inside it depends on the ticket's n
        o_acnt[k] = carve(64 + max_points * 5 + 64);
        o_aidx[k] = o_alab[k] = 0;
    }
    const size_t o_scroll = carve(a.gp2_stride * 8); // one layer in its device element order (map scroll) / two planes (images)
    const size_t o_image = carve(3 * Cpad * 4);
    const size_t o_bounds = carve(64);
    const size_t gp_valid_words = ((size_t)a.gpl.elems + 31) / 32;
    const size_t o_gpvalid = carve(gp_valid_words * 4);
    std::vector<int> gp_border; // the cells no sweep visits: ring >= c
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row)
            if (std::max(std::abs(row - a.gpl.c), std::abs(col - a.gpl.c)) >= a.gpl.c) gp_border.push_back(gp_index(a.gpl, row, col));
    a.gp_border_n = (int)gp_border.size();
    const size_t o_gpborder = carve(std::max<size_t>(gp_border.size() * 4, 64));
    a.gp_fresh_cell = 1 + (a.gpl.VS - 1) * 64; // side 0, group 0, the last sheared position of ring 1: beyond the map's last column
    const size_t o_dbg = carve(64 * 8); // sweep timing
    const size_t o_pdbg = carve(2048 * 8); // pair sweep timing
    const bool k2_timing = getenv("GG_K2_DEBUG") && (atoi(getenv("GG_K2_DEBUG")) == 9 || atoi(getenv("GG_K2_DEBUG")) == 5 || atoi(getenv("GG_K2_DEBUG")) == 6);
    const size_t o_k2dbg = carve(k2_timing ? (size_t)K2_DBG_WGS * 32 * 8 : 64);
    ctx->arena_bytes = off;
    CREATE_CHK(hipMalloc(&ctx->d_arena, ctx->arena_bytes));
    char *base = (char *)ctx->d_arena;
    CREATE_CHK(hipMemsetAsync(base, 0, ctx->arena_bytes, ctx->stream));

    a.expected = (const float *)(base + o_expected);
    a.patch_table = (const float4 *)(base + o_ptable);
    a.tile_rank = (const uint16_t *)(base + o_trank);
    a.rank_tile = (const uint16_t *)(base + o_rtile);
    a.rank_cell0 = (const uint32_t *)(base + o_rcell0);
    a.layers = (float *)(base + o_layers);
    a.gp2 = (float2 *)(base + o_gp2);
    a.gp_bits = (unsigned long long *)(base + o_gp2) + a.gp_bits_off;
    a.slot_layer_stride = percall_slot_floats;
    a.rec = (uint2 *)(base + o_rec);
    a.sorted = (uint2 *)(base + o_sorted);
    a.zcell = (float *)(base + o_zcell);
    a.point_stride = Npad;
    a.hist = (uint32_t *)(base + o_hist);
    a.chunk_emit = (uint32_t *)(base + o_emit);
    a.totals = (uint32_t *)(base + o_totals);
    a.tile_start = (uint32_t *)(base + o_tstart);
    a.tile_live = (uint32_t *)(base + o_tlive);
    a.tile_list = (uint4 *)(base + o_tlist);
    a.tile_list_cnt = (uint32_t *)(base + o_tlcnt);
    a.front_sync = (uint32_t *)(base + o_fsync);
    a.sweep_sync = (uint32_t *)(base + o_ssync);
    a.front_sync2 = (uint32_t *)(base + o_fsync2);
    a.sweep_sync2 = (uint32_t *)(base + o_ssync2);
    a.scan_sync = (unsigned long long *)(base + o_scansync);
    a.scan_sync2 = (unsigned long long *)(base + o_scansync2);
    {
        const uint32_t first_epoch[4] = {0u, 0u, 1u, 0u}; // (the exchange region starts zeroed: tag 0 is never current)
        CREATE_CHK(hipMemcpyAsync(a.sweep_sync, first_epoch, sizeof first_epoch, hipMemcpyHostToDevice, ctx->stream));
        const uint32_t first_epoch2[4] = {0u, 0u, 0x80000001u, 0u}; // (the second set's epochs carry bit 31: k4_sweep.hip)
        CREATE_CHK(hipMemcpyAsync(a.sweep_sync2, first_epoch2, sizeof first_epoch2, hipMemcpyHostToDevice, ctx->stream));
        CREATE_CHK(hipStreamSynchronize(ctx->stream));
    }
    {
        // one word the kernels can reach and the host can read without a copy: a bounded wait that ran out reports here
        CREATE_CHK(hipHostMalloc((void **)&ctx->h_dev_error, 64, hipHostMallocMapped));
        *ctx->h_dev_error = 0u;
        void *dptr = nullptr;
        CREATE_CHK(hipHostGetDevicePointer(&dptr, (void *)ctx->h_dev_error, 0));
        a.dev_error = (uint32_t *)dptr;
    }
    a.sweep_xchg = (unsigned long long *)(base + o_xchg);
    a.sweep_rec = a.sweep_rec_clouds ? (float *)(base + o_srec) : nullptr;
    a.pair_dbg = getenv("GG_PAIR_TIMING") ? (unsigned long long *)(base + o_pdbg) : nullptr;
    a.flags = 0;
    a.k2_debug = getenv("GG_K2_DEBUG") ? atoi(getenv("GG_K2_DEBUG")) : 0;
    a.k2_dbg = (unsigned long long *)(base + o_k2dbg);
    a.k3_debug = getenv("GG_K3_DEBUG") ? atoi(getenv("GG_K3_DEBUG")) : 0;
    a.k5_debug = getenv("GG_K5_DEBUG") ? atoi(getenv("GG_K5_DEBUG")) : 0;
    ctx->d_params = (CloudParams *)(base + o_params);
    ctx->d_gparams = (CloudParams *)(base + o_gparams);
    ctx->d_stage_pts = (gg_point16 *)(base + o_spts);
    ctx->d_stage_labels = (uint8_t *)(base + o_slab);
    ctx->d_stage_index = (int32_t *)(base + o_sidx);
    ctx->d_stage_counts = (int32_t *)(base + o_scnt);
    ctx->d_stage_class = (uint8_t *)(base + o_scls);
    ctx->d_stage_cell = (int32_t *)(base + o_scell);
    for (int k = 0; k < GG_ASYNC_DEPTH; ++k) {
        gg_context::AsyncSlot &as = ctx->async_slot[k];
        as.d_pts = (gg_point16 *)(base + o_apts[k]);
        as.d_counts = (int32_t *)(base + o_acnt[k]);
        as.d_index = (int32_t *)(base + o_acnt[k] + 64);
        as.d_labels = nullptr; // (placed per ticket: behind the n index entries)
    }
    ctx->d_scroll_scratch = (float *)(base + o_scroll);
    ctx->d_image = (float *)(base + o_image);
    ctx->d_bounds = (float *)(base + o_bounds);
    a.gp_valid = (const uint32_t *)(base + o_gpvalid);
    a.gp_border = (const int *)(base + o_gpborder);
    {
        int row, col;
        if (gp_cell_of(a.gpl, a.gp_fresh_cell, row, col)) { // (cannot happen: sheared position VS - 1 of ring 1 is column n + 125)
            gg_destroy(ctx);
            return GG_ERR_INVALID;
        }
        CREATE_CHK(hipMemcpyAsync(base + o_gpborder, gp_border.data(), gp_border.size() * 4, hipMemcpyHostToDevice, ctx->stream));
        CREATE_CHK(hipStreamSynchronize(ctx->stream)); // (the vectors go out of scope)
    }
    if (getenv("GG_SWEEP_TIMING")) {
        ctx->d_sweep_dbg = (unsigned long long *)(base + o_dbg);
        const unsigned long long mode = getenv("GG_SWEEP_DEBUG") ? strtoull(getenv("GG_SWEEP_DEBUG"), nullptr, 0) : 0ull;
        hipStreamSynchronize(ctx->stream); // (the arena memset above)
        hipMemcpy(ctx->d_sweep_dbg + 63, &mode, 8, hipMemcpyHostToDevice);
    }

    std::vector<uint32_t> gp_valid(gp_valid_words, 0u);
    for (int col = 0; col < n; ++col)
        for (int row = 0; row < n; ++row) {
            const int e = gp_index(a.gpl, row, col);
            gp_valid[(size_t)e >> 5] |= 1u << (e & 31);
        }
    {
        // The mask only steers the fill of gg_reset_maps, and the padding is never read: every 128-byte line (16 elements) that
        // holds a cell is filled whole.  More than half of the lines with cells are only partly cells (the shear), and a partly
        // written line costs the memory a read-modify-write: 1024 maps take 0.60 ms with the exact mask (1.09 GB), 0.41 ms with
        // whole 64-byte sectors, 0.36 ms with whole lines (1.52 GB, 4.3 TB/s).
        const int group = getenv("GG_FILL_GROUP") ? atoi(getenv("GG_FILL_GROUP")) : 16; // elements; 1 = exact mask (measurement)
        if (group == 8 || group == 16 || group == 32)
            for (size_t w = 0; w < gp_valid_words; ++w) {
                uint32_t v = gp_valid[w], o = 0u;
                for (int b = 0; b < 32; b += group) {
                    const uint32_t gm = (group == 32 ? 0xFFFFFFFFu : ((1u << group) - 1u)) << b;
                    if (v & gm) o |= gm;
                }
                gp_valid[w] = o;
            }
    }
    CREATE_CHK(hipMemcpyAsync(base + o_gpvalid, gp_valid.data(), gp_valid_words * 4, hipMemcpyHostToDevice, ctx->stream));
    CREATE_CHK(hipMemcpyAsync(base + o_expected, ctx->h_expected.data(), C * 4, hipMemcpyHostToDevice, ctx->stream));
    CREATE_CHK(hipMemcpyAsync(base + o_trank, tile_rank.data(), (size_t)g.T * 2, hipMemcpyHostToDevice, ctx->stream));
    CREATE_CHK(hipMemcpyAsync(base + o_rtile, rank_tile.data(), (size_t)g.T * 2, hipMemcpyHostToDevice, ctx->stream));
    std::vector<uint32_t> rank_cell0(g.T);
    for (int r = 0; r < g.T; ++r) {
        const int tile = rank_tile[r];
        rank_cell0[r] = (uint32_t)((tile % g.tiles_r) * TILE) | ((uint32_t)((tile / g.tiles_r) * TILE) << 16);
    }
    CREATE_CHK(hipMemcpyAsync(base + o_rcell0, rank_cell0.data(), (size_t)g.T * 4, hipMemcpyHostToDevice, ctx->stream));
    CREATE_CHK(hipStreamSynchronize(ctx->stream)); // the host vectors above go out of scope

    CREATE_CHK(hipHostMalloc((void **)&ctx->h_params, sizeof(CloudParams) * PARAM_RING * n_slots, hipHostMallocDefault));
    CREATE_CHK(hipHostMalloc((void **)&ctx->h_stage_pts, max_points * sizeof(gg_point16), hipHostMallocDefault));
    CREATE_CHK(hipHostMalloc((void **)&ctx->h_stage_labels, max_points, hipHostMallocDefault));
    CREATE_CHK(hipHostMalloc((void **)&ctx->h_stage_index, max_points * 4, hipHostMallocDefault));
    CREATE_CHK(hipHostMalloc((void **)&ctx->h_stage_counts, 64, hipHostMallocDefault));
    for (int i = 0; i < PARAM_RING; ++i) CREATE_CHK(hipEventCreateWithFlags(&ctx->ring_done[i], hipEventDisableTiming));
    for (int i = 0; i < PARAM_RING; ++i) CREATE_CHK(hipEventCreateWithFlags(&ctx->ring_done2[i], hipEventDisableTiming));
    for (int k = 0; k < GG_ASYNC_DEPTH; ++k) {
        gg_context::AsyncSlot &as = ctx->async_slot[k];
        CREATE_CHK(hipHostMalloc((void **)&as.h_pts, max_points * sizeof(gg_point16), hipHostMallocDefault));
        CREATE_CHK(hipHostMalloc((void **)&as.h_counts, 64 + max_points * 5 + 64, hipHostMallocDefault)); // (the same block on the host)
        as.h_index = as.h_counts + 16;
        as.h_labels = nullptr;
        if (hipHostGetDevicePointer((void **)&as.hd_counts, as.h_counts, 0) != hipSuccess) { // (no device view of pinned memory: the results take the copy)
            as.hd_counts = nullptr;
            (void)hipGetLastError();
        }
        CREATE_CHK(hipEventCreateWithFlags(&as.uploaded, hipEventDisableTiming));
        CREATE_CHK(hipEventCreateWithFlags(&as.computed, hipEventDisableTiming));
        CREATE_CHK(hipEventCreateWithFlags(&as.downloaded, hipEventDisableTiming));
    }

    ctx->helper.configure((getenv("GG_HOST_THREADS") ? std::max(1, std::min(atoi(getenv("GG_HOST_THREADS")), 16)) : 8) - 1); // (never more than the usable CPUs - 1: host_helper.h)
    {
        const int rc = rebuild_patch_table(ctx, a.cfg);
        if (rc != GG_OK) {
            gg_destroy(ctx);
            return rc;
        }
    }
    {
        const int rc = gg_reset_maps(ctx, 0, n_slots, 0.0, 0.0, 0.0f, 0, nullptr); // (one strided fill per layer for all slots)
        if (rc != GG_OK) {
            gg_destroy(ctx);
            return rc;
        }
    }
    CREATE_CHK(hipStreamSynchronize(ctx->stream));
#undef CREATE_CHK
    *out = ctx;
    return GG_OK;
}

void gg_destroy(gg_context *ctx)
{
    if (!ctx) return;
    ctx->helper.stop();
    hipSetDevice(ctx->device);
    if (ctx->have_batch_event) hipEventSynchronize(ctx->batch_event);
    if (ctx->half_stream) hipStreamSynchronize(ctx->half_stream);
    if (ctx->h2d_stream) hipStreamSynchronize(ctx->h2d_stream);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->d2h_stream) hipStreamSynchronize(ctx->d2h_stream);
    for (auto &p : ctx->pending) {
        if (p.owns_start) hipEventDestroy(p.start);
        hipEventDestroy(p.stop);
    }
    for (hipEvent_t e : ctx->free_single_events) hipEventDestroy(e);
    for (int i = 0; i < PARAM_RING; ++i)
        if (ctx->ring_done[i]) hipEventDestroy(ctx->ring_done[i]);
    for (int i = 0; i < PARAM_RING; ++i)
        if (ctx->ring_done2[i]) hipEventDestroy(ctx->ring_done2[i]);
    if (ctx->half_fork) hipEventDestroy(ctx->half_fork);
    if (ctx->half_done) hipEventDestroy(ctx->half_done);
    if (ctx->half_stream) hipStreamDestroy(ctx->half_stream);
    for (int k = 0; k < GG_ASYNC_DEPTH; ++k) {
        gg_context::AsyncSlot &as = ctx->async_slot[k];
        if (as.h_pts) hipHostFree(as.h_pts);
        if (as.h_counts) hipHostFree(as.h_counts); // (index and labels live in the same block)
        if (as.uploaded) hipEventDestroy(as.uploaded);
        if (as.computed) hipEventDestroy(as.computed);
        if (as.downloaded) hipEventDestroy(as.downloaded);
    }
    if (ctx->map_event) hipEventDestroy(ctx->map_event);
    if (ctx->batch_event) hipEventDestroy(ctx->batch_event);
    if (ctx->gather_event) hipEventDestroy(ctx->gather_event);
    drop_graphs(ctx);
    if (ctx->fork_event) hipEventDestroy(ctx->fork_event);
    if (ctx->join_event) hipEventDestroy(ctx->join_event);
    if (ctx->layers_event) hipEventDestroy(ctx->layers_event);
    for (auto &r : ctx->registered) hipHostUnregister(r.first);
    ctx->registered.clear();
    if (ctx->h2d_stream) hipStreamDestroy(ctx->h2d_stream);
    if (ctx->d2h_stream) hipStreamDestroy(ctx->d2h_stream);
    if (ctx->h_dev_error) hipHostFree((void *)ctx->h_dev_error);
    if (ctx->h_params) hipHostFree(ctx->h_params);
    if (ctx->h_stage_pts) hipHostFree(ctx->h_stage_pts);
    if (ctx->h_stage_labels) hipHostFree(ctx->h_stage_labels);
    if (ctx->h_stage_index) hipHostFree(ctx->h_stage_index);
    if (ctx->h_stage_counts) hipHostFree(ctx->h_stage_counts);
    if (ctx->d_planes) hipFree(ctx->d_planes);
    if (ctx->d_pc2) hipFree(ctx->d_pc2);
    if (ctx->h_pc2) hipHostFree(ctx->h_pc2);
    if (ctx->h_planes) hipHostFree(ctx->h_planes);
    if (ctx->d_arena) hipFree(ctx->d_arena);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

// The per-cell constants of detect_ground_patches (gg_internal.h Arena::patch_table), in the reference's own arithmetic: every cell
// used to recompute them for every cloud -- a dozen binary64 operations and the sheared layer index per cell and call, a sixth of
// k_patch's instructions -- although they depend on the cell and the configuration only.
// `cfg`: the configuration the table is for -- the caller commits it to the context only when the table is in place, so that k_patch's
// thresholds and the rest of the path never come from two configurations.
static int rebuild_patch_table(gg_context *ctx, const DevConfig &cfg)
{
    const Geometry &g = ctx->arena.g;
    const int rows = g.rows, cols = g.cols;
    std::vector<float> t((size_t)g.C * 4);
    const double res2 = (double)g.resolution_f * (double)g.resolution_f;
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i) {
            float *e = &t[((size_t)i + (size_t)j * rows) * 4];
            const float expected = ctx->h_expected[(size_t)i + (size_t)j * rows];
            // the four quadrants (:325-328) cover rows [2, 2 * (cols / 2) - 2) and cols [2, rows - 2)
            const bool visited = i >= 2 && j >= 2 && !(i >= 2 * (cols / 2) - 2 || j >= rows - 2);
            const double di = (double)i - (double)rows / 2.0, dj = (double)j - (double)cols / 2.0;
            const float sqdist = (float)((di * di + dj * dj) * res2);                          // :332
            const bool near = (double)sqdist <= cfg.patch_size_change_distance_sq;             // :334
            const double S = near ? 3.0 : 5.0;
            const double thr = std::max(floor(cfg.gpd_min_point_count_threshold * S * (double)expected), 3.0); // :364 (an integer-valued double)
            const float thr_f = (float)thr;
            e[0] = expected;
            // (a threshold too large for an exact float -- absurd configurations -- can never be met by a count below 2^24 either)
            e[1] = !visited || !(thr < 16777216.0) ? INFINITY : (near ? -thr_f : thr_f);
            e[2] = (float)std::min(std::max((double)sqdist * cfg.distance_factor_sq, cfg.minimum_distance_factor_sq), cfg.minimum_distance_factor_x10_sq); // :369
            const int gidx = gp_index(ctx->arena.gpl, i, j);
            memcpy(&e[3], &gidx, 4);
        }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (batches may be reading the old table: wait for them; gg_set_config is a rare, synchronous call)
    if (ctx->have_batch_event) HIPCHK(ctx, hipEventSynchronize(ctx->batch_event));
    if (ctx->have_half_event) HIPCHK(ctx, hipEventSynchronize(ctx->half_done));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    HIPCHK(ctx, hipMemcpy(const_cast<float4 *>(ctx->arena.patch_table), t.data(), t.size() * 4, hipMemcpyHostToDevice));
    return GG_OK;
}

int gg_set_config(gg_context *ctx, const gg_config *cfg)
{
    if (!ctx || !cfg) return GG_ERR_INVALID;
    // src/GroundSegmentation.cpp:468-471.  Unlike the reference's struct copy this call BLOCKS: it waits for the batches in flight
    // (they read the old table), rebuilds the per-cell table of detect_ground_patches on the host (O(cells)) and uploads it.
    DevConfig dev;
    make_dev_config(*cfg, dev);
    if (const int rc = rebuild_patch_table(ctx, dev)) return rc; // (on failure the context keeps its old configuration AND table)
    ctx->cfg = *cfg;
    ctx->arena.cfg = dev;
    drop_graphs(ctx); // (captured launches carry the old configuration by value)
    return GG_OK;
}

int gg_get_config(const gg_context *ctx, gg_config *cfg)
{
    if (!ctx || !cfg) return GG_ERR_INVALID;
    *cfg = ctx->cfg;
    return GG_OK;
}

int gg_set_flags(gg_context *ctx, unsigned flags)
{
    if (!ctx) return GG_ERR_INVALID;
    // (leaving GG_FLAG_MINIMAL_LAYERS needs no repair: the next cloud writes all nine layers in the half columns it marks live, and
    // every other column logically holds the reset values anyway -- gg_internal.h tile_live)
    if (flags != ctx->flags) drop_graphs(ctx);
    ctx->flags = flags;
    return GG_OK;
}

int gg_set_conventions(gg_context *ctx, const gg_conventions *conv)
{
    if (!ctx || !conv) return GG_ERR_INVALID;
    if (conv->eigen_reduction != GG_EIGEN_33 && conv->eigen_reduction != GG_EIGEN_34_SSE) return fail(ctx, GG_ERR_INVALID, "eigen_reduction");
    for (int r : conv->reserved)
        if (r != 0) return fail(ctx, GG_ERR_INVALID, "gg_conventions.reserved must be 0");
    ctx->conv = *conv;
    drop_graphs(ctx);
    return GG_OK;
}

int gg_get_conventions(const gg_context *ctx, gg_conventions *conv)
{
    if (!ctx || !conv) return GG_ERR_INVALID;
    *conv = ctx->conv;
    return GG_OK;
}

// host arithmetic only (this file is compiled with -ffp-contract=off: no fused multiply-adds, like the x86-64 reference)
int gg_rotation_from_quaternion(int convention, const double q[4], double R[9])
{
    if (!q || !R) return GG_ERR_INVALID;
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    if (convention == GG_ROT_TF2) { // tf2/LinearMath/Matrix3x3.h setRotation
        const double d = x * x + y * y + z * z + w * w;
        const double s = 2.0 / d;
        const double xs = x * s, ys = y * s, zs = z * s;
        const double wx = w * xs, wy = w * ys, wz = w * zs;
        const double xx = x * xs, xy = x * ys, xz = x * zs;
        const double yy = y * ys, yz = y * zs, zz = z * zs;
        R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
        R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
        R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
    } else if (convention == GG_ROT_KDL) { // orocos_kdl frames.cpp Rotation::Quaternion
        const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
        R[0] = w2 + x2 - y2 - z2;     R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
        R[3] = 2 * x * y + 2 * w * z; R[4] = w2 - x2 + y2 - z2;     R[5] = 2 * y * z - 2 * w * x;
        R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = w2 - x2 - y2 + z2;
    } else {
        return GG_ERR_INVALID;
    }
    return GG_OK;
}

int gg_transform_from_pose(int convention, const double pose7[7], double out12[12])
{
    if (!pose7 || !out12) return GG_ERR_INVALID;
    double R[9];
    const int rc = gg_rotation_from_quaternion(convention, pose7 + 3, R);
    if (rc != GG_OK) return rc;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out12[r * 4 + c] = R[r * 3 + c];
        out12[r * 4 + 3] = pose7[r];
    }
    return GG_OK;
}

int gg_get_size(const gg_context *ctx, int *rows, int *cols)
{
    if (!ctx) return GG_ERR_INVALID;
    if (rows) *rows = ctx->arena.g.rows;
    if (cols) *cols = ctx->arena.g.cols;
    return GG_OK;
}

int gg_get_geometry(const gg_context *ctx, double *resolution, double *length_x, double *length_y)
{
    if (!ctx) return GG_ERR_INVALID;
    if (resolution) *resolution = ctx->arena.g.resolution;
    if (length_x) *length_x = ctx->arena.g.length0;
    if (length_y) *length_y = ctx->arena.g.length1;
    return GG_OK;
}

const char *gg_last_error(const gg_context *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int gg_reset_maps(gg_context *ctx, int first_slot, int n, double pos_x, double pos_y, float odom_z, int persistent_only, void *stream)
{
    if (!ctx) return GG_ERR_INVALID;
    if (n < 0 || first_slot < 0 || first_slot + n > ctx->n_slots) return GG_ERR_CAPACITY;
    if (n == 0) return GG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipStream_t st = pick_stream(ctx, stream);
    for (int s = first_slot; s < first_slot + n; ++s) ctx->fresh[s] = 0; // (whatever they were: rewritten now)
    // GG_FLAG_CONCURRENT_HALVES: on a caller stream the maps of the upper half of the slots are re-initialised on the side stream, where
    // their batches run -- the caller's stream never has to wait for the second half of the batch before
    const int boundary = (ctx->n_slots + 1) / 2;
    const bool split = halves_enabled(ctx) && st != ctx->stream && st != ctx->half_stream && st != nullptr && first_slot < boundary && first_slot + n > boundary;
    if (st == ctx->stream) {
        if (const int rc = own_stream_waits_for_batches(ctx, false)) return rc;
    } else { // ordered like a batch on the caller's stream
        if (ctx->map_event_pending) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->map_event, 0));
        if (ctx->have_batch_event && ctx->last_batch_stream != st && !ctx->probe_unordered_streams) HIPCHK(ctx, hipStreamWaitEvent(st, ctx->batch_event, 0));
        if (!(split && ctx->last_batch_stream == st))
            if (const int rc = stream_waits_for_second_half(ctx, st)) return rc;
    }
    const Arena &a = ctx->arena;
    for (int s = first_slot; s < first_slot + n; ++s) {
        ctx->no_confidence[s] = 1; // groundpatch := 1e-7 everywhere (scrolling keeps that: exposed cells get 0)
        ctx->pos_x[s] = pos_x;
        ctx->pos_y[s] = pos_y;
    }
    // src/GroundGrid.cpp:71-75; the layers filter_cloud adds later (:61-75) start at 0.  The slots' regions are equally
    // spaced, so one strided fill per layer covers all n slots.
    const float init[GG_NUM_LAYERS] = {0.0f, odom_z, (float)0.0000001, (float)100.0, (float)-100.0, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto fill = [&](int first, int count, hipStream_t on) {
        if (count <= 0) return;
        if (!persistent_only) {
            for (int s = first; s < first + count; ++s) ctx->lazy_pending[s] = 0;
            launch_fill_percall(a, first, count, init, on);
            // these are GroundGrid's initial values, not filter_cloud's per-call reset values: the next cloud rewrites every tile
            launch_fill_bytes((uint8_t *)(a.tile_live + (size_t)first * a.tile_live_stride), (size_t)count * a.tile_live_stride * 4, 0xFF, on);
        }
        if (ctx->fresh_enabled) { // the interior stays unwritten: the next batch's sweep rewrites it anyway (make_real() for everybody else)
            static const bool probe_all = getenv("GG_FRESH_MAPS") && atoi(getenv("GG_FRESH_MAPS")) == 2; // (measurement: every cell written AND marked)
            if (probe_all) launch_fill2_strided(gp2_ptr(a, first), (size_t)a.gpl.elems, a.gp2_stride, count, init[GG_LAYER_GROUND], init[GG_LAYER_GROUNDPATCH], a.gp_valid, on);
            launch_reset_fresh(a, first, count, init[GG_LAYER_GROUND], init[GG_LAYER_GROUNDPATCH], on, probe_all ? 1 : 0);
            for (int s = first; s < first + count; ++s) ctx->fresh[s] = 1, ctx->fresh_z[s] = odom_z;
        } else
            launch_fill2_strided(gp2_ptr(a, first), (size_t)a.gpl.elems, a.gp2_stride, count, init[GG_LAYER_GROUND], init[GG_LAYER_GROUNDPATCH], a.gp_valid, on);
    };
    if (split) {
        if (!ctx->probe_no_fork) {
            HIPCHK(ctx, hipEventRecord(ctx->half_fork, st));
            HIPCHK(ctx, hipStreamWaitEvent(ctx->half_stream, ctx->half_fork, 0));
        }
        fill(first_slot, boundary - first_slot, st);
        fill(boundary, first_slot + n - boundary, ctx->half_stream);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(ctx->half_done, ctx->half_stream));
        ctx->have_half_event = true;
    } else {
        fill(first_slot, n, st);
        HIPCHK(ctx, hipGetLastError());
    }
    if (st == ctx->stream) return own_stream_mutated_map(ctx);
    HIPCHK(ctx, hipEventRecord(ctx->batch_event, st));
    ctx->have_batch_event = true;
    ctx->last_batch_stream = st;
    return GG_OK;
}

int gg_reset_map(gg_context *ctx, int slot, double pos_x, double pos_y, float odom_z)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    return gg_reset_maps(ctx, slot, 1, pos_x, pos_y, odom_z, 0, nullptr);
}

int gg_set_map_position(gg_context *ctx, int slot, double pos_x, double pos_y)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    ctx->pos_x[slot] = pos_x;
    ctx->pos_y[slot] = pos_y;
    return GG_OK;
}

int gg_move_map(gg_context *ctx, int slot, double odom_x, double odom_y, const double base_plane[4], int shift_out[2])
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!base_plane) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const double res = ctx->arena.g.resolution;
    // grid_map_core getIndexShiftFromPositionShift: round half away from zero, map frame -> buffer order (sign flip)
    const double odom[2] = {odom_x, odom_y};
    const double pos[2] = {ctx->pos_x[slot], ctx->pos_y[slot]};
    int s[2];
    for (int i = 0; i < 2; ++i) {
        const double tmp = (odom[i] - pos[i]) / res;
        s[i] = -(int)(tmp + 0.5 * (tmp > 0 ? 1 : -1));
    }
    if (shift_out) {
        shift_out[0] = s[0];
        shift_out[1] = s[1];
    }
    if (s[0] == 0 && s[1] == 0) return GG_OK; // src/GroundGrid.cpp:135-137
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    // getPositionShiftFromIndexShift: the position advances by whole cells, not to the odometry position
    ctx->pos_x[slot] += (double)(-s[0]) * res;
    ctx->pos_y[slot] += (double)(-s[1]) * res;
    launch_scroll(ctx->arena, slot, reinterpret_cast<float2 *>(ctx->d_scroll_scratch), s[0], s[1], ctx->pos_x[slot], ctx->pos_y[slot], base_plane, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    return own_stream_mutated_map(ctx);
}

int gg_get_map_position(const gg_context *ctx, int slot, double *pos_x, double *pos_y)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (pos_x) *pos_x = ctx->pos_x[slot];
    if (pos_y) *pos_y = ctx->pos_y[slot];
    return GG_OK;
}

int gg_set_layer(gg_context *ctx, int slot, int layer, const float *src)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!src || layer < 0 || layer >= GG_NUM_LAYERS) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    if (layer == GG_LAYER_GROUNDPATCH) ctx->no_confidence[slot] = 0;
    if (layer == GG_LAYER_GROUND || layer == GG_LAYER_GROUNDPATCH) { // de-interleave at the host boundary
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_image, src, (size_t)ctx->arena.g.C * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_plane_insert(ctx->arena, slot, layer == GG_LAYER_GROUNDPATCH, ctx->d_image, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
    } else {
        // the per-call layers are stored sparsely behind ONE set of liveness masks (gg_internal.h tile_live): make all nine dense
        // (reset values into the dead half columns, every half column live), then overwrite this one with the host's matrix
        if (const int rc = ensure_lazy_layers(ctx, slot, true)) return rc;
        launch_materialise_layers(ctx->arena, slot, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_image, src, (size_t)ctx->arena.g.C * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_layer_insert(ctx->arena, slot, layer, ctx->d_image, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
    }
    if (const int rc = own_stream_mutated_map(ctx)) return rc;
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

int gg_get_layer(gg_context *ctx, int slot, int layer, float *dst)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!dst || layer < 0 || layer >= GG_NUM_LAYERS) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    // both kinds of layer have a device representation of their own (sheared pairs / sparse tile blocks): extract the dense plane
    if (const int rc = ensure_lazy_layers(ctx, slot, lazy_layer(layer))) return rc;
    if (layer == GG_LAYER_GROUND || layer == GG_LAYER_GROUNDPATCH)
        launch_plane_extract(ctx->arena, slot, layer == GG_LAYER_GROUNDPATCH, ctx->d_image, ctx->stream);
    else
        launch_layer_extract(ctx->arena, slot, layer, ctx->d_image, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    const float *plane = ctx->d_image;
    HIPCHK(ctx, hipMemcpyAsync(dst, plane, (size_t)ctx->arena.g.C * 4, hipMemcpyDeviceToHost, ctx->stream));
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

// dst[l] (nullable) receives layer l; the destinations need not be aligned (they may sit inside a serialised message)
static int get_layers_impl(gg_context *ctx, int slot, void *const dst[GG_NUM_LAYERS])
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    const size_t plane = align_up((size_t)ctx->arena.g.C * 4, 256) / 4;
    if (!ctx->d_planes) HIPCHK(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)GG_NUM_LAYERS * plane * sizeof(float)));
    if (!ctx->h_planes) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_planes, (size_t)GG_NUM_LAYERS * plane * sizeof(float), hipHostMallocDefault));
    // The requested planes are extracted side by side, come down in ONE copy into pinned memory (a copy into the caller's
    // pageable matrices is staged by the runtime in small pieces: 0.7 ms for eleven 364 x 364 layers, against 0.15 ms), and the
    // context's host threads move them on to where the caller wants them.
    int want[GG_NUM_LAYERS], n_want = 0;
    for (int l = 0; l < GG_NUM_LAYERS; ++l)
        if (const int rc = ensure_lazy_layers(ctx, slot, dst[l] && lazy_layer(l))) return rc;
    unsigned want_mask = 0u;
    for (int l = 0; l < GG_NUM_LAYERS; ++l) {
        if (!dst[l]) continue;
        want_mask |= 1u << l;
        want[n_want++] = l;
    }
    if (n_want) launch_layers_extract(ctx->arena, slot, want_mask, ctx->d_planes, plane, ctx->stream); // (plane k = the k-th requested layer)
    if (n_want == 0) return GG_OK;
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_planes, ctx->d_planes, (size_t)n_want * plane * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    const size_t C = (size_t)ctx->arena.g.C;
    ctx->helper.split((size_t)n_want * C, [&](size_t lo, size_t hi) {
        for (size_t k = lo / C; k < (size_t)n_want && k * C < hi; ++k) {
            const size_t a0 = std::max(lo, k * C) - k * C, a1 = std::min(hi, (k + 1) * C) - k * C;
            memcpy((char *)dst[want[k]] + a0 * sizeof(float), ctx->h_planes + k * plane + a0, (a1 - a0) * sizeof(float));
        }
    });
    return GG_OK;
}

int gg_get_layers(gg_context *ctx, int slot, float *const dst[GG_NUM_LAYERS])
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!dst) return GG_ERR_INVALID;
    void *d[GG_NUM_LAYERS];
    for (int l = 0; l < GG_NUM_LAYERS; ++l) d[l] = dst[l];
    return get_layers_impl(ctx, slot, d);
}

// grid_map_msgs/GridMap, ROS 1 serialisation (include/groundgrid_hip.h): the skeleton is written by the host, the layer planes land
// in their data[] arrays straight from the pinned download
int gg_get_gridmap_message(gg_context *ctx, int slot, unsigned layer_mask, const gg_gridmap_header *header, uint8_t *dst, size_t capacity, size_t *size)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!size) return fail(ctx, GG_ERR_INVALID, "gg_get_gridmap_message: null size");
    static const char *const names[GG_NUM_LAYERS] = {"points", "ground", "groundpatch", "minGroundHeight", "maxGroundHeight", "groundCandidates",
                                                     "planeDist", "m2", "meanVariance", "pointsRaw", "variance"}; // src/GroundGrid.cpp:55, src/GroundSegmentation.cpp:61-75
    if (layer_mask == 0u) layer_mask = (1u << GG_NUM_LAYERS) - 1u;
    layer_mask &= (1u << GG_NUM_LAYERS) - 1u;
    const gg_gridmap_header none{};
    const gg_gridmap_header &h = header ? *header : none;
    const char *frame = h.frame_id ? h.frame_id : "map";
    const Geometry &g = ctx->arena.g;
    const size_t C = (size_t)g.C;
    // two passes over the same writer: the first only measures
    size_t data_at[GG_NUM_LAYERS] = {};
    auto write = [&](uint8_t *out) -> size_t {
        size_t at = 0;
        auto put = [&](const void *p, size_t n) {
            if (out) memcpy(out + at, p, n);
            at += n;
        };
        auto u32 = [&](uint32_t v) { put(&v, 4); };
        auto f64 = [&](double v) { put(&v, 8); };
        auto str = [&](const char *t) {
            u32((uint32_t)strlen(t));
            put(t, strlen(t));
        };
        u32(h.seq); // info.header
        u32(h.stamp_sec);
        u32(h.stamp_nsec);
        str(frame);
        f64(g.resolution);
        f64(g.length0);
        f64(g.length1);
        f64(ctx->pos_x[slot]); // pose.position
        f64(ctx->pos_y[slot]);
        f64(0.0);
        f64(0.0); // pose.orientation
        f64(0.0);
        f64(0.0);
        f64(1.0);
        u32((uint32_t)__builtin_popcount(layer_mask)); // layers
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if ((layer_mask >> l) & 1u) str(names[l]);
        u32((uint32_t)__builtin_popcount(h.basic_layers & layer_mask)); // basic_layers
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if ((h.basic_layers & layer_mask) >> l & 1u) str(names[l]);
        u32((uint32_t)__builtin_popcount(layer_mask)); // data
        for (int l = 0; l < GG_NUM_LAYERS; ++l) {
            if (!((layer_mask >> l) & 1u)) continue;
            u32(2u); // layout.dim (GridMapMsgHelpers: column-major storage -> "column_index" first)
            str("column_index");
            u32((uint32_t)g.cols);
            u32((uint32_t)C);
            str("row_index");
            u32((uint32_t)g.rows);
            u32((uint32_t)g.rows);
            u32(0u); // layout.data_offset
            u32((uint32_t)C);
            data_at[l] = at;
            at += C * 4;
        }
        const uint16_t zero = 0;
        put(&zero, 2); // outer_start_index
        put(&zero, 2); // inner_start_index
        return at;
    };
    const size_t need = write(nullptr);
    *size = need;
    if (!dst) return GG_OK;
    if (capacity < need) return fail(ctx, GG_ERR_CAPACITY, "gg_get_gridmap_message: buffer smaller than the message");
    write(dst);
    void *planes[GG_NUM_LAYERS];
    for (int l = 0; l < GG_NUM_LAYERS; ++l) planes[l] = ((layer_mask >> l) & 1u) ? (void *)(dst + data_at[l]) : nullptr;
    return get_layers_impl(ctx, slot, planes);
}

int gg_get_layer_image_u8(gg_context *ctx, int slot, int layer, uint8_t *dst, float *lower, float *upper)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!dst || layer < 0 || layer >= GG_NUM_LAYERS) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    const Geometry &g = ctx->arena.g;
    uint8_t *d_img = reinterpret_cast<uint8_t *>(ctx->d_image);
    if (const int rc = ensure_lazy_layers(ctx, slot, lazy_layer(layer))) return rc;
    if (layer == GG_LAYER_GROUND || layer == GG_LAYER_GROUNDPATCH)
        launch_plane_extract(ctx->arena, slot, layer == GG_LAYER_GROUNDPATCH, ctx->d_scroll_scratch, ctx->stream);
    else
        launch_layer_extract(ctx->arena, slot, layer, ctx->d_scroll_scratch, ctx->stream);
    const float *plane = ctx->d_scroll_scratch;
    launch_layer_to_u8(plane, g.rows, g.cols, ctx->d_bounds, d_img, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    float b[2];
    HIPCHK(ctx, hipMemcpyAsync(dst, d_img, (size_t)g.C, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(b, ctx->d_bounds, sizeof b, hipMemcpyDeviceToHost, ctx->stream));
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (lower) *lower = b[0];
    if (upper) *upper = b[1];
    return GG_OK;
}

int gg_get_terrain_image(gg_context *ctx, int slot, float *dst)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!dst) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    const Geometry &g = ctx->arena.g;
    launch_terrain_image(ctx->arena, slot, ctx->d_image, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(dst, ctx->d_image, (size_t)g.C * 3 * 4, hipMemcpyDeviceToHost, ctx->stream));
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

// sensor_msgs/PointCloud2 payload -> the packed records the device reads, in one host pass (the reference goes
// wire -> 32-byte PCL points first, pcl::fromROSMsg at src/GroundGridNodelet.cpp:120)
int gg_filter_cloud_pc2(gg_context *ctx, int slot, const uint8_t *data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                        size_t off_z, size_t off_ring, const double *map_from_cloud, const float origin[3], double base_z,
                        uint8_t *out_label, int32_t *out_index, size_t *out_n)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if ((!data && n) || !origin) return fail(ctx, GG_ERR_INVALID, "null data / origin");
    if (n > ctx->max_points) return fail(ctx, GG_ERR_CAPACITY, "cloud larger than max_points");
    if (off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step || off_ring + 2 > point_step)
        return fail(ctx, GG_ERR_INVALID, "field offset outside point_step");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    for (size_t i = 0; i < n; ++i) {
        const uint8_t *p = data + i * point_step;
        gg_point16 &d = ctx->h_stage_pts[i];
        memcpy(&d.x, p + off_x, 4);
        memcpy(&d.y, p + off_y, 4);
        memcpy(&d.z, p + off_z, 4);
        memcpy(&d.ring, p + off_ring, 2);
        d.pad = 0;
    }
    if (n) HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage_pts, ctx->h_stage_pts, n * sizeof(gg_point16), hipMemcpyHostToDevice, s));
    const int32_t n32 = (int32_t)n;
    gg_batch b{};
    b.n_clouds = 1;
    b.first_slot = slot;
    b.point_format = GG_POINT16;
    b.d_points = ctx->d_stage_pts;
    b.cloud_stride = ctx->max_points;
    b.n_points = &n32;
    b.origins = origin;
    b.base_z = &base_z;
    b.transforms = map_from_cloud;
    b.d_labels = ctx->d_stage_labels;
    b.d_out_index = ctx->d_stage_index;
    b.d_out_counts = ctx->d_stage_counts;
    const int rc = enqueue_batch(ctx, &b, s);
    if (rc != GG_OK) return rc;
    if (n && out_label) HIPCHK(ctx, hipMemcpyAsync(out_label, ctx->d_stage_labels, n, hipMemcpyDeviceToHost, s));
    if (n && out_index) HIPCHK(ctx, hipMemcpyAsync(out_index, ctx->d_stage_index, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_stage_counts, ctx->d_stage_counts, 16, hipMemcpyDeviceToHost, s));
    SYNCCHK(ctx, hipStreamSynchronize(s));
    if (out_n) *out_n = (size_t)ctx->h_stage_counts[0];
    return GG_OK;
}

// ... and with the returned cloud coming back as 18-byte PointCloud2 records written by the label kernel (gg_batch.d_out_pc2)
int gg_filter_cloud_pc2_out(gg_context *ctx, int slot, const uint8_t *data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                            size_t off_z, size_t off_ring, const double *map_from_cloud, const float origin[3], double base_z,
                            uint8_t *out_data, size_t *out_n)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if ((!data && n) || !origin || (!out_data && n)) return fail(ctx, GG_ERR_INVALID, "null data / origin / out_data");
    if (n > ctx->max_points) return fail(ctx, GG_ERR_CAPACITY, "cloud larger than max_points");
    if (off_x + 4 > point_step || off_y + 4 > point_step || off_z + 4 > point_step || off_ring + 2 > point_step)
        return fail(ctx, GG_ERR_INVALID, "field offset outside point_step");
    if (ctx->next_ticket != ctx->oldest_ticket) return fail(ctx, GG_ERR_INVALID, "gg_filter_cloud_pc2_out while async tickets are outstanding");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t block = 64 + ctx->max_points * GG_PC2_POINT_STEP;
    if (!ctx->d_pc2) HIPCHK(ctx, hipMalloc((void **)&ctx->d_pc2, block));
    if (!ctx->h_pc2) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_pc2, block, hipHostMallocDefault));
    hipStream_t s = ctx->stream;
    for (int c = 0; c < 2; ++c) { // pack and upload in two pieces, as the 32-byte entry point does
        const size_t lo = n * c / 2, hi = n * (c + 1) / 2;
        if (hi == lo) continue;
        ctx->helper.split(hi - lo, [&](size_t a0, size_t a1) {
            for (size_t i = lo + a0; i < lo + a1; ++i) {
                const uint8_t *p = data + i * point_step;
                gg_point16 &d = ctx->h_stage_pts[i];
                memcpy(&d.x, p + off_x, 4);
                memcpy(&d.y, p + off_y, 4);
                memcpy(&d.z, p + off_z, 4);
                memcpy(&d.ring, p + off_ring, 2);
                d.pad = 0;
            }
        });
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage_pts + lo, ctx->h_stage_pts + lo, (hi - lo) * sizeof(gg_point16), hipMemcpyHostToDevice, s));
    }
    const int32_t n32 = (int32_t)n;
    gg_batch b{};
    b.n_clouds = 1;
    b.first_slot = slot;
    b.point_format = GG_POINT16;
    b.d_points = ctx->d_stage_pts;
    b.cloud_stride = ctx->max_points;
    b.n_points = &n32;
    b.origins = origin;
    b.base_z = &base_z;
    b.transforms = map_from_cloud;
    b.d_out_counts = reinterpret_cast<int32_t *>(ctx->d_pc2);
    b.d_out_pc2 = ctx->d_pc2 + 64;
    const int rc = enqueue_batch(ctx, &b, s);
    if (rc != GG_OK) return rc;
    // counts and records in one copy: the size of the returned cloud is only known on the device, and it is nearly n
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pc2, ctx->d_pc2, 64 + n * GG_PC2_POINT_STEP, hipMemcpyDeviceToHost, s));
    SYNCCHK(ctx, hipStreamSynchronize(s));
    const size_t got = (size_t)reinterpret_cast<const int32_t *>(ctx->h_pc2)[0];
    if (out_n) *out_n = got;
    const uint8_t *src = ctx->h_pc2 + 64;
    ctx->helper.split(got * GG_PC2_POINT_STEP, [&](size_t a0, size_t a1) { memcpy(out_data + a0, src + a0, a1 - a0); });
    return GG_OK;
}

int gg_get_expected_points(const gg_context *ctx, float *dst)
{
    if (!ctx || !dst) return GG_ERR_INVALID;
    memcpy(dst, ctx->h_expected.data(), ctx->h_expected.size() * 4);
    return GG_OK;
}

int gg_filter_batch(gg_context *ctx, const gg_batch *b, void *stream)
{
    if (!ctx || !b) return GG_ERR_INVALID;
    if (b->n_clouds < 0 || b->n_clouds > ctx->n_slots) return fail(ctx, GG_ERR_CAPACITY, "slot range");
    if (!b->slots && (b->first_slot < 0 || b->first_slot + b->n_clouds > ctx->n_slots)) return fail(ctx, GG_ERR_CAPACITY, "slot range");
    if (b->n_clouds == 0) return GG_OK;
    if (b->slots) { // distinct and in range: two clouds of one launch on one map would race
        std::vector<char> &seen = ctx->slot_seen;
        seen.assign((size_t)ctx->n_slots, 0);
        for (int i = 0; i < b->n_clouds; ++i) {
            const int s = b->slots[i];
            if (s < 0 || s >= ctx->n_slots) return fail(ctx, GG_ERR_CAPACITY, "gg_batch.slots entry outside the context");
            if (seen[(size_t)s]) return fail(ctx, GG_ERR_INVALID, "gg_batch.slots entries must be distinct");
            seen[(size_t)s] = 1;
        }
    }
    if (!b->d_points || !b->n_points || !b->origins || !b->base_z) return fail(ctx, GG_ERR_INVALID, "null batch field");
    if (b->point_format != GG_POINT32 && b->point_format != GG_POINT16) return fail(ctx, GG_ERR_INVALID, "point_format");
    if (b->d_out_clouds && b->point_format != GG_POINT32) return fail(ctx, GG_ERR_INVALID, "d_out_clouds needs GG_POINT32 input");
    if (b->d_label_masks && (b->cloud_stride & 3u)) return fail(ctx, GG_ERR_INVALID, "d_label_masks needs cloud_stride % 4 == 0");
    for (int i = 0; i < b->n_clouds; ++i)
        if (b->n_points[i] < 0 || (size_t)b->n_points[i] > ctx->max_points || (size_t)b->n_points[i] > b->cloud_stride)
            return fail(ctx, GG_ERR_CAPACITY, "n_points exceeds max_points / cloud_stride");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return enqueue_batch(ctx, b, pick_stream(ctx, stream), nullptr, true);
}

int gg_device_error(gg_context *ctx, int clear)
{
    if (!ctx || !ctx->h_dev_error) return GG_ERR_INVALID;
    const uint32_t code = *ctx->h_dev_error;
    if (clear) *ctx->h_dev_error = GG_DEVERR_NONE;
    return (int)code;
}

int gg_batch_fence(gg_context *ctx, void *stream)
{
    if (!ctx) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return stream_waits_for_second_half(ctx, pick_stream(ctx, stream));
}

int gg_synchronize(gg_context *ctx)
{
    if (!ctx) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc; // batches on caller streams (and their second halves) included
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

// one ticket: pack + upload, the seven kernels, the download.  `pipelined`: uploads and downloads on their own streams so that
// they overlap the neighbouring tickets' kernels (gg_filter_cloud_async); a synchronous call has nothing to overlap with and puts
// everything on the context's stream instead -- no cross-stream events (four API calls, ~15 us per cloud)
static int enqueue_ticket(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *tf, const float origin[3], double base_z,
                          int *ticket, bool pipelined, const LayerPlan *plan = nullptr)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if ((!cloud && n) || !origin || !ticket) return fail(ctx, GG_ERR_INVALID, "null cloud / origin / ticket");
    if (n > ctx->max_points) return fail(ctx, GG_ERR_CAPACITY, "cloud larger than max_points");
    if (ctx->next_ticket - ctx->oldest_ticket >= GG_ASYNC_DEPTH) return fail(ctx, GG_ERR_CAPACITY, "GG_ASYNC_DEPTH tickets outstanding: wait for the oldest first");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // (a synchronous call has no ticket outstanding: always the same staging set, so that its launches are one kind for the graph cache)
    const int set = pipelined ? ctx->next_ticket % GG_ASYNC_DEPTH : 0;
    ctx->slot_of_ticket[ctx->next_ticket % GG_ASYNC_DEPTH] = set;
    gg_context::AsyncSlot &as = ctx->async_slot[set];

    static const bool host_timing = getenv("GG_HOST_TIMING") != nullptr; // (tools: where does the host call spend its time)
    const auto t_pack0 = std::chrono::steady_clock::now();
    // pack and upload in pieces (each packed by all of the context's host threads): the copy of the first travels while the
    // second is packed (and all of it overlaps the device work of the previous ticket)
    const int pieces = std::max(1, std::min(ctx->upload_pieces, 8));
    for (int c = 0; c < pieces; ++c) {
        const size_t lo = n * c / pieces, hi = n * (c + 1) / pieces;
        if (hi == lo) continue;
        ctx->helper.split(hi - lo, [&](size_t a0, size_t a1) { pack_points(cloud + lo + a0, as.h_pts + lo + a0, a1 - a0); });
        HIPCHK(ctx, hipMemcpyAsync(as.d_pts + lo, as.h_pts + lo, (hi - lo) * sizeof(gg_point16), hipMemcpyHostToDevice, pipelined ? ctx->h2d_stream : ctx->stream));
    }
    if (host_timing) ctx->host_t[0] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack0).count();
    if (pipelined) {
        HIPCHK(ctx, hipEventRecord(as.uploaded, ctx->h2d_stream));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, as.uploaded, 0));
    }

    const int32_t n32 = (int32_t)n;
    gg_batch b{};
    b.n_clouds = 1;
    b.first_slot = slot;
    b.point_format = GG_POINT16;
    b.d_points = as.d_pts;
    b.cloud_stride = ctx->max_points;
    b.n_points = &n32;
    b.origins = origin;
    b.base_z = &base_z;
    b.transforms = tf;
    as.d_labels = reinterpret_cast<uint8_t *>(as.d_index + n); // [counts][index: n][labels: n]
    as.h_labels = reinterpret_cast<uint8_t *>(as.h_index + n);
    const bool direct = ctx->results_direct != 0 && as.hd_counts != nullptr;
    int32_t *const res_counts = direct ? as.hd_counts : as.d_counts, *const res_index = direct ? as.hd_counts + 16 : as.d_index;
    b.d_labels = reinterpret_cast<uint8_t *>(res_index); // (+ label_shift = 4 n on the device: the launch's arguments do not depend on n)
    ctx->next_label_shift = (int)(n * 4);
    b.d_out_index = res_index;
    b.d_out_clouds = nullptr;
    b.d_out_counts = res_counts;
    const int rc = enqueue_batch(ctx, &b, ctx->stream, plan);
    ctx->next_label_shift = 0;
    if (rc != GG_OK) return rc;
    if (direct) {
        HIPCHK(ctx, hipEventRecord(as.downloaded, ctx->stream)); // (the results are in host memory when k_label is done)
    } else {
        // results come back on their own stream, so that the next ticket's kernels do not queue behind this download
        const hipStream_t down = pipelined ? ctx->d2h_stream : ctx->stream;
        if (pipelined) {
            HIPCHK(ctx, hipEventRecord(as.computed, ctx->stream));
            HIPCHK(ctx, hipStreamWaitEvent(down, as.computed, 0));
        }
        HIPCHK(ctx, hipMemcpyAsync(as.h_counts, as.d_counts, 64 + n * 5, hipMemcpyDeviceToHost, down)); // counts + index + labels: one copy
        HIPCHK(ctx, hipEventRecord(as.downloaded, down));
    }
    if (plan && (plan->mask & ~EARLY_LAYERS)) { // the layers that are only final now (ground, groundpatch, points) follow the results: the host
        Arena a = ctx->arena;                   // assembles the returned cloud while they travel
        enqueue_layer_downloads(ctx, a, slot, plan->mask & ~EARLY_LAYERS, *plan, ctx->stream);
        HIPCHK(ctx, hipGetLastError());
    }
    if (plan) HIPCHK(ctx, hipEventRecord(ctx->layers_event, ctx->stream));

    as.cloud = cloud;
    as.n = n;
    as.has_tf = tf != nullptr;
    if (tf) memcpy(as.tf, tf, sizeof as.tf);
    as.ticket = ctx->next_ticket;
    *ticket = ctx->next_ticket++;
    if (host_timing) ctx->host_t[1] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_pack0).count();
    return GG_OK;
}

int gg_filter_cloud_async(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *tf, const float origin[3],
                          double base_z, int *ticket)
{
    return enqueue_ticket(ctx, slot, cloud, n, tf, origin, base_z, ticket, true);
}

int gg_filter_cloud_wait(gg_context *ctx, int ticket, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label, int32_t *out_index)
{
    if (!ctx) return GG_ERR_INVALID;
    if (ticket != ctx->oldest_ticket || ticket >= ctx->next_ticket) return fail(ctx, GG_ERR_INVALID, "tickets are waited for in issue order");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    gg_context::AsyncSlot &as = ctx->async_slot[ctx->slot_of_ticket[ticket % GG_ASYNC_DEPTH]];
    ctx->oldest_ticket = ticket + 1; // (also on error below: the slot is reusable either way)
    static const bool host_timing = getenv("GG_HOST_TIMING") != nullptr;
    const auto t_w0 = std::chrono::steady_clock::now();
    SYNCCHK(ctx, hipEventSynchronize(as.downloaded));
    const auto t_w1 = std::chrono::steady_clock::now();
    const size_t n = as.n;
    if (out_n) *out_n = (size_t)as.h_counts[0];
    if (n && (out_label || out_index || out_cloud)) {
        // the returned cloud (:173-189): the host owns the input, so it assembles the output from index + label -- every input point has
        // its own position in the returned cloud (the pieces of the input write disjoint records); labels and index are copied out by
        // the same threads
        const gg_point32 *cloud = as.cloud;
        const double *tf = as.has_tf ? as.tf : nullptr;
        const int32_t *h_index = as.h_index;
        const uint8_t *h_labels = as.h_labels;
        ctx->helper.split(n, [&](size_t i0, size_t i1) {
            if (out_label) memcpy(out_label + i0, h_labels + i0, i1 - i0);
            if (out_index) memcpy(out_index + i0, h_index + i0, (i1 - i0) * 4);
            if (out_cloud) assemble_returned_cloud(cloud, h_index, h_labels, tf, out_cloud, i0, i1);
        });
    }
    if (host_timing) {
        ctx->host_t[2] += std::chrono::duration<double>(t_w1 - t_w0).count();
        ctx->host_t[3] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w1).count();
        if (++ctx->host_calls % 256 == 0) {
            fprintf(stderr, "gg host path, ms per call: pack+upload calls %.3f, whole enqueue %.3f, wait %.3f, copy-out + assemble %.3f\n",
                    1e3 * ctx->host_t[0] / 256, 1e3 * ctx->host_t[1] / 256, 1e3 * ctx->host_t[2] / 256, 1e3 * ctx->host_t[3] / 256);
            ctx->host_t[0] = ctx->host_t[1] = ctx->host_t[2] = ctx->host_t[3] = 0.0;
        }
    }
    return GG_OK;
}

// the synchronous reference-shaped call = one ticket, waited for at once
static int filter_cloud_impl(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *tf, const float origin[3],
                             double base_z, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label, int32_t *out_index)
{
    if (!ctx) return GG_ERR_INVALID;
    if (ctx->next_ticket != ctx->oldest_ticket) return fail(ctx, GG_ERR_INVALID, "gg_filter_cloud while async tickets are outstanding");
    int ticket = -1;
    const int rc = enqueue_ticket(ctx, slot, cloud, n, tf, origin, base_z, &ticket, false);
    if (rc != GG_OK) return rc;
    return gg_filter_cloud_wait(ctx, ticket, out_cloud, out_n, out_label, out_index);
}

int gg_filter_cloud(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const float origin[3], double base_z,
                    gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label, int32_t *out_index)
{
    return filter_cloud_impl(ctx, slot, cloud, n, nullptr, origin, base_z, out_cloud, out_n, out_label, out_index);
}

int gg_filter_cloud_tf(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double map_from_cloud[12],
                       const float origin[3], double base_z, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label,
                       int32_t *out_index)
{
    if (!map_from_cloud) return GG_ERR_INVALID;
    return filter_cloud_impl(ctx, slot, cloud, n, map_from_cloud, origin, base_z, out_cloud, out_n, out_label, out_index);
}

// gg_filter_cloud + the layers the caller publishes, as one call (include/groundgrid_hip.h): the eight layers that are final after
// k_reduce are extracted and downloaded on a side branch while the stencil and the sweep run, the other three follow the results
static bool host_range_registered(const gg_context *ctx, const void *p, size_t bytes)
{
    const char *c = static_cast<const char *>(p);
    for (const auto &r : ctx->registered)
        if (c >= r.first && c + bytes <= r.first + r.second) return true;
    return false;
}

int gg_host_register(gg_context *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || !bytes) return GG_ERR_INVALID;
    if (host_range_registered(ctx, ptr, bytes)) return GG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    ctx->registered.emplace_back(static_cast<char *>(ptr), bytes);
    return GG_OK;
}

int gg_host_unregister(gg_context *ctx, void *ptr)
{
    if (!ctx || !ptr) return GG_ERR_INVALID;
    for (auto it = ctx->registered.begin(); it != ctx->registered.end(); ++it)
        if (it->first == static_cast<char *>(ptr)) {
            HIPCHK(ctx, hipSetDevice(ctx->device));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); // (nothing in flight may still write into it)
            HIPCHK(ctx, hipStreamSynchronize(ctx->d2h_stream));
            drop_graphs(ctx);                               // (captured downloads name the range)
            HIPCHK(ctx, hipHostUnregister(ptr));
            ctx->registered.erase(it);
            return GG_OK;
        }
    return fail(ctx, GG_ERR_INVALID, "gg_host_unregister: not a registered range");
}

int gg_filter_cloud_layers(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *map_from_cloud, const float origin[3], double base_z,
                           gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label, int32_t *out_index, float *const layers[GG_NUM_LAYERS])
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (ctx->next_ticket != ctx->oldest_ticket) return fail(ctx, GG_ERR_INVALID, "gg_filter_cloud_layers while async tickets are outstanding");
    unsigned mask = 0u;
    if (layers)
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if (layers[l]) mask |= 1u << l;
    if (!mask || (ctx->flags & GG_FLAG_MINIMAL_LAYERS)) { // nothing to fuse / three layers are owed and computed on demand: one after the other
        const int rc = filter_cloud_impl(ctx, slot, cloud, n, map_from_cloud, origin, base_z, out_cloud, out_n, out_label, out_index);
        if (rc != GG_OK || !mask) return rc;
        void *d[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) d[l] = layers[l];
        return get_layers_impl(ctx, slot, d);
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t C = (size_t)ctx->arena.g.C, plane = align_up(C * 4, 256) / 4;
    if (!ctx->d_planes) HIPCHK(ctx, hipMalloc((void **)&ctx->d_planes, (size_t)GG_NUM_LAYERS * plane * sizeof(float)));
    if (!ctx->h_planes) HIPCHK(ctx, hipHostMalloc((void **)&ctx->h_planes, (size_t)GG_NUM_LAYERS * plane * sizeof(float), hipHostMallocDefault));
    LayerPlan plan;
    plan.mask = mask;
    bool staged[GG_NUM_LAYERS] = {};
    for (int l = 0; l < GG_NUM_LAYERS; ++l) {
        if (!((mask >> l) & 1u)) continue;
        staged[l] = !host_range_registered(ctx, layers[l], C * sizeof(float));
        plan.dst[l] = staged[l] ? (void *)(ctx->h_planes + (size_t)l * plane) : (void *)layers[l];
    }
    int ticket = -1;
    const int rc = enqueue_ticket(ctx, slot, cloud, n, map_from_cloud, origin, base_z, &ticket, false, &plan);
    if (rc != GG_OK) {
        // the side branch may already copy into the caller's (registered) planes: nothing of this call is in flight when it returns
        (void)hipStreamSynchronize(ctx->d2h_stream);
        (void)hipStreamSynchronize(ctx->stream);
        return rc;
    }
    // planes that came down into the staging block move on to the caller's: the early ones WHILE the device sweeps, the late ones at the end
    auto copy_staged = [&](unsigned group) {
        int want[GG_NUM_LAYERS], n_want = 0;
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if (staged[l] && ((group >> l) & 1u)) want[n_want++] = l;
        if (n_want)
            ctx->helper.split((size_t)n_want * C, [&](size_t lo, size_t hi) {
                for (size_t k = lo / C; k < (size_t)n_want && k * C < hi; ++k) {
                    const size_t a0 = std::max(lo, k * C) - k * C, a1 = std::min(hi, (k + 1) * C) - k * C;
                    memcpy(layers[want[k]] + a0, ctx->h_planes + (size_t)want[k] * plane + a0, (a1 - a0) * sizeof(float));
                }
            });
    };
    if (mask & EARLY_LAYERS) {
        HIPCHK(ctx, hipEventSynchronize(ctx->join_event)); // (the side branch: behind k_reduce, beside k_patch / k_sweep)
        copy_staged(EARLY_LAYERS);
    }
    const int rc_wait = gg_filter_cloud_wait(ctx, ticket, out_cloud, out_n, out_label, out_index); // (assembles the returned cloud while the late layers travel)
    SYNCCHK(ctx, hipEventSynchronize(ctx->layers_event));
    if (rc_wait != GG_OK) return rc_wait;
    if (ctx->layer_copy_failed) {
        ctx->layer_copy_failed = false;
        ctx->last_error = "gg_filter_cloud_layers: a layer download could not be enqueued";
        return GG_ERR_HIP;
    }
    copy_staged(~EARLY_LAYERS);
    return GG_OK;
}

// ---- RCCL, bound at run time (include/groundgrid_hip.h "the one collective of the path") --------------------------------
extern "C++" {
namespace {
struct Rccl {
    // the NCCL API as RCCL exports it (rccl.h): ncclResult_t == int, ncclSuccess == 0, ncclUint8 == 1, ncclComm_t / hipStream_t opaque
    struct UniqueId {
        char internal[128];
    };
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    Rccl()
    {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(h, "ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(h, "ncclAllGather"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        ok = GetUniqueId && CommInitRank && CommDestroy && AllGather;
    }
};
Rccl &rccl()
{
    static Rccl r; // (dlopen once, on first use)
    return r;
}
} // namespace
} // extern "C++"

int gg_collective_available(void) { return rccl().ok ? 1 : 0; }

int gg_comm_unique_id(uint8_t id_out[128])
{
    if (!id_out) return GG_ERR_INVALID;
    if (!rccl().ok) return GG_ERR_NO_DEVICE;
    Rccl::UniqueId id;
    if (rccl().GetUniqueId(&id) != 0) return GG_ERR_HIP;
    memcpy(id_out, id.internal, 128);
    return GG_OK;
}

int gg_comm_init_rank(const uint8_t id_in[128], int n_ranks, int rank, void **comm_out)
{
    if (!id_in || !comm_out || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return GG_ERR_INVALID;
    *comm_out = nullptr;
    if (!rccl().ok) return GG_ERR_NO_DEVICE;
    Rccl::UniqueId id;
    memcpy(id.internal, id_in, 128);
    if (rccl().CommInitRank(comm_out, n_ranks, id, rank) != 0) return GG_ERR_HIP;
    return GG_OK;
}

int gg_comm_init_rank_for(gg_context *ctx, const uint8_t id_in[128], int n_ranks, int rank, void **comm_out)
{
    if (!ctx) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = gg_comm_init_rank(id_in, n_ranks, rank, comm_out);
    if (rc != GG_OK) fail(ctx, rc, rc == GG_ERR_NO_DEVICE ? "librccl.so could not be loaded" : "ncclCommInitRank failed");
    return rc;
}

int gg_comm_destroy(void *comm)
{
    if (!comm) return GG_ERR_INVALID;
    if (!rccl().ok) return GG_ERR_NO_DEVICE;
    return rccl().CommDestroy(comm) == 0 ? GG_OK : GG_ERR_HIP;
}

int gg_allgather_label_masks(gg_context *ctx, void *comm, const uint8_t *d_send, uint8_t *d_recv, size_t bytes_per_rank, void *stream)
{
    if (!ctx) return GG_ERR_INVALID;
    if (!comm) return fail(ctx, GG_ERR_INVALID, "gg_allgather_label_masks: null communicator");
    if (!d_send || !d_recv || bytes_per_rank == 0) return fail(ctx, GG_ERR_INVALID, "gg_allgather_label_masks: null buffer / zero size");
    if (!rccl().ok) return fail(ctx, GG_ERR_NO_DEVICE, "librccl.so could not be loaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const hipStream_t s = pick_stream(ctx, stream);
    // the masks are written by the last batch: order the gather after it when it runs on another stream
    if (ctx->have_batch_event && ctx->last_batch_stream != s) HIPCHK(ctx, hipStreamWaitEvent(s, ctx->batch_event, 0));
    if (const int rc_half = stream_waits_for_second_half(ctx, s)) return rc_half;
    const int rc = rccl().AllGather(d_send, d_recv, bytes_per_rank, /* ncclUint8 */ 1, comm, s);
    if (rc != 0) {
        char buf[256];
        snprintf(buf, sizeof buf, "ncclAllGather: %s", rccl().GetErrorString ? rccl().GetErrorString(rc) : "error");
        ctx->last_error = buf;
        return GG_ERR_HIP;
    }
    // a later batch on another stream that writes into the buffer this gather still reads must wait for it (enqueue_batch)
    HIPCHK(ctx, hipEventRecord(ctx->gather_event, s));
    ctx->gather_stream = s;
    ctx->gather_lo = d_send;
    ctx->gather_hi = d_send + bytes_per_rank;
    return GG_OK;
}

int gg_get_point_classes(gg_context *ctx, int slot, size_t n, uint8_t *out_class, int32_t *out_cell)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (n > ctx->max_points) return GG_ERR_CAPACITY;
    if (n == 0) return GG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    launch_decode_classes(ctx->arena, slot, n, ctx->d_stage_class, ctx->d_stage_cell, ctx->stream);
    HIPCHK(ctx, hipGetLastError());
    if (out_class) HIPCHK(ctx, hipMemcpyAsync(out_class, ctx->d_stage_class, n, hipMemcpyDeviceToHost, ctx->stream));
    if (out_cell) HIPCHK(ctx, hipMemcpyAsync(out_cell, ctx->d_stage_cell, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    SYNCCHK(ctx, hipStreamSynchronize(ctx->stream));
    return GG_OK;
}

// The public stage members of the reference's class (include/groundgrid_hip.h "the stage members"), on the slot's layers as they stand.
int gg_run_stage(gg_context *ctx, int slot, int stage, const gg_stage_args *args)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!args) return fail(ctx, GG_ERR_INVALID, "gg_run_stage: null arguments");
    const Geometry &g = ctx->arena.g;
    switch (stage) {
    case GG_STAGE_DETECT_GROUND_PATCHES:
        if (args->section < -1 || args->section > 3) return fail(ctx, GG_ERR_INVALID, "gg_run_stage: section must be 0..3 or -1");
        break;
    case GG_STAGE_SPIRAL_GROUND_INTERPOLATION: break;
    case GG_STAGE_DETECT_GROUND_PATCH_3:
    case GG_STAGE_DETECT_GROUND_PATCH_5: {
        const int h = stage == GG_STAGE_DETECT_GROUND_PATCH_3 ? 1 : 2; // (the S x S blocks of :355 must stay inside the map: UB in the reference)
        if (args->i < h || args->j < h || args->i >= g.rows - h || args->j >= g.cols - h) return fail(ctx, GG_ERR_INVALID, "gg_run_stage: the cell's block leaves the map");
        break;
    }
    case GG_STAGE_INTERPOLATE_CELL:
        if (args->i < 1 || args->j < 1 || args->i >= g.rows - 1 || args->j >= g.cols - 1) return fail(ctx, GG_ERR_INVALID, "gg_run_stage: the cell's block leaves the map");
        break;
    default: return fail(ctx, GG_ERR_INVALID, "gg_run_stage: unknown stage");
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    const hipStream_t s = ctx->stream;
    Arena a = ctx->arena;
    a.flags = ctx->flags;
    a.eigen_reduction = ctx->conv.eigen_reduction;
    if (stage == GG_STAGE_DETECT_GROUND_PATCHES || stage == GG_STAGE_SPIRAL_GROUND_INTERPOLATION) {
        // the many-cell stages are the path's own kernels: they take their slot and base_z from a parameter record like a batch of one
        const int r = ctx->ring_next;
        ctx->ring_next = (r + 1) % PARAM_RING;
        if (ctx->ring_used[r]) HIPCHK(ctx, hipEventSynchronize(ctx->ring_done[r]));
        CloudParams *hp = ctx->h_params + (size_t)r * ctx->n_slots, *dp = ctx->d_params + (size_t)r * ctx->n_slots;
        hp[0] = CloudParams{};
        hp[0].slot = slot;
        hp[0].base_z = (float)args->base_z;
        hp[0].pos_x = ctx->pos_x[slot];
        hp[0].pos_y = ctx->pos_y[slot];
        HIPCHK(ctx, hipMemcpyAsync(dp, hp, sizeof(CloudParams), hipMemcpyHostToDevice, s));
        if (stage == GG_STAGE_DETECT_GROUND_PATCHES) {
            launch_patch_stage(a, dp, slot, args->section, s);
        } else {
            sweep::Params sp = ctx->sweep_params;
            sp.decrease = ctx->cfg.occupied_cells_decrease_factor;
            sp.inv_decrease = 1.0 / sp.decrease;
            sp.decay_fast = sp.decrease >= 1.25 && sp.decrease < 1e300;
            sp.keep_points = 1;
            launch_sweep(a, sp, dp, 1, s, nullptr);
        }
        HIPCHK(ctx, hipGetLastError());
        HIPCHK(ctx, hipEventRecord(ctx->ring_done[r], s));
        ctx->ring_used[r] = true;
    } else {
        launch_stage_cell(a, slot, stage, args->i, args->j, s);
        HIPCHK(ctx, hipGetLastError());
    }
    ctx->no_confidence[slot] = 0; // (every stage may write confidences)
    if (const int rc = own_stream_mutated_map(ctx)) return rc;
    SYNCCHK(ctx, hipStreamSynchronize(s));
    return GG_OK;
}

// GroundSegmentation::insert_cloud on the slot's map as it stands (include/groundgrid_hip.h): k_classify -> scan -> scatter on the range,
// then k_stage_insert continues the recurrences in the (densified) layers.
int gg_insert_cloud(gg_context *ctx, int slot, const gg_point32 *cloud, size_t start, size_t end, const float origin[3], uint8_t *out_class, int32_t *out_cell)
{
    if (!slot_ok(ctx, slot)) return GG_ERR_CAPACITY;
    if (!origin || end < start || (end > start && !cloud)) return fail(ctx, GG_ERR_INVALID, "gg_insert_cloud: null cloud / origin or end < start");
    const size_t n = end - start;
    if (n > ctx->max_points) return fail(ctx, GG_ERR_CAPACITY, "gg_insert_cloud: range larger than max_points");
    if (n == 0) return GG_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int rc = own_stream_waits_for_batches(ctx)) return rc;
    const hipStream_t s = ctx->stream;
    // the per-call layers dense: the three lazily kept ones computed, the reset values written into the dead half columns
    if (const int rc = ensure_lazy_layers(ctx, slot, true)) return rc;
    Arena a = ctx->arena;
    a.flags = ctx->flags & ~(unsigned)GG_FLAG_MINIMAL_LAYERS;
    a.eigen_reduction = ctx->conv.eigen_reduction;
    launch_materialise_layers(a, slot, s);
    HIPCHK(ctx, hipGetLastError());
    for (size_t i = 0; i < n; ++i) {
        const gg_point32 &p = cloud[start + i];
        gg_point16 &d = ctx->h_stage_pts[i];
        d.x = p.x, d.y = p.y, d.z = p.z, d.ring = p.ring, d.pad = 0;
    }
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage_pts, ctx->h_stage_pts, n * sizeof(gg_point16), hipMemcpyHostToDevice, s));
    const int r = ctx->ring_next;
    ctx->ring_next = (r + 1) % PARAM_RING;
    if (ctx->ring_used[r]) HIPCHK(ctx, hipEventSynchronize(ctx->ring_done[r]));
    CloudParams *hp = ctx->h_params + (size_t)r * ctx->n_slots, *dp = ctx->d_params + (size_t)r * ctx->n_slots;
    hp[0] = CloudParams{};
    hp[0].slot = slot;
    hp[0].n_points = (int)n;
    hp[0].ox = origin[0], hp[0].oy = origin[1], hp[0].oz = origin[2];
    hp[0].pos_x = ctx->pos_x[slot];
    hp[0].pos_y = ctx->pos_y[slot];
    hp[0].no_confidence = ctx->no_confidence[slot] ? 1 : 0;
    hp[0].fresh = 0;
    HIPCHK(ctx, hipMemcpyAsync(dp, hp, sizeof(CloudParams), hipMemcpyHostToDevice, s));
    BatchIO io{};
    io.d_points = ctx->d_stage_pts;
    io.cloud_stride = ctx->max_points;
    io.point_format = GG_POINT16;
    const int front = launch_classify(a, dp, io, 1, (int)n, s);
    if (front == FRONT_THREE_LAUNCHES) launch_scan(a, dp, 1, s);
    if (front != FRONT_ONE_LAUNCH) launch_scatter(a, dp, 1, (int)n, s);
    launch_stage_insert(a, dp, s);
    // (k_scan dropped the liveness masks of the tiles this range left empty; their values are all still there: every half column holds its values)
    launch_fill_bytes((uint8_t *)(a.tile_live + (size_t)slot * a.tile_live_stride), (size_t)a.g.T * 4, 0xFF, s);
    launch_decode_classes(a, slot, n, ctx->d_stage_class, ctx->d_stage_cell, s);
    HIPCHK(ctx, hipGetLastError());
    if (out_class) HIPCHK(ctx, hipMemcpyAsync(out_class, ctx->d_stage_class, n, hipMemcpyDeviceToHost, s));
    if (out_cell) HIPCHK(ctx, hipMemcpyAsync(out_cell, ctx->d_stage_cell, n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipEventRecord(ctx->ring_done[r], s));
    ctx->ring_used[r] = true;
    ctx->lazy_pending[slot] = 0; // (all nine layers are maintained and dense)
    if (const int rc = own_stream_mutated_map(ctx)) return rc;
    SYNCCHK(ctx, hipStreamSynchronize(s));
    return GG_OK;
}

// tools and tests only (not in the header): override the launch geometry the library would pick from the batch size (0 = back
// to the default).  key: "sweep_waves", "k2_per_cloud", "k2_dense_share".  Returns the chunk size PW for key "pw" (read-only:
// the arena is carved for it at gg_create; set GG_PW in the environment before gg_create to change it).
extern "C" int gg_debug_set_tuning(gg_context *ctx, const char *key, int value)
{
    if (!ctx || !key || value < 0) return GG_ERR_INVALID;
    if (!strcmp(key, "pw")) return ctx->arena.PW;
    if (!strcmp(key, "graph_replays")) return (int)std::min<long>(ctx->graph_replays, 1 << 30); // (read-only: how many calls replayed a captured graph)
    if (!strcmp(key, "graphs")) { // 0 = every call launches eagerly, 1 = one cloud per call replays a captured graph (the default)
        ctx->graphs_enabled = value != 0;
        drop_graphs(ctx);
        return GG_OK;
    }
    drop_graphs(ctx); // (every other key changes what the launchers pick)
    if (!strcmp(key, "sweep_waves")) ctx->arena.tune_sweep_waves = value;
    else if (!strcmp(key, "sweep_gpw")) ctx->arena.tune_sweep_gpw = value;
    else if (!strcmp(key, "sweep_split")) ctx->arena.tune_sweep_split = value;
    else if (!strcmp(key, "sweep_pair")) ctx->arena.tune_sweep_pair = value;
    else if (!strcmp(key, "sweep_pair_wgs")) ctx->arena.tune_sweep_pair_wgs = value;
    else if (!strcmp(key, "sweep_pair_waves")) ctx->arena.tune_sweep_pair_waves = value;
    else if (!strcmp(key, "fresh_maps")) ctx->fresh_enabled = value != 0;
    else if (!strcmp(key, "front")) ctx->arena.tune_front = std::min(value, 3);
    else if (!strcmp(key, "sweep_poll_cap")) ctx->arena.tune_sweep_poll_cap = value;
    else if (!strcmp(key, "scan_parts")) ctx->arena.tune_scan_parts = value;
    else if (!strcmp(key, "scan_poll_cap")) ctx->arena.tune_scan_poll_cap = value;
    else if (!strcmp(key, "upload_pieces")) ctx->upload_pieces = value;
    else if (!strcmp(key, "results_direct")) ctx->results_direct = value; // (A/B: 0 = results into HBM and a copy behind k_label, as before round 5)
    else if (!strcmp(key, "halves_min_clouds")) ctx->halves_min_clouds = std::max(2, value); // (tests: GG_FLAG_CONCURRENT_HALVES on small batches)
    else if (!strcmp(key, "halves_no_fork")) ctx->probe_no_fork = value != 0; // (measurement only: the side stream does not wait for the caller's)
    else if (!strcmp(key, "scan_fault")) ctx->arena.tune_scan_fault = value;
    else if (!strcmp(key, "sweep_fault")) ctx->arena.tune_sweep_fault = value;
    else if (!strcmp(key, "probe_unordered_streams")) ctx->probe_unordered_streams = value != 0; // (tools/fill_overlap_probe.py: the CALLER orders its streams)
    else if (!strcmp(key, "clear_device_error")) *ctx->h_dev_error = 0u;
    else if (!strcmp(key, "k2_per_cloud")) ctx->arena.tune_k2_per_cloud = value;
    else if (!strcmp(key, "k2_dense_share")) ctx->arena.tune_k2_dense_share = std::min(value, 15);
    else return GG_ERR_INVALID;
    return GG_OK;
}

// tools only (not in the header): cycle counters written by the last sweep when GG_SWEEP_TIMING=1 was set at gg_create
extern "C" int gg_debug_sweep_timing(gg_context *ctx, unsigned long long out[64])
{
    if (!ctx || !out || !ctx->d_sweep_dbg) return GG_ERR_INVALID;
    if (hipMemcpy(out, ctx->d_sweep_dbg, 64 * 8, hipMemcpyDeviceToHost) != hipSuccess) return GG_ERR_HIP;
    return GG_OK;
}

// tools only: what the pair sweep's wavefronts of the last launch's cloud 0 recorded (GG_PAIR_TIMING=1 at gg_create; tools/pair_timing.py)
extern "C" int gg_debug_pair_timing(gg_context *ctx, unsigned long long out[2048])
{
    if (!ctx || !out || !ctx->arena.pair_dbg) return GG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GG_ERR_HIP;
    if (hipMemcpy(out, ctx->arena.pair_dbg, 2048 * 8, hipMemcpyDeviceToHost) != hipSuccess) return GG_ERR_HIP;
    return GG_OK;
}

// tools only: k_reduce's work-group census by CU (GG_K2_DEBUG=5): 1024 keys x 8 counters; reset != 0 re-arms it
extern "C" int gg_debug_k2_census(gg_context *ctx, unsigned long long *out, int reset)
{
    if (!ctx || !out || ctx->arena.k2_debug != 5) return GG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GG_ERR_HIP;
    if (hipMemcpy(out, ctx->arena.k2_dbg, 1024 * 8 * 8, hipMemcpyDeviceToHost) != hipSuccess) return GG_ERR_HIP;
    if (reset) {
        std::vector<unsigned long long> z(1024 * 8, 0ull);
        for (int k = 0; k < 1024; ++k) z[(size_t)k * 8 + 4] = ~0ull; // (first start: a minimum)
        if (hipMemcpy(ctx->arena.k2_dbg, z.data(), z.size() * 8, hipMemcpyHostToDevice) != hipSuccess) return GG_ERR_HIP;
    }
    return GG_OK;
}

// tools only: k_reduce's per-work-group records (GG_K2_DEBUG=6): n x 4 words
extern "C" int gg_debug_k2_trace(gg_context *ctx, unsigned long long *out, int n)
{
    if (!ctx || !out || ctx->arena.k2_debug != 6 || n < 1 || n > 65536) return GG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GG_ERR_HIP;
    if (hipMemcpy(out, ctx->arena.k2_dbg, (size_t)n * 4 * 8, hipMemcpyDeviceToHost) != hipSuccess) return GG_ERR_HIP;
    return GG_OK;
}

// tools only: k_reduce's per-phase cycle sums (GG_K2_DEBUG=9); reset != 0 clears them
extern "C" int gg_debug_k2_phases(gg_context *ctx, unsigned long long out[64], int reset)
{
    if (!ctx || !out || ctx->arena.k2_debug != 9) return GG_ERR_INVALID;
    if (hipDeviceSynchronize() != hipSuccess) return GG_ERR_HIP;
    std::vector<unsigned long long> all((size_t)32768 * 32);
    if (hipMemcpy(all.data(), ctx->arena.k2_dbg, all.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return GG_ERR_HIP;
    for (int k = 0; k < 64; ++k) out[k] = 0;
    for (size_t wg = 0; wg < 32768; ++wg)
        for (int k = 0; k < 32; ++k) out[k] += all[wg * 32 + k];
    if (reset && hipMemset(ctx->arena.k2_dbg, 0, all.size() * 8) != hipSuccess) return GG_ERR_HIP;
    return GG_OK;
}

int gg_get_kernel_times(gg_context *ctx, double ms[GG_NUM_KERNELS], int64_t launches[GG_NUM_KERNELS], int reset)
{
    if (!ctx) return GG_ERR_INVALID;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int rc = drain_profile(ctx);
    if (rc != GG_OK) return rc;
    for (int k = 0; k < GG_NUM_KERNELS; ++k) {
        if (ms) ms[k] = ctx->k_ms[k];
        if (launches) launches[k] = ctx->k_launches[k];
        if (reset) {
            ctx->k_ms[k] = 0.0;
            ctx->k_launches[k] = 0;
        }
    }
    return GG_OK;
}

} // extern "C"
