// gg_internal.h -- device/host shared structures of libgroundgrid_hip (gfx950 only).
//
// HBM layout of one context (see DESIGN.md "Data layout"):
//   shared   : expectedPoints[C]  spiral schedule  tile rank tables
//   per slot : the nine per-call layers tile by tile (percall_index below; ground / groundpatch: gp2, gp_layout.h)
//              rec[Nmax]    (z, key)     cloud order     written by K1
//              sorted[Nmax] (z, key)     Morton-tile order, cloud order inside a tile (stable)
//              hist[NCH][T] per-wave-chunk tile histogram -> exclusive offsets after the scan
//              chunk_emit[NCH][4], totals[4], tile_start[T+1]
//              labels[Nmax], out_index[Nmax], pts16[Nmax] (host staging path)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "groundgrid_hip.h"
#include "gp_layout.h"

namespace gg {

constexpr int TILE = 16;             // cells per tile edge (K2 work-group = one tile, one thread per cell)
constexpr int TILE_CELLS = TILE * TILE;
constexpr uint32_t KEY_OUTSIDE = 0xFFFFFFFFu;
constexpr int K2_DBG_WGS = 32768;  // slots of k_reduce's phase counters (GG_K2_DEBUG=9)
// Launch geometry that changes with the batch size -- named here because docs and tests refer to it:
constexpr int PW_BIG_CONTEXT_SLOTS = 128;      // contexts with at least this many slots use 2048-point wave chunks (else 1024)
constexpr int SWEEP_LATENCY_MAX_CLOUDS = 256;  // (= CUs) k_sweep: launches of at most this many work-groups give every 64-ring group of a side
                                               // its own wavefront (up to 3 per side), and clouds are cut into as many work-groups
                                               // ("parts") as keep the launch within it; bigger launches use make_params' throughput setting
constexpr int K2_MIN_GROUPS_PER_CLOUD = 64;    // k_reduce: work-groups per cloud = max(4096 / clouds, this)
constexpr int PACKED_TILE_COUNTERS_MIN_T = 1024; // maps with more tiles keep K1's / k_scatter's per-wavefront tile counters as 16-bit halves
                                                 // (a chunk has fewer than 65536 points): 71 -> 40 KB of LDS per work-group at 3969 tiles
constexpr int K2_LIGHT_MAX = 512; // tiles with at most this many records are reduced by a single wavefront (k2_reduce.hip)
// key = tile_rank << 12 | emit << 10 | class << 8 | cell_in_tile (row_in_tile | col_in_tile << 4)
constexpr int KEY_TILE_SHIFT = 12;
constexpr int SCAN_MAX_PARTS = 16, SCAN_SYNC_WORDS = 1 + SCAN_MAX_PARTS; // (k_sort.hip k_scan_parts)
constexpr uint32_t KEY_EMIT_BIT = 1u << 10;
constexpr int KEY_CLASS_SHIFT = 8;

// device copy of gg_config plus the constants derived from it once per set_config
struct DevConfig {
    int point_count_cell_variance_threshold;
    int max_ring;
    double outlier_tolerance;
    double gpd_min_point_count_threshold;   // ground_patch_detection_minimum_point_count_threshold
    double patch_size_change_distance_sq;   // pow(patch_size_change_distance, 2.0)
    double occupied_cells_decrease_factor;
    double occupied_cells_point_count_factor;
    double occupied_cells_point_count_factor_x2; // factor * (double)2.0f
    double min_outlier_detection_ground_confidence;
    double distance_factor_sq;              // pow(distance_factor, 2.0)
    double minimum_distance_factor_sq;      // pow(minimum_distance_factor, 2.0)
    double minimum_distance_factor_x10_sq;  // pow(minimum_distance_factor*10, 2.0)
    double min_dist_fac;                    // minimum_distance_factor * 5   (:154)
    double min_point_height_thres;          // :155
    double min_point_height_obs_thres;      // :156
};

struct Geometry {
    int rows, cols;          // grid_map size
    int C;                   // rows * cols
    int tiles_r, tiles_c, T; // tile grid
    double resolution;       // (double)resolution_f
    double inv_resolution;   // 1.0 / resolution (fast path of the index division, gg_device.h)
    double length0, length1; // size * resolution
    double half0, half1;     // 0.5 * length  (getVectorToOrigin)
    float resolution_f;      // (float)map.getResolution()
    float min_dist_squared;
    int center;              // rows/2 - 1
};

// per-cloud parameters of one batched call (device array, one entry per cloud of the batch)
struct CloudParams {
    int slot;
    int n_points;
    float ox, oy, oz;
    float base_z;   // (float)translation.z
    double pos_x, pos_y;
    int has_tf;     // points are in the sensor frame: p_map = (float)(R p + t) first (src/GroundGridNodelet.cpp:166-181)
    int fresh;      // the slot's (ground, confidence) layer holds the reset values by DEFINITION -- only its never-swept border, Arena::gp_fresh_cell
                    // and what k_patch writes in this call (Arena::gp_bits) are in memory; the sweep of this call makes the layer real
    int no_confidence; // the slot's groundpatch layer is known to hold nothing above 0.01 (fresh or only scrolled since
                    // gg_reset_map): the line-of-sight test (:269 needs groundpatch(I) > 0.01f) cannot fire, K1 skips the walks
    double tf[12];  // map <- cloud frame, 3x4 row-major (R | t)
    int label_shift; // bytes added to this cloud's d_labels row (the host call places a cloud's labels right behind its n index entries -- one
                     // download for both -- and a captured launch must not bake that n in: it travels here)
    int io_index;    // the cloud's row of the batch's I/O buffers (d_points, d_labels, ...): its position in gg_batch, which is not its position
                     // in the parameter array when a batch runs as two halves (GG_FLAG_CONCURRENT_HALVES)
};

// everything a kernel needs to find its data
struct Arena {
    Geometry g;
    DevConfig cfg;
    // shared
    const float *expected;        // [C]
    const float4 *patch_table;    // [C] what detect_ground_patches needs to know about a cell that depends on the geometry and the
                                  // configuration only (rebuilt by gg_set_config, k3_patch.hip PatchCell): x = expectedPoints (:358),
                                  // y = the point-count threshold of :364-365, NEGATIVE when the cell takes 3 x 3 blocks (:334), +inf
                                  // when the quadrant loops never visit it (:325-328); z = varThresholdsq (:369); w = the cell's
                                  // element in the (ground, confidence) layer (an int, gp_layout.h)
    const uint16_t *tile_rank;    // [T] tile (tr + tc*tiles_r) -> Morton rank
    const uint16_t *rank_tile;    // [T] Morton rank -> tile
    const uint32_t *rank_cell0;   // [T] Morton rank -> first row | first col << 16 of the tile (per-point lookups: no division)
    // per slot (slot s at base + s * stride)
    float *layers;  size_t slot_layer_stride;  // the nine per-call layers of slot s, TILE BY TILE (percall_block below): layers + s*slot_layer_stride
    const uint32_t *gp_valid;     // one bit per element of the gp2 order: does gg_reset_maps fill it (the 128-byte lines that hold cells; the rest is padding)
    float2 *gp2;    size_t gp2_stride;    GpLayout gpl;              // (ground, confidence) of slot s: gp2 + s*gp2_stride, element order gp_layout.h
    // FRESH maps (gg_reset_maps did not write the interior: CloudParams::fresh).  gp_bits: per slot one bit per element of the gp2 order, bit
    // (e - 1) of the slot's words -- a wave-step of k_sweep (64 consecutive elements from 1 + 64 k) is ONE 64-bit word; zeroed by the reset,
    // set by k_patch for the cells it writes.  gp_fresh_cell: a padding element that holds the reset values (ground = odom_z, confidence
    // = 1e-7): where a fresh map's unwritten cells are read from
    unsigned long long *gp_bits;  size_t gp_bits_stride;  int gp_fresh_cell;
    int gp_bits_off, gp_bits_words;           // the bits of a slot sit behind its layer: 8-byte element gp_bits_off of the slot's gp2 region (gp_bits = gp2 + that)
    const int *gp_border;  int gp_border_n;   // the elements of the cells no sweep visits (ring >= c): k_sweep<FRESH> writes the reset's pair into them
    int fresh_launch;                         // this launch's maps are all fresh (CloudParams::fresh): k_sweep's FRESH variant
    uint2 *rec;     uint2 *sorted;  size_t point_stride;            // per slot Nmax
    float *zcell;   size_t zcell_stride;  // per slot: the KEPT heights grouped by cell (K2's stable cell sort), Nmax + 32 T + 64
    uint32_t *hist;        size_t hist_stride;   // NCH * hist_pitch
    int hist_pitch;        // words per chunk row of `hist`: T rounded up to a multiple of 4 (rows are read and written 16 bytes at a time)
    uint32_t *front_sync;  // [2 n_slots + 16] words the fused front end (k1_classify.hip) synchronises through, zeroed before every launch
                           // that uses them: arrivals per slot, "scanned" flag per slot, 8 ticket counters (one per XCD)
    uint32_t *dev_error;   // one word of host-mapped pinned memory: a kernel that gives up a bounded wait leaves a GG_DEVERR_* code
                           // here and the next gg_* call that synchronises reports GG_ERR_HIP instead of hanging
    uint32_t *chunk_emit;  size_t emit_stride;   // NCH * 4
    uint32_t *totals;      // [slot][4]  (emitted kept, emitted ignored, outliers, in-map)
    uint32_t *tile_start;  size_t tile_start_stride; // T + 1
    uint32_t *tile_live;   size_t tile_live_stride;  // [slot][T] by Morton rank: bit k = HALF COLUMN k of the tile (cells 8k .. 8k + 7 of the
                                                     // tile's cell order row + 16 col: 8 cells, one 32-byte sector per layer) physically
                                                     // HOLDS its values in the nine per-call layers.  A half column whose bit is clear
                                                     // holds stale bytes and logically has the per-call reset values (:61-75) -- the
                                                     // per-call layers are stored SPARSELY: K2 writes (and marks) exactly the half columns
                                                     // that hold an in-map record of this cloud, k_scan clears the masks of tiles without
                                                     // records, and every reader (K3's staging, gg_get_layer, the image kernels)
                                                     // substitutes the reset values through cell_is_live().  Nothing ever has to "clean"
                                                     // the cells a previous cloud left behind.  (Half columns: a scan line crosses a
                                                     // tile as an arc that touches most columns in one or two cells -- 27 % fewer layer
                                                     // bytes than with whole columns on a street scene.)
    uint4 *tile_list;      size_t tile_list_stride;  // [slot][T] K2's work lists (k_scan): light tiles from the front, dense tiles from the back; an
                                                     // entry is everything K2 needs to know about the tile without another dependent
                                                     // lookup: x = Morton rank, y / z = first / end
                                                     // of its records in `sorted`, w = first row | first col << 16
    uint32_t *tile_list_cnt;                         // [slot][2] number of light / dense tiles
    unsigned long long *scan_sync; // [n_slots][SCAN_SYNC_WORDS] k_scan as several work-groups per cloud (sort_core.h "PARTS"): ticket counter, then one
                                   // word per part; zeroed before every such launch
    uint32_t *front_sync2, *sweep_sync2; // the same two regions once more, for the half of a batch that runs on the library's side stream
    unsigned long long *scan_sync2;      // ... and the scan's (its own region: consecutive divided batches need not have equal halves)
    uint32_t *sweep_sync; // [4] ticket counter, finished work-groups, epoch of k_sweep launches with several work-groups per cloud (k4_sweep.hip)
    unsigned long long *sweep_xchg; size_t sweep_xchg_stride; // [slot] exchange region between the work-groups of one sweep (sweep_core.h "Parts"), in 64-bit words
    float *sweep_rec; size_t sweep_rec_stride; int sweep_rec_clouds; // scratch of the pair sweep (k4p_sweep_pair.hip): the records of up to
                                                                     // sweep_rec_clouds clouds of ONE launch (cloud b of the launch at sweep_rec + b * stride floats)
    unsigned long long *pair_dbg; // tools (GG_PAIR_TIMING=1 at gg_create): cycle counters of the pair sweep's wavefronts, cloud 0 of a launch; else nullptr
    int n_slots; // independent map states of the context
    int PW;    // points per wave-chunk
    int NCH;   // chunks per cloud (capacity)
    // launch geometry overrides (0 = the launchers' defaults).  Set at gg_create from the environment (GG_SWEEP_WAVES,
    // GG_K2_PER_CLOUD, GG_K2_DENSE_SHARE; GG_PW above) or per context by gg_debug_set_tuning: tools measure with them, and the
    // parity tests force every geometry the launchers can pick (tests/test_gpu_parity.py) at small batch sizes.
    int tune_sweep_waves;   // chain wavefronts per side of k_sweep
    int tune_sweep_gpw;     // ring groups per work-group of k_sweep (sweep_core.h "Parts"); default min(groups, 3)
    int tune_sweep_split;   // k_sweep "split steps": 0 = when a launch has one ring group per work-group and at most 256 work-groups, 1 = whenever gpw == 1, 2 = never
    int tune_sweep_pair;    // the pair sweeps: 0 = sweep_pair.h for launches of at most sweep_rec_clouds clouds, k_sweep for larger ones; 2 = never (k_sweep
                            // always); 4 = sweep_pairb.h (the pair sweep on the layer in place) for every launch
    int tune_sweep_pair_wgs; // work-groups per cloud of the pair sweep: 0 / 2 = one per pair of sides (two CUs), 1 = both pairs in one
    int tune_sweep_pair_waves; // chain wavefronts per pair (0 = one per 32-ring group, as many as fit)
    int tune_sweep_poll_cap; // tests: polls after which k_sweep's waits give up (0 = about a second)
    int tune_sweep_fault;   // tests: sweep::Params::debug_fault
    int tune_scan_fault;    // tests: 1 = the first part of a cloud's scan never publishes its sums (the waits of the others run out: GG_DEVERR_SCAN_WAIT)
    int tune_scan_poll_cap; // tests: polls after which a scan part's wait gives up (0 = sort_core.h SCAN_WAIT_POLLS, several seconds)
    int tune_scan_parts;    // tests: work-groups per cloud in k_scan (0 = the launcher's choice, k_sort.hip scan_parts)
    int tune_front;         // the front end (classify + tile sort): 0 = the launcher's choice, 1 = three launches (k_classify, k_scan,
                            // k_scatter), 2 = the scan inside k_classify (the last work-group of a cloud to finish scans it), 3 = one launch
                            // (after the scan every work-group scatters its own chunks)
    int tune_k2_per_cloud;  // minimum work-groups per cloud of k_reduce
    int tune_k2_dense_share; // sixteenths of them that walk the dense list
    int k2_skip;             // measurement (GG_K2_SKIP): 1 = k_reduce leaves the light tiles out, 2 = the dense tiles
    unsigned flags;
    int k2_debug;        // env GG_K2_DEBUG (measurement only, libraries built with -DGG_INSTRUMENT: gg_device.h GG_DEBUG_SWITCH): 1 = k_reduce stops after the tile lookup, 2 = after step 1, 3 = after step 3,
                         // 9 = per-phase cycle counters into k2_dbg (tools/k2_phases.py)
    unsigned long long *k2_dbg; // [64] when k2_debug == 9
    int k5_debug;        // env GG_K5_DEBUG (measurement only, results void), bits: 1 = no gathers (every lane reads element 0), 2 = no label / index stores,
                         // 4 = no `points` atomics
    int k3_debug;        // env GG_K3_DEBUG (measurement only): 1 = k_patch stops after the tile check, 2 = after staging, 3 = after the first test (:364)
    int eigen_reduction; // gg_conventions::eigen_reduction (GG_EIGEN_33 / GG_EIGEN_34_SSE): order of the 5x5 block sums in K3
};

// The nine per-call layers (everything but ground / groundpatch) are stored tile by tile: the 16x16 tile of Morton rank r owns the
// block [r * PERCALL_BLOCK, (r + 1) * PERCALL_BLOCK) of its slot, laid out [layer position][column in tile][row in tile] -- 9 x 256
// floats, 9 KiB, every (layer, column) a 64-byte line segment.  K2 writes a tile's nine layers as ONE contiguous region instead of
// 144 segments 1456 bytes apart in nine planes (measured: k_reduce 1.47 -> 1.31 ms per 1024 clouds with the same bytes), K5
// addresses a point's cell straight from its key (tile rank, cell in tile) without the tile's origin, and K3's block of 8 columns
// is 512 contiguous bytes per layer.  The order of the layers inside a block puts the three K3 reads first and the three that
// GG_FLAG_MINIMAL_LAYERS leaves out last.  The dense column-major matrices of the reference exist at the host boundary only
// (gg_get_layer / gg_set_layer, k6_wire.hip).
constexpr int PERCALL_LAYERS = 9;
constexpr int PERCALL_BLOCK = PERCALL_LAYERS * TILE * TILE;
enum : int { PL_POINTS = 0, PL_VARIANCE = 1, PL_MINGROUNDHEIGHT = 2, PL_M2 = 3, PL_POINTSRAW = 4, PL_MEANVARIANCE = 5, PL_MAXGROUNDHEIGHT = 6,
             PL_GROUNDCANDIDATES = 7, PL_PLANEDIST = 8 };
__host__ __device__ inline int percall_position(int layer) // gg_layer -> position in a block (-1: ground / groundpatch live elsewhere)
{
    switch (layer) {
    case GG_LAYER_POINTS: return PL_POINTS;
    case GG_LAYER_VARIANCE: return PL_VARIANCE;
    case GG_LAYER_MINGROUNDHEIGHT: return PL_MINGROUNDHEIGHT;
    case GG_LAYER_M2: return PL_M2;
    case GG_LAYER_POINTSRAW: return PL_POINTSRAW;
    case GG_LAYER_MEANVARIANCE: return PL_MEANVARIANCE;
    case GG_LAYER_MAXGROUNDHEIGHT: return PL_MAXGROUNDHEIGHT;
    case GG_LAYER_GROUNDCANDIDATES: return PL_GROUNDCANDIDATES;
    case GG_LAYER_PLANEDIST: return PL_PLANEDIST;
    default: return -1;
    }
}
__host__ __device__ inline float *percall_ptr(const Arena &a, int slot) { return a.layers + (size_t)slot * a.slot_layer_stride; }
// element of (tile rank, layer position, cell in tile = row in tile + 16 * column in tile)
__host__ __device__ inline size_t percall_index(int rank, int position, int cell) { return (size_t)rank * PERCALL_BLOCK + (size_t)position * (TILE * TILE) + (size_t)cell; }
// ... of map cell (row, col)
__device__ inline size_t percall_index_of(const Arena &a, int position, int row, int col)
{
    return percall_index(a.tile_rank[(row / TILE) + (col / TILE) * a.g.tiles_r], position, (row % TILE) + (col % TILE) * TILE);
}

// `ground` and `groundpatch` (confidence) are always used together -- confidence-weighted height -- and are the only
// state that persists from cloud to cloud.  They are stored INTERLEAVED as float2 (x = ground, y = confidence), one 8-byte
// request instead of two 4-byte ones everywhere, in the sheared element order of gp_layout.h (the terrain sweep's
// wavefronts then read and write 512 contiguous bytes per access); the plane slots GG_LAYER_GROUND / GG_LAYER_GROUNDPATCH
// of `layers` are unused.  gg_get_layer / gg_set_layer convert at the host boundary.
// the per-call reset value of a layer (:61-75, :147): what a cell outside the live columns logically holds
__host__ __device__ inline float layer_reset_value(int layer)
{
    return layer == GG_LAYER_MINGROUNDHEIGHT ? 3.402823466e+38f /* FLT_MAX, :72 */ : layer == GG_LAYER_MAXGROUNDHEIGHT ? 1.175494351e-38f /* FLT_MIN (sic), :73 */ : 0.0f;
}
// the bit of tile_live that covers `cell` = row in tile + 16 * column in tile
__host__ __device__ inline int live_bit(int cell) { return cell >> 3; }
// does cell (row, col) of `slot` physically hold its per-call layer values (Arena::tile_live)?
__device__ inline bool cell_is_live(const Arena &a, int slot, int row, int col)
{
    const int rank = a.tile_rank[(row / TILE) + (col / TILE) * a.g.tiles_r];
    return ((a.tile_live[(size_t)slot * a.tile_live_stride + rank] >> live_bit((row % TILE) + (col % TILE) * TILE)) & 1u) != 0u;
}

__host__ __device__ inline float2 *gp2_ptr(const Arena &a, int slot) { return a.gp2 + (size_t)slot * a.gp2_stride; }
__host__ __device__ inline int gp_idx(const Arena &a, int row, int col) { return gp_index(a.gpl, row, col); }

// I/O pointers of one batched call
struct BatchIO {
    const void *d_points;
    size_t cloud_stride;
    int point_format;
    uint8_t *d_labels;
    int32_t *d_out_index;
    gg_point32 *d_out_clouds;
    int32_t *d_out_counts;
    uint8_t *d_label_masks;
    uint8_t *d_out_pc2;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device setting: the launchers that need more than 64 KiB of dynamic
// LDS opt in once per DEVICE (a process may hold contexts on several GPUs).  `opt_in` runs under a lock and the device is marked
// only after it returned, so a second thread launching on the same device either sees the mark (the attribute is set) or waits for
// the lock; devices beyond the 64 the mask has bits for opt in on every launch (idempotent).
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    std::mutex lock;
    template <class F> void run(F opt_in)
    {
        int dev = 0;
        (void)hipGetDevice(&dev);
        const bool tracked = dev >= 0 && dev < 64;
        const uint64_t bit = tracked ? 1ull << dev : 0ull;
        if (tracked && (done.load(std::memory_order_acquire) & bit)) return;
        std::lock_guard<std::mutex> g(lock);
        if (tracked && (done.load(std::memory_order_relaxed) & bit)) return;
        opt_in();
        if (tracked) done.fetch_or(bit, std::memory_order_release);
    }
};

// codes a kernel leaves in Arena::dev_error when it gives up a bounded wait
enum : uint32_t { GG_DEVERR_NONE = 0, GG_DEVERR_FRONT_WAIT = 1 /* k_classify: a cloud's scan never completed */, GG_DEVERR_SWEEP_WAIT = 2 /* k_sweep: a hand-over never arrived */,
                  GG_DEVERR_SCAN_WAIT = 3 /* k_scan in parts: an earlier part's sums never arrived */ };
// the front end's launch shapes (Arena::tune_front)
enum : int { FRONT_AUTO = 0, FRONT_THREE_LAUNCHES = 1, FRONT_SCAN_IN_CLASSIFY = 2, FRONT_ONE_LAUNCH = 3 };
constexpr int FRONT_DEFAULT_SHAPE = FRONT_THREE_LAUNCHES;

// kernel launchers (one per .hip file)
int launch_classify(const Arena &a, const CloudParams *d_params, const BatchIO &io, int n_clouds, int max_n, hipStream_t s); // returns the FRONT_* shape it
                                                                                                                               // ran: the caller adds launch_scan / launch_scatter for what is left
void launch_scan(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s);
void launch_scatter(const Arena &a, const CloudParams *d_params, int n_clouds, int max_n, hipStream_t s);
void launch_reduce(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s);
void launch_stage_insert(const Arena &a, const CloudParams *d_params, hipStream_t s); // gg_insert_cloud: :282-309 continued from the layers as they stand (one slot, dense layers)
void launch_reduce_lazy(const Arena &a, const CloudParams &cp, hipStream_t s); // (GG_FLAG_MINIMAL_LAYERS: the other three layers, one slot)
void launch_patch(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s);
void launch_patch_stage(const Arena &a, const CloudParams *d_params, int slot, int section, hipStream_t s); // gg_run_stage: :323 + one quadrant (-1: all) on the slot's layers as they stand
void launch_stage_cell(const Arena &a, int slot, int stage, int i, int j, hipStream_t s);                    // gg_run_stage: detect_ground_patch<S> / interpolate_cell of one cell
namespace sweep {
struct Params;
Params make_params(int n, double resolution, float min_dist_squared, double decrease); // sweep_emul.hip (host)
} // namespace sweep
void launch_sweep(const Arena &a, const sweep::Params &P, const CloudParams *d_params, int n_clouds, hipStream_t s,
                  unsigned long long *dbg = nullptr); // k4_sweep.hip; dbg: 16 x 4 cycle counters of cloud 0's wavefronts (tools)
bool launch_sweep_pair(const Arena &a, const sweep::Params &P, const CloudParams *d_params, int n_clouds, hipStream_t s); // k4p_sweep_pair.hip; false: not this launch
bool sweep_takes_fresh(const Arena &a, const sweep::Params &P, int n_clouds); // k4_sweep.hip: would launch_sweep run k_sweep without split steps (what a launch of fresh maps needs)?
bool launch_sweep_pair_batch(const Arena &a, const sweep::Params &P, const CloudParams *d_params, int n_clouds, hipStream_t s); // k4b_sweep_pair_batch.hip; false: not this launch
size_t sweep_pair_rec_floats(const sweep::Params &P); // scratch floats per cloud of a launch (0: the geometry cannot take the pair sweep)
constexpr int SWEEP_PAIR_MAX_CLOUDS = 16;              // launches of more clouds keep k_sweep
size_t sweep_lds_bytes(const sweep::Params &P);
size_t sweep_xchg_entries(const sweep::Params &P);
void launch_label(const Arena &a, const CloudParams *d_params, const BatchIO &io, int n_clouds, int max_n, hipStream_t s);
void launch_fill(float *dst, size_t n, float v, hipStream_t s);
void launch_fill_bytes(uint8_t *dst, size_t n, uint8_t v, hipStream_t s);
void launch_fill_strided(float *dst, size_t n, size_t stride, int count, float v, hipStream_t s);   // count regions of n floats, `stride` apart
void launch_fill_percall(const Arena &a, int first_slot, int n_slots, const float init[GG_NUM_LAYERS], hipStream_t s); // every per-call layer of the slots := init[layer]
void launch_layer_insert(const Arena &a, int slot, int layer, const float *src, hipStream_t s);    // dense column-major plane -> per-call layer (all columns live)
void launch_fill2_strided(float2 *dst, size_t n, size_t stride, int count, float x, float y, const uint32_t *valid, hipStream_t s);
void launch_fill2(float2 *dst, size_t n, float x, float y, hipStream_t s);
void launch_reset_fresh(const Arena &a, int first_slot, int count, float x, float y, hipStream_t s, int all_ones = 0); // k1_classify.hip k_reset_fresh
void launch_plane_extract(const Arena &a, int slot, int comp, float *dst, hipStream_t s); // sheared layer -> column-major plane
void launch_plane_insert(const Arena &a, int slot, int comp, const float *src, hipStream_t s);
void launch_layer_extract(const Arena &a, int slot, int layer, float *dst, hipStream_t s);
void launch_layers_extract(const Arena &a, int slot, unsigned want, float *dst, size_t plane_floats, hipStream_t s); // (all requested layers, one launch) // per-call layer -> dense column-major plane (reset values outside the live columns)
void launch_materialise_layers(const Arena &a, int slot, hipStream_t s);                  // write the reset values into every dead column of the slot's per-call layers, mark all live
void launch_layer_to_u8(const float *layer, int rows, int cols, float *d_bounds, uint8_t *d_img, hipStream_t s);
void launch_terrain_image(const Arena &a, int slot, float *d_img, hipStream_t s);
void launch_scroll(const Arena &a, int slot, float2 *scratch, int s0, int s1, double pos_x, double pos_y, const double plane[4], hipStream_t s);
void launch_pack16(const gg_point32 *src, gg_point16 *dst, size_t n, hipStream_t s);
void launch_decode_classes(const Arena &a, int slot, size_t n, uint8_t *d_class, int32_t *d_cell, hipStream_t s);

} // namespace gg
