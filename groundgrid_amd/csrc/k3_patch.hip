// K3 -- detect_ground_patches / detect_ground_patch<3|5> (src/GroundSegmentation.cpp:314-395).
//
// A pure read-neighbours / write-self stencil: each interior cell reads an SxS block (S = 3 inside
// patch_size_change_distance, else 5) of `points`, `variance`, `minGroundHeight` and updates only its own
// `ground` / `groundpatch`, so the reference's four quadrant threads are order-free and one thread per
// cell is exact.  Blocks of 32 rows x 8 cols of cells (rows are the contiguous dimension of the
// column-major layers), inputs staged in LDS with a 2-cell halo; one work-group walks a band of blocks (k_patch).  The block sums use Eigen's unrolled
// tree order (gg_device.h tree9/tree25).
//
// Algorithmic bytes per cell: 6 layers read (points, variance, min, ground, groundpatch, expectedPoints),
// 2 written.
#include "gg_device.h"

#include <float.h>

#include <algorithm>

namespace gg {

constexpr int PR = 32, PC = 8, HALO = 2;
constexpr int LR = PR + 2 * HALO, LC = PC + 2 * HALO; // 36 x 12

// What a cell's visit has found out from the staged layers, carried to the next block's turn: the cell's old (ground,
// confidence) is loaded from HBM meanwhile, so that no work-group waits for that gather (nor for `expectedPoints`, which is
// requested a block ahead together with the layers).
struct PatchCarry {
    bool live;          // the block sum passed :364-365
    int gidx;           // element of the cell in the (ground, confidence) layer
    float2 old;         // (ground, confidence), in flight
    float pointsblockSum, expected, localmin, maxVar, groundlevel, varThresholdsq;
    int S;
};

// Eigen's block-sum orders over elements delivered one at a time, in column-major order s = 0 .. S*S-1, in groups separated by
// a compiler fence: the LDS reads of a group are issued, waited for and consumed before the next group's -- at most a dozen
// values are live at any time.  (With the 25 values of each of the three blocks of a cell read up front, the kernel needed
// 150 VGPRs: three wavefronts per SIMD for a kernel that lives on hiding LDS and memory latency.)
GG_DEV void group_fence() { __asm__ volatile("" ::: "memory"); }

template <class F> GG_DEV float stream_tree9(F get)
{
    const float x0 = get(0), x1 = get(1), x2 = get(2), x3 = get(3);
    const float l = (x0 + x1) + (x2 + x3);
    group_fence();
    const float x4 = get(4), x5 = get(5), x6 = get(6), x7 = get(7), x8 = get(8);
    return l + ((x4 + x5) + (x6 + (x7 + x8)));
}
template <class F> GG_DEV float stream_six(F get, int s0)
{
    const float x0 = get(s0), x1 = get(s0 + 1), x2 = get(s0 + 2), x3 = get(s0 + 3), x4 = get(s0 + 4), x5 = get(s0 + 5);
    return (x0 + (x1 + x2)) + (x3 + (x4 + x5));
}
template <class F> GG_DEV float stream_tree25(F get)
{
    const float a = stream_six(get, 0);
    group_fence();
    const float b = stream_six(get, 6);
    group_fence();
    const float ab = a + b;
    const float c = stream_six(get, 12);
    group_fence();
    const float x18 = get(18), x19 = get(19), x20 = get(20), x21 = get(21), x22 = get(22), x23 = get(23), x24 = get(24);
    const float d = (x18 + (x19 + x20)) + ((x21 + x22) + (x23 + x24));
    return ab + (c + d);
}
template <class F> GG_DEV float stream_tree25_eigen34(F get)
{
    float p[4], tail[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = get(5 * j + r);
            p[r] = j == 0 ? x : p[r] + x;
        }
        tail[j] = get(5 * j + 4);
        group_fence();
    }
    float res = (p[0] + p[2]) + (p[1] + p[3]);
#pragma unroll
    for (int j = 0; j < 5; ++j) res = res + tail[j];
    return res;
}

// first half: the two weighted sums (:372-375) of a cell that passed the point-count test of :364-365 (k_patch computes the
// count itself, as a box filter).  Everything that needs the LDS window.
template <int S>
GG_DEV void patch_sums(const Arena &a, const float (*pts)[LR], const float (*var)[LR], const float (*mnl)[LR], int lr, int lc, PatchCarry &pc)
{
    constexpr int ci = S / 2;
    const DevConfig &cfg = a.cfg;
    const bool e34 = a.eigen_reduction == GG_EIGEN_34_SSE;
    auto block_sum = [&](auto get) { return (S == 3) ? stream_tree9(get) : e34 ? stream_tree25_eigen34(get) : stream_tree25(get); };
    auto P = [&](int s) { return pts[lc - ci + s / S][lr - ci + s % S]; };
    const float pointsblockSum = pc.pointsblockSum;
    // :374
    if (pts[lc][lr] >= (float)cfg.point_count_cell_variance_threshold)
        pc.maxVar = var[lc][lr]; // :372
    else
        pc.maxVar = block_sum([&](int s) { return P(s) * var[lc - ci + s / S][lr - ci + s % S]; }) / pointsblockSum;
    group_fence();
    // :373 minCoeff (this layer never holds NaN: v_min) and :375
    float localmin = FLT_MAX;
    pc.groundlevel = block_sum([&](int s) {
                         const float m = mnl[lc - ci + s / S][lr - ci + s % S];
                         localmin = fminf(localmin, m);
                         return P(s) * m;
                     }) /
                     pointsblockSum;
    pc.localmin = localmin;
}

// second half (:360-393): the decision against the old cell, one block later
// bits: the fresh map's written-cell bits (Arena::gp_bits of the slot), or null
GG_DEV void detect_ground_patch_b(const Arena &a, const PatchCarry &pc, float2 *gp2, unsigned long long *bits)
{
    if (!pc.live) return;
    const DevConfig &cfg = a.cfg;
    const float oldConfidence = pc.old.y;   // :360
    const float oldGroundheight = pc.old.x; // :361
    const float pointsblockSum = pc.pointsblockSum, groundlevel = pc.groundlevel, maxVar = pc.maxVar;
    const float varThresholdsq = pc.varThresholdsq; // :369, from the per-cell table (gg_internal.h Arena::patch_table)
    // :376
    const float groundDiff = std_max((groundlevel - oldGroundheight) * (2.0f * oldConfidence), 1.0f);
    // :379-380
    if ((double)oldConfidence > 0.5 && (double)groundlevel >= (double)oldGroundheight + cfg.outlier_tolerance) return;
    // :382
    if ((double)varThresholdsq > (double)maxVar * (double)maxVar && maxVar > 0.0f &&
        (double)pointsblockSum > (double)((groundDiff * pc.expected) * (float)pc.S) * cfg.gpd_min_point_count_threshold) {
        const float newConfidence = (float)std_min((double)pointsblockSum / cfg.occupied_cells_point_count_factor, 1.0); // :383
        const float G = (groundlevel * newConfidence + (oldConfidence * oldGroundheight) * 2.0f) / (newConfidence + oldConfidence * 2.0f); // :385
        const float Cf =
            (float)std_min(((double)pointsblockSum / cfg.occupied_cells_point_count_factor_x2 + (double)oldConfidence) / 2.0, 1.0); // :387
        gp2[pc.gidx] = make_float2(G, Cf);
    } else if (pc.localmin < oldGroundheight) { // :389
        gp2[pc.gidx] = make_float2(pc.localmin, std_min(oldConfidence + 0.1f, 0.5f)); // :391, :393
    } else
        return;
    if (bits) { // (the centre cell, element 0, is the sweep's own: :405-411 overwrite it)
        const unsigned e = (unsigned)pc.gidx - 1u;
        if (pc.gidx > 0) __hip_atomic_fetch_or(bits + (e >> 6), 1ull << (e & 63u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// One work-group per (cloud, band of PR rows); it walks the band's blocks of PC columns from left to right with a rolling
// window of LC = PC + 4 columns in LDS (a ring of 32 column slots; slots 0..3 are mirrored at 32..35 so that every
// window is 12 CONSECUTIVE slots), keeps the next block's columns in flight in registers while it computes the current
// one, and skips the blocks that no KEPT point can reach: a cell only changes if its block of `points` sums to at least 3
// (:364), and `points` is zero in every 16x16 tile without records (K2 reset it, :61-75) -- the record counts of the band's
// three tile rows are folded into one flag per tile column when the work-group starts.  (540 blocks per cloud as separate
// work-groups spent 0.6 ms of 1.6 on launching work-groups and another 0.4 on their first loads.)
constexpr int RING = 32, SLOTS = RING + 4, MAXTC = 256;

// STAGE: the kernel as the stand-alone stage detect_ground_patches(map, section) (gg_run_stage, include/groundgrid/GroundSegmentation.h:59):
// the layers are whatever the slot holds -- not what k_reduce left a moment ago -- so no block is skipped on the strength of the last
// cloud's record counts, the S x S point count of :359 is summed in Eigen's order like the two weighted sums (a host-written `points`
// layer need not hold integers), and only the cells of the quadrant `bounds` = {i_lo, i_hi, j_lo, j_hi} (:325-328) are visited.
template <bool STAGE>
__global__ __launch_bounds__(256) void k_patch(const Arena a, const CloudParams *__restrict__ params, int n_bands, int blocks_per_segment, const int4 bounds)
{
    __shared__ __attribute__((aligned(16))) float pts[SLOTS][LR], var[SLOTS][LR], mnl[SLOTS][LR]; // (LR * 4 bytes = 9 x 16: every row quad is 16-byte aligned)
    // vertical partial sums of `points` over the window's 12 columns and the block's 32 rows: 5 rows (v5) and the middle 3 (v3).
    // `points` holds COUNTS -- integer-valued floats far below 2^24 -- so the S x S block sum of :359 is an exact integer in ANY
    // order of addition: the one float sum of the path that may be taken as a separable box filter (5 + 5 taps instead of 25,
    // the vertical half shared by the cells of a column) and still equals Eigen's tree sum bit for bit.  It is the sum EVERY
    // visited cell pays for (:364 decides on it); the two weighted sums (:374-375) keep Eigen's order.
    __shared__ float v5[LC][PR], v3[LC][PR];
    __shared__ uint32_t col_has_points[MAXTC]; // per tile column: records in the band's tile rows
    __shared__ uint32_t live_cols[3][MAXTC];   // the band's (up to three) tile rows: tile_live (which half columns physically hold values)
    __shared__ uint16_t band_rank[3][MAXTC];   // ... and the tiles' Morton ranks (where their blocks of the per-call layers are)
    // XCD-aware (gg_device.h): the bands of one cloud run on one XCD.  A launch with few clouds cuts every band into segments
    // of blocks_per_segment blocks (one work-group each) so that the chip is still covered.
    const uint32_t item = xcd_contiguous_item(blockIdx.x + gridDim.x * blockIdx.y, gridDim.x * gridDim.y);
    const int cloud = (int)(item / gridDim.x), band = (int)(item % gridDim.x) % n_bands, segment = (int)(item % gridDim.x) / n_bands;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows, cols = a.g.cols;
    const int r0 = HALO + band * PR; // first output row of the band
    const int b_first = segment * blocks_per_segment;
    const int n_blocks = min((cols - 2 * HALO + PC - 1) / PC, b_first + blocks_per_segment); // (end of this work-group's blocks)
    const int tid = threadIdx.x;

    // ---- which tile columns hold records in the band's tile rows
    const int tiles_c = a.g.tiles_c;
    for (int k = tid; k < tiles_c; k += 256) col_has_points[k] = 0u;
    __syncthreads();
    {
        const uint32_t *tile_start = a.tile_start + (size_t)cp.slot * a.tile_start_stride;
        const int tr_lo = (r0 - HALO) / TILE, tr_hi = min((r0 + PR + HALO - 1) / TILE, a.g.tiles_r - 1);
        const int ntr = tr_hi - tr_lo + 1;
        const uint32_t *tile_live = a.tile_live + (size_t)cp.slot * a.tile_live_stride;
        for (int k = tid; k < 3 * tiles_c; k += 256) live_cols[k / tiles_c][k % tiles_c] = 0u;
        __syncthreads();
        for (int k = tid; k < ntr * tiles_c; k += 256) {
            const int tc = k / ntr, tile = (tr_lo + k % ntr) + tc * a.g.tiles_r;
            const int rank = a.tile_rank[tile];
            if (STAGE || tile_start[rank + 1] != tile_start[rank]) col_has_points[tc] = 1u;
            band_rank[k % ntr][tc] = (uint16_t)rank;
            live_cols[k % ntr][tc] = tile_live[rank]; // (PR + 2 HALO rows starting at a multiple of TILE: at most three tile rows)
        }
    }
    __syncthreads();
    if (GG_DEBUG_SWITCH(a, k3_debug) == 1) return;
    auto block_has_points = [&](int b) { // reads cover columns [PC b, PC b + LC)
        const int tc_lo = (PC * b) / TILE, tc_hi = min((PC * b + LC - 1) / TILE, tiles_c - 1);
        return (col_has_points[tc_lo] | col_has_points[tc_hi]) != 0u; // (LC <= TILE: at most two tile columns)
    };
    auto next_block = [&](int b) {
        while (b < n_blocks && !block_has_points(b)) ++b;
        return b;
    };

    // (the per-call layers are stored tile by tile, gg_internal.h percall_index: K3's three come first in a tile's block)
    const float *L = percall_ptr(a, cp.slot);
    const float *gp_pts = L + percall_index(0, PL_POINTS, 0);
    const float *gp_var = L + percall_index(0, PL_VARIANCE, 0);
    const float *gp_min = L + percall_index(0, PL_MINGROUNDHEIGHT, 0);
    float2 *gp2 = gp2_ptr(a, cp.slot);
    // a FRESH map: every cell's old (ground, confidence) is the reset's pair -- read from the one padding element that holds it --, and the
    // cells written here are marked for the sweep, which reads nothing else of the layer
    const bool fresh = !STAGE && (cp.fresh != 0 || GG_DEBUG_SWITCH(a, k3_debug) == 7);
    unsigned long long *bits = fresh ? a.gp_bits + (size_t)cp.slot * a.gp_bits_stride : nullptr;

    // staging registers: LC columns x LR rows = 12 x 9 QUADS of four rows, the three layers of one quad per thread (threads
    // 0 .. 107): a tile's block of a layer holds a column's 16 rows contiguously and the window starts at a tile row, so a quad
    // is 16 aligned bytes of one half column -- one 16-byte load costs the CU's vector-memory front end what a 4-byte one does
    // (tools/ubench/ta_lines.hip), and the liveness of a half column (8 rows) covers the whole quad
    constexpr int QR = LR / 4; // quads per window column
    static_assert(LR % 4 == 0 && TILE % 4 == 0, "row quads must not straddle half columns");
    float4 sp4 = make_float4(0.f, 0.f, 0.f, 0.f), sv4 = sp4, sm4 = sp4;
    float4 cell_next = make_float4(0.0f, __builtin_inff(), 0.0f, 0.0f); // the per-cell constants (Arena::patch_table) of this thread's cell in the requested block
    const int tr = tid % PR, tcl = tid / PR;
    const int i = r0 + tr;
    const int q_lc = tid / QR, q_lr = 4 * (tid % QR); // this thread's quad: window column, first window row
    // Every load of the walk is UNCONDITIONAL (lanes with nothing to fetch read element 0, one broadcast line): loads and
    // stores share one in-order counter, and the compiler can only leave younger loads in flight across a wait when it knows how
    // many there are.  With a load under a branch every wait became vmcnt(0): the block's sums waited for the NEXT block's
    // columns, requested a moment earlier.
    auto request = [&](int b, int first_col, int n_cols) { // columns [first_col, first_col + n_cols) of block b's window
        {
            const int jj = HALO + PC * b + tcl;
            const bool v = i < rows && jj < cols && (!STAGE || (i >= bounds.x && i < bounds.y && jj >= bounds.z && jj < bounds.w));
            const float4 e = a.patch_table[v ? (size_t)i + (size_t)jj * rows : (size_t)0]; // :358, :334, :364, :369 and the layer element
            cell_next = make_float4(e.x, v ? e.y : __builtin_inff(), e.z, e.w); // (no such cell: never visited)
        }
        const int gr = r0 - HALO + q_lr, gcol = first_col + q_lc;
        const bool ok = q_lc < n_cols && gr < rows && gcol < cols;
        // the per-call layers are sparse (gg_internal.h tile_live): a half column that holds no record of this cloud has stale
        // bytes and logically the reset values of :61-75 -- points 0, variance 0 / (0 + FLT_MIN) = 0, minGroundHeight FLT_MAX.
        // Its lanes fetch element 0 like the out-of-range ones (still unconditional loads): no HBM traffic for dead half columns
        const int btr = ok ? q_lr / TILE : 0, btc = ok ? gcol / TILE : 0;
        const int cell_in_tile = (gr % TILE) + (gcol % TILE) * TILE;
        const bool live = ok && ((live_cols[btr][btc] >> live_bit(cell_in_tile)) & 1u) != 0u;
        const size_t idx = live ? percall_index((int)band_rank[btr][btc], 0, cell_in_tile) : (size_t)0;
        const float4 p = *reinterpret_cast<const float4 *>(gp_pts + idx), v = *reinterpret_cast<const float4 *>(gp_var + idx),
                     m = *reinterpret_cast<const float4 *>(gp_min + idx);
        // (rows of a border tile beyond the map's last row exist in the block and hold nothing: they count as outside)
        const bool l0 = live, l1 = live && gr + 1 < rows, l2 = live && gr + 2 < rows, l3 = live && gr + 3 < rows;
        const bool o0 = ok, o1 = ok && gr + 1 < rows, o2 = ok && gr + 2 < rows, o3 = ok && gr + 3 < rows;
        sp4 = make_float4(l0 ? p.x : 0.0f, l1 ? p.y : 0.0f, l2 ? p.z : 0.0f, l3 ? p.w : 0.0f);
        sv4 = make_float4(l0 ? v.x : 0.0f, l1 ? v.y : 0.0f, l2 ? v.z : 0.0f, l3 ? v.w : 0.0f);
        sm4 = make_float4(l0 ? m.x : (o0 ? FLT_MAX : 0.0f), l1 ? m.y : (o1 ? FLT_MAX : 0.0f), l2 ? m.z : (o2 ? FLT_MAX : 0.0f), l3 ? m.w : (o3 ? FLT_MAX : 0.0f));
    };
    auto deposit = [&](int first_col, int n_cols) {
        if (q_lc < n_cols) { // (threads beyond the window's quads hold nothing)
            const int slot = (first_col + q_lc) & (RING - 1);
            *reinterpret_cast<float4 *>(&pts[slot][q_lr]) = sp4;
            *reinterpret_cast<float4 *>(&var[slot][q_lr]) = sv4;
            *reinterpret_cast<float4 *>(&mnl[slot][q_lr]) = sm4;
            if (slot < SLOTS - RING) { // the mirror
                *reinterpret_cast<float4 *>(&pts[slot + RING][q_lr]) = sp4;
                *reinterpret_cast<float4 *>(&var[slot + RING][q_lr]) = sv4;
                *reinterpret_cast<float4 *>(&mnl[slot + RING][q_lr]) = sm4;
            }
        }
    };

    int b = next_block(b_first);
    int req_first = PC * b, req_cols = LC; // what the staging registers hold
    if (b < n_blocks) request(b, req_first, req_cols);
    // One block: `produce` receives what this block's cells found out (and their old cell, in flight), `consume` is the block
    // before, decided now.  Two carries take turns (the loop below is unrolled by two) so that the old cell's load lands in the
    // register it is consumed from one block later: a copy at the loop's back edge would wait for the load where it is issued.
    // Vector-memory schedule of a block: 7 loads (the next block's columns and expectedPoints), the stores of the block before
    // (conditional), 1 load (this block's old cells) -- the next wait for the columns leaves that one load in flight.
    auto block_step = [&](PatchCarry &produce, PatchCarry &consume) {
        deposit(req_first, req_cols);
        const float4 cell_const = cell_next;
        // the next block that can change anything: its new columns travel while this one is computed (past the last block: a
        // request that fetches nothing); requested as soon as the staging registers are free, before the barrier
        const int nb = next_block(b + 1);
        req_first = nb == b + 1 ? PC * nb + (LC - PC) : PC * nb; // (a neighbour: only the PC columns beyond this window)
        req_cols = nb == b + 1 ? PC : LC;
        request(nb, req_first, req_cols);
        __syncthreads();
        detect_ground_patch_b(a, consume, gp2, bits); // the previous block's cell: its old (ground, confidence) has arrived meanwhile
        consume.live = false;

        const int base = (PC * b) & (RING - 1); // 0, 8, 16 or 24: the window is slots base .. base + LC - 1
        const int lr = tr + HALO, lc = tcl + HALO;
        // :332-334 and the right side of :364-365 come from the per-cell table: the threshold's sign says 3 x 3 or 5 x 5 blocks,
        // +inf that the quadrant loops (:325-328) never visit the cell
        const bool near = cell_const.y < 0.0f;
        const float threshold = fabsf(cell_const.y);
        // :359 the block's point count from the vertical partial sums (exact: integers, see v5 / v3)
        if (tid < LC * (PR / 2)) { // two vertically adjacent outputs per thread: six reads instead of ten (192 threads, one pass)
            const int c = tid / (PR / 2), r = 2 * (tid % (PR / 2));
            const float *col = pts[base + c] + r; // rows r .. r + 4 of the window = output row r - 2 .. r + 2
            const float x0 = col[0], x1 = col[1], x2 = col[2], x3 = col[3], x4 = col[4], x5 = col[5];
            const float mid0 = (x1 + x2) + x3, mid1 = (x2 + x3) + x4;
            v3[c][r] = mid0;
            v5[c][r] = (x0 + x4) + mid0;
            v3[c][r + 1] = mid1;
            v5[c][r + 1] = (x1 + x5) + mid1;
        }
        __syncthreads();
        float pointsblockSum = near ? (v3[tcl + 1][tr] + v3[tcl + 2][tr]) + v3[tcl + 3][tr]
                                    : ((v5[tcl][tr] + v5[tcl + 1][tr]) + (v5[tcl + 2][tr] + v5[tcl + 3][tr])) + v5[tcl + 4][tr];
        if (STAGE) { // arbitrary layer contents: :359 in Eigen's order
            const float(*w)[LR] = pts + base;
            if (near)
                pointsblockSum = stream_tree9([&](int s) { return w[lc - 1 + s / 3][lr - 1 + s % 3]; });
            else if (a.eigen_reduction == GG_EIGEN_34_SSE)
                pointsblockSum = stream_tree25_eigen34([&](int s) { return w[lc - 2 + s / 5][lr - 2 + s % 5]; });
            else
                pointsblockSum = stream_tree25([&](int s) { return w[lc - 2 + s / 5][lr - 2 + s % 5]; });
        }
        const int S = near ? 3 : 5;
        // :364-365 (count and threshold are integer-valued floats: the comparison is the reference's binary64 one)
        const bool pass = !(pointsblockSum < threshold) && GG_DEBUG_SWITCH(a, k3_debug) != 3;
        produce.gidx = pass ? __float_as_int(cell_const.w) : 0; // the (ground, confidence) layer has its own element order (gp_layout.h)
        produce.old = gp2[(cp.fresh != 0 && !STAGE) ? a.gp_fresh_cell : produce.gidx]; // :360-361, used one block later
        produce.live = pass;
        produce.S = S;
        produce.pointsblockSum = pointsblockSum;
        produce.expected = cell_const.x;
        produce.varThresholdsq = cell_const.z;
        group_fence();
        if (pass) {
            if (near)
                patch_sums<3>(a, pts + base, var + base, mnl + base, lr, lc, produce);
            else
                patch_sums<5>(a, pts + base, var + base, mnl + base, lr, lc, produce);
        }
        // The next block's deposit overwrites column slots.  When it is the neighbour, its 8 new columns go to the slots BEHIND this
        // window (the ring holds 32: 12 of this window + 8 new ones never meet), so wavefronts without cells in the weighted sums go
        // on -- deposit, request the block after -- while the others finish; v5 / v3 are rewritten only after the next step's
        // first barrier, which everybody reaches after its reads here.  A block further away lands on any slots: barrier.
        if (nb != b + 1) __syncthreads();
        b = nb;
    };
    PatchCarry c0, c1;
    c0.live = c1.live = false;
    while (b < n_blocks) {
        if (GG_DEBUG_SWITCH(a, k3_debug) == 2) return;
        block_step(c0, c1);
        if (b >= n_blocks) break;
        block_step(c1, c0);
    }
    detect_ground_patch_b(a, c0, gp2, bits);
    detect_ground_patch_b(a, c1, gp2, bits);
}

void launch_patch(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    const int ir = a.g.rows - 2 * HALO, ic = a.g.cols - 2 * HALO;
    if (ir <= 0 || ic <= 0 || a.g.tiles_c > MAXTC) return;
    const int n_bands = (ir + PR - 1) / PR, n_blocks = (ic + PC - 1) / PC;
    // about 8192 work-groups per launch: whole bands when there are many clouds, segments of a band when there are few
    const int segments = std::max(1, std::min(n_blocks, 8192 / std::max(1, n_clouds * n_bands)));
    const int per_segment = (n_blocks + segments - 1) / segments;
    dim3 grid(n_bands * ((n_blocks + per_segment - 1) / per_segment), n_clouds);
    hipLaunchKernelGGL(k_patch<false>, grid, dim3(256), 0, s, a, d_params, n_bands, per_segment, make_int4(0, 0, 0, 0));
}

// :323 on its own: variance := m2 ./ (points + FLT_MIN) over the slot's live half columns (a dead one logically holds
// 0 / (0 + FLT_MIN) = 0, the layer's reset value: nothing to write)
__global__ __launch_bounds__(256) void k_variance(const Arena a, int slot)
{
    float *L = percall_ptr(a, slot);
    const uint32_t *tile_live = a.tile_live + (size_t)slot * a.tile_live_stride;
    for (int rank = blockIdx.x; rank < a.g.T; rank += gridDim.x) {
        const uint32_t live = tile_live[rank];
        const int cell = threadIdx.x;
        if (!((live >> live_bit(cell)) & 1u)) continue;
        L[percall_index(rank, PL_VARIANCE, cell)] = L[percall_index(rank, PL_M2, cell)] / (L[percall_index(rank, PL_POINTS, cell)] + FLT_MIN);
    }
}

// detect_ground_patches(map, section) as a stage of its own (gg_run_stage): section 0..3 = one quadrant (:325-328), -1 = all four
void launch_patch_stage(const Arena &a, const CloudParams *d_params, int slot, int section, hipStream_t s)
{
    hipLaunchKernelGGL(k_variance, dim3(std::min(a.g.T, 1024)), dim3(TILE_CELLS), 0, s, a, slot);
    const int ir = a.g.rows - 2 * HALO, ic = a.g.cols - 2 * HALO;
    if (ir <= 0 || ic <= 0 || a.g.tiles_c > MAXTC) return;
    const int gcols = a.g.cols, grows = a.g.rows;
    int4 b = make_int4(0, grows, 0, gcols); // (the table's "visited" flag already restricts to the union of the quadrants)
    if (section >= 0) {
        b.x = 2 + section % 2 * (gcols / 2 - 2);        // :325 (the reference's `i`, used as the ROW of detect_ground_patch)
        b.y = gcols / 2 + section % 2 * (gcols / 2 - 2); // :327
        b.z = section >= 2 ? grows / 2 : 2;              // :326 (`j`, the column)
        b.w = section >= 2 ? grows - 2 : grows / 2;      // :328
    }
    const int n_bands = (ir + PR - 1) / PR, n_blocks = (ic + PC - 1) / PC;
    const int segments = std::max(1, std::min(n_blocks, 8192 / std::max(1, n_bands)));
    const int per_segment = (n_blocks + segments - 1) / segments;
    dim3 grid(n_bands * ((n_blocks + per_segment - 1) / per_segment), 1);
    hipLaunchKernelGGL(k_patch<true>, grid, dim3(256), 0, s, a, d_params, n_bands, per_segment, b);
}

} // namespace gg
