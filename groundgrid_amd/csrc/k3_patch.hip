// K3 -- detect_ground_patches / detect_ground_patch<3|5> (src/GroundSegmentation.cpp:314-395).
//
// A pure read-neighbours / write-self stencil: each interior cell reads an SxS block (S = 3 inside
// patch_size_change_distance, else 5) of `points`, `variance`, `minGroundHeight` and updates only its own
// `ground` / `groundpatch`, so the reference's four quadrant threads are order-free and one thread per
// cell is exact.  Work-group = 32 rows x 8 cols of cells (rows are the contiguous dimension of the
// column-major layers), inputs staged in LDS with a 2-cell halo.  The block sums use Eigen's unrolled
// tree order (gg_device.h tree9/tree25).
//
// Algorithmic bytes per cell: 6 layers read (points, variance, min, ground, groundpatch, expectedPoints),
// 2 written.
#include "gg_device.h"

namespace gg {

constexpr int PR = 32, PC = 8, HALO = 2;
constexpr int LR = PR + 2 * HALO, LC = PC + 2 * HALO; // 36 x 12

template <int S>
GG_DEV void detect_ground_patch(const Arena &a, const float (*pts)[LR], const float (*var)[LR], const float (*mnl)[LR],
                                int lr, int lc, int i, int j, float sqdist, float2 *gp2)
{
    constexpr int SS = S * S;
    constexpr int ci = S / 2; // :352
    const DevConfig &cfg = a.cfg;
    const int rows = a.g.rows;
    float e[SS];
    // :355 pointsBlock, column-major linear index s -> (row s % S, col s / S)
#pragma unroll
    for (int s = 0; s < SS; ++s) e[s] = pts[lc - ci + s / S][lr - ci + s % S];
    const bool e34 = a.eigen_reduction == GG_EIGEN_34_SSE; // (uniform) which Eigen the reference was built against
    auto sum25 = [&](const float *v) { return e34 ? tree25_eigen34(v) : tree25(v); };
    const float pointsblockSum = (S == 3) ? tree9(e) : sum25(e); // :359
    const size_t idx = (size_t)i + (size_t)j * rows;
    const float expected = a.expected[idx]; // :358

    // :364-365
    if ((double)pointsblockSum < std_max(floor(cfg.gpd_min_point_count_threshold * (double)S * (double)expected), 3.0)) return;

    const int gidx = gp_idx(a, i, j); // the (ground, confidence) layer has its own element order (gp_layout.h)
    const float2 old = gp2[gidx];
    const float oldConfidence = old.y;   // :360
    const float oldGroundheight = old.x; // :361

    // :369
    const float varThresholdsq =
        (float)std_min(std_max((double)sqdist * cfg.distance_factor_sq, cfg.minimum_distance_factor_sq), cfg.minimum_distance_factor_x10_sq);
    const float variance = var[lc][lr]; // :372
    float localmin = mnl[lc - ci][lr - ci]; // :373 minCoeff (this layer never holds NaN)
#pragma unroll
    for (int s = 1; s < SS; ++s) {
        const float v = mnl[lc - ci + s / S][lr - ci + s % S];
        if (v < localmin) localmin = v;
    }
    // :374
    float maxVar;
    if (e[ci + ci * S] >= (float)cfg.point_count_cell_variance_threshold) {
        maxVar = variance;
    } else {
        float pr[SS];
#pragma unroll
        for (int s = 0; s < SS; ++s) pr[s] = e[s] * var[lc - ci + s / S][lr - ci + s % S];
        maxVar = ((S == 3) ? tree9(pr) : sum25(pr)) / pointsblockSum;
    }
    // :375
    float pm[SS];
#pragma unroll
    for (int s = 0; s < SS; ++s) pm[s] = e[s] * mnl[lc - ci + s / S][lr - ci + s % S];
    const float groundlevel = ((S == 3) ? tree9(pm) : sum25(pm)) / pointsblockSum;
    // :376
    const float groundDiff = std_max((groundlevel - oldGroundheight) * (2.0f * oldConfidence), 1.0f);

    // :379-380
    if ((double)oldConfidence > 0.5 && (double)groundlevel >= (double)oldGroundheight + cfg.outlier_tolerance) return;

    // :382
    if ((double)varThresholdsq > (double)maxVar * (double)maxVar && maxVar > 0.0f &&
        (double)pointsblockSum > (double)((groundDiff * expected) * (float)S) * cfg.gpd_min_point_count_threshold) {
        const float newConfidence = (float)std_min((double)pointsblockSum / cfg.occupied_cells_point_count_factor, 1.0); // :383
        const float G = (groundlevel * newConfidence + (oldConfidence * oldGroundheight) * 2.0f) / (newConfidence + oldConfidence * 2.0f); // :385
        const float Cf =
            (float)std_min(((double)pointsblockSum / cfg.occupied_cells_point_count_factor_x2 + (double)oldConfidence) / 2.0, 1.0); // :387
        gp2[gidx] = make_float2(G, Cf);
    } else if (localmin < oldGroundheight) { // :389
        gp2[gidx] = make_float2(localmin, std_min(oldConfidence + 0.1f, 0.5f)); // :391, :393
    }
}

__global__ __launch_bounds__(256) void k_patch(const Arena a, const CloudParams *__restrict__ params)
{
    __shared__ float pts[LC][LR], var[LC][LR], mnl[LC][LR];
    const int cloud = blockIdx.z;
    const CloudParams cp = params[cloud];
    const int rows = a.g.rows, cols = a.g.cols;
    const int r0 = HALO + blockIdx.x * PR, c0 = HALO + blockIdx.y * PC; // first output cell of this block

    const float *L = a.layers + (size_t)cp.slot * a.slot_layer_stride;
    const float *gp_pts = L + GG_LAYER_POINTS * a.layer_stride;
    const float *gp_var = L + GG_LAYER_VARIANCE * a.layer_stride;
    const float *gp_min = L + GG_LAYER_MINGROUNDHEIGHT * a.layer_stride;

    for (int k = threadIdx.x; k < LR * LC; k += 256) {
        const int lr = k % LR, lc = k / LR;
        const int gr = r0 - HALO + lr, gcol = c0 - HALO + lc;
        float p = 0.0f, v = 0.0f, m = 0.0f;
        if (gr < rows && gcol < cols) {
            const size_t idx = (size_t)gr + (size_t)gcol * rows;
            p = gp_pts[idx];
            v = gp_var[idx];
            m = gp_min[idx];
        }
        pts[lc][lr] = p;
        var[lc][lr] = v;
        mnl[lc][lr] = m;
    }
    __syncthreads();

    const int tr = threadIdx.x % PR, tcl = threadIdx.x / PR;
    const int i = r0 + tr, j = c0 + tcl;
    // the four quadrants (:325-328) cover rows [2, 2 * (cols / 2) - 2) -- the FIRST loop variable, bounded by cols / 2, is
    // used as the row index -- and cols [2, rows - 2): for odd sizes row n - 3 is never visited
    if (i >= 2 * (cols / 2) - 2 || j >= rows - 2) return;

    // :332
    const double di = (double)i - (double)rows / 2.0, dj = (double)j - (double)cols / 2.0;
    const float sqdist = (float)((di * di + dj * dj) * ((double)a.g.resolution_f * (double)a.g.resolution_f));
    float2 *gp2 = gp2_ptr(a, cp.slot);
    if ((double)sqdist <= a.cfg.patch_size_change_distance_sq) // :334
        detect_ground_patch<3>(a, pts, var, mnl, tr + HALO, tcl + HALO, i, j, sqdist, gp2);
    else
        detect_ground_patch<5>(a, pts, var, mnl, tr + HALO, tcl + HALO, i, j, sqdist, gp2);
}

void launch_patch(const Arena &a, const CloudParams *d_params, int n_clouds, hipStream_t s)
{
    if (n_clouds == 0) return;
    const int ir = a.g.rows - 2 * HALO, ic = a.g.cols - 2 * HALO;
    if (ir <= 0 || ic <= 0) return;
    dim3 grid((ir + PR - 1) / PR, (ic + PC - 1) / PC, n_clouds);
    hipLaunchKernelGGL(k_patch, grid, dim3(256), 0, s, a, d_params);
}

} // namespace gg
