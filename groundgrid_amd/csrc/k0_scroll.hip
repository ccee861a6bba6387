// K0 -- GroundGrid::update (src/GroundGrid.cpp:83-147) on the device: the map follows the vehicle.
//
// grid_map::GridMap::move shifts the map by a whole number of cells (the host computes the shift with grid_map's
// rounding, gg_context.hip gg_move_map), drops the rows / columns that fall out and exposes new ones;
// GroundGrid::update fills the exposed cells with ground = -(z of the cell centre expressed in base_link) and
// groundpatch = 0 (:121-131) and re-linearises the ring buffer (:143), which the hot path relies on (it indexes raw
// matrices).  In default-start-index terms that is one gather per cell:
//     new(i, j) = exposed(i, j) ? fill(i, j) : old((i + s0) mod n, (j + s1) mod n)
// Only `ground` and `groundpatch` persist across clouds (every other layer is rewritten by the next filter_cloud
// before anyone can observe it), so only those two are moved: 4 layer-passes per cloud instead of 22.
#include "gg_device.h"

namespace gg {

struct ScrollParams {
    int s0, s1;          // index shift (rows, cols), buffer order
    double pos_x, pos_y; // map position AFTER the move
    double first0, first1; // L/2 - res/2  (getVectorToFirstCell)
    double res;
    double m20, m21, m22, tz; // third row of the base_link<-map rotation and translation z (gg_move_map base_plane)
};

__global__ __launch_bounds__(256) void k_scroll(const Arena a, int slot, float2 *__restrict__ out, const ScrollParams sp)
{
    const float2 *gp2 = gp2_ptr(a, slot);
    const int rows = a.g.rows, cols = a.g.cols;
    const int i = blockIdx.x * 64 + (threadIdx.x & 63);
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= rows || j >= cols) return;
    const bool all = abs(sp.s0) >= rows || abs(sp.s1) >= cols;
    int bi = (i + sp.s0) % rows, bj = (j + sp.s1) % cols;
    if (bi < 0) bi += rows;
    if (bj < 0) bj += cols;
    const bool new0 = sp.s0 > 0 ? bi < sp.s0 : (sp.s0 < 0 ? bi >= rows + sp.s0 : false);
    const bool new1 = sp.s1 > 0 ? bj < sp.s1 : (sp.s1 < 0 ? bj >= cols + sp.s1 : false);
    float g, w;
    if (all || new0 || new1) {
        // grid_map getPositionFromIndex: position = mapPosition + offset + resolution * (-index)
        const double px = (sp.pos_x + sp.first0) + sp.res * (double)(-i);
        const double py = (sp.pos_y + sp.first1) + sp.res * (double)(-j);
        // doTransform: v_out.z = (m20 * x + m21 * y + m22 * 0) + origin.z ; ground = -z (:130), groundpatch = 0 (:131)
        const double z = ((sp.m20 * px + sp.m21 * py) + sp.m22 * 0.0) + sp.tz;
        g = (float)(-z);
        w = 0.0f;
    } else {
        const float2 v = gp2[gp_idx(a, bi, bj)];
        g = v.x;
        w = v.y;
    }
    out[gp_idx(a, i, j)] = make_float2(g, w); // same element order as the layer: copied back whole
}

void launch_scroll(const Arena &a, int slot, float2 *scratch, int s0, int s1, double pos_x, double pos_y, const double plane[4],
                   hipStream_t s)
{
    ScrollParams sp;
    sp.s0 = s0;
    sp.s1 = s1;
    sp.pos_x = pos_x;
    sp.pos_y = pos_y;
    sp.res = a.g.resolution;
    sp.first0 = a.g.half0 - 0.5 * a.g.resolution;
    sp.first1 = a.g.half1 - 0.5 * a.g.resolution;
    // third row of the base_link <- map rotation and translation z, as the binding built them (gg_move_map)
    sp.m20 = plane[0];
    sp.m21 = plane[1];
    sp.m22 = plane[2];
    sp.tz = plane[3];
    dim3 grid((a.g.rows + 63) / 64, (a.g.cols + 3) / 4);
    hipLaunchKernelGGL(k_scroll, grid, dim3(256), 0, s, a, slot, scratch, sp);
    hipMemcpyAsync(gp2_ptr(a, slot), scratch, (size_t)a.gpl.elems * 8, hipMemcpyDeviceToDevice, s); // (elements no cell maps to are never read)
}

} // namespace gg
