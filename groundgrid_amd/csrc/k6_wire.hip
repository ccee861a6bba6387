// K6 -- output wire formats of the nodelet (src/GroundGridNodelet.cpp:211-291), produced on the device so that only what
// is subscribed to crosses PCIe:
//   * per-layer 8-bit image: grid_map::GridMapCvConverter::toImage<unsigned char, 1> (Nodelet.cpp:239) -- the layer
//     normalised between the min and max of its finite cells, NaN / inf cells left 0.  (cv::applyColorMap, :240, is an
//     OpenCV lookup table applied by the host afterwards.)
//   * the 32FC3 "terrain" image (Nodelet.cpp:247-268): (ground, 3x3 pointsRaw sum >= 27 ? 1 : 0, pointsRaw) per cell.
// Images are row-major (cv::Mat), layers column-major (Eigen): the kernels transpose through the index math.
#include "gg_device.h"

#include <float.h>
#include <algorithm>

namespace gg {

// min / max over the finite cells of a layer (Eigen minCoeffOfFinites / maxCoeffOfFinites); out[0] = min, out[1] = max
__global__ __launch_bounds__(1024) void k_minmax_finite(const float *__restrict__ layer, int C, float *__restrict__ out)
{
    __shared__ float smin[16], smax[16];
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int k = threadIdx.x; k < C; k += 1024) {
        const float v = layer[k];
        if (isfinite(v)) {
            lo = fminf(lo, v);
            hi = fmaxf(hi, v);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, d, 64));
        hi = fmaxf(hi, __shfl_xor(hi, d, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        smin[threadIdx.x >> 6] = lo;
        smax[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) {
            lo = fminf(lo, smin[w]);
            hi = fmaxf(hi, smax[w]);
        }
        out[0] = lo;
        out[1] = hi;
    }
}

// GridMapCvConverter::toImage<unsigned char,1>: imageValue = (uchar)(((clamp(v, lo, hi) - lo) / (hi - lo)) * 255.f)
__global__ __launch_bounds__(256) void k_layer_to_u8(const float *__restrict__ layer, int rows, int cols, const float *__restrict__ bounds,
                                                     uint8_t *__restrict__ img)
{
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= rows || j >= cols) return;
    const float lo = bounds[0], hi = bounds[1];
    const float v = layer[(size_t)i + (size_t)j * rows];
    uint8_t o = 0;
    if (isfinite(v)) {
        const float c = v < lo ? lo : (v > hi ? hi : v);
        o = (uint8_t)(((c - lo) / (hi - lo)) * 255.0f);
    }
    img[(size_t)i * cols + j] = o;
}

// Nodelet.cpp:258-268; the reference reads block<3,3>(i-1, j-1) also on the border (UB): border cells get 0 for the flag.
__global__ __launch_bounds__(256) void k_terrain_image(const Arena a, int slot, float *__restrict__ img)
{
    const float2 *gp2 = gp2_ptr(a, slot);
    const float *percall = percall_ptr(a, slot);
    const int rows = a.g.rows, cols = a.g.cols;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= rows || j >= cols) return;
    float flag = 0.0f;
    // (pointsRaw is a sparse per-call layer: 0 outside the live half columns, gg_internal.h tile_live)
    auto raw_at = [&](int r, int c) { return cell_is_live(a, slot, r, c) ? percall[percall_index_of(a, PL_POINTSRAW, r, c)] : 0.0f; };
    if (i >= 1 && j >= 1 && i + 1 < rows && j + 1 < cols) {
        float e[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) e[s] = raw_at(i - 1 + s % 3, j - 1 + s / 3);
        flag = tree9(e) >= 27.0f ? 1.0f : 0.0f;
    }
    float *px = img + ((size_t)i * cols + j) * 3;
    px[0] = gp2[gp_idx(a, i, j)].x;
    px[1] = flag;
    px[2] = raw_at(i, j);
}

// One per-call layer as the dense column-major matrix the reference holds: stored values in the live half columns, the per-call reset
// value (:61-75) everywhere else (gg_internal.h tile_live).  The host boundary of the sparse layers (gg_get_layer, image getters).
__global__ __launch_bounds__(256) void k_layer_extract(const Arena a, int slot, int layer, float *__restrict__ dst)
{
    const float *src = percall_ptr(a, slot);
    const int rows = a.g.rows, position = percall_position(layer);
    const float dead = layer_reset_value(layer);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x)
        dst[i] = cell_is_live(a, slot, i % rows, i / rows) ? src[percall_index_of(a, position, i % rows, i / rows)] : dead;
}
void launch_layer_extract(const Arena &a, int slot, int layer, float *dst, hipStream_t s)
{
    const int blocks = (a.g.C + 255) / 256 < 2048 ? (a.g.C + 255) / 256 : 2048;
    hipLaunchKernelGGL(k_layer_extract, dim3(blocks), dim3(256), 0, s, a, slot, layer, dst);
}

// ... and all the layers a caller asked for in one launch (gg_get_layers): `want` = bit per gg_layer, plane k of `dst` = the k-th
// requested layer in gg_layer order.  The cell's liveness, its place in the tile blocks and its element of the sheared
// (ground, confidence) layer are worked out once for all of them -- eleven launches of the kernels above are eleven launch
// latencies for 0.5 MB each.
__global__ __launch_bounds__(256) void k_layers_extract(const Arena a, int slot, unsigned want, float *__restrict__ dst, size_t plane)
{
    const float *src = percall_ptr(a, slot);
    const float2 *gp2 = gp2_ptr(a, slot);
    const int rows = a.g.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x) {
        const int r = i % rows, c = i / rows;
        const bool live = cell_is_live(a, slot, r, c);
        const size_t at = percall_index_of(a, 0, r, c);
        float2 g = make_float2(0.f, 0.f);
        if (want & ((1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH))) g = gp2[gp_idx(a, r, c)];
        int k = 0;
#pragma unroll
        for (int l = 0; l < GG_NUM_LAYERS; ++l) {
            if (!((want >> l) & 1u)) continue; // (uniform)
            float v;
            if (l == GG_LAYER_GROUND) v = g.x;
            else if (l == GG_LAYER_GROUNDPATCH) v = g.y;
            else v = live ? src[at + (size_t)percall_position(l) * (TILE * TILE)] : layer_reset_value(l);
            dst[(size_t)k * plane + i] = v;
            ++k;
        }
    }
}
void launch_layers_extract(const Arena &a, int slot, unsigned want, float *dst, size_t plane_floats, hipStream_t s)
{
    const int blocks = (a.g.C + 255) / 256 < 2048 ? (a.g.C + 255) / 256 : 2048;
    hipLaunchKernelGGL(k_layers_extract, dim3(blocks), dim3(256), 0, s, a, slot, want, dst, plane_floats);
}

// Make a slot's per-call layers dense in place: every dead column receives the reset values, then every column is marked live.
// Needed before the host overwrites ONE per-call layer (gg_set_layer): the liveness masks are shared by the nine layers.
__global__ __launch_bounds__(256) void k_materialise(const Arena a, int slot)
{
    const int rows = a.g.rows;
    float *L = percall_ptr(a, slot);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x) {
        if (cell_is_live(a, slot, i % rows, i / rows)) continue;
        const size_t at = percall_index_of(a, 0, i % rows, i / rows);
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if (percall_position(l) >= 0) L[at + (size_t)percall_position(l) * (TILE * TILE)] = layer_reset_value(l);
    }
}
void launch_materialise_layers(const Arena &a, int slot, hipStream_t s)
{
    const int blocks = (a.g.C + 255) / 256 < 2048 ? (a.g.C + 255) / 256 : 2048;
    hipLaunchKernelGGL(k_materialise, dim3(blocks), dim3(256), 0, s, a, slot);
    launch_fill_bytes((uint8_t *)(a.tile_live + (size_t)slot * a.tile_live_stride), (size_t)a.g.T * 4, 0xFF, s); // (after the kernel, same stream)
}

// The host's dense column-major matrix into one per-call layer (gg_set_layer, after launch_materialise_layers: every column live).
__global__ __launch_bounds__(256) void k_layer_insert(const Arena a, int slot, int layer, const float *__restrict__ src)
{
    float *dst = percall_ptr(a, slot);
    const int rows = a.g.rows, position = percall_position(layer);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.g.C; i += gridDim.x * blockDim.x) dst[percall_index_of(a, position, i % rows, i / rows)] = src[i];
}
void launch_layer_insert(const Arena &a, int slot, int layer, const float *src, hipStream_t s)
{
    const int blocks = (a.g.C + 255) / 256 < 2048 ? (a.g.C + 255) / 256 : 2048;
    hipLaunchKernelGGL(k_layer_insert, dim3(blocks), dim3(256), 0, s, a, slot, layer, src);
}

// GroundGrid's initial values (src/GroundGrid.cpp:71-75) into every per-call layer of n slots: element e of a slot's region belongs
// to layer position (e / 256) % 9
struct PercallInit {
    float v[PERCALL_LAYERS];
};
__global__ __launch_bounds__(256) void k_fill_percall(float *__restrict__ dst, size_t n, size_t stride, const PercallInit init)
{
    float *d = dst + (size_t)blockIdx.y * stride;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = init.v[(i / (TILE * TILE)) % PERCALL_LAYERS];
}
void launch_fill_percall(const Arena &a, int first_slot, int n_slots, const float init[GG_NUM_LAYERS], hipStream_t s)
{
    if (n_slots <= 0) return;
    PercallInit pi;
    for (int l = 0; l < GG_NUM_LAYERS; ++l)
        if (percall_position(l) >= 0) pi.v[percall_position(l)] = init[l];
    const size_t n = (size_t)a.g.T * PERCALL_BLOCK;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, n_slots >= 64 ? (size_t)64 : (size_t)1024);
    hipLaunchKernelGGL(k_fill_percall, dim3(blocks, n_slots), dim3(256), 0, s, percall_ptr(a, first_slot), n, a.slot_layer_stride, pi);
}

void launch_layer_to_u8(const float *layer, int rows, int cols, float *d_bounds, uint8_t *d_img, hipStream_t s)
{
    hipLaunchKernelGGL(k_minmax_finite, dim3(1), dim3(1024), 0, s, layer, rows * cols, d_bounds);
    dim3 grid((cols + 63) / 64, (rows + 3) / 4);
    hipLaunchKernelGGL(k_layer_to_u8, grid, dim3(256), 0, s, layer, rows, cols, d_bounds, d_img);
}

void launch_terrain_image(const Arena &a, int slot, float *d_img, hipStream_t s)
{
    dim3 grid((a.g.cols + 63) / 64, (a.g.rows + 3) / 4);
    hipLaunchKernelGGL(k_terrain_image, grid, dim3(256), 0, s, a, slot, d_img);
}

} // namespace gg
