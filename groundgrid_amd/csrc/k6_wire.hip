// K6 -- output wire formats of the nodelet (src/GroundGridNodelet.cpp:211-291), produced on the device so that only what
// is subscribed to crosses PCIe:
//   * per-layer 8-bit image: grid_map::GridMapCvConverter::toImage<unsigned char, 1> (Nodelet.cpp:239) -- the layer
//     normalised between the min and max of its finite cells, NaN / inf cells left 0.  (cv::applyColorMap, :240, is an
//     OpenCV lookup table applied by the host afterwards.)
//   * the 32FC3 "terrain" image (Nodelet.cpp:247-268): (ground, 3x3 pointsRaw sum >= 27 ? 1 : 0, pointsRaw) per cell.
// Images are row-major (cv::Mat), layers column-major (Eigen): the kernels transpose through the index math.
#include "gg_device.h"

#include <float.h>

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
    const float *raw = layer_ptr(a, slot, GG_LAYER_POINTSRAW);
    const int rows = a.g.rows, cols = a.g.cols;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= rows || j >= cols) return;
    float flag = 0.0f;
    if (i >= 1 && j >= 1 && i + 1 < rows && j + 1 < cols) {
        float e[9];
#pragma unroll
        for (int s = 0; s < 9; ++s) e[s] = raw[(size_t)(i - 1 + s % 3) + (size_t)(j - 1 + s / 3) * rows];
        flag = tree9(e) >= 27.0f ? 1.0f : 0.0f;
    }
    float *px = img + ((size_t)i * cols + j) * 3;
    px[0] = gp2[gp_idx(a, i, j)].x;
    px[1] = flag;
    px[2] = raw[(size_t)i + (size_t)j * rows];
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
