// gg_device.h -- device-side scalar building blocks shared by the kernels.
//
// Every kernel file is compiled with -ffp-contract=off and without fast-math: the reference is
// x86-64 SSE2 code with no FMA contraction and IEEE divide/sqrt, and ground / non-ground labels
// are threshold tests on these values, so each float and double operation below is written in
// exactly the order and precision of src/GroundSegmentation.cpp (see oracle/gg_oracle.c for the
// same expressions on the CPU).
#pragma once

#include "gg_internal.h"

namespace gg {

// The kernels' measurement switches (env GG_K2_DEBUG / GG_K3_DEBUG / GG_K5_DEBUG -> Arena::k2_debug ...: early returns, skipped gathers or
// stores, cycle counters; results void) exist only in libraries built with -DGG_INSTRUMENT (tools/build_variant.sh inst "-DGG_INSTRUMENT"):
// the production kernels are compiled as if they were 0.
#ifdef GG_INSTRUMENT
#define GG_DEBUG_SWITCH(a, field) ((a).field)
#else
#define GG_DEBUG_SWITCH(a, field) 0
#endif


#define GG_DEV __device__ __forceinline__

// libstdc++ std::min / std::max (NaN: return the first argument when the comparison is false)
GG_DEV double std_min(double a, double b) { return (b < a) ? b : a; }
GG_DEV double std_max(double a, double b) { return (a < b) ? b : a; }
GG_DEV float std_min(float a, float b) { return (b < a) ? b : a; }
GG_DEV float std_max(float a, float b) { return (a < b) ? b : a; }

// Eigen 3.3.7 redux_novec_unroller order for a 3x3 / 5x5 fixed-size block, e[] in column-major
// linear order of the block (oracle/gg_oracle.c tree9 / tree25).
GG_DEV float tree9(const float *e)
{
    return ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + (e[7] + e[8])));
}
GG_DEV float tree25(const float *e)
{
    const float a = (e[0] + (e[1] + e[2])) + (e[3] + (e[4] + e[5]));
    const float b = (e[6] + (e[7] + e[8])) + (e[9] + (e[10] + e[11]));
    const float c = (e[12] + (e[13] + e[14])) + (e[15] + (e[16] + e[17]));
    const float d = (e[18] + (e[19] + e[20])) + ((e[21] + e[22]) + (e[23] + e[24]));
    return (a + b) + (c + d);
}

// Eigen 3.4.x, SSE2 build (Packet4f): Block<MatrixXf,5,5>::sum() takes SliceVectorizedTraversal -- rows 0..3 of the five
// columns are accumulated lane-wise, column by column, reduced with predux = (a0 + a2) + (a1 + a3), then row 4 of every
// column is added in column order (Redux.h redux_impl<..., SliceVectorizedTraversal, ...>; oracle/gg_oracle.c tree25_eigen34).
GG_DEV float tree25_eigen34(const float *e)
{
    float p[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = (((e[r] + e[5 + r]) + e[10 + r]) + e[15 + r]) + e[20 + r];
    float res = (p[0] + p[2]) + (p[1] + p[3]);
#pragma unroll
    for (int j = 0; j < 5; ++j) res = res + e[4 + 5 * j];
    return res;
}

// x86-64 cvttsd2si: truncation toward zero, INT_MIN for NaN / out of range
GG_DEV int trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
    return (int)v;
}

// trunc_to_int(-(a / res)) without the f64 divide on the common path.  q = a * (1/res) differs from the correctly
// rounded quotient fl(a / res) by less than |q| * 2^-51 (one rounding in 1/res, one in the product, half an ulp in the
// true division), so whenever q is further than |q| * 2^-49 from the nearest integer both values truncate to the same
// integer; otherwise (probability ~1e-13 per point, or a == 0, NaN, inf) the exact IEEE division decides.
GG_DEV int index_of(double a, double res, double inv_res)
{
    const double q = a * inv_res;
    const double r = rint(q);
    if (fabs(q - r) > fabs(q) * 0x1p-49) return trunc_to_int(-q);
    return trunc_to_int(-(a / res));
}

// grid_map_core getIndexFromPosition (value part): index = (int)(-(((p - L/2) - mapPos) / res))
GG_DEV void index_from_position(const Geometry &g, double pos_x, double pos_y, double px, double py, int &row, int &col)
{
    row = index_of((px - g.half0) - pos_x, g.resolution, g.inv_resolution);
    col = index_of((py - g.half1) - pos_y, g.resolution, g.inv_resolution);
}

// grid_map_core checkIfPositionWithinMap: t = M ((p - mapPos) - L/2) with M = [[-1, 0], [0, -1]] (the buffer-order transform);
// 0 <= t < L per axis.  t_x = -1.0 * a_x + 0.0 * a_y equals -a_x whenever a_y is finite (the products are exact, adding a signed
// zero changes at most the sign of a zero, which no comparison sees), and when a_y is NaN or infinite t_y = 0.0 * a_x - a_y fails
// its own range test -- so "inside" is exactly the four comparisons on -a_x, -a_y (the same holds with x and y exchanged); the
// oracle keeps the matrix form (oracle/gg_oracle.c ggo_get_index), the parity tests hold the two together (NaN / inf points in
// test_edge_cases and the fuzz scenes).
GG_DEV bool position_inside(const Geometry &g, double pos_x, double pos_y, double px, double py)
{
    const double ax = (px - pos_x) - g.half0;
    const double ay = (py - pos_y) - g.half1;
    const double tx = -ax, ty = -ay;
    return tx >= 0.0 && ty >= 0.0 && tx < g.length0 && ty < g.length1;
}

// glibc e_hypotf.c: (float)sqrt((double)x*x + (double)y*y) for finite arguments
GG_DEV float ref_hypotf(float x, float y)
{
    if (isinf(x) || isinf(y)) return __builtin_inff();
    if (isnan(x) || isnan(y)) return x + y;
    const double dx = (double)x, dy = (double)y;
    return (float)sqrt(dx * dx + dy * dy);
}

// tf2::doTransform of one point (src/GroundGridNodelet.cpp:166-181): v_out = basis * v + origin with the dot products
// evaluated left to right in double, each coordinate cast back to float.
GG_DEV void transform_point(const double (&tf)[12], float &x, float &y, float &z)
{
    const double dx = (double)x, dy = (double)y, dz = (double)z;
    const double ox = ((tf[0] * dx + tf[1] * dy) + tf[2] * dz) + tf[3];
    const double oy = ((tf[4] * dx + tf[5] * dy) + tf[6] * dz) + tf[7];
    const double oz = ((tf[8] * dx + tf[9] * dy) + tf[10] * dz) + tf[11];
    x = (float)ox;
    y = (float)oy;
    z = (float)oz;
}

GG_DEV int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// number of set bits of `mask` below this lane
GG_DEV int rank_below(unsigned long long mask)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// Runs.  A scan line crosses a cell several returns at a time, so the records of a 64-record window mostly come as runs of
// consecutive lanes with the same cell -- and same-address LDS atomics execute one lane at a time.  The first lane of a run
// speaks for the whole run: `head` and the run's lane mask.  `id` = the lane's cell, or any value the runs of interest never
// take for lanes that do not take part (they split runs, which is only conservative).
struct LaneRun {
    bool head;
    unsigned long long mask;
};
GG_DEV LaneRun lane_run(uint32_t id, int lane)
{
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp((int)id, (int)id, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
    LaneRun r;
    r.head = lane == 0 || id != prev;
    const unsigned long long hm = __ballot(r.head);
    const unsigned long long above = (hm >> 1) >> lane; // heads after this lane
    const unsigned long long upto = above & (0ull - above); // the next head, as a bit relative to lane + 1 (0: none)
    r.mask = ((upto << 1) - 1ull) << lane; // lanes [lane, next head); upto == 0 -> all lanes from this one up
    return r;
}

// cell (row, col) of a key
GG_DEV void key_to_cell(const Arena &a, uint32_t key, int &row, int &col)
{
    const uint32_t c0 = a.rank_cell0[key >> KEY_TILE_SHIFT];
    row = (int)(c0 & 0xFFFFu) + (int)(key & 15u);
    col = (int)(c0 >> 16) + (int)((key >> 4) & 15u);
}


// XCD-aware work distribution.  The 8 XCDs (each with its own 4 MiB L2) receive work-groups round-robin by dispatch order,
// so neighbouring blockIdx values land on DIFFERENT L2s: two tiles that share 128-byte lines would each leave half-written
// lines in two caches.  This remaps the linear dispatch id so that XCD x works through one contiguous range of the
// (cloud-major, item-minor) work list: consecutive items -- vertically adjacent tiles, the same cloud -- meet in one L2.
// Returns the item index in [0, n_items).
GG_DEV uint32_t xcd_contiguous_item(uint32_t lin, uint32_t n_items)
{
    const uint32_t xcd = lin & 7u, idx = lin >> 3;
    const uint32_t q = n_items >> 3, r = n_items & 7u;
    return xcd * q + (xcd < r ? xcd : r) + idx;
}

} // namespace gg
