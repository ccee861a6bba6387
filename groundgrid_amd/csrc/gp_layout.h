// gp_layout.h -- where cell (row, col) of the interleaved (ground, confidence) layer lives in HBM.
//
// `ground` and `groundpatch` are the only state that persists from cloud to cloud, they are always used together, and the
// kernel that dominates their traffic is the terrain sweep (k_sweep): per step, the 64 lanes of a wavefront -- 64 consecutive
// rings of one side of the spiral, ring r + 1 SKEW steps behind ring r -- each touch ONE cell.  In a row- or column-major
// matrix those 64 cells lie in 64 different 128-byte lines (the texture-address unit then spends ~128 cycles per wave
// instruction: measured TA_BUSY = kernel time).  The layout below shears each 64-ring wedge so that exactly those 64 cells
// are 64 CONSECUTIVE elements:
//
//   c = n/2 - 1,  dx = row - c,  dy = col - c,  r = max(|dx|, |dy|)  (the ring),  g = (r - 1) / 64,  l = (r - 1) % 64
//   (SH = GP_SHEAR = SKEW + 1)
//   side A (row = c - r, the cells the sweep's first side walks):   dx == -r && dy <  r      v = col + SH l
//   side C (row = c + r):                                            dx ==  r                 v = (n - 1 - col) + SH l
//   side B (col = c - r, without the two corners):                   dy == -r                 v = row + SH l
//   side D (col = c + r, with the corner (c - r, c + r)):            otherwise                v = (n - 1 - row) + SH l
//   index = 1 + ((side * G + g) * VS + v) * 64 + l          (index 0 = the centre cell;  VS = n + SH * 63)
//
// A chain lane walks its side with stride 64 elements; at wave-step t every lane of a wavefront is at the same v (its ring
// is l steps of SKEW behind and l cells of 1 further in: SKEW + 1 = the shear), so a wavefront reads and writes 512
// contiguous bytes per access.  The outer line of ring r is the own line of ring r + 1: element index + SH * 64 + 1.
// Every other kernel addresses the layer through gp_index(): their accesses to it are per-point gathers or one element
// per cell, a small part of their traffic.  The host boundary (gg_get_layer / gg_set_layer) converts to and from Eigen's
// column-major matrices.  Footprint: 4 * G * (n + 63 SH) * 64 * 8 B (3.0 MB for n = 364 and SH = 2 instead of 1.06 MB), of which only
// the n * n cells are ever touched.
#pragma once

#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define GPL_HD __host__ __device__ __forceinline__
#else
#define GPL_HD inline
#endif

// steps between neighbouring rings in the sweep (sweep_core.h SKEW) and the shear that follows from it
#ifndef GG_SWEEP_SKEW
#define GG_SWEEP_SKEW 1
#endif

namespace gg {

constexpr int GP_SHEAR = GG_SWEEP_SKEW + 1;

struct GpLayout {
    int n, c;  // rows = cols, centre index
    int G;     // storage groups of 64 rings (rings 1 .. n - 1 - c)
    int VS;    // n + GP_SHEAR * 63: sheared positions per (side, group)
    int elems; // 1 + 4 * G * VS * 64
};

GPL_HD GpLayout make_gp_layout(int n)
{
    GpLayout L;
    L.n = n;
    L.c = n / 2 - 1;
    const int rmax = n - 1 - L.c;
    L.G = (rmax - 1) / 64 + 1;
    L.VS = n + GP_SHEAR * 63;
    L.elems = 1 + 4 * L.G * L.VS * 64;
    return L;
}

GPL_HD int gp_index(const GpLayout &L, int row, int col)
{
    const int dx = row - L.c, dy = col - L.c;
    const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
    const int r = ax > ay ? ax : ay;
    if (r == 0) return 0;
    const int g = (r - 1) >> 6, l = (r - 1) & 63;
    int side, v;
    if (dx == -r && dy < r) {
        side = 0;
        v = col;
    } else if (dx == r) {
        side = 2;
        v = L.n - 1 - col;
    } else if (dy == -r) {
        side = 1;
        v = row;
    } else {
        side = 3;
        v = L.n - 1 - row;
    }
    return 1 + ((side * L.G + g) * L.VS + v + GP_SHEAR * l) * 64 + l;
}

// the inverse: which cell element `e` holds (false: the centre's slot 0 is cell (c, c); padding elements hold no cell)
GPL_HD bool gp_cell_of(const GpLayout &L, int e, int &row, int &col)
{
    if (e == 0) {
        row = col = L.c;
        return true;
    }
    if (e < 0 || e >= L.elems) return false;
    const int q = e - 1, l = q & 63, rest = q >> 6;
    const int vs = rest % L.VS, sg = rest / L.VS, g = sg % L.G, side = sg / L.G;
    const int r = 64 * g + l + 1, v = vs - GP_SHEAR * l;
    if (side > 3 || v < 0 || v >= L.n) return false;
    if (side == 0) {
        row = L.c - r;
        col = v;
    } else if (side == 2) {
        row = L.c + r;
        col = L.n - 1 - v;
    } else if (side == 1) {
        col = L.c - r;
        row = v;
    } else {
        col = L.c + r;
        row = L.n - 1 - v;
    }
    if (row < 0 || row >= L.n || col < 0 || col >= L.n) return false;
    return gp_index(L, row, col) == e; // (the side's own range of v: the other positions of the plane are padding)
}

} // namespace gg
