/*
 * gg_oracle.c -- CPU restatement of GroundGrid's per-cloud hot path.  TEST INFRASTRUCTURE ONLY
 * (see gg_oracle.h).  PARITY UNPINNED: no reference test vectors exist and the reference cannot
 * be built in this image; every function below cites the reference lines it restates.
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fexcess-precision=standard
 * (x86-64 SSE2 scalar math, FLT_EVAL_METHOD == 0, no FMA contraction: what `catkin build
 * -DCMAKE_BUILD_TYPE=Release` produces for the reference on x86-64).
 *
 * Precision notes (C++ overload resolution of the reference, g++ 11):
 *   std::pow(float,double) -> double     std::pow(float,float) -> float (folded to x*x)
 *   std::hypot(float,float) -> float     float / double -> double     int * float -> float
 *   std::min/max(a,b) return a when the comparison with NaN is false.
 * Every implicit promotion of the reference is written here as an explicit cast.
 */
#include "gg_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

#define L(m, id, i, j) ((m)->layer[(id)][(size_t)(i) + (size_t)(j) * (size_t)(m)->rows])

/* double -> int the way x86-64 cvttsd2si does it (Eigen's cast<int>() on the reference's
 * platform): truncation toward zero, "integer indefinite" (INT_MIN) for NaN / out of range. */
static int trunc_to_int(double v)
{
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

/* std::min / std::max exactly as libstdc++ defines them (NaN behaviour matters at :171) */
static double std_min_d(double a, double b) { return (b < a) ? b : a; }
static double std_max_d(double a, double b) { return (a < b) ? b : a; }
static float std_min_f(float a, float b) { return (b < a) ? b : a; }
static float std_max_f(float a, float b) { return (a < b) ? b : a; }
static int std_max_i(int a, int b) { return (a < b) ? b : a; }

/* Eigen 3.3.7 Redux.h redux_novec_unroller<Func,Derived,Start,Length>:
 *   f(Start,Length) = f(Start,Length/2) + f(Start+Length/2, Length-Length/2), f(s,1) = coeff(s)
 * with s the column-major linear index of the fixed-size block (DefaultTraversal +
 * CompleteUnrolling is what a Block<MatrixXf,S,S> (S = 3,5) sum()/product-sum selects). */
float ggo_tree_sum(const float *e, int len)
{
    if (len == 1) return e[0];
    int half = len / 2;
    float a = ggo_tree_sum(e, half);
    float b = ggo_tree_sum(e + half, len - half);
    return a + b;
}

/* The same order written out for the two block sizes the path uses (so the compiler can keep
 * everything in registers, as Eigen's complete unrolling does for the reference). */
#define GGO_ALWAYS_INLINE static inline __attribute__((always_inline))
GGO_ALWAYS_INLINE float tree9(const float *e)
{
    return ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + (e[7] + e[8])));
}
GGO_ALWAYS_INLINE float tree25(const float *e)
{
    const float a = (e[0] + (e[1] + e[2])) + (e[3] + (e[4] + e[5]));
    const float b = (e[6] + (e[7] + e[8])) + (e[9] + (e[10] + e[11]));
    const float c = (e[12] + (e[13] + e[14])) + (e[15] + (e[16] + e[17]));
    const float d = (e[18] + (e[19] + e[20])) + ((e[21] + e[22]) + (e[23] + e[24]));
    return (a + b) + (c + d);
}
/* Eigen 3.4.x Redux.h, SSE2 build (Packet4f = 4 floats): for a 5x5 block SliceVectorizedWork = (5 / 4) * 5 = 5 >= 3, so
 * redux_impl<Func, Evaluator, SliceVectorizedTraversal, NoUnrolling> runs:
 *   packet_res = packet(0,0);  for j in 0..4: for i in (j == 0 ? 4 : 0) .. 4 step 4: packet_res += packet(j, i)
 *   res = predux(packet_res)            SSE: tmp = a + movehl(a,a); tmp[0] + tmp[1]  ==  (a0 + a2) + (a1 + a3)
 *   for j in 0..4: for i in 4..5: res = res + coeff(j, i)         (row 4 of every column, in column order)
 * (3x3 blocks: SliceVectorizedWork = 0, they stay on redux_novec_unroller.)  cwiseProduct().sum() applies the same
 * traversal to the element-wise products. */
GGO_ALWAYS_INLINE float tree25_eigen34(const float *e)
{
    float p[4];
    for (int r = 0; r < 4; ++r) p[r] = (((e[r] + e[5 + r]) + e[10 + r]) + e[15 + r]) + e[20 + r];
    float res = (p[0] + p[2]) + (p[1] + p[3]);
    for (int j = 0; j < 5; ++j) res = res + e[4 + 5 * j];
    return res;
}
static int g_eigen_reduction = 0;
void ggo_set_eigen_reduction(int order) { g_eigen_reduction = order ? 1 : 0; }
int ggo_get_eigen_reduction(void) { return g_eigen_reduction; }
GGO_ALWAYS_INLINE float treeSS(const float *e, const int S)
{
    return S == 3 ? tree9(e) : (g_eigen_reduction ? tree25_eigen34(e) : tree25(e));
}

float ggo_block_sum(const float *e, int S) { return treeSS(e, S); }

/* Quaternion -> rotation matrix, the two candidates behind tf2::doTransform (see gg_oracle.h):
 *   0  tf2/LinearMath/Matrix3x3.h setRotation: d = |q|^2, s = 2/d, xs = x*s ..., rows
 *        (1-(yy+zz), xy-wz, xz+wy), (xy+wz, 1-(xx+zz), yz-wx), (xz-wy, yz+wx, 1-(xx+yy))
 *   1  orocos_kdl frames.cpp Rotation::Quaternion(x,y,z,w): x2 = x*x ..., rows
 *        (w2+x2-y2-z2, 2xy-2wz, 2xz+2wy), (2xy+2wz, w2-x2+y2-z2, 2yz-2wx), (2xz-2wy, 2yz+2wx, w2-x2-y2+z2) */
void ggo_rotation_from_quaternion(int convention, const double q[4], double R[9])
{
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    if (convention == 0) {
        const double d = x * x + y * y + z * z + w * w; /* length2(): x*x + y*y + z*z + w*w, left to right */
        const double s = 2.0 / d;
        const double xs = x * s, ys = y * s, zs = z * s;
        const double wx = w * xs, wy = w * ys, wz = w * zs;
        const double xx = x * xs, xy = x * ys, xz = x * zs;
        const double yy = y * ys, yz = y * zs, zz = z * zs;
        R[0] = 1.0 - (yy + zz); R[1] = xy - wz;         R[2] = xz + wy;
        R[3] = xy + wz;         R[4] = 1.0 - (xx + zz); R[5] = yz - wx;
        R[6] = xz - wy;         R[7] = yz + wx;         R[8] = 1.0 - (xx + yy);
    } else {
        const double x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
        R[0] = w2 + x2 - y2 - z2;       R[1] = 2 * x * y - 2 * w * z;   R[2] = 2 * x * z + 2 * w * y;
        R[3] = 2 * x * y + 2 * w * z;   R[4] = w2 - x2 + y2 - z2;       R[5] = 2 * y * z - 2 * w * x;
        R[6] = 2 * x * z - 2 * w * y;   R[7] = 2 * y * z + 2 * w * x;   R[8] = w2 - x2 - y2 + z2;
    }
}

/* glibc sysdeps/ieee754/flt-32/e_hypotf.c (2.13 .. 2.35): finite, non-zero arguments take
 * (float)sqrt((double)x*x + (double)y*y); inf/NaN/zero special cases as IEEE hypot. */
float ggo_hypotf(float x, float y)
{
    if (isinf(x) || isinf(y)) return INFINITY;
    if (isnan(x) || isnan(y)) return x + y;
    double dx = (double)x, dy = (double)y;
    return (float)sqrt(dx * dx + dy * dy);
}

void ggo_default_config(ggo_config *c)
{
    /* cfg/GroundGrid.cfg:8-21 */
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

/* ------------------------------------------------------------------------------------------ */
/* geometry: grid_map_core 1.6.x (third party, not under /root/reference)                     */
/* ------------------------------------------------------------------------------------------ */

/* GridMapMath.cpp getIndexFromPosition (value) + checkIfPositionWithinMap (return) as used by
 * GridMap::getIndex / GridMap::isInside at src/GroundSegmentation.cpp:228,230,261.
 *   indexVector = ((position - offset - mapPosition) / resolution), offset = 0.5 * mapLength
 *   index = (-indexVector).cast<int>()            (start index 0: GroundGrid.cpp:143)
 *   inside: t = -I * (position - mapPosition - offset); 0 <= t < mapLength per axis          */
int ggo_get_index(const ggo_map *m, double px, double py, int *row, int *col)
{
    const double offx = 0.5 * m->length[0];
    const double offy = 0.5 * m->length[1];
    const double ivx = ((px - offx) - m->position[0]) / m->resolution;
    const double ivy = ((py - offy) - m->position[1]) / m->resolution;
    *row = trunc_to_int(-ivx);
    *col = trunc_to_int(-ivy);
    const double ax = (px - m->position[0]) - offx;
    const double ay = (py - m->position[1]) - offy;
    /* Eigen 2x2 (-Identity).cast<double>() * vector, written out */
    const double tx = -1.0 * ax + 0.0 * ay;
    const double ty = 0.0 * ax + -1.0 * ay;
    return tx >= 0.0 && ty >= 0.0 && tx < m->length[0] && ty < m->length[1];
}

static void fill(float *p, size_t n, float v)
{
    for (size_t i = 0; i < n; ++i) p[i] = v;
}

void ggo_map_reset_state(ggo_map *m, double pos_x, double pos_y, float odom_z)
{
    const size_t C = (size_t)m->rows * (size_t)m->cols;
    m->position[0] = pos_x;
    m->position[1] = pos_y;
    /* src/GroundGrid.cpp:71-75 */
    fill(m->layer[GGO_POINTS], C, 0.0f);
    fill(m->layer[GGO_GROUND], C, odom_z);
    fill(m->layer[GGO_GROUNDPATCH], C, (float)0.0000001);
    fill(m->layer[GGO_MINGROUNDHEIGHT], C, (float)100.0);
    fill(m->layer[GGO_MAXGROUNDHEIGHT], C, (float)-100.0);
    /* layers added later by filter_cloud (:61-75) start at 0 */
    fill(m->layer[GGO_GROUNDCANDIDATES], C, 0.0f);
    fill(m->layer[GGO_PLANEDIST], C, 0.0f);
    fill(m->layer[GGO_M2], C, 0.0f);
    fill(m->layer[GGO_MEANVARIANCE], C, 0.0f);
    fill(m->layer[GGO_POINTSRAW], C, 0.0f);
    fill(m->layer[GGO_VARIANCE], C, 0.0f);
}

ggo_map *ggo_map_create(float length_f, float resolution_f, double pos_x, double pos_y, float odom_z)
{
    /* grid_map::GridMap::setGeometry(Length(l,l), resolution, position) as called at
     * src/GroundGrid.cpp:58: size = round(length / resolution) in double; length_ = size * res */
    const double res = (double)resolution_f;
    const double len = (double)length_f;
    const int n = (int)round(len / res);

    /* GroundSegmentation::init (src/GroundSegmentation.cpp:38): the nodelet passes 120.0f into a
     * size_t parameter (GroundGridNodelet.cpp:95); size_t / float -> float; std::round(float). */
    const size_t dimension = (size_t)length_f;
    const size_t cellCount = (size_t)roundf((float)dimension / resolution_f);
    if (n <= 0 || (size_t)n != cellCount) return NULL;

    ggo_map *m = (ggo_map *)calloc(1, sizeof(ggo_map));
    m->rows = n;
    m->cols = n;
    m->resolution = res;
    m->length[0] = (double)n * res;
    m->length[1] = (double)n * res;
    m->verticalPointAngDist = (float)(0.00174532925 * 2); /* GroundSegmentation.h:69 */
    m->minDistSquared = 12.0f;                            /* GroundSegmentation.h:70 */
    const size_t C = (size_t)n * (size_t)n;
    for (int l = 0; l < GGO_NUM_LAYERS; ++l) m->layer[l] = (float *)malloc(C * sizeof(float));
    m->expectedPoints = (float *)malloc(C * sizeof(float));

    /* src/GroundSegmentation.cpp:40-46 */
    for (size_t i = 0; i < cellCount; ++i) {
        for (size_t j = 0; j < cellCount; ++j) {
            const float dist = (float)hypot((double)i - (double)cellCount / 2.0,
                                            (double)j - (double)cellCount / 2.0);
            m->expectedPoints[i + j * cellCount] = atanf(1 / dist) / m->verticalPointAngDist;
        }
    }
    ggo_map_reset_state(m, pos_x, pos_y, odom_z);
    return m;
}

void ggo_map_destroy(ggo_map *m)
{
    if (!m) return;
    for (int l = 0; l < GGO_NUM_LAYERS; ++l) free(m->layer[l]);
    free(m->expectedPoints);
    free(m);
}

/* ------------------------------------------------------------------------------------------ */
/* N1: GroundGrid::update (src/GroundGrid.cpp:83-147) + grid_map::GridMap::move / getPosition /  */
/*     convertToDefaultStartIndex (grid_map_core 1.6.x) + tf2 doTransform of a point            */
/* ------------------------------------------------------------------------------------------ */
int ggo_map_update(ggo_map *m, double odom_x, double odom_y, const double plane[4], int shift[2])
{
    const int n[2] = {m->rows, m->cols};
    const double res = m->resolution;
    /* GridMapMath.cpp getIndexShiftFromPositionShift: round half away from zero, then map frame -> buffer order */
    const double odom[2] = {odom_x, odom_y};
    int s[2];
    for (int i = 0; i < 2; ++i) {
        const double tmp = (odom[i] - m->position[i]) / res;
        const int v = (int)(tmp + 0.5 * (tmp > 0 ? 1 : -1));
        s[i] = -v;
    }
    if (shift) {
        shift[0] = s[0];
        shift[1] = s[1];
    }
    if (s[0] == 0 && s[1] == 0) return 0; /* GroundGrid.cpp:135-137: damage empty, nothing to do */

    /* getPositionShiftFromIndexShift: position += (-indexShift) * resolution (snapped, not the odometry position) */
    m->position[0] += (double)(-s[0]) * res;
    m->position[1] += (double)(-s[1]) * res;

    /* third row of the rotation + translation z of base_link <- map (ggo_rotation_from_quaternion picks the convention) */
    const double m20 = plane[0], m21 = plane[1], m22 = plane[2], tz = plane[3];

    const size_t C = (size_t)n[0] * (size_t)n[1];
    float *tmp = (float *)malloc(C * sizeof(float));
    const double first0 = 0.5 * m->length[0] - 0.5 * res, first1 = 0.5 * m->length[1] - 0.5 * res; /* getVectorToFirstCell */
    for (int l = 0; l < GGO_NUM_LAYERS; ++l) {
        float *L = m->layer[l];
        for (int j = 0; j < n[1]; ++j) {
            for (int i = 0; i < n[0]; ++i) {
                /* convertToDefaultStartIndex: new (unwrapped) index (i,j) <- buffer index ((i + s0) mod n, (j + s1) mod n);
                 * the rows / cols move() dropped from the buffer are the newly exposed cells */
                const int all0 = abs(s[0]) >= n[0], all1 = abs(s[1]) >= n[1];
                int bi = (i + s[0]) % n[0], bj = (j + s[1]) % n[1];
                if (bi < 0) bi += n[0];
                if (bj < 0) bj += n[1];
                const int new0 = all0 || all1 || (s[0] > 0 ? bi < s[0] : (s[0] < 0 ? bi >= n[0] + s[0] : 0));
                const int new1 = all0 || all1 || (s[1] > 0 ? bj < s[1] : (s[1] < 0 ? bj >= n[1] + s[1] : 0));
                float v;
                if (new0 || new1) {
                    if (l == GGO_GROUND) {
                        /* getPositionFromIndex: position = mapPosition + offset + resolution * (-index) */
                        const double px = (m->position[0] + first0) + res * (double)(-i);
                        const double py = (m->position[1] + first1) + res * (double)(-j);
                        /* doTransform: v_out.z = (m20 * x + m21 * y + m22 * 0) + origin.z; ground = -z (:130) */
                        const double z = ((m20 * px + m21 * py) + m22 * 0.0) + tz;
                        v = (float)(-z);
                    } else if (l == GGO_GROUNDPATCH) {
                        v = 0.0f; /* :131 */
                    } else {
                        v = NAN; /* grid_map::GridMap::clearRows / clearCols on every layer */
                    }
                } else {
                    v = L[(size_t)bi + (size_t)bj * n[0]];
                }
                tmp[(size_t)i + (size_t)j * n[0]] = v;
            }
        }
        memcpy(L, tmp, C * sizeof(float));
    }
    free(tmp);
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* R2: filter_cloud prologue (src/GroundSegmentation.cpp:61-75)                               */
/* ------------------------------------------------------------------------------------------ */
void ggo_stage_reset(ggo_map *m)
{
    const size_t C = (size_t)m->rows * (size_t)m->cols;
    fill(m->layer[GGO_GROUNDCANDIDATES], C, 0.0f); /* :61,70 */
    fill(m->layer[GGO_PLANEDIST], C, 0.0f);        /* :62 */
    fill(m->layer[GGO_M2], C, 0.0f);               /* :63 */
    fill(m->layer[GGO_MEANVARIANCE], C, 0.0f);     /* :64 */
    fill(m->layer[GGO_POINTSRAW], C, 0.0f);        /* :67 */
    fill(m->layer[GGO_POINTS], C, 0.0f);           /* :71 */
    fill(m->layer[GGO_MINGROUNDHEIGHT], C, FLT_MAX); /* :72 */
    fill(m->layer[GGO_MAXGROUNDHEIGHT], C, FLT_MIN); /* :73 numeric_limits<float>::min() (sic) */
    fill(m->layer[GGO_VARIANCE], C, 0.0f);         /* :75 */
}

/* ------------------------------------------------------------------------------------------ */
/* R3 + R4: insert_cloud (src/GroundSegmentation.cpp:200-311), start = 0, end = n             */
/* ------------------------------------------------------------------------------------------ */
void ggo_stage_insert(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n,
                      const float origin[3], uint8_t *cls, int32_t *cell)
{
    const int size0 = m->rows, size1 = m->cols;
    const float ox = origin[0], oy = origin[1], oz = origin[2];

    for (size_t i = 0; i < n; ++i) {
        const ggo_point *point = &cloud[i];
        /* :222-223 */
        const double posx = (double)point->x, posy = (double)point->y;
        const float dx = point->x - ox, dy = point->y - oy;
        const float sqdist = (float)((double)dx * (double)dx + (double)dy * (double)dy);

        int gi0, gi1;
        const int inside = ggo_get_index(m, posx, posy, &gi0, &gi1); /* :228,230 */
        if (!inside || gi0 < 0 || gi1 < 0 || gi0 >= size0 || gi1 >= size1) {
            /* (index out of range with isInside true is UB in the reference; cannot happen for
             * finite inputs away from the last representable double below the map edge) */
            if (cls) cls[i] = GGO_OUTSIDE;
            if (cell) cell[i] = -1;
            continue; /* :231 */
        }
        if (cell) cell[i] = gi0 + gi1 * size0;

        L(m, GGO_POINTSRAW, gi0, gi1) += 1.0f; /* :234 */

        if ((int)point->ring > cfg->max_ring || sqdist < m->minDistSquared) { /* :237 */
            if (cls) cls[i] = GGO_IGNORED;
            continue;
        }

        /* Outlier detection test :243-275 */
        int toSkip = 0;
        const float oldgroundheight = L(m, GGO_GROUND, gi0, gi1);
        if ((double)point->z < (double)oldgroundheight - 0.2) { /* :244 */
            float vx = point->x - ox; /* :248-250 */
            float vy = point->y - oy;
            float vz = point->z - oz;
            const float len = sqrtf(vx * vx + vy * vy + vz * vz); /* :252 pow(float,2.0f) -> x*x */
            vx /= len; /* :253-255 */
            vy /= len;
            vz /= len;
            const double len2 = (double)len * (double)len;
            for (int step = 3; step < GGO_WALK_MAX_STEP; ++step) { /* :258; the bound is the library's (see gg_oracle.h) */
                const float sx = (float)step * vx, sy = (float)step * vy, sz = (float)step * vz;
                const double d2 = (double)sx * (double)sx + (double)sy * (double)sy + (double)sz * (double)sz;
                if (!(d2 < len2 && vz < -0.01f)) break;
                /* :260-261 */
                const float ipx = sx + ox, ipy = sy + oy;
                int I0, I1;
                (void)ggo_get_index(m, (double)ipx, (double)ipy, &I0, &I1);
                if (I0 <= 0 || I1 <= 0 || I0 >= size0 - 1 || I1 >= size1 - 1) continue; /* :264-265 */
                /* :268 block<3,3>(max(I0-1,2), max(I1-1,2)) of groundpatch */
                const int r0 = std_max_i(I0 - 1, 2), c0 = std_max_i(I1 - 1, 2);
                float e[9];
                for (int s = 0; s < 9; ++s) e[s] = L(m, GGO_GROUNDPATCH, r0 + s % 3, c0 + s / 3);
                const float bsum = tree9(e);
                /* :269 */
                if ((double)bsum > cfg->min_outlier_detection_ground_confidence &&
                    L(m, GGO_GROUNDPATCH, I0, I1) > 0.01f &&
                    (double)L(m, GGO_GROUND, I0, I1) >= (double)(sz + oz) + cfg->outlier_tolerance) {
                    toSkip = 1;
                    break;
                }
            }
        }
        if (toSkip) {
            if (cls) cls[i] = GGO_OUTLIER;
            continue; /* :278-279 */
        }
        if (cls) cls[i] = GGO_KEPT;

        /* :282-309 */
        float *groundheight = &L(m, GGO_GROUNDCANDIDATES, gi0, gi1);
        float *mean = &L(m, GGO_MEANVARIANCE, gi0, gi1);
        float *points = &L(m, GGO_POINTS, gi0, gi1);
        float *maxHeight = &L(m, GGO_MAXGROUNDHEIGHT, gi0, gi1);
        float *minHeight = &L(m, GGO_MINGROUNDHEIGHT, gi0, gi1);
        float *planeDistMap = &L(m, GGO_PLANEDIST, gi0, gi1);
        float *m2 = &L(m, GGO_M2, gi0, gi1);

        const float planeDist = point->z - oz;                                                   /* :295 */
        *groundheight = (float)((double)(point->z + *points * *groundheight) / ((double)*points + 1.0)); /* :296 */

        if ((double)*mean == 0.0) *mean = planeDist; /* :298-299 */
        if (!isnan(planeDist)) {                     /* :300 */
            const float delta = planeDist - *mean;   /* :301 */
            *mean += delta / (*points + 1);          /* :302 float / float */
            *planeDistMap = (float)((double)(planeDist + *points * *planeDistMap) / ((double)*points + 1.0)); /* :303 */
            *m2 += delta * (planeDist - *mean);      /* :304 */
        }
        *maxHeight = std_max_f(*maxHeight, point->z);           /* :307 */
        *minHeight = std_min_f(*minHeight, point->z - 0.0001f); /* :308 */
        *points = (float)((double)*points + 1.0);               /* :309 */
    }
}

/* ------------------------------------------------------------------------------------------ */
/* R5 + R6: detect_ground_patches / detect_ground_patch<S> (src/GroundSegmentation.cpp:314-395)*/
/* ------------------------------------------------------------------------------------------ */
GGO_ALWAYS_INLINE void detect_ground_patch(ggo_map *m, const ggo_config *cfg, const int S, size_t i, size_t j)
{
    const int center_idx = S / 2; /* :352 floor(S/2) with integer division */
    const int r0 = (int)i - center_idx, c0 = (int)j - center_idx;
    const int size0 = m->rows, size1 = m->cols;
    const float resolution = (float)m->resolution; /* :351 static const float */
    float pts[25], var[25], mn[25], prod[25];
    const int SS = S * S;
    for (int s = 0; s < SS; ++s) /* :355 block, column-major linear index */
        pts[s] = L(m, GGO_POINTS, r0 + s % S, c0 + s / S);
    /* :356 */
    const double di = (double)i - (double)size0 / 2.0, dj = (double)j - (double)size1 / 2.0;
    const float sqdist = (float)((di * di + dj * dj) * ((double)resolution * (double)resolution));
    const int patchSize = S;
    const float expected = m->expectedPoints[i + j * (size_t)m->rows]; /* :358 */
    const float pointsblockSum = treeSS(pts, S);                 /* :359 */
    float *oldConfidence = &L(m, GGO_GROUNDPATCH, i, j);                /* :360 */
    float *oldGroundheight = &L(m, GGO_GROUND, i, j);                   /* :361 */

    /* :364-365 */
    if ((double)pointsblockSum <
        std_max_d(floor(cfg->ground_patch_detection_minimum_point_count_threshold * (double)patchSize * (double)expected), 3.0))
        return;

    for (int s = 0; s < SS; ++s) { /* :370,371 blocks */
        var[s] = L(m, GGO_VARIANCE, r0 + s % S, c0 + s / S);
        mn[s] = L(m, GGO_MINGROUNDHEIGHT, r0 + s % S, c0 + s / S);
    }
    /* :369 */
    const double df2 = cfg->distance_factor * cfg->distance_factor;
    const double mdf2 = cfg->minimum_distance_factor * cfg->minimum_distance_factor;
    const double mdf10 = cfg->minimum_distance_factor * 10;
    const float varThresholdsq = (float)std_min_d(std_max_d((double)sqdist * df2, mdf2), mdf10 * mdf10);
    const float variance = var[center_idx + center_idx * S]; /* :372 */
    float localmin = mn[0];                                   /* :373 minCoeff (no NaN can occur in this layer) */
    for (int s = 1; s < SS; ++s)
        if (mn[s] < localmin) localmin = mn[s];
    /* :374 */
    float maxVar;
    if (pts[center_idx + center_idx * S] >= (float)cfg->point_count_cell_variance_threshold) {
        maxVar = variance;
    } else {
        for (int s = 0; s < SS; ++s) prod[s] = pts[s] * var[s];
        maxVar = treeSS(prod, S) / pointsblockSum;
    }
    /* :375 */
    for (int s = 0; s < SS; ++s) prod[s] = pts[s] * mn[s];
    const float groundlevel = treeSS(prod, S) / pointsblockSum;
    /* :376 */
    const float groundDiff = std_max_f((groundlevel - *oldGroundheight) * (2.0f * *oldConfidence), 1.0f);

    /* :379-380 */
    if ((double)*oldConfidence > 0.5 && (double)groundlevel >= (double)*oldGroundheight + cfg->outlier_tolerance)
        return;

    /* :382 */
    if ((double)varThresholdsq > (double)maxVar * (double)maxVar && maxVar > 0 &&
        (double)pointsblockSum >
            (double)((groundDiff * expected) * (float)patchSize) * cfg->ground_patch_detection_minimum_point_count_threshold) {
        /* :383 */
        const float newConfidence = (float)std_min_d((double)pointsblockSum / cfg->occupied_cells_point_count_factor, 1.0);
        /* :385 */
        *oldGroundheight = (groundlevel * newConfidence + (*oldConfidence * *oldGroundheight) * 2) /
                           (newConfidence + *oldConfidence * 2);
        /* :387 */
        *oldConfidence = (float)std_min_d(
            ((double)pointsblockSum / (cfg->occupied_cells_point_count_factor * (double)2.0f) + (double)*oldConfidence) / 2.0, 1.0);
    } else if (localmin < *oldGroundheight) { /* :389 */
        *oldGroundheight = localmin;                                /* :391 */
        *oldConfidence = std_min_f(*oldConfidence + 0.1f, 0.5f);    /* :393 */
    }
}

/* :323 variance = m2 ./ (points + FLT_MIN): every call of detect_ground_patches recomputes the whole layer */
static void detect_variance(ggo_map *m)
{
    const size_t C = (size_t)m->rows * (size_t)m->cols;
    for (size_t k = 0; k < C; ++k)
        m->layer[GGO_VARIANCE][k] = m->layer[GGO_M2][k] / (m->layer[GGO_POINTS][k] + FLT_MIN);
}

/* detect_ground_patches(map, section) (src/GroundSegmentation.cpp:314-340): one quadrant, 0: top-left .. 3: bottom-right */
void ggo_stage_detect_section(ggo_map *m, const ggo_config *cfg, unsigned short section)
{
    detect_variance(m);
    const int size0 = m->rows, size1 = m->cols;
    const float resolution = (float)m->resolution; /* :318 */
    const int gcols = m->cols, grows = m->rows;
    const int cols_start = 2 + section % 2 * (gcols / 2 - 2);             /* :325 */
    const int rows_start = section >= 2 ? grows / 2 : 2;                   /* :326 */
    const int cols_end = (gcols) / 2 + section % 2 * (gcols / 2 - 2);      /* :327 */
    const int rows_end = section >= 2 ? grows - 2 : (grows) / 2;           /* :328 */
    for (int i = cols_start; i < cols_end; ++i) {
        for (int j = rows_start; j < rows_end; ++j) {
            /* :332 */
            const double di = (double)i - (double)size0 / 2.0, dj = (double)j - (double)size1 / 2.0;
            const float sqdist = (float)((di * di + dj * dj) * ((double)resolution * (double)resolution));
            /* :334 */
            if ((double)sqdist <= cfg->patch_size_change_distance * cfg->patch_size_change_distance)
                detect_ground_patch(m, cfg, 3, (size_t)i, (size_t)j);
            else
                detect_ground_patch(m, cfg, 5, (size_t)i, (size_t)j);
        }
    }
}

void ggo_stage_detect(ggo_map *m, const ggo_config *cfg)
{
    /* :130-131, order irrelevant: no inter-cell dependency (a cell reads neighbours' points / variance / minGroundHeight and
     * writes only its own ground / groundpatch) */
    for (unsigned short section = 0; section < 4; ++section) ggo_stage_detect_section(m, cfg, section);
}

/* detect_ground_patch<S>(map, i, j) (:343-395) on its own: S = 3 or 5; reads `variance` as it stands (the caller ran :323) */
void ggo_detect_ground_patch(ggo_map *m, const ggo_config *cfg, int S, size_t i, size_t j)
{
    if (S == 3)
        detect_ground_patch(m, cfg, 3, i, j);
    else
        detect_ground_patch(m, cfg, 5, i, j);
}

/* ------------------------------------------------------------------------------------------ */
/* R7 + R8: spiral_ground_interpolation / interpolate_cell (src/GroundSegmentation.cpp:398-465)*/
/* ------------------------------------------------------------------------------------------ */
GGO_ALWAYS_INLINE void interpolate_cell(ggo_map *m, const ggo_config *cfg, const size_t x, const size_t y)
{
    const int center_idx = m->rows / 2 - 1; /* :447 */
    float g[9], w[9], prod[9];
    for (int s = 0; s < 9; ++s) { /* :453,458 block<3,3>(x-1,y-1) */
        w[s] = L(m, GGO_GROUNDPATCH, x - 1 + s % 3, y - 1 + s / 3);
        g[s] = L(m, GGO_GROUND, x - 1 + s % 3, y - 1 + s / 3);
    }
    float *height = &L(m, GGO_GROUND, x, y);        /* :455 */
    float *occupied = &L(m, GGO_GROUNDPATCH, x, y); /* :456 */
    const float gvlSum = tree9(w) + FLT_MIN; /* :457 */
    for (int s = 0; s < 9; ++s) prod[s] = w[s] * g[s];
    const float avg = tree9(prod) / gvlSum;  /* :458 */

    *height = (1.0f - *occupied) * avg + *occupied * *height; /* :460 */

    /* :463-464 */
    const float fx = (float)x - (float)center_idx, fy = (float)y - (float)center_idx;
    const double d2 = ((double)fx * (double)fx + (double)fy * (double)fy) * (m->resolution * m->resolution);
    if (d2 > (double)m->minDistSquared)
        *occupied = (float)std_max_d((double)*occupied - (double)*occupied / cfg->occupied_cells_decrease_factor, 0.001);
}

void ggo_stage_spiral(ggo_map *m, const ggo_config *cfg, double base_z)
{
    const int center_idx = m->rows / 2 - 1; /* :403 */
    L(m, GGO_GROUNDPATCH, center_idx, center_idx) = 1.0f;      /* :405 */
    /* :406-411: tf2::doTransform of the zero point == the transform's translation */
    L(m, GGO_GROUND, center_idx, center_idx) = (float)base_z;

    for (int i = center_idx - 1; i >= 1; --i) { /* :413 */
        int rectangle_pos = i;                                   /* :415 */
        const int side_length = (center_idx - rectangle_pos) * 2; /* :418 */
        for (short side = 0; side < 2; ++side) {                 /* :421 */
            for (int pos = rectangle_pos; pos < rectangle_pos + side_length; ++pos) {
                const int x = side % 2 ? pos : rectangle_pos;
                const int y = side % 2 ? rectangle_pos : pos;
                interpolate_cell(m, cfg, (size_t)x, (size_t)y);
            }
        }
        rectangle_pos += side_length;                            /* :431 */
        for (short side = 0; side < 2; ++side) {                 /* :432 */
            for (int pos = rectangle_pos; pos >= rectangle_pos - side_length; --pos) {
                const int x = side % 2 ? pos : rectangle_pos;
                const int y = side % 2 ? rectangle_pos : pos;
                interpolate_cell(m, cfg, (size_t)x, (size_t)y);
            }
        }
    }
}

/* interpolate_cell(map, x, y) (:445-465) on its own */
void ggo_interpolate_cell(ggo_map *m, const ggo_config *cfg, size_t x, size_t y) { interpolate_cell(m, cfg, x, y); }

size_t ggo_spiral_visit_count(int rows)
{
    const int center_idx = rows / 2 - 1;
    size_t cnt = 0;
    for (int i = center_idx - 1; i >= 1; --i) {
        const int side_length = (center_idx - i) * 2;
        cnt += 2 * (size_t)side_length + 2 * ((size_t)side_length + 1);
    }
    return cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* filter_cloud (src/GroundSegmentation.cpp:50-197)                                           */
/* ------------------------------------------------------------------------------------------ */
size_t ggo_filter_cloud(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n,
                        const float origin[3], double base_z,
                        ggo_point *out_points, uint8_t *out_label, int32_t *out_index,
                        uint8_t *out_class, int32_t *out_cell)
{
    uint8_t *cls = out_class ? out_class : (uint8_t *)malloc(n ? n : 1);
    int32_t *cell = out_cell ? out_cell : (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));

    ggo_stage_reset(m);                              /* :61-75 */
    ggo_stage_insert(m, cfg, cloud, n, origin, cls, cell); /* :101-117 with one thread */
    ggo_stage_detect(m, cfg);                        /* :130-134 */
    ggo_stage_spiral(m, cfg, base_z);                /* :142 */

    const size_t C = (size_t)m->rows * (size_t)m->cols;
    fill(m->layer[GGO_POINTS], C, 0.0f);             /* :147 */

    /* :154-156 */
    const double min_dist_fac = cfg->minimum_distance_factor * 5;
    const double min_point_height_thres = cfg->miminum_point_height_threshold;
    const double min_point_height_obs_thres = cfg->minimum_point_height_obstacle_threshold;
    const int size0 = m->rows, size1 = m->cols;
    const float ox = origin[0], oy = origin[1];

    if (out_label) memset(out_label, GGO_DROPPED, n);
    if (out_index)
        for (size_t i = 0; i < n; ++i) out_index[i] = -1;

    size_t out_n = 0;
    /* :150,158: point_index (kept, cloud order) then ignored (cloud order) */
    for (int pass = 0; pass < 2; ++pass) {
        const uint8_t want = pass == 0 ? GGO_KEPT : GGO_IGNORED;
        for (size_t i = 0; i < n; ++i) {
            if (cls[i] != want) continue;
            const ggo_point *point = &cloud[i];
            const int gi0 = cell[i] % size0, gi1 = cell[i] / size0;
            const double groundheight = (double)L(m, GGO_GROUND, gi0, gi1); /* :162 */
            const float variance = L(m, GGO_VARIANCE, gi0, gi1);           /* :165 */
            if (size0 <= gi0 + 3 || size1 <= gi1 + 3) continue;            /* :167-168 */
            const float dist = ggo_hypotf(point->x - ox, point->y - oy);   /* :170 */
            /* :171 */
            const double tolerance = std_max_d(
                std_min_d((min_dist_fac * (double)dist) / (double)variance * min_point_height_thres, min_point_height_thres),
                min_point_height_obs_thres);
            uint8_t label;
            if (tolerance + groundheight < (double)point->z) { /* :173 */
                label = GGO_NONGROUND_LABEL;                   /* :175 */
                L(m, GGO_POINTS, gi0, gi1) += 1.0f;            /* :176 */
            } else {
                label = GGO_GROUND_LABEL;                      /* :180 */
            }
            if (out_points) {
                out_points[out_n] = *point;
                out_points[out_n].intensity = (float)label;
            }
            if (out_label) out_label[i] = label;
            if (out_index) out_index[i] = (int32_t)out_n;
            ++out_n;
        }
    }
    /* :185-189 outliers, cloud order */
    for (size_t i = 0; i < n; ++i) {
        if (cls[i] != GGO_OUTLIER) continue;
        if (out_points) {
            out_points[out_n] = cloud[i];
            out_points[out_n].intensity = 49;
        }
        if (out_label) out_label[i] = GGO_GROUND_LABEL;
        if (out_index) out_index[i] = (int32_t)out_n;
        ++out_n;
    }

    if (!out_class) free(cls);
    if (!out_cell) free(cell);
    return out_n;
}

/* ------------------------------------------------------------------------------------------ */
/* TIMING ONLY: the reference's default threading shape (see gg_oracle.h)                     */
/* ------------------------------------------------------------------------------------------ */
struct insert_job {
    ggo_map *m;
    const ggo_config *cfg;
    const ggo_point *cloud;
    size_t start, end;
    const float *origin;
    uint8_t *cls;
    int32_t *cell;
};
static void *insert_thread(void *p)
{
    struct insert_job *j = (struct insert_job *)p;
    /* :101-106 insert_cloud(cloud, start, end, ...) on the shared map: same statements as ggo_stage_insert on a sub-range */
    ggo_stage_insert(j->m, j->cfg, j->cloud + j->start, j->end - j->start, j->origin, j->cls + j->start, j->cell + j->start);
    return NULL;
}
struct detect_job {
    ggo_map *m;
    const ggo_config *cfg;
    int section;
};
static void detect_section(ggo_map *m, const ggo_config *cfg, int section)
{
    const int size0 = m->rows, size1 = m->cols;
    const float resolution = (float)m->resolution;
    const int gcols = m->cols, grows = m->rows;
    const int cols_start = 2 + section % 2 * (gcols / 2 - 2);             /* :325 */
    const int rows_start = section >= 2 ? grows / 2 : 2;                   /* :326 */
    const int cols_end = (gcols) / 2 + section % 2 * (gcols / 2 - 2);      /* :327 */
    const int rows_end = section >= 2 ? grows - 2 : (grows) / 2;           /* :328 */
    for (int i = cols_start; i < cols_end; ++i)
        for (int j = rows_start; j < rows_end; ++j) {
            const double di = (double)i - (double)size0 / 2.0, dj = (double)j - (double)size1 / 2.0;
            const float sqdist = (float)((di * di + dj * dj) * ((double)resolution * (double)resolution));
            if ((double)sqdist <= cfg->patch_size_change_distance * cfg->patch_size_change_distance)
                detect_ground_patch(m, cfg, 3, (size_t)i, (size_t)j);
            else
                detect_ground_patch(m, cfg, 5, (size_t)i, (size_t)j);
        }
}
static void *detect_thread(void *p)
{
    struct detect_job *j = (struct detect_job *)p;
    detect_section(j->m, j->cfg, j->section);
    return NULL;
}

size_t ggo_filter_cloud_threads(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n, const float origin[3],
                                double base_z, int t_insert, uint8_t *out_label)
{
    if (t_insert < 1) t_insert = 1;
    if (t_insert > 64) t_insert = 64;
    uint8_t *cls = (uint8_t *)malloc(n ? n : 1);
    int32_t *cell = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    ggo_stage_reset(m); /* :61-75 */
    {                   /* :98-109: thread_count threads over consecutive point ranges, joined before the detection */
        pthread_t th[64];
        struct insert_job job[64];
        const size_t per = n / (size_t)t_insert;
        for (int t = 0; t < t_insert; ++t) {
            job[t].m = m;
            job[t].cfg = cfg;
            job[t].cloud = cloud;
            job[t].start = per * (size_t)t;
            job[t].end = t == t_insert - 1 ? n : per * (size_t)(t + 1);
            job[t].origin = origin;
            job[t].cls = cls;
            job[t].cell = cell;
            pthread_create(&th[t], NULL, insert_thread, &job[t]);
        }
        for (int t = 0; t < t_insert; ++t) pthread_join(th[t], NULL);
    }
    { /* :323 (done by every detection thread in the reference), then :128-134 four quadrant threads */
        const size_t C = (size_t)m->rows * (size_t)m->cols;
        for (size_t k = 0; k < C; ++k) m->layer[GGO_VARIANCE][k] = m->layer[GGO_M2][k] / (m->layer[GGO_POINTS][k] + FLT_MIN);
        pthread_t th[4];
        struct detect_job job[4];
        for (int s4 = 0; s4 < 4; ++s4) {
            job[s4].m = m;
            job[s4].cfg = cfg;
            job[s4].section = s4;
            pthread_create(&th[s4], NULL, detect_thread, &job[s4]);
        }
        for (int s4 = 0; s4 < 4; ++s4) pthread_join(th[s4], NULL);
    }
    ggo_stage_spiral(m, cfg, base_z); /* :142 */
    const size_t C = (size_t)m->rows * (size_t)m->cols;
    fill(m->layer[GGO_POINTS], C, 0.0f); /* :147 */
    const double min_dist_fac = cfg->minimum_distance_factor * 5;
    const double thres = cfg->miminum_point_height_threshold, obs = cfg->minimum_point_height_obstacle_threshold;
    const int size0 = m->rows, size1 = m->cols;
    size_t out_n = 0;
    if (out_label) memset(out_label, GGO_DROPPED, n);
    for (int pass = 0; pass < 2; ++pass) { /* :150-183 */
        const uint8_t want = pass == 0 ? GGO_KEPT : GGO_IGNORED;
        for (size_t i = 0; i < n; ++i) {
            if (cls[i] != want) continue;
            const ggo_point *point = &cloud[i];
            const int gi0 = cell[i] % size0, gi1 = cell[i] / size0;
            if (size0 <= gi0 + 3 || size1 <= gi1 + 3) continue;
            const double groundheight = (double)L(m, GGO_GROUND, gi0, gi1);
            const float variance = L(m, GGO_VARIANCE, gi0, gi1);
            const float dist = ggo_hypotf(point->x - origin[0], point->y - origin[1]);
            const double tolerance = std_max_d(std_min_d((min_dist_fac * (double)dist) / (double)variance * thres, thres), obs);
            uint8_t label = GGO_GROUND_LABEL;
            if (tolerance + groundheight < (double)point->z) {
                label = GGO_NONGROUND_LABEL;
                L(m, GGO_POINTS, gi0, gi1) += 1.0f;
            }
            if (out_label) out_label[i] = label;
            ++out_n;
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (cls[i] == GGO_OUTLIER) {
            if (out_label) out_label[i] = GGO_GROUND_LABEL;
            ++out_n;
        }
    free(cls);
    free(cell);
    return out_n;
}
