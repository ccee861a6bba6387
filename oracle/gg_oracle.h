/*
 * gg_oracle.h -- CPU restatement of GroundGrid's per-cloud hot path (TEST INFRASTRUCTURE).
 *
 * This is the checker, not the product.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may link or load it.  The product (libgroundgrid_hip.so)
 * never calls into it and has no CPU fallback.
 *
 * PARITY UNPINNED: the reference (dcmlr/groundgrid) ships no tests, golden vectors or
 * fixtures for this path, and it cannot be built in this image (it needs ROS Noetic,
 * grid_map_core, PCL, Eigen, tf2 -- none present, no network).  This restatement follows
 * the reference source text line by line (citations below are into /root/reference) and
 * restates the two third-party pieces of arithmetic it relies on from their published
 * sources: grid_map_core 1.6.x (GridMapMath.cpp getIndexFromPosition /
 * checkIfPositionWithinMap, GridMap::setGeometry) and Eigen 3.3.7 (Redux.h
 * redux_novec_unroller, the fixed-size Block<...,S,S>::sum() order).  Neither version
 * is pinned by the reference (package.xml:29, CMakeLists.txt:41).
 *
 * Deterministic configuration: the reference is only deterministic at thread_count = 1
 * (its multi-thread insertion is an unsynchronised data race,
 * src/GroundSegmentation.cpp:101-106 -> :282-309).  The oracle IS thread_count = 1.
 */
#ifndef GG_ORACLE_H
#define GG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* include/velodyne_pointcloud/point_types.h:27-33  (PCL_ADD_POINT4D + intensity + ring, 16-B aligned) */
typedef struct ggo_point {
    float x, y, z, pad0;
    float intensity;
    uint16_t ring;
    uint16_t pad1;
    uint32_t pad2[2];
} ggo_point; /* 32 bytes */

/* cfg/GroundGrid.cfg:8-21 (dynamic_reconfigure: int_t -> int, double_t -> double) */
typedef struct ggo_config {
    int point_count_cell_variance_threshold;                      /* 10 */
    int max_ring;                                                 /* 1024 */
    double groundpatch_detection_minimum_threshold;               /* 0.01 (unused by the path) */
    double distance_factor;                                       /* 0.0001 */
    double minimum_distance_factor;                               /* 0.0005 */
    double miminum_point_height_threshold;                        /* 0.3 (sic) */
    double minimum_point_height_obstacle_threshold;               /* 0.1 */
    double outlier_tolerance;                                     /* 0.1 */
    double ground_patch_detection_minimum_point_count_threshold;  /* 0.25 */
    double patch_size_change_distance;                            /* 20 */
    double occupied_cells_decrease_factor;                        /* 5 */
    double occupied_cells_point_count_factor;                     /* 20 */
    double min_outlier_detection_ground_confidence;               /* 1.25 */
    int thread_count;                                             /* 8 (ignored: oracle == 1 thread) */
} ggo_config;

enum ggo_layer {
    GGO_POINTS = 0,
    GGO_GROUND = 1,
    GGO_GROUNDPATCH = 2,
    GGO_MINGROUNDHEIGHT = 3,
    GGO_MAXGROUNDHEIGHT = 4,
    GGO_GROUNDCANDIDATES = 5,
    GGO_PLANEDIST = 6,
    GGO_M2 = 7,
    GGO_MEANVARIANCE = 8,
    GGO_POINTSRAW = 9,
    GGO_VARIANCE = 10,
    GGO_NUM_LAYERS = 11
};

/* per-input-point classification (what insert_cloud decides) */
enum ggo_class { GGO_OUTSIDE = 0, GGO_IGNORED = 1, GGO_OUTLIER = 2, GGO_KEPT = 3 };

/* per-input-point label in the output mask; 49/99 are the intensity codes the reference writes */
enum ggo_label { GGO_DROPPED = 0, GGO_GROUND_LABEL = 49, GGO_NONGROUND_LABEL = 99 };

typedef struct ggo_map {
    int rows, cols;        /* grid_map size (setGeometry)            */
    double resolution;     /* (double)resolution_f                   */
    double length[2];      /* size * resolution                      */
    double position[2];    /* map centre in the map frame            */
    float verticalPointAngDist; /* GroundSegmentation.h:69 */
    float minDistSquared;       /* GroundSegmentation.h:70 */
    float *layer[GGO_NUM_LAYERS]; /* column-major rows x cols (Eigen::MatrixXf) */
    float *expectedPoints;        /* R1 table, column-major                     */
} ggo_map;

/* Documented deviation shared with the library: the line-of-sight walk (:258) has no bound in the reference (a corrupt
 * z of -1e9 walks 1e9 steps, past 2^31 its int step overflows); steps >= GGO_WALK_MAX_STEP are not evaluated.  Only
 * points more than 65 km from the sensor can tell the difference. */
#define GGO_WALK_MAX_STEP (1 << 16)

/* Third-party conventions (unpinned, see tools/pin/): which Eigen the reference is built against decides the order of
 * the 5x5 block sums (:359, :374-375).  0 = Eigen 3.3.x redux_novec_unroller (default, ROS Noetic / Ubuntu 20.04),
 * 1 = Eigen 3.4.x SSE2 slice-vectorised reduction.  Process-wide (test infrastructure). */
void ggo_set_eigen_reduction(int order);
int ggo_get_eigen_reduction(void);

/* Quaternion (x,y,z,w) -> row-major rotation: convention 0 = tf2::Matrix3x3::setRotation, 1 = KDL::Rotation::Quaternion
 * (what tf2_geometry_msgs' doTransform(PointStamped) uses in ROS Noetic). */
void ggo_rotation_from_quaternion(int convention, const double q_xyzw[4], double rot[9]);

void ggo_default_config(ggo_config *c);

/* grid_map::GridMap::setGeometry + GroundSegmentation::init + GroundGrid::initGroundGrid state.
 * length_f/resolution_f are the reference's float constants (GroundGrid.h:70-71). Returns NULL
 * if the two cell counts (grid_map's and init()'s) disagree. */
ggo_map *ggo_map_create(float length_f, float resolution_f, double pos_x, double pos_y, float odom_z);
void ggo_map_destroy(ggo_map *m);
/* re-apply GroundGrid.cpp:71-75 initial layer values */
void ggo_map_reset_state(ggo_map *m, double pos_x, double pos_y, float odom_z);

/* GroundSegmentation::filter_cloud (src/GroundSegmentation.cpp:50-197), thread_count = 1.
 *   out_points : capacity n; receives the returned cloud (order: kept, ignored, outliers)
 *   out_label  : per INPUT point, ggo_label            (may be NULL)
 *   out_index  : per INPUT point, position in out_points or -1 (may be NULL)
 *   out_class  : per INPUT point, ggo_class            (may be NULL)
 *   out_cell   : per INPUT point, row + col*rows or -1 if outside (may be NULL)
 * returns number of output points. */
size_t ggo_filter_cloud(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n,
                        const float origin[3], double base_z,
                        ggo_point *out_points, uint8_t *out_label, int32_t *out_index,
                        uint8_t *out_class, int32_t *out_cell);

/* TIMING ONLY -- the reference's default threading shape (cfg/GroundGrid.cfg:21 thread_count = 8,
 * src/GroundSegmentation.cpp:98-134): t_insert threads run insert_cloud on consecutive point ranges of the cloud
 * CONCURRENTLY ON THE SHARED LAYERS, without synchronisation, exactly as the reference does (:101-109) -- a data race there and
 * here, so layers and labels of this variant are NOT deterministic and are never used as a checker; then 4 threads run
 * detect_ground_patches on the four quadrants (:128-134), then the serial sweep and label loop.  Returns the number of
 * output points.  (The per-point lists are concatenated in thread order, :112-117.) */
size_t ggo_filter_cloud_threads(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n, const float origin[3],
                                double base_z, int t_insert, uint8_t *out_label);

/* stage entry points (each follows the reference function of the same name) */
void ggo_stage_reset(ggo_map *m);                                   /* :61-75  */
void ggo_stage_insert(ggo_map *m, const ggo_config *cfg, const ggo_point *cloud, size_t n,
                      const float origin[3], uint8_t *cls, int32_t *cell); /* :200-311 */
void ggo_stage_detect(ggo_map *m, const ggo_config *cfg);           /* :314-395 */
void ggo_stage_spiral(ggo_map *m, const ggo_config *cfg, double base_z); /* :398-465 */
/* the public stage members of include/groundgrid/GroundSegmentation.h:59-62 one by one */
void ggo_stage_detect_section(ggo_map *m, const ggo_config *cfg, unsigned short section); /* detect_ground_patches(map, section), :314-340 */
void ggo_detect_ground_patch(ggo_map *m, const ggo_config *cfg, int S, size_t i, size_t j); /* detect_ground_patch<S>(map, i, j), :343-395 */
void ggo_interpolate_cell(ggo_map *m, const ggo_config *cfg, size_t x, size_t y);         /* interpolate_cell(map, x, y), :445-465 */

/* GroundGrid::update (src/GroundGrid.cpp:83-147) on an already initialised map: grid_map::GridMap::move to the odometry
 * position (snapped to whole cells), newly exposed cells get ground = -(z of (cell centre, 0) in base_link) and
 * groundpatch = 0 (:121-131), every other layer NaN there (grid_map clears dropped rows/cols of all layers),
 * then convertToDefaultStartIndex (:143).  base_plane = {r20, r21, r22, tz}: third row of the rotation and translation z
 * of the transform lookupTransform("base_link", "map") returns (:103), built from the quaternion by whichever
 * convention doTransform(PointStamped) follows (ggo_rotation_from_quaternion).  shift[2] receives the index shift
 * (rows, cols); returns 1 if the map moved. */
int ggo_map_update(ggo_map *m, double odom_x, double odom_y, const double base_plane[4], int shift[2]);

/* helpers exported for unit tests */
int ggo_get_index(const ggo_map *m, double px, double py, int *row, int *col); /* returns isInside */
float ggo_tree_sum(const float *e, int len);   /* Eigen 3.3.7 redux_novec_unroller order */
float ggo_block_sum(const float *e, int S);    /* Block<MatrixXf,S,S>::sum(), S = 3 or 5, under the selected Eigen order */
float ggo_hypotf(float x, float y);            /* glibc e_hypotf.c: (float)sqrt((double)x*x+(double)y*y) */
size_t ggo_spiral_visit_count(int rows);

#ifdef __cplusplus
}
#endif
#endif
