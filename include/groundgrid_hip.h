/*
 * groundgrid_hip.h -- C ABI of libgroundgrid_hip.so: GroundGrid's per-cloud hot path on MI355X (gfx950).
 *
 * The reference has no FFI of its own for this path: it is the C++ class
 * groundgrid::GroundSegmentation in the separately linked library
 * groundgrid_groundsegmentation_lib (/root/reference/CMakeLists.txt:112-125), used by the nodelet
 * at src/GroundGridNodelet.cpp:95 (init), :301 (setConfig) and :196 (filter_cloud).  A drop-in
 * replacement of that library keeps the class (see groundgrid_amd/host/GroundSegmentation.hpp and
 * INTEGRATION.md) and forwards to the entry points below.  Each entry point cites the reference
 * interface it replaces.
 *
 * Conventions the reference never had: every function returns gg_status (0 = OK, negative =
 * error), nothing throws across the boundary, all state lives in an opaque gg_context.  Calls on
 * one context must be externally serialised (the reference's callbacks are serialised by the ROS
 * spinner, src/GroundGridNode.cpp:42); different contexts are independent.
 *
 * Plain pointers and sizes only -- no torch / HIP types in any signature (a HIP stream is passed
 * as void*).  The library fails loudly (GG_ERR_NO_DEVICE / GG_ERR_HIP): there is no CPU fallback.
 */
#ifndef GROUNDGRID_HIP_H
#define GROUNDGRID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GG_ABI_VERSION 6

typedef enum gg_status {
    GG_OK = 0,
    GG_ERR_INVALID = -1,   /* null pointer / bad argument                        */
    GG_ERR_GEOMETRY = -2,  /* grid_map size and GroundSegmentation::init cell count disagree */
    GG_ERR_NOMEM = -3,     /* device or host allocation failed                   */
    GG_ERR_HIP = -4,       /* a HIP runtime call or kernel failed (see gg_last_error) */
    GG_ERR_CAPACITY = -5,  /* cloud larger than max_points / slot out of range   */
    GG_ERR_NO_DEVICE = -6  /* no gfx950 device visible                           */
} gg_status;

/* velodyne_pointcloud::PointXYZIR, include/velodyne_pointcloud/point_types.h:27-33 (32 B, 16-B aligned) */
typedef struct gg_point32 {
    float x, y, z, pad0;
    float intensity;
    uint16_t ring;
    uint16_t pad1;
    uint32_t pad2[2];
} gg_point32;

/* Device-native packed record (what the host staging path uploads: everything the algorithm
 * reads of a point -- x, y, z, ring -- in one 16-B load). */
typedef struct gg_point16 {
    float x, y, z;
    uint16_t ring;
    uint16_t pad;
} gg_point16;

typedef enum gg_point_format { GG_POINT32 = 0, GG_POINT16 = 1 } gg_point_format;

/* groundgrid::GroundGridConfig, generated from cfg/GroundGrid.cfg:8-21 (int_t -> int, double_t -> double) */
typedef struct gg_config {
    int point_count_cell_variance_threshold;
    int max_ring;
    double groundpatch_detection_minimum_threshold;
    double distance_factor;
    double minimum_distance_factor;
    double miminum_point_height_threshold; /* sic */
    double minimum_point_height_obstacle_threshold;
    double outlier_tolerance;
    double ground_patch_detection_minimum_point_count_threshold;
    double patch_size_change_distance;
    double occupied_cells_decrease_factor;
    double occupied_cells_point_count_factor;
    double min_outlier_detection_ground_confidence;
    int thread_count; /* accepted and ignored: results are those of thread_count = 1 */
} gg_config;

/* Compile-time constants of the reference made run-time parameters:
 * GroundGrid::mDimension / mResolution (include/groundgrid/GroundGrid.h:70-71) and
 * GroundSegmentation::verticalPointAngDist / minDistSquared (include/groundgrid/GroundSegmentation.h:69-70).
 * Zero selects the reference value. */
typedef struct gg_geometry {
    float length;                  /* 120.0f */
    float resolution;              /* .33f   */
    float vertical_point_ang_dist; /* (float)(0.00174532925*2) */
    float min_dist_squared;        /* 12.0f  */
} gg_geometry;

/* grid_map layer names used by the path (src/GroundGrid.cpp:55, src/GroundSegmentation.cpp:61-75) */
typedef enum gg_layer {
    GG_LAYER_POINTS = 0,
    GG_LAYER_GROUND = 1,
    GG_LAYER_GROUNDPATCH = 2,
    GG_LAYER_MINGROUNDHEIGHT = 3,
    GG_LAYER_MAXGROUNDHEIGHT = 4,
    GG_LAYER_GROUNDCANDIDATES = 5,
    GG_LAYER_PLANEDIST = 6,
    GG_LAYER_M2 = 7,
    GG_LAYER_MEANVARIANCE = 8,
    GG_LAYER_POINTSRAW = 9,
    GG_LAYER_VARIANCE = 10,
    GG_NUM_LAYERS = 11
} gg_layer;

/* per-input-point label: the intensity codes filter_cloud writes (src/GroundSegmentation.cpp:175,180,188);
 * 0 = the point is not in the returned cloud (outside the map, :230-231, or border, :167-168). */
enum { GG_LABEL_DROPPED = 0, GG_LABEL_GROUND = 49, GG_LABEL_NONGROUND = 99 };

/* per-input-point class decided by insert_cloud (src/GroundSegmentation.cpp:230-279) */
enum { GG_CLASS_OUTSIDE = 0, GG_CLASS_IGNORED = 1, GG_CLASS_OUTLIER = 2, GG_CLASS_KEPT = 3 };

/* gg_set_flags bits */
enum {
    GG_FLAG_MINIMAL_LAYERS = 1, /* the three layers nothing in the path reads (groundCandidates, planeDist, maxGroundHeight,
                                   src/GroundSegmentation.cpp:296,303,307) are not maintained per cloud; a reader of one of them
                                   (gg_get_layer, gg_get_layers, gg_get_layer_image_u8, gg_set_layer) has them computed first, from
                                   the tile-sorted records the slot's last cloud left on the device -- so every layer reads at all
                                   times as the reference's would.  Default: off for the calls that take or return host buffers (the
                                   binding publishes every layer), ON for gg_filter_batch -- device-resident clouds, nobody has asked
                                   for a layer -- unless GG_FLAG_EAGER_LAYERS is set */
    GG_FLAG_PROFILE = 2,        /* bracket every kernel with events on the launch stream (gg_get_kernel_times) */
    GG_FLAG_EAGER_LAYERS = 8,   /* gg_filter_batch maintains all nine per-call layers for every cloud as well (the behaviour up to ABI v5:
                                   k_reduce 1.27 instead of 1.11 ms per 1024 clouds) */
    GG_FLAG_CONCURRENT_HALVES = 4 /* a gg_filter_batch of at least 256 clouds runs as TWO independent launch sequences side by side: the
                                   clouds whose map slot is in the lower half of the context's slots on the caller's stream, the others
                                   on a stream of the library's own, and gg_reset_maps on a caller stream divides its fills the same way.
                                   A map slot is only ever touched from "its" stream, so the two sequences never wait for each other:
                                   they drift apart, kernels of different kinds overlap and fill each other's tails (+4 % clouds/s at
                                   1024 clouds per call).  An OUTPUT row follows a cloud's position in the batch, its half the cloud's slot: while
                                   every row keeps its half from batch to batch (or the output buffers are fresh) nothing joins; a batch that
                                   would write a row from the other stream than the batch before is ordered behind both first.  The price: the CALLER'S STREAM IS NOT ORDERED AFTER THE SECOND HALF.  Work the
                                   caller enqueues itself behind the call (copies of the outputs, its own kernels) must follow
                                   gg_batch_fence(ctx, stream) first; every gg_* entry point orders itself (getters, gg_synchronize,
                                   gg_allgather_label_masks, batches on other streams).  Ignored on the legacy default stream (its implicit synchronisation
                                   with every other stream makes two halves slower than one sequence) and while GG_FLAG_PROFILE is set: with two
                                   kernels sharing the device an event pair times half a machine, not a kernel.  Results identical. */
};

typedef struct gg_context gg_context;

/* ---- lifetime ------------------------------------------------------------------------------ */

void gg_default_config(gg_config *cfg);     /* cfg/GroundGrid.cfg defaults */
void gg_default_geometry(gg_geometry *g);   /* GroundGrid.h:70-71, GroundSegmentation.h:69-70 */

/* GroundSegmentation::init (src/GroundSegmentation.cpp:37-48) + the map geometry GroundGrid creates
 * (grid_map::setGeometry, src/GroundGrid.cpp:58), for n_slots independent map states ("streams")
 * that can be processed in one batched launch.  max_points = capacity per cloud. */
int gg_create(const gg_geometry *geom, int n_slots, size_t max_points, int device, gg_context **out);
void gg_destroy(gg_context *ctx);

/* GroundSegmentation::setConfig (src/GroundSegmentation.cpp:468-471).  Blocking, unlike the reference's struct copy: waits for the
 * batches in flight and rebuilds the per-cell threshold table of detect_ground_patches (O(cells) host work + one upload).  On failure
 * the context keeps its previous configuration entirely. */
int gg_set_config(gg_context *ctx, const gg_config *cfg);
int gg_get_config(const gg_context *ctx, gg_config *cfg);
int gg_set_flags(gg_context *ctx, unsigned flags);

/* Third-party conventions the reference inherits from the libraries it is built against (versions unpinned by the
 * reference: package.xml:29, CMakeLists.txt:41).  Each is one swappable function on the device and in the oracle;
 * tools/pin/ holds the probe that decides them on a real ROS box.
 *   eigen_reduction: order of Block<MatrixXf,5,5>::sum() / cwiseProduct().sum() (src/GroundSegmentation.cpp:359,374-375)
 *     GG_EIGEN_33       Eigen 3.3.x (Ubuntu 20.04 / ROS Noetic): DefaultTraversal + CompleteUnrolling, redux_novec_unroller
 *     GG_EIGEN_34_SSE   Eigen 3.4.x built for SSE2 (Packet4f): SliceVectorizedTraversal for the 5x5 blocks -- four row
 *                       lanes accumulated column by column, predux (a0+a2)+(a1+a3), then row 4 of every column
 *   The 3x3 blocks (:268, :457-458) take redux_novec_unroller under both versions. */
enum { GG_EIGEN_33 = 0, GG_EIGEN_34_SSE = 1 };
typedef struct gg_conventions {
    int eigen_reduction; /* GG_EIGEN_33 (default) */
    int reserved[7];     /* must be 0 */
} gg_conventions;
int gg_set_conventions(gg_context *ctx, const gg_conventions *conv);
int gg_get_conventions(const gg_context *ctx, gg_conventions *conv);

/* Quaternion (x, y, z, w) -> row-major 3x3 rotation the way the two candidates behind tf2::doTransform do it (host
 * arithmetic, no device involved):
 *   GG_ROT_TF2  tf2::Matrix3x3::setRotation (s = 2 / |q|^2; entries 1 - (yy + zz), xy - wz, ...): what
 *               doTransform(geometry_msgs::Point / Vector3, ...) and tf2::Transform use
 *   GG_ROT_KDL  KDL::Rotation::Quaternion (entries w2 + x2 - y2 - z2, 2xy - 2wz, ..., no normalisation): what
 *               tf2_geometry_msgs' doTransform(PointStamped) goes through (gmTransformToKDL) in ROS Melodic / Noetic --
 *               the overload the reference calls at src/GroundGrid.cpp:129 and src/GroundGridNodelet.cpp:146,176 */
enum { GG_ROT_TF2 = 0, GG_ROT_KDL = 1 };
int gg_rotation_from_quaternion(int convention, const double q_xyzw[4], double rot[9]);
/* {tx, ty, tz, qx, qy, qz, qw} -> 3x4 row-major (R | t) for gg_filter_cloud_tf / gg_batch.transforms, and the
 * {r20, r21, r22, tz} plane of gg_move_map */
int gg_transform_from_pose(int convention, const double pose7[7], double out12[12]);

int gg_get_size(const gg_context *ctx, int *rows, int *cols);          /* grid_map::GridMap::getSize */
int gg_get_geometry(const gg_context *ctx, double *resolution, double *length_x, double *length_y);
const char *gg_last_error(const gg_context *ctx);

/* ---- map state (what GroundGrid owns; the path borrows it by reference, :50) --------------- */

/* GroundGrid::initGroundGrid layer values (src/GroundGrid.cpp:71-75) + map position */
int gg_reset_map(gg_context *ctx, int slot, double pos_x, double pos_y, float odom_z);
/* The same for n_slots consecutive map states in one launch (position (pos_x, pos_y) and height odom_z for all).  With
 * persistent_only != 0 only the state that outlives a cloud is re-initialised -- ground := odom_z, groundpatch := 1e-7 -- which
 * is all a "cold" start needs: the nine per-call layers are rewritten by the next filter call anyway (:61-75).
 * `stream`: NULL = the context's stream (like every other map mutation); a caller stream (or GG_STREAM_DEFAULT) enqueues the
 * fills there, ordered like a batch on that stream -- a server that re-initialises maps between batches on its own stream
 * then has no cross-stream hand-over in its loop.
 * Fresh maps (ABI v6, nothing to do for the caller): the (ground, groundpatch) layer of a re-initialised map is not written cell by cell
 * -- the library notes that it holds the reset's values, and a large gg_filter_batch of such maps sweeps them as they are (0.01 instead
 * of 0.35 ms per 1024 maps of 364 x 364); every other entry point that reads or edits the
 * layer (getters, setters, gg_move_map, the stage calls, single clouds, small or mixed batches) fills it first.  Every getter returns
 * the values above at all times.  Environment GG_FRESH_MAPS=0: write every cell, as before. */
int gg_reset_maps(gg_context *ctx, int first_slot, int n_slots, double pos_x, double pos_y, float odom_z, int persistent_only, void *stream);
/* map position after grid_map::move (src/GroundGrid.cpp:97); layers unchanged */
int gg_set_map_position(gg_context *ctx, int slot, double pos_x, double pos_y);
/* GroundGrid::update for an initialised map (src/GroundGrid.cpp:83-147): grid_map::GridMap::move to the odometry
 * position (whole cells; the map position is snapped), newly exposed cells get ground = -(z of the cell centre in
 * base_link) and groundpatch = 0 (:121-131), then convertToDefaultStartIndex (:143) -- on the device, so that the two
 * persistent layers never leave HBM between clouds.
 * base_plane = {r20, r21, r22, tz}: third row of the rotation and z of the translation of
 * lookupTransform("base_link", "map") (:103), i.e. z_base(p) = ((r20 * p.x + r21 * p.y) + r22 * p.z) + tz, evaluated in
 * this order in double (:129).  ABI v2: the caller hands over matrix entries, NOT a quaternion -- which rotation matrix
 * tf2::doTransform(PointStamped) builds from the quaternion (tf2::Matrix3x3::setRotation or KDL::Rotation::Quaternion)
 * is the binding's decision (gg_rotation_from_quaternion offers both).  shift (nullable) receives the index shift
 * (rows, cols). */
int gg_move_map(gg_context *ctx, int slot, double odom_x, double odom_y, const double base_plane[4], int shift[2]);
int gg_get_map_position(const gg_context *ctx, int slot, double *pos_x, double *pos_y);
/* any of the 11 layers, column-major rows x cols float32 (Eigen::MatrixXf), host memory */
int gg_set_layer(gg_context *ctx, int slot, int layer, const float *src);
int gg_get_layer(gg_context *ctx, int slot, int layer, float *dst);
/* several layers in one go: dst[l] (nullable) receives layer l.  What the nodelet's publishers need after a cloud
 * (src/GroundGridNodelet.cpp:211-224 publishes every layer that has a subscriber): the extraction kernels and the downloads of
 * all requested layers are enqueued back to back and waited for once, instead of one synchronisation per layer. */
int gg_get_layers(gg_context *ctx, int slot, float *const dst[GG_NUM_LAYERS]);
/* GroundSegmentation::expectedPoints (src/GroundSegmentation.cpp:40-46), host copy */
int gg_get_expected_points(const gg_context *ctx, float *dst);

/* ---- the hot path -------------------------------------------------------------------------- */

/* GroundSegmentation::filter_cloud (include/groundgrid/GroundSegmentation.h:54,
 * src/GroundSegmentation.cpp:50-197): host buffers in, host buffers out, synchronous.
 *   cloud, n    : input cloud already in the map frame (Nodelet.cpp:166-181)
 *   origin      : cloudOrigin x,y,z
 *   base_z      : mapToBase.transform.translation.z (the only field of the transform the path uses, :406-411)
 *   out_cloud   : capacity n (nullable); receives the returned cloud: kept, then ignored, then outliers
 *   out_n       : number of points in the returned cloud
 *   out_label   : per input point GG_LABEL_* (nullable)
 *   out_index   : per input point position in the returned cloud, -1 if dropped (nullable) */
int gg_filter_cloud(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const float origin[3],
                    double base_z, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label,
                    int32_t *out_index);

/* filter_cloud for a cloud that is still in the sensor frame: the per-point transform of points_callback
 * (src/GroundGridNodelet.cpp:148-184) is fused into the first kernel.  map_from_cloud = 3x4 row-major (R | t) of
 * lookupTransform("map", cloud frame).  The returned cloud is in the map frame, as in the reference. */
int gg_filter_cloud_tf(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double map_from_cloud[12],
                       const float origin[3], double base_z, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label,
                       int32_t *out_index);

/* Pipelined form of gg_filter_cloud / gg_filter_cloud_tf (map_from_cloud nullable): packs the cloud into one of
 * GG_ASYNC_DEPTH pinned staging buffers, enqueues upload + kernels + download and returns a ticket without waiting;
 * gg_filter_cloud_wait blocks until that cloud is done and hands out the results.  With two clouds in flight the
 * host-side packing and the H2D copy of cloud k+1 overlap the kernels of cloud k, and the result download / returned-
 * cloud assembly of cloud k overlaps the kernels of cloud k+1 (clouds of one slot still execute in call order: cloud k+1
 * reads the map state cloud k left).  `cloud` must stay valid until the matching wait when out_cloud is requested
 * there.  Tickets must be waited for in issue order; at most GG_ASYNC_DEPTH may be outstanding (GG_ERR_CAPACITY). */
#define GG_ASYNC_DEPTH 2
int gg_filter_cloud_async(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *map_from_cloud,
                          const float origin[3], double base_z, int *ticket);
int gg_filter_cloud_wait(gg_context *ctx, int ticket, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label,
                         int32_t *out_index);

/* filter_cloud AND the layers a publisher needs afterwards, as one call: what the nodelet does per cloud is filter_cloud followed
 * by reading every layer of the map (grid_map message + images, src/GroundGridNodelet.cpp:196-228).  layers[l] (nullable) receives
 * layer l, column-major rows x cols, like gg_get_layers -- but the eight layers that are final once the insertion has run
 * (minGroundHeight, maxGroundHeight, groundCandidates, planeDist, m2, meanVariance, pointsRaw, variance) are extracted and
 * downloaded on a side branch WHILE the patch stencil and the terrain sweep run, and ground / groundpatch / points follow the
 * per-point results while the host assembles the returned cloud.  map_from_cloud nullable (as gg_filter_cloud_async).
 * Destinations inside a range passed to gg_host_register are written by the device directly (no staging copy on the host): a host
 * whose map planes keep their addresses from cloud to cloud -- grid_map::GridMap does -- registers them once. */
int gg_filter_cloud_layers(gg_context *ctx, int slot, const gg_point32 *cloud, size_t n, const double *map_from_cloud, const float origin[3],
                           double base_z, gg_point32 *out_cloud, size_t *out_n, uint8_t *out_label, int32_t *out_index,
                           float *const layers[GG_NUM_LAYERS]);
/* hipHostRegister / hipHostUnregister of a host range for a caller that has no HIP headers: downloads of this context whose
 * destination lies inside a registered range (gg_filter_cloud_layers) land there directly.  Unregister before freeing the memory;
 * gg_destroy unregisters what is left. */
int gg_host_register(gg_context *ctx, void *ptr, size_t bytes);
int gg_host_unregister(gg_context *ctx, void *ptr);

/* With GG_GRAPH=1 in the environment, one cloud per call (gg_filter_cloud*, gg_filter_batch with n_clouds == 1 on a stream other
 * than the legacy default one) replays a captured HIP graph from the third call of a kind on: the seven launches of the path reach
 * the device as one submission.  Off by default: measured, a replay is no faster than the eager launches. */

/* Batched, device-resident form of the same call: n_clouds independent (cloud, map-state) pairs in
 * one set of launches, slot first_slot + b (or slots[b]) for cloud b.  Pointers prefixed d_ are device memory.
 * Enqueues on `stream` (a hipStream_t passed as void*; NULL = the context's own stream, GG_STREAM_DEFAULT = the legacy
 * default ("null") stream, which as a hipStream_t is itself 0) and returns without waiting.
 * Ordering across streams is the library's job, not the caller's: a batch waits (hipStreamWaitEvent) for every earlier
 * map mutation of the context (gg_reset_map, gg_move_map, gg_set_layer, earlier batches on other streams), and every
 * later entry point that reads or writes map state on the context's own stream (gg_get_layer, gg_set_layer,
 * gg_move_map, gg_reset_map, the image / class getters, gg_filter_cloud*) waits for the batch.  The caller only has to
 * keep the buffers named in gg_batch alive and unmodified until the stream has passed the batch. */
typedef struct gg_batch {
    int n_clouds;
    int first_slot;
    int point_format;        /* gg_point_format */
    const void *d_points;    /* [n_clouds][cloud_stride] records of point_format */
    size_t cloud_stride;     /* in points; <= max_points */
    const int32_t *n_points; /* host [n_clouds] */
    const float *origins;    /* host [n_clouds][3] */
    const double *base_z;    /* host [n_clouds] */
    const double *transforms; /* host [n_clouds][12], nullable: map <- cloud frame as 3x4 row-major (R | t).  When given,
                                the points are still in the sensor frame and are transformed on the device exactly like
                                the nodelet does per point (tf2::doTransform in double, cast to float,
                                src/GroundGridNodelet.cpp:166-181); labels / returned clouds refer to map-frame points */
    uint8_t *d_labels;       /* [n_clouds][cloud_stride], nullable */
    int32_t *d_out_index;    /* [n_clouds][cloud_stride], nullable */
    gg_point32 *d_out_clouds; /* [n_clouds][cloud_stride], nullable; needs point_format == GG_POINT32 */
    int32_t *d_out_counts;   /* [n_clouds][4]: returned-cloud size, kept, ignored, outliers; nullable */
    uint8_t *d_label_masks;  /* [n_clouds][(cloud_stride + 3) / 4], nullable: the labels as a 2-bit mask, point p in bits
                                2*(p%4).. of byte p/4: 0 dropped, 1 ground (49), 2 non-ground (99) -- what a multi-GPU
                                caller all-gathers (a quarter of d_labels).  cloud_stride must be a multiple of 4; only the bytes
                                covering points < n_points (rounded up to a multiple of 64) are written */
    const int32_t *slots;    /* host [n_clouds], nullable (ABI v3): cloud b meets map slot slots[b] instead of first_slot + b
                                (first_slot is ignored then).  Entries must be distinct and inside the context: a server that
                                holds many streams' maps filters whichever of them received a cloud, in one set of launches */
    uint8_t *d_out_pc2;      /* [n_clouds][cloud_stride * GG_PC2_POINT_STEP], nullable (ABI v5): the returned cloud of cloud b --
                                kept, then ignored, then outliers, d_out_counts[b][0] records -- as sensor_msgs/PointCloud2 data in
                                the 18-byte layout of scripts/kitti_data_publisher.py:139-150 (x@0 y@4 z@8 intensity@12 float32,
                                ring@16 uint16; intensity = 49 / 99), written by the label kernel itself from either point format:
                                what a publisher sends (src/GroundGridNodelet.cpp:196-200) comes down as 18 B x returned points with
                                no host assembly.  (pcl::toROSMsg of PointXYZIR publishes the 32-byte struct as is: that layout is
                                d_out_clouds.) */
} gg_batch;
#define GG_PC2_POINT_STEP 18
#define GG_STREAM_DEFAULT ((void *)(intptr_t)-1)
int gg_filter_batch(gg_context *ctx, const gg_batch *batch, void *stream);
/* GG_FLAG_CONCURRENT_HALVES: orders `stream` (same convention) after both halves of every batch enqueued so far.  A no-op otherwise. */
int gg_batch_fence(gg_context *ctx, void *stream);
int gg_synchronize(gg_context *ctx);
/* The few places where a kernel waits for ANOTHER work-group (the sweep cut into parts, the tile scan cut into parts, the fused front
 * end) bound their waits; a wait that runs out leaves a code in a host-visible word instead of hanging.  Every gg_* call that
 * synchronises reports it ONCE as GG_ERR_HIP (text in gg_last_error) and clears it: the outputs of the batches enqueued since the
 * previous report are void and the map states they touched should be re-initialised (gg_reset_map); the context itself keeps
 * working.  A caller that synchronises its own stream instead of calling into the library asks here: returns the pending code
 * (0 = none, > 0 = a wait ran out) without synchronising; clear != 0 also clears it. */
int gg_device_error(gg_context *ctx, int clear);

/* ---- the one collective of the path: the all-gather of the per-cloud label masks (BASELINE configs[2]) ----
 * The reference has no distributed code; clouds shard as independent (cloud, map) pairs and the only exchange is that every
 * rank ends up with every cloud's labels.  These entry points let a C / C++ host run that configuration without Python:
 * they bind RCCL (librccl.so, the ROCm build of the NCCL API) at run time with dlopen, so the library has no link-time
 * dependency on it and a single-GPU user never loads it.
 *
 *   gg_comm_unique_id       rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by whatever means
 *                           the host has (MPI, a socket, torch.distributed ...)
 *   gg_comm_init_rank       every rank: ncclCommInitRank on the CURRENT HIP device -> an opaque communicator (ncclComm_t)
 *   gg_comm_init_rank_for   the same on the device of `ctx` (selected first): what a caller that holds a context wants -- with one
 *                           rank, or a transport that never touched HIP, nothing else has selected a device yet (ABI v4)
 *   gg_allgather_label_masks  ncclAllGather(d_send, d_recv, bytes_per_rank, ncclUint8) on `stream` (same convention as
 *                           gg_filter_batch: NULL = the context's stream, GG_STREAM_DEFAULT = the legacy default stream); d_send =
 *                           this rank's gg_batch.d_label_masks (or d_labels), d_recv = [world][bytes_per_rank].  The call is
 *                           ordered after the context's last batch (event wait when the streams differ) and returns without
 *                           waiting; `comm` may be any ncclComm_t, also one the host created itself.  The library remembers the
 *                           send buffer: a later gg_filter_batch on ANOTHER stream whose d_label_masks / d_labels overlap it is
 *                           ordered after the gather (event wait) -- batches that write other buffers (double buffering) are not
 *   gg_comm_destroy
 * GG_ERR_NO_DEVICE when librccl.so cannot be loaded, GG_ERR_HIP for RCCL errors (text in gg_last_error). */
int gg_collective_available(void);
int gg_comm_unique_id(uint8_t id_out[128]);
int gg_comm_init_rank(const uint8_t id[128], int n_ranks, int rank, void **comm_out);
int gg_comm_init_rank_for(gg_context *ctx, const uint8_t id[128], int n_ranks, int rank, void **comm_out);
int gg_comm_destroy(void *comm);
int gg_allgather_label_masks(gg_context *ctx, void *comm, const uint8_t *d_send, uint8_t *d_recv, size_t bytes_per_rank, void *stream);

/* ---- wire formats around the path (src/GroundGridNodelet.cpp:120, :211-291) -------------------------------------- */

/* filter_cloud straight from a sensor_msgs/PointCloud2 payload (e.g. the KITTI player's 18-byte records,
 * scripts/kitti_data_publisher.py:139-150: x@0 y@4 z@8 intensity@12 ring@16): one host pass packs the fields the path
 * reads, instead of pcl::fromROSMsg into 32-byte points first (Nodelet.cpp:120).  map_from_cloud (nullable) as in
 * gg_filter_cloud_tf.  Results per input point; the caller owns the payload and can assemble whatever message it needs. */
int gg_filter_cloud_pc2(gg_context *ctx, int slot, const uint8_t *data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                        size_t off_z, size_t off_ring, const double *map_from_cloud, const float origin[3], double base_z,
                        uint8_t *out_label, int32_t *out_index, size_t *out_n);
/* PointCloud2 payload in, PointCloud2 payload out: the same call with the RETURNED CLOUD written as 18-byte records (x, y, z,
 * intensity = 49 / 99, ring; the layout above, gg_batch.d_out_pc2) by the label kernel -- order kept, ignored, outliers as in
 * src/GroundSegmentation.cpp:150-189 -- so that the download is 18 B per returned point and the host assembles nothing.  out_data
 * needs room for n * GG_PC2_POINT_STEP bytes; *out_n = points of the returned cloud (= width x height of the message, row_step =
 * 18 * width).  The input layout is free (point_step / offsets) as for gg_filter_cloud_pc2. */
int gg_filter_cloud_pc2_out(gg_context *ctx, int slot, const uint8_t *data, size_t n, size_t point_step, size_t off_x, size_t off_y,
                            size_t off_z, size_t off_ring, const double *map_from_cloud, const float origin[3], double base_z,
                            uint8_t *out_data, size_t *out_n);

/* The map as the serialised grid_map_msgs/GridMap the nodelet publishes per cloud (src/GroundGridNodelet.cpp:211-214:
 * grid_map::GridMapRosConverter::toMessage, info.header.stamp := the cloud's stamp): ROS 1 wire format, little endian --
 *   info   { header {seq, stamp, frame_id}, resolution, length_x, length_y, pose {position (map x, map y, 0), orientation (0,0,0,1)} }
 *   layers[], basic_layers[]    names in the order the reference adds them (src/GroundGrid.cpp:55, src/GroundSegmentation.cpp:61-75)
 *   data[]  one std_msgs/Float32MultiArray per layer: dim[0] {"column_index", cols, rows*cols}, dim[1] {"row_index", rows, rows},
 *           data_offset 0, data = the layer column-major (Eigen's storage, copied as is)
 *   outer_start_index = inner_start_index = 0 (GroundGrid::update ends with convertToDefaultStartIndex, src/GroundGrid.cpp:143)
 * restated from grid_map_ros 1.6.x (GridMapRosConverter::toMessage, GridMapMsgHelpers: not under /root/reference, unpinned).
 * layer_mask: bit per gg_layer (0 = all eleven).  The layer planes are extracted by one kernel, come down in one copy and are
 * placed straight into the message.  dst == NULL or capacity too small: only *size is set (GG_ERR_CAPACITY in the second case). */
typedef struct gg_gridmap_header {
    uint32_t seq;
    uint32_t stamp_sec, stamp_nsec;
    const char *frame_id;   /* NULL = "map" (src/GroundGrid.cpp:57) */
    unsigned basic_layers;  /* bit per gg_layer; the reference declares none */
} gg_gridmap_header;
int gg_get_gridmap_message(gg_context *ctx, int slot, unsigned layer_mask, const gg_gridmap_header *header, uint8_t *dst, size_t capacity, size_t *size);

/* grid_map::GridMapCvConverter::toImage<unsigned char, 1> of one layer (Nodelet.cpp:239): rows x cols row-major bytes,
 * the layer normalised between the min and max of its finite cells (returned in lower / upper), non-finite cells 0.
 * cv::applyColorMap (:240) is left to the host. */
int gg_get_layer_image_u8(gg_context *ctx, int slot, int layer, uint8_t *dst, float *lower, float *upper);
/* the 32FC3 terrain image (Nodelet.cpp:247-268): rows x cols x 3 floats (ground, 3x3 pointsRaw sum >= 27, pointsRaw) */
int gg_get_terrain_image(gg_context *ctx, int slot, float *dst);

/* insert_cloud's per-point decision (include/groundgrid/GroundSegmentation.h:55): after a filter call,
 * class (GG_CLASS_*) and cell (row + col*rows, -1 outside) of every input point of `slot`. */
int gg_get_point_classes(gg_context *ctx, int slot, size_t n, uint8_t *out_class, int32_t *out_cell);

/* ---- the stage members of the reference's class, one by one (include/groundgrid/GroundSegmentation.h:59-62) --------
 * filter_cloud runs them as stages of one launch sequence; the header also declares them public, so the drop-in library
 * answers a caller that invokes one on its own.  Each runs on the slot's layers AS THEY STAND (whatever the last cloud,
 * gg_set_layer or an earlier stage left) and leaves its results in the slot, synchronously:
 *   GG_STAGE_DETECT_GROUND_PATCHES        detect_ground_patches(map, section), src/GroundSegmentation.cpp:314-340:
 *                                         variance := m2 ./ (points + FLT_MIN) over the whole layer (:323), then
 *                                         detect_ground_patch<3|5> for every cell of quadrant `section` (0: top-left, 1: top-right,
 *                                         2: bottom-left, 3: bottom-right, :325-328; -1: all four, what filter_cloud's four threads do)
 *   GG_STAGE_SPIRAL_GROUND_INTERPOLATION  spiral_ground_interpolation(map, toBase), :398-441, base_z = toBase.transform.translation.z
 *                                         (the only field the function uses, :406-411).  filter_cloud's reset of `points` (:147)
 *                                         is NOT part of it.
 *   GG_STAGE_DETECT_GROUND_PATCH_3 / _5   detect_ground_patch<S>(map, i, j), :343-395 (reads `variance` as it stands)
 *   GG_STAGE_INTERPOLATE_CELL             interpolate_cell(map, x = i, y = j), :445-465
 * The many-cell stages are the kernels of the path (k_patch without the skipping of blocks the last cloud left empty and with
 * the :359 count in Eigen's order, k_sweep); the single-cell ones are one lane of a kernel of their own.  Cell indices whose
 * blocks would leave the map -- UB in the reference -- are GG_ERR_INVALID. */
enum {
    GG_STAGE_DETECT_GROUND_PATCHES = 1,
    GG_STAGE_SPIRAL_GROUND_INTERPOLATION = 2,
    GG_STAGE_DETECT_GROUND_PATCH_3 = 3,
    GG_STAGE_DETECT_GROUND_PATCH_5 = 4,
    GG_STAGE_INTERPOLATE_CELL = 5
};
typedef struct gg_stage_args {
    int section;   /* GG_STAGE_DETECT_GROUND_PATCHES: 0..3, or -1 for all four quadrants */
    int i, j;      /* the single-cell stages: row and column */
    double base_z; /* GG_STAGE_SPIRAL_GROUND_INTERPOLATION */
} gg_stage_args;
int gg_run_stage(gg_context *ctx, int slot, int stage, const gg_stage_args *args);

/* GroundSegmentation::insert_cloud as the member it is (include/groundgrid/GroundSegmentation.h:55, src/GroundSegmentation.cpp:200-311):
 * the points cloud[start, end) against the slot's map AS IT STANDS -- no per-call reset (:61-75 is filter_cloud's): `pointsRaw` and the seven
 * recurrences of :296-309 continue from what the layers hold, a cell's count from wherever an earlier range left it; the outlier test of
 * :243-279 reads ground / groundpatch as they are.  out_class / out_cell (each end - start entries, either may be null) = per point of the
 * range, in cloud order, its GG_CLASS_* and its cell (row + col * rows; -1 outside the map): the three lists the reference appends to
 * are the points of class KEPT (`point_index`, with their cells), IGNORED (`ignored`, with their cells) and OUTLIER (`outliers`), each
 * in cloud order -- the host mirrors build them from these two arrays.  Synchronous; ABI v6. */
int gg_insert_cloud(gg_context *ctx, int slot, const gg_point32 *cloud, size_t start, size_t end, const float origin[3], uint8_t *out_class, int32_t *out_cell);

/* ---- measurement --------------------------------------------------------------------------- */

enum {
    GG_K_CLASSIFY = 0, /* K1: insert_cloud classify + tile key (:219-279)                 */
    GG_K_SCAN = 1,     /*     exclusive scan of tile histograms                            */
    GG_K_SCATTER = 2,  /*     stable scatter into Morton-tile order                        */
    GG_K_REDUCE = 3,   /* K2: ordered per-cell reductions (:282-309) + variance (:323)    */
    GG_K_PATCH = 4,    /* K3: detect_ground_patch<3|5> stencil (:330-395)                 */
    GG_K_SPIRAL = 5,   /* K4: spiral_ground_interpolation (:398-465)                       */
    GG_K_LABEL = 6,    /* K5: label loop (:147-189)                                        */
    GG_NUM_KERNELS = 7
};
/* With GG_FLAG_PROFILE: accumulated milliseconds and launch counts per kernel since the last reset. */
int gg_get_kernel_times(gg_context *ctx, double ms[GG_NUM_KERNELS], int64_t launches[GG_NUM_KERNELS], int reset);
const char *gg_kernel_name(int k);
int gg_abi_version(void);

/* Testing hook, runs on the host (no GPU): the terrain sweep the device runs (ring-per-lane dataflow, csrc/sweep_core.h) --
 * the same per-lane code -- emulated wavefront by wavefront with a seeded interleaving of the wavefronts (seed 0 = round
 * robin) and, if late_loads, every layer load resolved only when it is used.  gp2 = interleaved (ground, groundpatch)
 * [n*n][2] in Eigen's column-major cell order, updated in place like spiral_ground_interpolation
 * (src/GroundSegmentation.cpp:398-465) would.
 * stats (nullable, 8 longs): wave-steps, stalls, layer loads, stores, LDS operations, LDS bytes, wavefronts, decay radius^2.
 * Returns 0, or -10 if the wavefronts deadlock. */
int gg_debug_emulate_ring_sweep(int n, double resolution, float min_dist_squared, float *gp2, float base_z,
                                double occupied_cells_decrease_factor, unsigned seed, int late_loads, long *stats);
/* Testing hook, host only: the sweep's wait test -- one compare per half step against the last step the progress counters read last
 * cover (csrc/sweep_core.h ChainSync::cover) -- held to the closed-form needs it inverts, for every side, ring group, counter value and
 * wave-step of an n x n map.  Returns the number of disagreements (0), -1 for n < 8. */
long gg_debug_sweep_sync_selftest(int n);

#ifdef __cplusplus
}
#endif
#endif
