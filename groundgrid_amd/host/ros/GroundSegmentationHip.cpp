// GroundSegmentationHip.cpp -- the reference-side binding: compile this INSTEAD of src/GroundSegmentation.cpp into
// groundgrid_groundsegmentation_lib (/root/reference/CMakeLists.txt:112-125) and link libgroundgrid_hip.so.
//
// It defines groundgrid::GroundSegmentation's member functions with the exact signatures of the reference's own header
// (include/groundgrid/GroundSegmentation.h:53-62, included below -- the header, the nodelet, the launch files and the
// evaluation scripts stay untouched) and forwards to the C ABI of include/groundgrid_hip.h.
//
// This file is compile-checked in this repository (tests/test_ros_binding_cpu.py) against the reference's header and
// DECLARATION-ONLY stand-ins of the ROS / PCL / grid_map types (tests/cpp/decl_only/): that checks OUR text against the
// reference's class declaration; it builds nothing of the reference and pins no parity.
//
// Map ownership.  The reference borrows a grid_map::GridMap& that GroundGrid::update edits on the host between clouds
// (src/GroundGrid.cpp:97-143).  Here the authoritative copy of the layers lives in HBM; the binding
//   * uploads `ground` / `groundpatch` only when the host can have changed them: the first call, and whenever the map
//     position differs from the one seen last (GroundGrid::update changes layer contents only together with a move);
//   * downloads after every call the layers selected by GROUNDGRID_HIP_LAYERS (default "all": the nodelet publishes every
//     layer, src/GroundGridNodelet.cpp:211-224; "state" = ground, groundpatch, points, pointsRaw: what GroundGrid::update
//     and the terrain image read, :251-253; 2 x 0.5 MB up / 11 x 0.5 MB down per cloud at most).
// INTEGRATION.md 2b shows the variant that moves GroundGrid::update onto the device as well (gg_move_map).
#include <cstdlib>
#include <cstring>

#include <groundgrid/GroundSegmentation.h>

#include "groundgrid_hip.h"

namespace groundgrid {

namespace {

// one map per process, like the reference's function-local `static grid_map::Matrix&` references
// (src/GroundSegmentation.cpp:76-78,203-213)
gg_context *g_ctx = nullptr;
bool g_have_position = false;
double g_pos_x = 0.0, g_pos_y = 0.0;
gg_geometry g_geometry;
gg_config g_config;
bool g_have_config = false;
size_t g_capacity = 0; // points per cloud the context was created for; grows on demand (ensure_capacity)

size_t initial_capacity()
{
    // GROUNDGRID_HIP_MAX_POINTS: expected cloud size (default 400 000: three times an HDL-64E revolution).  Only a starting point:
    // a larger cloud makes the binding re-create the context with room for it.
    const char *e = std::getenv("GROUNDGRID_HIP_MAX_POINTS");
    const long v = e ? std::atol(e) : 0;
    return v > 0 ? static_cast<size_t>(v) : 400000;
}

// (re)create the device context for clouds of up to `points` points; the map state is uploaded again by the next filter_cloud
bool create_context(size_t points)
{
    if (g_ctx) gg_destroy(g_ctx);
    g_ctx = nullptr;
    g_have_position = false;
    g_capacity = 0;
    const int rc = gg_create(&g_geometry, 1, points, 0, &g_ctx);
    if (rc != GG_OK) {
        ROS_FATAL("groundgrid_hip: gg_create failed with status %d (no gfx950 device, out of memory, or grid_map and init() disagree on the cell count)", rc);
        g_ctx = nullptr;
        return false;
    }
    g_capacity = points;
    if (g_have_config) {
        const int rc2 = gg_set_config(g_ctx, &g_config);
        if (rc2 != GG_OK) ROS_ERROR("groundgrid_hip: gg_set_config failed with status %d", rc2);
    }
    return true;
}

bool ensure_capacity(size_t points)
{
    if (g_ctx && points <= g_capacity) return true;
    const size_t want = points + points / 2; // headroom: sensor clouds vary by a few percent from revolution to revolution
    ROS_WARN("groundgrid_hip: a cloud of %zu points exceeds the context's capacity of %zu: re-creating it for %zu", points, g_capacity, want);
    return create_context(want);
}

static_assert(sizeof(velodyne_pointcloud::PointXYZIR) == sizeof(gg_point32), "PointXYZIR must be the 32-byte record of point_types.h:27-33");

const char *const kLayerNames[GG_NUM_LAYERS] = {"points",           "ground",    "groundpatch", "minGroundHeight",
                                                "maxGroundHeight", "groundCandidates", "planeDist",   "m2",
                                                "meanVariance",    "pointsRaw", "variance"};

bool download_all_layers()
{
    const char *e = std::getenv("GROUNDGRID_HIP_LAYERS");
    return !(e && std::strcmp(e, "state") == 0);
}

} // namespace

// src/GroundSegmentation.cpp:37-48
void GroundSegmentation::init(ros::NodeHandle &nodeHandle, const size_t dimension, const float &resolution)
{
    (void)nodeHandle; // unused by the reference as well
    gg_default_geometry(&g_geometry);
    g_geometry.length = static_cast<float>(dimension); // the nodelet passes 120.0f into the size_t parameter (Nodelet.cpp:95)
    g_geometry.resolution = resolution;
    g_geometry.vertical_point_ang_dist = verticalPointAngDist;
    g_geometry.min_dist_squared = minDistSquared;
    if (!create_context(initial_capacity())) return;
    int rows = 0, cols = 0;
    gg_get_size(g_ctx, &rows, &cols);
    expectedPoints.resize(rows, cols); // :40-46, kept for callers that inspect the member
    gg_get_expected_points(g_ctx, expectedPoints.data());
}

// src/GroundSegmentation.cpp:468-471
void GroundSegmentation::setConfig(const groundgrid::GroundGridConfig &config)
{
    mConfig = config;
    gg_config k;
    gg_default_config(&k);
    k.point_count_cell_variance_threshold = config.point_count_cell_variance_threshold;
    k.max_ring = config.max_ring;
    k.groundpatch_detection_minimum_threshold = config.groundpatch_detection_minimum_threshold;
    k.distance_factor = config.distance_factor;
    k.minimum_distance_factor = config.minimum_distance_factor;
    k.miminum_point_height_threshold = config.miminum_point_height_threshold;
    k.minimum_point_height_obstacle_threshold = config.minimum_point_height_obstacle_threshold;
    k.outlier_tolerance = config.outlier_tolerance;
    k.ground_patch_detection_minimum_point_count_threshold = config.ground_patch_detection_minimum_point_count_threshold;
    k.patch_size_change_distance = config.patch_size_change_distance;
    k.occupied_cells_decrease_factor = config.occupied_cells_decrease_factor;
    k.occupied_cells_point_count_factor = config.occupied_cells_point_count_factor;
    k.min_outlier_detection_ground_confidence = config.min_outlier_detection_ground_confidence;
    k.thread_count = config.thread_count; // accepted, ignored: results are those of thread_count = 1
    g_config = k; // (kept: a context re-created for a larger cloud gets the same configuration)
    g_have_config = true;
    if (!g_ctx) return;
    const int rc = gg_set_config(g_ctx, &k);
    if (rc != GG_OK) ROS_ERROR("groundgrid_hip: gg_set_config failed with status %d", rc);
}

// src/GroundSegmentation.cpp:50-197
pcl::PointCloud<GroundSegmentation::PCLPoint>::Ptr GroundSegmentation::filter_cloud(const pcl::PointCloud<PCLPoint>::Ptr cloud,
                                                                                     const PCLPoint &cloudOrigin,
                                                                                     const geometry_msgs::TransformStamped &mapToBase,
                                                                                     grid_map::GridMap &map)
{
    pcl::PointCloud<PCLPoint>::Ptr filtered_cloud(new pcl::PointCloud<PCLPoint>);
    filtered_cloud->header = cloud->header;
    if (!g_ctx) {
        ROS_ERROR("groundgrid_hip: filter_cloud before a successful init");
        return filtered_cloud;
    }
    // the layers filter_cloud adds (:61-75) must exist for the publishers even if they are not downloaded
    for (int l = 0; l < GG_NUM_LAYERS; ++l)
        if (!map.exists(kLayerNames[l])) map.add(kLayerNames[l], 0.0);

    const size_t n = cloud->points.size();
    if (!ensure_capacity(n)) return filtered_cloud; // (a re-created context forgets the map: uploaded again just below)

    // state the host may have edited since the last cloud: GroundGrid::update moves the map and seeds the exposed cells
    const double px = map.getPosition().x(), py = map.getPosition().y();
    if (!g_have_position || px != g_pos_x || py != g_pos_y) {
        int rc_up = gg_set_map_position(g_ctx, 0, px, py);
        if (rc_up == GG_OK) rc_up = gg_set_layer(g_ctx, 0, GG_LAYER_GROUND, map["ground"].data()); // Eigen::MatrixXf is column-major: as is
        if (rc_up == GG_OK) rc_up = gg_set_layer(g_ctx, 0, GG_LAYER_GROUNDPATCH, map["groundpatch"].data());
        if (rc_up != GG_OK) {
            // the device would keep filtering against a stale terrain: report, return nothing, and try the upload again next time
            ROS_ERROR("groundgrid_hip: uploading the map state failed with status %d (%s)", rc_up, gg_last_error(g_ctx));
            g_have_position = false;
            return filtered_cloud;
        }
        g_have_position = true;
        g_pos_x = px;
        g_pos_y = py;
    }

    filtered_cloud->points.resize(n);
    const float origin[3] = {cloudOrigin.x, cloudOrigin.y, cloudOrigin.z};
    size_t n_out = 0;
    const int rc = gg_filter_cloud(g_ctx, 0, reinterpret_cast<const gg_point32 *>(cloud->points.data()), n, origin,
                                   mapToBase.transform.translation.z, // the only field of the transform the path uses, :406-411
                                   reinterpret_cast<gg_point32 *>(filtered_cloud->points.data()), &n_out, nullptr, nullptr);
    if (rc != GG_OK) {
        ROS_ERROR("groundgrid_hip: %s", gg_last_error(g_ctx));
        n_out = 0;
    }
    filtered_cloud->points.resize(n_out);

    // the layers the publishers (and GroundGrid::update) read, in one go: the extraction kernels and the downloads are enqueued
    // back to back and waited for once (gg_get_layers)
    const bool all = download_all_layers();
    float *dst[GG_NUM_LAYERS];
    for (int l = 0; l < GG_NUM_LAYERS; ++l) {
        const bool state = l == GG_LAYER_GROUND || l == GG_LAYER_GROUNDPATCH || l == GG_LAYER_POINTS || l == GG_LAYER_POINTSRAW;
        dst[l] = (all || state) ? map[kLayerNames[l]].data() : nullptr;
    }
    const int rc_down = gg_get_layers(g_ctx, 0, dst);
    if (rc_down != GG_OK) ROS_ERROR("groundgrid_hip: downloading the layers failed with status %d (%s)", rc_down, gg_last_error(g_ctx));
    return filtered_cloud;
}

// src/GroundSegmentation.cpp:200-311.  The reference calls this from its own filter_cloud only; here the per-point
// decisions of the LAST filter_cloud call are handed out in the same three lists, restricted to [start, end).
void GroundSegmentation::insert_cloud(const pcl::PointCloud<PCLPoint>::Ptr cloud, const size_t start, const size_t end,
                                      const PCLPoint &cloudOrigin, std::vector<std::pair<size_t, grid_map::Index>> &point_index,
                                      std::vector<std::pair<size_t, grid_map::Index>> &ignored, std::vector<size_t> &outliers,
                                      grid_map::GridMap &map)
{
    (void)cloudOrigin;
    (void)map;
    if (!g_ctx) return;
    const size_t n = cloud->points.size();
    std::vector<uint8_t> cls(n);
    std::vector<int32_t> cell(n);
    if (gg_get_point_classes(g_ctx, 0, n, cls.data(), cell.data()) != GG_OK) return;
    int rows = 0, cols = 0;
    gg_get_size(g_ctx, &rows, &cols);
    for (size_t i = start; i < end && i < n; ++i) {
        if (cls[i] == GG_CLASS_OUTSIDE) continue;
        const grid_map::Index gi(cell[i] % rows, cell[i] / rows);
        if (cls[i] == GG_CLASS_KEPT)
            point_index.push_back(std::make_pair(i, gi));
        else if (cls[i] == GG_CLASS_IGNORED)
            ignored.push_back(std::make_pair(i, gi));
        else
            outliers.push_back(i);
    }
}

// The stage functions are public in the header (.h:59-62) but have no caller outside filter_cloud; on the device they are
// stages of one fused launch sequence, not separately callable.  They are defined so that the library exports every
// symbol the header declares.
void GroundSegmentation::detect_ground_patches(grid_map::GridMap &, unsigned short) const
{
    ROS_ERROR("groundgrid_hip: detect_ground_patches is part of filter_cloud on the device and cannot be called on its own");
}
template <int S> void GroundSegmentation::detect_ground_patch(grid_map::GridMap &, size_t, size_t) const
{
    ROS_ERROR("groundgrid_hip: detect_ground_patch is part of filter_cloud on the device and cannot be called on its own");
}
template void GroundSegmentation::detect_ground_patch<3>(grid_map::GridMap &, size_t, size_t) const;
template void GroundSegmentation::detect_ground_patch<5>(grid_map::GridMap &, size_t, size_t) const;
void GroundSegmentation::spiral_ground_interpolation(grid_map::GridMap &, const geometry_msgs::TransformStamped &) const
{
    ROS_ERROR("groundgrid_hip: spiral_ground_interpolation is part of filter_cloud on the device and cannot be called on its own");
}
void GroundSegmentation::interpolate_cell(grid_map::GridMap &, const size_t, const size_t) const
{
    ROS_ERROR("groundgrid_hip: interpolate_cell is part of filter_cloud on the device and cannot be called on its own");
}

} // namespace groundgrid
