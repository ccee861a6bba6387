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
// Map ownership, per map (groundgrid_amd/host/binding_core.hpp does the work, this file only converts types):
//   * HOST-MANAGED, the reference's division of labour: GroundGrid::update edits `ground` / `groundpatch` on the host between
//     clouds (src/GroundGrid.cpp:97-143).  The binding uploads the two layers only when the host can have changed them -- the first
//     call, and whenever the map position differs from the one seen last (update changes layer contents only together with a
//     move) -- and downloads after every call the layers selected by GROUNDGRID_HIP_LAYERS (default "all": the nodelet publishes
//     every layer, src/GroundGridNodelet.cpp:211-224; "state" = ground, groundpatch, points, pointsRaw: what GroundGrid::update
//     and the terrain image read, :251-253; "none"; or a comma-separated list of layer names).
//   * DEVICE-RESIDENT: when ros/GroundGridHip.cpp is compiled instead of src/GroundGrid.cpp as well, GroundGrid::update runs on
//     the device (gg_move_map) and binds the map object it returns to this object's context: nothing is uploaded, and only the
//     layers somebody subscribed to come back (default "state" unless GROUNDGRID_HIP_LAYERS says otherwise; a warning says so once).
// One device context per GroundSegmentation OBJECT (keyed by `this`: the reference's header cannot grow a member): two objects
// in one process keep separate maps, unlike the reference's function-local statics (src/GroundSegmentation.cpp:76-78,203-213).
#include <cstdlib>
#include <cstring>

#include <groundgrid/GroundSegmentation.h>

#include "../binding_core.hpp"

namespace groundgrid {

namespace {

using groundgrid_hip::Core;
using groundgrid_hip::Registry;

size_t initial_capacity()
{
    // GROUNDGRID_HIP_MAX_POINTS: expected cloud size (default 400 000: three times an HDL-64E revolution).  Only a starting point:
    // a larger cloud makes the binding re-create the context with room for it.
    const char *e = std::getenv("GROUNDGRID_HIP_MAX_POINTS");
    const long v = e ? std::atol(e) : 0;
    return v > 0 ? static_cast<size_t>(v) : 400000;
}

static_assert(sizeof(velodyne_pointcloud::PointXYZIR) == sizeof(gg_point32), "PointXYZIR must be the 32-byte record of point_types.h:27-33");

} // namespace

// src/GroundSegmentation.cpp:37-48
void GroundSegmentation::init(ros::NodeHandle &nodeHandle, const size_t dimension, const float &resolution)
{
    (void)nodeHandle; // unused by the reference as well
    gg_geometry geometry;
    gg_default_geometry(&geometry);
    geometry.length = static_cast<float>(dimension); // the nodelet passes 120.0f into the size_t parameter (Nodelet.cpp:95)
    geometry.resolution = resolution;
    geometry.vertical_point_ang_dist = verticalPointAngDist;
    geometry.min_dist_squared = minDistSquared;
    Core *core = Registry::instance().core_of_object(this, true);
    if (core->ok()) Registry::instance().unbind_maps_of(core); // a second init(), or a new object at a recycled address: start clean
    if (!core->create(geometry, initial_capacity())) {
        ROS_FATAL("groundgrid_hip: %s", core->last_error().c_str());
        return;
    }
    int rows = 0, cols = 0;
    gg_get_size(core->context(), &rows, &cols);
    expectedPoints.resize(rows, cols); // :40-46, kept for callers that inspect the member
    gg_get_expected_points(core->context(), expectedPoints.data());
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
    // (the nodelet calls this once before init(), :97: the object's Core exists from the first call on and keeps the values)
    Registry::instance().core_of_object(this, true)->set_config(k);
}

// src/GroundSegmentation.cpp:50-197
pcl::PointCloud<GroundSegmentation::PCLPoint>::Ptr GroundSegmentation::filter_cloud(const pcl::PointCloud<PCLPoint>::Ptr cloud,
                                                                                     const PCLPoint &cloudOrigin,
                                                                                     const geometry_msgs::TransformStamped &mapToBase,
                                                                                     grid_map::GridMap &map)
{
    pcl::PointCloud<PCLPoint>::Ptr filtered_cloud(new pcl::PointCloud<PCLPoint>);
    filtered_cloud->header = cloud->header;
    // a map GroundGridHip keeps on the device names its Core; any other map is host-managed and served by this object's own
    Core *core = Registry::instance().core_of_map(&map);
    if (!core) core = Registry::instance().core_of_object(this, false);
    if (!core || !core->ok()) {
        ROS_ERROR("groundgrid_hip: filter_cloud before a successful init");
        return filtered_cloud;
    }
    // the layers filter_cloud adds (:61-75) must exist for the publishers even if they are not downloaded
    const char *const *names = groundgrid_hip::layer_names();
    for (int l = 0; l < GG_NUM_LAYERS; ++l)
        if (!map.exists(names[l])) map.add(names[l], 0.0);

    groundgrid_hip::MapView view;
    view.pos_x = map.getPosition().x();
    view.pos_y = map.getPosition().y();
    for (int l = 0; l < GG_NUM_LAYERS; ++l) view.layer[l] = map[names[l]].data(); // (Eigen::MatrixXf, column-major: as the library wants it)
    // Default: host-managed maps get every layer back (the nodelet publishes the whole grid_map per cloud, Nodelet.cpp:211-214); a
    // device-resident map gets what GroundGrid::update and the terrain image read (LAYERS_STATE) -- the other planes of the host
    // object are NOT refreshed unless GROUNDGRID_HIP_LAYERS says so, and a publisher of the full grid_map should set it to "all"
    // (or publish gg_get_gridmap_message, which serialises straight from the device).
    const unsigned download = groundgrid_hip::layers_from_env(core->device_resident() ? groundgrid_hip::LAYERS_STATE : groundgrid_hip::LAYERS_ALL);
    if (core->device_resident() && download != groundgrid_hip::LAYERS_ALL)
        ROS_WARN_ONCE("groundgrid_hip: the map lives on the device; only the layers selected by GROUNDGRID_HIP_LAYERS (default: ground, groundpatch, points, "
                      "pointsRaw) are copied into the host grid_map per cloud -- set GROUNDGRID_HIP_LAYERS=all if every layer is published");

    const size_t n = cloud->points.size();
    filtered_cloud->points.resize(n);
    const float origin[3] = {cloudOrigin.x, cloudOrigin.y, cloudOrigin.z};
    size_t n_out = 0;
    const int rc = core->filter(view, download, reinterpret_cast<const gg_point32 *>(cloud->points.data()), n, origin,
                                mapToBase.transform.translation.z, // the only field of the transform the path uses, :406-411
                                reinterpret_cast<gg_point32 *>(filtered_cloud->points.data()), &n_out);
    if (rc != GG_OK) ROS_ERROR("groundgrid_hip: filter_cloud failed with status %d (%s)", rc, core->last_error().c_str());
    filtered_cloud->points.resize(n_out);
    return filtered_cloud;
}

// src/GroundSegmentation.cpp:200-311, the member itself: cloud[start, end) is classified against `map` as it stands and INSERTED -- pointsRaw
// and the recurrences of :296-309 continue in the map's layers, which a host-managed map sends up and gets back -- and the three lists
// receive the range's points in cloud order (gg_insert_cloud).  (The per-point decisions of the last filter_cloud call, without
// inserting anything, are gg_get_point_classes.)
void GroundSegmentation::insert_cloud(const pcl::PointCloud<PCLPoint>::Ptr cloud, const size_t start, const size_t end,
                                      const PCLPoint &cloudOrigin, std::vector<std::pair<size_t, grid_map::Index>> &point_index,
                                      std::vector<std::pair<size_t, grid_map::Index>> &ignored, std::vector<size_t> &outliers,
                                      grid_map::GridMap &map)
{
    Core *core = Registry::instance().core_of_map(&map);
    if (!core) core = Registry::instance().core_of_object(this, false);
    if (!core || !core->ok()) {
        ROS_ERROR("groundgrid_hip: insert_cloud was called before a successful init");
        return;
    }
    const size_t last = std::min(end, cloud->points.size());
    if (start >= last) return;
    const char *const *names = groundgrid_hip::layer_names();
    groundgrid_hip::MapView view;
    view.pos_x = map.getPosition().x();
    view.pos_y = map.getPosition().y();
    for (int l = 0; l < GG_NUM_LAYERS; ++l) view.layer[l] = map.exists(names[l]) ? map[names[l]].data() : nullptr;
    std::vector<uint8_t> cls(last - start);
    std::vector<int32_t> cell(last - start);
    const float origin[3] = {cloudOrigin.x, cloudOrigin.y, cloudOrigin.z};
    const int rc = core->insert(view, reinterpret_cast<const gg_point32 *>(cloud->points.data()), start, last, origin, cls.data(), cell.data());
    if (rc != GG_OK) {
        ROS_ERROR("groundgrid_hip: insert_cloud failed with status %d (%s)", rc, core->last_error().c_str());
        return;
    }
    int rows = 0, cols = 0;
    gg_get_size(core->context(), &rows, &cols);
    for (size_t k = 0; k < cls.size(); ++k) {
        if (cls[k] == GG_CLASS_OUTSIDE) continue;
        const grid_map::Index gi(cell[k] % rows, cell[k] / rows);
        if (cls[k] == GG_CLASS_KEPT)
            point_index.push_back(std::make_pair(start + k, gi));
        else if (cls[k] == GG_CLASS_IGNORED)
            ignored.push_back(std::make_pair(start + k, gi));
        else
            outliers.push_back(start + k);
    }
}

// The stage functions (.h:59-62).  Nothing outside filter_cloud calls them in the reference; here each runs on `map` as it stands
// through gg_run_stage -- the many-cell ones are the path's own kernels (k_patch on one quadrant, k_sweep), the single-cell ones a
// kernel of their own -- so that the library answers every public member of the class the way the reference's does.
namespace {

void run_stage(const GroundSegmentation *self, grid_map::GridMap &map, int stage, int section, int i, int j, double base_z)
{
    Core *core = Registry::instance().core_of_map(&map);
    if (!core) core = Registry::instance().core_of_object(self, false);
    if (!core || !core->ok()) {
        ROS_ERROR("groundgrid_hip: a stage of filter_cloud was called before a successful init");
        return;
    }
    const char *const *names = groundgrid_hip::layer_names();
    groundgrid_hip::MapView view;
    view.pos_x = map.getPosition().x();
    view.pos_y = map.getPosition().y();
    for (int l = 0; l < GG_NUM_LAYERS; ++l) view.layer[l] = map.exists(names[l]) ? map[names[l]].data() : nullptr;
    const int rc = core->run_stage(view, stage, section, i, j, base_z);
    if (rc != GG_OK) ROS_ERROR("groundgrid_hip: stage %d failed with status %d (%s)", stage, rc, core->last_error().c_str());
}

} // namespace

// src/GroundSegmentation.cpp:314-340
void GroundSegmentation::detect_ground_patches(grid_map::GridMap &map, unsigned short section) const
{
    run_stage(this, map, GG_STAGE_DETECT_GROUND_PATCHES, section, 0, 0, 0.0);
}
// src/GroundSegmentation.cpp:343-395
template <int S> void GroundSegmentation::detect_ground_patch(grid_map::GridMap &map, size_t i, size_t j) const
{
    run_stage(this, map, S == 3 ? GG_STAGE_DETECT_GROUND_PATCH_3 : GG_STAGE_DETECT_GROUND_PATCH_5, 0, static_cast<int>(i), static_cast<int>(j), 0.0);
}
template void GroundSegmentation::detect_ground_patch<3>(grid_map::GridMap &, size_t, size_t) const;
template void GroundSegmentation::detect_ground_patch<5>(grid_map::GridMap &, size_t, size_t) const;
// src/GroundSegmentation.cpp:398-441 (of toBase only the translation's z is used, :406-411)
void GroundSegmentation::spiral_ground_interpolation(grid_map::GridMap &map, const geometry_msgs::TransformStamped &toBase) const
{
    run_stage(this, map, GG_STAGE_SPIRAL_GROUND_INTERPOLATION, 0, 0, 0, toBase.transform.translation.z);
}
// src/GroundSegmentation.cpp:445-465
void GroundSegmentation::interpolate_cell(grid_map::GridMap &map, const size_t x, const size_t y) const
{
    run_stage(this, map, GG_STAGE_INTERPOLATE_CELL, 0, static_cast<int>(x), static_cast<int>(y), 0.0);
}

} // namespace groundgrid
