// GroundSegmentation.hpp -- C++ host-side mirror of groundgrid::GroundSegmentation on top of the C ABI.
//
// Same method names, argument meaning and call sequence as the reference class
// (/root/reference/include/groundgrid/GroundSegmentation.h:48-71):
//     init(dimension, resolution)            .h:53   (the unused ros::NodeHandle& is dropped)
//     setConfig(config)                      .h:56
//     filter_cloud(cloud, origin, mapToBase, map) -> cloud        .h:54
//     insert_cloud(cloud, start, end, origin, point_index, ignored, outliers, map)   .h:55
//     detect_ground_patches(map, section), detect_ground_patch<S>(map, i, j), spiral_ground_interpolation(map, toBase),
//     interpolate_cell(map, x, y)            .h:59-62
// but with dependency-free value types, because PCL / grid_map / ROS headers do not exist in this image:
//     pcl::PointCloud<PointXYZIR>::Ptr        -> std::vector<gg_point32>          (identical 32-byte records)
//     geometry_msgs::TransformStamped         -> double mapToBase_z              (the only field used, .cpp:406-411)
//     grid_map::GridMap&                      -> groundgrid_hip::GridMap         (same layers, column-major floats)
// INTEGRATION.md shows the 40-line variant with the real ROS types that a maintainer drops into the nodelet build.
//
// Header-only; link with -lgroundgrid_hip.  Errors throw std::runtime_error (the reference has no error path at all).
#pragma once

#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "groundgrid_hip.h"

namespace groundgrid_hip {

// The reference's map is a grid_map::GridMap owned by GroundGrid and borrowed by filter_cloud.  Here the
// authoritative copy of the layers lives in HBM (one slot of the context); this object is the host-side view:
// layer(id) downloads lazily, setLayer(id, data) uploads (GroundGrid::update edits `ground` / `groundpatch` of newly
// exposed cells between clouds, src/GroundGrid.cpp:121-133).
class GridMap {
  public:
    GridMap(gg_context *ctx, int slot) : ctx_(ctx), slot_(slot) { gg_get_size(ctx_, &rows_, &cols_); }
    int rows() const { return rows_; }
    int cols() const { return cols_; }
    int slot() const { return slot_; }
    // grid_map::GridMap::setGeometry position + GroundGrid::initGroundGrid layer values (src/GroundGrid.cpp:58,71-75)
    void reset(double pos_x, double pos_y, float odom_z) { check(gg_reset_map(ctx_, slot_, pos_x, pos_y, odom_z), "gg_reset_map"); }
    // grid_map::GridMap::move result (src/GroundGrid.cpp:97)
    void setPosition(double x, double y) { check(gg_set_map_position(ctx_, slot_, x, y), "gg_set_map_position"); }
    // map[layer] as a column-major rows x cols matrix (Eigen::MatrixXf layout)
    std::vector<float> layer(gg_layer id) const
    {
        std::vector<float> v((size_t)rows_ * cols_);
        check(gg_get_layer(ctx_, slot_, id, v.data()), "gg_get_layer");
        return v;
    }
    // all 11 layers with one synchronisation (what a publisher loop reads after a cloud, src/GroundGridNodelet.cpp:211-224)
    std::vector<std::vector<float>> layers() const
    {
        std::vector<std::vector<float>> all((size_t)GG_NUM_LAYERS, std::vector<float>((size_t)rows_ * cols_));
        float *dst[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) dst[l] = all[(size_t)l].data();
        check(gg_get_layers(ctx_, slot_, dst), "gg_get_layers");
        return all;
    }
    void setLayer(gg_layer id, const std::vector<float> &v)
    {
        if (v.size() != (size_t)rows_ * cols_) throw std::runtime_error("setLayer: size mismatch");
        check(gg_set_layer(ctx_, slot_, id, v.data()), "gg_set_layer");
    }

  private:
    void check(int rc, const char *what) const
    {
        if (rc != GG_OK) throw std::runtime_error(std::string(what) + ": " + gg_last_error(ctx_));
    }
    gg_context *ctx_;
    int slot_, rows_ = 0, cols_ = 0;
};

class GroundSegmentation {
  public:
    typedef gg_point32 PCLPoint; // velodyne_pointcloud::PointXYZIR
    typedef std::array<int, 2> Index; // grid_map::Index

    GroundSegmentation() = default;
    GroundSegmentation(const GroundSegmentation &) = delete;
    GroundSegmentation &operator=(const GroundSegmentation &) = delete;
    ~GroundSegmentation() { gg_destroy(ctx_); }

    // GroundSegmentation::init (src/GroundSegmentation.cpp:37-48).  n_maps independent map states, max_points per cloud.
    void init(const size_t dimension, const float &resolution, int n_maps = 1, size_t max_points = 200000, int device = 0)
    {
        gg_destroy(ctx_);
        ctx_ = nullptr;
        // (the structs this header passes by pointer -- gg_batch, gg_config -- are this version's: a library of another version
        // would read past them)
        if (gg_abi_version() != GG_ABI_VERSION)
            throw std::runtime_error("libgroundgrid_hip.so has ABI version " + std::to_string(gg_abi_version()) + ", this header is version " +
                                     std::to_string(GG_ABI_VERSION));
        gg_geometry g;
        gg_default_geometry(&g);
        g.length = (float)dimension; // the nodelet passes 120.0f (Nodelet.cpp:95)
        g.resolution = resolution;
        const int rc = gg_create(&g, n_maps, max_points, device, &ctx_);
        if (rc != GG_OK) throw std::runtime_error("gg_create failed with status " + std::to_string(rc));
        maps_.clear();
        for (int s = 0; s < n_maps; ++s) maps_.emplace_back(ctx_, s);
    }

    GridMap &map(int slot = 0) { return maps_.at((size_t)slot); }

    // GroundSegmentation::setConfig (src/GroundSegmentation.cpp:468-471)
    void setConfig(const gg_config &config)
    {
        if (gg_set_config(ctx_, &config) != GG_OK) throw std::runtime_error("gg_set_config");
    }

    // GroundSegmentation::filter_cloud (src/GroundSegmentation.cpp:50-197)
    std::vector<PCLPoint> filter_cloud(const std::vector<PCLPoint> &cloud, const PCLPoint &cloudOrigin, double mapToBase_z, GridMap &map)
    {
        std::vector<PCLPoint> out(cloud.size());
        const float origin[3] = {cloudOrigin.x, cloudOrigin.y, cloudOrigin.z};
        size_t out_n = 0;
        labels_.resize(cloud.size());
        index_.resize(cloud.size());
        const int rc = gg_filter_cloud(ctx_, map.slot(), cloud.data(), cloud.size(), origin, mapToBase_z, out.data(), &out_n,
                                       labels_.data(), index_.data());
        if (rc != GG_OK) throw std::runtime_error(std::string("gg_filter_cloud: ") + gg_last_error(ctx_));
        out.resize(out_n);
        last_n_ = cloud.size();
        return out;
    }
    // BASELINE.json's north_star names the entry point segment(); the reference has no such function -- alias.
    std::vector<PCLPoint> segment(const std::vector<PCLPoint> &cloud, const PCLPoint &cloudOrigin, double mapToBase_z, GridMap &map)
    {
        return filter_cloud(cloud, cloudOrigin, mapToBase_z, map);
    }

    // GroundSegmentation::insert_cloud (include/groundgrid/GroundSegmentation.h:55, src/GroundSegmentation.cpp:200-311): the points
    // cloud[start, end) INTO `map` as it stands -- pointsRaw and the recurrences of :296-309 continue from what the layers hold, no reset
    // -- and appended to the three lists in cloud order (gg_insert_cloud).
    void insert_cloud(const std::vector<PCLPoint> &cloud, const size_t start, const size_t end, const PCLPoint &cloudOrigin,
                      std::vector<std::pair<size_t, Index>> &point_index, std::vector<std::pair<size_t, Index>> &ignored, std::vector<size_t> &outliers,
                      GridMap &map)
    {
        const size_t last = end < cloud.size() ? end : cloud.size();
        if (start >= last) return;
        std::vector<uint8_t> cls(last - start);
        std::vector<int32_t> cell(last - start);
        const float origin[3] = {cloudOrigin.x, cloudOrigin.y, cloudOrigin.z};
        if (gg_insert_cloud(ctx_, map.slot(), reinterpret_cast<const gg_point32 *>(cloud.data()), start, last, origin, cls.data(), cell.data()) != GG_OK)
            throw std::runtime_error(std::string("gg_insert_cloud: ") + gg_last_error(ctx_));
        append_lists(cls, cell, start, map.rows(), point_index, ignored, outliers);
    }
    // ... and the per-point decisions of the LAST filter_cloud call on `map`, restricted to [start, end), in the same three lists (nothing
    // is inserted: gg_get_point_classes)
    void last_call_lists(const size_t start, const size_t end, std::vector<std::pair<size_t, Index>> &point_index,
                         std::vector<std::pair<size_t, Index>> &ignored, std::vector<size_t> &outliers, GridMap &map)
    {
        std::vector<uint8_t> cls(last_n_);
        std::vector<int32_t> cell(last_n_);
        if (gg_get_point_classes(ctx_, map.slot(), last_n_, cls.data(), cell.data()) != GG_OK)
            throw std::runtime_error("gg_get_point_classes");
        const size_t last = end < last_n_ ? end : last_n_;
        if (start >= last) return;
        append_lists(std::vector<uint8_t>(cls.begin() + start, cls.begin() + last), std::vector<int32_t>(cell.begin() + start, cell.begin() + last), start,
                     map.rows(), point_index, ignored, outliers);
    }

    // The stage members (include/groundgrid/GroundSegmentation.h:59-62), each on `map` as it stands (gg_run_stage)
    void detect_ground_patches(GridMap &map, unsigned short section) const { stage(map, GG_STAGE_DETECT_GROUND_PATCHES, section, 0, 0, 0.0); }
    template <int S> void detect_ground_patch(GridMap &map, size_t i, size_t j) const
    {
        static_assert(S == 3 || S == 5, "the reference instantiates detect_ground_patch<3> and <5> (src/GroundSegmentation.cpp:335,337)");
        stage(map, S == 3 ? GG_STAGE_DETECT_GROUND_PATCH_3 : GG_STAGE_DETECT_GROUND_PATCH_5, 0, (int)i, (int)j, 0.0);
    }
    void spiral_ground_interpolation(GridMap &map, double toBase_z) const { stage(map, GG_STAGE_SPIRAL_GROUND_INTERPOLATION, 0, 0, 0, toBase_z); }
    void interpolate_cell(GridMap &map, const size_t x, const size_t y) const { stage(map, GG_STAGE_INTERPOLATE_CELL, 0, (int)x, (int)y, 0.0); }

    // per input point of the last filter_cloud call: GG_LABEL_* and position in the returned cloud (-1: dropped)
    const std::vector<uint8_t> &labels() const { return labels_; }
    const std::vector<int32_t> &out_index() const { return index_; }
    gg_context *context() { return ctx_; }

  private:
    static void append_lists(const std::vector<uint8_t> &cls, const std::vector<int32_t> &cell, size_t start, int rows,
                             std::vector<std::pair<size_t, Index>> &point_index, std::vector<std::pair<size_t, Index>> &ignored, std::vector<size_t> &outliers)
    {
        for (size_t k = 0; k < cls.size(); ++k) {
            const Index gi = {cell[k] % rows, cell[k] / rows};
            if (cls[k] == GG_CLASS_KEPT) point_index.emplace_back(start + k, gi);
            else if (cls[k] == GG_CLASS_IGNORED) ignored.emplace_back(start + k, gi);
            else if (cls[k] == GG_CLASS_OUTLIER) outliers.push_back(start + k);
        }
    }
    void stage(GridMap &map, int which, int section, int i, int j, double base_z) const
    {
        gg_stage_args a{};
        a.section = section;
        a.i = i;
        a.j = j;
        a.base_z = base_z;
        if (gg_run_stage(ctx_, map.slot(), which, &a) != GG_OK) throw std::runtime_error(std::string("gg_run_stage: ") + gg_last_error(ctx_));
    }
    gg_context *ctx_ = nullptr;
    std::vector<GridMap> maps_;
    std::vector<uint8_t> labels_;
    std::vector<int32_t> index_;
    size_t last_n_ = 0;
};

} // namespace groundgrid_hip
