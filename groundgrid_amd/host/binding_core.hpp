// binding_core.hpp -- what the reference-typed bindings (ros/GroundSegmentationHip.cpp, ros/GroundGridHip.cpp) do between the
// reference's class interfaces and the C ABI of include/groundgrid_hip.h, with no ROS / PCL / grid_map type in sight: plain
// pointers in, plain pointers out.  The two ROS translation units are thin shells around it (they convert the reference's types
// and nothing else), so everything that can go wrong -- per-object contexts, who owns the map state, which layers travel -- is
// exercised on the GPU by tests/cpp/test_binding_core.cpp without a ROS installation.
//
// Map ownership, two modes per map:
//   * HOST-MANAGED (the reference's division of labour): GroundGrid::update edits `ground` / `groundpatch` on the host between
//     clouds (src/GroundGrid.cpp:97-143).  filter() uploads the two layers whenever the map's position changed since the last
//     call (update changes layer contents only together with a move) and downloads the requested layers after every call.
//   * DEVICE-RESIDENT: GroundGrid::update runs on the device as well (reset_map / move_map = gg_reset_map / gg_move_map); the
//     persistent layers never leave HBM, and only the layers somebody has subscribed to are downloaded.
// One Core per groundgrid::GroundSegmentation OBJECT (the reference's function-local statics bind the first map ever passed,
// src/GroundSegmentation.cpp:76-78,203-213, so a process could only ever serve one; SURVEY H7 asks not to copy that): the
// registry below maps objects -- and, for device-resident maps, the grid_map::GridMap objects GroundGrid hands to filter_cloud
// -- to their Core.
#pragma once

#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "groundgrid_hip.h"

namespace groundgrid_hip {

inline const char *const *layer_names()
{
    static const char *const names[GG_NUM_LAYERS] = {"points",           "ground",    "groundpatch", "minGroundHeight",
                                                     "maxGroundHeight", "groundCandidates", "planeDist",   "m2",
                                                     "meanVariance",    "pointsRaw", "variance"};
    return names;
}

constexpr unsigned LAYERS_ALL = (1u << GG_NUM_LAYERS) - 1u;
constexpr unsigned LAYERS_NONE = 0u;
// what GroundGrid::update and the terrain image read (src/GroundGrid.cpp:130-131, src/GroundGridNodelet.cpp:251-253)
constexpr unsigned LAYERS_STATE = (1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH) | (1u << GG_LAYER_POINTS) | (1u << GG_LAYER_POINTSRAW);

// GROUNDGRID_HIP_LAYERS: "all", "state", "none" or a comma-separated list of layer names -> bit mask over gg_layer.
// `fallback` when the variable is not set (a nodelet that knows its subscribers passes the mask itself instead).
inline unsigned layers_from_env(unsigned fallback)
{
    const char *e = std::getenv("GROUNDGRID_HIP_LAYERS");
    if (!e || !*e) return fallback;
    if (!std::strcmp(e, "all")) return LAYERS_ALL;
    if (!std::strcmp(e, "state")) return LAYERS_STATE;
    if (!std::strcmp(e, "none")) return LAYERS_NONE;
    unsigned mask = 0u;
    std::string s(e);
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t end = s.find(',', pos) == std::string::npos ? s.size() : s.find(',', pos);
        const std::string name = s.substr(pos, end - pos);
        for (int l = 0; l < GG_NUM_LAYERS; ++l)
            if (name == layer_names()[l]) mask |= 1u << l;
        pos = end + 1;
    }
    return mask;
}

// the caller's grid_map::GridMap as plain data: position and one column-major plane per layer (nullptr: the map does not hold
// it / the caller does not want it written)
struct MapView {
    double pos_x = 0.0, pos_y = 0.0;
    float *layer[GG_NUM_LAYERS] = {};
};

class Core {
  public:
    Core() = default;
    Core(const Core &) = delete;
    Core &operator=(const Core &) = delete;
    ~Core() { gg_destroy(ctx_); }

    // GroundSegmentation::init: one map state, room for `capacity` points per cloud (grows on demand, ensure_capacity)
    bool create(const gg_geometry &geometry, size_t capacity, int device = 0)
    {
        geometry_ = geometry;
        device_ = device;
        return recreate(capacity);
    }
    bool ok() const { return ctx_ != nullptr; }
    gg_context *context() const { return ctx_; }
    const std::string &last_error() const { return error_; }
    size_t capacity() const { return capacity_; }
    bool device_resident() const { return device_resident_; }

    void set_config(const gg_config &c)
    {
        config_ = c; // (kept: a context re-created for a larger cloud gets the same configuration)
        have_config_ = true;
        if (ctx_ && gg_set_config(ctx_, &c) != GG_OK) note("gg_set_config");
    }

    // ---- device-resident map (GroundGrid on the device) ----
    // GroundGrid::initGroundGrid (src/GroundGrid.cpp:50-80): fresh layers around (x, y), ground := odom z
    int reset_map(double x, double y, float odom_z)
    {
        if (!ctx_) return GG_ERR_INVALID;
        const int rc = gg_reset_map(ctx_, 0, x, y, odom_z);
        if (rc != GG_OK) return note("gg_reset_map"), rc;
        device_resident_ = true;
        have_position_ = true;
        return gg_get_map_position(ctx_, 0, &pos_x_, &pos_y_);
    }
    // GroundGrid::update (src/GroundGrid.cpp:83-147): the map follows the vehicle, the exposed cells are re-seeded on the device.
    // base_plane = third row of the base_link <- map rotation and its z translation (include/groundgrid_hip.h gg_move_map).
    // Returns the status; *moved, the cell shift and the map's new (snapped) position come back for the host-side GridMap.
    int move_map(double odom_x, double odom_y, const double base_plane[4], bool *moved, double *snapped_x, double *snapped_y)
    {
        if (!ctx_ || !device_resident_) return GG_ERR_INVALID;
        int shift[2] = {0, 0};
        const int rc = gg_move_map(ctx_, 0, odom_x, odom_y, base_plane, shift);
        if (rc != GG_OK) return note("gg_move_map"), rc;
        if (moved) *moved = shift[0] != 0 || shift[1] != 0;
        const int rc2 = gg_get_map_position(ctx_, 0, &pos_x_, &pos_y_);
        if (snapped_x) *snapped_x = pos_x_;
        if (snapped_y) *snapped_y = pos_y_;
        return rc2;
    }

    // ---- GroundSegmentation::filter_cloud (src/GroundSegmentation.cpp:50-197) ----
    // out must have room for n points; *n_out = points of the returned cloud.  `download` = layers written back into view.layer
    // after the call (bit mask over gg_layer; layers whose pointer is null are skipped).
    int filter(const MapView &view, unsigned download, const gg_point32 *cloud, size_t n, const float origin[3], double base_z, gg_point32 *out,
               size_t *n_out)
    {
        if (n_out) *n_out = 0;
        if (!ctx_) return GG_ERR_INVALID;
        if (!ensure_capacity(n)) return GG_ERR_NOMEM;
        if (!device_resident_ && (!have_position_ || view.pos_x != pos_x_ || view.pos_y != pos_y_)) {
            // state the host may have edited since the last cloud: GroundGrid::update moves the map and seeds the exposed cells
            if (!view.layer[GG_LAYER_GROUND] || !view.layer[GG_LAYER_GROUNDPATCH]) return error_ = "host-managed map without ground / groundpatch planes", GG_ERR_INVALID;
            int rc = gg_set_map_position(ctx_, 0, view.pos_x, view.pos_y);
            if (rc == GG_OK) rc = gg_set_layer(ctx_, 0, GG_LAYER_GROUND, view.layer[GG_LAYER_GROUND]); // (Eigen::MatrixXf is column-major: as is)
            if (rc == GG_OK) rc = gg_set_layer(ctx_, 0, GG_LAYER_GROUNDPATCH, view.layer[GG_LAYER_GROUNDPATCH]);
            if (rc != GG_OK) { // the device would keep filtering against a stale terrain: report and try the upload again next time
                have_position_ = false;
                return note("uploading the map state"), rc;
            }
            have_position_ = true;
            pos_x_ = view.pos_x;
            pos_y_ = view.pos_y;
        }
        // the cloud and the layers somebody reads afterwards as ONE call: the eight layers the insertion finishes travel while the
        // stencil and the sweep run, the other three while the returned cloud is assembled (gg_filter_cloud_layers)
        float *dst[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) dst[l] = ((download >> l) & 1u) ? view.layer[l] : nullptr;
        if (pin_planes_) pin(dst);
        size_t got = 0;
        const int rc = gg_filter_cloud_layers(ctx_, 0, cloud, n, nullptr, origin, base_z, out, &got, nullptr, nullptr, dst);
        if (rc != GG_OK) return note("gg_filter_cloud_layers"), rc;
        if (n_out) *n_out = got;
        return GG_OK;
    }

    // Planes the device may write directly (gg_host_register): only for a map whose owner tells the Core when its planes go away
    // (release_planes) -- a registration that outlives its memory would send a later download to pages nobody reads.
    void set_pin_planes(bool on)
    {
        if (!on) release_planes();
        pin_planes_ = on;
    }
    void release_planes()
    {
        for (void *p : pinned_)
            if (ctx_) gg_host_unregister(ctx_, p);
        pinned_.clear();
    }

    // ---- the stage members (include/groundgrid/GroundSegmentation.h:59-62) on the caller's map ----
    // A host-managed map holds the layers the stage reads on the host: they are uploaded first (detect_ground_patches /
    // detect_ground_patch<S>: points, m2, minGroundHeight, variance, ground, groundpatch; the sweep and interpolate_cell: ground,
    // groundpatch); a device-resident map already has them.  What the stage writes -- ground, groundpatch, and `variance` for
    // detect_ground_patches (:323) -- is written back into the view's planes either way: these are rare, explicit calls.
    int run_stage(const MapView &view, int stage, int section, int i, int j, double base_z)
    {
        if (!ctx_) return GG_ERR_INVALID;
        const bool detect = stage == GG_STAGE_DETECT_GROUND_PATCHES || stage == GG_STAGE_DETECT_GROUND_PATCH_3 || stage == GG_STAGE_DETECT_GROUND_PATCH_5;
        const unsigned reads = (1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH) |
                               (detect ? (1u << GG_LAYER_POINTS) | (1u << GG_LAYER_M2) | (1u << GG_LAYER_MINGROUNDHEIGHT) | (1u << GG_LAYER_VARIANCE) : 0u);
        const unsigned writes = (1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH) | (stage == GG_STAGE_DETECT_GROUND_PATCHES ? 1u << GG_LAYER_VARIANCE : 0u);
        if (!device_resident_) {
            int rc = gg_set_map_position(ctx_, 0, view.pos_x, view.pos_y);
            for (int l = 0; l < GG_NUM_LAYERS && rc == GG_OK; ++l)
                if (((reads >> l) & 1u) && view.layer[l]) rc = gg_set_layer(ctx_, 0, l, view.layer[l]);
            if (rc != GG_OK) return note("uploading the layers of a stage"), rc;
            have_position_ = true;
            pos_x_ = view.pos_x;
            pos_y_ = view.pos_y;
        }
        gg_stage_args a{};
        a.section = section;
        a.i = i;
        a.j = j;
        a.base_z = base_z;
        const int rc = gg_run_stage(ctx_, 0, stage, &a);
        if (rc != GG_OK) return note("gg_run_stage"), rc;
        float *dst[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) dst[l] = ((writes >> l) & 1u) ? view.layer[l] : nullptr;
        const int rc_down = gg_get_layers(ctx_, 0, dst);
        if (rc_down != GG_OK) note("downloading the layers of a stage");
        return rc_down;
    }

    // GroundSegmentation::insert_cloud (.h:55, :200-311) on the caller's map as it stands: a host-managed map's ground / groundpatch and the
    // eight layers the insertion reads and continues are uploaded first, what it wrote comes back into the view's planes; per point of
    // [start, end) the GG_CLASS_* and the cell (gg_insert_cloud).
    int insert(const MapView &view, const gg_point32 *cloud, size_t start, size_t end, const float origin[3], uint8_t *cls, int32_t *cell)
    {
        if (!ctx_) return GG_ERR_INVALID;
        if (end > start && !ensure_capacity(end - start)) return GG_ERR_CAPACITY;
        const unsigned writes = (1u << GG_LAYER_POINTS) | (1u << GG_LAYER_POINTSRAW) | (1u << GG_LAYER_GROUNDCANDIDATES) | (1u << GG_LAYER_MEANVARIANCE) |
                                (1u << GG_LAYER_PLANEDIST) | (1u << GG_LAYER_M2) | (1u << GG_LAYER_MAXGROUNDHEIGHT) | (1u << GG_LAYER_MINGROUNDHEIGHT);
        const unsigned reads = writes | (1u << GG_LAYER_GROUND) | (1u << GG_LAYER_GROUNDPATCH);
        if (!device_resident_) {
            int rc = gg_set_map_position(ctx_, 0, view.pos_x, view.pos_y);
            for (int l = 0; l < GG_NUM_LAYERS && rc == GG_OK; ++l)
                if (((reads >> l) & 1u) && view.layer[l]) rc = gg_set_layer(ctx_, 0, l, view.layer[l]);
            if (rc != GG_OK) return note("uploading the layers of insert_cloud"), rc;
            have_position_ = true;
            pos_x_ = view.pos_x;
            pos_y_ = view.pos_y;
        }
        const int rc = gg_insert_cloud(ctx_, 0, cloud, start, end, origin, cls, cell);
        if (rc != GG_OK) return note("gg_insert_cloud"), rc;
        float *dst[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) dst[l] = ((writes >> l) & 1u) ? view.layer[l] : nullptr;
        const int rc_down = gg_get_layers(ctx_, 0, dst);
        if (rc_down != GG_OK) note("downloading the layers of insert_cloud");
        return rc_down;
    }

    // a cloud larger than the context was created for: re-create it with headroom.  A device-resident map is carried over
    // (ground, groundpatch and the position take the round trip through the host once); a host-managed one is uploaded again
    // by the next filter() anyway.
    bool ensure_capacity(size_t points)
    {
        if (ctx_ && points <= capacity_) return true;
        const size_t want = points + points / 2; // (sensor clouds vary by a few percent from revolution to revolution)
        std::vector<float> ground, patch;
        double px = 0.0, py = 0.0;
        const bool carry = ctx_ && device_resident_;
        if (carry) {
            int rows = 0, cols = 0;
            gg_get_size(ctx_, &rows, &cols);
            ground.resize((size_t)rows * cols);
            patch.resize((size_t)rows * cols);
            if (gg_get_layer(ctx_, 0, GG_LAYER_GROUND, ground.data()) != GG_OK || gg_get_layer(ctx_, 0, GG_LAYER_GROUNDPATCH, patch.data()) != GG_OK ||
                gg_get_map_position(ctx_, 0, &px, &py) != GG_OK)
                return note("saving the device-resident map"), false;
        }
        if (!recreate(want)) return false;
        if (carry) {
            if (gg_set_map_position(ctx_, 0, px, py) != GG_OK || gg_set_layer(ctx_, 0, GG_LAYER_GROUND, ground.data()) != GG_OK ||
                gg_set_layer(ctx_, 0, GG_LAYER_GROUNDPATCH, patch.data()) != GG_OK)
                return note("restoring the device-resident map"), false;
            device_resident_ = true;
            have_position_ = true;
            pos_x_ = px;
            pos_y_ = py;
        }
        return true;
    }

  private:
    void pin(float *const dst[GG_NUM_LAYERS])
    {
        int rows = 0, cols = 0;
        gg_get_size(ctx_, &rows, &cols);
        for (int l = 0; l < GG_NUM_LAYERS; ++l) {
            if (!dst[l]) continue;
            bool known = false;
            for (void *p : pinned_) known |= p == dst[l];
            if (known) continue;
            if (pinned_.size() >= 4 * GG_NUM_LAYERS) { // planes that keep changing their address: not worth pinning, and never a leak
                set_pin_planes(false);
                return;
            }
            if (gg_host_register(ctx_, dst[l], (size_t)rows * cols * sizeof(float)) == GG_OK) pinned_.push_back(dst[l]);
        }
    }
    bool recreate(size_t capacity)
    {
        pinned_.clear(); // (gg_destroy unregisters what the old context had registered)
        if (gg_abi_version() != GG_ABI_VERSION) return error_ = "libgroundgrid_hip.so and groundgrid_hip.h disagree on the ABI version", false;
        gg_destroy(ctx_);
        ctx_ = nullptr;
        have_position_ = false;
        device_resident_ = false;
        capacity_ = 0;
        const int rc = gg_create(&geometry_, 1, capacity, device_, &ctx_);
        if (rc != GG_OK) {
            ctx_ = nullptr;
            error_ = "gg_create failed with status " + std::to_string(rc) + " (no gfx950 device, out of memory, or grid_map and init() disagree on the cell count)";
            return false;
        }
        capacity_ = capacity;
        if (have_config_ && gg_set_config(ctx_, &config_) != GG_OK) note("gg_set_config");
        return true;
    }
    void note(const char *what) { error_ = std::string(what) + ": " + (ctx_ ? gg_last_error(ctx_) : "no context"); }

    gg_context *ctx_ = nullptr;
    gg_geometry geometry_{};
    gg_config config_{};
    bool have_config_ = false;
    int device_ = 0;
    size_t capacity_ = 0;
    bool have_position_ = false, device_resident_ = false;
    bool pin_planes_ = false;
    std::vector<void *> pinned_;
    double pos_x_ = 0.0, pos_y_ = 0.0;
    std::string error_;
};

// Who serves what.  Keys are object addresses (the reference's headers cannot grow members): a GroundSegmentation object owns
// its Core from init() on; a GroundGrid object that keeps its map on the device binds the map object it hands out to the Core
// that will filter against it -- the first Core that has no map yet, i.e. the GroundSegmentation of the same nodelet
// (src/GroundGridNodelet.cpp:89-95 constructs the pair).
// Lifetime.  The reference's GroundSegmentation has no destructor to hook (its header cannot change), so a Core outlives its
// object until the address is used again -- init() on a known address re-creates the context and drops every map bound to it -- or until
// the host calls forget_object (a nodelet's own destructor can: groundgrid_hip::Registry::instance().forget_object(&ground_segmentation_)).
// GroundGrid does have a destructor: it unbinds its map (forget_map), and initGroundGrid unbinds the map it replaces.  The registry
// itself is never destroyed: tearing down HIP contexts from a static destructor can run after the HIP runtime is gone.
class Registry {
  public:
    static Registry &instance()
    {
        static Registry *const r = new Registry(); // (leaked on purpose, see above)
        return *r;
    }
    Core *core_of_object(const void *segmentation, bool create)
    {
        std::lock_guard<std::mutex> g(m_);
        auto it = by_object_.find(segmentation);
        if (it != by_object_.end()) return it->second.get();
        if (!create) return nullptr;
        Core *c = new Core();
        by_object_[segmentation].reset(c);
        order_.push_back(c);
        return c;
    }
    Core *core_of_map(const void *map)
    {
        std::lock_guard<std::mutex> g(m_);
        auto it = by_map_.find(map);
        return it == by_map_.end() ? nullptr : it->second;
    }
    // the oldest created Core that serves no device-resident map yet; binds it to `map`
    Core *bind_map(const void *map)
    {
        std::lock_guard<std::mutex> g(m_);
        auto it = by_map_.find(map);
        if (it != by_map_.end()) return it->second;
        for (Core *c : order_) {
            bool taken = false;
            for (const auto &kv : by_map_) taken |= kv.second == c;
            if (!taken && c->ok()) return by_map_[map] = c;
        }
        return nullptr;
    }
    // the map object is going away (GroundGrid::~GroundGrid) or being replaced (initGroundGrid): its Core serves no map again
    void forget_map(const void *map)
    {
        std::lock_guard<std::mutex> g(m_);
        auto it = by_map_.find(map);
        if (it == by_map_.end()) return;
        it->second->release_planes(); // (the map's planes are about to be freed)
        by_map_.erase(it);
    }
    // init() on an object that already has a Core (a re-initialised nodelet, or a new object at a recycled address): the maps bound
    // to the old context name nothing any more
    void unbind_maps_of(const Core *c)
    {
        std::lock_guard<std::mutex> g(m_);
        for (auto m = by_map_.begin(); m != by_map_.end();) m = m->second == c ? by_map_.erase(m) : std::next(m);
    }
    void forget_object(const void *segmentation)
    {
        std::lock_guard<std::mutex> g(m_);
        auto it = by_object_.find(segmentation);
        if (it == by_object_.end()) return;
        Core *c = it->second.get();
        for (auto m = by_map_.begin(); m != by_map_.end();) m = m->second == c ? by_map_.erase(m) : std::next(m);
        for (auto o = order_.begin(); o != order_.end();) o = *o == c ? order_.erase(o) : std::next(o);
        by_object_.erase(it);
    }

  private:
    std::mutex m_;
    std::unordered_map<const void *, std::unique_ptr<Core>> by_object_;
    std::unordered_map<const void *, Core *> by_map_;
    std::vector<Core *> order_;
};

} // namespace groundgrid_hip
