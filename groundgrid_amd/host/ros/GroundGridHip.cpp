// GroundGridHip.cpp -- the map manager on the device: compile this INSTEAD of src/GroundGrid.cpp into the `groundgrid` library
// (/root/reference/CMakeLists.txt:96-110), next to ros/GroundSegmentationHip.cpp, and link libgroundgrid_hip.so.
//
// It defines groundgrid::GroundGrid's member functions with the exact signatures of the reference's own header
// (include/groundgrid/GroundGrid.h:50-85, included below; the header, the nodelet and the launch files stay untouched).  What
// changes is WHERE the map lives: `ground` and `groundpatch` -- the only state that outlives a cloud
// (src/GroundSegmentation.cpp:243-275, :376-393) -- stay in HBM from cloud to cloud.  initGroundGrid becomes gg_reset_map,
// update becomes gg_move_map (the scroll and the seeding of the exposed cells, src/GroundGrid.cpp:97-143, as one kernel), and
// the grid_map::GridMap object update() hands to the nodelet (src/GroundGridNodelet.cpp:110 -> :196) is the map's NAME on the
// host: geometry, position, and planes that are filled only with the layers somebody subscribed to (GROUNDGRID_HIP_LAYERS).
// With the host-managed binding every map move costs two 530 KB uploads and every cloud up to eleven downloads; here a cloud
// costs its points up and its labels down.
//
// Which device context?  The map belongs with the GroundSegmentation that will filter against it: the registry of
// groundgrid_amd/host/binding_core.hpp binds the GridMap object this class creates to the first GroundSegmentation context
// that serves no map yet -- in the nodelet, the `ground_segmentation_` member constructed next to this object
// (src/GroundGridNodelet.cpp:89-95).  Several (GroundGrid, GroundSegmentation) pairs in one process pair up in construction
// order and keep separate maps.
//
// Compile-checked against the reference's header and declaration-only stand-ins of the ROS / tf2 / grid_map types
// (tests/test_ros_binding_cpu.py); the logic below the types is exercised on the GPU through binding_core.hpp
// (tests/cpp/test_binding_core.cpp).
#include <groundgrid/GroundGrid.h>

#include "../binding_core.hpp"

namespace groundgrid {

using groundgrid_hip::Core;
using groundgrid_hip::Registry;

GroundGrid::GroundGrid() : mTf2_listener(mTfBuffer) {}

GroundGrid::~GroundGrid()
{
    if (mMap_ptr) Registry::instance().forget_map(mMap_ptr.get()); // the context serves no map until a new one is bound
}

void GroundGrid::setConfig(groundgrid::GroundGridConfig &config) { config_ = config; }

// src/GroundGrid.cpp:50-80
void GroundGrid::initGroundGrid(const nav_msgs::OdometryConstPtr &inOdom)
{
    // the host object: frame, geometry and the five layers the reference creates (:55).  Their contents live on the device.
    if (mMap_ptr) Registry::instance().forget_map(mMap_ptr.get()); // (re-initialisation: the old object's address may be recycled)
    mMap_ptr = std::make_shared<grid_map::GridMap, const std::vector<std::string>>({"points", "ground", "groundpatch", "minGroundHeight", "maxGroundHeight"});
    grid_map::GridMap &map = *mMap_ptr;
    map.setFrameId("map");
    map.setGeometry(grid_map::Length(mDimension, mDimension), mResolution, grid_map::Position(inOdom->pose.pose.position.x, inOdom->pose.pose.position.y));
    ROS_INFO("Created map with size %f x %f m (%i x %i cells).", map.getLength().x(), map.getLength().y(), map.getSize()(0), map.getSize()(1));

    geometry_msgs::PoseWithCovarianceStamped odomPose;
    odomPose.pose = inOdom->pose;
    odomPose.header = inOdom->header;
    mLastPose = odomPose;

    // :71-75 on the device: points := 0, ground := odom z, groundpatch := 1e-7, minGroundHeight := 100, maxGroundHeight := -100
    Core *core = Registry::instance().bind_map(mMap_ptr.get());
    if (!core) { // GroundSegmentation::init has not run yet: update() binds the map as soon as a context exists
        ROS_WARN("groundgrid_hip: no GroundSegmentation context yet; the map is initialised on the device at the first update() after its init()");
        return;
    }
    const int rc = core->reset_map(inOdom->pose.pose.position.x, inOdom->pose.pose.position.y, static_cast<float>(inOdom->pose.pose.position.z));
    if (rc != GG_OK) ROS_ERROR("groundgrid_hip: initialising the map on the device failed with status %d (%s)", rc, core->last_error().c_str());
    // this object owns the host map and says when its planes go away (forget_map above and in the destructor): the layers that are
    // downloaded may be written by the device straight into them (GROUNDGRID_HIP_PIN_LAYERS=0: through a staging copy instead)
    const char *pin = std::getenv("GROUNDGRID_HIP_PIN_LAYERS");
    core->set_pin_planes(!pin || std::atoi(pin) != 0);
}

// src/GroundGrid.cpp:83-147
std::shared_ptr<grid_map::GridMap> GroundGrid::update(const nav_msgs::OdometryConstPtr &inOdom)
{
    if (!mMap_ptr) {
        initGroundGrid(inOdom);
        return mMap_ptr;
    }
    grid_map::GridMap &map = *mMap_ptr;
    Core *core = Registry::instance().core_of_map(mMap_ptr.get());
    if (!core) { // (see initGroundGrid: the segmentation's context did not exist then)
        core = Registry::instance().bind_map(mMap_ptr.get());
        if (!core) return mMap_ptr;
        map.setPosition(grid_map::Position(inOdom->pose.pose.position.x, inOdom->pose.pose.position.y));
        const int rc = core->reset_map(inOdom->pose.pose.position.x, inOdom->pose.pose.position.y, static_cast<float>(inOdom->pose.pose.position.z));
        if (rc != GG_OK) ROS_ERROR("groundgrid_hip: initialising the map on the device failed with status %d (%s)", rc, core->last_error().c_str());
        mLastPose.pose = inOdom->pose;
        mLastPose.header = inOdom->header;
        return mMap_ptr;
    }

    // static so if the new transform is not yet available, we can use the last one (as the reference, :100)
    static geometry_msgs::TransformStamped base_to_map;
    try {
        base_to_map = mTfBuffer.lookupTransform("base_link", "map", inOdom->header.stamp);
    } catch (tf2::LookupException &e) {
        ROS_WARN("no transform? -> error: %s", e.what()); // potentially degraded performance
    } catch (tf2::ExtrapolationException &e) {
        ROS_DEBUG("need to extrapolate a transform? -> error: %s", e.what()); // the old one is used instead
    }

    // What :121-131 needs of that transform: z of a map point in base_link = third row of the rotation . p + t.z.  The rotation
    // is the one tf2_geometry_msgs' doTransform(PointStamped) builds from the quaternion (KDL::Rotation::Quaternion on Noetic;
    // GROUNDGRID_HIP_ROTATION=tf2 selects tf2::Matrix3x3::setRotation -- tools/pin/ says which one a given installation uses).
    const char *conv = std::getenv("GROUNDGRID_HIP_ROTATION");
    const double q[4] = {base_to_map.transform.rotation.x, base_to_map.transform.rotation.y, base_to_map.transform.rotation.z, base_to_map.transform.rotation.w};
    double R[9];
    gg_rotation_from_quaternion(conv && !std::strcmp(conv, "tf2") ? GG_ROT_TF2 : GG_ROT_KDL, q, R);
    const double base_plane[4] = {R[6], R[7], R[8], base_to_map.transform.translation.z};

    bool moved = false;
    double snapped_x = 0.0, snapped_y = 0.0;
    const int rc = core->move_map(inOdom->pose.pose.position.x, inOdom->pose.pose.position.y, base_plane, &moved, &snapped_x, &snapped_y);
    if (rc != GG_OK) {
        ROS_ERROR("groundgrid_hip: moving the map on the device failed with status %d (%s)", rc, core->last_error().c_str());
        return mMap_ptr;
    }
    // We havent moved so we have nothing to do (:136-137)
    if (!moved) return mMap_ptr;

    map.setPosition(grid_map::Position(snapped_x, snapped_y)); // the host object follows: grid_map::move snaps to whole cells (:97)
    mLastPose.pose = inOdom->pose;
    mLastPose.header = inOdom->header;
    return mMap_ptr;
}

} // namespace groundgrid
