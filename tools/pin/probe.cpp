// probe.cpp -- pinning kit for the oracle's third-party conventions (run on a ROS box, NOT in the build image).
//
// dcmlr/groundgrid inherits arithmetic from libraries whose versions it does not pin (package.xml:29, CMakeLists.txt:41):
// Eigen (order of fixed-size block sums), grid_map_core (index rounding, move / re-linearisation), tf2_geometry_msgs
// (which rotation matrix doTransform(PointStamped) builds), glibc (hypotf, atanf).  oracle/gg_oracle.c restates them from
// their published sources; this program links the REAL libraries and prints what they compute on seeded inputs, as one JSON
// document.  tools/pin/compare.py regenerates the same inputs, evaluates every variant the oracle offers and reports
// which one reproduces the vectors bit for bit -- that pins the oracle (and therefore the HIP path, which is
// bit-identical to it) to the reference's actual build environment.
//
// Each section cites the reference call site it stands for.  Nothing here is copied from the reference: the calls are
// the library entry points the reference uses, on inputs of our own.
//
// Build (ROS Noetic, Ubuntu 20.04):
//   source /opt/ros/noetic/setup.bash
//   g++ -O2 -std=c++14 probe.cpp -o probe $(pkg-config --cflags eigen3) -I/opt/ros/noetic/include \
//       -L/opt/ros/noetic/lib -lgrid_map_core -ltf2 -lorocos-kdl -lrostime -lcpp_common -lroscpp_serialization \
//       -Wl,-rpath,/opt/ros/noetic/lib
//   ./probe > vectors.json && python3 compare.py vectors.json
// Use the SAME optimisation / -march flags as the groundgrid build under test (catkin Release: -O2, no -march=native):
// Eigen's packet type, and with it the 5x5 reduction order, depends on them.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include <Eigen/Dense>
#include <geometry_msgs/PointStamped.h>
#include <geometry_msgs/TransformStamped.h>
#include <grid_map_core/grid_map_core.hpp>
#include <tf2_geometry_msgs/tf2_geometry_msgs.h>

#ifdef PIN_WITH_GRID_MAP_ROS // optional section 6 (the serialised grid_map_msgs/GridMap): add -DPIN_WITH_GRID_MAP_ROS -lgrid_map_ros to the build line
#include <grid_map_ros/grid_map_ros.hpp>
#include <ros/serialization.h>
#endif

#include "pin_inputs.h" // Lcg: the seeded input generator shared with compare.py (mirrored there in Python)

static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static uint64_t bits(double d) { uint64_t u; std::memcpy(&u, &d, 8); return u; }
static bool first_item;
static void sep() { if (!first_item) std::printf(","); first_item = false; }
static void open_array(const char *name) { std::printf(",\n\"%s\": [", name); first_item = true; }
static void close_array() { std::printf("]"); }
static void put(float f) { sep(); std::printf("\"%08x\"", bits(f)); }
static void put(double d) { sep(); std::printf("\"%016llx\"", (unsigned long long)bits(d)); }
static void put(int v) { sep(); std::printf("%d", v); }

// src/GroundSegmentation.cpp:355-375, :453-458: the exact expression shapes the reference uses on Block<MatrixXf,S,S>
template <int S> static void eigen_blocks(const Eigen::MatrixXf &A, const Eigen::MatrixXf &B, const char *tag)
{
    char name[64];
    std::snprintf(name, sizeof name, "eigen_sum%d", S);
    open_array(name);
    for (int j = 0; j + S <= A.cols(); j += 3)
        for (int i = 0; i + S <= A.rows(); i += 3) {
            const auto &blk = A.block<S, S>(i, j);
            put((float)blk.sum()); // :359 pointsBlock.sum(), :457 gvlblock.sum()
        }
    close_array();
    std::snprintf(name, sizeof name, "eigen_prod%d", S);
    open_array(name);
    for (int j = 0; j + S <= A.cols(); j += 3)
        for (int i = 0; i + S <= A.rows(); i += 3) {
            const auto &a = A.block<S, S>(i, j);
            const auto &b = B.block<S, S>(i, j);
            put((float)a.cwiseProduct(b).sum()); // :375, :458
        }
    close_array();
    std::snprintf(name, sizeof name, "eigen_arrprod%d", S);
    open_array(name);
    for (int j = 0; j + S <= A.cols(); j += 3)
        for (int i = 0; i + S <= A.rows(); i += 3) {
            const auto &a = A.block<S, S>(i, j);
            const auto &b = B.block<S, S>(i, j);
            put((float)a.array().cwiseProduct(b.array()).sum()); // :374
        }
    close_array();
    std::snprintf(name, sizeof name, "eigen_min%d", S);
    open_array(name);
    for (int j = 0; j + S <= A.cols(); j += 3)
        for (int i = 0; i + S <= A.rows(); i += 3) put((float)A.block<S, S>(i, j).minCoeff()); // :373
    close_array();
    (void)tag;
}

int main()
{
    std::printf("{\n\"format\": 1");
    std::printf(",\n\"eigen_version\": [%d, %d, %d]", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
    std::printf(",\n\"eigen_packet_floats\": %d", (int)Eigen::internal::packet_traits<float>::size);
#ifdef EIGEN_VECTORIZE_AVX
    std::printf(",\n\"eigen_avx\": 1");
#else
    std::printf(",\n\"eigen_avx\": 0");
#endif

    // ---- 1. Eigen fixed-size block reductions -------------------------------------------------------------------
    {
        const int n = 41;
        Eigen::MatrixXf A(n, n), B(n, n);
        Lcg g(0xE16E0001u);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                A(i, j) = g.wide_float();
                B(i, j) = g.wide_float();
            }
        eigen_blocks<3>(A, B, "3");
        eigen_blocks<5>(A, B, "5");
        // :323 whole-layer element-wise quotient (no reduction, listed for completeness)
        open_array("eigen_variance");
        Eigen::MatrixXf P = A.cwiseAbs(), V;
        V = B.array() / (P.array() + std::numeric_limits<float>::min());
        for (int k = 0; k < 64; ++k) put(V(k % n, k / n));
        close_array();
    }

    // ---- 2. grid_map_core: setGeometry / getIndex / isInside (src/GroundGrid.cpp:58, src/GroundSegmentation.cpp:228-230,261)
    {
        const float mDimension = 120.0f, mResolution = .33f; // include/groundgrid/GroundGrid.h:70-71
        grid_map::GridMap map({"ground", "groundpatch"});
        const double px = 12.34, py = -7.77;
        map.setGeometry(grid_map::Length(mDimension, mDimension), mResolution, grid_map::Position(px, py));
        std::printf(",\n\"gm_size\": [%d, %d]", map.getSize()(0), map.getSize()(1));
        open_array("gm_geometry");
        put(map.getResolution());
        put(map.getLength().x());
        put(map.getLength().y());
        put(map.getPosition().x());
        put(map.getPosition().y());
        close_array();
        open_array("gm_index"); // per probe: isInside, getIndex return value, row, col
        const double res = map.getResolution(), half = 0.5 * map.getLength().x();
        Lcg g(0x61D00002u);
        for (int k = -2; k <= map.getSize()(0) + 2; ++k)
            for (int v = 0; v < 3; ++v) { // exactly on a cell edge, one ulp either side
                double x = (px + half) - (double)k * res;
                if (v == 1) x = std::nextafter(x, 1e300);
                if (v == 2) x = std::nextafter(x, -1e300);
                const double y = (py + half) - ((double)(k * 7 % 364) + g.unit()) * res;
                grid_map::Index idx(-7, -7);
                const grid_map::Position pos(x, y);
                const bool inside = map.isInside(pos);
                const bool ok = map.getIndex(pos, idx);
                put((int)inside);
                put((int)ok);
                put((int)idx(0));
                put((int)idx(1));
                // and the transposed probe (edge in y)
                grid_map::Index idy(-7, -7);
                const grid_map::Position pos2(y - py + px, x - px + py);
                put((int)map.isInside(pos2));
                put((int)map.getIndex(pos2, idy));
                put((int)idy(0));
                put((int)idy(1));
            }
        close_array();
        open_array("gm_index_random");
        for (int k = 0; k < 4000; ++k) {
            const double rx = px + (g.unit() - 0.5) * 130.0; // (two statements: argument evaluation order is unspecified)
            const double ry = py + (g.unit() - 0.5) * 130.0;
            const grid_map::Position pos(rx, ry);
            grid_map::Index idx(-7, -7);
            put((int)map.isInside(pos));
            map.getIndex(pos, idx);
            put((int)idx(0));
            put((int)idx(1));
        }
        close_array();
    }

    // ---- 3. GroundGrid::update as the reference runs it (src/GroundGrid.cpp:83-147): move + exposed-cell fill through
    //         tf2::doTransform(PointStamped) + convertToDefaultStartIndex, on a 64 x 64 map -------------------------------
    {
        const double moves[][2] = {{0.16, -0.16}, {0.17, 0.0}, {0.7, -0.34}, {-3.0, 5.2}, {-3.0, 5.2}, {2.475, 5.2}, {40.0, 5.2}, {40.0, -90.0}};
        const double poses[][7] = {{0, 0, 1, 0, 0, 0, 1},
                                   {0.3, 0.2, 1.5, 0.02, -0.01, 0.3, 0.9533},
                                   {0.0, 0.0, 1.25, 0, 0, 0, 1},
                                   {1, 2, 3, 0, 0, 0.70710678118654757, 0.70710678118654757},
                                   {1, 2, 3, 0, 0, 0.70710678118654757, 0.70710678118654757},
                                   {-4.0, 1.0, 0.5, 0.0499791692706783, 0.0, 0.0, 0.9987502603949663},
                                   {0, 0, 0.5, 0, 0, 0, 1},
                                   {0, 0, 0.5, 0.1, 0.2, 0.3, 0.9273618495495703}};
        grid_map::GridMap map({"ground", "groundpatch"});
        map.setGeometry(grid_map::Length(21.12f, 21.12f), .33f, grid_map::Position(0.0, 0.0));
        const int n = map.getSize()(0);
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) {
                map["ground"](i, j) = (float)(i + j * n) * 0.125f;
                map["groundpatch"](i, j) = 0.5f + (float)((i * 31 + j * 17) % 64) / 256.0f;
            }
        std::printf(",\n\"update\": [");
        for (size_t m = 0; m < sizeof moves / sizeof moves[0]; ++m) {
            std::vector<grid_map::BufferRegion> damage;
            const bool moved = map.move(grid_map::Position(moves[m][0], moves[m][1]), damage);
            geometry_msgs::TransformStamped base_to_map;
            base_to_map.transform.translation.x = poses[m][0];
            base_to_map.transform.translation.y = poses[m][1];
            base_to_map.transform.translation.z = poses[m][2];
            base_to_map.transform.rotation.x = poses[m][3];
            base_to_map.transform.rotation.y = poses[m][4];
            base_to_map.transform.rotation.z = poses[m][5];
            base_to_map.transform.rotation.w = poses[m][6];
            geometry_msgs::PointStamped ps;
            grid_map::Position pos;
            for (auto region : damage)
                for (auto it = grid_map::SubmapIterator(map, region); !it.isPastEnd(); ++it) {
                    auto idx = *it;
                    map.getPosition(idx, pos);
                    ps.point.x = pos(0);
                    ps.point.y = pos(1);
                    ps.point.z = 0;
                    tf2::doTransform(ps, ps, base_to_map);
                    map.at("ground", idx) = -ps.point.z;
                    map.at("groundpatch", idx) = 0.0;
                }
            if (!damage.empty()) map.convertToDefaultStartIndex();
            std::printf("%s\n {\"moved\": %d, \"damage\": %d, \"position\": [\"%016llx\", \"%016llx\"], \"ground\": [", m ? "," : "", (int)moved,
                        (int)damage.size(), (unsigned long long)bits(map.getPosition().x()), (unsigned long long)bits(map.getPosition().y()));
            for (int k = 0; k < n * n; ++k) std::printf("%s\"%08x\"", k ? "," : "", bits(map["ground"](k % n, k / n)));
            std::printf("], \"groundpatch\": [");
            for (int k = 0; k < n * n; ++k) std::printf("%s\"%08x\"", k ? "," : "", bits(map["groundpatch"](k % n, k / n)));
            std::printf("]}");
        }
        std::printf("]");
    }

    // ---- 4. tf2::doTransform(PointStamped): the per-point cloud transform and the cloud origin
    //         (src/GroundGridNodelet.cpp:146,176) ---------------------------------------------------------------------------
    {
        open_array("do_transform"); // per case: x, y, z (doubles)
        Lcg g(0x7F200004u);
        for (int k = 0; k < 600; ++k) {
            double q[4] = {g.unit() - 0.5, g.unit() - 0.5, g.unit() - 0.5, g.unit() - 0.5};
            if (k % 4) { // unit quaternions, normalised the way a publisher would (double sqrt); every 4th left unnormalised
                const double nrm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
                for (double &v : q) v /= nrm;
            }
            geometry_msgs::TransformStamped t;
            t.transform.translation.x = (g.unit() - 0.5) * 2000.0;
            t.transform.translation.y = (g.unit() - 0.5) * 2000.0;
            t.transform.translation.z = (g.unit() - 0.5) * 20.0;
            t.transform.rotation.x = q[0];
            t.transform.rotation.y = q[1];
            t.transform.rotation.z = q[2];
            t.transform.rotation.w = q[3];
            geometry_msgs::PointStamped p;
            p.point.x = (double)(float)((g.unit() - 0.5) * 160.0); // cloud points are floats widened to double (:172-174)
            p.point.y = (double)(float)((g.unit() - 0.5) * 160.0);
            p.point.z = (double)(float)((g.unit() - 0.5) * 10.0);
            tf2::doTransform(p, p, t);
            put(p.point.x);
            put(p.point.y);
            put(p.point.z);
        }
        close_array();
    }

    // ---- 5. libm: std::hypot(float, float) (src/GroundSegmentation.cpp:170), std::atan(float) (:44) --------------------
    {
        Lcg g(0x11B30005u);
        open_array("hypotf");
        for (int k = 0; k < 4000; ++k) {
            const float x = g.wide_float() * 0.01f, y = g.wide_float() * 0.01f;
            put((float)std::hypot(x, y));
        }
        close_array();
        open_array("expected_points"); // :40-46 for a 364-cell map, sampled
        const size_t cellCount = 364;
        for (size_t i = 0; i < cellCount; i += 7)
            for (size_t j = 0; j < cellCount; j += 11) {
                const float dist = std::hypot(i - cellCount / 2.0, j - cellCount / 2.0);
                put((float)(std::atan(1 / dist) / (float)(0.00174532925 * 2)));
            }
        close_array();
    }
    #ifdef PIN_WITH_GRID_MAP_ROS
    // ---- 6. the map as the nodelet publishes it: grid_map::GridMapRosConverter::toMessage + ROS 1 serialisation
    //         (src/GroundGridNodelet.cpp:211-214); libgroundgrid_hip's gg_get_gridmap_message writes these bytes itself
    {
        grid_map::GridMap map({"points", "ground"});
        map.setFrameId("map");
        map.setGeometry(grid_map::Length(1.65, 1.32), 0.33, grid_map::Position(0.5, -0.25)); // 5 x 4 cells
        Lcg g(0x6D5A0006u);
        for (const char *layer : {"points", "ground"})
            for (int j = 0; j < map.getSize()(1); ++j)
                for (int i = 0; i < map.getSize()(0); ++i) map[layer](i, j) = g.wide_float();
        grid_map_msgs::GridMap msg;
        grid_map::GridMapRosConverter::toMessage(map, msg);
        msg.info.header.stamp = ros::Time(1234, 5678); // :213
        const uint32_t len = ros::serialization::serializationLength(msg);
        std::vector<uint8_t> buf(len);
        ros::serialization::OStream stream(buf.data(), len);
        ros::serialization::serialize(stream, msg);
        open_array("gridmap_msg");
        for (uint8_t b : buf) put((int)b);
        close_array();
    }
#endif
    std::printf("\n}\n");
    return 0;
}
