// GPU test of groundgrid_amd/host/binding_core.hpp -- everything the reference-typed bindings (ros/GroundSegmentationHip.cpp,
// ros/GroundGridHip.cpp) do below the ROS types -- against the C oracle, bit for bit:
//   * two "GroundSegmentation objects" (registry keys) in one process keep SEPARATE device contexts and maps, interleaved calls;
//   * object A's map is HOST-MANAGED: the test plays GroundGrid::update on the host (the oracle's own map is the host GridMap),
//     the binding uploads ground / groundpatch when the position changed and downloads all layers after every cloud;
//   * object B's map is DEVICE-RESIDENT: bound through the registry like GroundGridHip does, reset_map / move_map per frame, no
//     layer leaves the device during the drive; a cloud larger than the context's capacity re-creates the context in mid-drive
//     and the map survives; at the end all eleven layers are downloaded once and compared.
// Built and run by tests/test_cpp_adapter.py (g++; links libgroundgrid_hip.so and oracle/libgg_oracle.so).
//   exit 0 = identical, 1 = mismatch, 77 = no GPU (skipped)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "binding_core.hpp"
#include "gg_oracle.h"

using groundgrid_hip::Core;
using groundgrid_hip::MapView;
using groundgrid_hip::Registry;

static bool same_floats(const float *a, const float *b, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(a[i] == b[i] || (std::isnan(a[i]) && std::isnan(b[i])))) return false;
    return true;
}

static std::vector<gg_point32> make_cloud(size_t n, unsigned seed, float cx, float cy)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> uxy(-62.f, 62.f), u01(0.f, 1.f);
    std::normal_distribution<float> nz(0.f, 0.03f);
    std::vector<gg_point32> cloud(n);
    std::memset(cloud.data(), 0, n * sizeof(gg_point32));
    for (auto &p : cloud) {
        p.x = cx + uxy(rng);
        p.y = cy + uxy(rng);
        p.z = -1.7f + 0.015f * p.x + nz(rng) + (u01(rng) < 0.2f ? 2.f * u01(rng) : 0.f) - (u01(rng) < 0.04f ? 1.f : 0.f);
        p.intensity = u01(rng);
        p.ring = (uint16_t)(rng() % 64);
    }
    return cloud;
}

struct Drive { // one vehicle: pose per frame and the base_link <- map plane GroundGrid::update seeds exposed cells with
    double x, y;
    double plane[4];
};
static Drive pose_of(int frame, double speed, double side)
{
    Drive d;
    d.x = speed * frame;
    d.y = side * 0.4 * frame * frame;
    const double pitch = 0.01 * frame;
    d.plane[0] = -std::sin(pitch);
    d.plane[1] = 0.002 * frame;
    d.plane[2] = std::cos(pitch);
    d.plane[3] = 1.73 + 0.01 * frame;
    return d;
}

int main()
{
    gg_geometry g;
    gg_default_geometry(&g);
    const int key_a = 0, key_b = 0, map_b = 0; // (only their addresses matter: the registry's keys)
    Core *a = Registry::instance().core_of_object(&key_a, true);
    Core *b = Registry::instance().core_of_object(&key_b, true);
    if (!a->create(g, 60000) || !b->create(g, 20000)) { // (B starts too small on purpose)
        std::printf("no device: %s\n", a->last_error().c_str());
        return 77;
    }
    bool ok = a != b && a->context() != b->context();
    ok &= Registry::instance().core_of_object(&key_a, false) == a && Registry::instance().core_of_object(&key_b, false) == b;
    gg_config cfg;
    gg_default_config(&cfg);
    cfg.max_ring = 55;
    a->set_config(cfg);
    cfg.max_ring = 40;
    b->set_config(cfg);
    ggo_config rcfg_a, rcfg_b;
    ggo_default_config(&rcfg_a);
    ggo_default_config(&rcfg_b);
    rcfg_a.max_ring = 55;
    rcfg_b.max_ring = 40;

    // A: host-managed.  The oracle's map doubles as the host-side GridMap whose planes the binding reads and writes.
    ggo_map *host_a = ggo_map_create(120.0f, 0.33f, 0.0, 0.0, -0.2f), *ref_a = ggo_map_create(120.0f, 0.33f, 0.0, 0.0, -0.2f);
    // B: device-resident, bound like GroundGridHip::initGroundGrid does: the oldest Core without a map is A (construction order),
    // so the pairing gives A the first map and B the second
    const int map_a_dummy = 0;
    ok &= Registry::instance().bind_map(&map_a_dummy) == a;
    ok &= Registry::instance().bind_map(&map_b) == b && Registry::instance().core_of_map(&map_b) == b;
    ggo_map *ref_b = ggo_map_create(120.0f, 0.33f, 3.0, -2.0, 0.1f);
    ok &= b->reset_map(3.0, -2.0, 0.1f) == GG_OK && b->device_resident() && !a->device_resident();
    const size_t C = (size_t)ref_a->rows * ref_a->cols;

    for (int frame = 0; frame < 5 && ok; ++frame) {
        // ---- object A ----
        const Drive da = pose_of(frame, 1.1, 1.0);
        if (frame > 0) { // GroundGrid::update on the host, twice: the "host GridMap" and the reference
            int sh[2];
            ggo_map_update(host_a, da.x, da.y, da.plane, sh);
            ggo_map_update(ref_a, da.x, da.y, da.plane, sh);
        }
        const std::vector<gg_point32> cloud_a = make_cloud(40000, 100u + (unsigned)frame, (float)da.x, (float)da.y);
        MapView va;
        va.pos_x = host_a->position[0];
        va.pos_y = host_a->position[1];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) va.layer[l] = host_a->layer[l];
        const float org_a[3] = {(float)da.x, (float)da.y, 0.1f};
        std::vector<gg_point32> out(cloud_a.size()), rout(cloud_a.size());
        size_t n_out = 0;
        ok &= a->filter(va, groundgrid_hip::LAYERS_ALL, cloud_a.data(), cloud_a.size(), org_a, -1.8, out.data(), &n_out) == GG_OK;
        const size_t rn = ggo_filter_cloud(ref_a, &rcfg_a, reinterpret_cast<const ggo_point *>(cloud_a.data()), cloud_a.size(), org_a, -1.8,
                                           reinterpret_cast<ggo_point *>(rout.data()), nullptr, nullptr, nullptr, nullptr);
        ok &= n_out == rn && std::memcmp(out.data(), rout.data(), rn * sizeof(gg_point32)) == 0;
        for (int l = 0; l < GG_NUM_LAYERS; ++l) ok &= same_floats(host_a->layer[l], ref_a->layer[l], C); // downloaded into the host map
        std::printf("frame %d A (host-managed): %zu points returned -> %s\n", frame, n_out, ok ? "identical" : "MISMATCH");

        // ---- object B, interleaved ----
        const Drive db = pose_of(frame, -0.9, -1.0);
        if (frame > 0) {
            int sh[2], rsh[2];
            bool moved = false;
            double sx = 0, sy = 0;
            ok &= b->move_map(3.0 + db.x, -2.0 + db.y, db.plane, &moved, &sx, &sy) == GG_OK;
            const int rmoved = ggo_map_update(ref_b, 3.0 + db.x, -2.0 + db.y, db.plane, rsh);
            (void)sh;
            ok &= moved == (rmoved != 0) && sx == ref_b->position[0] && sy == ref_b->position[1];
        }
        // frame 3: a cloud three times the context's capacity -- the context is re-created, the device-resident map must survive
        const std::vector<gg_point32> cloud_b = make_cloud(frame == 3 ? 70000 : 18000, 200u + (unsigned)frame, 3.0f + (float)db.x, -2.0f + (float)db.y);
        MapView vb; // (no planes wanted during the drive)
        vb.pos_x = ref_b->position[0];
        vb.pos_y = ref_b->position[1];
        const float org_b[3] = {3.0f + (float)db.x, -2.0f + (float)db.y, 0.3f};
        out.assign(cloud_b.size(), gg_point32());
        rout.assign(cloud_b.size(), gg_point32());
        ok &= b->filter(vb, groundgrid_hip::LAYERS_NONE, cloud_b.data(), cloud_b.size(), org_b, -1.7, out.data(), &n_out) == GG_OK;
        const size_t rnb = ggo_filter_cloud(ref_b, &rcfg_b, reinterpret_cast<const ggo_point *>(cloud_b.data()), cloud_b.size(), org_b, -1.7,
                                            reinterpret_cast<ggo_point *>(rout.data()), nullptr, nullptr, nullptr, nullptr);
        ok &= n_out == rnb && std::memcmp(out.data(), rout.data(), rnb * sizeof(gg_point32)) == 0;
        ok &= b->device_resident() && (frame < 3 || b->capacity() >= 70000);
        std::printf("frame %d B (device-resident, capacity %zu): %zu points returned -> %s\n", frame, b->capacity(), n_out, ok ? "identical" : "MISMATCH");
    }
    // B's layers, once, after the drive: what a late subscriber would get
    {
        std::vector<std::vector<float>> planes(GG_NUM_LAYERS, std::vector<float>(C));
        float *dst[GG_NUM_LAYERS];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) dst[l] = planes[(size_t)l].data();
        ok &= gg_get_layers(b->context(), 0, dst) == GG_OK;
        for (int l = 0; l < GG_NUM_LAYERS; ++l) {
            const bool same = same_floats(planes[(size_t)l].data(), ref_b->layer[l], C);
            if (!same) std::printf("B layer %s differs\n", groundgrid_hip::layer_names()[l]);
            ok &= same;
        }
    }
    // ---- the stage members of the class on their own (include/groundgrid/GroundSegmentation.h:59-62) ----
    {
        // A, host-managed: the host map's planes are uploaded, the stage runs on the device, what it wrote comes back into the planes
        MapView va;
        va.pos_x = host_a->position[0];
        va.pos_y = host_a->position[1];
        for (int l = 0; l < GG_NUM_LAYERS; ++l) va.layer[l] = host_a->layer[l];
        // (the host may have edited its map since the last cloud: give two cells counts and heights no cloud produced)
        for (ggo_map *m : {host_a, ref_a}) {
            m->layer[GGO_POINTS][200 + 180 * (size_t)m->rows] = 25.f;
            m->layer[GGO_MINGROUNDHEIGHT][200 + 180 * (size_t)m->rows] = -1.9f;
            m->layer[GGO_M2][200 + 180 * (size_t)m->rows] = 1e-4f;
            m->layer[GGO_GROUNDPATCH][150 + 150 * (size_t)m->rows] = 0.95f;
        }
        ok &= a->run_stage(va, GG_STAGE_DETECT_GROUND_PATCHES, 1, 0, 0, 0.0) == GG_OK;
        ggo_stage_detect_section(ref_a, &rcfg_a, 1);
        ok &= a->run_stage(va, GG_STAGE_DETECT_GROUND_PATCH_5, 0, 200, 180, 0.0) == GG_OK;
        ggo_detect_ground_patch(ref_a, &rcfg_a, 5, 200, 180);
        ok &= a->run_stage(va, GG_STAGE_INTERPOLATE_CELL, 0, 150, 151, 0.0) == GG_OK;
        ggo_interpolate_cell(ref_a, &rcfg_a, 150, 151);
        ok &= a->run_stage(va, GG_STAGE_SPIRAL_GROUND_INTERPOLATION, 0, 0, 0, -1.6) == GG_OK;
        ggo_stage_spiral(ref_a, &rcfg_a, -1.6);
        for (int l : {(int)GG_LAYER_GROUND, (int)GG_LAYER_GROUNDPATCH, (int)GG_LAYER_VARIANCE, (int)GG_LAYER_POINTS}) ok &= same_floats(host_a->layer[l], ref_a->layer[l], C);
        std::printf("stages on the host-managed map -> %s\n", ok ? "identical" : "MISMATCH");
        // B, device-resident: nothing is uploaded; the view's planes only receive what the stage wrote
        std::vector<float> gb(C), pb(C), vbv(C);
        MapView vb;
        vb.pos_x = ref_b->position[0];
        vb.pos_y = ref_b->position[1];
        vb.layer[GG_LAYER_GROUND] = gb.data();
        vb.layer[GG_LAYER_GROUNDPATCH] = pb.data();
        vb.layer[GG_LAYER_VARIANCE] = vbv.data();
        ok &= b->run_stage(vb, GG_STAGE_DETECT_GROUND_PATCHES, 2, 0, 0, 0.0) == GG_OK;
        ggo_stage_detect_section(ref_b, &rcfg_b, 2);
        ok &= b->run_stage(vb, GG_STAGE_SPIRAL_GROUND_INTERPOLATION, 0, 0, 0, 0.4) == GG_OK;
        ggo_stage_spiral(ref_b, &rcfg_b, 0.4);
        ok &= same_floats(gb.data(), ref_b->layer[GGO_GROUND], C) && same_floats(pb.data(), ref_b->layer[GGO_GROUNDPATCH], C) && same_floats(vbv.data(), ref_b->layer[GGO_VARIANCE], C);
        ok &= b->run_stage(vb, GG_STAGE_DETECT_GROUND_PATCH_3, 0, 0, 7, 0.0) == GG_ERR_INVALID; // (the block would leave the map: UB in the reference)
        std::printf("stages on the device-resident map -> %s\n", ok ? "identical" : "MISMATCH");
    }
    // ---- lifetime: a map that goes away frees its Core for the next map; a re-initialised object drops its maps ----
    {
        const int map_a2 = 0;
        ok &= Registry::instance().bind_map(&map_a2) == nullptr; // (both Cores serve a map)
        Registry::instance().forget_map(&map_a_dummy);           // GroundGrid::~GroundGrid / initGroundGrid of a new map
        ok &= Registry::instance().core_of_map(&map_a_dummy) == nullptr && Registry::instance().bind_map(&map_a2) == a;
        Registry::instance().unbind_maps_of(a);                   // GroundSegmentation::init on an object that already had a context
        ok &= Registry::instance().core_of_map(&map_a2) == nullptr && Registry::instance().core_of_map(&map_b) == b;
    }
    // the layer selection of GROUNDGRID_HIP_LAYERS
    setenv("GROUNDGRID_HIP_LAYERS", "ground,variance", 1);
    ok &= groundgrid_hip::layers_from_env(0u) == ((1u << GG_LAYER_GROUND) | (1u << GG_LAYER_VARIANCE));
    setenv("GROUNDGRID_HIP_LAYERS", "state", 1);
    ok &= groundgrid_hip::layers_from_env(0u) == groundgrid_hip::LAYERS_STATE;
    unsetenv("GROUNDGRID_HIP_LAYERS");
    ok &= groundgrid_hip::layers_from_env(groundgrid_hip::LAYERS_ALL) == groundgrid_hip::LAYERS_ALL;
    Registry::instance().forget_object(&key_b);
    ok &= Registry::instance().core_of_map(&map_b) == nullptr && Registry::instance().core_of_object(&key_b, false) == nullptr;
    ggo_map_destroy(host_a);
    ggo_map_destroy(ref_a);
    ggo_map_destroy(ref_b);
    std::printf("%s\n", ok ? "binding core OK" : "binding core MISMATCH");
    return ok ? 0 : 1;
}
