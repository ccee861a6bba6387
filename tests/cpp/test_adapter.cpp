// C++ parity test of the host adapter (groundgrid_amd/host/GroundSegmentation.hpp) against the C oracle.
// Reads like a test of the reference class: init, setConfig, filter_cloud over a few frames, compare everything.
// Built and run by tests/test_cpp_adapter.py (g++; links libgroundgrid_hip.so and oracle/libgg_oracle.so).
//   exit 0 = bit-identical, 1 = mismatch, 77 = no GPU (skipped)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "GroundSegmentation.hpp"
#include "gg_oracle.h"

static bool same_floats(const float *a, const float *b, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(a[i] == b[i] || (std::isnan(a[i]) && std::isnan(b[i])))) return false;
    return true;
}

int main()
{
    groundgrid_hip::GroundSegmentation seg;
    try {
        seg.init(120, 0.33f, 1, 50000);
    } catch (const std::exception &e) {
        std::printf("no device: %s\n", e.what());
        return 77;
    }
    gg_config cfg;
    gg_default_config(&cfg);
    cfg.max_ring = 50;
    seg.setConfig(cfg);

    ggo_map *ref = ggo_map_create(120.0f, 0.33f, 2.0, -1.0, -0.5f);
    ggo_config rcfg;
    ggo_default_config(&rcfg);
    rcfg.max_ring = 50;
    seg.map().reset(2.0, -1.0, -0.5f);

    std::mt19937 rng(42);
    std::uniform_real_distribution<float> uxy(-65.f, 65.f), u01(0.f, 1.f);
    std::normal_distribution<float> nz(0.f, 0.03f);
    std::vector<gg_point32> cloud(40000);
    std::memset(cloud.data(), 0, cloud.size() * sizeof(gg_point32));
    for (auto &p : cloud) {
        p.x = uxy(rng);
        p.y = uxy(rng);
        p.z = -1.7f + 0.02f * p.x + nz(rng) + (u01(rng) < 0.2f ? 2.f * u01(rng) : 0.f) - (u01(rng) < 0.03f ? 1.f : 0.f);
        p.intensity = u01(rng);
        p.ring = (uint16_t)(rng() % 64);
    }
    const gg_point32 origin = {1.5f, -0.5f, 0.2f, 0.f, 0.f, 0, 0, {0, 0}};
    const float org[3] = {origin.x, origin.y, origin.z};

    bool ok = true;
    for (int frame = 0; frame < 3 && ok; ++frame) {
        std::vector<gg_point32> out = seg.filter_cloud(cloud, origin, -1.9, seg.map());
        std::vector<gg_point32> rout(cloud.size());
        std::vector<uint8_t> rlabel(cloud.size()), rcls(cloud.size());
        std::vector<int32_t> rindex(cloud.size()), rcell(cloud.size());
        const size_t rn = ggo_filter_cloud(ref, &rcfg, reinterpret_cast<const ggo_point *>(cloud.data()), cloud.size(), org, -1.9,
                                           reinterpret_cast<ggo_point *>(rout.data()), rlabel.data(), rindex.data(), rcls.data(), rcell.data());
        ok &= out.size() == rn && std::memcmp(out.data(), rout.data(), rn * sizeof(gg_point32)) == 0;
        ok &= std::memcmp(seg.labels().data(), rlabel.data(), cloud.size()) == 0;
        ok &= std::memcmp(seg.out_index().data(), rindex.data(), cloud.size() * 4) == 0;
        const std::vector<std::vector<float>> all = seg.map().layers(); // one synchronisation for the eleven (gg_get_layers)
        for (int l = 0; l < GG_NUM_LAYERS && ok; ++l) {
            const std::vector<float> a = seg.map().layer((gg_layer)l);
            ok &= same_floats(a.data(), ref->layer[l], a.size());
            ok &= same_floats(all[(size_t)l].data(), ref->layer[l], a.size());
        }
        std::vector<std::pair<size_t, groundgrid_hip::GroundSegmentation::Index>> pi, ig;
        std::vector<size_t> outl;
        seg.last_call_lists(0, cloud.size(), pi, ig, outl, seg.map()); // (the decisions of the call above; nothing is inserted)
        size_t nk = 0, ni = 0, no = 0;
        for (size_t i = 0; i < cloud.size(); ++i) {
            nk += rcls[i] == GGO_KEPT;
            ni += rcls[i] == GGO_IGNORED;
            no += rcls[i] == GGO_OUTLIER;
        }
        ok &= pi.size() == nk && ig.size() == ni && outl.size() == no;
        std::printf("frame %d: %zu points returned, kept %zu ignored %zu outliers %zu -> %s\n", frame, out.size(), nk, ni, no, ok ? "identical" : "MISMATCH");
    }
    // GroundSegmentation::insert_cloud as the member it is: two ranges of the cloud INTO the map as the three frames left it (the second range
    // continues the counts of the first), lists and all eleven layers against the oracle's insert_cloud on the same state
    {
        const size_t half = cloud.size() / 2;
        std::vector<std::pair<size_t, groundgrid_hip::GroundSegmentation::Index>> pi, ig;
        std::vector<size_t> outl;
        seg.insert_cloud(cloud, 0, half, origin, pi, ig, outl, seg.map());
        seg.insert_cloud(cloud, half, cloud.size(), origin, pi, ig, outl, seg.map());
        std::vector<uint8_t> rcls(cloud.size());
        std::vector<int32_t> rcell(cloud.size());
        ggo_stage_insert(ref, &rcfg, reinterpret_cast<const ggo_point *>(cloud.data()), half, org, rcls.data(), rcell.data());
        ggo_stage_insert(ref, &rcfg, reinterpret_cast<const ggo_point *>(cloud.data()) + half, cloud.size() - half, org, rcls.data() + half, rcell.data() + half);
        size_t k = 0, g = 0, o = 0;
        for (size_t i = 0; i < cloud.size() && ok; ++i) {
            if (rcls[i] == GGO_KEPT) ok &= k < pi.size() && pi[k].first == i && pi[k].second[0] + pi[k].second[1] * seg.map().rows() == rcell[i], ++k;
            else if (rcls[i] == GGO_IGNORED) ok &= g < ig.size() && ig[g].first == i && ig[g].second[0] + ig[g].second[1] * seg.map().rows() == rcell[i], ++g;
            else if (rcls[i] == GGO_OUTLIER) ok &= o < outl.size() && outl[o] == i, ++o;
        }
        ok &= k == pi.size() && g == ig.size() && o == outl.size();
        for (int l = 0; l < GG_NUM_LAYERS && ok; ++l) {
            const std::vector<float> a = seg.map().layer((gg_layer)l);
            ok &= same_floats(a.data(), ref->layer[l], a.size());
        }
        std::printf("insert_cloud in two ranges: kept %zu ignored %zu outliers %zu -> %s\n", k, g, o, ok ? "identical" : "MISMATCH");
    }
    ggo_map_destroy(ref);
    return ok ? 0 : 1;
}
