/* Cross-stream ordering through the C ABI alone (no Python, no torch): a batch launched on a FOREIGN stream followed at
 * once by gg_get_layer / gg_set_layer / gg_reset_map / gg_move_map on the context's own stream must behave like the
 * serial call sequence ("a C caller reads stale layers").  The foreign stream is kept busy with a long fill so
 * that a missing dependency shows as stale data instead of passing by luck.  Also drives the pipelined host call
 * (gg_filter_cloud_async / gg_filter_cloud_wait) two clouds deep and compares every result with the C oracle; and a batch divided
 * into two concurrent halves (GG_FLAG_CONCURRENT_HALVES) behind a busy queue, with a getter and a fenced copy behind it.
 *   exit 0 = bit-identical, 1 = mismatch, 77 = no GPU (skipped).  Built and run by tests/test_cpp_adapter.py. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "gg_oracle.h"
#include "groundgrid_hip.h"

#define CHECK(call)                                                                 \
    do {                                                                            \
        int rc__ = (call);                                                          \
        if (rc__ != 0) {                                                            \
            printf("%s failed: %d (%s)\n", #call, rc__, ctx ? gg_last_error(ctx) : ""); \
            return 1;                                                               \
        }                                                                           \
    } while (0)

static unsigned lcg(unsigned *s) { return *s = *s * 1664525u + 1013904223u; }
static float u01(unsigned *s) { return (float)(lcg(s) >> 8) / 16777216.0f; }

static int same_floats(const float *a, const float *b, size_t n, const char *what)
{
    for (size_t i = 0; i < n; ++i)
        if (!(a[i] == b[i] || (isnan(a[i]) && isnan(b[i])))) {
            printf("MISMATCH %s at %zu: %g vs %g\n", what, i, a[i], b[i]);
            return 0;
        }
    return 1;
}

static void make_cloud(gg_point32 *c, size_t n, unsigned seed, float shift)
{
    memset(c, 0, n * sizeof *c);
    for (size_t i = 0; i < n; ++i) {
        c[i].x = 110.0f * u01(&seed) - 55.0f + shift;
        c[i].y = 110.0f * u01(&seed) - 55.0f;
        c[i].z = -1.7f + 0.02f * c[i].x + 0.05f * (u01(&seed) - 0.5f) + (u01(&seed) < 0.2f ? 2.0f * u01(&seed) : 0.0f);
        c[i].ring = (uint16_t)(lcg(&seed) % 64);
    }
}

int main(void)
{
    gg_context *ctx = NULL;
    enum { N = 60000 };
    int rc = gg_create(NULL, 1, N, 0, &ctx);
    if (rc == GG_ERR_NO_DEVICE) {
        printf("no device\n");
        return 77;
    }
    if (rc != GG_OK) {
        printf("gg_create: %d\n", rc);
        return 1;
    }
    int rows = 0, cols = 0;
    gg_get_size(ctx, &rows, &cols);
    const size_t C = (size_t)rows * cols;

    ggo_map *ref = ggo_map_create(120.0f, 0.33f, 0.0, 0.0, 0.0f);
    ggo_config rcfg;
    ggo_default_config(&rcfg);

    gg_point32 *cloud = (gg_point32 *)malloc(N * sizeof *cloud), *cloud2 = (gg_point32 *)malloc(N * sizeof *cloud2);
    make_cloud(cloud, N, 7u, 0.0f);
    make_cloud(cloud2, N, 99u, 1.0f);
    float *layer = (float *)malloc(C * 4), *custom = (float *)malloc(C * 4);
    uint8_t *label = (uint8_t *)malloc(N), *rlabel = (uint8_t *)malloc(N);
    int32_t *index = (int32_t *)malloc(N * 4), *rindex = (int32_t *)malloc(N * 4);
    gg_point32 *out = (gg_point32 *)malloc(N * sizeof *out), *rout = (gg_point32 *)malloc(N * sizeof *rout);

    hipStream_t foreign;
    if (hipStreamCreateWithFlags(&foreign, hipStreamNonBlocking) != hipSuccess) return 1;
    gg_point32 *d_cloud;
    uint8_t *d_labels;
    char *d_busy;
    const size_t BUSY = (size_t)1 << 31;
    if (hipMalloc((void **)&d_cloud, N * sizeof *cloud) != hipSuccess || hipMalloc((void **)&d_labels, N) != hipSuccess ||
        hipMalloc((void **)&d_busy, BUSY) != hipSuccess)
        return 1;
    hipMemcpy(d_cloud, cloud, N * sizeof *cloud, hipMemcpyHostToDevice);

    const float org[3] = {0.5f, -0.25f, 0.1f};
    const double base_z = -1.73;
    const int32_t n32 = N;
    gg_batch b;
    memset(&b, 0, sizeof b);
    b.n_clouds = 1;
    b.first_slot = 0;
    b.point_format = GG_POINT32;
    b.d_points = d_cloud;
    b.cloud_stride = N;
    b.n_points = &n32;
    b.origins = org;
    b.base_z = &base_z;
    b.d_labels = d_labels;

    int ok = 1;
    /* 1. reset (context stream) -> batch (busy foreign stream) -> get_layer (context stream), no synchronisation by us */
    CHECK(gg_reset_map(ctx, 0, 0.0, 0.0, -0.25f));
    ggo_map_reset_state(ref, 0.0, 0.0, -0.25f);
    for (int k = 0; k < 12; ++k) hipMemsetAsync(d_busy, k, BUSY, foreign); /* ~10 ms of queue ahead of the batch */
    CHECK(gg_filter_batch(ctx, &b, foreign));
    CHECK(gg_get_layer(ctx, 0, GG_LAYER_GROUND, layer));
    ggo_filter_cloud(ref, &rcfg, (const ggo_point *)cloud, N, org, base_z, NULL, rlabel, NULL, NULL, NULL);
    ok &= same_floats(layer, ref->layer[GGO_GROUND], C, "ground after batch on a foreign stream");
    CHECK(gg_get_layer(ctx, 0, GG_LAYER_POINTS, layer));
    ok &= same_floats(layer, ref->layer[GGO_POINTS], C, "points after batch on a foreign stream");
    hipStreamSynchronize(foreign); /* d_labels is the caller's buffer: the caller orders its own reads */
    hipMemcpy(label, d_labels, N, hipMemcpyDeviceToHost);
    ok &= memcmp(label, rlabel, N) == 0;
    if (!ok) printf("step 1 failed\n");

    /* 2. set_layer (context stream) right behind a busy batch, then another batch: the host edit must land between them */
    for (size_t i = 0; i < C; ++i) custom[i] = -1.5f + 0.001f * (float)(i % 97);
    for (int k = 0; k < 12; ++k) hipMemsetAsync(d_busy, k, BUSY, foreign);
    CHECK(gg_filter_batch(ctx, &b, foreign));
    ggo_filter_cloud(ref, &rcfg, (const ggo_point *)cloud, N, org, base_z, NULL, NULL, NULL, NULL, NULL);
    CHECK(gg_set_layer(ctx, 0, GG_LAYER_GROUND, custom));
    memcpy(ref->layer[GGO_GROUND], custom, C * 4);
    CHECK(gg_filter_batch(ctx, &b, foreign));
    ggo_filter_cloud(ref, &rcfg, (const ggo_point *)cloud, N, org, base_z, NULL, NULL, NULL, NULL, NULL);
    CHECK(gg_get_layer(ctx, 0, GG_LAYER_GROUND, layer));
    ok &= same_floats(layer, ref->layer[GGO_GROUND], C, "ground after set_layer between two foreign-stream batches");
    CHECK(gg_get_layer(ctx, 0, GG_LAYER_GROUNDPATCH, layer));
    ok &= same_floats(layer, ref->layer[GGO_GROUNDPATCH], C, "groundpatch after set_layer between two foreign-stream batches");

    /* 3. move_map behind a busy batch, batch again */
    {
        const double pose[7] = {0.3, -0.1, 1.7, 0.01, -0.02, 0.05, 0.998};
        double m[12];
        CHECK(gg_transform_from_pose(GG_ROT_KDL, pose, m));
        const double plane[4] = {m[8], m[9], m[10], m[11]};
        int sh[2], rsh[2];
        for (int k = 0; k < 12; ++k) hipMemsetAsync(d_busy, k, BUSY, foreign);
        CHECK(gg_filter_batch(ctx, &b, foreign));
        ggo_filter_cloud(ref, &rcfg, (const ggo_point *)cloud, N, org, base_z, NULL, NULL, NULL, NULL, NULL);
        CHECK(gg_move_map(ctx, 0, 2.4, -1.1, plane, sh));
        ggo_map_update(ref, 2.4, -1.1, plane, rsh);
        ok &= sh[0] == rsh[0] && sh[1] == rsh[1];
        CHECK(gg_filter_batch(ctx, &b, foreign));
        ggo_filter_cloud(ref, &rcfg, (const ggo_point *)cloud, N, org, base_z, NULL, NULL, NULL, NULL, NULL);
        CHECK(gg_get_layer(ctx, 0, GG_LAYER_GROUND, layer));
        ok &= same_floats(layer, ref->layer[GGO_GROUND], C, "ground after move_map between two foreign-stream batches");
    }
    CHECK(gg_synchronize(ctx));

    /* 4. the pipelined host call, two clouds in flight, against the serial oracle */
    {
        int t[2];
        size_t out_n = 0;
        const gg_point32 *seq[6] = {cloud, cloud2, cloud, cloud2, cloud2, cloud};
        CHECK(gg_filter_cloud_async(ctx, 0, seq[0], N, NULL, org, base_z, &t[0]));
        for (int k = 0; k < 6; ++k) {
            if (k + 1 < 6) CHECK(gg_filter_cloud_async(ctx, 0, seq[k + 1], N, NULL, org, base_z, &t[(k + 1) & 1]));
            if (k == 0) { /* a third ticket does not fit */
                int t3;
                ok &= gg_filter_cloud_async(ctx, 0, seq[0], N, NULL, org, base_z, &t3) == GG_ERR_CAPACITY;
            }
            CHECK(gg_filter_cloud_wait(ctx, t[k & 1], out, &out_n, label, index));
            const size_t rn = ggo_filter_cloud(ref, &rcfg, (const ggo_point *)seq[k], N, org, base_z, (ggo_point *)rout, rlabel, rindex, NULL, NULL);
            ok &= rn == out_n && memcmp(label, rlabel, N) == 0 && memcmp(index, rindex, N * 4) == 0 &&
                  memcmp(out, rout, rn * sizeof *out) == 0;
            if (!ok) {
                printf("async frame %d differs (out_n %zu vs %zu)\n", k, out_n, rn);
                break;
            }
        }
        CHECK(gg_get_layer(ctx, 0, GG_LAYER_GROUND, layer));
        ok &= same_floats(layer, ref->layer[GGO_GROUND], C, "ground after the async sequence");
    }

    /* 5. GG_FLAG_CONCURRENT_HALVES: a batch of four clouds on four maps runs as two halves -- slots 0, 1 on the foreign stream, slots
     *    2, 3 on the library's side stream -- with NO join between calls.  Three steps back to back behind a busy queue (map
     *    re-initialisation on the foreign stream, divided the same way, in front of the second), then: a getter orders itself after
     *    both halves; the caller's own copy of the labels does so with gg_batch_fence. */
    {
        enum { B = 4, M = 20000 };
        extern int gg_debug_set_tuning(gg_context *, const char *, int);
        gg_context *ctx2 = NULL;
        if (gg_create(NULL, B, M, 0, &ctx2) != GG_OK) return 1;
        gg_context *const ctx_outer = ctx;
        ctx = ctx2; /* (CHECK reports against this context) */
        CHECK(gg_set_flags(ctx, GG_FLAG_CONCURRENT_HALVES));
        CHECK(gg_debug_set_tuning(ctx, "halves_min_clouds", 2));
        ggo_map *refs[B];
        gg_point32 *hc = (gg_point32 *)malloc((size_t)B * M * sizeof *hc), *dc = NULL;
        uint8_t *dl = NULL, *hl = (uint8_t *)malloc((size_t)B * M);
        for (int k = 0; k < B; ++k) {
            refs[k] = ggo_map_create(120.0f, 0.33f, 0.0, 0.0, 0.0f);
            make_cloud(hc + (size_t)k * M, M, 1000u + (unsigned)k, 0.5f * (float)k);
        }
        if (hipMalloc((void **)&dc, (size_t)B * M * sizeof *dc) != hipSuccess || hipMalloc((void **)&dl, (size_t)B * M) != hipSuccess) return 1;
        hipMemcpy(dc, hc, (size_t)B * M * sizeof *dc, hipMemcpyHostToDevice);
        int32_t np[B], slots[B];
        float orgs[B * 3];
        double bzs[B];
        for (int k = 0; k < B; ++k) np[k] = M, orgs[3 * k] = 0.1f * (float)k, orgs[3 * k + 1] = 0.0f, orgs[3 * k + 2] = 0.0f, bzs[k] = -1.73;
        gg_batch bb;
        memset(&bb, 0, sizeof bb);
        bb.n_clouds = B;
        bb.point_format = GG_POINT32;
        bb.d_points = dc;
        bb.cloud_stride = M;
        bb.n_points = np;
        bb.origins = orgs;
        bb.base_z = bzs;
        bb.d_labels = dl;
        bb.slots = slots;
        for (int k = 0; k < 12; ++k) hipMemsetAsync(d_busy, k, BUSY, foreign);
        for (int step = 0; step < 3; ++step) {
            for (int k = 0; k < B; ++k) slots[k] = (k + step) % B; /* cloud k meets map (k + step) mod 4: both halves change hands */
            if (step == 1) {
                CHECK(gg_reset_maps(ctx, 0, B, 0.0, 0.0, 0.0f, 1, foreign));
                for (int k = 0; k < B; ++k)
                    for (size_t i = 0; i < C; ++i) refs[k]->layer[GGO_GROUND][i] = 0.0f, refs[k]->layer[GGO_GROUNDPATCH][i] = (float)0.0000001;
            }
            CHECK(gg_filter_batch(ctx, &bb, foreign));
            for (int k = 0; k < B; ++k)
                ggo_filter_cloud(refs[slots[k]], &rcfg, (const ggo_point *)(hc + (size_t)k * M), M, orgs + 3 * k, bzs[k], NULL, rlabel + 0, NULL, NULL, NULL);
        }
        for (int k = 0; k < B; ++k) { /* getters: ordered after both halves by the library */
            CHECK(gg_get_layer(ctx, k, GG_LAYER_GROUND, layer));
            ok &= same_floats(layer, refs[k]->layer[GGO_GROUND], C, "ground of a map after three divided batches");
            CHECK(gg_get_layer(ctx, k, GG_LAYER_POINTS, layer));
            ok &= same_floats(layer, refs[k]->layer[GGO_POINTS], C, "points of a map after three divided batches");
        }
        /* the caller's own read of the outputs: behind the fence, on its stream */
        CHECK(gg_filter_batch(ctx, &bb, foreign));
        CHECK(gg_batch_fence(ctx, foreign));
        hipMemcpyAsync(hl, dl, (size_t)B * M, hipMemcpyDeviceToHost, foreign);
        hipStreamSynchronize(foreign);
        for (int k = 0; k < B; ++k) {
            ggo_filter_cloud(refs[slots[k]], &rcfg, (const ggo_point *)(hc + (size_t)k * M), M, orgs + 3 * k, bzs[k], NULL, rlabel, NULL, NULL, NULL);
            ok &= memcmp(hl + (size_t)k * M, rlabel, M) == 0;
        }
        if (!ok) printf("step 5 (concurrent halves) failed\n");
        CHECK(gg_synchronize(ctx));
        for (int k = 0; k < B; ++k) ggo_map_destroy(refs[k]);
        hipFree(dc);
        hipFree(dl);
        free(hc);
        free(hl);
        gg_destroy(ctx2);
        ctx = ctx_outer;
    }

    printf(ok ? "streams + async: bit-identical to the oracle\n" : "FAILED\n");
    hipFree(d_cloud);
    hipFree(d_labels);
    hipFree(d_busy);
    hipStreamDestroy(foreign);
    gg_destroy(ctx);
    ggo_map_destroy(ref);
    return ok ? 0 : 1;
}
