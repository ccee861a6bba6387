// CPU-only exercise of the host-side loops of the host-buffer entry points (groundgrid_amd/csrc/host_helper.h: the helper
// threads of a context, pack_points, assemble_returned_cloud), built without HIP so that it can run under ThreadSanitizer:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -I include -I groundgrid_amd/csrc tests/cpp/test_host_helper.cpp -lpthread
// (tests/test_sanitizers_cpu.py).  The reference's own host threads were never run under a sanitizer and race
// (src/GroundSegmentation.cpp:101-106); these must not.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>

#include <vector>

#include "host_helper.h"

static int failures = 0;
#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); \
            ++failures;                                              \
        }                                                            \
    } while (0)

static std::vector<gg_point32> make_cloud(size_t n, unsigned seed)
{
    std::vector<gg_point32> c(n);
    memset(c.data(), 0, n * sizeof(gg_point32));
    for (size_t i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        c[i].x = (float)(seed >> 8) * 1e-5f;
        c[i].y = (float)(i % 977) * 0.25f;
        c[i].z = -1.7f + (float)(i % 13) * 0.01f;
        c[i].intensity = 0.5f;
        c[i].ring = (uint16_t)(i % 64);
    }
    return c;
}

// one "call": pack the cloud in parts, then assemble a returned cloud in parts; both against the serial loops
static void one_call(HostHelper &h, size_t n, unsigned seed, bool with_tf)
{
    const std::vector<gg_point32> cloud = make_cloud(n, seed);
    std::vector<gg_point16> packed(n), packed_ref(n);
    memset(packed.data(), 0xAB, n * sizeof(gg_point16));
    memset(packed_ref.data(), 0xAB, n * sizeof(gg_point16));
    h.split(n, [&](size_t a0, size_t a1) { pack_points(cloud.data() + a0, packed.data() + a0, a1 - a0); });
    pack_points(cloud.data(), packed_ref.data(), n);
    CHECK(memcmp(packed.data(), packed_ref.data(), n * sizeof(gg_point16)) == 0);

    // every third point is dropped, the others keep their order (kept first, as the path emits them)
    std::vector<int32_t> index(n);
    std::vector<uint8_t> label(n);
    int32_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        index[i] = (i % 3 == 2) ? -1 : k++;
        label[i] = (i % 5 == 0) ? 99 : 49;
    }
    const double tf[12] = {0.6, -0.8, 0.0, 1.5, 0.8, 0.6, 0.0, -2.0, 0.0, 0.0, 1.0, 0.25};
    std::vector<gg_point32> out((size_t)k), out_ref((size_t)k);
    memset(out.data(), 0, out.size() * sizeof(gg_point32));
    memset(out_ref.data(), 0, out_ref.size() * sizeof(gg_point32));
    h.split(n, [&](size_t i0, size_t i1) { assemble_returned_cloud(cloud.data(), index.data(), label.data(), with_tf ? tf : nullptr, out.data(), i0, i1); });
    assemble_returned_cloud(cloud.data(), index.data(), label.data(), with_tf ? tf : nullptr, out_ref.data(), 0, n);
    CHECK(memcmp(out.data(), out_ref.data(), out.size() * sizeof(gg_point32)) == 0);
}

int main()
{
    CHECK(HostHelper::usable_cpus() >= 1);
    for (int helpers : {0, 1, 3, 7}) {
        HostHelper h;
        h.configure(helpers);
        for (int rep = 0; rep < 40; ++rep) one_call(h, rep % 7 == 0 ? 100 : 20000 + 997 * (size_t)rep, 17u * (unsigned)rep + (unsigned)helpers, rep % 2 == 1);
    }
    {
        // two contexts' helpers side by side (two GroundSegmentation objects in one process), interleaved calls
        HostHelper a, b;
        a.configure(3);
        b.configure(2);
        for (int rep = 0; rep < 20; ++rep) {
            one_call(a, 30000, 5u + (unsigned)rep, false);
            one_call(b, 12345, 9u + (unsigned)rep, true);
        }
    }
#if !defined(__SANITIZE_THREAD__) // (ThreadSanitizer refuses to start threads in the child of a multi-threaded fork)
    {
        // fork(): the child inherits the object but not its threads and must start its own instead of waiting for the parent's
        HostHelper h;
        h.configure(3);
        one_call(h, 50000, 3u, false);
        fflush(stderr);
        const pid_t pid = fork();
        if (pid == 0) {
            const int before = failures;
            one_call(h, 50000, 4u, true);
            _exit(failures == before ? 0 : 1);
        }
        int status = 0;
        CHECK(pid > 0 && waitpid(pid, &status, 0) == pid && WIFEXITED(status) && WEXITSTATUS(status) == 0);
        one_call(h, 50000, 5u, false);
    }
#endif
    if (failures) {
        fprintf(stderr, "%d check(s) failed\n", failures);
        return 1;
    }
    printf("host helper OK\n");
    return 0;
}
