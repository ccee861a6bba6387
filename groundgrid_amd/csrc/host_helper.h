// host_helper.h -- the host-side loops of the host-buffer entry points (gg_filter_cloud and friends) and the helper threads
// that share them: plain C++17, no HIP.  gg_context.hip is its only user in the library; tests/cpp/test_host_helper.cpp builds
// it alone under ThreadSanitizer (the reference was never run under a sanitizer, its insertion threads race).
#pragma once

#include <sched.h>
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "groundgrid_hip.h"

// Helper threads of a context for the host-buffer entry points: packing the input cloud and assembling the returned cloud are two
// memory-bound loops of ~0.07 and ~0.12 ms per HDL-64E cloud on one core, a third of what a synchronous gg_filter_cloud costs
// beyond its kernels.  A range is cut into equal parts, the caller's thread takes the first, the helpers the others.
// GG_HOST_THREADS = threads per context including the caller's (default: 8, but never more than the CPUs this process may run
// on -- sched_getaffinity, so a cgroup / taskset limit counts; 1 = everything on the caller's thread).  The helpers are created
// on the first split that is large enough to want them, not at gg_create (a context that only ever runs batches has none), and
// a child process after fork() -- which inherits the object but not the threads -- starts its own.  The caller waits for the
// parts with a short spin, then yields, then sleeps on the condition variable: a descheduled helper costs the caller a
// context switch, not its time slice.  An idle helper does the same from its side (loop()).
class HostHelper {
  public:
    HostHelper() = default;
    HostHelper(const HostHelper &) = delete;
    ~HostHelper() { stop(); }
    void configure(int helpers) { wanted_ = std::max(0, helpers); }
    static int usable_cpus()
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof set, &set) == 0) {
            const int n = CPU_COUNT(&set);
            if (n > 0) return n;
        }
        const unsigned hc = std::thread::hardware_concurrency();
        return hc ? (int)hc : 1;
    }
    void stop()
    {
        if (!st_) return;
        if (st_->owner == getpid()) {
            {
                std::lock_guard<std::mutex> g(st_->m);
                st_->quit = true;
                ++st_->epoch;
            }
            st_->cv.notify_all();
            for (auto &t : st_->threads) t.join();
            delete st_;
        }
        // (a forked child: the threads, and whatever they were doing with the mutex and the condition variables, stayed in the
        // parent -- the copy is abandoned, not destroyed)
        st_ = nullptr;
    }
    // fn(lo, hi) over [0, n) in parts() pieces; returns when all of them are done
    template <class F> void split(size_t n, F fn)
    {
        if (wanted_ == 0 || n < 4096) {
            fn((size_t)0, n);
            return;
        }
        ensure_threads();
        State &st = *st_;
        const size_t parts = st.threads.size() + 1;
        if (parts == 1) {
            fn((size_t)0, n);
            return;
        }
        {
            std::lock_guard<std::mutex> g(st.m);
            st.job = [&fn, n, parts](int k) { fn(n * (size_t)(k + 1) / parts, n * (size_t)(k + 2) / parts); };
            st.pending.store((int)st.threads.size(), std::memory_order_relaxed);
            ++st.epoch;
        }
        st.cv.notify_all();
        fn((size_t)0, n / parts);
        for (int spins = 0; st.pending.load(std::memory_order_acquire) != 0; ++spins) {
            if (spins < 2000) { // (the parts are equal: normally a few hundred nanoseconds)
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
            } else if (spins < 2200) {
                std::this_thread::yield();
            } else { // a helper is not running (oversubscribed cores, a CPU limit): sleep until the last one reports
                // (the last helper notifies under the lock, after its decrement: no wake-up is lost.  A plain wait, not wait_for:
                // gcc 11's ThreadSanitizer does not know pthread_cond_clockwait and reports phantom double locks around it.)
                std::unique_lock<std::mutex> lk(st.m);
                st.done_cv.wait(lk, [&] { return st.pending.load(std::memory_order_acquire) == 0; });
            }
        }
    }

  private:
    struct State { // everything the helper threads touch, in one heap object (see stop() for why)
        std::vector<std::thread> threads;
        std::mutex m;
        std::condition_variable cv, done_cv;
        std::function<void(int)> job;
        std::atomic<int> pending{0};
        std::atomic<unsigned long> epoch{0};
        std::atomic<bool> quit{false};
        pid_t owner = 0;
        int idle_spins = 4000; // ~60 us (GG_HOST_HELPER_SPINS: 0 = sleep at once; measured 0 / 1500 / 4000: fused call 0.598 / 0.589 / 0.576 ms)
    };
    void ensure_threads()
    {
        if (st_ && st_->owner == getpid()) return;
        st_ = new State(); // (first use, or a forked child: the parent's State is abandoned)
        st_->owner = getpid();
        if (const char *e = getenv("GG_HOST_HELPER_SPINS")) st_->idle_spins = std::max(0, atoi(e));
        const int n = std::min(wanted_, usable_cpus() - 1);
        State *st = st_;
        for (int k = 0; k < n; ++k) st->threads.emplace_back([st, k] { loop(*st, k); }); // (epoch 0 is what they have seen so far)
    }
    static void loop(State &st, int k)
    {
        unsigned long seen = 0;
        for (;;) {
            // a split often follows another within microseconds (the two pieces of a cloud being packed, the copy-out behind a short
            // call): look for it with a short spin (~60 us) before sleeping on the condition variable -- a futex wake-up of seven
            // threads is 10-20 us of the caller's critical path
            bool found = false;
            for (int spins = 0; spins < st.idle_spins && !found; ++spins) {
                found = st.epoch.load(std::memory_order_acquire) != seen;
#if defined(__x86_64__)
                if (!found) __builtin_ia32_pause();
#endif
            }
            if (!found) {
                std::unique_lock<std::mutex> lk(st.m);
                st.cv.wait(lk, [&] { return st.epoch.load(std::memory_order_acquire) != seen; });
            }
            seen = st.epoch.load(std::memory_order_acquire); // (the job was stored before the epoch moved; the next epoch cannot come before this job is done)
            if (st.quit.load(std::memory_order_acquire)) return;
            st.job(k); // (the job stays valid until `pending` reaches 0: split() does not return before)
            const bool last = st.pending.fetch_sub(1, std::memory_order_acq_rel) == 1;
            if (last) {
                std::lock_guard<std::mutex> g(st.m); // (under the lock, after the decrement: no wake-up of a sleeping caller is lost)
                st.done_cv.notify_all();
            }
        }
    }
    State *st_ = nullptr;
    int wanted_ = 0;
};

// pack PointXYZIR -> 16-B records while copying into pinned staging (halves PCIe and HBM traffic; only x, y, z, ring are
// ever read, :222-250).  One 16-byte load + the ring per point, one 16-byte store: the loop vectorises to SSE moves.
inline void pack_points(const gg_point32 *__restrict__ cloud, gg_point16 *__restrict__ dst, size_t n)
{
    for (size_t i = 0; i < n; ++i) {
        gg_point16 d;
        d.x = cloud[i].x;
        d.y = cloud[i].y;
        d.z = cloud[i].z;
        d.ring = cloud[i].ring;
        d.pad = 0;
        dst[i] = d;
    }
}

// The returned cloud (src/GroundSegmentation.cpp:173-189) for input points [i0, i1): the host owns the input cloud, so it
// assembles the output from what the device sent back -- position in the returned cloud (or -1: dropped) and label per input
// point.  Every input point has its own position, so ranges of the input write disjoint records.  `tf` (or null): the cloud came
// in the sensor frame and the returned cloud is in the map frame, same arithmetic as the device (build with -ffp-contract=off).
inline void assemble_returned_cloud(const gg_point32 *__restrict__ cloud, const int32_t *__restrict__ h_index, const uint8_t *__restrict__ h_labels,
                                    const double *tf, gg_point32 *__restrict__ out_cloud, size_t i0, size_t i1)
{
    for (size_t i = i0; i < i1; ++i) {
        const int32_t k = h_index[i];
        if (k < 0) continue;
        out_cloud[k] = cloud[i];
        if (tf) {
            const double dx = (double)cloud[i].x, dy = (double)cloud[i].y, dz = (double)cloud[i].z;
            out_cloud[k].x = (float)(((tf[0] * dx + tf[1] * dy) + tf[2] * dz) + tf[3]);
            out_cloud[k].y = (float)(((tf[4] * dx + tf[5] * dy) + tf[6] * dz) + tf[7]);
            out_cloud[k].z = (float)(((tf[8] * dx + tf[9] * dy) + tf[10] * dz) + tf[11]);
        }
        out_cloud[k].intensity = (float)h_labels[i];
    }
}
