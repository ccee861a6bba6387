// What a wave instruction costs the CU's texture addresser / L1 as a function of how many cache lines its 64 lanes name.
// Every work-group re-reads (or re-writes) its own 32 KB of a buffer (cache resident after the first pass); lane l of an
// instruction addresses element l * stride: stride 1 = 256 contiguous bytes (2 lines), 32 = one 128-byte line per lane (64 lines).
// Many wavefronts per CU, independent loads: the time per instruction is throughput, not latency.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/ta_lines.hip -o /tmp/ta_lines && /tmp/ta_lines
#include <hip/hip_runtime.h>
#include <cstdio>

template <int STORE>
__global__ __launch_bounds__(256) void k(float *buf, int stride, int iters, float *sink)
{
    float *base = buf + (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048; // 8 KB per wavefront
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (lane * stride + (i + u) * 64 * stride) & 2047;
            if (STORE) base[e] = (float)i;
            else acc += base[e];
        }
    }
    if (!STORE && acc == 123.456f) sink[0] = acc;
}

// ... with only every `every`-th lane taking part (one line per active lane)
__global__ __launch_bounds__(256) void ka(float *buf, int every, int iters, float *sink)
{
    float *base = buf + (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048;
    const int lane = threadIdx.x & 63;
    float acc = 0.0f;
    if (lane % every == 0)
        for (int i = 0; i < iters; i += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += base[((lane + (i + u) * 3) * 32) & 2047];
        }
    if (acc == 123.456f) sink[0] = acc;
}

// ... and as a function of the access WIDTH of a fully coalesced instruction (lane l reads / writes W consecutive dwords at l * W)
template <int STORE, int W>
__global__ __launch_bounds__(256) void kw(float *buf, int iters, float *sink)
{
    typedef float vec __attribute__((ext_vector_type(W)));
    vec *base = reinterpret_cast<vec *>(buf + (size_t)blockIdx.x * 8192 + (threadIdx.x >> 6) * 2048); // 8 KB per wavefront
    const int lane = threadIdx.x & 63;
    constexpr int N = 2048 / W; // vectors in the wavefront's region
    vec acc = 0.0f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = (lane + (i + u) * 64) & (N - 1);
            if (STORE) base[e] = (vec)(float)i;
            else acc += base[e];
        }
    }
    float a0 = 0.f;
    for (int k = 0; k < W; ++k) a0 += acc[k];
    if (!STORE && a0 == 123.456f) sink[0] = a0;
}

int main()
{
    const int wgs = 256 * 8, iters = 4096;
    float *buf, *sink;
    hipMalloc(&buf, (size_t)wgs * 8192 * 4);
    hipMalloc(&sink, 4);
    hipMemset(buf, 0, (size_t)wgs * 8192 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int store = 0; store < 2; ++store)
        for (int stride : {1, 2, 4, 8, 16, 32}) {
            float ms = 0.f;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (store) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(256), 0, 0, buf, stride, iters, sink);
                else hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(256), 0, 0, buf, stride, iters, sink);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
            }
            const double instr_per_cu = (double)wgs * 4 * iters / 256.0; // wave instructions per CU
            const int lines = stride >= 32 ? 64 : 2 * stride;
            printf("%s stride %2d (%2d lines per instruction): %.3f ms, %.1f ns = ~%.0f cycles (2.4 GHz) per wave instruction and CU\n", store ? "store" : "load ",
                   stride, lines, ms, 1e6 * ms / instr_per_cu, 2.4 * 1e6 * ms / instr_per_cu);
        }
#define RUNW(ST, W)                                                                                                       \
    {                                                                                                                     \
        float ms = 0.f;                                                                                                   \
        for (int rep = 0; rep < 2; ++rep) {                                                                               \
            hipEventRecord(a);                                                                                            \
            hipLaunchKernelGGL((kw<ST, W>), dim3(wgs), dim3(256), 0, 0, buf, iters, sink);                                \
            hipEventRecord(b);                                                                                            \
            hipEventSynchronize(b);                                                                                       \
            hipEventElapsedTime(&ms, a, b);                                                                               \
        }                                                                                                                 \
        const double instr_per_cu = (double)wgs * 4 * iters / 256.0;                                                      \
        printf("%s coalesced, %2d bytes per lane: %.3f ms, ~%.0f cycles per wave instruction and CU (%.1f bytes per cycle)\n", ST ? "store" : "load ", \
               4 * W, ms, 2.4 * 1e6 * ms / instr_per_cu, 256.0 * W / (2.4 * 1e6 * ms / instr_per_cu));                     \
    }
    for (int every : {1, 2, 4, 8, 16}) {
        float ms = 0.f;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            hipLaunchKernelGGL(ka, dim3(wgs), dim3(256), 0, 0, buf, every, iters, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            hipEventElapsedTime(&ms, a, b);
        }
        printf("load, one line per active lane, %2d lanes active: ~%.0f cycles per wave instruction and CU\n", 64 / every, 2.4 * 1e6 * ms / ((double)wgs * 4 * iters / 256.0));
    }
    RUNW(0, 1) RUNW(0, 2) RUNW(0, 4) RUNW(1, 1) RUNW(1, 2) RUNW(1, 4)
    return 0;
}
