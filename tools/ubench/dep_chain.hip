// Dependent-chain latencies of the candidate quotient sequences of k_reduce (one wavefront, s_memtime around 4096 iterations).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/dep_chain.hip -o /tmp/dep_chain && /tmp/dep_chain
#include <hip/hip_runtime.h>
#include <cstdio>

template <int V>
__global__ void k(float *out, unsigned long long *cyc, float x0, float b, double r, float rf)
{
    float x = x0 + threadIdx.x;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
    for (int i = 0; i < 4096; ++i) {
        if (V == 0) x = __builtin_fmaf(x, rf, b);                         // 1 fma
        if (V == 1) x = (float)((double)x * r) + b;                        // cvt, mul64, cvt, add
        if (V == 2) {                                                      // mul + 4 fma (+ add)
            const float q0 = x * rf;
            const float e0 = __builtin_fmaf(-b, q0, x);
            const float q1 = __builtin_fmaf(e0, rf, q0);
            const float e1 = __builtin_fmaf(-b, q1, x);
            x = __builtin_fmaf(e1, rf, q1) + b;
        }
        if (V == 3) x = x / b + b;                                         // IEEE division + add
        if (V == 4) x = (float)((double)x * r);                            // cvt, mul64, cvt
        if (V == 5) { double d = (double)x; d = d * r; d = d * r; d = d * r; d = d * r; x = (float)d; } // cvt + 4 mul64 + cvt
        if (V == 6) { x = x + b; x = x * rf; x = x - b; x = x * rf; }      // 4 plain f32 ops
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
    float *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 256);
    hipMalloc(&cyc, 8);
    const char *names[] = {"fma", "cvt+mul64+cvt+add", "mul+4fma+add", "div+add", "cvt+mul64+cvt", "cvt+4mul64+cvt", "4 f32 ops"};
#define RUN(V)                                                                                   \
    for (int rep = 0; rep < 2; ++rep) {                                                          \
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(64), 0, 0, out, cyc, 1.5f, 7.0f, 1.0 / 7.0, 1.0f / 7.0f); \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                            \
    }                                                                                            \
    printf("%-20s %7.1f cycles per iteration\n", names[V], (double)h / 4096.0);
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}
