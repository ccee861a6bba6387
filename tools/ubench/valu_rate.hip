// VALU issue rates on gfx950: how many cycles a SIMD needs per wave64 instruction of each kind, with W wavefronts per SIMD
// (independent instruction streams inside each wave, so that neither dependences nor memory limit the rate).
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ void k(float *out, unsigned long long *cyc, float b, double rd)
{
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3;
    const f2 bb = {b, b};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 4
    for (int i = 0; i < 2048; ++i) {
        if (V == 0) { // 8 independent v_fma_f32
            x0 = __builtin_fmaf(x0, b, b); x1 = __builtin_fmaf(x1, b, b); x2 = __builtin_fmaf(x2, b, b); x3 = __builtin_fmaf(x3, b, b);
            x4 = __builtin_fmaf(x4, b, b); x5 = __builtin_fmaf(x5, b, b); x6 = __builtin_fmaf(x6, b, b); x7 = __builtin_fmaf(x7, b, b);
        }
        if (V == 1) { // 4 independent v_pk_fma_f32 (8 fmas), then 4 more = 8 instructions
            p0 = __builtin_elementwise_fma(p0, bb, bb); p1 = __builtin_elementwise_fma(p1, bb, bb); p2 = __builtin_elementwise_fma(p2, bb, bb); p3 = __builtin_elementwise_fma(p3, bb, bb);
            p0 = __builtin_elementwise_fma(p0, bb, bb); p1 = __builtin_elementwise_fma(p1, bb, bb); p2 = __builtin_elementwise_fma(p2, bb, bb); p3 = __builtin_elementwise_fma(p3, bb, bb);
        }
        if (V == 2) { // 8 v_mul_f64 (4 streams x 2)
            d0 *= rd; d1 *= rd; d2 *= rd; d3 *= rd; d0 *= rd; d1 *= rd; d2 *= rd; d3 *= rd;
        }
        if (V == 3) { // 8 conversions f32 -> f64 -> f32 (4 streams)
            x0 = (float)((double)x0); x1 = (float)((double)x1); x2 = (float)((double)x2); x3 = (float)((double)x3);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
        }
        if (V == 4) { // 8 v_add_u32 (integer)
            asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1\n v_add_u32 %4, %4, 1\n v_add_u32 %5, %5, 1\n v_add_u32 %6, %6, 1\n v_add_u32 %7, %7, 1"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
        if (V == 5) { // 8 v_cndmask / v_min3 mix
            x0 = fminf(x0, b); x1 = fminf(x1, b); x2 = fminf(x2, b); x3 = fminf(x3, b); x4 = fmaxf(x4, b); x5 = fmaxf(x5, b); x6 = fmaxf(x6, b); x7 = fmaxf(x7, b);
            asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + (float)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out;
    unsigned long long *cyc, h;
    hipMalloc(&out, 4096 * 4);
    hipMalloc(&cyc, 64);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_mul_f64", "cvt f32<->f64", "v_add_u32", "v_min/max_f32"};
#define RUN(V, WAVES)                                                                                    \
    for (int rep = 0; rep < 2; ++rep) {                                                                  \
        hipLaunchKernelGGL(k<V>, dim3(1), dim3(64 * WAVES), 0, 0, out, cyc, 1.0000001f, 1.0000000001);  \
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                    \
    }                                                                                                    \
    printf("%-16s %2d waves in one work-group (%d per SIMD): %6.2f cycles per instruction per wave  -> %5.2f per SIMD\n", names[V], WAVES, (WAVES + 3) / 4, (double)h / (2048.0 * 8.0), (double)h / (2048.0 * 8.0) / ((WAVES + 3) / 4));
    RUN(0, 1) RUN(0, 4) RUN(0, 8) RUN(0, 16)
    RUN(1, 1) RUN(1, 4) RUN(1, 8) RUN(1, 16)
    RUN(2, 1) RUN(2, 4) RUN(2, 8)
    RUN(3, 1) RUN(3, 4) RUN(3, 8)
    RUN(4, 1) RUN(4, 8)
    RUN(5, 1) RUN(5, 8)
    return 0;
}
