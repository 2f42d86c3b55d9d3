// mfma_peak.hip -- what the fp32 matrix pipe of this MI355X actually sustains, and what other instructions
// in the loop cost it.
//
// (1) A kernel that does nothing but v_mfma_f32_32x32x2_f32 (no memory, no LDS, no barrier): NACC independent
//     accumulator chains per wave, WPS waves per SIMD, every CU filled.  The number is the practical ceiling
//     the implicit-GEMM convolution is compared with in DESIGN.md (the nominal 157.3 TFLOP/s assumes 2.4 GHz
//     and a back-to-back issue with no bubble).
// (2) The same loop with NV vector-ALU instructions after every MFMA (full-rate v_add_u32, or quarter-rate
//     v_mul_lo_u32): how many VALU instructions per MFMA hide under the 64-cycle MFMA, with 1 and 2 waves
//     per SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/_build/mfma_peak && tools/_build/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int NV, int KIND>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float x = a + threadIdx.x * 1e-9f, y = b;
    unsigned c0 = threadIdx.x, c1 = threadIdx.x * 3u, c2 = threadIdx.x * 5u, c3 = threadIdx.x * 7u, d = 0x9E3779B9u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    unsigned& c = (v & 3) == 0 ? c0 : (v & 3) == 1 ? c1 : (v & 3) == 2 ? c2 : c3;
                    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                    else asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f || (c0 ^ c1 ^ c2 ^ c3) == 0x12345u) out[0] = s;
}

template <int NACC, int NV, int KIND>
static void run(int blocks_per_cu, int iters) {
    float* d;
    if (hipMalloc(&d, 4) != hipSuccess) exit(1);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL((mfma_loop<NACC, NV, KIND>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC, NV, KIND>), dim3(grid), dim3(256), 0, 0, d, iters, 1.f, 2.f);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flops = (double)grid * 4 /*waves*/ * iters * 4.0 * NACC * 4096.0;
    printf("chains %d, waves/SIMD %d, %d x %s per MFMA: %.3f ms, %.1f TFLOP/s\n", NACC, blocks_per_cu, NV,
           KIND == 0 ? "v_add_u32" : "v_mul_lo_u32", best, flops / best * 1e-9);
    (void)hipFree(d);
}

int main() {
    const int iters = 4000;
    run<1, 0, 0>(1, iters); run<4, 0, 0>(1, iters); run<4, 0, 0>(2, iters); run<4, 0, 0>(4, iters);
    printf("-- VALU instructions in the shadow of the MFMAs\n");
    run<4, 1, 0>(1, iters); run<4, 2, 0>(1, iters); run<4, 4, 0>(1, iters); run<4, 8, 0>(1, iters); run<4, 12, 0>(1, iters);
    run<4, 16, 0>(1, iters);
    run<4, 1, 0>(2, iters); run<4, 2, 0>(2, iters); run<4, 4, 0>(2, iters); run<4, 8, 0>(2, iters);
    run<4, 1, 1>(1, iters); run<4, 2, 1>(1, iters); run<4, 4, 1>(1, iters);
    run<4, 1, 1>(2, iters); run<4, 2, 1>(2, iters);
    return 0;
}
