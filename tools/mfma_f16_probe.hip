// mfma_f16_probe.hip -- what v_mfma_f32_32x32x16_f16 does with the operands of a split-f16 ("hi + lo") fp32 emulation:
// (1) numerics: subnormal f16 inputs, exactness of the 16-term internal sum, v_cvt_f16_f32 rounding / subnormal results;
// (2) rate: MFMA-only loop, and the same loop with NV vector-ALU instructions after every MFMA (1 and 2 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_probe.hip -o tools/_build/mfma_f16_probe && tools/_build/mfma_f16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// ---- numerics: one MFMA, A row i / B col j given per test ------------------------------------------------------------
// A operand: lane l holds row l % 32, k = 8 * (l / 32) + 0..7; B likewise for columns.  D[row][col]: col = lane & 31,
// row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
__global__ void one_mfma(const _Float16* a, const _Float16* b, const float* c, float* d) {
    const int l = threadIdx.x;
    f16x8 av, bv;
    for (int q = 0; q < 8; ++q) { av[q] = a[(l % 32) * 16 + 8 * (l / 32) + q]; bv[q] = b[(l % 32) * 16 + 8 * (l / 32) + q]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = c[0];
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[r];
}
__global__ void cvt_probe(const float* x, float* hi, float* lo, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    const _Float16 h = (_Float16)x[i];
    hi[i] = (float)h;
    lo[i] = (float)(_Float16)(x[i] - (float)h);
}

static float run_row0(const float* arow, const float* brow, float cval) {   // D[0][0] for A row 0 = arow[16], B col 0 = brow[16]
    _Float16 ha[32 * 16], hb[32 * 16];
    memset(ha, 0, sizeof ha); memset(hb, 0, sizeof hb);
    for (int k = 0; k < 16; ++k) { ha[k] = (_Float16)arow[k]; hb[k] = (_Float16)brow[k]; }
    _Float16 *da, *db; float *dc, *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dc, 4); hipMalloc(&dd, 32 * 32 * 4);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipMemcpy(dc, &cval, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(one_mfma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
    float out[32 * 32];
    hipMemcpy(out, dd, sizeof out, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dc); hipFree(dd);
    return out[0];
}

// ---- rate --------------------------------------------------------------------------------------------------------------
template <int NACC, int NV, int KIND>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 x, y;
    for (int q = 0; q < 8; ++q) { x[q] = (_Float16)(a + threadIdx.x * 1e-3f); y[q] = (_Float16)b; }
    unsigned c0 = threadIdx.x, c1 = threadIdx.x * 3u, c2 = threadIdx.x * 5u, c3 = threadIdx.x * 7u, d = 0x9E3779B9u;
    float f0 = a, f1 = b, f2 = a + b, f3 = a - b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[i], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    unsigned& c = (v & 3) == 0 ? c0 : (v & 3) == 1 ? c1 : (v & 3) == 2 ? c2 : c3;
                    float& f = (v & 3) == 0 ? f0 : (v & 3) == 1 ? f1 : (v & 3) == 2 ? f2 : f3;
                    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                    else if (KIND == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(d));
                    else if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f) : "v"(b));
                    else asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(c) : "v"(f));
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f || (c0 ^ c1 ^ c2 ^ c3) == 0x12345u || f0 + f1 + f2 + f3 == 1.2345f) out[0] = s;
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
    const double flops = (double)grid * 4 * iters * 4.0 * NACC * 32768.0;
    static const char* kn[] = {"v_add_u32", "v_mul_lo_u32", "v_fma_f32", "v_cvt_f16_f32"};
    printf("chains %d, waves/SIMD %d, %d x %s per MFMA: %.3f ms, %.1f TFLOP/s (f16), %.1f as 3-product fp32\n", NACC, blocks_per_cu, NV,
           kn[KIND], best, flops / best * 1e-9, flops / best * 1e-9 / 3);
    (void)hipFree(d);
}

int main() {
    // ---- numerics ----
    float a[16] = {0}, b[16] = {0};
    a[0] = ldexpf(1.f, -20); b[0] = 1.f;                       // subnormal f16 A input
    printf("subnormal A (2^-20) * 1           = %.9g (expect %.9g)\n", run_row0(a, b, 0.f), ldexpf(1.f, -20));
    a[0] = ldexpf(1.f, -24); b[0] = ldexpf(1.f, -24);          // product 2^-48: fine in fp32
    printf("smallest subnormal squared        = %.9g (expect %.9g)\n", run_row0(a, b, 0.f), ldexpf(1.f, -48));
    for (int k = 0; k < 16; ++k) { a[k] = 1.f; b[k] = 1.f; }
    a[0] = 2048.f; b[0] = 8192.f;                              // 2^24 + 15 ones: exact sum 16777231 (odd: not representable; 16777232 by RNE)
    printf("2^24 + 15 ones                    = %.9g (exact 16777231; fp32 chain of +1 gives 16777216)\n", run_row0(a, b, 0.f));
    a[1] = -2048.f; b[1] = 8192.f;                             // 2^24 - 2^24 + 14 ones = 14
    printf("2^24 - 2^24 + 14 ones             = %.9g (expect 14)\n", run_row0(a, b, 0.f));
    for (int k = 0; k < 16; ++k) { a[k] = 1.f + ldexpf(1.f, -10); b[k] = 1.f + ldexpf(1.f, -10); }   // (1+2^-10)^2 = 1 + 2^-9 + 2^-20: exact in fp32
    printf("16 x (1+2^-10)^2                  = %.9g (expect %.9g)\n", run_row0(a, b, 0.f), 16.f * (1.f + ldexpf(1.f, -9) + ldexpf(1.f, -20)));
    for (int k = 0; k < 16; ++k) { a[k] = 1.f; b[k] = 1.f; }
    printf("C = 2^26, + 16 products of 1      = %.9g (exact 67108880)\n", run_row0(a, b, 67108864.f));
    for (int k = 0; k < 16; ++k) { a[k] = 0.f; b[k] = 0.f; }
    a[0] = 1.f; b[0] = 1.f; a[8] = ldexpf(1.f, -12); b[8] = ldexpf(1.f, -12);        // 1 + 2^-24: rounds to 1 (tie to even) in fp32
    a[9] = ldexpf(1.f, -12); b[9] = ldexpf(1.f, -12);                                // 1 + 2^-23 exactly if summed wide
    printf("1 + 2^-24 + 2^-24                 = %.9g (wide sum %.9g, fp32 chain 1)\n", run_row0(a, b, 0.f), 1.f + ldexpf(1.f, -23));
    {   // conversions
        const int n = 8;
        float x[n] = {1.0004883f, 3.1415927f, 1e-5f, 3e-8f, 6.1e-5f, 65520.f, 0.1f, -2.7182817f}, hi[n], lo[n];
        float *dx, *dh, *dl;
        hipMalloc(&dx, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
        hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_probe, dim3(1), dim3(64), 0, 0, dx, dh, dl, n);
        hipMemcpy(hi, dh, n * 4, hipMemcpyDeviceToHost); hipMemcpy(lo, dl, n * 4, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            printf("x = %-14.9g hi = %-14.9g lo = %-14.9g  x - (hi + lo) = %.3g (rel %.3g)\n", x[i], hi[i], lo[i],
                   (double)x[i] - ((double)hi[i] + (double)lo[i]), ((double)x[i] - ((double)hi[i] + (double)lo[i])) / x[i]);
    }
    // ---- rate ----
    const int iters = 4000;
    run<1, 0, 0>(1, iters); run<2, 0, 0>(1, iters); run<1, 0, 0>(2, iters); run<2, 0, 0>(2, iters);
    run<4, 0, 0>(1, iters); run<4, 0, 0>(2, iters); run<8, 0, 0>(1, iters);
    printf("-- VALU instructions in the shadow of the MFMAs\n");
    run<4, 1, 0>(1, iters); run<4, 2, 0>(1, iters); run<4, 4, 0>(1, iters); run<4, 6, 0>(1, iters); run<4, 8, 0>(1, iters);
    run<4, 2, 0>(2, iters); run<4, 4, 0>(2, iters); run<4, 8, 0>(2, iters);
    run<4, 2, 2>(1, iters); run<4, 4, 2>(1, iters); run<4, 4, 2>(2, iters);
    run<4, 2, 3>(1, iters); run<4, 4, 3>(1, iters); run<4, 4, 3>(2, iters);
    run<4, 1, 1>(1, iters); run<4, 2, 1>(2, iters);
    return 0;
}
