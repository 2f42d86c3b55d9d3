// mfma_f16_power_probe.hip -- the SUSTAINED v_mfma_f32_32x32x16_f16 rate under the socket power cap: the same MFMA-only loop with
// constant operands (what tools/mfma_f16_probe.hip measures: nothing toggles) and with random fp16 operands that change
// every iteration (register-resident, no memory traffic), for ~1 s each, reporting the rate of the last 100 ms.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_power_probe.hip -o tools/_build/mfma_f16_power_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0 constant, 1 random operands rotated through 8 register sets, 2 random A only changes
__global__ __launch_bounds__(256) void loop(float* out, int iters, unsigned seed) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a[8], b[8];
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    for (int k = 0; k < 8; ++k) {
        u32x4 ua, ub;
        for (int q = 0; q < 4; ++q) {
            s = s * 1664525u + 1013904223u; const unsigned r1 = s; s = s * 1664525u + 1013904223u; const unsigned r2 = s;
            // two fp16 in [1, 2) x sign: exponent 15, random mantissa -- finite, no growth problems in fp32 accumulators
            ua[q] = MODE ? ((r1 & 0x83FF83FFu) | 0x3C003C00u) : 0x3C003C00u;
            ub[q] = MODE ? ((r2 & 0x83FF83FFu) | 0x3C003C00u) : 0x40004000u;
        }
        a[k] = __builtin_bit_cast(f16x8, ua); b[k] = __builtin_bit_cast(f16x8, ub);
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + i) & 7], b[MODE == 2 ? 0 : ((k + 2 * i) & 7)], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    if (t == 12345.678f) out[0] = t;
}

template <int MODE> static void run(const char* what) {
    float* d; if (hipMalloc(&d, 4) != hipSuccess) exit(1);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 512, iters = 20000;     // 2 waves per SIMD; one launch ~ 10 ms
    const double flops = (double)grid * 4 * iters * 32.0 * 32768.0;
    float last = 0.f, first = 0.f;
    for (int rep = 0; rep < 100; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(loop<MODE>, dim3(grid), dim3(256), 0, 0, d, iters, 1234u + rep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep == 1) first = ms;
        if (rep >= 90) last += ms / 10.f;
    }
    printf("%-34s launch 2: %.2f ms = %.0f TFLOP/s   sustained (launches 91-100): %.2f ms = %.0f TFLOP/s fp16 = %.0f as 3-product fp32\n", what, first,
           flops / first * 1e-9, last, flops / last * 1e-9, flops / last * 1e-9 / 3);
    (void)hipFree(d);
}
int main() { run<0>("constant operands"); run<1>("random operands (A and B)"); run<2>("random A, fixed B"); run<0>("constant operands again"); return 0; }
