// l1_l2_bw_probe.hip -- how many bytes per clock a CU pulls through its vector L1 with buffer_load_dwordx4 (1 KB per wave
// instruction, coalesced), by working-set size: per-workgroup 16 KB (L1-resident) .. 1 MB, chip-wide region 8 MB (L2),
// 128 MB (Infinity Cache), 512 MB (HBM).  The split-f16 convolution needs 32 .. 48 KB per K-tile per workgroup.
//   hipcc --offload-arch=gfx950 -O3 tools/l1_l2_bw_probe.hip -o tools/_build/l1_l2_bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int INFLIGHT>
__global__ __launch_bounds__(256) void rd(const float* base, unsigned wg_bytes, unsigned region_bytes, int iters, float* out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0xFFFFFFFFu, 0x00020000);
    const unsigned wg_base = (unsigned)(((unsigned long long)blockIdx.x * wg_bytes) % region_bytes);
    unsigned off = threadIdx.x * 16;            // 256 threads x 16 B = 4 KB per "row" of the workgroup
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        f32x4 v[INFLIGHT];
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) {
            v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, wg_base + off, 0, 0));
            off += 4096; if (off >= wg_bytes) off -= wg_bytes;
        }
#pragma unroll
        for (int k = 0; k < INFLIGHT; ++k) acc += v[k];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = acc[0];
}

int main() {
    const size_t region = 512u << 20;
    float *buf, *out;
    if (hipMalloc(&buf, region) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
    (void)hipMemset(buf, 0, region);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const unsigned wgs[] = {16u << 10, 64u << 10, 256u << 10, 1u << 20};
    const unsigned regions[] = {8u << 20, 128u << 20, 512u << 20};
    for (int wpc = 1; wpc <= 2; ++wpc)
        for (unsigned wg : wgs)
            for (unsigned reg : regions) {
                const int grid = 256 * wpc, iters = 2000;
                auto launch = [&] { hipLaunchKernelGGL(rd<8>, dim3(grid), dim3(256), 0, 0, buf, wg, reg, iters, out); };
                launch(); (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                const double bytes = (double)grid * 256 * 16 * 8 * iters;
                printf("wg/CU %d  per-wg set %4u KB  region %3u MB: %.3f ms  %.2f TB/s  %.1f B/clk/CU (at 2.4 GHz)\n", wpc, wg >> 10, reg >> 20, ms,
                       bytes / ms * 1e-9, bytes / (ms * 1e-3) / 256 / 2.4e9);
            }
    return 0;
}
