// winograd.hip -- Winograd F(2x2, 3x3) around the implicit-GEMM kernel, for the large 3x3 / stride-1
// convolutions (lib_yolo/layers.py:545-575 with kernel_size 3): 16 multiplies per 2x2 output tile and channel
// pair instead of 36, i.e. 2.25x fewer matrix-pipe cycles than the direct form.
//
//   V[xi][p][c] = (B^T d B)[xi]      input transform   (this file; one 4x4 input patch per output tile p)
//   M[xi][p][n] = sum_c V[xi][p][c] * U[xi][c][n]      16 GEMMs = ONE launch of conv_igemm_kernel over
//                                                       16*P rows with a per-row-block weight matrix
//   Y[2x2 of p][n] = (A^T M A)       output transform + the fused epilogue of the direct kernel
//                                     (dropout mask, BN scale / shift, leaky, residual)   (this file)
//   U[xi][c][n] = (G g G^T)[xi]      weight transform, once, on the host (byolo_finalize)
//
// with the standard matrices
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// The transforms are HBM-bound streaming kernels (V is 4x the input, M 4x the output), so the samples of a
// layer are processed in chunks whose V and M stay in a bounded scratch region (byolo_plan.hip).
// Same arithmetic type (fp32) as the direct path; the rounding differs (tests/: same tolerance).
#include <hip/hip_runtime.h>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "epilogue.h"

namespace byk {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t wdiv(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shr; }

// thread = (tile p, 4 channels): 16 x 16-byte loads, 16 x 16-byte stores
// V is written with non-temporal stores, as in wino_split.hip (there: the GEMM that reads V 1.3 - 5 % faster; here, fp32 mode at config 4:
// 190.4 / 189.5 -> 191.1 / 191.7 img/s, inside the noise); BYOLO_WF_NT_STORE=0: the A/B build
#ifndef BYOLO_WF_NT_STORE
#define BYOLO_WF_NT_STORE 1
#endif
__device__ __forceinline__ void vstore(float* at, const f32x4 v) {
    if constexpr (BYOLO_WF_NT_STORE != 0) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(at));
    else *reinterpret_cast<f32x4*>(at) = v;
}
__global__ __launch_bounds__(256) void wino_input_kernel(const WinoParams p) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c4n = (uint32_t)p.C >> 2;
    const uint32_t t = wdiv(gid, p.d_c4), c4 = gid - t * c4n;
    if (t >= (uint32_t)p.P) return;
    const uint32_t tt = (uint32_t)(p.th * p.tw);
    const uint32_t s = wdiv(t, p.d_tt), r = t - s * tt;
    const uint32_t ty = wdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
    const float* img = p.x + ((size_t)(p.s0 + s) * p.H * p.W) * p.C + c4 * 4;
    const int y0 = 2 * (int)ty - 1, x0 = 2 * (int)tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + i, x = x0 + j;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            d[i][j] = ok ? *reinterpret_cast<const f32x4*>(img + ((size_t)y * p.W + x) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    f32x4 u[4][4];                                   // B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[0][j] = d[0][j] - d[2][j];
        u[1][j] = d[1][j] + d[2][j];
        u[2][j] = d[2][j] - d[1][j];
        u[3][j] = d[1][j] - d[3][j];
    }
    float* v = p.v + (size_t)t * p.C + c4 * 4;
    const size_t xi_stride = (size_t)p.P_pad * p.C;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                    // (B^T d) B
        vstore(v + (size_t)(i * 4 + 0) * xi_stride, u[i][0] - u[i][2]);
        vstore(v + (size_t)(i * 4 + 1) * xi_stride, u[i][1] + u[i][2]);
        vstore(v + (size_t)(i * 4 + 2) * xi_stride, u[i][2] - u[i][1]);
        vstore(v + (size_t)(i * 4 + 3) * xi_stride, u[i][1] - u[i][3]);
    }
}

// thread = (tile p, 4 output channels): 16 x 16-byte loads, up to 4 x 16-byte stores; epilogue as in conv_igemm.hip
__global__ __launch_bounds__(256) void wino_output_kernel(const WinoParams p) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t n4n = (uint32_t)p.N >> 2;
    const uint32_t t = wdiv(gid, p.d_n4), n4 = gid - t * n4n;
    if (t >= (uint32_t)p.P) return;
    const uint32_t tt = (uint32_t)(p.th * p.tw);
    const uint32_t s = wdiv(t, p.d_tt), r = t - s * tt;
    const uint32_t ty = wdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
    const int n0 = (int)n4 * 4;
    const float* m = p.m + (size_t)t * p.N + n0;
    const size_t xi_stride = (size_t)p.P_pad * p.N;
    f32x4 a[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[i][j] = *reinterpret_cast<const f32x4*>(m + (size_t)(i * 4 + j) * xi_stride);
    f32x4 u[2][4];                                   // A^T m
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[0][j] = a[0][j] + a[1][j] + a[2][j];
        u[1][j] = a[1][j] - a[2][j] - a[3][j];
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(p.scale + n0);     // includes 1 / (1 - p) when the masks are on
    const f32x4 sf = *reinterpret_cast<const f32x4*>(p.shift + n0);
    const bool do_drop = p.flags & EPI_DROPOUT;
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const uint32_t oy = 2 * ty + dy, ox = 2 * tx + dx;
            if (oy >= (uint32_t)p.H || ox >= (uint32_t)p.W) continue;
            const f32x4 yv = dx == 0 ? u[dy][0] + u[dy][1] + u[dy][2] : u[dy][1] - u[dy][2] - u[dy][3];   // (A^T m) A
            const uint64_t pix = ((uint64_t)(p.s0 + s) * p.H + oy) * p.W + ox;
            const uint64_t idx0 = p.idx_base + pix * (uint64_t)p.N + (uint64_t)n0;
            bool keep[4] = {true, true, true, true};
            if (do_drop) epi::keep4(epi::DropRow(idx0, p.k1), 0, p.k0, p.thr, keep);     // N % 4 == 0: idx0 is a multiple of 4
            f32x4 o = epi::bn_act4(yv, sc, sf, keep, slope);
            const size_t off = (size_t)pix * p.N + n0;
            if (p.residual) o += *reinterpret_cast<const f32x4*>(p.residual + off);
            *reinterpret_cast<f32x4*>(p.y + off) = o;
        }
}

hipError_t launch_wino_input(const WinoParams& p, hipStream_t st) {
    const uint64_t total = (uint64_t)p.P * (p.C >> 2);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}
hipError_t launch_wino_output(const WinoParams& p, hipStream_t st) {
    const uint64_t total = (uint64_t)p.P * (p.N >> 2);
    hipLaunchKernelGGL(wino_output_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// U[xi] = G g G^T for one (cin, cout) pair: g[ky][kx] -> u[16]
void wino_weight_transform(const float g[9], float u[16]) {
    static const double G[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
    double t[4][3];                                  // in double, rounded once: the weights are transformed once
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) u[i * 4 + j] = (float)(t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2]);
}

}  // namespace byk
