// conv_kernels.hip -- the convolution work that is NOT the implicit GEMM of conv_igemm.hip: the direct
// 3-channel stem convolution and the BN-calibration helpers.
#include <hip/hip_runtime.h>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "epilogue.h"

namespace byk {
// finish_upsampled_kernel's stores (a wave writes whole cache lines, 16 bytes per lane, to a tensor the next launch streams once): non-temporal,
// like the Winograd transform's V (wino_split.hip vstore): 0.352 -> 0.323 ms per step at config 4, +0.25 % img/s; 0 = the A/B build.  The
// stem's stores measured +-0 with the hint and keep the plain form.
#ifndef BYOLO_NT_MISC
#define BYOLO_NT_MISC 1
#endif
__device__ __forceinline__ void st4(float* at, const epi::f32x4 v) {
    if constexpr (BYOLO_NT_MISC != 0) __builtin_nontemporal_store(v, reinterpret_cast<epi::f32x4*>(at));
    else *reinterpret_cast<epi::f32x4*>(at) = v;
}


typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shr; }

// ---------------------------------------------------------------------------------------------
// The stem (lib_yolo/darknet.py:10): 3x3, 3 -> NOUT channels, no dropout / residual.  Writes 128 B per pixel
// and reads 12: too little K for the matrix cores (27), so vector FMAs -- thread = one output pixel x ALL
// NOUT channels: its 27 inputs are loaded once, the 27*NOUT weights and the folded BN scale / shift are
// wave-uniform (scalar loads, SGPR operands of v_fmac), and each thread stores one full 128-byte line.
// Same summation order (ky, kx, c) as conv_direct_kernel.
// ---------------------------------------------------------------------------------------------
template <int NOUT>
__global__ __launch_bounds__(256) void conv_stem3x3_kernel(const ConvParams p) {
    // The 256 pixels of a block are consecutive rows of the [M][NOUT] output: 256 * NOUT * 4 contiguous bytes.  A thread that
    // stored its own NOUT * 4 = 128 bytes would touch 64 different lines per wave-store (0.9 TB/s measured); the rows go through
    // LDS instead (row stride NOUT * 4 + 16: conflict-free 16-byte accesses) and leave as whole lines, 1 KB per wave-store.
    constexpr int ROWB = NOUT * 4 + 16, PIECES = NOUT / 4;
    __shared__ __attribute__((aligned(16))) char lds[256 * ROWB];
    const uint32_t m0 = blockIdx.x * 256u, m = m0 + threadIdx.x;
    const bool live = m < (uint32_t)p.M;
    const uint32_t hw = (uint32_t)(p.Hout * p.Wout);
    const uint32_t mm = live ? m : 0u;
    const uint32_t s = fdiv(mm, p.d_hw), rem = mm - s * hw;
    const uint32_t oy = fdiv(rem, p.d_wout), ox = rem - oy * (uint32_t)p.Wout;
    const float* img = p.src0 + (size_t)fdiv(s, p.d_sdiv0) * p.Hs0 * p.Ws0 * 3;
    float x[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = (int)oy * p.stride - p.pad + ky, ix = (int)ox * p.stride - p.pad + kx;
            const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
            const float* px = img + ((size_t)(ok ? iy : 0) * p.Ws0 + (ok ? ix : 0)) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) x[(ky * 3 + kx) * 3 + c] = ok ? px[c] : 0.f;
        }
    // HWIO weights [27][NOUT] through LDS, read as broadcast 16-byte vectors (every lane the same address: no conflict).  As
    // wave-uniform scalar loads -- 27 * NOUT s_load results feeding v_fmac -- the kernel spent its time waiting for the scalar
    // cache: 0.43 ms per 8 images against 0.03 ms of FMA work (profiles/r3_a_bench_kernel_stats.md).
    f32x4* wl = reinterpret_cast<f32x4*>(lds);
    for (int i = threadIdx.x; i < 27 * NOUT / 4; i += 256) wl[i] = reinterpret_cast<const f32x4*>(p.wpk)[i];
    __syncthreads();
    float acc[NOUT];
#pragma unroll
    for (int n = 0; n < NOUT; ++n) acc[n] = 0.f;
#pragma unroll
    for (int k = 0; k < 27; ++k)
#pragma unroll
        for (int n = 0; n < NOUT; n += 4) {
            const f32x4 w4 = wl[k * (NOUT / 4) + n / 4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[n + q] = fmaf(x[k], w4[q], acc[n + q]);
        }
    __syncthreads();                              // the weights are read out: the rows take their place
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
    float vmax = 0.f;
#pragma unroll
    for (int n = 0; n < NOUT; n += 4) {
        f32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float y = acc[n + q] * p.scale[n + q] + p.shift[n + q];
            v[q] = fmaxf(y, slope * y);
        }
        vmax = epi::absmax4(vmax, v);
        *reinterpret_cast<f32x4*>(lds + threadIdx.x * ROWB + n * 4) = (p.split & 2) ? epi::split_encode4(v) : v;    // split precision: [4 hi | 4 lo]
    }
    if (live && (p.split & 2) && p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }   // conv_igemm.hip finish_tile
    __syncthreads();
    float* d = p.dst + (size_t)m0 * p.ldc;          // ldc == NOUT (launcher)
#pragma unroll
    for (int k = 0; k < PIECES; ++k) {
        const uint32_t q = k * 256u + threadIdx.x, row = q / PIECES, pc = q % PIECES;
        if (m0 + row < (uint32_t)p.M)
            *reinterpret_cast<f32x4*>(d + (size_t)q * 4) = *reinterpret_cast<const f32x4*>(lds + row * ROWB + pc * 16);
    }
}

// ---------------------------------------------------------------------------------------------
// Direct convolution, the general fallback: whatever the implicit-GEMM kernel does not take (an input-channel count
// that is not a multiple of 32 -- the stem's 3 when the fast stem kernel above does not apply, or a user graph's
// 24 / 40 / ...), with everything its loader folds in: two concatenated sources, nearest x2 upsampling, the T-fold
// sample tile, any output-channel count, the detection bias.  thread = (output pixel, group of 8 output channels);
// weights (HWIO, k*k*Cin x cout) are broadcast from LDS when they fit in 48 KB, read through the caches otherwise.
// Correct, not fast: Darknet-53 / YOLOv3 never come here except for the stem.
// ---------------------------------------------------------------------------------------------
// element c of a pixel's channel run starting at px: plain fp32, or hi + lo of a split-f16 tensor ([4 hi | 4 lo] per 4 channels)
__device__ __forceinline__ float act_at(const float* px, int c, bool split) {
    if (!split) return px[c];
    const _Float16* g = reinterpret_cast<const _Float16*>(px + (c & ~3));
    return (float)g[c & 3] + (float)g[4 + (c & 3)];
}

__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p, const int w_in_lds) {
    extern __shared__ __attribute__((aligned(16))) float wl[];    // [K][N] if w_in_lds
    const int Cin = p.C0 + p.C1, K = p.ksize * p.ksize * Cin, N = p.N;
    if (w_in_lds) {
        for (int i = threadIdx.x; i < K * N; i += blockDim.x) wl[i] = p.wpk[i];
        __syncthreads();
    }
    const float* wsrc = w_in_lds ? wl : p.wpk;
    const int groups = (N + 7) >> 3;
    const int64_t total = (int64_t)p.M * groups;
    const int hw = p.Hout * p.Wout;
    const bool in_split = p.split & 1, out_split = p.split & 2;      // split precision (N % 4 == 0 when out_split)
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(gid / groups), g = (int)(gid - (int64_t)m * groups);
        const int s = m / hw, rem = m - s * hw;
        const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
        const int nv = N - g * 8 < 8 ? N - g * 8 : 8;          // valid channels of this group
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
        for (int ky = 0; ky < p.ksize; ++ky) {
            const int iy = oy * p.stride - p.pad + ky;
            for (int kx = 0; kx < p.ksize; ++kx) {
                const int ix = ox * p.stride - p.pad + kx;
                if ((unsigned)iy >= (unsigned)p.Hin || (unsigned)ix >= (unsigned)p.Win) continue;     // zero padding
                const float* w = wsrc + (size_t)((ky * p.ksize + kx) * Cin) * N + g * 8;
                // source 0 then source 1 (channel concat, layers.py:588); each at its own resolution / sample divisor
                const float* px = p.src0 + (((size_t)(s / p.sdiv0) * p.Hs0 + (iy >> p.sh0)) * p.Ws0 + (ix >> p.sh0)) * p.C0;
                for (int c = 0; c < p.C0; ++c, w += N) {
                    const float x = act_at(px, c, in_split);
                    for (int o = 0; o < nv; ++o) acc[o] = fmaf(x, w[o], acc[o]);
                }
                if (p.C1) {
                    px = p.src1 + (((size_t)(s / p.sdiv1) * p.Hs1 + (iy >> p.sh1)) * p.Ws1 + (ix >> p.sh1)) * p.C1;
                    for (int c = 0; c < p.C1; ++c, w += N) {
                        const float x = act_at(px, c, in_split);
                        for (int o = 0; o < nv; ++o) acc[o] = fmaf(x, w[o], acc[o]);
                    }
                }
            }
        }
        float* d = p.dst + (size_t)m * p.ldc + g * 8;
        float res[8];
        for (int o = 0; o < nv; ++o) {
            const int n = g * 8 + o;
            float v = acc[o] * p.scale[n];                     // scale includes 1 / (1 - p) when the masks are on
            if (p.flags & EPI_DROPOUT) {
                const uint64_t el = (uint64_t)m * (uint64_t)N + (uint64_t)n;
                const bool keep = p.mask_bits ? ((p.mask_bits[el >> 5] >> (uint32_t)(el & 31u)) & 1u)
                                              : byolo_keep(p.idx_base + el, p.k0, p.k1, p.thr);
                if (!keep) v = 0.f;
            }
            v += p.shift[n];
            if (p.flags & EPI_LEAKY) v = fmaxf(v, 0.1f * v);
            if (p.flags & EPI_RESIDUAL) v += act_at(p.residual + (size_t)m * p.ldc, n, in_split);
            res[o] = v;
        }
        if (out_split) {
            float vmax = 0.f;
            for (int o = 0; o < nv; o += 4) {
                const f32x4 v{res[o], res[o + 1], res[o + 2], res[o + 3]};
                vmax = epi::absmax4(vmax, v);
                *reinterpret_cast<f32x4*>(d + o) = epi::split_encode4(v);
            }
            if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
        } else {
            for (int o = 0; o < nv; ++o) d[o] = res[o];
        }
    }
}

hipError_t launch_conv_direct(const ConvParams& p, hipStream_t st) {
    if (p.ksize == 3 && p.C0 == 3 && p.C1 == 0 && p.sh0 == 0 && p.N == 32 && (p.flags & ~EPI_LEAKY) == 0 && !(p.split & 1) &&
        p.ldc == 32 && p.rep == 1 && !p.addend) {
        hipLaunchKernelGGL(conv_stem3x3_kernel<32>, dim3((unsigned)((p.M + 255) / 256)), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    if (p.rep != 1 || p.addend) return hipErrorInvalidValue;              // the de-duplicated forms are implicit-GEMM only
    if ((p.split & 2) && (p.N & 3)) return hipErrorInvalidValue;
    const size_t wbytes = (size_t)p.ksize * p.ksize * (p.C0 + p.C1) * p.N * sizeof(float);
    const int w_in_lds = wbytes <= 48 * 1024;
    const int64_t total = (int64_t)p.M * ((p.N + 7) >> 3);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)blocks), dim3(256), w_in_lds ? wbytes : 0, st, p, w_in_lds);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// A view as a tensor (STEP_GATHER of byolo_api.hip): dst[m][c] = the element the convolution loader would have read --
// source 0 or 1 by channel, nearest x2 upsampling as (y >> 1, x >> 1), the T-fold sample tile as sample / T.
// And the residual add that does not ride in a convolution's epilogue.  Neither is on the reference models' path.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void view_gather_kernel(const ConvParams p) {
    const int C = p.C0 + p.C1, hw = p.Hout * p.Wout;
    const int64_t total = (int64_t)p.M * C;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (int64_t)gridDim.x * blockDim.x) {
        const int64_t m = gid / C;
        const int c = (int)(gid - m * C);
        const int s = (int)(m / hw), rem = (int)(m - (int64_t)s * hw);
        const int y = rem / p.Wout, x = rem - y * p.Wout;
        float v;
        if (c < p.C0) v = p.src0[(((size_t)(s / p.sdiv0) * p.Hs0 + (y >> p.sh0)) * p.Ws0 + (x >> p.sh0)) * p.C0 + c];
        else v = p.src1[(((size_t)(s / p.sdiv1) * p.Hs1 + (y >> p.sh1)) * p.Ws1 + (x >> p.sh1)) * p.C1 + (c - p.C0)];
        p.dst[gid] = v;
    }
}
// STEP_FINISH (byolo_kernels.h FinishParams): thread = (source pixel, 4-channel group) -> its 2 x 2 output pixels: one load of the
// low-resolution accumulators, four loads of the per-image partial sums and four 16-byte stores in flight per thread (HBM-bound)
template <int MODE>
__global__ __launch_bounds__(256) void finish_upsampled_kernel(const FinishParams p) {
    const uint32_t n4 = (uint32_t)p.N >> 2, W = (uint32_t)p.W, hw = (uint32_t)(p.H * p.W), lw = W >> 1, lhw = hw >> 2;
    const uint64_t total = (uint64_t)p.S * lhw * n4;
    const bool do_drop = p.flags & EPI_DROPOUT;
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
    float vmax = 0.f;
    for (uint64_t gid = (uint64_t)blockIdx.x * 256u + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * 256u) {
        const uint32_t lm = (uint32_t)(gid / n4), g = (uint32_t)(gid - (uint64_t)lm * n4);        // (rows < 2^31: check_run)
        const uint32_t s = fdiv(lm, p.d_hw), lr = lm - s * lhw;                                   // d_hw divides by lhw, d_w by lw
        const uint32_t ly = fdiv(lr, p.d_w), lx = lr - ly * lw;
        const f32x4 lo4 = *reinterpret_cast<const f32x4*>(p.low + (size_t)lm * p.N + 4 * g);
        const size_t part_img = (size_t)fdiv(s, p.d_T) * hw;
        uint32_t r[4];
        f32x4 a4[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            r[o] = (2 * ly + (uint32_t)(o >> 1)) * W + 2 * lx + (uint32_t)(o & 1);
            a4[o] = lo4;
            if (p.part) a4[o] += *reinterpret_cast<const f32x4*>(p.part + (part_img + r[o]) * p.N + 4 * g);
        }
        f32x4 sc4, sf4;
        if constexpr (MODE != 0) { sc4 = *reinterpret_cast<const f32x4*>(p.scale + 4 * g); sf4 = *reinterpret_cast<const f32x4*>(p.shift + 4 * g); }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            const size_t m = (size_t)s * hw + r[o];
            float* d = p.dst + m * p.N + 4 * g;
            if constexpr (MODE == 0) { *reinterpret_cast<f32x4*>(d) = a4[o]; continue; }
            bool keep[4] = {true, true, true, true};
            if (do_drop) {
                const uint64_t idx = p.idx_base + (uint64_t)m * (uint64_t)p.N + 4u * g;
                const epi::DropRow drow(idx, p.k1);
                if (p.mask_bits) {                                      // injected masks (conv_igemm.hip finish_tile)
                    const uint32_t el = drow.el_lo();
                    const uint32_t w = p.mask_bits[el >> 5] >> (el & 31u);
#pragma unroll
                    for (int q = 0; q < 4; ++q) keep[q] = (w >> q) & 1u;
                } else epi::keep4(drow, 0, p.k0, p.thr, keep);
            }
            const f32x4 v = epi::bn_act4(a4[o], sc4, sf4, keep, slope);
            if constexpr (MODE == 2) { vmax = epi::absmax4(vmax, v); st4(d, epi::split_encode4(v)); }
            else st4(d, v);
        }
    }
    if constexpr (MODE == 2) {
        if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
    }
}
hipError_t launch_finish_upsampled(const FinishParams& p, hipStream_t st) {
    if ((p.N & 3) || (p.H & 1) || (p.W & 1) || p.S < 1 || p.T < 1 || !p.low || !p.dst) return hipErrorInvalidValue;
    const uint64_t total = (uint64_t)p.S * (p.H >> 1) * (p.W >> 1) * (p.N >> 2);
    const unsigned grid = (unsigned)std::min<uint64_t>((total + 255) / 256, 256u * 64u);
    if (p.mode == 0) hipLaunchKernelGGL(finish_upsampled_kernel<0>, dim3(grid), dim3(256), 0, st, p);
    else if (p.mode == 1) hipLaunchKernelGGL(finish_upsampled_kernel<1>, dim3(grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(finish_upsampled_kernel<2>, dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void tensor_add_kernel(const float* a, const float* b, float* d, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = a[i] + b[i];
}
__global__ __launch_bounds__(256) void tensor_add_split_kernel(const f32x4* a, const f32x4* b, f32x4* d, int64_t n4, unsigned* status, int layer) {
    float vmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = epi::split_decode4(a[i]) + epi::split_decode4(b[i]);
        vmax = epi::absmax4(vmax, v);
        d[i] = epi::split_encode4(v);
    }
    if (status && vmax >= 65520.f) { atomicOr(status, 1u); atomicMin(status + 1, (unsigned)layer); }
}
__global__ __launch_bounds__(256) void f32_to_split_kernel(const f32x4* s, f32x4* d, int64_t n4, float mul, unsigned* status) {
    float vmax = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const f32x4 v = s[i] * mul;
        vmax = epi::absmax4(vmax, v);
        d[i] = epi::split_encode4(v);
    }
    if (status && vmax >= 65520.f) { atomicOr(status, 1u); atomicMin(status + 1, 0u); }   // the image itself: reported as layer 0's input
}
__global__ __launch_bounds__(256) void split_to_f32_kernel(const f32x4* s, f32x4* d, int64_t n4, float mul) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = epi::split_decode4(s[i]) * mul;
}
// tf.image.convert_image_dtype(uint8 -> float32) of decode_img (lib_yolo/dataset_utils.py:6-11): float(u8) * (1 / 255) in fp32,
// bit for bit what the host's `astype(float32) * float32(1 / 255)` gives; 4 pixels-channels per thread (one dword in, 16 bytes out)
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const uint32_t* s, f32x4* d, int64_t n4) {
    const float k = 1.0f / 255.0f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t w = __builtin_nontemporal_load(s + i);
        f32x4 v;
        v[0] = (float)(w & 0xFFu) * k; v[1] = (float)((w >> 8) & 0xFFu) * k; v[2] = (float)((w >> 16) & 0xFFu) * k; v[3] = (float)(w >> 24) * k;
        d[i] = v;
    }
}
__global__ void u8_to_f32_tail_kernel(const uint8_t* s, float* d, int64_t lo, int64_t n) {
    const int64_t i = lo + threadIdx.x;
    if (i < n) d[i] = (float)s[i] * (1.0f / 255.0f);
}
static unsigned grid_for(int64_t n) { int64_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 256 * 32 ? 256 * 32 : b)); }
hipError_t launch_view_gather(const ConvParams& p, hipStream_t st) {
    hipLaunchKernelGGL(view_gather_kernel, dim3(grid_for((int64_t)p.M * (p.C0 + p.C1))), dim3(256), 0, st, p);
    return hipGetLastError();
}
hipError_t launch_tensor_add(const float* a, const float* b, float* dst, int64_t n, bool split, hipStream_t st, unsigned* status, int layer) {
    if (split) {
        if (n & 3) return hipErrorInvalidValue;
        hipLaunchKernelGGL(tensor_add_split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(a),
                           reinterpret_cast<const f32x4*>(b), reinterpret_cast<f32x4*>(dst), n / 4, status, layer);
    } else hipLaunchKernelGGL(tensor_add_kernel, dim3(grid_for(n)), dim3(256), 0, st, a, b, dst, n);
    return hipGetLastError();
}
hipError_t launch_f32_to_split(const float* src, float* dst, int64_t n, float mul, hipStream_t st, unsigned* status) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(f32_to_split_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n / 4, mul, status);
    return hipGetLastError();
}
// The split-K tickets / unit claims of a forward, zeroed by a KERNEL of the library (not hipMemsetAsync): inside a captured launch
// graph the runtime's memset node went wrong after ~8 192 graph operations of a process holding three or more executable graphs --
// the ticket words then held garbage and every split-K / stream-K launch of a replayed forward reduced slabs nobody had written
// (tools/graph_stress.py, profiles/r6_small_configs.md: onset at replay 274 with three handles, 206 with four, never with eager
// launches or with one or two graphs).  A kernel node has no such history.
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* p, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = 0u;
}
hipError_t launch_zero_words(void* p, int64_t n_words, hipStream_t st) {
    if (n_words > 0) hipLaunchKernelGGL(zero_words_kernel, dim3(grid_for(n_words)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), n_words);
    return hipGetLastError();
}
hipError_t launch_u8_to_f32(const uint8_t* src, float* dst, int64_t n, hipStream_t st) {
    const int64_t n4 = n / 4;
    if (n4) hipLaunchKernelGGL(u8_to_f32_kernel, dim3(grid_for(n4)), dim3(256), 0, st, reinterpret_cast<const uint32_t*>(src),
                               reinterpret_cast<f32x4*>(dst), n4);
    if (n & 3) hipLaunchKernelGGL(u8_to_f32_tail_kernel, dim3(1), dim3(4), 0, st, src, dst, n4 * 4, n);
    return hipGetLastError();
}
hipError_t launch_split_to_f32(const float* src, float* dst, int64_t n, float mul, hipStream_t st) {
    if (n & 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL(split_to_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, st, reinterpret_cast<const f32x4*>(src),
                       reinterpret_cast<f32x4*>(dst), n / 4, mul);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// calibration helpers (byolo_calibrate_bn): per-channel batch statistics, device-side BN fold,
// in-place BN + leaky [+ residual].  Not on the inference hot path.
// ---------------------------------------------------------------------------------------------
__global__ void channel_stats_partial(const float* x, int64_t M, int C, double* tmp /*[blocks][2][C]*/) {
    // block handles a strided set of rows; thread c-strided over channels
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        double s = 0.0, q = 0.0;
        for (int64_t m = blockIdx.x; m < M; m += gridDim.x) {
            const double v = x[m * C + c];
            s += v; q += v * v;
        }
        tmp[((size_t)blockIdx.x * 2 + 0) * C + c] = s;
        tmp[((size_t)blockIdx.x * 2 + 1) * C + c] = q;
    }
}
__global__ void channel_stats_final(const double* tmp, int blocks, int64_t M, int C, float* mean, float* var) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = 0; b < blocks; ++b) { s += tmp[((size_t)b * 2) * C + c]; q += tmp[((size_t)b * 2 + 1) * C + c]; }
    const double mu = s / (double)M;
    double v = q / (double)M - mu * mu;
    if (v < 0) v = 0;
    mean[c] = (float)mu; var[c] = (float)v;
}
static constexpr int STATS_BLOCKS = 1024;
hipError_t launch_channel_stats(const float* x, int64_t M, int C, float* d_mean, float* d_var, double* d_tmp,
                                hipStream_t st) {
    hipLaunchKernelGGL(channel_stats_partial, dim3(STATS_BLOCKS), dim3(256), 0, st, x, M, C, d_tmp);
    hipLaunchKernelGGL(channel_stats_final, dim3((C + 255) / 256), dim3(256), 0, st, d_tmp, STATS_BLOCKS, M, C,
                       d_mean, d_var);
    return hipGetLastError();
}

__global__ void fold_bn_kernel(const float* g, const float* b, const float* m, const float* v, float eps,
                               float* scale, float* shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float inv = g[c] * (1.0f / sqrtf(v[c] + eps));
    scale[c] = inv;
    shift[c] = b[c] - m[c] * inv;
}
hipError_t launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C, hipStream_t st) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3((C + 255) / 256), dim3(256), 0, st, gamma, beta, mean, var, eps, scale,
                       shift, C);
    return hipGetLastError();
}

__global__ void bn_act_inplace_kernel(float* x, int64_t total, int C, const float* scale, const float* shift,
                                      const float* residual, int leaky) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        float v = x[i] * scale[c] + shift[c];
        if (leaky) v = fmaxf(v, 0.1f * v);
        if (residual) v += residual[i];
        x[i] = v;
    }
}
// split precision: fp32 in, [4 hi | 4 lo] groups out (in place, one group per thread); the residual is a split-f16 tensor
__global__ void bn_act_inplace_split_kernel(f32x4* x, int64_t total4, int C4, const float* scale, const float* shift,
                                            const f32x4* residual, int leaky) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        f32x4 v = x[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q] = v[q] * scale[c + q] + shift[c + q];
            if (leaky) v[q] = fmaxf(v[q], 0.1f * v[q]);
        }
        if (residual) v += epi::split_decode4(residual[i]);
        x[i] = epi::split_encode4(v);
    }
}
hipError_t launch_bn_act_inplace(float* x, int64_t M, int C, const float* scale, const float* shift,
                                 const float* residual, int leaky, bool split, hipStream_t st) {
    const int64_t total = M * C;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (split) {
        if (C & 3) return hipErrorInvalidValue;
        hipLaunchKernelGGL(bn_act_inplace_split_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<f32x4*>(x), total / 4,
                           C / 4, scale, shift, reinterpret_cast<const f32x4*>(residual), leaky);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(bn_act_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, total, C, scale, shift,
                       residual, leaky);
    return hipGetLastError();
}

}  // namespace byk
