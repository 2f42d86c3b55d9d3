// epilogue.h -- the ONE definition of the convolution epilogue's arithmetic on a group of 4 consecutive output
// channels of one pixel (lib_yolo/layers.py:560-574: conv -> dropout -> BN -> leaky, dropout BETWEEN conv and BN):
//
//     v = leaky( keep * (acc * scale) + shift )          scale = gamma * rsqrt(var + eps)  [/ (1 - p) with the masks on]
//                                                         shift = beta - mean * gamma * rsqrt(var + eps)   (not masked)
//
// used by every kernel that finishes a convolution: conv_igemm.hip, gemm_stream.hip (1x1 convolutions), wino_fused.hip
// and winograd.hip (output transform).  Epilogue instructions are paid in matrix-pipe time (mfma_pipe.h), so the forms
// here are the cheapest measured: mask and scale as ONE select feeding ONE fused multiply-add; leaky as max(x, slope*x)
// with slope = 1 for linear layers; the dropout mask of the group from one hash + one derived word (byolo_rng.h) whose key word
// of the index's high half is prepared once per pixel by the caller.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "byolo_rng.h"

namespace byk {
namespace epi {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Dropout position of one pixel's channel run: group-of-four index of its first element and the key word of the index's high
// half.  `idx` = NHWC element index of the run's first channel in the dropout input [S, h, w, cout] (a multiple of 4).
// (`el_lo()`: the element index's low word, which the injected-mask paths turn into a bit position.)
struct DropRow {
    uint32_t g4_lo, k1h;
    __device__ __forceinline__ DropRow(uint64_t idx, uint32_t k1) : g4_lo((uint32_t)(idx >> 2)), k1h(k1 + (uint32_t)(idx >> 34) * 0x9E3779B9u) {}
    __device__ __forceinline__ uint32_t el_lo() const { return g4_lo << 2; }
};

// keep bits of the 4-channel group `dn` channels further along the row (dn % 4 == 0): one hash, one derived word, the four
// 16-bit fields compared in place (byolo_rng.h).  A group further along the row may sit past a 2^32 group boundary: the
// high-half key word then moves on by one step.
__device__ __forceinline__ void keep4(const DropRow& r, int dn, uint32_t k0, uint32_t thr, bool (&keep)[4]) {
    const uint32_t g_lo = r.g4_lo + (uint32_t)(dn >> 2);
    const uint32_t k1h = g_lo < r.g4_lo ? r.k1h + 0x9E3779B9u : r.k1h;
    const uint32_t h0 = byolo_pair_hash(g_lo, k0, k1h);
    const uint32_t h1 = byolo_next_word(h0);
    const uint32_t t = thr << 16;                                      // thr <= 65535 (byolo_layer_keys): block-uniform, scalar
    keep[0] = (h0 << 16) < t; keep[1] = h0 < t;
    keep[2] = (h1 << 16) < t; keep[3] = h1 < t;
}

// leaky(keep * (a * scale) + shift), slope = 0.1 (leaky) or 1 (linear)
__device__ __forceinline__ f32x4 bn_act4(const f32x4 a, const f32x4 sc, const f32x4 sf, const bool (&keep)[4], float slope) {
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float x = __builtin_fmaf(a[q], keep[q] ? sc[q] : 0.f, sf[q]);      // mask * scale, + shift
        v[q] = fmaxf(x, slope * x);
    }
    return v;
}
// the same with the fused multiply-add and the slope product as packed fp32 operations (v_pk_fma_f32 / v_pk_mul_f32: two values per
// instruction, IEEE per element -- the bits of the scalar form); for the straight-line epilogues, which have the aligned register
// pairs to spare (the kernels that sit at 256 registers keep the scalar form)
__device__ __forceinline__ f32x4 bn_act4_pk(const f32x4 a, const f32x4 sc, const f32x4 sf, const bool (&keep)[4], float slope) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x4 v;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x2 a2 = {a[2 * h], a[2 * h + 1]}, f2 = {sf[2 * h], sf[2 * h + 1]};
        const f32x2 s2 = {keep[2 * h] ? sc[2 * h] : 0.f, keep[2 * h + 1] ? sc[2 * h + 1] : 0.f};
        const f32x2 x2 = __builtin_elementwise_fma(a2, s2, f2);
        const f32x2 y2 = x2 * f32x2{slope, slope};
        v[2 * h] = fmaxf(x2[0], y2[0]); v[2 * h + 1] = fmaxf(x2[1], y2[1]);
    }
    return v;
}


// max(m, |v0|, .., |v3|): two v_max3_f32 with |.| source modifiers (a NaN operand is ignored, as maxNum does)
__device__ __forceinline__ float absmax4(float m, const f32x4 v) {
    m = fmaxf(fmaxf(m, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
    return fmaxf(fmaxf(m, __builtin_fabsf(v[2])), __builtin_fabsf(v[3]));
}

// Split-f16 storage of 4 consecutive channels (mfma_pipe.h): 16 bytes = [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3],
// hi = RNE_f16(x), lo = RNE_f16(x - hi).  x is in the tensor's pre-scaled domain (byolo_pack.hip folds the powers of two
// into scale / shift); |x| >= 65520 overflows to infinity like any fp16.
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// (hi as a packed pair by v_cvt_pk_f16_f32; lo = RNE_f16(x - hi) by v_fma_mix{lo,hi}_f16: fma(hi read as an f16 source, -1.0, x)
//  computed in f32 -- exact, x - hi has at most 13 significant bits -- and rounded ONCE to f16 into the low / high half of the
//  destination: 6 instructions per 4 values where convert / convert back / subtract / convert / pack took 13, and the same bits --
//  tools/enc_mix_check.hip compares the two forms on 16.7 M random words with infinities, NaNs, the overflow boundary, subnormals)
__device__ __forceinline__ f32x4 split_encode4(const f32x4 v) {
    uint32_t h01, h23, l01, l23;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h01) : "v"(v[0]), "v"(v[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(h23) : "v"(v[2]), "v"(v[3]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l01) : "v"(h01), "v"(v[0]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l01) : "v"(h01), "v"(v[1]));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l23) : "v"(h23), "v"(v[2]));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l23) : "v"(h23), "v"(v[3]));
    return f32x4{__builtin_bit_cast(float, h01), __builtin_bit_cast(float, h23), __builtin_bit_cast(float, l01), __builtin_bit_cast(float, l23)};
}
__device__ __forceinline__ f32x4 split_decode4(const f32x4 w) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f16x4 hi = __builtin_bit_cast(f16x4, f32x2{w[0], w[1]}), lo = __builtin_bit_cast(f16x4, f32x2{w[2], w[3]});
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (float)hi[q] + (float)lo[q];
    return v;
}

}  // namespace epi
}  // namespace byk
