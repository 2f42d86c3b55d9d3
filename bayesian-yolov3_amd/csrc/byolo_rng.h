// byolo_rng.h -- the build-defined MC-dropout Bernoulli stream (host + device).
//
// The reference never seeds tf.layers.dropout (lib_yolo/layers.py:521-524), so its masks are
// irreproducible; the build defines them as a pure function of
//   (seed, dropout-layer ordinal, NHWC element index of the dropout input [S,h,w,cout])
// keep <=> mix32(mix32(lo32(idx) + k0) ^ (hi32(idx) + k1)) < floor((1-p) * 2^32)
// with (k0, k1) = layer_keys(seed, ordinal) computed once per layer on the host.
// oracle/rng.py restates this bit-exactly in numpy.  Each wavefront evaluates the hash for the
// 64x16 accumulator elements it owns, in registers, inside the conv epilogue.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BYOLO_HD __host__ __device__ __forceinline__
#else
#define BYOLO_HD inline
#endif

BYOLO_HD uint32_t byolo_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x21F0AAADu;
    x ^= x >> 15; x *= 0x735A2D97u;
    x ^= x >> 15;
    return x;
}

struct byolo_drop_keys { uint32_t k0, k1, thr; };

inline byolo_drop_keys byolo_layer_keys(uint64_t seed, uint32_t layer, double drop_prob) {
    byolo_drop_keys k;
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    k.k0 = byolo_mix32(lo ^ (0x9E3779B9u * (layer + 1u)));
    k.k1 = byolo_mix32(hi + k.k0 + layer);
    k.thr = (uint32_t)((1.0 - drop_prob) * 4294967296.0);
    return k;
}

BYOLO_HD bool byolo_keep(uint64_t idx, uint32_t k0, uint32_t k1, uint32_t thr) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    return byolo_mix32(byolo_mix32(lo + k0) ^ (hi + k1)) < thr;
}
