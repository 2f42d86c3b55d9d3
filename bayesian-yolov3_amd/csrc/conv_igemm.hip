// conv_igemm.hip -- K1..K5 of SURVEY.md section 2.1 as ONE fused implicit-GEMM kernel family for
// gfx950 (CDNA4), exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces, per call, the reference's op chain of lib_yolo/layers.py:545-575
//   tf.layers.conv2d(no bias) -> tf.layers.dropout -> tf.layers.batch_normalization -> leaky_relu
// plus, folded into the operand loader / epilogue, layers.py:505-507 (residual add), :578-580
// (nearest x2 upsample), :583-592 (channel concat), :595-597 (T-fold batch tile), :533-537
// (darknet stride-2 padding) and :600-613 (detection conv + bias).
//
// GEMM view: M = S*Hout*Wout output pixels, N = cout, K = ksize^2 * Cin, NHWC activations
// (a K-slice of 32 channels of one tap is 128 contiguous bytes per pixel), weights pre-packed at
// byolo_finalize() as [K/32][Npad][32] so a block's B tile is one contiguous BN*128-byte read.
// Block = WM x WN wave64; block tile BM x BN x 32, each wave owns TM x TN tiles of 32x32 accumulated in
// registers; operands staged global -> VGPR -> LDS (row stride 36 floats: the ds_read_b128 fragment
// reads and the ds_write_b128 staging writes are bank-conflict free), double-buffered.
// The K order inside a 32-slice is permuted (lane-half h of MFMA step j consumes k = 8q+4h+j) so
// that every lane fetches its four A (and B) operands of four MFMA steps with ONE ds_read_b128.
//
// The design rule of the K loop (measured, tools/mfma_peak.hip): a vector-ALU instruction is NOT hidden
// under a v_mfma_f32_32x32x2_f32 -- every one costs the SIMD ~3-4 cycles of matrix-pipe time, with one
// or two waves per SIMD alike (1 VALU per MFMA: 155.5 -> 145.7 TFLOP/s).  Global loads, LDS reads/writes,
// scalar ALU and waits do hide.  So the steady-state loop carries NO vector-ALU instruction at all:
//   * operands are fetched with buffer loads: per-row 32-bit byte offset in a VGPR that changes only
//     when the filter tap changes, the running channel offset in an SGPR (scalar ALU); rows in the zero
//     padding hold an out-of-range offset and the buffer bounds check returns 0 for them;
//   * every LDS address is a loop-invariant VGPR + an immediate (the loop is unrolled x2 so that the
//     double-buffer index is a compile-time constant);
//   * the K-tile sequencing (tap, channel chunk, weight offset) lives in SGPRs.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <type_traits>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "mfma_pipe.h"
#include "epilogue.h"

namespace byk {

using namespace pipe;                         // tile geometry, LDS image, fragment scheme, MFMA group, K-tile schedule: mfma_pipe.h

// Timing ablations of the K loop (build.py --ablate N -> libbyolo_ablN.so, loaded with BYOLO_LIB=...; the
// results are WRONG by construction): 1 no global loads, 2 no LDS staging writes (the loads are still
// waited for), 4 no barrier, 8 no fragment reads.
#ifndef BYOLO_CONV_ABLATE
#define BYOLO_CONV_ABLATE 0
#endif
static constexpr int ABL = BYOLO_CONV_ABLATE;

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Blocks are dispatched round-robin over the 8 XCDs (bid % 8); give each XCD a contiguous
    // range of logical tiles so neighbouring tiles (same A rows / same weights) share its L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// One BM x BN output tile (logical tile index -> (tile_m, tile_n)).
// FAST: one plain source (no concat, no upsample), ksize <= 3 -- every backbone conv and every 3x3 /
//       1x1 head conv after the concat de-duplication.  The ksize^2 tap addresses of a row differ by a
//       block-uniform delta, so a row keeps the tap-(0,0) byte offset + a validity bit per tap, and a tap
//       switch is one add + select per row.  !FAST re-derives the row offsets per (tap, source).
// Split-K slices (ConvParams::ksplit > 1, the tiles of the last, partial wave of a launch): the block
// computes K-tiles [kt_begin, kt_end) of tile `logical`, writes its raw accumulators to slab
// (slice_tile, slice) and draws a ticket; the block that draws the last ticket of the tile sums the slabs
// in slice order (deterministic) and runs the epilogue.
// How a partial tile (K-tiles [kt_begin, kt_end) of a tile) is handed over: the block writes its raw accumulators
// to slab `my_slab` and draws a ticket from counters[counter]; the block that draws ticket nseg - 1 sums the slabs
// of segments k = 0 .. nseg-1 -- slab index base + k * stride (+ first_add for k = 0) -- in that order and runs
// the epilogue.  counter < 0: the block computes the whole tile.
struct TileShare { int counter, my_slab, nseg, base, stride, first_add; };

// The end of a tile, shared by every K loop of this file: split-K hand-off, then the fused epilogue on the accumulators
// (wave (wm, wn) of the WM x WN block owns TM x TN blocks of 32 x 32; lane = pixel li (+32 per block row), 4 groups of 4
// consecutive channels from 4 * lh).
template <int BM, int BN, int WM, int WN, bool SPLIT>
__device__ __forceinline__ void finish_tile(const ConvParams& p, float* smem, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                            const uint32_t tile_m, const uint32_t tile_n, const TileShare sh) {
    constexpr int NT = 64 * WM * WN, TM = BM / WM / 32, TN = BN / WN / 32;
    // injected dropout masks (ConvParams::mask_bits): every build but the fp32 128 x 128 tile, which sits at exactly 256
    // registers and would spill -- byolo_plan.hip plans the 64-wide tile for such a call in the fp32 mode
    constexpr bool INJECT = SPLIT || BN < 128;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const uint32_t hw = (uint32_t)(p.Hout * p.Wout);
    // ---- split-K hand-off (block-uniform): slab write, ticket, ordered reduce by the last arriver ---------
    // Per-XCD L2s are not coherent with each other and a CU's L1 is not refreshed by other CUs' stores, so
    // the slabs travel with sc1 (write-through / system-coherent) stores and loads: once a wave's vmcnt has
    // drained its slab is visible to the whole device, and the reducer's sc1 loads cannot hit a stale line.
    // No release/acquire fence (an agent-scope release = buffer_wbl2 writes back the whole L2; hundreds of
    // slice blocks doing that at the end of a launch cost more than the split saved).  Correct for any
    // placement of a tile's slices on XCDs / CUs.
    if (sh.counter >= 0) {
        constexpr uint32_t SLAB_B = BM * BN * 4;     // bytes; lane-linear image: float4 q of thread t at (q * NT + t) * 16
        constexpr int SC1 = 16;
        const __amdgpu_buffer_rsrc_t s_rsrc = make_rsrc(p.slabs, p.slab_bytes);
        const uint32_t my_off = (uint32_t)sh.my_slab * SLAB_B;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), s_rsrc, tid * 16,
                                                           my_off + ((i * TN + j) * 4 + g) * (NT * 16), SC1);
                }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                         // also: every wave is done with the LDS tiles
        int* flag = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(p.counters + sh.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool is_last = ticket == (unsigned)sh.nseg - 1u;
            *flag = is_last;
            // leave the counter at zero for the next launch that uses it (several launches of one step share them)
            if (is_last) __hip_atomic_store(p.counters + sh.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        const int last = *flag;
        __syncthreads();                         // the flag word is LDS the next tile of this block overwrites
        if (!last) return;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int sl = 0; sl < sh.nseg; ++sl) {       // segment order: the sum does not depend on arrival order
            const uint32_t off = (uint32_t)(sh.base + sl * sh.stride + (sl == 0 ? sh.first_add : 0)) * SLAB_B;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const f32x4 v = buffer_load_x4<SC1>(s_rsrc, tid * 16, off + ((i * TN + j) * 4 + g) * (NT * 16));
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[i][j][4 * g + q] += v[q];
                    }
        }
    }

    // ---- fused epilogue: [+ addend] [dropout mask] * scale, + shift, leaky, [+ residual] ---------------
    // The MFMAs compute the TRANSPOSED tile (srcA = weights, srcB = pixels), so in the 32x32 C/D map
    //   col = lane & 31 -> pixel,  row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) -> channel
    // every lane owns, per 32x32 tile, ONE pixel and 4 groups of 4 CONSECUTIVE channels: NHWC stores
    // (and residual loads) are 16-byte vectors.  The row part of every address (dst, residual, addend,
    // dropout element index) is derived once per row; the (j, g) channel-group part is an immediate.
    const bool do_leaky = p.flags & EPI_LEAKY, do_drop = p.flags & EPI_DROPOUT, do_res = p.flags & EPI_RESIDUAL;
    const float slope = do_leaky ? 0.1f : 1.f;
    const bool split_out = !(p.flags & EPI_F32OUT);            // SPLIT: encode the output (everything but a detection head)
    // T-invariant de-duplication (lowering in byolo_api.hip):
    //   rep > 1     the conv ran once per IMAGE (its input does not depend on the MC sample); only the
    //               dropout mask differs between the T samples, so the epilogue is replayed T times and
    //               writes the T stacked outputs (row m = img*hw + pix  ->  (img*rep + t)*hw + pix);
    //   addend      the T-invariant half of a concat input was convolved once per image into `addend`
    //               (raw accumulators, [B*hw][N]); it joins the accumulator here, before scale / mask.
    const int rep = p.rep;
    const int nb = (int)(tile_n * BN) + wn * TN * 32 + 4 * lh;       // first channel of this lane's (j=0, g=0) group
    // EPI_RAW (the Winograd-domain GEMM): the accumulators are the result; nothing but the 16-byte stores
    if ((p.flags & EPI_RAW) && ((p.N | p.ldc) & 3) == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t m = tile_m * BM + wm * TM * 32 + i * 32 + li;
            if (m >= (uint32_t)p.M) continue;
            float* d = p.dst + (size_t)m * p.ldc + nb;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dn = j * 32 + 8 * g;
                    if (nb + dn >= p.N) continue;
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q];
                    *reinterpret_cast<f32x4*>(d + dn) = v;
                }
        }
        return;
    }
    uint32_t row_m[TM], row_img[TM], row_pix[TM];
    const float* add_row[TM];
    float vmax = 0.f;                                          // SPLIT: largest |value| this lane stores
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        row_m[i] = tile_m * BM + wm * TM * 32 + i * 32 + li;
        const uint32_t mm = row_m[i] < (uint32_t)p.M ? row_m[i] : 0u;
        row_img[i] = fdiv(mm, p.d_hw);
        row_pix[i] = mm - row_img[i] * hw;
        add_row[i] = p.addend ? p.addend + (size_t)(fdiv(row_img[i], p.d_addT) * hw + row_pix[i]) * p.N + nb : nullptr;
    }
    // VEC: N % 4 == 0 and ldc % 4 == 0 (every convolution but the 3*(5+C)-channel detection heads): a lane's
    //      4-channel group is entirely inside or entirely outside N, its element index is a multiple of 4,
    //      so the group is one 16-byte store and its dropout bits are two pair hashes.
    auto epilogue = [&](auto vec_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;
        for (int t = 0; t < rep; ++t) {
            float* dst_row[TM];
            const float* res_row[TM];
            uint64_t idx_row[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const uint32_t mo = rep > 1 ? (row_img[i] * rep + t) * hw + row_pix[i] : row_m[i];
                dst_row[i] = p.dst + (size_t)mo * p.ldc + nb;
                res_row[i] = do_res ? p.residual + (size_t)mo * p.ldc + nb : nullptr;
                idx_row[i] = p.idx_base + (uint64_t)mo * (uint64_t)p.N + (uint64_t)nb;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (row_m[i] >= (uint32_t)p.M) continue;
                const epi::DropRow drow(idx_row[i], p.k1);        // VEC: pair index / high-half key word, once per row
                // VEC: the row's addend (joins before scale) or residual (joins after the activation) values are
                // fetched up front, all TN*4 groups in flight at once -- a load + wait per group serialises the
                // epilogue on memory latency (the staging / fragment registers of the K loop are dead here)
                f32x4 extra[TN * 4];
                const float* extra_row = p.addend ? add_row[i] : res_row[i];
                if constexpr (VEC) {
                    if (extra_row) {
#pragma unroll
                        for (int k = 0; k < TN * 4; ++k) {
                            const int dn = (k >> 2) * 32 + 8 * (k & 3);
                            extra[k] = nb + dn < p.N ? *reinterpret_cast<const f32x4*>(extra_row + dn) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int dn = j * 32 + 8 * g;                        // channel offset of the group from nb
                        const int n0 = nb + dn;                               // 4 channels n0 .. n0+3
                        if (n0 >= p.N) continue;
                        // arrays are padded to Npad; with the masks on, scale already holds 1 / (1 - p) (host)
                        const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.scale + n0);
                        const f32x4 sf4 = *reinterpret_cast<const f32x4*>(p.shift + n0);
                        f32x4 a4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) a4[q] = acc[i][j][4 * g + q];
                        if (p.addend) {
                            if constexpr (VEC) a4 += extra[j * 4 + g];
                            else {
                                const float* ad = add_row[i] + dn;
#pragma unroll
                                for (int q = 0; q < 4; ++q) if (n0 + q < p.N) a4[q] += ad[q];
                            }
                        }
                        const uint64_t idx0 = idx_row[i] + (uint64_t)dn;
                        bool keep[4] = {true, true, true, true};
                        if (do_drop) {
                            if constexpr (VEC) {
                                if (INJECT && p.mask_bits) {                            // injected masks: bit i of the layer = element i of this call's tensor
                                    // (idx_base is 0 on such a call, the tensor has < 2^32 elements, cout % 4 == 0: byolo_forward checks;
                                    //  element index % 4 == 0: the group's bits sit in one word)
                                    const uint32_t el = drow.el_lo() + (uint32_t)dn;
                                    const uint32_t w = p.mask_bits[el >> 5] >> (el & 31u);
#pragma unroll
                                    for (int q = 0; q < 4; ++q) keep[q] = (w >> q) & 1u;
                                } else epi::keep4(drow, dn, p.k0, p.thr, keep);
                            } else {
#pragma unroll
                                for (int q = 0; q < 4; ++q) keep[q] = byolo_keep(idx0 + q, p.k0, p.k1, p.thr);
                            }
                        }
                        f32x4 v = epi::bn_act4(a4, sc4, sf4, keep, slope);    // slope = 0.1 (leaky) or 1 (linear)
                        float* d = dst_row[i] + dn;
                        if constexpr (VEC) {
                            // (a launch with BOTH an addend and a residual -- a de-duplicated concat convolution
                            //  followed by a residual add: no reference model has one -- prefetched the addend)
                            if (do_res) {
                                const f32x4 r4 = p.addend ? *reinterpret_cast<const f32x4*>(res_row[i] + dn) : extra[j * 4 + g];
                                v += SPLIT ? epi::split_decode4(r4) : r4;
                            }
                            if constexpr (SPLIT) vmax = epi::absmax4(vmax, v);      // range check of the hi/lo encoding (below)
                            *reinterpret_cast<f32x4*>(d) = (SPLIT && split_out) ? epi::split_encode4(v) : v;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                if (n0 + q < p.N) d[q] = do_res ? v[q] + res_row[i][dn + q] : v[q];
                        }
                    }
                }
            }
        }
    };
    if (((p.N | p.ldc) & 3) == 0) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    // Split-f16 storage ends at |x| = 65504 (epilogue.h): an output beyond it would be stored as infinity where the reference's
    // float32 tensor (lib_yolo/layers.py:550-574) holds a number.  The lane's largest stored magnitude costs one v_max3 per two
    // values; a hit raises the handle's sticky status word, which byolo_forward / byolo_status turn into BYOLO_ERR_RANGE.
    if constexpr (SPLIT) {
        if (split_out && p.status && vmax >= 65520.f) {
            atomicOr(p.status, 1u);
            atomicMin(p.status + 1, (unsigned)p.layer_idx);
        }
    }
}

// The epilogue of the PLAIN case (ConvParams::plain, set by the launcher): a split-f16 launch without addend, residual, T-replay,
// raw / fp32 output or injected masks, cout % 32 == 0, 16-byte rows, a whole tile -- which is every large launch of the reference's
// models (the head convolutions, most of the backbone).  finish_tile decides all of that per channel group at run time: its
// unrolled body carries a dozen block-uniform branches, scalar spills read back through v_readlane and hazard no-ops per group
// -- 24 vector instructions per output value measured (PMC, round 4) where the arithmetic needs ~14, and a timing ablation without
// the epilogue's arithmetic ran the 76x76 shared-tap launches 29 % faster: vector instructions are paid in matrix-pipe time
// (mfma_pipe.h), the epilogue is the largest single cost after the MFMAs themselves.  Here: ONE decision per tile (DROP is a
// template parameter), column block outermost so that one block's scale / shift vectors are live at a time, no branches inside.
// Same arithmetic in the same order per value (epilogue.h): the same bits.
// RES: + the residual (a split-f16 tensor shaped like the output: the darknet blocks' `inputs + shortcut`, lib_yolo/layers.py:505-507),
// after the activation; a column block's residual groups of all rows are fetched before its arithmetic starts.
template <int BM, int BN, int WM, int WN, bool DROP, bool RES = false>
__device__ __forceinline__ void finish_plain(const ConvParams& p, f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const uint32_t tile_m, const uint32_t tile_n) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
    const int nb = (int)(tile_n * BN) + wn * TN * 32 + 4 * lh;
    float vmax = 0.f;
    if constexpr ((ABL & 64) != 0) {              // timing ablation: no epilogue arithmetic, one 16-byte store per accumulator block
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t m = tile_m * BM + wm * TM * 32 + i * 32 + li;
            if (m >= (uint32_t)p.M) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = acc[i][j][q] + acc[i][j][4 + q] + acc[i][j][8 + q] + acc[i][j][12 + q];
                *reinterpret_cast<f32x4*>(p.dst + (size_t)m * p.ldc + nb + j * 32) = a4;
            }
        }
        return;
    }
    uint32_t row_m[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) row_m[i] = tile_m * BM + wm * TM * 32 + i * 32 + li;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        if (nb - 4 * lh + j * 32 >= p.N) continue;                      // (wave-uniform) a column block of the padding
        f32x4 sc4[4], sf4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(p.scale + nb + j * 32 + 8 * g);
            sf4[g] = *reinterpret_cast<const f32x4*>(p.shift + nb + j * 32 + 8 * g);
        }
        f32x4 res[RES ? TM : 1][4];
        if constexpr (RES) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const float* r = p.residual + (size_t)(row_m[i] < (uint32_t)p.M ? row_m[i] : 0u) * p.ldc + nb + j * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) res[i][g] = *reinterpret_cast<const f32x4*>(r + 8 * g);
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            if (row_m[i] >= (uint32_t)p.M) continue;
            const uint64_t idx_row = p.idx_base + (uint64_t)row_m[i] * (uint64_t)p.N + (uint64_t)nb;
            const epi::DropRow drow(idx_row, p.k1);
            float* d = p.dst + (size_t)row_m[i] * p.ldc + nb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dn = j * 32 + 8 * g;
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = acc[i][j][4 * g + q];
                bool keep[4] = {true, true, true, true};
                if constexpr (DROP) epi::keep4(drow, dn, p.k0, p.thr, keep);
                f32x4 v = epi::bn_act4_pk(a4, sc4[g], sf4[g], keep, slope);
                if constexpr (RES) v += epi::split_decode4(res[i][g]);
                vmax = epi::absmax4(vmax, v);
                *reinterpret_cast<f32x4*>(d + dn) = epi::split_encode4(v);
            }
        }
    }
    if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
}
// the end of a tile of a split-f16 launch: the straight-line epilogue where the launch and the tile allow it
// (WITH_RES = false: a kernel that never sees a residual -- the 8-wave tile of the head layers -- does not carry that instantiation,
//  which cost it its last registers)
template <int BM, int BN, int WM, int WN, bool WITH_RES = true>
__device__ __forceinline__ void finish_split(const ConvParams& p, float* smem, f32x16 (&acc)[BM / WM / 32][BN / WN / 32],
                                             const uint32_t tile_m, const uint32_t tile_n, const TileShare sh) {
    if (p.plain && sh.counter < 0 && (WITH_RES || !(p.flags & EPI_RESIDUAL))) {
        if constexpr (WITH_RES) {
            if (p.flags & EPI_RESIDUAL) { finish_plain<BM, BN, WM, WN, false, true>(p, acc, tile_m, tile_n); return; }      // (never with dropout: conv_epilogue_is_plain)
        }
        if (p.flags & EPI_DROPOUT) finish_plain<BM, BN, WM, WN, true>(p, acc, tile_m, tile_n);
        else finish_plain<BM, BN, WM, WN, false>(p, acc, tile_m, tile_n);
    } else finish_tile<BM, BN, WM, WN, true>(p, smem, acc, tile_m, tile_n, sh);
}

// SPLIT: split-f16 operands and activations (mfma_pipe.h): sources, weights and residual are [4 hi | 4 lo] groups, three
// fp16 MFMAs per product into the same fp32 accumulators; the output is encoded the same way unless EPI_F32OUT / EPI_RAW.
template <int BM, int BN, int WM, int WN, bool FAST, bool SPLIT>
__device__ __forceinline__ void conv_tile(const ConvParams& p, float* smem, const int logical, const int kt_begin,
                                          const int kt_end, const TileShare sh) {
    using BT = std::conditional_t<SPLIT, SplitTile<BM, BN, WM, WN>, BlockTile<BM, BN, WM, WN>>;
    constexpr int NT = BT::NT, TM = BT::TM, TN = BT::TN, A_LD = BT::A_LD;
    constexpr int B_LD = SPLIT ? 1 : BN * 8 / NT;            // SPLIT: the weight operand does not pass through LDS
    const BT bt(smem);
    const int tid = bt.tid;
    const uint32_t n_tiles = (uint32_t)p.Npad / BN;
    const uint32_t tile_m = fdiv((uint32_t)logical, p.d_ntiles), tile_n = (uint32_t)logical - tile_m * n_tiles;

    // ---- per-thread A-row bookkeeping (4 rows at BM = 128, 256 threads) -----------------------------
    // Each thread stages the same A_LD rows of every K-tile.  Row state = output pixel (sample, oy, ox).
    const int a_q = bt.a_q, a_r = bt.a_r;
    const uint32_t hw = (uint32_t)(p.Hout * p.Wout);
    uint32_t a_voff[A_LD];                       // byte offset of the row for the current (tap, source)
    uint32_t a_off00[A_LD], a_mask[A_LD];        // FAST: tap (0,0) offset, validity bit per tap
    int a_iy0[A_LD], a_ix0[A_LD];                // !FAST: input coordinate of tap (0,0)
    uint32_t a_s0[A_LD], a_s1[A_LD];             //        sample index in each source
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const uint32_t m = tile_m * BM + a_r + (NT / 8) * j;
        const bool row_ok = m < (uint32_t)p.M;
        const uint32_t mm = row_ok ? m : 0u;
        const uint32_t s = fdiv(mm, p.d_hw), rem = mm - s * hw;
        const uint32_t oy = fdiv(rem, p.d_wout), ox = rem - oy * (uint32_t)p.Wout;
        const int iy0 = (int)oy * p.stride - p.pad, ix0 = (int)ox * p.stride - p.pad;
        if constexpr (FAST) {
            unsigned ym = 0, xm = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                ym |= (t < p.ksize && (unsigned)(iy0 + t) < (unsigned)p.Hin) ? (1u << t) : 0u;
                xm |= (t < p.ksize && (unsigned)(ix0 + t) < (unsigned)p.Win) ? (1u << t) : 0u;
            }
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) mk |= ((ym >> t) & 1u) ? xm << (t * p.ksize) : 0u;
            a_mask[j] = row_ok ? mk : 0u;
            // 32-bit wrap-around arithmetic: the sum for a VALID tap is the true offset (< 2^32)
            const uint32_t s0 = fdiv(s, p.d_sdiv0);
            a_off00[j] = ((((s0 * (uint32_t)p.Hs0 + (uint32_t)iy0) * (uint32_t)p.Ws0 + (uint32_t)ix0) * (uint32_t)p.C0) + a_q * 4) * 4u;
        } else {
            a_iy0[j] = row_ok ? iy0 : -(1 << 28);    // every tap out of bounds -> reads 0
            a_ix0[j] = ix0;
            a_s0[j] = fdiv(s, p.d_sdiv0);
            a_s1[j] = fdiv(s, p.d_sdiv1);
        }
    }

    // ---- K-tile sequencing: block-uniform, scalar registers ------------------------------------------
    int ld_tap = (int)fdiv((uint32_t)kt_begin, p.d_cin);     // (channel chunk, tap) of the NEXT tile to load
    int ld_chunk = kt_begin - ld_tap * p.cin_tiles;
    int ld_ky = (int)fdiv((uint32_t)ld_tap, p.d_ks), ld_kx = ld_tap - ld_ky * p.ksize;
    bool ld_first = true;                                   // the first tile sets the row offsets whatever its chunk
    uint32_t a_soff = 0;                                    // byte offset of that tile's channels inside a row
    const float* a_base = p.src0;
    uint32_t a_bytes = p.src0_bytes;
    const uint32_t w_step = (uint32_t)p.Npad * BK * 4;      // (SPLIT: = Npad / 32 blocks of SPLIT_WBLOCK bytes per K-tile)
    uint32_t w_soff = (uint32_t)kt_begin * w_step;          // byte offset of the next weight tile
    if constexpr (SPLIT) w_soff += tile_n * (BN / 32) * SPLIT_WBLOCK;
    if (p.wino_rows) w_soff += fdiv(tile_m * BM, p.d_wino) * p.wino_wstride;   // Winograd: row block xi has its own matrix
    const uint32_t b_voff = (tile_n * BN * BK + (uint32_t)tid * 4) * 4;

    auto next_tile = [&]() {
        if constexpr (FAST) {
            if (ld_chunk == 0 || ld_first) {
                const uint32_t delta = (uint32_t)((ld_ky * p.Ws0 + ld_kx) * p.C0) * 4u;
                const uint32_t bit = 1u << ld_tap;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) a_voff[j] = (a_mask[j] & bit) ? a_off00[j] + delta : CONV_OOB_OFFSET;
            }
            a_soff = (uint32_t)ld_chunk * (BK * 4);
        } else {
            const int cc = ld_chunk * BK;
            const bool second = cc >= p.C0;
            if (ld_chunk == 0 || cc == p.C0 || ld_first) {
                const uint32_t C = second ? p.C1 : p.C0, Hs = second ? p.Hs1 : p.Hs0, Ws = second ? p.Ws1 : p.Ws0;
                const int sh = second ? p.sh1 : p.sh0;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) {
                    const int iy = a_iy0[j] + ld_ky, ix = a_ix0[j] + ld_kx;
                    const bool ok = (unsigned)iy < (unsigned)p.Hin && (unsigned)ix < (unsigned)p.Win;
                    const uint32_t s = second ? a_s1[j] : a_s0[j];
                    const uint32_t off = ((((s * Hs + (uint32_t)(iy >> sh)) * Ws + (uint32_t)(ix >> sh)) * C) + a_q * 4) * 4u;
                    a_voff[j] = ok ? off : CONV_OOB_OFFSET;
                }
                a_base = second ? p.src1 : p.src0;
                a_bytes = second ? p.src1_bytes : p.src0_bytes;
            }
            a_soff = (uint32_t)(second ? cc - p.C0 : cc) * 4u;
        }
        ld_first = false;
        if (++ld_chunk == p.cin_tiles) {
            ld_chunk = 0; ++ld_tap;
            if (++ld_kx == p.ksize) { ld_kx = 0; ++ld_ky; }
        }
    };

    // DEEP (split launches on the 64- and 32-wide tiles: the HBM-latency-bound 1x1 convolutions and detection heads, which
    // have the registers for it): the activations of tile t+3 are fetched while tile t multiplies -- two staging sets, a
    // tile's loads have two K-tiles to arrive instead of one.
    constexpr bool DEEP = SPLIT && BN <= 64;
    f32x4 a_regs[DEEP ? 2 : 1][A_LD], b_reg[B_LD];     // staging registers (SPLIT: b_reg unused)
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wpk, p.w_bytes);
    auto issue_loads_to = [&](auto set_tag) {    // the buffer_load_dwordx4s of the tile set up by next_tile()
        f32x4 (&a_reg)[A_LD] = a_regs[DEEP ? decltype(set_tag)::value : 0];
        const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(a_base, a_bytes);
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
        if constexpr (!SPLIT) {
#pragma unroll
            for (int j = 0; j < B_LD; ++j) b_reg[j] = buffer_load_x4(w_rsrc, b_voff, w_soff + j * (NT * 16));
            w_soff += w_step;
        }
    };
    auto issue_loads = [&]() { issue_loads_to(std::integral_constant<int, 0>{}); };
    auto store_tile_from = [&](auto buf_tag, auto set_tag) {        // a staged K-tile -> LDS buffer BUF
        constexpr int BUF = decltype(buf_tag)::value;
        f32x4 (&a_reg)[A_LD] = a_regs[DEEP ? decltype(set_tag)::value : 0];
        if constexpr ((ABL & 2) != 0) {
#pragma unroll
            for (int j = 0; j < A_LD; ++j) asm volatile("" : : "v"(a_reg[j]));
            if constexpr (!SPLIT) {
#pragma unroll
                for (int j = 0; j < B_LD; ++j) asm volatile("" : : "v"(b_reg[j]));
            }
            return;
        }
        bt.template store_a<BUF>(a_reg);
        if constexpr (!SPLIT) bt.template store_b<BUF>(b_reg);
    };
    auto store_tile = [&](auto buf_tag) { store_tile_from(buf_tag, std::integral_constant<int, 0>{}); };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- software-pipelined K loop (schedule: mfma_pipe.h tile_body) ----------------------------------------------
    // The loop body is branch-free (tail tiles peeled).  The buffer loads of tile t+2 go into the staging registers
    // just freed, in the last MFMA group of tile t: three MFMA groups before the LDS write that waits for them
    // (issuing them in group 0 of tile t+1 measured -0.3 %).
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using yes = std::true_type;
    using no = std::false_type;
    const int KT = kt_end - kt_begin;            // K-tiles of this block
    // SPLIT: weight fragments of tiles t, t+1 in the two register sets (set = t & 1), fetched one K-tile ahead
    f16x8 bfr[SPLIT ? 2 : 1][2][TN][2];
    auto issue_b = [&](auto set_tag) {
        if constexpr (SPLIT) { bt.load_b(bfr[decltype(set_tag)::value], w_rsrc, w_soff); w_soff += w_step; }
    };
    next_tile();
    issue_loads();
    issue_b(std::integral_constant<int, 0>{});
    store_tile(std::integral_constant<int, 0>{});
    if constexpr (DEEP) {                         // tiles 1 and 2 wait in the two staging sets (past the end of K: zeros or unused bytes)
        next_tile(); issue_loads_to(std::integral_constant<int, 1>{});
        next_tile(); issue_loads_to(std::integral_constant<int, 0>{});
    } else if (KT > 1) { next_tile(); issue_loads(); }   // tile 1 waits in the staging registers
    __syncthreads();
    // tile t lives in LDS buffer BUF = t & 1.  HN: tile t+1 exists (stage it);  LD: tile t+2 exists (fetch it)
    auto run_tiles = [&](auto&& tile_body) {
        int kt = 0;
        if constexpr (DEEP) {                             // every tile with a successor stages t+1 and fetches t+3
            for (; kt + 2 < KT; kt += 2) {
                next_tile(); tile_body(c0{}, yes{}, yes{});
                next_tile(); tile_body(c1{}, yes{}, yes{});
            }
            if (kt + 1 < KT) { next_tile(); tile_body(c0{}, yes{}, yes{}); tile_body(c1{}, no{}, no{}); }
            else if (kt < KT) tile_body(c0{}, no{}, no{});
            return;
        }
        for (; kt + 3 < KT; kt += 2) {                      // tiles kt, kt+1: both stage t+1 and fetch t+2
            next_tile(); tile_body(c0{}, yes{}, yes{});
            next_tile(); tile_body(c1{}, yes{}, yes{});
        }
        auto tail = [&](auto buf_tag, const int t) {          // the last 1..3 tiles (block-uniform branches)
            if (t >= KT) return;
            if (t + 2 < KT) { next_tile(); tile_body(buf_tag, yes{}, yes{}); }
            else if (t + 1 < KT) tile_body(buf_tag, yes{}, no{});
            else tile_body(buf_tag, no{}, no{});
        };
        tail(c0{}, kt); tail(c1{}, kt + 1); tail(c0{}, kt + 2);
    };
    if constexpr (SPLIT) {
        f16x8 af0[TM][2], af1[TM][2];
        bt.template read_frags<0, 0>(af0);
        run_tiles([&](auto buf_tag, auto has_next_tag, auto load_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            constexpr bool HN = decltype(has_next_tag)::value, LD = !(ABL & 1), LD3 = decltype(load_tag)::value && LD;
            pipe::tile_body_split<BUF, HN, (HN && LD) ? BT::NBF : 0, LD3 ? A_LD : 0, (ABL & 2) ? 0 : A_LD, ABL>(
                bt, acc, af0, af1, bfr[BUF], [&] { issue_b(std::integral_constant<int, BUF ^ 1>{}); },
                [&] { issue_loads_to(std::integral_constant<int, BUF ^ 1>{}); },             // DEEP: tile t+3 into the set tile t+1 has just left
                [&] { store_tile_from(std::integral_constant<int, BUF ^ 1>{}, std::integral_constant<int, BUF ^ 1>{}); });
        });
    } else {
        f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
        bt.template read_frags<0, 0>(af0, bf0);
        auto mf = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], int) { mfma_group<TM, TN>(acc, af, bf); };
        auto none = [] {};
        // (s_setprio around this loop was measured: no effect.)
        run_tiles([&](auto buf_tag, auto has_next_tag, auto load_tag) {
            constexpr int BUF = decltype(buf_tag)::value;
            constexpr bool HN = decltype(has_next_tag)::value;
            constexpr bool LD3 = decltype(load_tag)::value && !(ABL & 1);
            constexpr int N_ST = (ABL & 2) ? 0 : BT::NLD;
            pipe::tile_body<BUF, HN, 0, LD3 ? BT::NLD : 0, N_ST, ABL>(
                bt, af0, bf0, af1, bf1, mf, none, issue_loads, [&] { store_tile(std::integral_constant<int, BUF ^ 1>{}); });
        });
    }

    if constexpr (SPLIT) finish_split<BM, BN, WM, WN>(p, smem, acc, tile_m, tile_n, sh);
    else finish_tile<BM, BN, WM, WN, SPLIT>(p, smem, acc, tile_m, tile_n, sh);
}

// Back-to-back fusion on the 8-wave 128 x 256 tile (ConvParams::f_wpk): the accumulators of a finished 3x3 tile -- 128 pixels x
// ALL 256 output channels of the layer, wave w owns channels 32 w .. 32 w + 31 -- go through this layer's epilogue (dropout mask,
// BN, leaky, hi/lo encoding, range check) into LDS as the activation image of a 1x1 convolution: 8 K-tiles of [128 rows][32 hi |
// 32 lo], exactly what the staging stores of conv_tile_p1 would have put there from memory.  A second MFMA pass multiplies them
// with the follower's weights (N2 <= 128 columns: 8 waves as WM2 x WN2 blocks of TM2 x 1), and the follower's epilogue (finish_tile
// on a copy of the parameters) stores ITS output.  The 3x3 layer's own output tensor -- 1.42 GB at config 4's 76x76 layers -- is
// never written or read, and the follower's launch disappears.
template <int WM2, int WN2>
__device__ __forceinline__ void fused_tail(const ConvParams& p, float* smem, f32x16 (&acc)[4][1], const uint32_t tile_m) {
    constexpr int BM = 128, ROWB = LD * 4, A_BUF = BM * ROWB, BN2 = 32 * WN2;
    using BT2 = SplitTile<BM, BN2, WM2, WN2>;
    constexpr int TM2 = BT2::TM;
    static_assert(BT2::TN == 1 && WM2 * WN2 == 8, "8 waves, one column block each");
    char* lds = reinterpret_cast<char*>(smem);
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));                 // nothing below is computed before, or carried through, the K loop above
    const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    // ---- this layer's epilogue -> LDS -------------------------------------------------------------------------------------
    // (straight-line like finish_plain: the dropout decision once per tile, packed BN; rows past the end of the tensor hold
    //  leaky(shift) -- their operand rows were zeros -- which the follower multiplies and never stores)
    auto to_lds = [&](auto drop_tag) {
        constexpr bool DROP = decltype(drop_tag)::value;
        const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
        const int nb = wave * 32 + 4 * lh;                    // first channel of the lane's group g = 0
        float vmax = 0.f;
        f32x4 sc4[4], sf4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(p.scale + nb + 8 * g);
            sf4[g] = *reinterpret_cast<const f32x4*>(p.shift + nb + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t rr = (uint32_t)(i * 32 + li), m = tile_m * BM + rr;
            const uint64_t idx_row = p.idx_base + (uint64_t)m * (uint64_t)p.N + (uint64_t)nb;
            const epi::DropRow drow(idx_row, p.k1);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dn = 8 * g;
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = acc[i][0][4 * g + q];
                bool keep[4] = {true, true, true, true};
                if constexpr (DROP) epi::keep4(drow, dn, p.k0, p.thr, keep);     // (injected masks: the planner does not fuse such a call)
                const f32x4 v = epi::bn_act4_pk(a4, sc4[g], sf4[g], keep, slope);
                vmax = epi::absmax4(vmax, v);
                const f32x4 e = epi::split_encode4(v);
                // K-tile `wave` of the follower's input; 4-channel group (lh + 2 g) of its 32 channels: hi at +8 q, lo at +64 + 8 q
                char* at = lds + wave * A_BUF + rr * ROWB + (lh + 2 * g) * 8;
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                *reinterpret_cast<f32x2*>(at) = f32x2{e[0], e[1]};
                *reinterpret_cast<f32x2*>(at + 64) = f32x2{e[2], e[3]};
            }
        }
        if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
    };
    if (p.flags & EPI_DROPOUT) to_lds(std::true_type{}); else to_lds(std::false_type{});
    __syncthreads();
    // ---- the follower's GEMM: [128 x 256] x [256 x N2] ---------------------------------------------------------------------
    const BT2 bt(smem, tid);
    f32x16 acc2[TM2][1];
#pragma unroll
    for (int i = 0; i < TM2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[i][0][r] = 0.f;
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.f_wpk, p.f_w_bytes);
    const uint32_t w_step = (uint32_t)p.f_Npad * BK * 4;
    // Software pipeline, pinned with scheduling barriers: while step s (16 channels) multiplies, the activation fragments of step s + 1
    // and (every other step) the weight fragments of the next K-tile are in flight into the other register sets.  (Left to itself the compiler sank every weight load to just in
    // front of its first use -- one fragment set, `s_waitcnt vmcnt(0)` after each of the 32 loads: 11 us of exposed latency per tile.)
    f16x8 bfr[2][2][1][2];                        // [set = K-tile & 1][step][column block][plane]
    f16x8 afr[2][TM2][2];                         // [set = step & 1][block][plane]
    bt.load_b(bfr[0], w_rsrc, 0u);
    bt.template read_frags<0, 0>(afr[0]);
    auto step2 = [&](auto s_tag) {                // step S (16 channels) of the 16: K-tile S / 2, half S % 2
        constexpr int S = decltype(s_tag)::value, KT2 = S >> 1, H = S & 1;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (H == 0 && KT2 < 7) bt.load_b(bfr[(KT2 + 1) & 1], w_rsrc, (uint32_t)(KT2 + 1) * w_step);
        if constexpr (S < 15) bt.template read_frags<((S + 1) >> 1) & 7, (S + 1) & 1>(afr[(S + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        mfma_step_split<TM2, 1>(acc2, afr[S & 1], bfr[KT2 & 1][H]);
        __builtin_amdgcn_sched_barrier(0);
    };
    step2(std::integral_constant<int, 0>{}); step2(std::integral_constant<int, 1>{}); step2(std::integral_constant<int, 2>{});
    step2(std::integral_constant<int, 3>{}); step2(std::integral_constant<int, 4>{}); step2(std::integral_constant<int, 5>{});
    step2(std::integral_constant<int, 6>{}); step2(std::integral_constant<int, 7>{}); step2(std::integral_constant<int, 8>{});
    step2(std::integral_constant<int, 9>{}); step2(std::integral_constant<int, 10>{}); step2(std::integral_constant<int, 11>{});
    step2(std::integral_constant<int, 12>{}); step2(std::integral_constant<int, 13>{}); step2(std::integral_constant<int, 14>{});
    step2(std::integral_constant<int, 15>{});
    __syncthreads();                              // every fragment is in registers: the next tile of this workgroup may stage again
    // ---- the follower's epilogue (the vector form of finish_tile: cout and row pitch are multiples of 4; no addend, no residual, one
    // sample per row -- the planner fuses nothing else); dropout and the output encoding decided once per tile ------------------------
    auto follower = [&](auto drop_tag, auto split_tag) {
        constexpr bool DROP = decltype(drop_tag)::value, SPLIT_OUT = decltype(split_tag)::value;
        const float slope = (p.f_flags & EPI_LEAKY) ? 0.1f : 1.f;
        const int nb = bt.wn * 32 + 4 * lh;
        float vmax = 0.f;
#pragma unroll
        for (int i = 0; i < TM2; ++i) {
            const uint32_t m = tile_m * BM + (uint32_t)((bt.wm * TM2 + i) * 32 + li);
            if (m >= (uint32_t)p.M) continue;
            const uint64_t idx_row = p.f_idx_base + (uint64_t)m * (uint64_t)p.f_N + (uint64_t)nb;
            const epi::DropRow drow(idx_row, p.f_k1);
            float* d = p.f_dst + (size_t)m * p.f_ldc + nb;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int dn = 8 * g, n0 = nb + dn;
                if (n0 >= p.f_N) continue;
                const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.f_scale + n0);
                const f32x4 sf4 = *reinterpret_cast<const f32x4*>(p.f_shift + n0);
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = acc2[i][0][4 * g + q];
                bool keep[4] = {true, true, true, true};
                if constexpr (DROP) epi::keep4(drow, dn, p.f_k0, p.f_thr, keep);
                const f32x4 v = epi::bn_act4_pk(a4, sc4, sf4, keep, slope);
                if constexpr (SPLIT_OUT) vmax = epi::absmax4(vmax, v);
                if constexpr (SPLIT_OUT) *reinterpret_cast<f32x4*>(d + dn) = epi::split_encode4(v);
                else *reinterpret_cast<f32x4*>(d + dn) = v;
            }
        }
        if (SPLIT_OUT && p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.f_layer_idx); }
    };
    if (p.f_flags & EPI_F32OUT) {                 // a detection head: plain fp32 rows
        if (p.f_flags & EPI_DROPOUT) follower(std::true_type{}, std::false_type{}); else follower(std::false_type{}, std::false_type{});
    } else if (p.f_flags & EPI_DROPOUT) follower(std::true_type{}, std::true_type{});
    else follower(std::false_type{}, std::true_type{});
}

// 3x3 / stride-1 / one plain source, split-f16 (SplitTileKx in mfma_pipe.h): stages [sg_begin, sg_end) of the tile, a
// stage = (filter row ky, 32-channel chunk) = the three K-tiles kx = 0, 1, 2 on one staged activation tile.  The body
// is uniform: every K-tile stages the weight tile after it and fetches the one after that, every stage fetches and
// stages its successor -- past the end of the range those loads read zeros / run past the buffer (bounds-checked) and
// what they stage is never used.
// WALK (the 8-wave 128 x 256 tile, whole tiles only): ONE workgroup per CU, so nothing covers the turn-over between two tiles
// there -- the workgroup walks the tile list itself (stride gridDim.x from tile `logical_or_first` = blockIdx.x), and fetches the first
// stage and weight tile of its NEXT tile before the epilogue of the current one: the loads fly while the epilogue computes.
template <int BM, int BN, int WM, int WN, bool WALK = false>
__device__ __forceinline__ void conv_tile_kx3(const ConvParams& p, float* smem, const int logical_or_first, const int sg_begin,
                                              const int sg_end, const TileShare sh) {
    using BT = SplitTileKx<BM, BN, WM, WN>;
    constexpr int NT = BT::NT, TM = BT::TM, TN = BT::TN, A_LD = BT::A_LD, A_LDX = BT::A_LDX, ROWB = BT::ROWB;
    const BT bt(smem);
    const uint32_t n_tiles = (uint32_t)p.Npad / BN;
    const uint32_t hw = (uint32_t)(p.Hout * p.Wout), W = (uint32_t)p.Wout;
    int v = logical_or_first;                                  // WALK: position in the tile list (before the XCD remap)
    uint32_t tile_m, tile_n;
    auto locate = [&](int logical) { tile_m = fdiv((uint32_t)logical, p.d_ntiles); tile_n = (uint32_t)logical - tile_m * n_tiles; };
    locate(WALK ? xcd_remap(v, p.full_tiles) : logical_or_first);

    uint32_t a_off0[A_LDX], a_vm[A_LDX], a_voff[A_LDX];   // offset of input pixel (y - 1, x); bit ky: row y + ky - 1 exists
    uint32_t fa[3][TM];
    int ld_ky = 0, ld_c = 0;
    bool ld_first = true;
    uint32_t a_soff = 0, w_soff = 0;
    const uint32_t w_step = (uint32_t)p.Npad * BK * 4;
    // per-tile bookkeeping for tile (tile_m, tile_n), its stages from `sgb` on
    auto setup = [&](const int sgb) {
        // ---- staging rows of this thread: A_LD rows of the tile + one halo row (threads 0 .. 15) -----------------------
#pragma unroll
        for (int j = 0; j < A_LDX; ++j) {
            const int rho = j < A_LD ? bt.a_r + (NT / 8) * j + 1 : (bt.a_r == 0 ? 0 : (bt.a_r == 1 ? BM + 1 : -1));
            const int64_t mm = (int64_t)tile_m * BM - 1 + rho;
            const bool ok = rho >= 0 && mm >= 0 && mm < (int64_t)p.M;
            const uint32_t m = ok ? (uint32_t)mm : 0u;
            const uint32_t sidx = fdiv(m, p.d_hw), rem = m - sidx * hw;
            const uint32_t oy = fdiv(rem, p.d_wout), ox = rem - oy * W;
            unsigned vm = 0;
#pragma unroll
            for (int t = 0; t < 3; ++t) vm |= ((unsigned)((int)oy + t - 1) < (unsigned)p.Hin) ? (1u << t) : 0u;
            a_vm[j] = ok ? vm : 0u;
            const uint32_t s0 = fdiv(sidx, p.d_sdiv0);
            a_off0[j] = ((((s0 * (uint32_t)p.Hs0 + (oy - 1u)) * (uint32_t)p.Ws0 + ox) * (uint32_t)p.C0) + bt.a_q * 4) * 4u;
            a_voff[j] = CONV_OOB_OFFSET;
        }
        // ---- fragment row addresses: block i of this wave, kx = 0 / 1 / 2 ---------------------------------------------
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t rr = bt.wm * TM * 32 + i * 32 + bt.li, m = tile_m * BM + rr;
            const uint32_t sidx = fdiv(m, p.d_hw), rem = m - sidx * hw;
            const uint32_t oy = fdiv(rem, p.d_wout), ox = rem - oy * W;
            fa[0][i] = (ox > 0 ? rr : (uint32_t)BT::ZROW) * ROWB + bt.lh * 16;
            fa[1][i] = (rr + 1) * ROWB + bt.lh * 16;
            fa[2][i] = (ox + 1 < W ? rr + 2 : (uint32_t)BT::ZROW) * ROWB + bt.lh * 16;
        }
        // ---- sequencing (scalar): the NEXT stage to load, the next weight tile --------------------------------------
        ld_ky = (int)fdiv((uint32_t)sgb, p.d_cin); ld_c = sgb - ld_ky * p.cin_tiles;
        ld_first = true;
        a_soff = 0;
        w_soff = (uint32_t)sgb * 3u * w_step + tile_n * (BN / 32) * SPLIT_WBLOCK;
    };
    setup(sg_begin);
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.src0, p.src0_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wpk, p.w_bytes);
    f32x4 a_reg[A_LDX];
    f16x8 bfr[2][2][TN][2];                       // weight fragments of K-tiles t, t+1 (set = t & 1), fetched one K-tile ahead
    auto next_stage = [&]() {
        if (ld_c == 0 || ld_first) {
            const uint32_t delta = (uint32_t)(ld_ky * p.Ws0 * p.C0) * 4u, bit = 1u << ld_ky;     // ky >= 3 (past the end): no bit
#pragma unroll
            for (int j = 0; j < A_LDX; ++j) a_voff[j] = (a_vm[j] & bit) ? a_off0[j] + delta : CONV_OOB_OFFSET;
        }
        a_soff = (uint32_t)ld_c * (BK * 4);
        ld_first = false;
        if (++ld_c == p.cin_tiles) { ld_c = 0; ++ld_ky; }
    };
    auto load_a = [&]() {
        if constexpr ((ABL & 1) != 0) return;
#pragma unroll
        for (int j = 0; j < A_LDX; ++j) a_reg[j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
    };
    auto load_b = [&](auto set_tag) {
        if constexpr ((ABL & 1) == 0) bt.load_b(bfr[decltype(set_tag)::value], w_rsrc, w_soff);
        w_soff += w_step;
    };

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    f32x16 acc[TM][TN];
    f16x8 af0[TM][2], af1[TM][2];
    next_stage(); load_a(); load_b(c0{});

    // K-tile Q (= kx) of a stage in activation buffer AP; its weight fragments are in set (AP + Q) & 1 (a stage flips both)
    auto ktile = [&](auto ap_tag, auto q_tag) {
        constexpr int AP = decltype(ap_tag)::value, Q = decltype(q_tag)::value, BS = (AP + Q) & 1;
        constexpr int G = BT::G, NFR = (ABL & 8) ? 0 : BT::NFR, NLDB = (ABL & 1) ? 0 : BT::NBF, NLDA = (ABL & 1) ? 0 : A_LDX;
        constexpr int NSTA = (ABL & 2) ? 0 : A_LDX;
        __builtin_amdgcn_sched_barrier(0);
        load_b(std::integral_constant<int, BS ^ 1>{});                              // weight fragments of K-tile t+1
        if constexpr (!(ABL & 8)) bt.template read_frags_kx<AP, 1>(fa[Q], af1);
        if constexpr (Q == 2 && !(ABL & 2)) bt.template store_stage<AP ^ 1>(a_reg);  // the next stage (fetched in K-tile 0)
        mfma_step_split<TM, TN>(acc, af0, bfr[BS][0]);
        sched_interleave<G, NLDB, NFR, Q == 2 ? NSTA : 0>();
        __builtin_amdgcn_sched_barrier(0);

        if constexpr (Q == 2 && !(ABL & 4)) __syncthreads();                       // the next stage is visible; this one is read out
        if constexpr (Q == 0) { next_stage(); load_a(); }                          // the next stage's activations
        if constexpr (!(ABL & 8)) {
            if constexpr (Q < 2) bt.template read_frags_kx<AP, 0>(fa[Q + 1], af0);
            else bt.template read_frags_kx<AP ^ 1, 0>(fa[0], af0);
        }
        mfma_step_split<TM, TN>(acc, af1, bfr[BS][1]);
        sched_interleave<G, Q == 0 ? NLDA : 0, NFR, 0>();
        __builtin_amdgcn_sched_barrier(0);
    };
    const int NS = sg_end - sg_begin;
    for (;;) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        bt.template store_stage<0>(a_reg);
        __syncthreads();
        bt.template read_frags_kx<0, 0>(fa[0], af0);
        int sg = 0;
        for (; sg + 1 < NS; sg += 2) {
            ktile(c0{}, c0{}); ktile(c0{}, c1{}); ktile(c0{}, c2{});
            ktile(c1{}, c0{}); ktile(c1{}, c1{}); ktile(c1{}, c2{});
        }
        if (sg < NS) { ktile(c0{}, c0{}); ktile(c0{}, c1{}); ktile(c0{}, c2{}); }
        __syncthreads();                          // the trailing LDS traffic of the uniform body is done before LDS is reused
        const uint32_t cur_m = tile_m, cur_n = tile_n;
        bool more = false;
        if constexpr (WALK) {
            v += (int)gridDim.x;
            more = v < p.full_tiles;
            if (more) {                           // the next tile's first stage and weight tile fly during this tile's epilogue
                locate(xcd_remap(v, p.full_tiles));
                setup(0);
                next_stage(); load_a(); load_b(c0{});
            }
        }
        bool fused = false;
        if constexpr (BN == 256 && WM == 1 && WN == 8) {
            if (p.f_wpk) {                        // back-to-back: the follower 1x1 / detection head in this launch (whole tiles only)
                if (p.f_Npad > 64) fused_tail<2, 4>(p, smem, acc, cur_m);
                else fused_tail<4, 2>(p, smem, acc, cur_m);
                fused = true;
            }
        }
        if (!fused) finish_split<BM, BN, WM, WN, BN != 256>(p, smem, acc, cur_m, cur_n, sh);
        if (!more) return;
    }
}

// 1x1 / stride-1 convolution over ONE plain source, split-f16: K-tiles [kt_begin, kt_end) of the tile on a UNIFORM, tail-free loop.
// These launches -- the 1x1 convolutions of the heads, the stacked half of the concat convolutions, the detection heads -- have
// short K loops (8 .. 32 K-tiles) and stream their input from HBM, and they were waiting: for loads fetched one K-tile ahead
// and at one barrier per 12 MFMAs on the 64-wide tile.  Here the activations of tile t + 3 are fetched while tile t multiplies
// (two staging sets), every K-tile stages its successor whether or not it exists (past the end the loads read the next rows'
// bytes or zeros, staged and never multiplied), and without tail copies of the body the 128 x 128 tile (24 MFMAs per wave and
// barrier, the input read once for 128 columns) stays inside 256 registers.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_tile_p1(const ConvParams& p, float* smem, const int logical, const int kt_begin,
                                             const int kt_end, const TileShare sh) {
    using BT = SplitTile<BM, BN, WM, WN>;
    constexpr int NT = BT::NT, TM = BT::TM, TN = BT::TN, A_LD = BT::A_LD;
    const BT bt(smem);
    const uint32_t n_tiles = (uint32_t)p.Npad / BN;
    const uint32_t tile_m = fdiv((uint32_t)logical, p.d_ntiles), tile_n = (uint32_t)logical - tile_m * n_tiles;
    const uint32_t hw = (uint32_t)(p.Hout * p.Wout);
    uint32_t a_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const uint32_t m = tile_m * BM + bt.a_r + (NT / 8) * j;
        const uint32_t mm = m < (uint32_t)p.M ? m : 0u;
        // the source may be a T-fold tile of an unstacked tensor (sample / T) and / or read through a nearest x2 upsample
        // ((y >> 1, x >> 1): the stacked half of the heads' concat convolutions, lib_yolo/layers.py:578-580)
        const uint32_t s = fdiv(mm, p.d_hw), rem = mm - s * hw;
        const uint32_t oy = fdiv(rem, p.d_wout), ox = rem - oy * (uint32_t)p.Wout;
        const uint32_t row = (fdiv(s, p.d_sdiv0) * (uint32_t)p.Hs0 + (oy >> p.sh0)) * (uint32_t)p.Ws0 + (ox >> p.sh0);
        a_voff[j] = m < (uint32_t)p.M ? (row * (uint32_t)p.C0 + (uint32_t)bt.a_q * 4u) * 4u : CONV_OOB_OFFSET;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.src0, p.src0_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.wpk, p.w_bytes);
    const uint32_t w_step = (uint32_t)p.Npad * BK * 4;
    uint32_t a_soff = (uint32_t)kt_begin * (BK * 4);                      // the NEXT tile to load
    uint32_t w_soff = (uint32_t)kt_begin * w_step + tile_n * (BN / 32) * SPLIT_WBLOCK;
    f32x4 a_reg[2][A_LD];
    f16x8 bfr[2][2][TN][2];
    auto load_a = [&](auto set_tag) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[decltype(set_tag)::value][j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
        a_soff += BK * 4;
    };
    auto load_b = [&](auto set_tag) { bt.load_b(bfr[decltype(set_tag)::value], w_rsrc, w_soff); w_soff += w_step; };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    f16x8 af0[TM][2], af1[TM][2];
    load_a(c0{}); load_b(c0{});
    bt.template store_a<0>(a_reg[0]);
    load_a(c1{}); load_a(c0{});                                           // tiles 1 and 2 wait in the sets 1 and 0
    __syncthreads();
    bt.template read_frags<0, 0>(af0);
    auto ktile = [&](auto buf_tag) {                                      // tile t in LDS buffer BUF = t & 1; set BUF ^ 1 holds tile t + 1, then t + 3
        constexpr int BUF = decltype(buf_tag)::value;
        using NX = std::integral_constant<int, BUF ^ 1>;
        pipe::tile_body_split<BUF, true, BT::NBF, A_LD, A_LD, 0>(
            bt, acc, af0, af1, bfr[BUF], [&] { load_b(NX{}); }, [&] { load_a(NX{}); }, [&] { bt.template store_a<BUF ^ 1>(a_reg[BUF ^ 1]); });
    };
    const int n = kt_end - kt_begin;
    int t = 0;
    for (; t + 1 < n; t += 2) { ktile(c0{}); ktile(c1{}); }
    if (t < n) ktile(c0{});
    __syncthreads();                              // the trailing LDS traffic of the uniform body is done before LDS is reused
    finish_split<BM, BN, WM, WN>(p, smem, acc, tile_m, tile_n, sh);
}

// Workgroups walk the tile list with stride gridDim.x: with gridDim.x == #tiles every workgroup owns one
// tile; with a smaller (persistent) grid a workgroup runs several tiles back to back.  Either way the
// tiles that are in flight on one XCD at a time are neighbours.
// (Measured and dropped: 8-wave 128x128 and 256x128 tiles, one workgroup per CU, a persistent grid, two
//  streams, s_setprio, and de-phasing the two co-resident workgroups of a CU at launch -- none moved the
//  number)
// 2nd launch bound = waves per SIMD: two workgroups per CU (what the LDS allows for the 128x128 tile) must
// also fit the register file, i.e. VGPRs + AGPRs <= 256 per wave.
// KX3: 0 the (tap, chunk) loop of conv_tile, 1 shared-tap 3x3 stages (conv_tile_kx3), 2 the 1x1 loop (conv_tile_p1)
template <int BM, int BN, int WM, int WN, bool FAST, bool SPLIT, int KX3 = 0>
__global__ __launch_bounds__(64 * WM * WN, 2) void conv_igemm_kernel(const ConvParams p) {      // two waves per SIMD: <= 256 registers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // One loop, one inlined conv_tile: the work items of this workgroup are either
    //  (stream-K, p.sk_grid > 0) the tiles its share of the launch's tiles * KT units overlaps.  Workgroup order =
    //    unit order per XCD (xcd_remap), so the tiles in flight on an XCD are neighbours and most split tiles stay on
    //    one XCD; a straddling tile's slabs are indexed by the workgroup that wrote them (2 per workgroup: the head and
    //    the tail of its range), its ticket counter by the workgroup that owns its first unit; or
    //  (otherwise) blocks [0, full_tiles): whole tiles, each XCD a contiguous range of them; blocks beyond: K slices of
    //    the remaining tiles -- slices of one tile on one XCD (block id % 8), neighbours in dispatch order:
    //    id = full_tiles + ((tile_local / 8) * ksplit + slice) * 8 + tile_local % 8
    if constexpr (KX3 == 1 && BN == 256) {        // the 8-wave tile: whole tiles only (byolo_plan.hip make_plan), walked by the workgroup itself
        if ((int)blockIdx.x < p.full_tiles)
            conv_tile_kx3<BM, BN, WM, WN, true>(p, smem, (int)blockIdx.x, 0, p.KT, TileShare{-1, 0, 1, 0, 0, 0});
        return;
    }
    const bool sk = p.sk_grid > 0;
    const uint32_t q = (uint32_t)p.sk_q, r = (uint32_t)p.sk_r, KT = (uint32_t)p.KT;
    auto start = [&](uint32_t w) { return w * q + (w < r ? w : r); };
    auto owner = [&](uint32_t u) { return u < r * (q + 1) ? fdiv(u, p.d_skq1) : r + fdiv(u - r * (q + 1), p.d_skq); };
    const uint32_t g = sk ? (uint32_t)xcd_remap(blockIdx.x, p.sk_grid) : 0u;
    uint32_t u = sk ? start(g) : 0u;
    const uint32_t u_end = sk ? start(g + 1) : 0u;
    const int total = p.full_tiles + p.split_blocks;
    int v = blockIdx.x;
    for (;;) {
        int logical, kb, ke;
        TileShare sh{-1, 0, 1, 0, 0, 0};
        if (sk) {
            if (u >= u_end) break;
            const uint32_t t = fdiv(u, p.d_kt), t0 = t * KT;
            logical = (int)t; kb = (int)(u - t0); ke = (int)((u_end - t0 < KT) ? u_end - t0 : KT);
            if (kb != 0 || ke != (int)KT) {                      // this tile straddles workgroups
                const uint32_t g_first = owner(t0), g_last = owner(t0 + KT - 1);
                const int first_add = start(g_first) != t0;     // g_first's segment is the tail of ITS range
                sh.counter = (int)g_first; sh.nseg = (int)(g_last - g_first + 1);
                sh.base = 2 * (int)g_first; sh.stride = 2; sh.first_add = first_add;
                sh.my_slab = 2 * (int)g + ((g == g_first && first_add) ? 1 : 0);
            }
            u = t0 + (uint32_t)ke;
        } else {
            if (v >= total) break;
            const int cur = v;
            v += gridDim.x;
            logical = xcd_remap(cur, p.full_tiles); kb = 0; ke = p.KT;
            if (cur >= p.full_tiles) {
                const uint32_t id = (uint32_t)(cur - p.full_tiles), i = id >> 3, x = id & 7u;
                const uint32_t grp = fdiv(i, p.d_ksplit);
                const int slice = (int)(i - grp * (uint32_t)p.ksplit), tile_local = (int)(grp * 8u + x);
                if (tile_local >= p.split_tiles) continue;
                logical = p.full_tiles + tile_local;
                kb = (int)fdiv((uint32_t)slice * (uint32_t)p.KT, p.d_ksplit);
                ke = (int)fdiv((uint32_t)(slice + 1) * (uint32_t)p.KT, p.d_ksplit);
                sh = TileShare{tile_local, tile_local * p.ksplit + slice, p.ksplit, tile_local * p.ksplit, 1, 0};
            }
        }
        if constexpr (KX3 == 1) conv_tile_kx3<BM, BN, WM, WN>(p, smem, logical, kb, ke, sh);    // p.KT, kb, ke count STAGES (3 K-tiles)
        else if constexpr (KX3 == 2) conv_tile_p1<BM, BN, WM, WN>(p, smem, logical, kb, ke, sh);
        else conv_tile<BM, BN, WM, WN, FAST, SPLIT>(p, smem, logical, kb, ke, sh);
    }
}

int conv_tile_bn(int tile) { return tile == TILE_128x256 ? 256 : (tile == TILE_128x128 ? 128 : (tile == TILE_128x64 ? 64 : 32)); }

// split precision: the tile configuration a launch really runs on
int conv_split_tile(int tile, bool wide) { return (tile == TILE_128x128 && !wide) ? TILE_128x64 : tile; }

int conv_pick_tile(int N) {
    if (N > 64) return TILE_128x128;
    if (N > 32) return TILE_128x64;
    return TILE_128x32;
}

template <int BM, int BN, int WM, int WN, bool FAST, bool SPLIT, int KX3 = 0>
static hipError_t launch_one(const ConvParams& p, int grid, hipStream_t st) {
    constexpr size_t lds_loop = KX3 == 1 ? (size_t)SplitTileKx<BM, BN, WM, WN>::LDS_BYTES : (SPLIT ? (size_t)SplitTile<BM, BN, WM, WN>::LDS_BYTES : (size_t)BlockTile<BM, BN, WM, WN>::LDS_BYTES);
    // the 8-wave shared-tap tile may carry a fused follower (fused_tail): 8 K-tiles of [128][144 bytes] of LDS
    constexpr size_t lds_fused = (KX3 == 1 && BN == 256) ? (size_t)8 * 128 * LD * 4 : 0;
    constexpr size_t lds_max = lds_fused > lds_loop ? lds_fused : lds_loop;
    const size_t lds = (lds_fused && p.f_wpk) ? lds_max : lds_loop;
    auto k = conv_igemm_kernel<BM, BN, WM, WN, FAST, SPLIT, KX3>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(k), lds_max, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), lds, st, p);
    return hipGetLastError();
}

// workgroups of a tile configuration resident on the chip: LDS-limited (160 KB per CU; 73.7 / 55.3 / 46.1 KB per
// workgroup), the register bound of __launch_bounds__ allows at least as many
static int tile_slots(int tile) { return 256 * (tile == TILE_128x256 ? 1 : (tile == TILE_128x32 ? 3 : 2)); }      // (the 8-wave tile: one workgroup per CU)

// Tile quantisation: a launch of `tiles` equal tiles on `slots` resident workgroups takes ceil(tiles/slots)
// rounds although the last one may be nearly empty (5416 tiles on 512 slots: 10.58 -> 11 rounds, 3.8 % of
// the chip-time idle).  The tiles of that last partial round are therefore cut into `ksplit` K slices so that
// they fill the chip once more with shorter blocks.  Cost model in tile-times of a whole tile; t_k, t_o from
// DESIGN.md section 3 (per-K-tile and per-tile overhead), a slice pays the overhead + the hand-off fences.
ConvSplit conv_plan_split(int M, int Npad, int KT, int tile, double tk_scale, int ksplit_knob, int streamk_knob) {
    const int BN = conv_tile_bn(tile), slots = tile_slots(tile);
    const int tiles = ((M + 127) / 128) * (Npad / BN);
    ConvSplit r; r.full_tiles = tiles; r.split_tiles = 0; r.ksplit = 1; r.split_blocks = 0; r.sk_grid = 0;
    // byolo_plan_opts.ksplit: -1 the cost model below, 0 = never split, n > 1 = always n slices (tests, A/B)
    const int knob = ksplit_knob;
    // microseconds: K-tile and tile overhead of a whole tile, overhead of a slice (prologue, sc1 slab
    // round trip, ticket; measured on the 19x19 .. 76x76 head shapes and the small backbone launches)
    const double t_k = 1.8 * BN / 128.0 * tk_scale, t_o = 3.3, t_slice = 16.0;    // tk_scale: K unit of the launch relative to an fp32 K-tile
    const double whole = KT * t_k + t_o;
    // ---- stream-K for small launches --------------------------------------------------------------------------
    // Fewer tiles than a few rounds of resident workgroups: a whole-tile schedule either leaves CUs idle (tiles <
    // CUs: a tile runs on ONE CU, however long its K loop) or pays a nearly empty last round (676 tiles on 512 slots
    // = 2 rounds for 1.32 rounds of work).  Here every resident workgroup takes an equal share of the launch's
    // tiles * KT K-tile units; cost in tile-times = share + hand-off of the (at most two) partial tiles of a workgroup.
    // byolo_plan_opts.streamk: 0 never, 1 when the model predicts a gain (default), 2 whenever admissible (tests).
    const int sk_knob = streamk_knob;
    if (sk_knob && knob < 0 && KT >= 2) {
        const int64_t U = (int64_t)tiles * KT;
        int G = (int)std::min<int64_t>(slots, U / (sk_knob >= 2 ? 1 : 6)) & ~7;       // >= 6 K-tiles per workgroup
        if (G >= 8 && U * G < ((int64_t)1 << 31) && U > G) {
            // In units of `whole` (one tile's time on a CU of its own pipe): two workgroups share a CU's matrix pipe, so a
            // round of `slots` tiles takes 2 wholes; a tile alone on its CU takes 1.16 (measured 0.58 of the shared time).
            const double rounds = (double)tiles / slots;
            const double plain = tiles * 2 <= slots ? 1.16 : 2.0 * std::ceil(rounds);
            const double sk = (double)tiles / G * (G * 2 <= slots ? 1.16 : 2.0) + t_slice / whole;
            if (sk_knob >= 2 || (rounds < 4.0 && sk < 0.93 * plain)) { r.sk_grid = G; return r; }
        }
    }
    const int rem = tiles % slots;
    if (knob == 0 || rem == 0) return r;
    // a round that leaves at most one workgroup per CU runs faster than a full one (a lone workgroup owns
    // the matrix pipe), IF the dispatcher spreads it -- it does not always: 0.8 / 0.7 are averages
    // (measured: bimodal 0.58 / 1.0 for the 128x128 tile, always spread for the 128x64 tile)
    const double base = rem > 256 ? 1.0 : (tile == TILE_128x128 ? 0.8 : 0.62);
    double best = base; int best_s = 1;
    if (whole < 60.0 && knob <= 1) return r;     // fixed launch costs dominate: nothing to win
    for (int s = 2; s <= 8; ++s) {
        if (KT / s < 8) break;
        const int blocks = rem * s, rounds = (blocks + slots - 1) / slots, last = blocks - (rounds - 1) * slots;
        const double frac = ((double)((KT + s - 1) / s) * t_k + t_slice) / whole;
        const double cost = ((rounds - 1) + (last > 256 ? 1.0 : 0.7)) * frac;
        if (cost < best - 0.08) { best = cost; best_s = s; }
    }
    if (knob > 1) best_s = (KT / knob >= 2) ? knob : 1;
    if (best_s == 1) return r;
    r.full_tiles = tiles - rem; r.split_tiles = rem; r.ksplit = best_s;
    r.split_blocks = ((rem + 7) / 8) * 8 * best_s;
    return r;
}
size_t conv_split_slab_bytes(const ConvSplit& sp, int tile) {
    if (sp.sk_grid > 0) return (size_t)2 * sp.sk_grid * 128 * conv_tile_bn(tile) * sizeof(float);
    return sp.ksplit > 1 ? (size_t)sp.split_tiles * sp.ksplit * 128 * conv_tile_bn(tile) * sizeof(float) : 0;
}

// ConvParams::plain (finish_plain)
static bool conv_epilogue_is_plain(const ConvParams& p) {
    // (p.no_plain: byolo_plan_opts.plain_epilogue = 0 -- finish_tile everywhere, the A/B of tests/test_gpu_parity.py)
    return !p.no_plain && p.split == 1 && !p.addend && p.rep <= 1 && !(p.flags & (EPI_F32OUT | EPI_RAW)) && !p.mask_bits &&
           !((p.flags & EPI_RESIDUAL) && (p.flags & EPI_DROPOUT)) && ((p.N | p.ldc) & 3) == 0 && (p.N % 32) == 0;
}

template <int BM, int BN, int WM, int WN, bool SPLITCFG = false>
static hipError_t launch_cfg(const ConvParams& p, hipStream_t st) {
    ConvParams q = p;
    q.plain = conv_epilogue_is_plain(p);
    q.d_ntiles = make_fastdiv((uint32_t)(p.Npad / BN));
    q.d_cin = make_fastdiv((uint32_t)p.cin_tiles);
    q.d_ks = make_fastdiv((uint32_t)p.ksize);
    const int tiles = ((p.M + BM - 1) / BM) * (p.Npad / BN);
    if (p.ksplit <= 1 || !p.slabs || !p.counters) { q.full_tiles = tiles; q.split_tiles = 0; q.split_blocks = 0; q.ksplit = 1; }
    if (!p.slabs || !p.counters || p.sk_grid <= 0 || (int64_t)tiles * p.KT <= p.sk_grid) q.sk_grid = 0;
    q.d_ksplit = make_fastdiv((uint32_t)q.ksplit);
    int grid = q.full_tiles + q.split_blocks;
    if (q.sk_grid > 0) {
        const uint32_t U = (uint32_t)tiles * (uint32_t)p.KT;
        q.sk_q = (int)(U / (uint32_t)q.sk_grid); q.sk_r = (int)(U % (uint32_t)q.sk_grid);
        q.d_skq = make_fastdiv((uint32_t)q.sk_q); q.d_skq1 = make_fastdiv((uint32_t)q.sk_q + 1u); q.d_kt = make_fastdiv((uint32_t)p.KT);
        grid = q.sk_grid;
    }
    // (a persistent grid for the 4-wave tiles measured +-0 in round 4 -- two workgroups per CU cover each other's turn-over -- and its
    //  knob is gone.)  The 8-wave tile runs ONE workgroup per CU: nothing covers the turn-over between two workgroups (dispatch,
    // kernel-argument loads, address prologue) there, so its workgroups walk the tile list themselves.
    if constexpr (BN == 256) {
        if (q.split_blocks != 0 || q.sk_grid != 0) return hipErrorInvalidValue;      // whole tiles only on this tile
        if (grid > 256) grid = 256;
    }
    const bool fast = p.C1 == 0 && p.sh0 == 0 && p.ksize <= 3;
    if (p.split != (SPLITCFG ? 1 : 0)) return hipErrorInvalidValue;
    if constexpr (SPLITCFG) {
        if constexpr (BN >= 64) {
            if (p.kx3 == 1) {
                if (!fast || p.ksize != 3 || p.stride != 1) return hipErrorInvalidValue;
                return launch_one<BM, BN, WM, WN, true, true, 1>(q, grid, st);
            }
            if constexpr (BN <= 128) {
                if (p.kx3 == 2) {
                    if (p.C1 != 0 || p.ksize != 1 || p.stride != 1) return hipErrorInvalidValue;
                    return launch_one<BM, BN, WM, WN, true, true, 2>(q, grid, st);
                }
            }
        }
        // (the plain split kernel is not built for the 128-wide tile: it needs more than 256 registers there, and this
        //  library keeps out of scratch memory -- conv_split_tile() maps such launches to the 64-wide tile)
        if constexpr (BN >= 128) return hipErrorInvalidValue;
        else return fast ? launch_one<BM, BN, WM, WN, true, true>(q, grid, st) : launch_one<BM, BN, WM, WN, false, true>(q, grid, st);
    } else
        return fast ? launch_one<BM, BN, WM, WN, true, false>(q, grid, st) : launch_one<BM, BN, WM, WN, false, false>(q, grid, st);
}

hipError_t launch_conv_igemm(const ConvParams& p, int tile, hipStream_t st) {
    if (p.split) {      // split-f16: the waves sit side by side along N (each fetches its own weight fragments, mfma_pipe.h)
        // (a 2 x 2 wave grid -- half the LDS fragment reads, every weight fragment fetched twice -- measured the same: 18.4 ms)
        switch (tile) {
            case TILE_128x256:                                                     // 8 waves of 128x32: the shared-tap 3x3 kernel only
                if (p.kx3 != 1) return hipErrorInvalidValue;
                return launch_cfg<128, 256, 1, 8, true>(p, st);
            case TILE_128x128: return launch_cfg<128, 128, 1, 4, true>(p, st);    // 4 waves of 128x32
            case TILE_128x64:  return launch_cfg<128, 64, 2, 2, true>(p, st);     // 4 waves of 64x32
            default:           return launch_cfg<128, 32, 4, 1, true>(p, st);     // 4 waves of 32x32
        }
    }
    switch (tile) {
        case TILE_128x128: return launch_cfg<128, 128, 2, 2>(p, st);      // 4 waves of 64x64
        case TILE_128x64:  return launch_cfg<128, 64, 2, 2>(p, st);       // 4 waves of 64x32
        default:           return launch_cfg<128, 32, 4, 1>(p, st);       // 4 waves of 32x32
    }
}

}  // namespace byk
