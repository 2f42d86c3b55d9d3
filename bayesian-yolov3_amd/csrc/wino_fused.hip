// wino_fused.hip -- the Winograd-domain GEMM with the OUTPUT TRANSFORM and the convolution's epilogue fused in:
// M = V * U is never written to memory.
//
// gemm_stream.hip computes M[xi] (4x the size of the layer's output), winograd.hip reads it back, applies
// Y = A^T M A and the epilogue.  At K = Cin = 128 that round trip is most of the traffic of a GEMM that already sits at
// the memory system's edge (43 FLOP/B).  Here a workgroup owns 128 output TILES (2x2 pixels each) x 64 output
// channels and runs, for one row tile after the other, ALL 16 transform points xi through one software pipeline
// (16 * Cin/32 K-tiles per row tile): after the K-tiles of xi its accumulators are folded into the four 2x2-output
// accumulators with the coefficients of A^T . A (0, +1, -1: 36 of the 64 (xi, output) pairs are non-zero), and after
// xi = 15 the lanes hold finished pre-activation outputs: dropout mask, BN scale / shift, leaky, residual, 16-byte NHWC
// stores -- the same epilogue as conv_igemm.hip / winograd.hip.
//
// Cost: the fold (36 packed multiply-adds per xi on average; no clears: the first MFMA of a transform point takes a zero
// C operand, the first fold into an output assigns) and the epilogue run on the SIMDs that multiply, where a vector-ALU
// instruction is paid in matrix-pipe time (tools/mfma_peak.hip): ~10 % at K = 128, ~3 % at K = 512 (ablation table in
// DESIGN.md section 3).  Gain: no M (write + read of 4x the output), no output-transform launch.
// Block = 4 waves as 2 (rows) x 2 (columns), wave tile 64 tiles x 32 channels; LDS image, fragment scheme and
// interleaving as in conv_igemm.hip; persistent grid and XCD placement as in gemm_stream.hip.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "mfma_pipe.h"
#include "epilogue.h"

namespace byk {

using namespace pipe;                             // tile geometry, LDS image, fragment scheme, K-tile schedule: mfma_pipe.h

namespace {

// timing ablations (build.py --ablate-wf N -> libbyolo_wfN.so, loaded with BYOLO_LIB; results are wrong by design):
// 1 no fold, 2 epilogue stores raw Y (no hash / BN / leaky), 4 no epilogue, 8 V loads pinned to the first K-tile rows,
// 16 weight loads pinned to xi = 0, 32 no global loads in the loop, 64 epilogue computes but does not store
#ifndef BYOLO_WF_ABLATE
#define BYOLO_WF_ABLATE 0
#endif
constexpr int WFA = BYOLO_WF_ABLATE;

}  // namespace

__global__ __launch_bounds__(256, 2) void wino_fused_kernel(const WinoFusedParams p) {
    using BT = BlockTile<128, 64, 2, 2>;          // 4 waves of 64 tiles x 32 channels
    constexpr int BM = BT::BM, BN = BT::BN, NT = BT::NT, TM = BT::TM, TN = BT::TN, A_LD = BT::A_LD, B_LD = BT::B_LD;
    static_assert(TM == 2 && TN == 1 && A_LD == 4 && B_LD == 2, "wave tile 64 x 32");
    constexpr int SS_BASE = BT::LDS_BYTES;        // scale[64], shift[64] of this workgroup's channels, after the tile buffers
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BT bt(smem);
    char* lds = bt.lds;
    const int tid = bt.tid;

    // ---- which tiles, which channels (placement as in gemm_stream.hip) ------------------------------------------
    const uint32_t b = blockIdx.x, x = b & 7u, i8 = b >> 3;
    const uint32_t sl = fdiv(i8, p.d_ntiles), tile_n = i8 - sl * (uint32_t)p.n_tiles;
    const uint32_t slot = x * ((uint32_t)p.slots >> 3) + sl;
    if (slot >= (uint32_t)p.slots) return;
    const uint32_t r0 = slot * (uint32_t)p.q + (slot < (uint32_t)p.rem ? slot : (uint32_t)p.rem);   // first row tile
    const int cnt = p.q + (slot < (uint32_t)p.rem ? 1 : 0);
    if (cnt <= 0) return;
    const int KT = p.KT;
    // The epilogue's per-channel constants live in LDS: read from global memory inside the epilogue, every group
    // of 4 channels waited on vmcnt -- which also counts the stores of the group before it -- and the epilogue ran at
    // memory latency (32 round trips per row tile).  (Visible after the pipeline's first barrier.)
    if (tid < 2 * BN)
        reinterpret_cast<float*>(lds + SS_BASE)[tid] = tid < BN ? p.scale[tile_n * BN + tid] : p.shift[tile_n * BN + tid - BN];

    // ---- load stream: K-tile (row tile, xi, chunk), chunk fastest -----------------------------------------------------
    const int a_q = bt.a_q, a_r = bt.a_r;
    uint32_t a_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_voff[j] = (((r0 * BM + a_r + (NT / 8) * j) * (uint32_t)p.C) + a_q * 4) * 4u;
    const uint32_t a_tile_step = (uint32_t)BM * p.C * 4u;
    const uint32_t w_step = (uint32_t)p.N * 32 * 4;
    const uint32_t b_voff = (tile_n * BN * 32 + (uint32_t)tid * 4) * 4;
    uint32_t a_xi_off = 0, w_base = 0, a_soff = 0, w_soff = 0;      // xi part of the V and U offsets (scalar)
    int ld_chunk = 0, ld_xi = 0;
    bool ld_first = true;
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.v, p.v_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, p.w_bytes);
    // staging registers: the V rows of the next K-tile; the weight tiles of the next TWO (the 16 weight matrices of a
    // layer are 2 .. 32 MB per XCD and round -- they come from the Infinity Cache, not from L2, every time)
    f32x4 a_reg[A_LD], b_reg0[B_LD], b_reg1[B_LD];

    auto next_tile = [&]() {
        if constexpr ((WFA & 8) != 0) { a_soff = (uint32_t)ld_chunk * 128u; }
        if constexpr ((WFA & 16) != 0) { w_soff = (uint32_t)ld_chunk * w_step; }
        if constexpr ((WFA & 24) == 24) { if (++ld_chunk == KT) ld_chunk = 0; return; }
        if (ld_chunk == 0 && !ld_first) {
            if (++ld_xi == 16) {                 // next row tile
                ld_xi = 0; a_xi_off = 0; w_base = 0;
#pragma unroll
                for (int j = 0; j < A_LD; ++j) a_voff[j] += (WFA & 8) ? 0u : a_tile_step;
            } else { a_xi_off += p.xi_stride; w_base += p.wstride; }
        }
        ld_first = false;
        if constexpr (!(WFA & 8)) a_soff = a_xi_off + (uint32_t)ld_chunk * 128u;
        if constexpr (!(WFA & 16)) w_soff = w_base + (uint32_t)ld_chunk * w_step;
        if (++ld_chunk == KT) ld_chunk = 0;
    };
    auto load_a = [&]() {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
    };
    auto load_b = [&](f32x4 (&b_reg)[B_LD]) {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) b_reg[j] = buffer_load_x4(w_rsrc, b_voff, w_soff + j * (NT * 16));
    };
    const int wm = bt.wm, wn = bt.wn, li = bt.li, lh = bt.lh;

    f32x16 acc[TM][TN];                           // M[xi] of this wave's 64 tiles x 32 channels (transposed: rows = channels)
    f32x16 Y[4][TM];                              // the four outputs (dy, dx) of every tile
    // Neither accumulator set is ever cleared with vector-ALU moves: the first MFMA of a transform point takes a zero C
    // operand (mfma_group<.., FIRST>), and the first fold into each output assigns instead of adding.

    // Y[dy][dx] += A^T[dy][i] * A^T[dx][j] * M[xi = 4 i + j];   A^T = [1 1 1 0; 0 1 -1 -1]
    auto fold = [&](const int xi) {
        if constexpr ((WFA & 1) != 0) { if (xi != 15) return; }
        const int i = xi >> 2, j = xi & 3;
        const float cy[2] = {i < 3 ? 1.f : 0.f, i == 0 ? 0.f : (i == 1 ? 1.f : -1.f)};
        const float cx[2] = {j < 3 ? 1.f : 0.f, j == 0 ? 0.f : (j == 1 ? 1.f : -1.f)};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float c = cy[dy] * cx[dx];
                if (c != 0.f) {                   // block-uniform
                    // the first transform point that reaches output (dy, dx) is xi = 5 dy ... : (0,0) <- 0, (0,1) <- 1,
                    // (1,0) <- 4, (1,1) <- 5
                    if (xi == 4 * dy + dx) {
#pragma unroll
                        for (int t = 0; t < TM; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) Y[dy * 2 + dx][t][r] = c * acc[t][0][r];
                    } else {
#pragma unroll
                        for (int t = 0; t < TM; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) Y[dy * 2 + dx][t][r] += c * acc[t][0][r];
                    }
                }
            }
    };

    // ---- epilogue of one finished row tile (128 output tiles): lane = tile li (+32 per t), 4 x 4 consecutive channels ----
    const bool do_drop = p.flags & EPI_DROPOUT, do_res = p.flags & EPI_RESIDUAL;
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
    const int nb = (int)(tile_n * BN) + wn * 32 + 4 * lh;          // first channel of this lane's group g = 0
    const uint32_t tt = (uint32_t)(p.th * p.tw);
    auto epilogue = [&](const uint32_t row_tile) {
        if constexpr ((WFA & 4) != 0) { if (p.P >= 0) return; }
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const uint32_t tile = row_tile * BM + wm * TM * 32 + t * 32 + li;
            if (tile < (uint32_t)p.P) {
                const uint32_t s = fdiv(tile, p.d_tt), r = tile - s * tt;
                const uint32_t ty = fdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const uint32_t oy = 2 * ty + (o >> 1), ox = 2 * tx + (o & 1);
                    if (oy >= (uint32_t)p.H || ox >= (uint32_t)p.W) continue;
                    const uint64_t pix = ((uint64_t)(p.s0 + s) * p.H + oy) * p.W + ox;
                    const size_t off = (size_t)pix * p.N + nb;
                    // pair index of the pixel's first element here and the key word of its high half: once per pixel
                    const epi::DropRow drow(p.idx_base + pix * (uint64_t)p.N + (uint64_t)nb, p.k1);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if constexpr ((WFA & 2) != 0) {
                            f32x4 raw;
#pragma unroll
                            for (int q = 0; q < 4; ++q) raw[q] = Y[o][t][4 * g + q];
                            *reinterpret_cast<f32x4*>(p.y + off + 8 * g) = raw;
                            continue;
                        }
                        // scale (x 1 / (1 - p) with the masks on: host) and shift of channels n0 .. n0 + 3
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(lds + SS_BASE + (wn * 32 + 4 * lh + 8 * g) * 4);
                        const f32x4 sf = *reinterpret_cast<const f32x4*>(lds + SS_BASE + (BN + wn * 32 + 4 * lh + 8 * g) * 4);
                        bool keep[4] = {true, true, true, true};
                        if (do_drop) epi::keep4(drow, 8 * g, p.k0, p.thr, keep);
                        f32x4 y4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) y4[q] = Y[o][t][4 * g + q];
                        f32x4 v = epi::bn_act4(y4, sc, sf, keep, slope);
                        if (do_res) v += *reinterpret_cast<const f32x4*>(p.residual + off + 8 * g);
                        if constexpr ((WFA & 64) != 0) { if (p.P >= 0) continue; }
                        *reinterpret_cast<f32x4*>(p.y + off + 8 * g) = v;
                    }
                }
            }
        }
    };

    // ---- the pipeline (mfma_pipe.h) ---------------------------------------------------------------------------------
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using yes = std::true_type;
    using no = std::false_type;
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    next_tile(); load_a(); load_b(b_reg0); bt.template store_a<0>(a_reg); bt.template store_b<0>(b_reg0);
    next_tile(); load_a(); load_b(b_reg1);        // total >= 16 * KT >= 32
    __syncthreads();
    bt.template read_frags<0, 0>(af0, bf0);

    // tile t+1's weights wait in set (t+1) & 1; tile t+2's are fetched into set t & 1 in group 0 already, its V rows in
    // group 3 (into the staging registers the LDS write of group 2 has just freed)
    auto tile_body = [&](auto buf_tag, auto first_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool FIRST = decltype(first_tag)::value, LD = !(WFA & 32);
        f32x4 (&b_far)[B_LD] = BUF == 0 ? b_reg0 : b_reg1;
        f32x4 (&b_near)[B_LD] = BUF == 0 ? b_reg1 : b_reg0;
        pipe::tile_body<BUF, true, LD ? B_LD : 0, LD ? A_LD : 0, BT::NLD>(
            bt, af0, bf0, af1, bf1,
            [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], int group) {
                if (FIRST && group == 0) mfma_group<TM, TN, true>(acc, af, bf);      // first MFMAs of a unit start from zero
                else mfma_group<TM, TN, false>(acc, af, bf);
            },
            [&] { load_b(b_far); }, load_a,
            [&] { bt.template store_a<BUF ^ 1>(a_reg); bt.template store_b<BUF ^ 1>(b_near); });
    };

    // One unit = one transform point of one row tile = KT K-tiles (KT is even: tiles go in pairs over the two LDS
    // buffers).  Every tile prefetches the tile two ahead, also the last two of the stream: those loads run past the
    // slot's rows (the next slot's first tile, or beyond the buffer where the bounds check returns 0) and are never used.
    const int half = KT >> 1, units = cnt * 16;
    int xi = 0;
    uint32_t row_tile = r0;
    for (int u = 0; u < units; ++u) {
        next_tile(); tile_body(c0{}, yes{});                          // first MFMAs of the unit start from zero
        next_tile(); tile_body(c1{}, no{});
        for (int pr = 1; pr < half; ++pr) {
            next_tile(); tile_body(c0{}, no{});
            next_tile(); tile_body(c1{}, no{});
        }
        fold(xi);
        if (++xi == 16) { xi = 0; epilogue(row_tile); ++row_tile; }
    }
}

hipError_t launch_wino_fused(const WinoFusedParams& p, hipStream_t st) {
    constexpr size_t lds = BlockTile<128, 64, 2, 2>::LDS_BYTES + 2 * 64 * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(wino_fused_kernel), lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(wino_fused_kernel, dim3(512), dim3(256), lds, st, p);
    return hipGetLastError();
}

bool wino_fused_ok(int C, int N) {
    const int nt = N / 64;
    return (N % 64) == 0 && (C % 64) == 0 && (nt == 1 || nt == 2 || nt == 4 || nt == 8 || nt == 16 || nt == 32);
}

}  // namespace byk
