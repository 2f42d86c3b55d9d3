// mfma_pipe.h -- the ONE definition of the fp32 matrix-pipe software pipeline every GEMM-shaped kernel of this library
// runs (conv_igemm.hip, gemm_stream.hip, wino_fused.hip): block tile BM x BN x 32 staged in LDS, fragment scheme,
// MFMA group, and the per-K-tile schedule that places every auxiliary instruction between two MFMAs.
//
// Design rule (measured, tools/mfma_peak.hip): fp32 MFMA (v_mfma_f32_32x32x2_f32) and the vector ALU share the SIMD's
// FP32 lanes -- a vector-ALU instruction is NOT hidden under an MFMA, global loads / LDS traffic / scalar ALU / waits
// are.  So the steady-state K loop carries no vector-ALU instruction: operands come through buffer loads whose
// addresses are a loop-invariant VGPR + an SGPR offset, every LDS address is a loop-invariant VGPR + an immediate
// (the loop is unrolled x2 so that the double-buffer index is a compile-time constant).
//
// LDS image (bytes): A[2][BM][LD] then B[2][BN][LD], LD = 36 floats per staged row (32 + 4 pad: the ds_write_b128
// staging stores and the ds_read_b128 fragment reads are bank-conflict free).  The K order inside a 32-slice is
// permuted: lane-half h of MFMA step j of quarter q consumes k = 8q + 4h + j, so ONE ds_read_b128 per operand feeds four
// MFMA steps.  The MFMAs compute the TRANSPOSED tile (srcA = B operand, srcB = A operand): in the 32x32 C/D map
//   col = lane & 31 -> A row (pixel),   row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) -> B row (channel)
// so a lane owns one pixel and 4 groups of 4 CONSECUTIVE channels -- NHWC stores are 16-byte vectors.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "byolo_kernels.h"

namespace byk {
namespace pipe {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

static constexpr int BK = 32;
static constexpr int LD = 36;                     // floats per staged row
static constexpr int RSRC_FLAGS = 0x00020000;     // raw buffer, 32-bit data format (gfx9 family)

__device__ __forceinline__ uint32_t fdiv(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shr; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, RSRC_FLAGS);
}
template <int AUX = 0>                                 // AUX: cache-policy bits of the instruction (16 = sc1)
__device__ __forceinline__ f32x4 buffer_load_x4(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}

// Scheduling pattern for one MFMA group: after every MFMA place ceil(aux / N_MFMA) auxiliary instructions, in the
// order global loads -> LDS reads -> LDS writes (LLVM SchedGroupMask: MFMA 0x8, VMEM_READ 0x20, DS_READ 0x100,
// DS_WRITE 0x200).
template <int N_MFMA, int N_VMEM, int N_DSR, int N_DSW>
__device__ __forceinline__ void sched_interleave() {
    constexpr int AUX = N_VMEM + N_DSR + N_DSW;
    constexpr int PER = (AUX + N_MFMA - 1) / N_MFMA;
#pragma unroll
    for (int k = 0; k < N_MFMA; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int q = k * PER + u;
            if (q < N_VMEM) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            else if (q < N_VMEM + N_DSR) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            else if (q < AUX) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    }
}

// Geometry of a block tile and the per-thread LDS addresses.  Block = WM x WN waves; a wave owns TM x TN accumulators
// of 32 x 32; every thread stages the same A_LD rows of A and B_LD rows of B of every K-tile (16 bytes each).
template <int BM_, int BN_, int WM_, int WN_>
struct BlockTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int A_LD = BM * 8 / NT, B_LD = BN * 8 / NT;
    static constexpr int ROWB = LD * 4, A_BUF = BM * ROWB, B_BUF = BN * ROWB, B_BASE = 2 * A_BUF;
    static constexpr int JSTEP = (NT / 8) * ROWB;            // staging rows of one thread are NT/8 apart
    static constexpr int LDS_BYTES = 2 * (BM + BN) * ROWB;
    static constexpr int G = 4 * TM * TN;                    // MFMAs per group (a quarter K-tile) per wave
    static constexpr int NFR = TM + TN;                      // fragment reads per group
    static constexpr int NLD = A_LD + B_LD;                  // 16-byte loads / LDS stores per K-tile per thread
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && B_LD >= 1 && BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile config");

    char* lds;
    int tid, wm, wn, li, lh;          // wave row / column in the block; lane & 31, lane >> 5
    int a_q, a_r;                     // staging: 16-byte column tid % 8, row tid / 8 (+ NT/8 per j)
    int st_off, fa_off, fb_off;

    __device__ __forceinline__ explicit BlockTile(float* smem) {
        lds = reinterpret_cast<char*>(smem);
        tid = threadIdx.x;
        const int wave = tid >> 6, lane = tid & 63;
        wm = wave / WN; wn = wave % WN;
        li = lane & 31; lh = lane >> 5;
        a_q = tid & 7; a_r = tid >> 3;
        st_off = (a_r * LD + a_q * 4) * 4;
        fa_off = ((wm * TM * 32 + li) * LD + lh * 4) * 4;
        fb_off = ((wn * TN * 32 + li) * LD + lh * 4) * 4;
    }
    template <int BUF> __device__ __forceinline__ void store_a(const f32x4 (&a)[A_LD]) const {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) *reinterpret_cast<f32x4*>(lds + st_off + (BUF * A_BUF + j * JSTEP)) = a[j];
    }
    template <int BUF> __device__ __forceinline__ void store_b(const f32x4 (&b)[B_LD]) const {
#pragma unroll
        for (int j = 0; j < B_LD; ++j) *reinterpret_cast<f32x4*>(lds + st_off + (B_BASE + BUF * B_BUF + j * JSTEP)) = b[j];
    }
    template <int BUF, int KQ> __device__ __forceinline__ void read_frags(f32x4 (&af)[TM], f32x4 (&bf)[TN]) const {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(lds + fa_off + (BUF * A_BUF + KQ * 32 + i * 32 * ROWB));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(lds + fb_off + (B_BASE + BUF * B_BUF + KQ * 32 + j * 32 * ROWB));
    }
};

// 4 k-steps x TM x TN MFMAs on one fragment set.  FIRST: the very first MFMA into each accumulator takes an inline-zero
// C operand instead of the accumulator (no vector-ALU clears).
template <int TM, int TN, bool FIRST = false>
__device__ __forceinline__ void mfma_group(f32x16 (&acc)[TM][TN], const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], (FIRST && s == 0) ? zero : acc[i][j], 0, 0, 0);
}

// One K-tile t of the software pipeline; tile t lives in LDS buffer BUF = t & 1, its first fragments already in
// (af0, bf0).  An fp32 MFMA occupies the matrix pipe for 64 cycles and a wave issues in order, so any RUN of non-MFMA
// instructions longer than that lets the pipe drain: the body is branch-free and every auxiliary instruction sits
// BETWEEN two MFMAs.  Per K-tile, 4 groups of G MFMAs per wave:
//   group 0 | [loads0: early global loads, N_LD0 instructions]  LDS fragment reads of group 1
//   group 1 | fragment reads of group 2
//   group 2 | fragment reads of group 3, then (HN) the LDS writes of tile t+1 into the other buffer (N_ST instructions)
//   barrier   (every read of the current buffer is in registers, tile t+1 is visible afterwards)
//   group 3 | [loads3: global loads of tile t+2 into the staging registers just freed, N_LD3 instructions],
//           | (HN) fragment reads of group 0 of tile t+1  -> barrier + LDS latency hide under group 3
// mfma(af, bf, group) issues one group; ABL = timing-ablation bits of the conv build (1: handled by the caller's
// loads, 2: by its store, 4 no barrier, 8 no fragment reads).
template <int BUF, bool HN, int N_LD0, int N_LD3, int N_ST, int ABL = 0, class BT, class Mfma, class L0, class L3, class St>
__device__ __forceinline__ void tile_body(const BT& t, f32x4 (&af0)[BT::TM], f32x4 (&bf0)[BT::TN], f32x4 (&af1)[BT::TM],
                                          f32x4 (&bf1)[BT::TN], Mfma&& mfma, L0&& loads0, L3&& loads3, St&& store_next) {
    constexpr int G = BT::G;
    constexpr bool FR = !(ABL & 8);
    constexpr int NFR = FR ? BT::NFR : 0;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (N_LD0 > 0) loads0();
    if constexpr (FR) t.template read_frags<BUF, 1>(af1, bf1);
    mfma(af0, bf0, 0);
    sched_interleave<G, N_LD0, NFR, 0>();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (FR) t.template read_frags<BUF, 2>(af0, bf0);
    mfma(af1, bf1, 1);
    sched_interleave<G, 0, NFR, 0>();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (FR) t.template read_frags<BUF, 3>(af1, bf1);
    if constexpr (HN) store_next();
    mfma(af0, bf0, 2);
    sched_interleave<G, 0, NFR, HN ? N_ST : 0>();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (!(ABL & 4)) __syncthreads();
    if constexpr (N_LD3 > 0) loads3();
    if constexpr (HN && FR) t.template read_frags<BUF ^ 1, 0>(af0, bf0);
    mfma(af1, bf1, 3);
    sched_interleave<G, N_LD3, HN ? NFR : 0, 0>();
    __builtin_amdgcn_sched_barrier(0);
}


// ---------------------------------------------------------------------------------------------------------------------
// Split-f16 operands ("hi + lo"): the same pipeline on v_mfma_f32_32x32x16_f16, three products per fp32 product.
//
// Every activation / weight element is stored as TWO fp16 values, hi = RNE_f16(x), lo = RNE_f16(x - hi) (x = hi + lo to
// ~23 significant bits; tensors are pre-scaled by powers of two -- byolo_pack.hip -- so that lo stays a normal fp16 for
// every value that matters).  The product
//     x * w  ~=  hi_x hi_w + hi_x lo_w + lo_x hi_w        (lo_x lo_w < 2^-22 |x w| is dropped)
// runs as three MFMAs into ONE fp32 accumulator: products of fp16 values are exact in fp32 and the instruction sums its
// 16 products before one rounding (tools/mfma_f16_probe.hip), so the result is fp32-grade at
// 3/16 of the fp32 MFMA's time per product (measured 2.1 - 2.45 PFLOP/s fp16 = 700 - 800 TFLOP/s fp32-equivalent).
//
// At that rate the LDS (a 16-byte staging store costs 13 cycles of its write path) and the CU's load path (64 bytes per
// clock through the vector L1, tools/l1_l2_bw_probe.hip) are the scarce units, so only the ACTIVATION operand is staged:
//   * activations live in memory in groups of 4 channels, 16 bytes = [hi0 hi1 hi2 hi3 | lo0 lo1 lo2 lo3]: a tensor keeps
//     4 bytes per element and 16 bytes per 4 channels -- every address of the fp32 kernels is unchanged.  LDS image: a
//     staged row = [32 hi | 32 lo] fp16 = 128 bytes + 16 pad (the 144-byte stride of the fp32 image: conflict-free
//     ds_read_b128); a 16-byte global load lands as two 8-byte halves (ds_write2_b64: hi at +8q, lo at +64+8q).  MFMA
//     step s (k = 16 channels) of lane-half h consumes channels 16s + 8h .. +7: ONE ds_read_b128 per plane;
//   * weights are packed on the host in FRAGMENT ORDER, [K-tile][32-channel column block][step][plane][lane][8 fp16]: a
//     wave fetches the operand registers of its column blocks straight from global memory (one coalesced 1 KB
//     buffer_load_dwordx4 per fragment, no LDS), one K-tile ahead, into a second register set.  The waves of a block sit
//     side by side along N (128 x 128: 1 x 4 waves of 128 x 32), so no two waves fetch the same weights.
// Same transposed C/D map as the fp32 kernels (lane = pixel, 4 groups of 4 channels).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

static constexpr int SPLIT_WBLOCK = 4096;          // bytes of one (K-tile, 32-column block) of packed weights

template <int BM_, int BN_, int WM_, int WN_>
struct SplitTile {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NT = 64 * WM * WN;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int A_LD = BM * 8 / NT;
    static constexpr int ROWB = LD * 4, A_BUF = BM * ROWB;
    static constexpr int JSTEP = (NT / 8) * ROWB;
    static constexpr int LDS_BYTES = 2 * A_BUF;
    static constexpr int G = 3 * TM * TN;                    // MFMAs per step (16 channels) per wave
    static constexpr int NFR = 2 * TM;                       // LDS fragment reads per step
    static constexpr int NBF = 4 * TN;                       // weight-fragment loads per K-tile per lane
    static_assert(TM >= 1 && TN >= 1 && A_LD >= 1 && BM * 8 % NT == 0, "tile config");

    char* lds;
    int tid, wm, wn, li, lh;
    int a_q, a_r;
    int st_off, fa_off;
    uint32_t b_voff;                  // this lane's byte offset inside a weight block, + the wave's first column block

    __device__ __forceinline__ explicit SplitTile(float* smem) : SplitTile(smem, (int)threadIdx.x) {}
    // (tid_ given by the caller: a value the compiler cannot see through keeps this bookkeeping from being hoisted above, and kept
    //  alive across, a loop in front of it -- conv_igemm.hip fused_tail)
    __device__ __forceinline__ SplitTile(float* smem, int tid_) {
        lds = reinterpret_cast<char*>(smem);
        tid = tid_;
        const int wave = tid >> 6, lane = tid & 63;
        wm = wave / WN; wn = wave % WN;
        li = lane & 31; lh = lane >> 5;
        // staging row of a thread: the two rows of a 16-lane store group are 4 rows apart (144-byte rows: 16 banks), so the
        // halves of a ds_write2_b64 do not collide (rows r, r + 1 share 12 of their 16 banks: a quarter of the LDS cycles of the 3x3 kernel were bank conflicts; +0.5 %)
        a_q = tid & 7; { const int t = tid >> 3; a_r = (t & ~7) | ((t & 1) << 2) | ((t >> 1) & 3); }
        st_off = a_r * ROWB + a_q * 8;
        fa_off = (wm * TM * 32 + li) * ROWB + lh * 16;
        b_voff = (uint32_t)(lane * 16 + wn * TN * SPLIT_WBLOCK);
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    __device__ __forceinline__ void put(char* at, const f32x4& v) const {        // hi half, lo half: one ds_write2_b64
        *reinterpret_cast<f32x2*>(at) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(at + 64) = f32x2{v[2], v[3]};
    }
    template <int BUF> __device__ __forceinline__ void store_a(const f32x4 (&a)[A_LD]) const {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) put(lds + st_off + (BUF * A_BUF + j * JSTEP), a[j]);
    }
    // activation fragments of step S (0 | 1) of the K-tile in buffer BUF: [0] = hi plane, [1] = lo plane
    template <int BUF, int S> __device__ __forceinline__ void read_frags(f16x8 (&af)[TM][2]) const {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                af[i][h] = *reinterpret_cast<const f16x8*>(lds + fa_off + (BUF * A_BUF + S * 32 + h * 64 + i * 32 * ROWB));
    }
    // weight fragments of one K-tile: [step][column block][plane]; soff = byte offset of (K-tile, column block 0 of the tile)
    __device__ __forceinline__ void load_b(f16x8 (&bf)[2][TN][2], __amdgpu_buffer_rsrc_t rsrc, uint32_t soff) const {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    bf[s][j][h] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, b_voff, soff + (uint32_t)(j * SPLIT_WBLOCK + (s * 2 + h) * 1024), 0));
    }
};

// One step (16 channels): hi*hi, hi*lo, lo*hi of every 32x32 block; product-major, so consecutive MFMAs go to different
// accumulators (a dependent MFMA would wait out the 8 passes of its predecessor).
// FIRST: the first product into each accumulator takes an inline-zero C operand instead of the accumulator -- the accumulators
// need no clearing (the same bits: 0 + p).
template <int TM, int TN, bool FIRST = false>
__device__ __forceinline__ void mfma_step_split(f32x16 (&acc)[TM][TN], const f16x8 (&af)[TM][2], const f16x8 (&bf)[TN][2]) {
    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j][pr == 2 ? 1 : 0], af[i][pr == 1 ? 1 : 0], (FIRST && pr == 0) ? zero : acc[i][j], 0, 0, 0);
}

// One K-tile (32 channels = 2 steps) of the split pipeline; tile t lives in LDS buffer BUF, the activation fragments of
// its step 0 already in af0, its weight fragments in bcur.
//   step 0 | [loads_b: the weight fragments of tile t+1 into the other register set], activation fragment reads of step 1,
//          | then (HN) the LDS writes of tile t+1 into the other buffer
//   barrier  (every read of the current buffer is in registers; tile t+1 is visible afterwards)
//   step 1 | [loads_a: global loads of tile t+2's activations into the staging registers just freed], (HN) fragment reads
//          | of step 0 of tile t+1
// ABL = timing-ablation bits of the conv build (1: handled by the caller's loads, 2: by its store, 4 no barrier, 8 no
// LDS fragment reads).
template <int BUF, bool HN, int N_LDB, int N_LDA, int N_ST, int ABL = 0, bool FIRST = false, class BT, class LB, class LA, class St>
__device__ __forceinline__ void tile_body_split(const BT& t, f32x16 (&acc)[BT::TM][BT::TN], f16x8 (&af0)[BT::TM][2], f16x8 (&af1)[BT::TM][2],
                                                const f16x8 (&bcur)[2][BT::TN][2], LB&& loads_b, LA&& loads_a, St&& store_next) {
    constexpr int G = BT::G;
    constexpr bool FR = !(ABL & 8);
    constexpr int NFR = FR ? BT::NFR : 0;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (N_LDB > 0) loads_b();
    if constexpr (FR) t.template read_frags<BUF, 1>(af1);
    if constexpr (HN) store_next();
    mfma_step_split<BT::TM, BT::TN, FIRST>(acc, af0, bcur[0]);
    sched_interleave<G, N_LDB, NFR, HN ? N_ST : 0>();
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (!(ABL & 4)) __syncthreads();
    if constexpr (N_LDA > 0) loads_a();
    if constexpr (HN && FR) t.template read_frags<BUF ^ 1, 0>(af0);
    mfma_step_split<BT::TM, BT::TN>(acc, af1, bcur[1]);
    sched_interleave<G, N_LDA, HN ? NFR : 0, 0>();
    __builtin_amdgcn_sched_barrier(0);
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride-1 convolutions in split-f16: the three horizontal taps of a filter row share ONE staged activation tile.
//
// The output rows of a tile are CONSECUTIVE pixels, so the operand of tap (ky, kx) is the operand of tap (ky, 1) shifted
// by kx - 1 rows.  A "stage" = (filter row ky, 32-channel chunk): BM + 2 rows (pixels m0 - 1 .. m0 + BM) are staged once,
// and the three K-tiles kx = 0, 1, 2 read their fragments from rows r + kx; a lane whose pixel sits in the first (last)
// image column reads the stage's ZERO row for kx = 0 (2) -- one address register per (block, kx), set up once per tile.
// Activation traffic (global and LDS stores) / 3, and a stage's loads have a K-tile and a half to arrive.  Weights are
// packed in (ky, chunk, kx) order.
template <int BM_, int BN_, int WM_, int WN_>
struct SplitTileKx : SplitTile<BM_, BN_, WM_, WN_> {
    using Base = SplitTile<BM_, BN_, WM_, WN_>;
    using Base::BM; using Base::TM; using Base::A_LD; using Base::ROWB; using Base::JSTEP; using Base::NT;
    static constexpr int ZROW = BM + 2;                      // rows 0 .. BM+1: pixels m0-1 .. m0+BM; row BM+2: zeros
    static constexpr int A_STAGE = (BM + 3) * ROWB;
    static constexpr int LDS_BYTES = 2 * A_STAGE;
    static constexpr int A_LDX = A_LD + 1;                   // + the halo load (two rows per stage; the other threads re-zero the zero row)
    static_assert(NT / 8 >= 2, "two halo rows");

    int sta_off, sth_off;

    __device__ __forceinline__ explicit SplitTileKx(float* smem) : Base(smem) {
        sta_off = (this->a_r + 1) * ROWB + this->a_q * 8;
        sth_off = (this->a_r == 0 ? 0 : (this->a_r == 1 ? BM + 1 : ZROW)) * ROWB + this->a_q * 8;
    }
    template <int BUF> __device__ __forceinline__ void store_stage(const f32x4 (&a)[A_LDX]) const {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) this->put(this->lds + sta_off + (BUF * A_STAGE + j * JSTEP), a[j]);
        this->put(this->lds + sth_off + BUF * A_STAGE, a[A_LD]);
    }
    // activation fragments of step S from stage buffer ABUF at the lane's row addresses `fa` (one per block, for the K-tile's kx)
    template <int ABUF, int S> __device__ __forceinline__ void read_frags_kx(const uint32_t (&fa)[TM], f16x8 (&af)[TM][2]) const {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                af[i][h] = *reinterpret_cast<const f16x8*>(this->lds + fa[i] + (ABUF * A_STAGE + S * 32 + h * 64));
    }
};

}  // namespace pipe
}  // namespace byk
