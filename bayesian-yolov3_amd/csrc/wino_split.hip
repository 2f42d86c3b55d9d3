// wino_split.hip -- Winograd F(2x2, 3x3) in split-f16 arithmetic for the large 3x3 / stride-1 convolutions of the heads
// (lib_yolo/layers.py:545-575 with kernel_size 3; lib_yolo/yolov3.py:543-622: the nine 3x3 convolutions of the three
// detection heads are 62 % of the step at BASELINE configs[3]).
//
// The default precision computes x * w as three fp16 matrix products (mfma_pipe.h), and the matrix pipe runs against the
// socket's power cap: the only way to a faster convolution is FEWER PRODUCTS.  F(2x2, 3x3) spends 16 multiplies per 2x2 output
// tile and channel pair instead of 36:
//
//   V[xi][p][c] = (B^T d B)[xi]        wino_split_input_kernel: one 4x4 input patch per output tile p, fp32 arithmetic on the
//                                      decoded hi + lo values, V stored as hi/lo pairs again (scale 1: |V| <= 4 |d| sits in
//                                      the fp16 range the activations' 4 * value occupies)
//   M[xi]       = V[xi] . U[xi]        wino_split_kernel: per transform point a 64 x 256 x C GEMM (per workgroup) on
//                                      v_mfma_f32_32x32x16_f16 (x_hi u_hi + x_hi u_lo + x_lo u_hi, fp32 accumulation) ...
//   Y           = A^T M A              ... whose accumulators are folded, point after point, into the four 2x2-output
//                                      accumulators with the coefficients of A^T . A (0, +-1); after point 15 the lanes hold
//                                      pre-activation outputs and run the convolution's epilogue (dropout mask, BN, leaky,
//                                      hi/lo encoding, range check).  M never exists in memory.
//   U[xi][c][n] = (G g G^T)[xi]        once, on the host, in double (byolo_finalize); one power-of-two scale per output channel;
//                                      packed per transform point like a 1x1 convolution's weights (fragment order)
//
// B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
//
// The default workgroup (8 waves side by side along N, each 64 x 32) owns 64 output tiles x 256 channels and keeps 4 x 32
// output accumulators + 32 product accumulators per lane (about 230 registers, two waves per SIMD); the template also builds
// 64 x 128 (4 waves) and 128 x 128 (4 waves of 128 x 32: 473 registers, one wave per SIMD; measured slower).  Five accumulator
// sets per tile element pin the tile at 64 x 256 per CU, and its weight-fragment stream (32 KB per 768 matrix-pipe cycles through
// the vector L1) is what bounds the launch: DESIGN.md 3.6.  Numerics: emulated around the oracle before the kernel was built
// (tests/test_split_numerics.py: 0.62 of the bound from float64 where float32 sits at 0.98).
#include <hip/hip_runtime.h>
#include <type_traits>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "mfma_pipe.h"
#include "epilogue.h"

namespace byk {

using namespace pipe;

// Timing ablations (build.py --ablate-ws N -> libbyolo_ws<N>.so, loaded with BYOLO_LIB=...; results are WRONG by construction):
// 1 no weight-fragment loads in the loop, 2 no fold, 4 no activation loads / LDS staging in the loop, 8 no accumulator clears,
// 16 no workgroup barrier in the K-tile body, 32 no LDS fragment reads.
#ifndef BYOLO_WS_ABLATE
#define BYOLO_WS_ABLATE 0
#endif
static constexpr int WS_ABL = BYOLO_WS_ABLATE;

// ---------------------------------------------------------------------------------------------------------------------
// input transform: thread = (tile p, 4 channels): 16 x 16-byte loads, 16 x 16-byte stores; hi/lo groups in and out
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino_split_input_kernel(const WinoParams p) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c4n = (uint32_t)p.C >> 2;
    const uint32_t t = fdiv(gid, p.d_c4), c4 = gid - t * c4n;
    if (t >= (uint32_t)p.P_pad) return;
    float* v = p.v + (size_t)t * p.C + c4 * 4;
    const size_t xi_stride = (size_t)p.P_pad * p.C;
    if (t >= (uint32_t)p.P) {                        // rows that pad the last row tile: zeros (they are multiplied, never stored)
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(v + (size_t)k * xi_stride) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const uint32_t tt = (uint32_t)(p.th * p.tw);
    const uint32_t s = fdiv(t, p.d_tt), r = t - s * tt;
    const uint32_t ty = fdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
    const float* img = p.x + ((size_t)(p.s0 + s) * p.H * p.W) * p.C + c4 * 4;
    const int y0 = 2 * (int)ty - 1, x0 = 2 * (int)tx - 1;
    f32x4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = y0 + i, x = x0 + j;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
            const f32x4 raw = ok ? *reinterpret_cast<const f32x4*>(img + ((size_t)y * p.W + x) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[i][j] = epi::split_decode4(raw);       // ACT_SCALE * value, exact
        }
    f32x4 u[4][4];                                   // B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        u[0][j] = d[0][j] - d[2][j];
        u[1][j] = d[1][j] + d[2][j];
        u[2][j] = d[2][j] - d[1][j];
        u[3][j] = d[1][j] - d[3][j];
    }
    // (B^T d) B, times 1 / ACT_SCALE (p.vmul, a power of two): |V| <= 4 |d| stays inside the range the inputs occupied
    const float m = p.vmul;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 0) * xi_stride) = epi::split_encode4((u[i][0] - u[i][2]) * m);
        *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 1) * xi_stride) = epi::split_encode4((u[i][1] + u[i][2]) * m);
        *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 2) * xi_stride) = epi::split_encode4((u[i][2] - u[i][1]) * m);
        *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 3) * xi_stride) = epi::split_encode4((u[i][1] - u[i][3]) * m);
    }
}

// The same transform, TWO horizontally adjacent output tiles per thread (round 5): their 4x4 patches share two of four columns, so a
// thread loads 4 x 6 pixels instead of 2 x 16 -- the one-tile kernel re-read the input 2.5x through the fabric (measured, FETCH_SIZE:
// 0.68 GB per launch against 0.27 GB of input; every pixel belongs to four patches and the L2 caught a third of the repeats).  Same
// arithmetic per tile, operation for operation: V is bit-identical.  thread = (sample, tile row, tile PAIR, 4 channels); the rows that
// pad V to P_pad are zeroed by the threads behind the last pair.
__global__ __launch_bounds__(256) void wino_split_input2_kernel(const WinoParams p, const FastDiv d_twp, const FastDiv d_ttp, const int twp) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c4n = (uint32_t)p.C >> 2;
    const uint32_t q = fdiv(gid, p.d_c4), c4 = gid - q * c4n;
    const size_t xi_stride = (size_t)p.P_pad * p.C;
    const uint32_t ttp = (uint32_t)(p.th * twp), n_pairs = (uint32_t)(p.P / (p.th * p.tw)) * ttp;
    if (q >= n_pairs) {
        const uint32_t t = (uint32_t)p.P + (q - n_pairs);
        if (t >= (uint32_t)p.P_pad) return;
        float* v = p.v + (size_t)t * p.C + c4 * 4;
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<f32x4*>(v + (size_t)k * xi_stride) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const uint32_t s = fdiv(q, d_ttp), r = q - s * ttp;
    const uint32_t ty = fdiv(r, d_twp), txp = r - ty * (uint32_t)twp;
    const uint32_t tx0 = 2u * txp;
    const bool two = tx0 + 1u < (uint32_t)p.tw;
    const float* img = p.x + ((size_t)(p.s0 + s) * p.H * p.W) * p.C + c4 * 4;
    const int y0 = 2 * (int)ty - 1, x0 = 2 * (int)tx0 - 1;
    f32x4 u[4][6];                                   // B^T d of the six patch columns x0 .. x0 + 5
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        f32x4 d[4];
        const int x = x0 + j;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = y0 + i;
            const bool ok = (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W && (j < 4 || two);
            const f32x4 raw = ok ? *reinterpret_cast<const f32x4*>(img + ((size_t)y * p.W + x) * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
            d[i] = epi::split_decode4(raw);
        }
        u[0][j] = d[0] - d[2];
        u[1][j] = d[1] + d[2];
        u[2][j] = d[2] - d[1];
        u[3][j] = d[1] - d[3];
    }
    const float m = p.vmul;
    const uint32_t t0 = ((s * (uint32_t)p.th + ty) * (uint32_t)p.tw + tx0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (k == 1 && !two) break;
        float* v = p.v + (size_t)(t0 + k) * p.C + c4 * 4;
        const int o = 2 * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 0) * xi_stride) = epi::split_encode4((u[i][o + 0] - u[i][o + 2]) * m);
            *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 1) * xi_stride) = epi::split_encode4((u[i][o + 1] + u[i][o + 2]) * m);
            *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 2) * xi_stride) = epi::split_encode4((u[i][o + 2] - u[i][o + 1]) * m);
            *reinterpret_cast<f32x4*>(v + (size_t)(i * 4 + 3) * xi_stride) = epi::split_encode4((u[i][o + 1] - u[i][o + 3]) * m);
        }
    }
}

hipError_t launch_wino_split_input(const WinoParams& p, hipStream_t st) {
    static const int pair = [] { const char* e = getenv("BYOLO_WINO_IN_PAIR"); return e ? atoi(e) : 1; }();
    if (pair) {
        const int twp = (p.tw + 1) / 2;
        const uint64_t rows = (uint64_t)(p.P / (p.th * p.tw)) * p.th * twp + (uint64_t)(p.P_pad - p.P);
        const uint64_t total = rows * (uint64_t)(p.C >> 2);
        hipLaunchKernelGGL(wino_split_input2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p,
                           make_fastdiv((uint32_t)twp), make_fastdiv((uint32_t)(p.th * twp)), twp);
        return hipGetLastError();
    }
    const uint64_t total = (uint64_t)p.P_pad * (p.C >> 2);
    hipLaunchKernelGGL(wino_split_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// ONE-DIMENSIONAL form, F(2,3) along W (round 5 experiment for the 128-channel 76x76 head convolutions):
// V[xi][(s, yp, j)][c] = (B^T d)[xi] over the four pixels 2j-1 .. 2j+2 of image row yp - 1; rows yp = 0 and yp = H + 1 of every
// sample are zeros (the filter rows above / below the image), so that the GEMM's operand for filter row ky is the SAME row
// sequence ky padded rows further on -- no validity logic in its loader.  V is 2x the input (the 2-D form: 4x).
// thread = (V row, 4 channels): 4 loads, 4 stores.  |V| <= 2 |d|: stored at scale 2 (vmul = 2 / ACT_SCALE), the range the inputs had.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void wino1d_input_kernel(const WinoParams p) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c4n = (uint32_t)p.C >> 2;
    const uint32_t t = fdiv(gid, p.d_c4), c4 = gid - t * c4n;
    if (t >= (uint32_t)p.P_pad) return;
    float* v = p.v + (size_t)t * p.C + c4 * 4;
    const size_t xi_stride = (size_t)p.P_pad * p.C;
    const uint32_t tt = (uint32_t)(p.th * p.tw);
    const uint32_t s = fdiv(t, p.d_tt), r = t - s * tt;
    const uint32_t yp = fdiv(r, p.d_tw), j = r - yp * (uint32_t)p.tw;
    const int y = (int)yp - 1;
    if (t >= (uint32_t)p.P || (unsigned)y >= (unsigned)p.H) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *reinterpret_cast<f32x4*>(v + (size_t)k * xi_stride) = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const float* row = p.x + (((size_t)(p.s0 + s) * p.H + y) * p.W) * p.C + c4 * 4;
    const int x0 = 2 * (int)j - 1;
    f32x4 d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int x = x0 + k;
        const f32x4 raw = (unsigned)x < (unsigned)p.W ? *reinterpret_cast<const f32x4*>(row + (size_t)x * p.C) : f32x4{0.f, 0.f, 0.f, 0.f};
        d[k] = epi::split_decode4(raw);
    }
    const float m = p.vmul;
    *reinterpret_cast<f32x4*>(v + 0 * xi_stride) = epi::split_encode4((d[0] - d[2]) * m);
    *reinterpret_cast<f32x4*>(v + 1 * xi_stride) = epi::split_encode4((d[1] + d[2]) * m);
    *reinterpret_cast<f32x4*>(v + 2 * xi_stride) = epi::split_encode4((d[2] - d[1]) * m);
    *reinterpret_cast<f32x4*>(v + 3 * xi_stride) = epi::split_encode4((d[1] - d[3]) * m);
}

hipError_t launch_wino1d_input(const WinoParams& p, hipStream_t st) {
    const uint64_t total = (uint64_t)p.P_pad * (p.C >> 2);
    hipLaunchKernelGGL(wino1d_input_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// fused GEMM + output transform + epilogue
// ---------------------------------------------------------------------------------------------------------------------
// WINO_BM = output tiles per workgroup (rows of the transform-domain GEMM): 64 -> 230 registers, two workgroups per CU;
// 128 -> 473 registers (the outputs in the accumulator file), one workgroup per CU
// WINO_BN = output channels per workgroup: 128 (4 waves) or 256 (8 waves side by side: ONE workgroup per CU stages a V row for 256
// columns -- the column tiles of a row tile re-read V through the fabric, measured 4.7 GB per launch against 0.8 .. 1.4 GB of V)
// ONED: the one-dimensional form (4 points, K = (filter row, chunk), two outputs per GEMM row; WinoSplitParams.oned)
template <int WINO_BM, int WINO_BN, bool ONED = false>
__global__ __launch_bounds__(WINO_BN * 2, WINO_BM == 64 ? 2 : 1) void wino_split_kernel(const WinoSplitParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    using BT = SplitTile<WINO_BM, WINO_BN, 1, WINO_BN / 32>;
    constexpr int TM = BT::TM, A_LD = BT::A_LD;            // 4 row blocks of 32 per wave; 4 staging rows per thread
    static_assert(BT::TN == 1, "one 32-column block per wave");
    const BT bt(smem);

    // unit -> (row tile, column tile): the column tiles of a row tile are neighbours on one XCD (V rows shared in its L2)
    const int nwg = (int)gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const uint32_t n_tiles = (uint32_t)p.n_tiles;
    const uint32_t unit = (uint32_t)((xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + bi);
    const uint32_t rt = fdiv(unit, p.d_ntiles), ct = unit - rt * n_tiles;

    // ---- staging rows (V row = output tile index inside the chunk; every row of the padded extent exists) -------------
    uint32_t a_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        uint32_t m = rt * (uint32_t)WINO_BM + (uint32_t)bt.a_r + (uint32_t)(BT::NT / 8) * j;
        if constexpr (ONED) {                                  // output pair (s, y, jj) -> V row (s, yp = y [+ ky], jj) of the padded extent:
            const uint32_t sm = fdiv(m, p.d_tt);               // two padded rows per earlier sample; the sample's own first padded row is ky = 0's
            m += 2u * sm * (uint32_t)p.tw;
        }
        a_voff[j] = (m * (uint32_t)p.C + (uint32_t)bt.a_q * 4u) * 4u;
    }
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.v, p.v_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, p.w_bytes);
    // K-tile sequence: (xi, chunk); scalar offsets
    const uint32_t KT = (uint32_t)p.KT;
    const uint32_t w_step = (uint32_t)p.N * BK * 4;                        // one K-tile of all column blocks (N / 32 blocks of 4 KB)
    uint32_t a_soff = 0, a_kt = 0, a_xi_base = 0;                          // the NEXT tile to load
    uint32_t a_ky = 0, a_ky_off = 0;                                       // ONED: filter row of the next tile and its byte offset
    const uint32_t CT = (uint32_t)p.C / BK;                                // ONED: chunks per filter row
    uint32_t w_soff = ct * (WINO_BN / 32) * SPLIT_WBLOCK;
    // V streams from HBM (a chunk is far larger than the caches) and a K-tile of this tile is short (12 MFMAs per wave): the
    // activations of tile t + 1 + NSET are fetched while tile t multiplies, into a ring of NSET staging sets (timing ablation
    // with one set: without the activation path the launch ran 35 % faster -- the loop was waiting for its loads)
    constexpr int NSET = WINO_BM == 64 ? 4 : 2;
    f32x4 a_reg[NSET][A_LD];
    f16x8 bfr[2][2][1][2];
    auto next_tile = [&]() {
        if constexpr (ONED) {
            a_soff = a_xi_base + a_ky_off + a_kt * (BK * 4);
            if (++a_kt == CT) { a_kt = 0; a_ky_off += p.ky_stride; if (++a_ky == 3) { a_ky = 0; a_ky_off = 0; a_xi_base += p.xi_stride; } }
        } else {
            a_soff = a_xi_base + a_kt * (BK * 4);
            if (++a_kt == KT) { a_kt = 0; a_xi_base += p.xi_stride; }      // past point 15: beyond v_bytes -> zeros
        }
    };
    auto load_a = [&](auto set_tag) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[decltype(set_tag)::value][j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
    };
    auto load_b = [&](auto set_tag) { bt.load_b(bfr[decltype(set_tag)::value], w_rsrc, w_soff); w_soff += w_step; };

    constexpr int NOUT = ONED ? 2 : 4, NPT = ONED ? 4 : 16;
    f32x16 Y[NOUT][TM], M[TM][1];
#pragma unroll
    for (int o = 0; o < NOUT; ++o)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) Y[o][i][r] = 0.f;

    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    f16x8 af0[TM][2], af1[TM][2];
    next_tile(); load_a(c0{}); load_b(c0{});
    bt.template store_a<0>(a_reg[0]);
    {   // tiles 1 .. NSET wait in the sets 1 .. NSET - 1, 0
        next_tile(); load_a(std::integral_constant<int, 1 % NSET>{});
        next_tile(); load_a(std::integral_constant<int, 2 % NSET>{});
        if constexpr (NSET == 4) { next_tile(); load_a(std::integral_constant<int, 3>{}); next_tile(); load_a(c0{}); }
    }
    __syncthreads();
    bt.template read_frags<0, 0>(af0);

    // one K-tile in LDS buffer BUF: the uniform body of the split pipeline (mfma_pipe.h tile_body_split) -- stage tile t+1,
    // fetch tile t+2 and the weight fragments of t+1; past the end the loads read zeros and what they stage is never used
    // K-tile t: LDS buffer BUF = t & 1; tile t + 1 is staged from set SET = (t + 1) % NSET, which then receives tile t + 1 + NSET
    auto ktile = [&](auto buf_tag, auto set_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        using ST = decltype(set_tag);
        pipe::tile_body_split<BUF, true, (WS_ABL & 1) ? 0 : BT::NBF, (WS_ABL & 4) ? 0 : A_LD, (WS_ABL & 4) ? 0 : A_LD, ((WS_ABL & 16) ? 4 : 0) | ((WS_ABL & 32) ? 8 : 0)>(
            bt, M, af0, af1, bfr[(WS_ABL & 1) ? 0 : BUF], [&] { load_b(std::integral_constant<int, BUF ^ 1>{}); },
            [&] { next_tile(); load_a(ST{}); }, [&] { if constexpr (!(WS_ABL & 4)) bt.template store_a<BUF ^ 1>(a_reg[ST::value]); });
    };
    auto run_point = [&]() {
        if constexpr (!(WS_ABL & 8)) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) M[i][0][r] = 0.f;
        }
        using c2 = std::integral_constant<int, 2>;
        using c3 = std::integral_constant<int, 3>;
        if constexpr (NSET == 4) { for (uint32_t kt = 0; kt < KT; kt += 4) { ktile(c0{}, c1{}); ktile(c1{}, c2{}); ktile(c0{}, c3{}); ktile(c1{}, c0{}); } }
        else { for (uint32_t kt = 0; kt < KT; kt += 2) { ktile(c0{}, c1{}); ktile(c1{}, c0{}); } }
    };
    // fold M into the four outputs: Y[a][b] += cA(a, i) * cA(b, j) * M for point xi = (i, j), cA = A^T = [1 1 1 0; 0 1 -1 -1].
    // The coefficients (0, +-1) are block-uniform scalars and every point runs the SAME 4 x 32 fused multiply-adds: a switch
    // over 16 specialised folds (36 of the 64 (point, output) pairs are non-zero) made the register allocator copy the output
    // accumulators at the join and spill.
    auto cA = [](int a, int i) -> float { return a == 0 ? (i < 3 ? 1.f : 0.f) : (i == 0 ? 0.f : (i == 1 ? 1.f : -1.f)); };
    for (int xi = 0; xi < NPT; ++xi) {
        run_point();
        const int I = xi >> 2, J = xi & 3;
        if constexpr (WS_ABL & 2) { if (xi != NPT - 1) continue; }
        // as packed fp32 fused multiply-adds (v_pk_fma_f32: two accumulators per instruction, IEEE per element): fp32 vector
        // instructions are paid in full in matrix-pipe time on this part (tools/mfma_valu_coexec_probe.hip), the fold is 128 of them per point
        typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int o = 0; o < NOUT; ++o) {
            const float c = ONED ? cA(o, xi) : cA(o >> 1, I) * cA(o & 1, J);       // ONED: Y = A^T M along W only
            const f32x2 c2 = {c, c};
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 y2 = __builtin_elementwise_fma(c2, f32x2{M[i][0][r], M[i][0][r + 1]}, f32x2{Y[o][i][r], Y[o][i][r + 1]});
                    Y[o][i][r] = y2[0]; Y[o][i][r + 1] = y2[1];
                }
        }
    }

    // ---- epilogue: lane = output tile li (+32 per block), 4 groups of 4 consecutive channels from 4 * lh (mfma_pipe.h) ----
    // Straight-line per dropout mode (0 none, 1 the library's hash, 2 injected bits), decided ONCE: vector instructions are paid in
    // matrix-pipe time, and a block-uniform branch per channel group costs scalar spills and hazard no-ops (conv_igemm.hip finish_plain)
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
        const uint32_t tt = (uint32_t)(p.th * p.tw);
        const int nb = (int)(ct * WINO_BN) + bt.wn * 32 + 4 * bt.lh;
        float vmax = 0.f;
        f32x4 sc4[4], sf4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(p.scale + nb + 8 * g);
            sf4[g] = *reinterpret_cast<const f32x4*>(p.shift + nb + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t t = rt * (uint32_t)WINO_BM + (uint32_t)i * 32u + (uint32_t)bt.li;
            if (t >= (uint32_t)p.P) continue;
            const uint32_t s = fdiv(t, p.d_tt), r = t - s * tt;
            const uint32_t ty = fdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
#pragma unroll
            for (int o = 0; o < NOUT; ++o) {
                const uint32_t oy = ONED ? ty : 2 * ty + (o >> 1), ox = 2 * tx + (ONED ? o : (o & 1));
                if (oy >= (uint32_t)p.H || ox >= (uint32_t)p.W) continue;
                const uint64_t pix = ((uint64_t)(p.s0 + s) * p.H + oy) * p.W + ox;
                const uint64_t idx_row = p.idx_base + pix * (uint64_t)p.N + (uint64_t)nb;
                const epi::DropRow drow(idx_row, p.k1);
                float* d = p.y + (size_t)pix * p.N + nb;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int dn = 8 * g;
                    f32x4 a4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) a4[q] = Y[o][i][4 * g + q];
                    bool keep[4] = {true, true, true, true};
                    if constexpr (MODE == 2) {                              // injected masks (conv_igemm.hip finish_tile)
                        const uint32_t el = drow.el_lo() + (uint32_t)dn;
                        const uint32_t w = p.mask_bits[el >> 5] >> (el & 31u);
#pragma unroll
                        for (int q = 0; q < 4; ++q) keep[q] = (w >> q) & 1u;
                    } else if constexpr (MODE == 1) epi::keep4(drow, dn, p.k0, p.thr, keep);
                    const f32x4 v = epi::bn_act4_pk(a4, sc4[g], sf4[g], keep, slope);
                    vmax = epi::absmax4(vmax, v);
                    *reinterpret_cast<f32x4*>(d + dn) = epi::split_encode4(v);
                }
            }
        }
        if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
    };
    if (!(p.flags & EPI_DROPOUT)) epilogue(std::integral_constant<int, 0>{});
    else if (p.mask_bits) epilogue(std::integral_constant<int, 2>{});
    else epilogue(std::integral_constant<int, 1>{});
}

bool wino_split_ok(int C, int N) { return C >= 128 && (C % 128) == 0 && N >= 128 && (N % 128) == 0; }     // K-tiles in groups of 4

template <int WINO_BM, int WINO_BN, bool ONED = false>
static hipError_t launch_wino_split_bm(const WinoSplitParams& p, hipStream_t st) {
    using BT = SplitTile<WINO_BM, WINO_BN, 1, WINO_BN / 32>;
    auto k = wino_split_kernel<WINO_BM, WINO_BN, ONED>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(k), BT::LDS_BYTES, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3((unsigned)p.units), dim3(WINO_BN * 2), BT::LDS_BYTES, st, p);
    return hipGetLastError();
}
hipError_t launch_wino_split(const WinoSplitParams& p, hipStream_t st) {
    if (p.oned) return (p.bn == 256 && p.bm == 64) ? launch_wino_split_bm<64, 256, true>(p, st) : hipErrorInvalidValue;
    if (p.bn == 256) return p.bm == 64 ? launch_wino_split_bm<64, 256>(p, st) : hipErrorInvalidValue;
    return p.bm == 64 ? launch_wino_split_bm<64, 128>(p, st) : launch_wino_split_bm<128, 128>(p, st);
}

}  // namespace byk
