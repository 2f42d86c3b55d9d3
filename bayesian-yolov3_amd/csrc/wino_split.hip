// wino_split.hip -- Winograd F(2x2, 3x3) in split-f16 arithmetic for the large 3x3 / stride-1 convolutions of the heads
// (lib_yolo/layers.py:545-575 with kernel_size 3; lib_yolo/yolov3.py:543-622: the nine 3x3 convolutions of the three
// detection heads are 62 % of the step at BASELINE configs[3]).
//
// The default precision computes x * w as three fp16 matrix products (mfma_pipe.h), and the matrix pipe runs against the
// socket's power cap: the only way to a faster convolution is FEWER PRODUCTS.  F(2x2, 3x3) spends 16 multiplies per 2x2 output
// tile and channel pair instead of 36:
//
//   V[xi][p][c] = (B^T d B)[xi]        wino_split_input2_kernel: one 4x4 input patch per output tile p, fp32 arithmetic on the
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
// 64 x 128 (4 waves) for layers whose output channels are not a multiple of 256.  Five accumulator sets per tile element pin the
// tile at 64 x 256 per CU, and its weight-fragment stream (32 KB per 768 matrix-pipe cycles through the vector L1) is what bounds
// the launch: DESIGN.md section 3.  (Rounds 3 - 5 also carried a 128 x 128 workgroup -- 473 registers, one wave per SIMD, 298 against
// 310 img/s --, a one-tile input transform and a one-dimensional F(2,3) form for the 128-channel layers, transform + GEMM 2.12 ms
// against the direct kernel's 2.01: all three lost their A/Bs and are gone; profiles/HISTORY.md, profiles/r5_wino1d.md.)
//
// Round 6: the workgroups can WALK the unit list (grid = resident workgroups; byolo_plan_opts.wino_split_persist = 1 | 2; VERDICT r5
// item 1).  A unit = 64 output tiles x WINO_BN channels through all 16 points; the V stream of a workgroup then runs on across its
// units: while unit u multiplies its last K-tiles, the V rows of unit u + 1's first tiles are already being fetched, staged and
// read, so that the pipeline fill and the workgroup turn-over (dispatch, kernel arguments, address prologue: one workgroup per CU,
// nothing covers it) are paid once per workgroup instead of once per unit.  Same K order, same arithmetic per output element: the
// rows are the rows of the one-unit-per-workgroup launch bit for bit (tools/rows_digest.py).  MEASURED SLOWER, so it is not the
// default (profiles/r6_wino_persist.md; same box, three interleaved runs each, ms per step of the six launches): one unit per
// workgroup 7.126, units claimed from a per-XCD counter (2) 7.180 (+0.8 %), a static list (1) 7.304 (+2.5 %).  The hardware
// dispatcher already hands the next unit to whichever CU is free -- units differ (edge tiles store 1 - 3 of their 4 outputs) and
// CUs do not run in step, so a static list runs at the pace of its unluckiest workgroup -- and the ~3 us of fill + turn-over it
// pays per unit are less than what a walking workgroup loses: its weight-fragment stream restarts behind every epilogue anyway
// (32 registers that cannot live through it), and the epilogue runs with the prefetched V sets and the stream's scalars live.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "mfma_pipe.h"
#include "epilogue.h"

namespace byk {

using namespace pipe;

// Timing ablations (build.py --ablate-ws N -> libbyolo_ws<N>.so, loaded with BYOLO_LIB=...; results are WRONG by construction):
// 1 no weight-fragment loads in the loop, 2 no fold, 4 no activation loads / LDS staging in the loop, (8: gone with the clears),
// 16 no workgroup barrier in the K-tile body, 32 no LDS fragment reads.
#ifndef BYOLO_WS_ABLATE
#define BYOLO_WS_ABLATE 0
#endif
static constexpr int WS_ABL = BYOLO_WS_ABLATE;
#ifndef BYOLO_WS_FOLD_SKIP                               // 0: the A/B build that runs all 64 (point, output) folds
#define BYOLO_WS_FOLD_SKIP 1
#endif
static constexpr bool WS_FOLD_SKIP = BYOLO_WS_FOLD_SKIP != 0;
#ifndef BYOLO_WS_RES_AHEAD                               // residual epilogue: outputs the fetch runs ahead (2: 255 registers, measured +-0)
#define BYOLO_WS_RES_AHEAD 1
#endif

// ---------------------------------------------------------------------------------------------------------------------
// input transform: thread = (sample, tile row, tile PAIR, 4 channels): 4 x 6 16-byte loads, 2 x 16 16-byte stores; hi/lo groups in
// and out.  TWO horizontally adjacent output tiles per thread (round 5): their 4x4 patches share two of four columns -- a one-tile
// thread re-read the input 2.5x through the fabric (measured, FETCH_SIZE: 0.68 GB per launch against 0.27 GB of input; every pixel
// belongs to four patches and the L2 caught a third of the repeats).  The rows that pad V to P_pad are zeroed by the threads behind
// the last pair.
// ---------------------------------------------------------------------------------------------------------------------
// V is written with NON-TEMPORAL stores (`global_store_dwordx4 ... nt`: a wave writes 1 KB runs, whole cache lines): the transform itself
// is 1 % faster and the GEMM that reads V 1.3 - 5 % (config 4: 6.98 -> 6.90 ms per step over the six launches, +0.7 % img/s; the 1024 x 1920
// workloads 6.46 -> 6.24; also where V would fit the Infinity Cache) -- profiles/r6_wino_small.md.  The same hint on the GEMM's OUTPUT
// stores, which complete a 128-byte line over four instructions of two lanes, costs 23 % of that kernel.  0 = the A/B build.
// Layout of V inside a transform point's plane.  0: [tile][channel] (a row = one output tile, C channels).  1: K-TILE MAJOR -- [row tile of 64
// output tiles][K-tile of 32 channels][64 rows][32 channels]: what a workgroup of the GEMM stages per K-tile is one contiguous 8 KB block
// instead of 64 pieces of 128 bytes 4 C bytes apart (fewer DRAM page openings per byte).  Same rows; one box, three interleaved runs each:
// config 4 381.5 -> 383.3 img/s, config 7 459.5 -> 461.7 (+0.5 % both), the GEMM's launches -0.6 % / -1.6 %, the transform's unchanged
// (profiles/r6_wino_small.md).  0 = the A/B build.
#ifndef BYOLO_WS_V_KTMAJOR
#define BYOLO_WS_V_KTMAJOR 1
#endif
__device__ __forceinline__ size_t v_index(uint32_t t, uint32_t c, uint32_t C) {          // float index of (output tile t, channel c) inside a plane
    if constexpr (BYOLO_WS_V_KTMAJOR != 0) return ((size_t)((t >> 6) * (C >> 5) + (c >> 5)) * 64u + (t & 63u)) * 32u + (c & 31u);
    else return (size_t)t * C + c;
}
#ifndef BYOLO_WS_IN_COLMAJOR
#define BYOLO_WS_IN_COLMAJOR 1
#endif
#ifndef BYOLO_WS_NT_STORE
#define BYOLO_WS_NT_STORE 1
#endif
__device__ __forceinline__ void vstore(float* at, const f32x4 v) {
    if constexpr (BYOLO_WS_NT_STORE != 0) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(at));
    else *reinterpret_cast<f32x4*>(at) = v;
}
__global__ __launch_bounds__(256) void wino_split_input2_kernel(const WinoParams p, const FastDiv d_twp, const FastDiv d_ttp, const int twp, const FastDiv d_th) {
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x;
    const uint32_t c4n = (uint32_t)p.C >> 2;
    const uint32_t q = fdiv(gid, p.d_c4), c4 = gid - q * c4n;
    const size_t xi_stride = (size_t)p.P_pad * p.C;
    const uint32_t ttp = (uint32_t)(p.th * twp), n_pairs = (uint32_t)(p.P / (p.th * p.tw)) * ttp;
    if (q >= n_pairs) {
        const uint32_t t = (uint32_t)p.P + (q - n_pairs);
        if (t >= (uint32_t)p.P_pad) return;
        float* v = p.v + v_index(t, c4 * 4, (uint32_t)p.C);
#pragma unroll
        for (int k = 0; k < 16; ++k) vstore(v + (size_t)k * xi_stride, f32x4{0.f, 0.f, 0.f, 0.f});
        return;
    }
    const uint32_t s = fdiv(q, d_ttp), r = q - s * ttp;
    // BYOLO_WS_IN_COLMAJOR: neighbouring threads take VERTICALLY adjacent tile pairs (their 4-row patches share two rows, half of what a
    // thread reads) instead of horizontally adjacent ones (two of six columns): the transform's launches 1.594 -> 1.561 ms per step at
    // config 4, 1.160 -> 1.127 at config 7 (one box, three interleaved runs each; img/s +0.15 % / +-0); 0 = the A/B build
    uint32_t ty, txp;
    if constexpr (BYOLO_WS_IN_COLMAJOR != 0) { txp = fdiv(r, d_th); ty = r - txp * (uint32_t)p.th; }
    else { ty = fdiv(r, d_twp); txp = r - ty * (uint32_t)twp; }
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
        float* v = p.v + v_index(t0 + (uint32_t)k, c4 * 4, (uint32_t)p.C);
        const int o = 2 * k;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            vstore(v + (size_t)(i * 4 + 0) * xi_stride, epi::split_encode4((u[i][o + 0] - u[i][o + 2]) * m));
            vstore(v + (size_t)(i * 4 + 1) * xi_stride, epi::split_encode4((u[i][o + 1] + u[i][o + 2]) * m));
            vstore(v + (size_t)(i * 4 + 2) * xi_stride, epi::split_encode4((u[i][o + 2] - u[i][o + 1]) * m));
            vstore(v + (size_t)(i * 4 + 3) * xi_stride, epi::split_encode4((u[i][o + 1] - u[i][o + 3]) * m));
        }
    }
}

hipError_t launch_wino_split_input(const WinoParams& p, hipStream_t st) {
    const int twp = (p.tw + 1) / 2;
    const uint64_t rows = (uint64_t)(p.P / (p.th * p.tw)) * p.th * twp + (uint64_t)(p.P_pad - p.P);
    const uint64_t total = rows * (uint64_t)(p.C >> 2);
    hipLaunchKernelGGL(wino_split_input2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p,
                       make_fastdiv((uint32_t)twp), make_fastdiv((uint32_t)(p.th * twp)), twp, make_fastdiv((uint32_t)p.th));
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// fused GEMM + output transform + epilogue
// ---------------------------------------------------------------------------------------------------------------------
// coefficient of point xi = (i, j) in output o = (a, b): cA(a, i) * cA(b, j), cA = A^T = [1 1 1 0; 0 1 -1 -1]
__host__ __device__ constexpr int wino_cA(int a, int i) { return a == 0 ? (i < 3 ? 1 : 0) : (i == 0 ? 0 : (i == 1 ? 1 : -1)); }

// WINO_BN = output channels per workgroup: 256 (8 waves side by side: ONE workgroup per CU stages a V row for 256 columns -- the
// column tiles of a row tile re-read V through the fabric, measured 4.7 GB per launch against 0.8 .. 1.4 GB of V with 128) or 128
// (4 waves, two workgroups per CU).  64 output tiles per workgroup (rows of the transform-domain GEMM): 230 registers.
template <int WINO_BN>
__global__ __launch_bounds__(WINO_BN * 2, 2) void wino_split_kernel(const WinoSplitParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ uint32_t s_claim;                          // persist == 2: the unit this workgroup has claimed for its next turn
    constexpr int WINO_BM = 64;
    using BT = SplitTile<WINO_BM, WINO_BN, 1, WINO_BN / 32>;
    constexpr int TM = BT::TM, A_LD = BT::A_LD;            // 2 row blocks of 32 per wave; 1 (8 waves) or 2 staging rows per thread
    static_assert(BT::TN == 1, "one 32-column block per wave");
    const BT bt(smem);

    // ---- this workgroup's units.  Workgroup b sits on XCD b & 7 (round-robin dispatch); the units are dealt out to the XCDs in
    // contiguous ranges, and inside an XCD unit x0 + k goes to the XCD's workgroup k mod (its workgroups): what the workgroups of one
    // XCD multiply at the same time are neighbouring units = the column tiles of a few row tiles (their V rows shared in that L2).
    // One unit per workgroup (grid = units): the same code, the list has one entry.
    const uint32_t nwg = gridDim.x, xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t n_units = (uint32_t)p.units, uq = n_units >> 3, ur = n_units & 7u;
    const uint32_t x0 = xcd * uq + (xcd < ur ? xcd : ur), xn = uq + (xcd < ur ? 1u : 0u);
    const uint32_t ustep = (nwg >> 3) + (xcd < (nwg & 7u) ? 1u : 0u);
    if (slot >= xn) return;
    const uint32_t n_tiles = (uint32_t)p.n_tiles;
    auto unit_of = [&](uint32_t k, uint32_t& rt, uint32_t& ct) __attribute__((always_inline)) { const uint32_t u = x0 + k; rt = fdiv(u, p.d_ntiles); ct = u - rt * n_tiles; };
    // staging rows (V row = output tile index inside the chunk; every row of the padded extent exists)
    auto voff_of = [&](uint32_t rt, uint32_t (&v)[A_LD]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const uint32_t row = (uint32_t)bt.a_r + (uint32_t)(BT::NT / 8) * j;
            if constexpr (BYOLO_WS_V_KTMAJOR != 0) v[j] = rt * (uint32_t)WINO_BM * (uint32_t)p.C * 4u + (row * 32u + (uint32_t)bt.a_q * 4u) * 4u;
            else v[j] = ((rt * (uint32_t)WINO_BM + row) * (uint32_t)p.C + (uint32_t)bt.a_q * 4u) * 4u;
        }
    };
    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.v, p.v_bytes);
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, p.w_bytes);
    const uint32_t KT = (uint32_t)p.KT;
    const uint32_t w_step = (uint32_t)p.N * BK * 4;                        // one K-tile of all column blocks (N / 32 blocks of 4 KB)

    // ---- the two operand streams, each one sequence over ALL units of this workgroup: K-tile order (unit, point, chunk).
    // V streams from HBM (a chunk is far larger than the caches) and a K-tile of this tile is short (12 MFMAs per wave): the
    // activations of tile t + 1 + NSET are fetched while tile t multiplies, into a ring of NSET staging sets (timing ablation
    // with one set: without the activation path the launch ran 35 % faster -- the loop was waiting for its loads); the weight
    // fragments of tile t + 1.  Past a unit's last tile the streams go on with the NEXT unit's first tiles; past the last unit's, with
    // addresses beyond the buffers' extents -> zeros, staged and never multiplied.
    constexpr int NSET = WINO_BN == 256 ? 4 : 2;       // (the 4-wave workgroup stages two rows per thread and sits at 256 registers: a ring of two)
    f32x4 a_reg[NSET][A_LD];
    f16x8 bfr[2][2][1][2];
    uint32_t a_voff[A_LD];                                                 // a lane's staging rows inside a row tile (the same for every unit)
    voff_of(0u, a_voff);
    // V stream, all scalar and branch-free (a branch inside the K loop splits the accumulators' live ranges: the first build with one
    // spilled 850 bytes per lane): the NEXT tile to load = (row tile's byte offset) + (point's) + (chunk's); when the unit's last tile
    // has been issued the stream moves to the row tile of this workgroup's next unit, which the unit loop has looked up (a_next_base)
    uint32_t a_base, a_soff, a_kt = 0, a_pt = 0, a_xi_base = 0, a_next_base = 0, a_has_next = 0;
    uint32_t w_soff = 0;                                                   // U stream: the next tile's offset (set at every unit's start)
    const uint32_t rt_bytes = (uint32_t)WINO_BM * (uint32_t)p.C * 4u;
    { uint32_t rt, ct; unit_of(slot, rt, ct); a_base = rt * rt_bytes; a_soff = a_base; }
    auto load_a = [&](auto set_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[decltype(set_tag)::value][j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
        ++a_kt;
        const uint32_t wrap = a_kt == KT ? 1u : 0u;
        a_kt = wrap ? 0u : a_kt;
        a_xi_base += wrap ? p.xi_stride : 0u;
        a_pt += wrap;
        const uint32_t roll = (a_pt == 16u ? 1u : 0u) & a_has_next;        // (without a next unit: on beyond V's extent, never multiplied)
        a_pt = roll ? 0u : a_pt;
        a_xi_base = roll ? 0u : a_xi_base;
        a_base = roll ? a_next_base : a_base;
        a_soff = a_base + a_xi_base + a_kt * (BYOLO_WS_V_KTMAJOR != 0 ? WINO_BM * BK * 4 : BK * 4);
    };
    // (the weight fragments do NOT run on across units: 32 registers that would have to live through the epilogue, which has none to
    //  spare -- the build with them spilled 1.2 KB per lane; a unit's first fragments are fetched behind the previous unit's epilogue:
    //  one L2 round trip per unit.  What a unit's last K-tile fetches lies beyond U's extent: zeros, never used.)
    auto load_b = [&](auto set_tag) __attribute__((always_inline)) { bt.load_b(bfr[decltype(set_tag)::value], w_rsrc, w_soff); w_soff += w_step; };

    f32x16 Y[4][TM], M[TM][1];
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    using c3 = std::integral_constant<int, 3>;
    f16x8 af0[TM][2], af1[TM][2];
    load_a(c0{});
    bt.template store_a<0>(a_reg[0]);
    if constexpr (NSET == 4) { load_a(c1{}); load_a(c2{}); load_a(c3{}); load_a(c0{}); }       // tiles 1 .. NSET wait in the sets 1 .. NSET - 1, 0
    else { load_a(c1{}); load_a(c0{}); }
    __syncthreads();

    // one K-tile in LDS buffer BUF: the uniform body of the split pipeline (mfma_pipe.h tile_body_split) -- stage tile t+1,
    // fetch tile t+1+NSET and the weight fragments of t+1
    // K-tile t: LDS buffer BUF = t & 1; tile t + 1 is staged from set SET = (t + 1) % NSET, which then receives tile t + 1 + NSET
    auto ktile = [&](auto buf_tag, auto set_tag, auto first_tag) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_tag)::value;
        using ST = decltype(set_tag);
        pipe::tile_body_split<BUF, true, (WS_ABL & 1) ? 0 : BT::NBF, (WS_ABL & 4) ? 0 : A_LD, (WS_ABL & 4) ? 0 : A_LD, ((WS_ABL & 16) ? 4 : 0) | ((WS_ABL & 32) ? 8 : 0), decltype(first_tag)::value>(
            bt, M, af0, af1, bfr[(WS_ABL & 1) ? 0 : BUF], [&]() __attribute__((always_inline)) { load_b(std::integral_constant<int, BUF ^ 1>{}); },
            [&]() __attribute__((always_inline)) { load_a(ST{}); }, [&]() __attribute__((always_inline)) { if constexpr (!(WS_ABL & 4)) bt.template store_a<BUF ^ 1>(a_reg[ST::value]); });
    };
    // a point's K loop; its first K-tile's first products start the product accumulators from an inline zero (no clears: 32 vector
    // moves per point that the matrix pipe would wait for)
    auto run_point = [&]() __attribute__((always_inline)) {
        constexpr std::true_type first{};
        constexpr std::false_type later{};
        if constexpr (NSET == 4) {
            ktile(c0{}, c1{}, first); ktile(c1{}, c2{}, later); ktile(c0{}, c3{}, later); ktile(c1{}, c0{}, later);
            for (uint32_t kt = 4; kt < KT; kt += 4) { ktile(c0{}, c1{}, later); ktile(c1{}, c2{}, later); ktile(c0{}, c3{}, later); ktile(c1{}, c0{}, later); }
        } else {
            ktile(c0{}, c1{}, first); ktile(c1{}, c0{}, later);
            for (uint32_t kt = 2; kt < KT; kt += 2) { ktile(c0{}, c1{}, later); ktile(c1{}, c0{}, later); }
        }
    };
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    // ---- epilogue: lane = output tile li (+32 per block), 4 groups of 4 consecutive channels from 4 * lh (mfma_pipe.h) ----
    // Straight-line per dropout mode (0 none, 1 the library's hash, 2 injected bits), decided ONCE: vector instructions are paid in
    // matrix-pipe time, and a block-uniform branch per channel group costs scalar spills and hazard no-ops (conv_igemm.hip finish_plain)
    auto epilogue = [&](auto mode_tag, uint32_t rt, uint32_t ct) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
        const uint32_t tt = (uint32_t)(p.th * p.tw);
        // the lane's position from a thread index the compiler cannot see through: otherwise everything the epilogue derives from it
        // is hoisted out of the unit loop and lives in registers through all K loops, which have none to spare (896 spilled bytes)
        int tid_e = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int e_wn = tid_e >> 6, e_li = tid_e & 31, e_lh = (tid_e >> 5) & 1;
        const int nb = (int)(ct * WINO_BN) + e_wn * 32 + 4 * e_lh;
        float vmax = 0.f;
        f32x4 sc4[4], sf4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(p.scale + nb + 8 * g);
            sf4[g] = *reinterpret_cast<const f32x4*>(p.shift + nb + 8 * g);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t t = rt * (uint32_t)WINO_BM + (uint32_t)i * 32u + (uint32_t)e_li;
            if (t >= (uint32_t)p.P) continue;
            const uint32_t s = fdiv(t, p.d_tt), r = t - s * tt;
            const uint32_t ty = fdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const uint32_t oy = 2 * ty + (o >> 1), ox = 2 * tx + (o & 1);
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

    // ---- the same with a residual added after the activation and no dropout (the darknet blocks' `inputs + shortcut`, lib_yolo/layers.py:
    // 505-507; byolo_plan.hip sends no layer with both here).  Straight line: every lane computes all of its 2 x 4 outputs at clamped
    // addresses and only the STORE is predicated (a lane's `continue` saves nothing while its wave runs on), so that the residual groups
    // of output o + 1 can be fetched, without a branch, as soon as those of output o have been added: a ring of four 16-byte groups,
    // one output's arithmetic ahead of its use.  Measured on 1024 x 1920 x 11 frames (profiles/r6_wino_small.md): the residual costs
    // 0.07 ms per launch of 173 MB (+22 % at 256 -> 512 channels, +8 % at 512 -> 1024) whatever the distance of the fetch -- all CUs
    // reach their epilogues together and the round's 67 MB of residual reads, which block, join its 67 MB of writes, which do not.
    auto epilogue_res = [&](uint32_t rt, uint32_t ct) __attribute__((always_inline)) {
        const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
        const uint32_t tt = (uint32_t)(p.th * p.tw);
        int tid_e = (int)threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int e_wn = tid_e >> 6, e_li = tid_e & 31, e_lh = (tid_e >> 5) & 1;
        const int nb = (int)(ct * WINO_BN) + e_wn * 32 + 4 * e_lh;
        float vmax = 0.f;
        f32x4 sc4[4], sf4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            sc4[g] = *reinterpret_cast<const f32x4*>(p.scale + nb + 8 * g);
            sf4[g] = *reinterpret_cast<const f32x4*>(p.shift + nb + 8 * g);
        }
        const bool keep[4] = {true, true, true, true};
        auto row_of = [&](int k, bool& ok) __attribute__((always_inline)) -> size_t {        // k = 4 i + o
            const uint32_t t0 = rt * (uint32_t)WINO_BM + (uint32_t)(k >> 2) * 32u + (uint32_t)e_li;
            const uint32_t t = t0 < (uint32_t)p.P ? t0 : (uint32_t)p.P - 1u;
            const uint32_t s = fdiv(t, p.d_tt), r = t - s * tt;
            const uint32_t ty = fdiv(r, p.d_tw), tx = r - ty * (uint32_t)p.tw;
            const uint32_t oy0 = 2 * ty + (uint32_t)((k >> 1) & 1), ox0 = 2 * tx + (uint32_t)(k & 1);
            ok = t0 < (uint32_t)p.P && oy0 < (uint32_t)p.H && ox0 < (uint32_t)p.W;
            const uint32_t oy = oy0 < (uint32_t)p.H ? oy0 : (uint32_t)p.H - 1u, ox = ox0 < (uint32_t)p.W ? ox0 : (uint32_t)p.W - 1u;
            return (size_t)(((uint64_t)(p.s0 + s) * p.H + oy) * p.W + ox) * p.N + nb;
        };
        constexpr int AHEAD = BYOLO_WS_RES_AHEAD;          // outputs the residual fetch runs ahead of the arithmetic
        f32x4 rr4[AHEAD][4];
        bool okv[AHEAD + 1]; size_t rowv[AHEAD + 1];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) {
            rowv[a] = row_of(a, okv[a]);
#pragma unroll
            for (int g = 0; g < 4; ++g) rr4[a][g] = *reinterpret_cast<const f32x4*>(p.residual + rowv[a] + 8 * g);
        }
#pragma unroll
        for (int k = 0; k < 4 * TM; ++k) {
            if (k + AHEAD < 4 * TM) rowv[AHEAD] = row_of(k + AHEAD, okv[AHEAD]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a4;
#pragma unroll
                for (int q = 0; q < 4; ++q) a4[q] = Y[k & 3][k >> 2][4 * g + q];
                f32x4 v = epi::bn_act4_pk(a4, sc4[g], sf4[g], keep, slope);
                v += epi::split_decode4(rr4[k % AHEAD][g]);
                if (k + AHEAD < 4 * TM) rr4[k % AHEAD][g] = *reinterpret_cast<const f32x4*>(p.residual + rowv[AHEAD] + 8 * g);
                if (okv[0]) {
                    vmax = epi::absmax4(vmax, v);
                    *reinterpret_cast<f32x4*>(p.y + rowv[0] + 8 * g) = epi::split_encode4(v);
                }
            }
#pragma unroll
            for (int a = 0; a < AHEAD; ++a) { okv[a] = okv[a + 1]; rowv[a] = rowv[a + 1]; }
        }
        if (p.status && vmax >= 65520.f) { atomicOr(p.status, 1u); atomicMin(p.status + 1, (unsigned)p.layer_idx); }
    };

    // The walk.  Static (persist 1): unit k + (workgroups of the XCD) is next.  Dynamic (persist 2): the XCD's workgroups claim units
    // from one counter as they get to them -- units differ (edge tiles store 1 - 3 of their 4 outputs, padding rows none) and CUs
    // do not run in step, and a static list gives the whole launch the pace of its unluckiest workgroup (measured: +2.5 % per launch
    // against one unit per workgroup, where the hardware dispatcher does exactly this balancing).  The claim is made at a unit's
    // start and read after its first point: the V stream needs it 5 K-tiles before the unit's end.
    const bool dyn = p.persist == 2;
    for (uint32_t k = slot, k_next; k < xn; k = k_next) {
        uint32_t rt, ct; unit_of(k, rt, ct);
        k_next = k + ustep;
        if (dyn) { if (threadIdx.x == 0) s_claim = ustep + atomicAdd(p.claims + xcd, 1u); a_has_next = 0; }
        else {
            a_has_next = k_next < xn ? 1u : 0u;
            uint32_t rt2, ct2; unit_of(a_has_next ? k_next : k, rt2, ct2); a_next_base = rt2 * rt_bytes;
        }
        // the unit's first weight fragments, and (again: the last K-tile of the unit before read them already, into registers the
        // epilogue then took) the activation fragments of its first K-tile, which sits in LDS buffer 0 since that K-tile's barrier
        __builtin_amdgcn_sched_barrier(0);
        w_soff = ct * (WINO_BN / 32) * SPLIT_WBLOCK;
        load_b(c0{});
        bt.template read_frags<0, 0>(af0);
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) Y[o][i][r] = 0.f;
            // fold M into the four outputs: Y[a][b] += cA(a, i) * cA(b, j) * M for point xi = (i, j), cA = A^T = [1 1 1 0; 0 1 -1 -1].
            // The coefficients (0, +-1) are block-uniform scalars; an output's 16 packed fused multiply-adds (v_pk_fma_f32: two
            // accumulators per instruction, IEEE per element) are SKIPPED by a scalar branch when its coefficient is zero -- 28 of the
            // 64 (point, output) pairs: fp32 vector instructions are paid in full in matrix-pipe time on this part
            // (tools/mfma_valu_coexec_probe.hip), the skip is -1.1 % on the kernel (profiles/r6_wino_zero_c.md), the same bits.  The
            // branch sits between a point's K loop and the next, around vector code only: no accumulator is copied, nothing spills.
            // What did NOT work: a switch over specialised folds (the register allocator copied the output accumulators at the join and
            // spilled; round 3), the point loop unrolled 16 x with compile-time coefficients (70 KB of code for a 64 KB instruction
            // cache, 565 bytes per lane spilled; round 6).
            for (int xi = 0; xi < 16; ++xi) {
                run_point();
                if (dyn && xi == 0) {                     // (every wave is past several barriers since thread 0 wrote the claim)
                    k_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)*reinterpret_cast<volatile uint32_t*>(&s_claim));
                    a_has_next = k_next < xn ? 1u : 0u;
                    uint32_t rt2, ct2; unit_of(a_has_next ? k_next : k, rt2, ct2); a_next_base = rt2 * rt_bytes;
                }
                if constexpr (WS_ABL & 2) { if (xi != 15) continue; }
                const int I = xi >> 2, J = xi & 3;
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float c = (float)(wino_cA(o >> 1, I) * wino_cA(o & 1, J));
                    if constexpr (WS_FOLD_SKIP) { if (c == 0.f) continue; }
                    const f32x2 cc = {c, c};
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const f32x2 y2 = __builtin_elementwise_fma(cc, f32x2{M[i][0][r], M[i][0][r + 1]}, f32x2{Y[o][i][r], Y[o][i][r + 1]});
                            Y[o][i][r] = y2[0]; Y[o][i][r + 1] = y2[1];
                        }
                }
            }
        if (p.residual) epilogue_res(rt, ct);
        else if (!(p.flags & EPI_DROPOUT)) epilogue(std::integral_constant<int, 0>{}, rt, ct);
        else if (p.mask_bits) epilogue(std::integral_constant<int, 2>{}, rt, ct);
        else epilogue(std::integral_constant<int, 1>{}, rt, ct);
    }
}

bool wino_split_ok(int C, int N) { return C >= 128 && (C % 128) == 0 && N >= 128 && (N % 128) == 0; }     // K-tiles in groups of 4

template <int WINO_BN>
static hipError_t launch_wino_split_bn(const WinoSplitParams& p, hipStream_t st) {
    using BT = SplitTile<64, WINO_BN, 1, WINO_BN / 32>;
    auto k = wino_split_kernel<WINO_BN>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(k), BT::LDS_BYTES, attr_done); e != hipSuccess) return e;
    // persistent: as many workgroups as the chip holds at once (256 CUs x 1 workgroup of 8 waves, or x 2 of 4 waves)
    const int resident = 256 * (WINO_BN == 256 ? 1 : 2);
    const int grid = (p.persist && p.units > resident) ? resident : p.units;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(WINO_BN * 2), BT::LDS_BYTES, st, p);
    return hipGetLastError();
}
hipError_t launch_wino_split(const WinoSplitParams& p, hipStream_t st) {
    if (p.bm != 64 || (p.residual && (p.flags & EPI_DROPOUT))) return hipErrorInvalidValue;      // (no reference model drops out in front of a residual add: byolo_plan.hip keeps such a layer direct)
    return p.bn == 256 ? launch_wino_split_bn<256>(p, st) : launch_wino_split_bn<128>(p, st);
}

}  // namespace byk
