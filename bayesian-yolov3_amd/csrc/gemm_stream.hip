// gemm_stream.hip -- row-streaming persistent GEMM  D[rows][N] = A[rows][C] * W[C][N]  on the fp32 matrix pipe, for the
// GEMM-shaped work whose K is short (C = 128 .. 1024, i.e. 4 .. 32 K-tiles per output tile):
//
//   EPI_RAW    the Winograd-domain GEMM  M[xi] = V[xi] (P_pad x C) * U[xi] (C x N),  xi = 0..15: 16 row blocks, one
//              weight matrix each, raw accumulators out (winograd.hip reads them back);
//   EPI_CONV   a 1x1 / stride-1 convolution over one plain NHWC source (lib_yolo/layers.py:545-575 with kernel_size 1:
//              the 1x1 convolutions of the three heads) with the convolution's fused epilogue: [+ per-image partial sum
//              of the T-invariant half of a concat input] -> dropout mask * scale -> + shift -> leaky;
//   EPI_BIAS   the detection heads (layers.py:600-613: 1x1 conv + bias, linear; N = 3*(5+C) or 3*2*(5+C), any count).
//
// Why not conv_igemm_kernel (which computes the same things, and did, first): a workgroup that computes ONE output tile
// pays its prologue (row bookkeeping, cold first loads, pipeline fill) and epilogue per 4-32 K-tiles: measured 57 / 122 /
// 134 TFLOP/s at K = 128 / 256 / 512 where the same loop reaches 144 on K = 1152 .. 4608.  Here a workgroup owns a
// column tile and a contiguous run of row tiles: the loads of the next row tile's first K-tiles are already in flight
// while the last K-tiles of the current one are multiplied; between two row tiles there is only the epilogue.
// Tile, LDS image, fragment scheme and MFMA / LDS / load interleaving: mfma_pipe.h.
// Measured (config 4 head layers, Winograd-domain, TFLOP/s executed):  K = 512: 131,  K = 256: 126,  K = 128: 113-116.
// Timing ablations: without the loads AND the stores the loop runs at 140-145 on all three; the stores alone cost 2 / 9 /
// 19 %, the loads 5 / 8 / 11 %.  It is the memory system, not the pipeline: at K = 128 the GEMM writes 1 KB and reads
// 0.5 KB per 65.5 kFLOP = 43 FLOP/B, i.e. 3.3 TB/s at 140 TFLOP/s.  (A second A register set fetched three tiles ahead:
// no gain.)
//
// Work split: grid = 512 workgroups (2 per CU, what the LDS allows); slot s = one of 512 / n_tiles row ranges
// (balanced to one row tile), its n_tiles workgroups (one per column tile) sit on the SAME XCD and run the same rows at
// the same time, so a row tile of A is fetched from HBM once.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "byolo_kernels.h"
#include "byolo_rng.h"
#include "mfma_pipe.h"
#include "epilogue.h"

namespace byk {

using namespace pipe;

enum : int { GS_RAW = 0, GS_CONV = 1, GS_BIAS = 2 };

// BN = 128: 4 waves of 64 x 64 (2 x 2 accumulators of 32 x 32); BN = 64: 4 waves of 64 x 32.  K-tile 32.
template <int BN, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_stream_kernel(const GemmStreamParams p) {
    using BT = BlockTile<128, BN, 2, 2>;
    constexpr int BM = BT::BM, NT = BT::NT, TM = BT::TM, TN = BT::TN, A_LD = BT::A_LD, B_LD = BT::B_LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const BT t(smem);
    const int tid = t.tid;

    // ---- which rows, which column tile -------------------------------------------------------------------
    const uint32_t b = blockIdx.x, x = b & 7u, i8 = b >> 3;
    const uint32_t sl = fdiv(i8, p.d_ntiles), tile_n = i8 - sl * (uint32_t)p.n_tiles;
    // XCD x owns slots [x * slots/8, (x+1) * slots/8): a contiguous 1/8 of the rows (Winograd: two of the 16 transform
    // points, so all workgroups of an XCD multiply with the same one or two L2-resident weight matrices)
    const uint32_t slot = x * ((uint32_t)p.slots >> 3) + sl;
    if (slot >= (uint32_t)p.slots) return;
    const uint32_t r0 = slot * (uint32_t)p.q + (slot < (uint32_t)p.rem ? slot : (uint32_t)p.rem);   // first row tile
    const int cnt = p.q + (slot < (uint32_t)p.rem ? 1 : 0);                                         // row tiles of this slot
    if (cnt <= 0) return;
    const int KT = p.KT, total = cnt * KT;

    // ---- load stream state (block-uniform except the per-row offsets) ------------------------------------------
    uint32_t a_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_voff[j] = (((r0 * BM + t.a_r + (NT / 8) * j) * (uint32_t)p.C) + t.a_q * 4) * 4u;
    const uint32_t a_tile_step = (uint32_t)BM * p.C * 4u;        // next row tile
    const uint32_t w_step = (uint32_t)p.Npad * 32 * 4;           // next K-tile of a weight matrix
    const uint32_t b_voff = (tile_n * BN * 32 + (uint32_t)tid * 4) * 4;
    const uint32_t xi0 = fdiv(r0, p.d_RT);
    int ld_in_xi = (int)(r0 - xi0 * (uint32_t)p.RT);             // row tile inside its weight block (load stream)
    uint32_t w_base = xi0 * p.wstride, w_soff = w_base, a_soff = 0;
    int ld_chunk = 0;
    bool ld_first = true;

    const __amdgpu_buffer_rsrc_t a_rsrc = make_rsrc(p.a, p.a_bytes);     // rows beyond the source read 0
    const __amdgpu_buffer_rsrc_t w_rsrc = make_rsrc(p.w, p.w_bytes);
    f32x4 a_reg[A_LD], b_reg[B_LD];

    auto next_tile = [&]() {                     // set up the K-tile the next issue_loads() fetches
        if (ld_chunk == 0 && !ld_first) {        // a new row tile: rows += 128; maybe the next weight matrix
#pragma unroll
            for (int j = 0; j < A_LD; ++j) a_voff[j] += a_tile_step;
            if (++ld_in_xi == p.RT) { ld_in_xi = 0; w_base += p.wstride; }
        }
        ld_first = false;
        a_soff = (uint32_t)ld_chunk * 128u;
        w_soff = w_base + (uint32_t)ld_chunk * w_step;
        if (++ld_chunk == KT) ld_chunk = 0;
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) a_reg[j] = buffer_load_x4(a_rsrc, a_voff[j], a_soff);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) b_reg[j] = buffer_load_x4(w_rsrc, b_voff, w_soff + j * (NT * 16));
    };

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };
    zero_acc();

    // ---- output: in the transposed 32x32 map a lane owns one row (pixel) and 4 x 4 consecutive columns -------------
    const int nb = (int)(tile_n * BN) + t.wn * TN * 32 + 4 * t.lh;
    uint32_t row_tile = r0;
    const uint32_t hw = (uint32_t)p.hw;
    const bool do_drop = p.flags & EPI_DROPOUT;
    const float slope = (p.flags & EPI_LEAKY) ? 0.1f : 1.f;
    auto flush = [&]() {                          // one finished row tile
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const uint32_t m = row_tile * BM + t.wm * TM * 32 + i * 32 + t.li;
            if constexpr (EPI == GS_RAW) {        // the accumulators are the result (rows are padded to the tile)
                float* d = p.dst + (size_t)m * p.ldc + nb;
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x4 v;
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q];
                        *reinterpret_cast<f32x4*>(d + (j * 32 + 8 * g)) = v;
                    }
            } else if (m < (uint32_t)p.M) {
                float* d = p.dst + (size_t)m * p.ldc + nb;
                if constexpr (EPI == GS_BIAS) {   // any N: scalar stores of the columns that exist
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int n = nb + j * 32 + 8 * g + q;
                                if (n < p.N) d[j * 32 + 8 * g + q] = acc[i][j][4 * g + q] + p.shift[n];
                            }
                } else {
                    // N % 4 == 0: a lane's 4-channel group is entirely inside or outside N, its element index is a
                    // multiple of 4 -> one 16-byte store, two pair hashes: the shared epilogue of epilogue.h
                    const float* add_row = nullptr;
                    if (p.addend) {               // raw partial sums of the concat's T-invariant half, one row per IMAGE pixel
                        const uint32_t img = fdiv(m, p.d_hw), pix = m - img * hw;
                        add_row = p.addend + (size_t)(fdiv(img, p.d_addT) * hw + pix) * p.N + nb;
                    }
                    const epi::DropRow drow(p.idx_base + (uint64_t)m * (uint64_t)p.N + (uint64_t)nb, p.k1);
                    f32x4 extra[TN * 4];
                    if (add_row) {
#pragma unroll
                        for (int k = 0; k < TN * 4; ++k) {
                            const int dn = (k >> 2) * 32 + 8 * (k & 3);
                            extra[k] = nb + dn < p.N ? *reinterpret_cast<const f32x4*>(add_row + dn) : f32x4{0.f, 0.f, 0.f, 0.f};
                        }
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int dn = j * 32 + 8 * g, n0 = nb + dn;
                            if (n0 >= p.N) continue;
                            const f32x4 sc4 = *reinterpret_cast<const f32x4*>(p.scale + n0);   // x 1 / (1 - p) with the masks on (host)
                            const f32x4 sf4 = *reinterpret_cast<const f32x4*>(p.shift + n0);
                            f32x4 a4;
#pragma unroll
                            for (int q = 0; q < 4; ++q) a4[q] = acc[i][j][4 * g + q];
                            if (add_row) a4 += extra[j * 4 + g];
                            bool keep[4] = {true, true, true, true};
                            if (do_drop) epi::keep4(drow, dn, p.k0, p.thr, keep);
                            *reinterpret_cast<f32x4*>(d + dn) = epi::bn_act4(a4, sc4, sf4, keep, slope);
                        }
                }
            }
        }
        ++row_tile;
        zero_acc();
    };

    // ---- the pipeline (mfma_pipe.h) over all K-tiles of all row tiles of this workgroup -------------------------
    using yes = std::true_type;
    using no = std::false_type;
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    next_tile(); issue_loads(); t.template store_a<0>(a_reg); t.template store_b<0>(b_reg);
    if (total > 1) { next_tile(); issue_loads(); }
    __syncthreads();
    t.template read_frags<0, 0>(af0, bf0);

    auto mf = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN], int) { mfma_group<TM, TN>(acc, af, bf); };
    auto none = [] {};
    auto body = [&](auto buf_tag, auto has_next_tag, auto load_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool HN = decltype(has_next_tag)::value, LDT = decltype(load_tag)::value;
        tile_body<BUF, HN, 0, LDT ? BT::NLD : 0, BT::NLD>(
            t, af0, bf0, af1, bf1, mf, none, issue_loads,
            [&] { t.template store_a<BUF ^ 1>(a_reg); t.template store_b<BUF ^ 1>(b_reg); });
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;

    // KT is even: a row tile ends after the second tile of a pair
    const int half = KT >> 1;
    int pair_in_row = 0;
    int kt = 0;
    for (; kt + 3 < total; kt += 2) {
        next_tile(); body(c0{}, yes{}, yes{});
        next_tile(); body(c1{}, yes{}, yes{});
        if (++pair_in_row == half) { pair_in_row = 0; flush(); }
    }
    // the last pair (total is even and >= 2)
    body(c0{}, yes{}, no{});
    body(c1{}, no{}, no{});
    flush();
}

template <int BN, int EPI>
static hipError_t launch_gs(const GemmStreamParams& p, hipStream_t st) {
    constexpr size_t lds = BlockTile<128, BN, 2, 2>::LDS_BYTES;
    auto k = gemm_stream_kernel<BN, EPI>;
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(k), lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(512), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gemm_stream(const GemmStreamParams& p, hipStream_t st) {
    if (p.epi == GS_RAW) return launch_gs<128, GS_RAW>(p, st);
    if (p.epi == GS_BIAS) return launch_gs<64, GS_BIAS>(p, st);
    return p.Npad % 128 == 0 ? launch_gs<128, GS_CONV>(p, st) : launch_gs<64, GS_CONV>(p, st);
}

// usable when the column tiles divide the 512 resident workgroups into whole XCD groups and K-tiles pair up
bool gemm_stream_ok(int C, int N) {
    const int nt = N / 128;
    return (N % 128) == 0 && (C % 64) == 0 && (nt == 1 || nt == 2 || nt == 4 || nt == 8);
}

// A 1x1 / stride-1 convolution of M rows as a row-streaming launch: tile width (128, or 64 for cout <= 64), or 0 when the
// shape does not fit (K-tiles must pair up, column tiles must form whole XCD groups, every slot needs a few row tiles:
// with fewer the static row split wastes more than the streaming saves -- conv_igemm.hip's stream-K takes those).
int conv1x1_stream_tile(int M, int C, int N, bool force) {
    if (C % 64) return 0;
    const int bn = N <= 64 ? 64 : 128;
    if (bn == 128 && !gemm_stream_ok(C, N)) return 0;
    if (force) return bn;                         // tests: every shape the kernel can express
    const int nt = bn == 64 ? 1 : N / 128, slots = 512 / nt, rt = (M + 127) / 128;
    const int q = rt / slots, rem = rt % slots;
    if (q < 2) return 0;
    if (rem && (double)(q + 1) / ((double)rt / slots) > 1.08) return 0;       // > 8 % of the launch idle in the last row tile
    return bn;
}

}  // namespace byk
