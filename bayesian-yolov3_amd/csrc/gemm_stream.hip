// gemm_stream.hip -- the Winograd-domain GEMM  M[xi] = V[xi] (P_pad x C) * U[xi] (C x N),  xi = 0..15, as ONE
// persistent launch whose workgroups stream CONSECUTIVE ROW TILES through one software pipeline.
//
// Why not conv_igemm_kernel (which computes the same thing as a 1x1 convolution, and did, first): Cin is the whole
// K here (128 / 256 / 512 -> 4 / 8 / 16 K-tiles of 32), and a workgroup that computes ONE tile pays its prologue
// (row bookkeeping, cold first loads, pipeline fill) and epilogue per 4-16 K-tiles: measured 57 / 122 / 134 TFLOP/s
// where the same loop reaches 144 on K = 1152 .. 4608.  Here a workgroup owns a column tile and a contiguous run of
// ~40 row tiles: the loads of the next row tile's first K-tiles are already in flight while the last K-tiles of the
// current one are multiplied; between two row tiles there is only the store of the 64 accumulator registers.
// Same tile shape, LDS image, fragment scheme and MFMA / LDS / load interleaving as conv_igemm.hip (see there).
// Measured (config 4 head layers, TFLOP/s executed):  K = 512: 131,  K = 256: 126,  K = 128: 113-116.  Timing ablations:
// without the loads AND the stores the loop runs at 140-145 on all three; the stores alone cost 2 / 9 / 19 %, the loads
// 5 / 8 / 11 %.  It is the memory system, not the pipeline: at K = 128 the GEMM writes 1 KB and reads 0.5 KB per 65.5
// kFLOP = 43 FLOP/B, i.e. 3.3 TB/s at 140 TFLOP/s.  (A second A register set fetched three tiles ahead: no gain.)
//
// Work split: grid = 512 workgroups (2 per CU, what the LDS allows); slot s = one of 512 / n_tiles row ranges
// (balanced to one row tile), its n_tiles workgroups (one per 128-column tile) sit on the SAME XCD and run the
// same rows at the same time, so a row tile of V is fetched from HBM once.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "byolo_kernels.h"

namespace byk {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GS_LD = 36;                 // floats per staged row (32 + 4 pad)
constexpr int GS_RSRC = 0x00020000;

__device__ __forceinline__ uint32_t gdiv(uint32_t n, FastDiv d) { return (__umulhi(n, d.mul) + n) >> d.shr; }

template <int N_MFMA, int N_VMEM, int N_DSR, int N_DSW>
__device__ __forceinline__ void gs_interleave() {
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

}  // namespace

// 128 x 128 block tile, 4 waves of 64 x 64 (2 x 2 accumulators of 32 x 32), K-tile 32
__global__ __launch_bounds__(256, 2) void gemm_stream_kernel(const GemmStreamParams p) {
    constexpr int BM = 128, BN = 128, NT = 256, TM = 2, TN = 2, A_LD = 4, B_LD = 4;
    constexpr int ROWB = GS_LD * 4, A_BUF = BM * ROWB, B_BUF = BN * ROWB, B_BASE = 2 * A_BUF, JSTEP = (NT / 8) * ROWB;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x;

    // ---- which rows, which column tile -------------------------------------------------------------------
    const uint32_t b = blockIdx.x, x = b & 7u, i8 = b >> 3;
    const uint32_t sl = gdiv(i8, p.d_ntiles), tile_n = i8 - sl * (uint32_t)p.n_tiles;
    // XCD x owns slots [x * slots/8, (x+1) * slots/8): a contiguous 1/8 of the rows = two of the 16 transform
    // points, so all workgroups of an XCD multiply with the same one or two weight matrices (2 x C x N x 4 bytes,
    // L2-resident) while the V rows stream through
    const uint32_t slot = x * ((uint32_t)p.slots >> 3) + sl;
    if (slot >= (uint32_t)p.slots) return;
    const uint32_t r0 = slot * (uint32_t)p.q + (slot < (uint32_t)p.rem ? slot : (uint32_t)p.rem);   // first row tile
    const int cnt = p.q + (slot < (uint32_t)p.rem ? 1 : 0);                                         // row tiles of this slot
    if (cnt <= 0) return;
    const int KT = p.KT, total = cnt * KT;

    // ---- load stream state (block-uniform except the per-row offsets) ------------------------------------------
    const int a_q = tid & 7, a_r = tid >> 3;
    uint32_t a_voff[A_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) a_voff[j] = (((r0 * BM + a_r + (NT / 8) * j) * (uint32_t)p.C) + a_q * 4) * 4u;
    const uint32_t a_tile_step = (uint32_t)BM * p.C * 4u;        // next row tile
    const uint32_t w_step = (uint32_t)p.N * 32 * 4;              // next K-tile of a weight matrix
    const uint32_t b_voff = (tile_n * BN * 32 + (uint32_t)tid * 4) * 4;
    const uint32_t xi0 = gdiv(r0, p.d_RT);
    int ld_in_xi = (int)(r0 - xi0 * (uint32_t)p.RT);             // row tile inside its xi block (load stream)
    uint32_t w_base = xi0 * p.wstride, w_soff = w_base, a_soff = 0;
    int ld_chunk = 0;
    bool ld_first = true;

    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, GS_RSRC);
    const __amdgpu_buffer_rsrc_t w_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, GS_RSRC);
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
        for (int j = 0; j < A_LD; ++j)
            a_reg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, a_voff[j], a_soff, 0));
#pragma unroll
        for (int j = 0; j < B_LD; ++j)
            b_reg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(w_rsrc, b_voff, w_soff + j * (NT * 16), 0));
    };
    const int st_off = (a_r * GS_LD + a_q * 4) * 4;
    auto store_tile = [&](auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
#pragma unroll
        for (int j = 0; j < A_LD; ++j) *reinterpret_cast<f32x4*>(lds + st_off + (BUF * A_BUF + j * JSTEP)) = a_reg[j];
#pragma unroll
        for (int j = 0; j < B_LD; ++j) *reinterpret_cast<f32x4*>(lds + st_off + (B_BASE + BUF * B_BUF + j * JSTEP)) = b_reg[j];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int fa_off = ((wm * TM * 32 + li) * GS_LD + lh * 4) * 4;
    const int fb_off = ((wn * TN * 32 + li) * GS_LD + lh * 4) * 4;
    auto read_frags = [&](auto buf_tag, auto kq_tag, f32x4 (&af)[TM], f32x4 (&bf)[TN]) {
        constexpr int BUF = decltype(buf_tag)::value, KQ = decltype(kq_tag)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(lds + fa_off + (BUF * A_BUF + KQ * 32 + i * 32 * ROWB));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(lds + fb_off + (B_BASE + BUF * B_BUF + KQ * 32 + j * 32 * ROWB));
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
    auto mfma_group = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bf[j][s], af[i][s], acc[i][j], 0, 0, 0);   // D^T: rows = columns of M
    };

    // ---- output: in the transposed 32x32 map a lane owns one row (pixel) and 4 x 4 consecutive columns -------------
    const int nb = (int)(tile_n * BN) + wn * TN * 32 + 4 * lh;
    float* d_row[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) d_row[i] = p.dst + (size_t)(r0 * BM + wm * TM * 32 + i * 32 + li) * p.N + nb;
    const size_t d_tile_step = (size_t)BM * p.N;
    auto flush = [&]() {                          // one finished row tile: 32 x 16-byte stores per lane, accumulators := 0
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q];
                    *reinterpret_cast<f32x4*>(d_row[i] + (j * 32 + 8 * g)) = v;
                }
            d_row[i] += d_tile_step;
        }
        zero_acc();
    };

    // ---- the pipeline: tile_body as in conv_igemm.hip, over all K-tiles of all row tiles of this workgroup -------
    constexpr int G = 4 * TM * TN, NFR = TM + TN, NLD = A_LD + B_LD;
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using c2 = std::integral_constant<int, 2>;
    using c3 = std::integral_constant<int, 3>;
    using yes = std::true_type;
    using no = std::false_type;
    f32x4 af0[TM], bf0[TN], af1[TM], bf1[TN];
    next_tile(); issue_loads(); store_tile(c0{});
    if (total > 1) { next_tile(); issue_loads(); }
    __syncthreads();
    read_frags(c0{}, c0{}, af0, bf0);

    auto tile_body = [&](auto buf_tag, auto has_next_tag, auto load_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        using cur = std::integral_constant<int, BUF>;
        using nxt = std::integral_constant<int, BUF ^ 1>;
        constexpr bool HN = decltype(has_next_tag)::value, LD = decltype(load_tag)::value;
        __builtin_amdgcn_sched_barrier(0);
        read_frags(cur{}, c1{}, af1, bf1);
        mfma_group(af0, bf0);
        gs_interleave<G, 0, NFR, 0>();
        __builtin_amdgcn_sched_barrier(0);

        read_frags(cur{}, c2{}, af0, bf0);
        mfma_group(af1, bf1);
        gs_interleave<G, 0, NFR, 0>();
        __builtin_amdgcn_sched_barrier(0);

        read_frags(cur{}, c3{}, af1, bf1);
        if constexpr (HN) store_tile(nxt{});
        mfma_group(af0, bf0);
        gs_interleave<G, 0, NFR, HN ? NLD : 0>();
        __builtin_amdgcn_sched_barrier(0);

        __syncthreads();
        if constexpr (LD) issue_loads();
        if constexpr (HN) read_frags(nxt{}, c0{}, af0, bf0);
        mfma_group(af1, bf1);
        gs_interleave<G, LD ? NLD : 0, HN ? NFR : 0, 0>();
        __builtin_amdgcn_sched_barrier(0);
    };

    // KT is even: a row tile ends after the second tile of a pair
    const int half = KT >> 1;
    int pair_in_row = 0;
    int t = 0;
    for (; t + 3 < total; t += 2) {
        next_tile(); tile_body(c0{}, yes{}, yes{});
        next_tile(); tile_body(c1{}, yes{}, yes{});
        if (++pair_in_row == half) { pair_in_row = 0; flush(); }
    }
    // the last pair (total is even and >= 2)
    tile_body(c0{}, yes{}, no{});
    tile_body(c1{}, no{}, no{});
    flush();
}

hipError_t launch_gemm_stream(const GemmStreamParams& p, hipStream_t st) {
    constexpr size_t lds = (size_t)2 * (128 + 128) * GS_LD * sizeof(float);
    static std::atomic<uint64_t> attr_done{0};
    if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(gemm_stream_kernel), lds, attr_done); e != hipSuccess) return e;
    hipLaunchKernelGGL(gemm_stream_kernel, dim3(512), dim3(256), lds, st, p);
    return hipGetLastError();
}

// usable when the column tiles divide the 512 resident workgroups into whole XCD groups and K-tiles pair up
bool gemm_stream_ok(int C, int N) {
    const int nt = N / 128;
    return (N % 128) == 0 && (C % 64) == 0 && (nt == 1 || nt == 2 || nt == 4 || nt == 8);
}

}  // namespace byk
