// byolo_kernels.h -- host-callable launchers of the gfx950 kernels (internal; the public
// boundary is include/byolo.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace byk {

// Opt a kernel into more than 64 KB of dynamic LDS, once per DEVICE (the attribute belongs to the device's copy of the
// function; a process may hold engines on several devices) and safely from several host threads.
inline hipError_t set_dynamic_lds_once(const void* fn, size_t bytes, std::atomic<uint64_t>& done) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
    return e;
}


enum : int { EPI_LEAKY = 1, EPI_DROPOUT = 2, EPI_RESIDUAL = 4, EPI_RAW = 8 /* store the accumulators as they are */,
              EPI_F32OUT = 16 /* split-f16 launch: plain fp32 output (detection heads) */ };

// n / d for n < 2^31 as (umulhi(n, mul) + n) >> shr: the kernels divide by launch constants only
// (h*w, w, T, #column tiles), and a 32-bit integer division costs ~30 vector-ALU instructions that
// are NOT hidden under the MFMAs (tools/mfma_peak.hip).
struct FastDiv { uint32_t mul, shr; };
inline FastDiv make_fastdiv(uint32_t d) {
    uint32_t l = 0;
    while (l < 32 && ((uint64_t)1 << l) < d) ++l;
    FastDiv f;
    f.mul = (uint32_t)(((((uint64_t)1 << l) - d) << 32) / d + 1);
    f.shr = l;
    return f;
}

// byte offsets into a source are 32-bit (buffer addressing); rows that fall into the zero padding get
// this offset, which is out of range for every admissible source (<= CONV_MAX_SRC_BYTES) -> they read 0
static constexpr uint32_t CONV_OOB_OFFSET = 0xC0000000u;
static constexpr uint64_t CONV_MAX_SRC_BYTES = 0xBFF00000ull;

// One fused convolution  dst = [residual +] leaky( mask * (conv(src) * scale [/keep]) + shift )
// as an implicit GEMM  M = S*Hout*Wout (pixels), N = cout, K = ksize^2 * (C0 + C1).
// The input is the channel concat of up to two NHWC sources, each optionally read through a
// nearest x2 upsample (sh = 1) and/or a T-fold batch tile (sdiv = T): the reference's
// tf.image.resize_nearest_neighbor / tf.concat(axis=3) / tf.concat([x]*T, axis=0)
// (lib_yolo/layers.py:578-597) are never materialised.
struct ConvParams {
    const float* src0; const float* src1;
    uint32_t src0_bytes, src1_bytes;  // extent of each source (buffer descriptor range)
    const float* wpk;                 // packed weights [K/32][Npad][32]   (direct kernel: HWIO as is)
    uint32_t w_bytes;
    const float* scale; const float* shift;   // per output channel
    const float* residual;            // [M][ldc] or null
    const float* addend;             // [B*hw][N] raw partial sums joined before scale (T-invariant half), or null
    int addend_T;                     // samples per image of THIS launch's rows (1 if rows are images)
    int rep;                          // >= 1: epilogue replays for `rep` MC samples per computed row
    float* dst;                       // [M][ldc]
    int C0, C1;                       // channels of the two sources (C1 == 0: single source)
    int Hs0, Ws0, Hs1, Ws1;           // physical spatial size of each source
    int sh0, sh1;                     // upsample shift per source (0 | 1)
    int sdiv0, sdiv1;                 // sample divisor per source (1 | T)
    int Hin, Win, Hout, Wout;         // logical conv input / output size
    int ksize, stride, pad;
    int M, N, Npad, ldc;
    int KT, cin_tiles;                // K/32, (C0+C1)/32
    int flags;                        // EPI_*
    int split;                        // matrix-pipe launch: 1 = split-f16 sources / weights / residual / output (mfma_pipe.h), 0 = fp32;
                                      // direct launch: bit 0 = sources and residual are hi/lo tensors, bit 1 = the output is
    int kx3;                          // split launch: 1 = 3x3 / stride 1 on shared-tap stages (conv_tile_kx3): weights packed in (ky, chunk, kx) order, KT counts
                                      // stages; 2 = 1x1 / stride 1 over one plain source on the uniform loop (conv_tile_p1)
    uint32_t k0, k1, thr;             // dropout keys (byolo_rng.h)
    uint64_t idx_base;                // dropout element index of dst[0] (sub-batch / shard of a logical batch)
    // injected dropout masks (byolo_forward's d_mask_bits; lib_yolo/layers.py:521-524 with the caller's own Bernoulli draw):
    // bit i decides element i of THIS CALL's dropout input [S,h,w,cout] (1 = keep; < 2^32 elements, idx_base = 0);
    // null = the counter hash
    const uint32_t* mask_bits;
    // numeric status of the handle (byolo_status): status[0] |= 1 when a split-f16 output leaves the fp16 range,
    // status[1] = min(status[1], layer) -- the first layer it happened in; null = not tracked
    unsigned* status; int layer_idx;
    // set by launch_conv_igemm: a split-f16 launch whose epilogue is the plain case -- no addend, residual, T-replay, raw / fp32 output
    // or injected masks, cout % 32 == 0, 16-byte rows -- and may run the straight-line epilogue (conv_igemm.hip finish_plain)
    int plain;
    int no_plain;              // byolo_plan_opts.plain_epilogue == 0: the general epilogue (finish_tile) on every launch (A/B)
    FastDiv d_hw, d_wout, d_sdiv0, d_sdiv1, d_addT;   // Hout*Wout, Wout, sdiv0, sdiv1, addend_T
    // split-K of the last partial round of tiles (conv_plan_split): blocks [0, full_tiles) compute whole
    // tiles, the remaining split_tiles tiles are computed by ksplit K-slice blocks each
    int full_tiles, split_tiles, split_blocks, ksplit;
    // stream-K (small launches: fewer tiles than a few rounds of resident workgroups): sk_grid > 0 workgroups share
    // the launch's tiles * KT K-tile units evenly -- workgroup g owns units [g*sk_q + min(g, sk_r), ...) -- a tile
    // that straddles workgroups is reduced through slabs by the last arriver, like a split-K tile
    int sk_grid, sk_q, sk_r;
    FastDiv d_skq, d_skq1, d_kt;      // sk_q, sk_q + 1, KT
    float* slabs;                     // [split_tiles][ksplit][128*BN] raw accumulators  (stream-K: [2 * sk_grid][128*BN])
    uint32_t slab_bytes;
    unsigned* counters;               // [split_tiles] arrival tickets, zero at launch
    // Winograd-domain batched GEMM (winograd.hip): the rows are 16 blocks of wino_rows (a multiple of 128) rows, block
    // xi multiplies with the weight matrix at wpk + xi * wino_wstride bytes; 0 = ordinary convolution
    uint32_t wino_rows, wino_wstride;
    FastDiv d_wino;
    // filled by the launcher for the tile it picked
    FastDiv d_ntiles, d_cin, d_ks, d_ksplit;   // Npad / BN, cin_tiles, ksize, ksplit
    // Back-to-back fusion (conv_igemm.hip fused_tail; the 8-wave 128 x 256 shared-tap tile with cout == 256 only): the 1x1 convolution
    // or detection head that is the ONLY reader of this convolution's output runs in the same launch -- this layer's epilogue
    // writes its hi/lo rows into LDS instead of memory, a second MFMA pass multiplies them with the follower's weights, the
    // follower's epilogue stores.  f_wpk == null: no follower.  The f_ fields are the follower's ConvParams fields of the same name.
    const float* f_wpk; uint32_t f_w_bytes;
    const float* f_scale; const float* f_shift;
    float* f_dst;
    int f_N, f_Npad, f_ldc, f_flags, f_layer_idx;
    uint32_t f_k0, f_k1, f_thr; uint64_t f_idx_base;
    const uint32_t* f_mask_bits;
};

// Winograd F(2x2, 3x3) transforms around the GEMM (winograd.hip)
struct WinoParams {
    const float* x; float* v;         // input [S,H,W,C];  V [16][P_pad][C] (this chunk)
    const float* m; float* y;         // M [16][P_pad][N] (this chunk);  output [S,H,W,N]
    const float* residual;            // [S,H,W,N] or null
    const float* scale; const float* shift;
    int H, W, C, N, th, tw;           // th, tw = ceil(H/2), ceil(W/2) output tiles per image
    int s0;                           // first sample of the chunk
    int P, P_pad;                     // tiles of the chunk (samples * th * tw), rounded up to 128
    int flags; uint32_t k0, k1, thr; uint64_t idx_base;
    FastDiv d_tt, d_tw, d_c4, d_n4;   // th*tw, tw, C/4, N/4
    float vmul;                       // split precision (wino_split.hip): V = vmul * (B^T d B) of the stored hi + lo values
};
// Winograd F(2x2,3x3) in split-f16 arithmetic (wino_split.hip): V [16][P_pad x C, K-tile major inside a plane: v_index()] as hi/lo groups -> output [S,H,W,N] as hi/lo
// groups, GEMM + output transform + epilogue in one launch of P_pad / 128 * N / 128 workgroups
struct WinoSplitParams {
    const float* v; uint32_t v_bytes;     // V of this chunk; rows beyond v_bytes read 0
    uint32_t xi_stride;                   // bytes between consecutive transform points in V (P_pad * C * 4)
    const float* w; uint32_t w_bytes;     // U: 16 * C / 32 K-tiles in (point, chunk) order, split-f16 fragment order (mfma_pipe.h)
    float* y;                             // output [S,H,W,N], hi/lo groups
    const float* residual;                // [S,H,W,N] hi/lo groups, added after the activation (layers.py:505-507), or null; not together with dropout
    const float* scale; const float* shift;
    int C, N, KT, n_tiles;                // KT = C / 32 (a multiple of 4), n_tiles = N / bn
    int H, W, th, tw, s0, P, P_pad;       // as WinoParams
    int bm, bn;                           // output tiles / channels per workgroup: 64 (P_pad is a multiple of it), 128 | 256
    int persist;                          // byolo_plan_opts.wino_split_persist: 0 one unit per workgroup, 1 the resident workgroups walk a static unit list,
                                          // 2 they claim the next unit of their XCD from `claims` (8 words, zero before the launch)
    unsigned* claims;
    int units;                            // P_pad / bm * n_tiles units of (64 output tiles, bn channels)
    int flags; uint32_t k0, k1, thr; uint64_t idx_base; const uint32_t* mask_bits;
    unsigned* status; int layer_idx;
    FastDiv d_ntiles, d_tt, d_tw;
};
bool wino_split_ok(int C, int N);
hipError_t launch_wino_split_input(const WinoParams& p, hipStream_t st);
hipError_t launch_wino_split(const WinoSplitParams& p, hipStream_t st);
// Row-streaming persistent GEMM (gemm_stream.hip): the Winograd-domain GEMM (epi 0: 16 row blocks of RT row tiles, one
// weight matrix each, raw accumulators out), a 1x1 / stride-1 convolution with its fused epilogue (epi 1), or a
// detection head (epi 2: + bias, any cout)
struct GemmStreamParams {
    const float* a; uint32_t a_bytes;     // A: [rows][C] (rows beyond a_bytes read 0)
    const float* w; uint32_t w_bytes;     // weight matrices, each packed [C/32][Npad][32], wstride bytes apart
    float* dst;                           // [rows][ldc]
    int C, N, KT, n_tiles;                // KT = C / 32 (even), n_tiles = column tiles (128 wide; 64 for cout <= 64)
    int RT;                               // row tiles per weight matrix (one matrix: > all row tiles)
    int slots, q, rem;                    // 512 / n_tiles row ranges of q (+1 for the first rem) row tiles
    uint32_t wstride;
    FastDiv d_ntiles, d_RT;
    int epi, M, Npad, ldc;                // epilogue kind; valid rows (epi != 0); weight rows per K-tile; dst row stride
    const float* scale; const float* shift;
    const float* addend; int hw;          // epi 1: raw partial sums [images*hw][N] joined before scale, or null; h*w
    int flags; uint32_t k0, k1, thr; uint64_t idx_base;      // EPI_LEAKY / EPI_DROPOUT; dropout keys (byolo_rng.h)
    FastDiv d_hw, d_addT;                 // hw, samples per image of the rows (addend)
};
int conv1x1_stream_tile(int M, int C, int N, bool force = false);    // tile width (128 | 64) if a 1x1 convolution of M rows fits the streaming launch, else 0
// The same GEMM with the output transform and the convolution's epilogue fused in (wino_fused.hip): no M
struct WinoFusedParams {
    const float* v; uint32_t v_bytes;     // V: [16][P_pad][C]
    const float* w; uint32_t w_bytes;     // 16 matrices, each packed [C/32][N][32]
    float* y; const float* residual;      // output / residual [S,H,W,N]
    const float* scale; const float* shift;
    int C, N, KT, n_tiles;                // KT = C / 32 (even), n_tiles = N / 64
    int H, W, th, tw, s0, P;              // as WinoParams
    int slots, q, rem;                    // 512 / n_tiles ranges of q (+1 for the first rem) row tiles (128 output tiles each)
    uint32_t xi_stride, wstride;          // bytes between consecutive transform points in V / in the weights
    int flags; uint32_t k0, k1, thr; uint64_t idx_base;
    FastDiv d_ntiles, d_tt, d_tw;
};
bool wino_fused_ok(int C, int N);
hipError_t launch_wino_fused(const WinoFusedParams& p, hipStream_t st);
bool gemm_stream_ok(int C, int N);
hipError_t launch_gemm_stream(const GemmStreamParams& p, hipStream_t st);
hipError_t launch_wino_input(const WinoParams& p, hipStream_t st);
hipError_t launch_side_delay(int microseconds, hipStream_t st);   // <= 48 registers: co-resident with wino_fused_kernel
hipError_t launch_wino_output(const WinoParams& p, hipStream_t st);
void wino_weight_transform(const float g[9], float u[16]);   // host: U = G g G^T

struct ConvSplit { int full_tiles, split_tiles, split_blocks, ksplit; int sk_grid = 0; };
// decision (shape-only, deterministic); KT in the launch's scheduling units, tk_scale = time of a unit / time of an fp32 K-tile
ConvSplit conv_plan_split(int M, int Npad, int KT, int tile, double tk_scale = 1.0, int ksplit_knob = -1, int streamk_knob = 1);      // knobs: byolo_plan_opts.ksplit / .streamk
size_t conv_split_slab_bytes(const ConvSplit& sp, int tile);

// tile configuration ids
enum : int { TILE_128x128 = 0, TILE_128x64 = 1, TILE_128x32 = 2, TILE_128x256 = 3 };      // 128x256: 8 waves, split-f16 shared-tap 3x3 only
int conv_tile_bn(int tile);           // BN of a tile config
int conv_pick_tile(int N);            // tile config for cout = N
int conv_split_tile(int tile, bool wide);  // split precision: 128-wide tiles exist for the shared-tap 3x3 and the 1x1 kernels only
hipError_t launch_conv_igemm(const ConvParams& p, int tile, hipStream_t st);
// a route / upsample / stack view copied into a dense [M][C0 + C1] tensor (sources, extents and dst as in ConvParams)
hipError_t launch_view_gather(const ConvParams& p, hipStream_t st);
hipError_t launch_tensor_add(const float* a, const float* b, float* dst, int64_t n, bool split, hipStream_t st, unsigned* status = nullptr, int layer = 0);   // dst = a + b (split: all three in [4 hi | 4 lo] groups)
hipError_t launch_zero_words(void* p, int64_t n_words, hipStream_t st);                    // the forward's ticket / claim words (a kernel, not a memset node: conv_kernels.hip)
hipError_t launch_u8_to_f32(const uint8_t* src, float* dst, int64_t n, hipStream_t st);   // float(u8) * (1 / 255), fp32
hipError_t launch_f32_to_split(const float* src, float* dst, int64_t n, float mul, hipStream_t st, unsigned* status = nullptr);   // hi/lo pairs of mul * src
hipError_t launch_split_to_f32(const float* src, float* dst, int64_t n, float mul, hipStream_t st);   // dst[i] = mul * (hi + lo) of a split-f16 tensor
hipError_t launch_conv_direct(const ConvParams& p, hipStream_t st);   // small-cin (stem) direct conv
// The finish of a 1x1 concat convolution whose stacked half is an UPSAMPLED tensor (byolo_api.hip STEP_FINISH): a 1x1 convolution
// commutes with nearest-neighbour upsampling, so that half was multiplied at the source's resolution (`low`, raw accumulators
// [S, H/2, W/2, N]); here every output pixel (s, y, x) takes low[s, y/2, x/2] + part[s / T, y, x] (the T-invariant half, raw
// accumulators per image, or null) through the convolution's epilogue -- the arithmetic, in the order, of finish_tile.
struct FinishParams {
    const float* low; const float* part; float* dst;
    int S, H, W, N, T;
    const float* scale; const float* shift;
    int flags; uint32_t k0, k1, thr; uint64_t idx_base; const uint32_t* mask_bits;
    unsigned* status; int layer_idx;
    int mode;                         // 0 the raw fp32 sum (BN calibration), 1 epilogue with fp32 output (fp32 mode), 2 epilogue with hi/lo output
    FastDiv d_hw, d_w, d_n4, d_T;     // (H / 2) * (W / 2), W / 2, N / 4, T
};
hipError_t launch_finish_upsampled(const FinishParams& p, hipStream_t st);

// ---- calibration helpers -------------------------------------------------------------------
// per-channel mean / population variance of x [M][C]  -> stats [2][C] (double accumulation)
hipError_t launch_channel_stats(const float* x, int64_t M, int C, float* d_mean, float* d_var, double* d_tmp,
                                hipStream_t st);
// in place: x = [residual +] leaky(x * scale + shift)
// split: x holds fp32 on entry and [4 hi | 4 lo] groups on exit (C % 4 == 0), the residual is a split-f16 tensor
hipError_t launch_bn_act_inplace(float* x, int64_t M, int C, const float* scale, const float* shift,
                                 const float* residual, int leaky, bool split, hipStream_t st);
// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale   (device-side fold)
hipError_t launch_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                          float* scale, float* shift, int C, hipStream_t st);

// ---- tail ------------------------------------------------------------------------------------
constexpr int BYOLO_MAX_CLASSES = 128;   // register capacity of the largest decode build (tail_kernels.hip)
struct DecodeParams {
    const float* raw;        // [S, lh, lw, 3*blk]
    float* boxes;            // [B, n_total, D]
    int B, T, lh, lw, C;     // C = cls_cnt (1 .. BYOLO_MAX_CLASSES)
    int64_t n_total, box_base;
    float ph[3], pw[3];
    int layer_id;
    int ld;                  // floats per cell of `raw` (0 = 3*blk, dense; the detection convolution pads its rows to a multiple of 4)
    unsigned* status;        // numeric status (byolo_status): status[0] |= 2 when a raw value is not finite; null = not tracked
    // epistemic decode only (T sharded over ranks, byolo_set_tshard): 0 = reduce over the T samples and decode (the normal call);
    // 1 = write the 21 + C running SUMS of this call's T samples into the row instead ([sum l x4 | sum l l^T upper triangle x10 |
    // sum e^logvar x4 | sum sigma(obj) | sum H(obj) | sum softmax x C | sum H(cls)]: exactly one row's worth); 2 = `boxes` holds such
    // sums over all T samples (added up across the ranks): finish them in place -- means, covariance, determinant, entropies, corners
    int mode;
};
hipError_t launch_decode(int kind, const DecodeParams& p, hipStream_t st);
// decode_epistemic's dict entries outside the box row: ev_loc [B,lh,lw,3,4], covar [B,lh,lw,3,4,4], obj / cls samples
hipError_t launch_epi_stats(const DecodeParams& p, float* ev_loc, float* covar, float* obj_s, float* cls_s, hipStream_t st);

size_t nms_workspace_bytes(int B, int64_t N);
struct NmsParams {
    const float* boxes;      // [B, N, D]
    int B; int64_t N; int D, obj_idx, cls_start;
    int two_class, max_out; float iou_thr;
    void* ws; size_t ws_bytes;
    float* rows; int32_t* kept; int32_t* count;
    int general_only;        // byolo_plan_opts.nms_general: the general path for every image (tests)
};
hipError_t launch_sort_nms(const NmsParams& p, hipStream_t st);

// ---- ground-truth encoding and training loss (train_kernels.hip) ------------------------
struct EncodeGtParams {
    const float* boxes;      // [B, max_boxes, 4]  ymin, xmin, ymax, xmax (image fractions)
    const int32_t* labels;   // [B, max_boxes]
    const int32_t* counts;   // [B] boxes of each image (null: max_boxes everywhere)
    int B, max_boxes, n_layers, N;      // N = prior boxes of all layers
    int lh[4], lw[4], base[4];          // grid and first prior box of each layer ([row, col, box] inside a layer)
    double ph[4][3], pw[4][3];          // priors as the Python doubles of lib_yolo/yolov3.py
    float ign_thresh;
    float* loc; float* obj; int32_t* cls; float* ign;      // [B, N, 4], [B, N], [B, N], [B, N]
    unsigned* best;          // workspace [B, max_boxes]: the maximum IoU of every box over all prior boxes, as float bits
};
size_t encode_gt_workspace_bytes(int B, int max_boxes);
hipError_t launch_encode_gt(const EncodeGtParams& p, hipStream_t st);
constexpr int LOSS_MAX_BLOCKS = 1024;
struct LossParams {
    const float* raw; int pitch;        // raw detection output [S, lh, lw, pitch >= 3 * blk]
    float* grad; int grad_pitch;        // d(loc + obj + cls) / d raw, or null
    int S, lh, lw, C, aleatoric, aleatoric_loss;
    const float* gt_loc; const float* gt_obj; const int32_t* gt_cls; const float* gt_ign;   // [S][gt_stride][4 | 1]: the layer's slice
    int64_t gt_stride;                  // prior boxes between consecutive images in the ground-truth arrays
    double* partial; double* out;       // LOSS_MAX_BLOCKS x 3 partial sums; out[3] = loc, obj, cls
};
size_t loss_workspace_bytes();
hipError_t launch_loss(const LossParams& p, hipStream_t st);

}  // namespace byk
