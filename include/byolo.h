/* byolo.h -- C-ABI of libbyolo.so: the MI355X-native (gfx950) Bayesian-YOLOv3 inference path.
 *
 * The reference (flkraus/bayesian-yolov3) has no FFI / plugin / operator interface: its hot path
 * sits behind Python function-level interfaces on top of the TensorFlow-1.x graph API.  This
 * header is therefore the boundary a maintainer would bind if the path were native: one entry
 * point per reference interface, each citing the reference symbol (file:line, relative to the
 * reference tree) it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every function returns int32 status: 0 = BYOLO_OK, negative = error; a message is
 *     available from byolo_last_error(handle) (handle-less failures: byolo_last_error(NULL));
 *   - no C++ exception crosses the ABI; no torch / HIP C++ types in signatures; `stream` is a
 *     hipStream_t passed as void* (NULL = the default stream);
 *   - all `d_*` pointers are CALLER-OWNED DEVICE memory (e.g. PyTorch-ROCm tensors); the library
 *     never frees them and never retains them past the call;  `h_*` pointers are host memory;
 *   - the library owns only the packed weights it uploads in byolo_finalize();
 *   - a handle is bound to one device and is not thread-safe: one handle per device/thread;
 *   - all tensors are float32, activations NHWC, convolution kernels HWIO
 *     (lib_yolo/layers.py:550, tf.layers.conv2d defaults), images in [0,1)
 *     (lib_yolo/dataset_utils.py:6-11).
 */
#ifndef BYOLO_H_
#define BYOLO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BYOLO_API __attribute__((visibility("default")))

#define BYOLO_OK 0
#define BYOLO_ERR_ARG (-1)      /* invalid argument / graph construction error (reference: assert) */
#define BYOLO_ERR_STATE (-2)    /* call order violated (e.g. forward before finalize)             */
#define BYOLO_ERR_HIP (-3)      /* HIP runtime error                                              */
#define BYOLO_ERR_NOMEM (-4)    /* workspace too small                                            */
#define BYOLO_ERR_RANGE (-5)    /* split-f16 only: a value the reference's float32 tensors hold does not fit the
                                   hi/lo fp16 storage (|activation| > 16376), or a raw detection output is inf / NaN   */

typedef struct byolo byolo_t;

/* decode kinds == the three detection-layer factories of lib_yolo/model.py:107 / :132 / :157 */
enum { BYOLO_DET_STANDARD = 0, BYOLO_DET_ALEATORIC = 1, BYOLO_DET_EPISTEMIC = 2 };
/* inference_epistemic.py:99-102 (class-agnostic) and :104-126 (commented-out 2-class variant) */
enum { BYOLO_NMS_AGNOSTIC = 0, BYOLO_NMS_TWO_CLASS = 1 };
/* normaliser list of lib_yolo/layers.py:556-571 */
enum { BYOLO_NORM_BN = 1, BYOLO_NORM_DROPOUT = 2 };

typedef struct byolo_cfg {
    int32_t img_h, img_w, img_c;  /* config['full_img_size'] (yolov3.py:207-208: h, w % 32 == 0)   */
    int32_t cls_cnt;              /* config['cls_cnt'], 1 .. 128                                    */
    float   drop_prob;            /* 0.1 hard-coded at lib_yolo/yolov3.py:462                       */
    int32_t max_out;              /* 1000: tf.image.non_max_suppression(.., 1000)                   */
    float   iou_thresh;           /* 0.5: TF default                                                */
    int32_t nms_mode;             /* BYOLO_NMS_*                                                    */
    int32_t keep_all_outputs;     /* 1 = no activation-buffer reuse, every layer readable (tests)   */
} byolo_cfg;

/* ---- lifetime ---------------------------------------------------------------------------- */
BYOLO_API int32_t byolo_create(const byolo_cfg* cfg, int32_t device, byolo_t** out);
BYOLO_API int32_t byolo_destroy(byolo_t* h);
/* Arithmetic of the convolution stack (the reference runs TensorFlow's float32 kernels, lib_yolo/layers.py:550; both
 * modes accumulate in fp32 and hold the 1e-4 contract against the float32 CPU restatement):
 *   BYOLO_PREC_F32        fp32 operands on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32), Winograd F(2x2,3x3)
 *                         for the large 3x3 convolutions;
 *   BYOLO_PREC_SPLIT_F16  every activation / weight as hi + lo, two fp16 values (~23 significant bits), three fp16
 *                         matrix products per fp32 product into fp32 accumulators (v_mfma_f32_32x32x16_f16).  Weights
 *                         carry one exact power-of-two scale per OUTPUT CHANNEL (folded into the BN scale), so a filter
 *                         far smaller than its neighbours keeps its bits.  Activations are stored as 4 * value: one
 *                         beyond +-16376 does not fit -- that is DETECTED (byolo_status), never stored silently: see
 *                         BYOLO_ERR_RANGE and byolo_set_async below.
 * Default: environment BYOLO_PRECISION = f32 | split, else BYOLO_PREC_SPLIT_F16.  Call before byolo_finalize (a finalized
 * handle must be finalized again).  byolo_finalize itself falls back to BYOLO_PREC_F32 for a graph split storage cannot
 * express (a convolution whose output channels are not a multiple of 4 -- no reference model has one):
 * byolo_get_precision reports the mode in effect, byolo_precision_note why ("" if the request stands). */
enum { BYOLO_PREC_F32 = 0, BYOLO_PREC_SPLIT_F16 = 1 };
BYOLO_API int32_t byolo_set_precision(byolo_t* h, int32_t precision);
BYOLO_API int32_t byolo_get_precision(const byolo_t* h);
BYOLO_API const char* byolo_precision_note(const byolo_t* h);
BYOLO_API const char* byolo_last_error(const byolo_t* h);
BYOLO_API const char* byolo_version(void);
/* Incremented on every incompatible change of a signature or of a buffer layout this header describes, so that a binding can
 * refuse a library it was not written for (byolo/_lib.py does).  4: byolo_forward takes d_mask_bits; detection rows handed out
 * by byolo_layer_output are padded to a multiple of 4 floats; the host-side feed / writer entry points exist.  5: byolo_encode_gt,
 * byolo_loss (ground-truth encoding and training loss) exist.  6: the dropout stream byolo_forward draws from `seed` is redefined
 * (one hash + one derived word per group of four channels, csrc/byolo_rng.h): the same seed gives other masks than ABI 5 did; the
 * bit order of injected masks (d_mask_bits) is unchanged.  7: the plan of a handle is part of the ABI (byolo_plan_opts, byolo_get_plan_opts /
 * byolo_set_plan_opts, byolo_graph_stats): the BYOLO_* plan variables are read ONCE, at byolo_create, and no longer reach an existing
 * handle; a drop_prob whose 16-bit threshold rounds to 2^16 (rate 0) is the identity. */
#define BYOLO_ABI_VERSION 7
BYOLO_API int32_t byolo_abi_version(void);

/* ---- the plan of a handle: which kernel carries which layer, and how a forward is put on the stream -------------------------------
 * The reference leaves all of this to TensorFlow's placer and kernel registry (`sess.run`, inference_epistemic.py:76); here it is a
 * per-HANDLE struct, so that two handles in one process may differ and nothing outside a handle steers it.  byolo_create fills it
 * with the defaults below and then applies the environment variables named in brackets (the A/B switches of tools/ and tests/;
 * read ONCE, at byolo_create, never again); byolo_set_plan_opts replaces it.  Fields marked [pack] change what byolo_finalize packs:
 * set them before byolo_finalize (a finalized handle is marked un-finalized when one of them changes); the others may change between
 * forwards (the next forward plans again).  Every setting computes the layer functions of lib_yolo/layers.py:545-575; settings that
 * change the order of a floating-point sum (split-K, stream-K, Winograd on / off) change results within the fp32 re-association
 * error, all others bit for bit -- tests/test_gpu_parity.py holds each of them to the oracle. */
typedef struct byolo_plan_opts {
    int32_t struct_bytes;          /* sizeof(byolo_plan_opts) of the caller's header: a mismatch is BYOLO_ERR_ARG                      */
    int32_t graphs;                /* launch-graph replay of a whole forward: 0 never, 1 forwards that do not fill the chip by
                                      themselves (< 1 TFLOP: detect.py's batch-1 loop, BASELINE configs[0..1]; default), 2 every forward
                                      that can be captured [BYOLO_GRAPHS]                                                             */
    int32_t serialize_convs;       /* forwards of one handle on several streams: convolution stacks one after the other: 0 never,
                                      1 forwards of >= 1 TFLOP (default), 2 always [BYOLO_SERIALIZE_CONVS]                            */
    int32_t serialize_heads;       /* with serialize_convs in effect: only the HEADS wait for the other stream's convolution stack; this
                                      forward's backbone (small launches that leave CUs idle) runs beside the previous forward's heads: 1
                                      (default; +1.8 % img/s at configs[3]), 0 = the whole stack waits.  A forward recorded at profiling
                                      level 2 and the one after it always wait as a whole [BYOLO_SERIALIZE_HEADS]                      */
    int32_t dedup;                 /* [pack] T-invariant de-duplication (a convolution over a T-fold tile runs once per image): 1
                                      [BYOLO_NO_DEDUP inverts]                                                                        */
    int32_t lowmain;               /* [pack] the 1x1 convolution over an upsampled source at the source's resolution: 1 [BYOLO_LOWMAIN] */
    int32_t kx3, p1;               /* [pack] split-f16 K loops: shared-tap 3x3 stages / uniform 1x1 loop: 1, 1 [BYOLO_KX3, BYOLO_P1]   */
    int32_t b2b;                   /* 3x3 + following 1x1 in one launch (76x76 head pairs): 0 never, 1 launches of >= 4 rounds
                                      (default), 2 every eligible pair [BYOLO_B2B]                                                    */
    int32_t kx3_wide;              /* the 8-wave 128 x 256 shared-tap tile without a follower: 0 (default; measured 4 % slower), 1, 2
                                      [BYOLO_KX3_WIDE]                                                                                */
    int32_t wino_split;            /* Winograd F(2x2,3x3) in split-f16: 0 never, 1 the layers with >= wino_split_min_c input channels the
                                      planner's time model (byolo_plan.hip) expects >= 3 % faster than the direct kernel (default),
                                      2 every eligible layer [BYOLO_WINO_SPLIT]                                                       */
    int32_t wino_split_min_c;      /* 256 [BYOLO_WINO_SPLIT_MIN_C]                                                                    */
    int32_t wino_split_bn;         /* output channels per workgroup: 0 the time model picks per layer (default: 256 for launches of
                                      many rounds, 128 where that fills more CUs), 256 (8 waves) or 128 forced [BYOLO_WINO_SPLIT_BN]  */
    int32_t wino_split_rounds;     /* experiment: chunks of k whole rounds of workgroups, 0 = off [BYOLO_WINO_SPLIT_ROUNDS]           */
    int32_t winograd;              /* fp32 mode: Winograd for the large 3x3 layers: 0, 1 (default), 2 every eligible [BYOLO_WINOGRAD] */
    int32_t wino_fused;            /* fp32 mode: GEMM + output transform in one kernel: 0, 1 (default), 2 [BYOLO_WINO_FUSED]          */
    int32_t stream1x1;             /* fp32 mode: row-streaming 1x1 launches: 0, 1 (default), 2 [BYOLO_STREAM1X1]                      */
    int32_t gemm_stream;           /* fp32 mode: the unfused Winograd GEMM on the row-streaming kernel: 1 [BYOLO_GEMM_STREAM]         */
    int32_t ksplit;                /* split-K of a launch's last partial round: -1 the cost model (default), 0 never, n > 1 always n
                                      slices [BYOLO_KSPLIT]                                                                           */
    int32_t streamk;               /* stream-K for small launches: 0 never, 1 the cost model (default), 2 whenever admissible
                                      [BYOLO_STREAMK]                                                                                 */
    int32_t plain_epilogue;        /* straight-line epilogues of the split-f16 convolutions (one decision per tile): 1; 0 = the general
                                      epilogue everywhere (A/B; the same bits) [BYOLO_PLAIN_EPILOGUE]                                 */
    int32_t wshift_per_layer;      /* [pack] split-f16 weights: ONE power-of-two scale per layer instead of one per output channel: 0; 1 = the
                                      A/B of tests/test_robustness.py (a filter far smaller than its neighbours loses bits)
                                      [BYOLO_WSHIFT_PER_LAYER]                                                                        */
    int32_t nms_general;           /* the tail's general path (sort + bit-matrix NMS over all candidates) for every image, never the
                                      top-k fast path: 0; 1 = tests / soak runs of that path [BYOLO_NMS_GENERAL]                      */
    int32_t wino_split_persist;    /* the Winograd GEMM's workgroups walk the unit list themselves, next unit prefetched: 0 one unit per
                                      workgroup (default), 1 a static list, 2 units claimed from a per-XCD counter; the same bits, and
                                      measured SLOWER: +2.5 % / +0.8 % per launch (profiles/r6_wino_persist.md) [BYOLO_WINO_SPLIT_PERSIST] */
    float   wino_split_min_gflop;  /* a floor under the time model: 30 [BYOLO_WINO_SPLIT_MIN_GFLOP]                                   */
    float   wino_split_chunk_mb;   /* V bytes of one chunk: 1500 [BYOLO_WINO_SPLIT_CHUNK_MB]                                          */
    float   wino_min_gflop;        /* fp32 mode: 10 [BYOLO_WINO_MIN_GFLOP]                                                            */
    float   wino_chunk_mb;         /* fp32 mode: V + M bytes of one chunk: 800 [BYOLO_WINO_CHUNK_MB]                                  */
    float   wino_min_ratio;        /* fp32 mode: Cin * cout / (Cin + cout) at least: 80 [BYOLO_WINO_MIN_RATIO]                        */
} byolo_plan_opts;
BYOLO_API int32_t byolo_get_plan_opts(const byolo_t* h, byolo_plan_opts* out);       /* out->struct_bytes is set                     */
BYOLO_API int32_t byolo_set_plan_opts(byolo_t* h, const byolo_plan_opts* opts);
/* Launch graphs kept by the handle (one per distinct argument set of a replayed forward; at most 8, least recently used dropped) and
 * how often a forward was replayed / captured / updated in place (a forward whose dropout seed differs from the captured one). */
BYOLO_API int32_t byolo_graph_stats(const byolo_t* h, int32_t* n_graphs, int64_t* replays, int64_t* captures, int64_t* updates);

/* ---- graph construction: one call per ModelBuilder.make_* (lib_yolo/model.py:52-185).
 * Each returns the new layer's index (>= 0) in the reference's `ModelBuilder.__layers` numbering
 * (model.py:40-41) or a negative error.  Layer references (`shortcut`, `routes`, `src`) follow
 * the reference: negative = relative to the end of the list, non-negative = absolute.
 *
 *
 * Route / upsample / stack layers are VIEWS: they exist only inside the loader of the convolution that reads them
 * (two concatenated sources, each upsampled at most once), and a residual add rides in the epilogue of the
 * convolution in front of it.  That covers everything the reference's three models build at no cost.  A general
 * ModelBuilder graph is accepted as well; what does not fit is lowered to plain steps of its own:
 *   - the inner view of a nested concat / double upsample, or a view used as an operand of a residual add, is
 *     copied into a tensor first;
 *   - a residual add whose left operand is not a convolution, or whose convolution has other readers, runs as an
 *     element-wise kernel;
 *   - input channels that are not a multiple of 32 (per concatenated source) take a general direct convolution
 *     instead of the matrix-pipe one -- correct, slow. ---------------------------------------------------------- */

/* make_conv_layer / make_downsample_layer / make_darknet_conv_layer / make_darknet_downsample_layer
 * (model.py:52-81 -> layers.conv, lib_yolo/layers.py:545-575): conv(no bias) -> [dropout] -> BN
 * (eps 1e-5) -> leaky-ReLU(0.1).  stride 2 = pad(1,1)+VALID (layers.py:533-537, :616-635).
 * `scope` = TF variable scope of the layer ("darknet53/conv_3", ...): names its variables
 * "<scope>/conv2d/kernel", "<scope>/batch_normalization/{gamma,beta,moving_mean,moving_variance}".
 * At most 65 536 filters and 2^28 weights per kernel (the reference's largest: 1024 / 4.7 M): beyond, BYOLO_ERR_ARG.
 * No entry point lets a C++ exception out: a host allocation that fails is BYOLO_ERR_NOMEM. */
BYOLO_API int32_t byolo_add_conv(byolo_t* h, const char* scope, int32_t filters, int32_t ksize, int32_t stride,
                       int32_t norm_flags);
/* make_residual_layer (model.py:91-94, layers.py:505-507) */
BYOLO_API int32_t byolo_add_residual(byolo_t* h, int32_t shortcut);
/* make_route_layer (model.py:83-86, layers.py:583-592): 1 route = identity, 2 = channel concat */
BYOLO_API int32_t byolo_add_route(byolo_t* h, const int32_t* routes, int32_t n_routes);
/* make_upsample_layer (model.py:101-105, layers.py:578-580): nearest x2 */
BYOLO_API int32_t byolo_add_upsample(byolo_t* h);
/* make_stack_feature_map_layer (model.py:88-89, layers.py:595-597): tile T x on the batch axis;
 * T is a run-time argument of byolo_forward.  Sample order: s = img*T + t. */
BYOLO_API int32_t byolo_add_stack(byolo_t* h, int32_t src);
/* make_detection_layer{,_aleatoric,_aleatoric_epistemic} (model.py:107-185): 1x1 conv + bias,
 * linear (layers.py:600-613), then split + decode (layers.py:11-84, :191-502).  priors_hw = the 3
 * (h, w) priors of this stride; layer_id = len(det_layers) so far (model.py:141, :167). */
BYOLO_API int32_t byolo_add_detection(byolo_t* h, const char* scope, int32_t kind, const float* priors_hw /*[3][2]*/);

/* `self.__darknet53_layer_cnt = mb.layer_cnt()` (lib_yolo/yolov3.py:246): everything added so far
 * is the backbone (used for load_darknet53_weights and for the per-stage timers). */
BYOLO_API int32_t byolo_mark_backbone_end(byolo_t* h);

/* ---- parameters: tf.train.Saver.restore / darknet.load_darknet_weights
 * (inference_epistemic.py:58, lib_yolo/darknet.py:42-122) ------------------------------------ */
BYOLO_API int32_t byolo_num_params(const byolo_t* h);
BYOLO_API int32_t byolo_param_info(const byolo_t* h, int32_t i, const char** name, int32_t* ndim, int64_t shape[4]);
BYOLO_API int32_t byolo_set_param(byolo_t* h, const char* name, const float* h_data, int64_t count);
BYOLO_API int32_t byolo_get_param(const byolo_t* h, const char* name, float* h_data, int64_t count);
/* fold BN (+ 1/keep_prob) into per-channel scale/shift, repack kernels for the MFMA tiles, upload.
 * May be called again after further byolo_set_param calls.  Packs on the host's cores (environment
 * BYOLO_FINALIZE_THREADS: default the hardware threads / LOCAL_WORLD_SIZE -- one process per GPU, the ranks of a node finalize
 * together --, at most 32; the packed bytes do not depend on it). */
BYOLO_API int32_t byolo_finalize(byolo_t* h);

/* ---- run: one sess.run([nms_op]) (inference_epistemic.py:76, inference_aleatoric.py:75) ------ */
BYOLO_API int32_t byolo_num_layers(const byolo_t* h);
BYOLO_API int32_t byolo_num_boxes(const byolo_t* h, int64_t* n_boxes, int32_t* row_len);   /* N, D of concat_bbox */
BYOLO_API int32_t byolo_workspace_bytes(byolo_t* h, int32_t B, int32_t T, size_t* out);
/* Introspection of the workspace plan of a (B, T) call, host only (no device, no byolo_finalize needed) -- what tests/test_planner.py
 * checks on the CPU: no tensor is written while another tensor that shares its memory is still to be read.
 *   byolo_plan_num     makes the plan (inject != 0: the plan of a call with d_mask_bits); steps, tensors, arena size;
 *   byolo_plan_step    the tensor step `step` writes; fuses_next = 1 when step + 1 runs INSIDE this step's launch (its output is
 *                      written during this step, this step's own output tensor never exists); the tensors the step's launch reads;
 *   byolo_plan_tensor  offset in the workspace (< 0: none in this plan), size, and whether anything reads it after the last step
 *                      (a detection layer's raw output: the decode launch; every layer under keep_all_outputs). */
BYOLO_API int32_t byolo_plan_num(byolo_t* h, int32_t B, int32_t T, int32_t inject, int32_t* n_steps, int32_t* n_tensors, int64_t* arena_bytes);
BYOLO_API int32_t byolo_plan_step(byolo_t* h, int32_t step, int32_t* out_tensor, int32_t* fuses_next, int32_t reads[8], int32_t* n_reads);
BYOLO_API int32_t byolo_plan_tensor(byolo_t* h, int32_t tensor, int64_t* offset, int64_t* bytes, int32_t* read_after_the_steps);
/* d_img [B,H,W,C] NHWC fp32.  T = MC samples (1 for graphs without stack layers).
 * Outputs (any may be NULL to skip that stage's export):
 *   d_boxes   [B,N,D]               pre-NMS rows in concat_bbox order (inference_*.py concat_bbox)
 *   d_rows    [B,max_out*(1|2),D]   NMS-kept rows, score order (2-class: ped rows then rider rows)
 *   d_kept    [B,max_out*(1|2)]     int32 global box indices of the kept rows
 *   d_count   [B,2]                 int32 {total kept, kept in first class (== total if agnostic)}
 * seed: dropout stream (see csrc/byolo_rng.h); dropout is active iff the layer has
 * BYOLO_NORM_DROPOUT and `dropout_on` != 0 (standard_test_dropout=True quirk, layers.py:567-568).
 * d_mask_bits (nullable): INJECTED dropout masks instead of the library's counter stream -- tf.layers.dropout draws
 *   its Bernoulli noise from an unseeded op (layers.py:521-524), so a caller that wants the reference's masks (or its
 *   own) passes them: one bit per element of each dropout layer's input [S,h,w,cout] of THIS call (S = B*T in the
 *   stacked part of the graph; byolo_set_first_image does not enter), 1 = keep, layer after layer at the bit offsets of
 *   byolo_mask_layout.
 * Numeric status (split precision): unless byolo_set_async(h, 1), the call waits for the stream and returns
 *   BYOLO_ERR_RANGE -- never rows of inf / NaN -- when an activation left the split-f16 range; outputs are then undefined.
 * Any B: a batch beyond byolo_max_images(h, T) runs inside this call as consecutive pieces of at most that many images in the
 *   same workspace (byolo_workspace_bytes sizes it for the largest piece), each drawing the dropout masks of its position in
 *   the batch -- the result does not depend on the cut (images are independent; the NMS is per image).  Injected masks
 *   (d_mask_bits) describe one piece: with them B must not exceed byolo_max_images.  byolo_layer_output then shows the LAST piece. */
BYOLO_API int32_t byolo_forward(byolo_t* h, const float* d_img, int32_t B, int32_t T, uint64_t seed, int32_t dropout_on,
                      const uint32_t* d_mask_bits, void* d_workspace, size_t workspace_bytes,
                      float* d_boxes, float* d_rows, int32_t* d_kept, int32_t* d_count, void* stream);
/* Dropout layers of the graph (creation order = the `ordinal` of csrc/byolo_rng.h), and where layer `ordinal`'s bits sit in
 * d_mask_bits for a (B, T) call: bit_offset (a multiple of 32) and element count S*h*w*cout; ordinal == byolo_num_dropout
 * returns the total length in bits (a multiple of 32) in bit_offset. */
BYOLO_API int32_t byolo_num_dropout(const byolo_t* h);
BYOLO_API int32_t byolo_mask_layout(byolo_t* h, int32_t B, int32_t T, int32_t ordinal, int64_t* bit_offset, int64_t* elements);
/* Numeric status of the handle's forwards in BYOLO_PREC_SPLIT_F16.  Every epilogue that encodes an activation tracks the
 * largest magnitude it stores, every decode launch checks the raw detection outputs; a hit raises a sticky device word:
 *   flags bit 0  an activation beyond the split-f16 range in layer `layer` (the first such layer; -1 otherwise)
 *   flags bit 1  a raw detection output is inf / NaN
 * byolo_status WAITS for `stream`, returns BYOLO_OK or BYOLO_ERR_RANGE (message: the layer's scope) and leaves the words
 * set; byolo_clear_status resets them (asynchronously, on `stream`).  byolo_set_async(h, 1): byolo_forward no longer waits
 * and checks -- a caller that synchronises anyway (to copy rows to the host) asks byolo_status once there.  Default 0. */
BYOLO_API int32_t byolo_set_async(byolo_t* h, int32_t on);
BYOLO_API int32_t byolo_status(byolo_t* h, void* stream, uint32_t* flags, int32_t* layer);
BYOLO_API int32_t byolo_clear_status(byolo_t* h, void* stream);
/* The dropout stream is indexed by the element index in the [S,h,w,c] tensor, S = images*T.  When one logical
 * batch is processed in several calls (sub-batches that respect byolo_max_images, or one shard per GPU), tell
 * every call where its first image sits in the logical batch: image j of the call then draws the masks of image
 * first_image + j, and the pieces equal the unsplit run.  Sticky per handle; 0 after byolo_create. */
BYOLO_API int32_t byolo_set_first_image(byolo_t* h, int64_t first_image);
/* T sharded over ranks -- the latency path for the reference's own default, ONE image per step (inference_epistemic.py:193
 * asserts batch_size == 1, :220 T = 50), where sharding the batch axis leaves N - 1 GPUs idle (the alternative):
 * every rank runs the backbone on the image and the heads on T_local = its share of the T_total MC samples.
 *   byolo_set_tshard(h, t0, T_total)   sticky; T_total = 0 switches it off.  A following byolo_forward(h, img, B = 1, T = T_local,
 *       ...) draws the dropout masks of samples t0 .. t0 + T_local - 1 of the image's T_total (the N ranks together draw what one
 *       call with T = T_total draws) and writes into d_boxes [B, N, 21 + C], instead of decoded rows, the per-box SUMS over its
 *       samples of the quantities lib_yolo/layers.py:377-395 reduces with reduce_mean: [sum l (4) | sum l l^T, upper triangle (10) |
 *       sum exp(logvar) (4) | sum sigmoid(obj) | sum H(sigmoid(obj)) | sum softmax(cls) (C) | sum H(softmax(cls))].  d_rows must be
 *       null (no NMS on sums).  Epistemic detection layers only.
 *   the caller adds the ranks' buffers element-wise (one all-reduce of N * (21 + C) floats: 2.1 MB at 608 x 608);
 *   byolo_finish_tshard(h, d_sums, B, T_total, stream)   turns the summed buffer IN PLACE into the rows a T_total-sample
 *       byolo_forward writes (means, population covariance, 4x4 determinant, entropies / mutual information, decoded corners);
 *       byolo_sort_nms then runs the tail.  The sums are added in another order than the one-call reduction: rows agree within
 *       float32 rounding (tested at the contract's bound 1e-4 * max(1, |ref|)), not bit for bit. */
BYOLO_API int32_t byolo_set_tshard(byolo_t* h, int32_t t0, int32_t T_total);
BYOLO_API int32_t byolo_finish_tshard(byolo_t* h, float* d_sums, int32_t B, int32_t T_total, void* stream);
/* Images ONE launch sequence carries at this T: the convolutions address their sources with 32-bit byte offsets, so every
 * activation tensor [B*T or B, h, w, c] of a piece stays below 3 GiB (18 images at 608x608, T=30).  byolo_forward cuts larger
 * batches into such pieces itself; the number matters to a caller that injects masks or reads byolo_layer_output. */
BYOLO_API int32_t byolo_max_images(byolo_t* h, int32_t T, int32_t* max_images);
/* After a forward with keep_all_outputs: device pointer + NHWC shape of layer `idx`'s output
 * (the reference's model.layers[idx], model.py:191); for detection layers the raw conv output
 * (DetLayer.raw_output, model.py:241) -- those are readable on any handle (they are never overwritten).
 * The pointer addresses the library's own storage: hi/lo pairs under BYOLO_PREC_SPLIT_F16, and a detection head's rows
 * are padded to a multiple of 4 floats (shape[3] = 21 / 42 / .. is the logical count): read tensors through
 * byolo_copy_layer_output, which hands out dense float32. */
BYOLO_API int32_t byolo_layer_output(const byolo_t* h, int32_t idx, const float** d_ptr, int64_t shape[4]);
/* The same tensor as float32 values, copied into caller-owned device memory d_dst[count] (count = the product of the
 * shape) on `stream`.  Under BYOLO_PREC_SPLIT_F16 the pointer of byolo_layer_output addresses the hi/lo pairs the
 * kernels exchange (only detection layers are plain float32 there): read activations through this call. */
BYOLO_API int32_t byolo_copy_layer_output(const byolo_t* h, int32_t idx, float* d_dst, int64_t count, void* stream);

/* ---- staged tail entry points (parity tests on oracle-provided inputs) ------------------------ */
/* decode one detection layer: d_raw [S,lh,lw,F] -> rows written at their concat_bbox position in
 * d_boxes [B,N_total,D] starting at box offset `box_base` (d_raw dense, F floats per cell); kind as above; EPISTEMIC reduces every
 * image's T samples (S = B*T). */
BYOLO_API int32_t byolo_decode(byolo_t* h, int32_t kind, const float* d_raw, int32_t B, int32_t T, int32_t lh, int32_t lw,
                     const float* priors_hw /*[3][2] host*/, int32_t layer_id, float* d_boxes,
                     int64_t n_total, int64_t box_base, void* stream);
/* The entries of decode_epistemic's dict (lib_yolo/layers.py:397-411) that are not columns of the box row, from the
 * raw output d_raw [B*T,lh,lw,3*2*(5+C)] of an epistemic detection layer, DENSE float32 (as byolo_copy_layer_output hands it
 * out; the library's own storage pads the rows, see byolo_layer_output): ev_loc [B,lh,lw,3,4]
 * (mean raw location logits), epi_covar_loc [B,lh,lw,3,4,4] (full covariance; its diagonal is columns 4..7 of the box
 * row), obj_samples [B*T,lh,lw,3] = sigmoid(obj), cls_samples [B*T,lh,lw,3,C] = softmax(cls).  Any output may be NULL.
 * (vis_uncertainty.py:49-163 and other consumers of DetLayer.det.) */
BYOLO_API int32_t byolo_epistemic_stats(byolo_t* h, const float* d_raw, int32_t B, int32_t T, int32_t lh, int32_t lw,
                                        float* d_ev_loc, float* d_epi_covar, float* d_obj_samples, float* d_cls_samples,
                                        void* stream);
/* tf.image.non_max_suppression + tf.gather per image on d_boxes [B,N,D] (scores = column obj_idx).
 * d_sort_ws: >= byolo_nms_workspace_bytes(B, N). */
BYOLO_API size_t  byolo_nms_workspace_bytes(int32_t B, int64_t N);
BYOLO_API int32_t byolo_sort_nms(byolo_t* h, const float* d_boxes, int32_t B, int64_t N, int32_t D, int32_t obj_idx,
                       int32_t cls_start_idx, int32_t nms_mode, int32_t max_out, float iou_thresh,
                       void* d_sort_ws, size_t ws_bytes,
                       float* d_rows, int32_t* d_kept, int32_t* d_count, void* stream);

/* ---- ground-truth encoding and training loss ------------------------
 * The reference builds both into its TRAINING graph; here they are two stand-alone device functions (the optimiser, batch-
 * statistics BN and back-propagation through the network stay out of scope).  `h` supplies the device and the error string
 * and may be NULL (current device, byolo_last_error(NULL)).
 *
 * byolo_encode_gt = lib_yolo/tfdata.py:77-171 `encode_boxes` (the map function of the training dataset,
 * lib_yolo/dataset_utils.py:58-63) for a batch of B images: boxes [B,max_boxes,4] as (ymin, xmin, ymax, xmax) image
 * fractions, labels [B,max_boxes], counts [B] (NULL: every image has max_boxes boxes); detection layers as
 * layer_hw [n_layers][2] and priors_hw [n_layers][3][2] = (h, w) DOUBLES (the Python floats of lib_yolo/yolov3.py:6-166: the
 * prior boxes are computed in double and rounded to float32 like lib_yolo/data.py:125-166 does).  Outputs, per image the prior
 * boxes of all layers back to back, inside a layer in the reference's [row, col, box] order (N = sum of lh * lw * 3): loc [B,N,4]
 * (logit / log targets), obj [B,N], cls [B,N] (int32), ign [B,N]; layer k's slice reshapes to the reference's
 * gt_k['loc'] [lh,lw,3,4] etc.  Later boxes overwrite earlier ones that claim the same prior box, as the reference's
 * sequential tf.while_loop does.  The masks are bit-identical to a float32 evaluation of the reference's formulas.  d_boxes 16-byte
 * aligned; d_workspace >= byolo_encode_gt_workspace_bytes(B, max_boxes) (every box's maximum IoU over all prior boxes). */
BYOLO_API int32_t byolo_encode_gt(byolo_t* h, int32_t n_layers, const int32_t* layer_hw, const double* priors_hw,
                                  const float* d_boxes, const int32_t* d_labels, const int32_t* d_counts, int32_t B,
                                  int32_t max_boxes, float ign_thresh, float* d_loc, float* d_obj, int32_t* d_cls,
                                  float* d_ign, void* d_workspace, size_t workspace_bytes, void* stream);
BYOLO_API size_t  byolo_encode_gt_workspace_bytes(int32_t B, int32_t max_boxes);
/* byolo_loss = lib_yolo/layers.py:126-188 `loss_tf` for ONE detection layer on its raw output d_raw [S,lh,lw,pitch]
 * (pitch = floats per cell, 0 = dense; the library's own storage pads, see byolo_layer_output), split as
 * lib_yolo/layers.py:11-84 does: kind BYOLO_DET_STANDARD (loc 4, obj, cls C per prior) or BYOLO_DET_ALEATORIC (loc 4,
 * log_loc_var 4, obj, log_obj_stddev, cls C, log_cls_stddev C); aleatoric_loss as the reference's flag (lib_yolo/layers.py:150-153,
 * log variance clipped to [-40, 40]).  Ground truth as byolo_encode_gt lays it out: pointers at the layer's first prior box,
 * gt_stride = prior boxes between consecutive images (N of byolo_encode_gt, or lh * lw * 3 for per-layer arrays).
 * d_loss[3] (double) = loc, obj, cls: sum / (2 S), sum / S, sum / S.  d_grad (or NULL) [S,lh,lw,grad_pitch] receives
 * d(loc + obj + cls) / d(raw) -- what lib_yolo/train.py:88 `optimizer.minimize` would send into the network.  Terms are
 * float32 like the graph's, sums double in a fixed order: results are deterministic.  d_workspace >= byolo_loss_workspace_bytes(). */
BYOLO_API size_t  byolo_loss_workspace_bytes(void);
BYOLO_API int32_t byolo_loss(byolo_t* h, int32_t kind, int32_t aleatoric_loss, int32_t cls_cnt, const float* d_raw, int32_t pitch,
                             int32_t S, int32_t lh, int32_t lw, const float* d_gt_loc, const float* d_gt_obj,
                             const int32_t* d_gt_cls, const float* d_gt_ign, int64_t gt_stride, double* d_loss,
                             float* d_grad, int32_t grad_pitch, void* d_workspace, size_t workspace_bytes, void* stream);

/* ---- synthetic-weight support: data-dependent BN initialisation (no reference counterpart;
 * replaces "train the network" for benchmarking with random-init weights): runs the graph on
 * d_img with dropout off, sets every BN's moving_mean/variance to the batch statistics of its conv
 * output layer by layer, re-folds.  Read the result back with byolo_get_param. */
BYOLO_API int32_t byolo_calibrate_bn(byolo_t* h, const float* d_img, int32_t B, void* d_workspace, size_t workspace_bytes,
                           void* stream);

/* ---- profiling hooks: per-stage device time of the LAST forward (hipEvents on `stream`);
 * enable with byolo_set_profiling(h, 1).  stage: 0 backbone, 1 heads, 2 decode, 3 sort+nms. */
BYOLO_API int32_t byolo_set_profiling(byolo_t* h, int32_t on);   /* 0 off, 1 per stage, 2 + per conv launch */
/* The same switch WITHOUT discarding the records already taken (byolo_set_profiling invalidates them): a caller that times a
 * run of forwards records every n-th one (an event costs the GPU a few microseconds; ~95 per forward) and reads them all afterwards. */
BYOLO_API int32_t byolo_resume_profiling(byolo_t* h, int32_t on);
BYOLO_API int32_t byolo_stage_ms(byolo_t* h, float ms[4]);
/* Keep the records of the last `depth` profiled forwards (default 1) and choose which one byolo_stage_ms /
 * byolo_num_steps / byolo_step_profile / byolo_step_split read: age 0 = the last forward, 1 = the one before, ...
 * A caller timing K back-to-back forwards sets depth = K and reads every record after the run -- no host
 * synchronisation inside the timed region (bench.py). */
BYOLO_API int32_t byolo_set_profile_depth(byolo_t* h, int32_t depth);
BYOLO_API int32_t byolo_select_profile(byolo_t* h, int32_t age);
/* per kernel launch of the convolution stack in the LAST forward (profiling level 2): graph layer, kernel variant --
 *   128 / 64 / 32   conv_igemm_kernel, BN of the tile (split precision: + 1000 the general loop, + 2000 the 1x1 loop, + 3000 the
 *                   shared-tap 3x3 loop; 140 / -4 Winograd in split arithmetic: fused launch / input transform)
 *                                                                -1   the direct small-Cin kernels
 *   -2 / -3         Winograd input / output transform            129  the row-streaming Winograd-domain GEMM
 *   130             the same with output transform + epilogue    131 / 132  a 1x1 convolution / detection head as a
 *                   fused in (wino_fused_kernel)                            row-streaming launch, 128- / 64-wide tile
 * -- the EXECUTED extents {M, N, K} (for a Winograd-domain GEMM: 16 * tiles rows, cout, cin), the launch's device time
 * (hipEvents on `stream` around it) and the ALGORITHMIC FLOPs it stands for (2*M*N*K of the layer as written; differs
 * from the executed work for the T-invariant de-duplicated launches -- conv once per image + T masked epilogues; the
 * per-image partial sum of a concat's tiled half carries 0 -- and for Winograd, where the GEMM launch carries the
 * direct-convolution FLOPs of its samples and the transforms carry 0).
 * byolo_num_steps = launches of the last profiled forward. */
BYOLO_API int32_t byolo_num_steps(const byolo_t* h);
BYOLO_API int32_t byolo_step_profile(byolo_t* h, int32_t i, int32_t* layer, int32_t* variant, int64_t mnk[3], float* ms,
                                     double* algo_flops);
/* the same launch's split-K plan: K slices per tile of its last partial round of tiles (1 = not split) and how many
 * tiles were cut; a NEGATIVE ksplit -G = a stream-K launch: G workgroups share all tiles * K-tiles units evenly
 * (conv_igemm.hip; decided per (B, T) shape, deterministic).  A Winograd GEMM entry (variant 140): ksplit 1, split_tiles = the
 * output channels per workgroup the planner chose (256 | 128). */
BYOLO_API int32_t byolo_step_split(byolo_t* h, int32_t i, int32_t* ksplit, int32_t* split_tiles);
/* analytic cost of one forward: conv FLOPs (2*MAC, graph as written) for B images x T samples */
BYOLO_API int32_t byolo_flops(byolo_t* h, int32_t B, int32_t T, double* flops);

/* ---- input feed helper (host only): CRC-32C of the TFRecord framing read by the reference's
 * tf.data.TFRecordDataset (lib_yolo/dataset_utils.py:188-199); returns the raw (unmasked) CRC. */
BYOLO_API uint32_t byolo_crc32c(const void* h_data, size_t n);


/* ---- input feed, device side: `tf.image.convert_image_dtype(img, tf.float32)` of decode_img
 * (lib_yolo/dataset_utils.py:6-11) on the device -- d_f32[i] = float(d_u8[i]) * (1 / 255) in fp32, bit for bit the value the
 * host conversion produces -- so that the feed ships one byte per pixel-channel over PCIe instead of four.  n = elements
 * (B * H * W * C); d_f32 is what byolo_forward takes as d_img. */
BYOLO_API int32_t byolo_normalize_u8(byolo_t* h, const uint8_t* d_u8, int64_t n, float* d_f32, void* stream);

/* The two status words (byolo_status: {flags, layer}) copied device-to-device into d_out[2] on `stream`, without waiting: a
 * multi-GPU driver appends them to the buffer its ONE all-gather carries, so that every rank learns of any rank's
 * BYOLO_ERR_RANGE from the gathered list and all ranks switch to the fp32 mode together (byolo/inference.py). */
BYOLO_API int32_t byolo_copy_status(byolo_t* h, uint32_t* d_out, void* stream);

/* ---- input feed, host side: `tf.image.decode_png(encoded, dtype=tf.uint8)` of decode_img for the n records of a batch on
 * `threads` host threads -- the `map(parse_example, num_parallel_calls=config['cpu_thread_cnt'])` stage of TestingDataset
 * (lib_yolo/dataset_utils.py:196).  h_out: n frames of img_h * img_w * img_c bytes (e.g. pinned memory); status[i] per record:
 * OK, UNSUPPORTED (interlaced / palette / 16-bit: the caller decodes that record with a general decoder), SHAPE (the PNG is
 * not img_h x img_w x img_c; found_shape[3*i..] = what it is, like the reference's set_shape failure) or CORRUPT (signature,
 * chunk CRC, zlib stream, filter type).  Returns the number of records that are not OK, or BYOLO_ERR_ARG. */
enum { BYOLO_PNG_OK = 0, BYOLO_PNG_UNSUPPORTED = 1, BYOLO_PNG_SHAPE = 2, BYOLO_PNG_CORRUPT = 3 };
BYOLO_API int32_t byolo_png_decode_batch(const uint8_t* const* h_png, const size_t* png_bytes, int32_t n, int32_t img_h,
                                         int32_t img_w, int32_t img_c, uint8_t* h_out, int32_t threads, int32_t* status,
                                         int32_t* found_shape);

/* The whole map stage for the n records of a batch, natively: record i = lengths[i] payload bytes at offsets[i] of the open file
 * fds[i] (TFRecord framing: the payload is followed by its masked CRC-32C, checked if verify_crc) -> tf.train.Example -> the
 * first bytes value of `image/encoded` decoded as above into frame i of h_out, the first value of `image/filename` into
 * h_names[i * name_cap ...] (NUL-terminated UTF-8).  status[i] = BYOLO_PNG_* or BYOLO_FEED_IO (short read) / _CRC / _PROTO
 * (malformed Example, no image, a name of name_cap bytes or more).  Returns the number of records that are not OK. */
enum { BYOLO_FEED_IO = 4, BYOLO_FEED_CRC = 5, BYOLO_FEED_PROTO = 6 };
BYOLO_API int32_t byolo_feed_records(const int32_t* fds, const int64_t* offsets, const int64_t* lengths, int32_t n, int32_t verify_crc,
                                     int32_t img_h, int32_t img_w, int32_t img_c, uint8_t* h_out, int32_t threads, char* h_names,
                                     int32_t name_cap, int32_t* status, int32_t* found_shape);

/* ---- output writer, host side: the text `json.dump({'children': [bbox_to_ecp_format(b, ...) for b in boxes]}, f)` writes for
 * the kept rows of ONE image (inference_epistemic.py:84-92 + :131-170; inference_aleatoric.py:139-178;
 * inference_standard_yolov3.py:128-150), byte for byte: keys in the reference's order, coordinates scaled in float32 then
 * widened, score = obj * cls[argmax] in double, every number as CPython's float repr (shortest round-trip digits, NaN /
 * Infinity spelt as json.dump does), the index quirks of the reference kept (aleatoric: cls_entropy / layer_id / prior_id all
 * read column cls_start + C; epistemic: ped_score / rider_score = columns 17 / 18).  kind = BYOLO_DET_*; h_rows [n_rows,
 * row_len]; identity = labels[argmax + implicit_background] if that entry exists and is not NULL (ASCII without quotes /
 * backslashes), else the integer.  Returns the length of the text; if it exceeds cap nothing useful is in h_out and the
 * return value is -(length) - 16. */
BYOLO_API int64_t byolo_format_ecp_json(int32_t kind, const float* h_rows, int32_t n_rows, int32_t row_len, int32_t img_h,
                                        int32_t img_w, int32_t cls_cnt, int32_t obj_idx, int32_t cls_start,
                                        int32_t implicit_background, const char* const* labels, int32_t n_labels,
                                        char* h_out, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* BYOLO_H_ */
