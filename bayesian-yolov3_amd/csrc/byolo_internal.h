// byolo_internal.h -- the handle behind include/byolo.h and what the translation units of the library share:
//   byolo_api.hip   handle, graph builder (lib_yolo/model.py ModelBuilder), parameter store, lowering of the layer list to steps,
//                   the forward driver and the staged / profiling entry points
//   byolo_pack.hip  byolo_finalize: BN folding, weight packing for the MFMA tiles (fp32 and split-f16, Winograd U), upload
//   byolo_plan.hip  the workspace planner: liveness-based arena, launch geometry per step, Winograd chunks, back-to-back
//                   fusion decisions -- host code only -- and its introspection entry points (tests/test_planner.py)
// (round 5: byolo_api.hip used to be one 2 400-line translation unit)
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/byolo.h"
#include "byolo_kernels.h"
#include "byolo_rng.h"

using namespace byk;

namespace byi {


enum Op { OP_CONV, OP_RESIDUAL, OP_ROUTE, OP_UPSAMPLE, OP_STACK, OP_DETECTION };

struct Param {
    std::string name;
    std::vector<int64_t> shape;
    std::vector<float> data;
    int64_t count() const { int64_t c = 1; for (auto s : shape) c *= s; return c; }
};

struct Layer {
    Op op;
    std::string scope;
    int filters = 0, ksize = 0, stride = 1, norm = 0;
    int prev = -1;                 // implicit input (previous layer; -1 = image)
    int ref[2] = {-1, -1};         // explicit absolute refs (shortcut / routes / stack src)
    int nref = 0;
    int det_kind = 0, det_id = 0;
    float priors[6] = {0, 0, 0, 0, 0, 0};
    // inferred output
    int C = 0, H = 0, W = 0;
    bool stacked = false;
    // params
    int p_kernel = -1, p_bias = -1, p_gamma = -1, p_beta = -1, p_mean = -1, p_var = -1;
    int drop_ordinal = -1;
    int Cin = 0;
    // lowering
    bool materialized = false;     // owns an activation tensor
    int out_tensor = -1;           // layer index whose tensor receives this conv's output
    int fused_residual = -1;       // residual layer fused into this conv's epilogue
    bool standalone = false;       // residual layer computed by its own element-wise step (STEP_ADD)
    int add_a = -1;                // ... its left operand (tensor id)
    // packed weights (offsets in floats into the device blob)
    size_t w_off = 0, scale_off = 0, shift_off = 0;
    size_t scalek_off = 0;             // dropout layers: scale / (1 - p), what the epilogue multiplies with when the masks are on
    int tile = 0, Npad = 0;
    bool direct = false;
    std::vector<int> wshift;       // split precision, per OUTPUT CHANNEL n: the packed weights hold w[.][n] * 2^wshift[n] (the channel's largest |w'| in [2^13, 2^14)); folded into scale[n]
    std::vector<int> wshift_u;     // split precision, Winograd (wino_split.hip): per output channel, for U = G g G^T
    size_t wscale_off = 0, wscalek_off = 0;    // ... and the per-channel scale arrays that go with it (the shift is the layer's)
    float in_scale = 1.f;          // split precision: scale of the layer's input (ACT_SCALE for activations, 1 for the fp32 image of a direct convolution)
    int64_t box_base = 0;
};

struct Src { int layer; int C; int sh; bool tile; };
struct View { Src s[2]; int n = 0; };

// One conv launch.  Normally one per conv/detection layer; the T-invariant de-duplication
//  lowers some layers of a stacked (MC-sample) graph differently:
//   STEP_REP      every input is a T-fold tile of an unstacked tensor and the layer has dropout: the
//                 conv runs once per image, the epilogue is replayed for the T samples (masks differ);
//   STEP_PARTIAL  the tiled (T-invariant) half of a channel concat, convolved once per image into an
//                 auxiliary raw-accumulator tensor ...
//   STEP_MAIN     ... which the conv over the stacked half picks up as an addend before scale / mask.
// Two more step kinds keep the builder general (the reference's models never need them):
//   STEP_GATHER   a route / upsample / stack VIEW that a loader cannot express on the fly (the inner view of a nested
//                 concat or double upsample, a view used as a residual shortcut) is copied into a tensor of its own;
//   STEP_ADD      a residual add that cannot ride in a convolution's epilogue (its left operand is not a convolution,
//                 or that convolution's output has other readers) runs as an element-wise kernel.
// And one that the reference's Bayesian model does need (round 4): the stacked half of its two concat convolutions is an
// UPSAMPLED tensor, and a 1x1 convolution commutes with nearest-neighbour upsampling -- its GEMM belongs at the source's
// resolution, a quarter of the rows:
//   STEP_PARTIAL with `low`  the stacked half, convolved per SAMPLE at the source's resolution into an auxiliary raw-accumulator
//                 tensor [S, H/2, W/2, N] (it also owns the layer's scale / shift arrays);
//   STEP_FINISH   output pixel (s, y, x) = epilogue(low[s, y/2, x/2] + partial[image, y, x]): an element-wise kernel
//                 (conv_kernels.hip finish_upsampled_kernel).  The same two numbers added in the same order as STEP_MAIN's
//                 accumulator + addend, the same epilogue arithmetic: the same bits.  BYOLO_LOWMAIN=0 keeps STEP_MAIN.
enum StepMode { STEP_NORMAL = 0, STEP_REP = 1, STEP_PARTIAL = 2, STEP_MAIN = 3, STEP_GATHER = 4, STEP_ADD = 5, STEP_FINISH = 6 };
struct Step {
    int layer; View in;
    int mode = STEP_NORMAL;
    bool is_conv() const { return mode <= STEP_MAIN; }
    int c_lo = 0, c_hi = 0;        // input-channel range of the layer's Cin this launch convolves
    int out_tensor = -1;           // tensor id written (layer index, or n_layers + aux index)
    int addend_tensor = -1;        // STEP_MAIN / STEP_FINISH: the PARTIAL result
    bool low = false;              // STEP_PARTIAL: the stacked half at the source's resolution (per sample, output H/2 x W/2)
    int low_tensor = -1;           // STEP_FINISH: that launch's result
    size_t w_off = 0; int Npad = 0, tile = 0;   // packed weights of this launch
    bool wino_ok = false;          // 3x3 / stride 1 over one plain source: Winograd F(2x2,3x3) is possible
    size_t wino_off = 0;           // the 16 transformed weight matrices U[xi], each packed [Cin/32][Npad][32]
    bool p1 = false;               // split precision: 1x1 / stride 1 over one plain source -- the uniform loop of conv_tile_p1 (BYOLO_P1=0: the general loop)
    bool kx3 = false;              // split precision: 3x3 / stride 1 over one plain source -- shared-tap stages (conv_tile_kx3), weights in (ky, chunk, kx) order
};
// per-(B, T) decision for a Winograd-capable step: samples per chunk (0 = direct convolution)
struct WinoPlan { int chunk = 0; int th = 0, tw = 0; size_t v_bytes = 0, m_bytes = 0; bool fused = false; int bm = 0, bn = 0; /* split precision: output tiles / channels per workgroup */ };
struct AuxTensor { int H, W, C; bool stacked = false; };      // stacked: one row per SAMPLE pixel (else per image pixel)

struct Plan {
    int B = -1, T = -1;
    std::vector<int64_t> off;      // per layer tensor offset in bytes (-1: none)
    size_t arena = 0, boxes_off = 0, nms_off = 0, stats_off = 0, total = 0;
    size_t img_split_off = 0;      // split precision: the image as hi/lo pairs, for a matrix-pipe convolution that reads it (img_c % 32 == 0)
    size_t slab_off = 0, slab_bytes = 0, cnt_off = 0, cnt_bytes = 0;   // split-K slabs (shared by all steps), per-step ticket counters
    std::vector<ConvSplit> split;  // per step
    std::vector<int> tile;         // per step: tile configuration of the launch
    std::vector<WinoPlan> wino;    // per step
    std::vector<int> stream1x1;    // per step: tile width of the row-streaming 1x1 launch (gemm_stream.hip), 0 = conv_igemm
    std::vector<char> fuse;        // per step: the NEXT step (a 1x1 convolution / detection head reading only this output) runs inside this launch
    size_t wino_off = 0;           // scratch for V and M of one chunk (shared by all steps)
};
static constexpr int CNT_PER_STEP = 1024;      // >= resident workgroups of any tile configuration
// split precision: every activation tensor holds ACT_SCALE * value, so that the lo half of a value >= 2^-4 is a normal
// fp16 (smaller values keep an absolute error of 2^-25 / ACT_SCALE = 7.5e-9 -- the fp32 rounding of a value of 0.125);
// an activation beyond 65504 / ACT_SCALE = 16376 overflows to infinity.  A power of two: folded into scale / shift
// exactly.  (Measured at 608x608, T=4 against the float64 oracle: scale 1, 4, 16 are indistinguishable -- DESIGN.md 5.)
static constexpr float ACT_SCALE = 4.f;


}  // namespace byi
using namespace byi;


struct byolo {
    byolo_cfg cfg;
    int device = 0;
    std::string err;
    std::vector<Layer> layers;
    std::vector<Param> params;
    std::map<std::string, int> pindex;
    int n_dropout = 0, n_det = 0;
    int backbone_end = -1;
    bool finalized = false;        // weights folded, packed and uploaded
    bool lowered = false;          // graph frozen and lowered to steps (host only)
    std::vector<char> need_mat;    // per layer: this view is copied into a tensor of its own (STEP_GATHER)
    mutable int want_mat = -1;     // lowering: the view whose materialisation would resolve the last failure
    std::vector<Step> steps;
    std::vector<AuxTensor> aux;    // auxiliary tensors (ids n_layers + k): partial sums of split convs
    byolo_plan_opts opts;          // include/byolo.h: defaults + the environment at byolo_create, byolo_set_plan_opts afterwards
    // Launch graphs of whole forwards (opts.graphs; byolo_api.hip forward_graph): keyed by everything a launch's arguments depend on
    struct GraphKey {
        const void* d_img; const void* d_mask_bits; const void* d_workspace; const void* d_boxes; const void* d_rows; const void* d_kept; const void* d_count;
        int32_t B, T, dropout_on, precision, plan_epoch, tshard_t0, tshard_T;
        bool same(const GraphKey& o) const { return !memcmp(this, &o, sizeof *this); }
    };
    struct GraphEntry { GraphKey key; uint64_t seed; int64_t first_image; hipGraphExec_t exec = nullptr; uint64_t used = 0; int seen = 0; bool no_graph = false; };
    std::vector<GraphEntry> graphs;
    hipStream_t cap_stream = nullptr;  // the capture stream (created with the first capture)
    uint64_t graph_clock = 0; int64_t graph_replays = 0, graph_captures = 0, graph_updates = 0;
    bool dedup = true;             // T-invariant de-duplication (opts.dedup)
    // Arithmetic of the convolution stack (byolo_set_precision; BYOLO_PRECISION=f32|split):
    //   0  fp32 operands on v_mfma_f32_32x32x2_f32 (+ Winograd F(2x2,3x3) where it pays)
    //   1  split-f16 operands ("hi + lo", ~23 significant bits, fp32 accumulation) on v_mfma_f32_32x32x16_f16:
    //      activations live in memory as [4 hi | 4 lo] groups holding ACT_SCALE * value, weights as 2^wshift * w
    bool img_split = false;        // split precision: some matrix-pipe convolution reads the image -> a hi/lo copy is made per forward
    int precision = 1;             // default: split-f16 (BYOLO_PRECISION=f32 selects the fp32 matrix instruction)
    int prec_requested = 1;        // what byolo_set_precision / BYOLO_PRECISION asked for
    std::string prec_note;         // why byolo_finalize fell back to BYOLO_PREC_F32 (empty: it did not)
    // Numeric status (byolo_status): two device words {flags, first layer} every split-f16 epilogue / decode launch of this
    // handle may raise (sticky until byolo_clear_status), and their pinned host mirror
    unsigned* d_status = nullptr; unsigned* h_status = nullptr;
    bool async_status = false;     // byolo_set_async: byolo_forward does not wait for the status words
    bool plan_inject = false;      // the current plan was made for injected dropout masks (fp32 mode: conv_igemm launches only)
    int plan_epoch = 0;            // bumped whenever the plan is invalidated (precision, finalize)
    int wsm_B = -1, wsm_T = -1, wsm_epoch = -1; size_t wsm_total = 0;     // byolo_workspace_bytes: size of the masked-call plan of (B, T)
    // Forwards of ONE handle alternating over several streams (a caller pipelining whole steps: bench.py --pipeline): the
    // convolution stacks run one after the other -- two of them sharing the chip gain nothing and blur every per-launch timing
    // -- while a step's latency-bound tail (decode, sort, NMS) overlaps the next step's convolutions.  ev_convs is recorded
    // behind the last convolution launch of a forward; a forward on ANOTHER stream waits for it before its first launch.
    hipEvent_t ev_convs = nullptr; hipStream_t convs_stream = nullptr; bool ev_convs_valid = false;
    bool quiet_next = false;       // the forward before this one was recorded at profiling level 2: this one does not run beside its heads
    std::vector<int> last_use;     // per tensor id: index of the last step reading it
    float* d_blob = nullptr;       // packed weights + scale/shift
    size_t blob_floats = 0;
    float* d_ones = nullptr; float* d_zeros = nullptr; int maxC = 0;
    int64_t n_boxes = 0; int row_len = 0, obj_idx = 0, cls_start = 0;
    Plan plan;
    void* last_ws = nullptr;
    int64_t first_image = 0;       // position of a call's first image in the logical batch (dropout stream)
    int tshard_t0 = 0, tshard_T = 0;   // byolo_set_tshard: this call's T samples are samples t0 .. t0 + T - 1 of tshard_T per image (0 = off)
    int profiling = 0;             // 0 off, 1 stage events, 2 + one event per conv launch
    // level 2: one entry per kernel launch of the convolution stack in a forward (a Winograd layer
    // contributes input transform / GEMM / output transform per chunk); event k is recorded before launch k
    // ev_begin / ev_end: indices into the slot's event pool (the end of a launch is the begin of the next one)
    struct Launch { int layer, variant; int64_t m, n, k; double algo_flops; int ksplit, split_tiles; int ev_begin, ev_end; };
    // The records of the last `depth` profiled forwards (byolo_set_profile_depth; 1 by default): a caller that times a
    // run of back-to-back forwards reads all of them AFTER the run instead of synchronising with every step.
    struct ProfSlot {
        hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        bool ev_valid = false;
        std::vector<Launch> launches;
        std::vector<hipEvent_t> step_ev;   // pool; n_ev in use
        int n_ev = 0, last_main = -1;      // events used by this forward; the last launch marked (its end = the next one's begin)
        bool step_valid = false;
    };
    std::vector<ProfSlot> prof = std::vector<ProfSlot>(1);
    int prof_w = 0;                    // slot of the most recent profiled forward
    int prof_age = 0;                  // which forward the read calls refer to: 0 = the last, 1 = the one before, ...
    ProfSlot& wslot() { return prof[prof_w]; }
    ProfSlot& rslot() { const int d = (int)prof.size(); return prof[((prof_w - prof_age) % d + d) % d]; }
};


// ---- shared helpers ------------------------------------------------------------------------------------------------------
int32_t byolo_fail(byolo_t* h, int32_t code, const char* fmt, ...);
#define fail byolo_fail
#define HIPCHK(h, call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(h, BYOLO_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline float* dptr(const byolo_t* h, size_t off) { return h->d_blob + off; }

// No C++ exception leaves the C-ABI: a host allocation that fails (a graph of absurd sizes) is BYOLO_ERR_NOMEM, like a workspace
// that is too small -- the caller is a ctypes / cgo / JNI binding that cannot unwind.
template <class F>
static int32_t guarded(byolo_t* h, const char* what, F&& f) {
    try { return f(); }
    catch (const std::bad_alloc&) { return fail(h, BYOLO_ERR_NOMEM, "%s: out of host memory", what); }
    catch (const std::exception& e) { return fail(h, BYOLO_ERR_ARG, "%s: %s", what, e.what()); }
    catch (...) { return fail(h, BYOLO_ERR_ARG, "%s: an exception that is not a std::exception", what); }      // nothing crosses the C-ABI (include/byolo.h)
}


#define lower byolo_lower
int32_t byolo_lower(byolo_t* h);                                   // byolo_api.hip: freeze the graph, lower it to steps (host only)
// byolo_pack.hip
void fold_layer(const byolo_t* h, const Layer& l, std::vector<float>& scale, std::vector<float>& shift);
float acc_scale_of(const Layer& l, int n);
void fold_split(const Layer& l, std::vector<float>& sc, std::vector<float>& sf);
void scale_keep(const byolo_t* h, std::vector<float>& sc);
void wino_scales(const byolo_t* h, const Layer& l, std::vector<float>& sc, std::vector<float>& sk);
// byolo_plan.hip
int layer_pitch(const Layer& l);
int64_t tensor_bytes(const byolo_t* h, int id, int B, int T);
void step_geometry(const byolo_t* h, const Step& st, int B, int T, int* M, int* KT);
void decide_loops(byolo_t* h);
void make_plan(byolo_t* h, int B, int T, bool inject = false);
int64_t piece_cap(const byolo_t* h, int32_t T);
int32_t check_run(byolo_t* h, int32_t B, int32_t T, const char* what, bool need_device = true);
