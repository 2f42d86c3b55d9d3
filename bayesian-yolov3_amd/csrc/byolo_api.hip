// byolo_api.hip -- implementation of include/byolo.h: graph builder (mirror of
// lib_yolo/model.py ModelBuilder), parameter store, lowering of the reference's layer list to steps,
// and the forward driver that turns the steps into fused gfx950 kernel launches on the caller's stream.
// (BN folding + weight packing: byolo_pack.hip; the liveness-based workspace planner: byolo_plan.hip;
//  the handle and what the three share: byolo_internal.h.)
//
// Lowering rules (what the TF graph of lib_yolo/yolov3.py:518-628 becomes):
//   conv -> [dropout] -> bn -> leaky                      one conv_igemm launch (fused epilogue)
//   conv3x3 followed by residual(-3)                      same launch, residual added in the epilogue
//   upsample / route / stack_feature_map                  never materialised: they become the operand
//                                                          view (two sources, x>>1 indexing, s/T
//                                                          sample broadcast) of the consuming conv
//   detection conv (+bias) -> split -> decode             conv_igemm (bias epilogue) + one decode launch
//                                                          writing rows at their concat_bbox offset
//   concat_bbox -> non_max_suppression -> gather          sort_keys + nms launches
#include "byolo_internal.h"

static thread_local std::string g_err;

int32_t byolo_fail(byolo_t* h, int32_t code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    if (h) h->err = buf; else g_err = buf;
    return code;
}

// ------------------------------------------------------------------------------------------------
extern "C" const char* byolo_version(void) { return "byolo 0.7 (gfx950; fp32 MFMA and split-f16 MFMA; abi 7)"; }
extern "C" int32_t byolo_abi_version(void) { return BYOLO_ABI_VERSION; }

extern "C" const char* byolo_last_error(const byolo_t* h) { return h ? h->err.c_str() : g_err.c_str(); }

static void plan_opts_defaults(byolo_plan_opts& o);
static void plan_opts_from_env(byolo_plan_opts& o);

extern "C" int32_t byolo_create(const byolo_cfg* cfg, int32_t device, byolo_t** out) {
    if (!cfg || !out) return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: null argument");
    if (cfg->img_h <= 0 || cfg->img_w <= 0 || cfg->img_c <= 0)
        return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: invalid image size");
    // lib_yolo/yolov3.py:207-208: input size must be a multiple of the biggest stride
    if (cfg->img_h % 32 || cfg->img_w % 32)
        return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: full_img_size must be a multiple of 32 (yolov3.py:207-208)");
    if (cfg->cls_cnt < 1 || cfg->cls_cnt > byk::BYOLO_MAX_CLASSES)
        return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: cls_cnt outside 1 .. 128");
    if (cfg->max_out < 1 || cfg->max_out > 2048) return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: max_out out of [1,2048]");
    if (!(cfg->drop_prob >= 0.f && cfg->drop_prob < 1.f)) return fail(nullptr, BYOLO_ERR_ARG, "byolo_create: drop_prob");
    byolo_t* h = new (std::nothrow) byolo();
    if (!h) return fail(nullptr, BYOLO_ERR_NOMEM, "byolo_create: out of host memory");
    h->cfg = *cfg;
    h->device = device;
    plan_opts_defaults(h->opts);
    plan_opts_from_env(h->opts);
    h->dedup = h->opts.dedup != 0;
    if (const char* e = getenv("BYOLO_PRECISION")) h->precision = (!strcmp(e, "f32") || !strcmp(e, "0")) ? 0 : 1;
    h->prec_requested = h->precision;
    *out = h;
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// plan options (include/byolo.h byolo_plan_opts)
// ------------------------------------------------------------------------------------------------
static void plan_opts_defaults(byolo_plan_opts& o) {
    memset(&o, 0, sizeof o);
    o.struct_bytes = (int32_t)sizeof o;
    o.graphs = 1; o.serialize_convs = 1; o.serialize_heads = 1; o.dedup = 1; o.lowmain = 1; o.kx3 = 1; o.p1 = 1; o.b2b = 1; o.kx3_wide = 0;
    o.wino_split = 1; o.wino_split_min_c = 256; o.wino_split_bn = 0; o.wino_split_rounds = 0;
    o.winograd = 1; o.wino_fused = 1; o.stream1x1 = 1; o.gemm_stream = 1; o.ksplit = -1; o.streamk = 1; o.plain_epilogue = 1; o.wino_split_persist = 0;
    o.wino_split_min_gflop = 30.f; o.wino_split_chunk_mb = 1500.f; o.wino_min_gflop = 10.f; o.wino_chunk_mb = 800.f; o.wino_min_ratio = 80.f;
}
// The environment is the default filler of a NEW handle and nothing else: the A/B scripts under tools/ and the tests set a variable,
// then build their model.  No other translation unit of the library reads a plan variable.
static void plan_opts_from_env(byolo_plan_opts& o) {
    auto geti = [](const char* name, int32_t& v) { if (const char* e = getenv(name)) v = atoi(e); };
    auto getf = [](const char* name, float& v) { if (const char* e = getenv(name)) v = (float)atof(e); };
    geti("BYOLO_GRAPHS", o.graphs); geti("BYOLO_SERIALIZE_CONVS", o.serialize_convs); geti("BYOLO_SERIALIZE_HEADS", o.serialize_heads);
    if (const char* e = getenv("BYOLO_NO_DEDUP")) o.dedup = atoi(e) ? 0 : 1;
    geti("BYOLO_LOWMAIN", o.lowmain); geti("BYOLO_KX3", o.kx3); geti("BYOLO_P1", o.p1); geti("BYOLO_B2B", o.b2b); geti("BYOLO_KX3_WIDE", o.kx3_wide);
    geti("BYOLO_WINO_SPLIT", o.wino_split); geti("BYOLO_WINO_SPLIT_MIN_C", o.wino_split_min_c); geti("BYOLO_WINO_SPLIT_BN", o.wino_split_bn);
    geti("BYOLO_WINO_SPLIT_ROUNDS", o.wino_split_rounds); geti("BYOLO_WINOGRAD", o.winograd);
    geti("BYOLO_WINO_FUSED", o.wino_fused); geti("BYOLO_STREAM1X1", o.stream1x1); geti("BYOLO_GEMM_STREAM", o.gemm_stream);
    geti("BYOLO_KSPLIT", o.ksplit); geti("BYOLO_STREAMK", o.streamk); geti("BYOLO_PLAIN_EPILOGUE", o.plain_epilogue); geti("BYOLO_WINO_SPLIT_PERSIST", o.wino_split_persist);
    geti("BYOLO_WSHIFT_PER_LAYER", o.wshift_per_layer); geti("BYOLO_NMS_GENERAL", o.nms_general);
    getf("BYOLO_WINO_SPLIT_MIN_GFLOP", o.wino_split_min_gflop); getf("BYOLO_WINO_SPLIT_CHUNK_MB", o.wino_split_chunk_mb);
    getf("BYOLO_WINO_MIN_GFLOP", o.wino_min_gflop); getf("BYOLO_WINO_CHUNK_MB", o.wino_chunk_mb); getf("BYOLO_WINO_MIN_RATIO", o.wino_min_ratio);
}

static void drop_graphs(byolo_t* h) {
    bool any = false;
    for (auto& g : h->graphs) any = any || g.exec;
    if (any) (void)hipDeviceSynchronize();                  // an executable graph may still be replaying on some stream of the caller's
    for (auto& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

extern "C" int32_t byolo_get_plan_opts(const byolo_t* h, byolo_plan_opts* out) {
    if (!h || !out) return fail(const_cast<byolo_t*>(h), BYOLO_ERR_ARG, "byolo_get_plan_opts: null argument");
    *out = h->opts;
    out->struct_bytes = (int32_t)sizeof *out;
    return BYOLO_OK;
}

extern "C" int32_t byolo_set_plan_opts(byolo_t* h, const byolo_plan_opts* o) {
    if (!h || !o) return fail(h, BYOLO_ERR_ARG, "byolo_set_plan_opts: null argument");
    if (o->struct_bytes != (int32_t)sizeof *o) return fail(h, BYOLO_ERR_ARG, "byolo_set_plan_opts: struct_bytes %d, this library's byolo_plan_opts has %d (include/byolo.h)", o->struct_bytes, (int)sizeof *o);
    if (o->graphs < 0 || o->graphs > 2 || o->serialize_convs < 0 || o->serialize_convs > 2 || o->b2b < 0 || o->b2b > 2 || o->kx3_wide < 0 || o->kx3_wide > 2 ||
        o->wino_split < 0 || o->wino_split > 2 || o->winograd < 0 || o->winograd > 2 || o->wino_fused < 0 || o->wino_fused > 2 || o->stream1x1 < 0 || o->stream1x1 > 2 ||
        o->streamk < 0 || o->streamk > 2 || o->ksplit < -1 || o->ksplit > 64 || (o->wino_split_bn != 0 && o->wino_split_bn != 128 && o->wino_split_bn != 256) ||
        o->wino_split_rounds < 0 || o->wino_split_persist < 0 || o->wino_split_persist > 2 || !(o->wino_split_chunk_mb > 0.f) || !(o->wino_chunk_mb > 0.f) || !(o->wino_split_min_gflop >= 0.f) || !(o->wino_min_gflop >= 0.f) || !(o->wino_min_ratio >= 0.f))
        return fail(h, BYOLO_ERR_ARG, "byolo_set_plan_opts: a field outside its range (include/byolo.h)");
    const byolo_plan_opts& c = h->opts;
    // what byolo_lower / byolo_finalize have baked into steps and packed weights
    const bool repack = (o->wshift_per_layer != 0) != (c.wshift_per_layer != 0) || (o->dedup != 0) != (c.dedup != 0) || (o->lowmain != 0) != (c.lowmain != 0) || (o->kx3 != 0) != (c.kx3 != 0) || (o->p1 != 0) != (c.p1 != 0);
    h->opts = *o;
    h->dedup = o->dedup != 0;
    h->plan.B = -1; h->plan.T = -1; ++h->plan_epoch; h->wsm_B = -1;
    if (h->d_blob || !h->graphs.empty()) { HIPCHK(h, hipSetDevice(h->device)); drop_graphs(h); }
    if (repack && h->lowered) { h->lowered = false; h->finalized = false; }
    return BYOLO_OK;
}

extern "C" int32_t byolo_graph_stats(const byolo_t* h, int32_t* n_graphs, int64_t* replays, int64_t* captures, int64_t* updates) {
    if (!h) return BYOLO_ERR_ARG;
    int n = 0; for (const auto& g : h->graphs) n += g.exec != nullptr;
    if (n_graphs) *n_graphs = n;
    if (replays) *replays = h->graph_replays;
    if (captures) *captures = h->graph_captures;
    if (updates) *updates = h->graph_updates;
    return BYOLO_OK;
}

extern "C" int32_t byolo_set_precision(byolo_t* h, int32_t precision) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_set_precision: null handle");
    if (precision != BYOLO_PREC_F32 && precision != BYOLO_PREC_SPLIT_F16) return fail(h, BYOLO_ERR_ARG, "byolo_set_precision: unknown precision %d", precision);
    h->prec_requested = precision;
    if (precision == BYOLO_PREC_F32) h->prec_note.clear();          // asked for, not fallen back to
    if (precision != h->precision) { h->precision = precision; h->prec_note.clear(); h->finalized = false; h->plan.B = -1; h->plan.T = -1; ++h->plan_epoch; }
    return BYOLO_OK;
}
extern "C" int32_t byolo_get_precision(const byolo_t* h) { return h ? h->precision : BYOLO_ERR_ARG; }

extern "C" int32_t byolo_destroy(byolo_t* h) {
    if (!h) return BYOLO_OK;
    bool on_device = h->d_blob || h->d_ones || h->d_status || !h->graphs.empty() || h->cap_stream;
    for (auto& ps : h->prof) on_device = on_device || ps.ev[0] || !ps.step_ev.empty();
    if (on_device) {                               // a handle that never ran (builder-only use, no GPU) touches no HIP call
        (void)hipSetDevice(h->device);
        drop_graphs(h);
        if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
        if (h->d_blob) (void)hipFree(h->d_blob);
        if (h->d_ones) (void)hipFree(h->d_ones);
        if (h->d_zeros) (void)hipFree(h->d_zeros);
        if (h->ev_convs) (void)hipEventDestroy(h->ev_convs);
        if (h->d_status) (void)hipFree(h->d_status);
        if (h->h_status) (void)hipHostFree(h->h_status);
        for (auto& ps : h->prof) {
            for (auto& e : ps.ev) if (e) (void)hipEventDestroy(e);
            for (auto& e : ps.step_ev) (void)hipEventDestroy(e);
        }
    }
    delete h;
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// graph construction
// ------------------------------------------------------------------------------------------------
static int add_param(byolo_t* h, const std::string& name, std::vector<int64_t> shape, float fill) {
    if (h->pindex.count(name)) return -1;
    Param p; p.name = name; p.shape = std::move(shape);
    p.data.assign((size_t)p.count(), fill);
    h->params.push_back(std::move(p));
    h->pindex[name] = (int)h->params.size() - 1;
    return (int)h->params.size() - 1;
}

static int resolve_ref(const byolo_t* h, int r) {        // reference indexing: negative = from the end
    const int n = (int)h->layers.size();
    const int a = r < 0 ? n + r : r;
    return (a < 0 || a >= n) ? -2 : a;
}

static void input_shape(const byolo_t* h, int prev, int& C, int& H, int& W, bool& stacked) {
    if (prev < 0) { C = h->cfg.img_c; H = h->cfg.img_h; W = h->cfg.img_w; stacked = false; }
    else { const Layer& l = h->layers[prev]; C = l.C; H = l.H; W = l.W; stacked = l.stacked; }
}

static int32_t begin_add(byolo_t* h, const char* what) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "%s: null handle", what);
    if (h->lowered) return fail(h, BYOLO_ERR_STATE, "%s: graph is frozen after byolo_finalize", what);
    return BYOLO_OK;
}

// Bounds of a convolution this library will hold on the host (the reference's largest: 1024 channels, 4.7 M weights): beyond them a
// call is a mistake (filters = INT_MAX would otherwise be a 200 GB std::vector filled with zeros)
static constexpr int MAX_CHANNELS = 1 << 16;
static constexpr int64_t MAX_KERNEL_WEIGHTS = (int64_t)1 << 28;

static int32_t add_conv_impl(byolo_t* h, const char* scope, int32_t filters, int32_t ksize, int32_t stride, int32_t norm_flags) {
    int32_t rc = begin_add(h, "byolo_add_conv"); if (rc) return rc;
    // lib_yolo/layers.py:546-547: assert kernel_size in [1,3], strides in [1,2]
    if (!scope || !(ksize == 1 || ksize == 3)) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: invalid kernel size");
    if (!(stride == 1 || stride == 2)) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: invalid strides");
    if (stride == 2 && ksize != 3) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: invalid kernel size (layers.py:629)");
    if (filters < 1) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: filters < 1");
    if (filters > MAX_CHANNELS) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: more than %d filters", MAX_CHANNELS);
    if (!(norm_flags & BYOLO_NORM_BN)) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: BN-less conv layers are not on this path");
    Layer l; l.op = OP_CONV; l.scope = scope; l.filters = filters; l.ksize = ksize; l.stride = stride; l.norm = norm_flags;
    l.prev = (int)h->layers.size() - 1;
    int C, H, W; bool st; input_shape(h, l.prev, C, H, W, st);
    if (stride == 2 && ((H | W) & 1)) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: stride 2 needs even input size");
    if ((int64_t)ksize * ksize * C * filters > MAX_KERNEL_WEIGHTS)
        return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: a kernel of %lld weights (limit 2^28)", (long long)ksize * ksize * C * filters);
    l.Cin = C; l.C = filters; l.H = H / stride; l.W = W / stride; l.stacked = st;
    const std::string s(scope);
    l.p_kernel = add_param(h, s + "/conv2d/kernel", {ksize, ksize, C, filters}, 0.f);
    l.p_gamma = add_param(h, s + "/batch_normalization/gamma", {filters}, 1.f);
    l.p_beta = add_param(h, s + "/batch_normalization/beta", {filters}, 0.f);
    l.p_mean = add_param(h, s + "/batch_normalization/moving_mean", {filters}, 0.f);
    l.p_var = add_param(h, s + "/batch_normalization/moving_variance", {filters}, 1.f);
    if (l.p_kernel < 0 || l.p_gamma < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_conv: duplicate scope '%s'", scope);
    if (norm_flags & BYOLO_NORM_DROPOUT) l.drop_ordinal = h->n_dropout++;
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}
extern "C" int32_t byolo_add_conv(byolo_t* h, const char* scope, int32_t filters, int32_t ksize, int32_t stride,
                                  int32_t norm_flags) {
    return guarded(h, "byolo_add_conv", [&] { return add_conv_impl(h, scope, filters, ksize, stride, norm_flags); });
}

extern "C" int32_t byolo_add_residual(byolo_t* h, int32_t shortcut) {
    int32_t rc = begin_add(h, "byolo_add_residual"); if (rc) return rc;
    Layer l; l.op = OP_RESIDUAL; l.prev = (int)h->layers.size() - 1;
    const int a = resolve_ref(h, shortcut);
    if (a < 0 || l.prev < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_residual: bad shortcut %d", shortcut);
    const Layer& x = h->layers[l.prev]; const Layer& s = h->layers[a];
    if (x.C != s.C || x.H != s.H || x.W != s.W || x.stacked != s.stacked)
        return fail(h, BYOLO_ERR_ARG, "byolo_add_residual: shape mismatch");
    l.ref[0] = a; l.nref = 1; l.C = x.C; l.H = x.H; l.W = x.W; l.stacked = x.stacked;
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}

extern "C" int32_t byolo_add_route(byolo_t* h, const int32_t* routes, int32_t n) {
    int32_t rc = begin_add(h, "byolo_add_route"); if (rc) return rc;
    // lib_yolo/layers.py:584-585
    if (!routes || n >= 3) return fail(h, BYOLO_ERR_ARG, "byolo_add_route: too many routes");
    if (n < 1) return fail(h, BYOLO_ERR_ARG, "byolo_add_route: too few routes");
    Layer l; l.op = OP_ROUTE; l.prev = (int)h->layers.size() - 1; l.nref = n;
    for (int i = 0; i < n; ++i) {
        l.ref[i] = resolve_ref(h, routes[i]);
        if (l.ref[i] < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_route: bad route %d", routes[i]);
    }
    const Layer& a = h->layers[l.ref[0]];
    l.C = a.C; l.H = a.H; l.W = a.W; l.stacked = a.stacked;
    if (n == 2) {
        const Layer& b = h->layers[l.ref[1]];
        if (a.H != b.H || a.W != b.W || a.stacked != b.stacked)
            return fail(h, BYOLO_ERR_ARG, "byolo_add_route: concat shape mismatch");
        l.C = a.C + b.C;
        if (l.C > MAX_CHANNELS) return fail(h, BYOLO_ERR_ARG, "byolo_add_route: more than %d channels", MAX_CHANNELS);
    }
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}

extern "C" int32_t byolo_add_upsample(byolo_t* h) {
    int32_t rc = begin_add(h, "byolo_add_upsample"); if (rc) return rc;
    Layer l; l.op = OP_UPSAMPLE; l.prev = (int)h->layers.size() - 1;
    if (l.prev < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_upsample: no input");
    const Layer& x = h->layers[l.prev];
    l.C = x.C; l.H = 2 * x.H; l.W = 2 * x.W; l.stacked = x.stacked;
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}

extern "C" int32_t byolo_add_stack(byolo_t* h, int32_t src) {
    int32_t rc = begin_add(h, "byolo_add_stack"); if (rc) return rc;
    Layer l; l.op = OP_STACK; l.prev = (int)h->layers.size() - 1;
    const int a = resolve_ref(h, src);
    if (a < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_stack: bad source %d", src);
    const Layer& x = h->layers[a];
    if (x.stacked) return fail(h, BYOLO_ERR_ARG, "byolo_add_stack: source is already stacked");
    l.ref[0] = a; l.nref = 1; l.C = x.C; l.H = x.H; l.W = x.W; l.stacked = true;
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}

static void row_layout(int kind, int C, int& D, int& obj, int& cls) {
    // lib_yolo/yolov3.py:183-184 / :321-322 / :464-465
    if (kind == BYOLO_DET_STANDARD) { D = 5 + C; obj = 4; cls = 5; }
    else if (kind == BYOLO_DET_ALEATORIC) { D = 14 + C; obj = 9; cls = 11; }
    else { D = 21 + C; obj = 14; cls = 17; }
}

static int32_t add_detection_impl(byolo_t* h, const char* scope, int32_t kind, const float* priors_hw) {
    int32_t rc = begin_add(h, "byolo_add_detection"); if (rc) return rc;
    if (!scope || !priors_hw || kind < 0 || kind > 2) return fail(h, BYOLO_ERR_ARG, "byolo_add_detection: bad argument");
    Layer l; l.op = OP_DETECTION; l.scope = scope; l.prev = (int)h->layers.size() - 1; l.det_kind = kind;
    if (l.prev < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_detection: no input");
    int C, H, W; bool st; input_shape(h, l.prev, C, H, W, st);
    const int cc = h->cfg.cls_cnt;
    l.filters = kind == BYOLO_DET_STANDARD ? 3 * (5 + cc) : 3 * 2 * (5 + cc);     // layers.py:601 / :609
    l.ksize = 1; l.stride = 1; l.Cin = C; l.C = l.filters; l.H = H; l.W = W; l.stacked = st;
    memcpy(l.priors, priors_hw, sizeof l.priors);
    l.det_id = h->n_det++;
    int D, obj, cls; row_layout(kind, cc, D, obj, cls);
    if (h->row_len && h->row_len != D) return fail(h, BYOLO_ERR_ARG, "byolo_add_detection: mixed detection kinds");
    h->row_len = D; h->obj_idx = obj; h->cls_start = cls;
    l.box_base = h->n_boxes;
    h->n_boxes += (int64_t)3 * H * W;
    const std::string s(scope);
    l.p_kernel = add_param(h, s + "/conv2d/kernel", {1, 1, C, l.filters}, 0.f);
    l.p_bias = add_param(h, s + "/conv2d/bias", {l.filters}, 0.f);
    if (l.p_kernel < 0 || l.p_bias < 0) return fail(h, BYOLO_ERR_ARG, "byolo_add_detection: duplicate scope '%s'", scope);
    h->layers.push_back(l);
    return (int32_t)h->layers.size() - 1;
}
extern "C" int32_t byolo_add_detection(byolo_t* h, const char* scope, int32_t kind, const float* priors_hw) {
    return guarded(h, "byolo_add_detection", [&] { return add_detection_impl(h, scope, kind, priors_hw); });
}

extern "C" int32_t byolo_mark_backbone_end(byolo_t* h) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "null handle");
    h->backbone_end = (int)h->layers.size();
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// parameters
// ------------------------------------------------------------------------------------------------
extern "C" int32_t byolo_num_params(const byolo_t* h) { return h ? (int32_t)h->params.size() : BYOLO_ERR_ARG; }

extern "C" int32_t byolo_param_info(const byolo_t* h, int32_t i, const char** name, int32_t* ndim, int64_t shape[4]) {
    if (!h || i < 0 || i >= (int)h->params.size()) return fail(const_cast<byolo_t*>(h), BYOLO_ERR_ARG, "byolo_param_info: bad index");
    const Param& p = h->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int32_t)p.shape.size();
    if (shape) for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
    return BYOLO_OK;
}

extern "C" int32_t byolo_set_param(byolo_t* h, const char* name, const float* data, int64_t count) {
    if (!h || !name || !data) return fail(h, BYOLO_ERR_ARG, "byolo_set_param: null argument");
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) return fail(h, BYOLO_ERR_ARG, "byolo_set_param: unknown variable '%s'", name);
    Param& p = h->params[it->second];
    if (count != p.count()) return fail(h, BYOLO_ERR_ARG, "byolo_set_param: '%s' expects %lld values, got %lld", name,
                                        (long long)p.count(), (long long)count);
    memcpy(p.data.data(), data, sizeof(float) * (size_t)count);
    return BYOLO_OK;
}

extern "C" int32_t byolo_get_param(const byolo_t* h, const char* name, float* data, int64_t count) {
    byolo_t* hh = const_cast<byolo_t*>(h);
    if (!h || !name || !data) return fail(hh, BYOLO_ERR_ARG, "byolo_get_param: null argument");
    auto it = h->pindex.find(name);
    if (it == h->pindex.end()) return fail(hh, BYOLO_ERR_ARG, "byolo_get_param: unknown variable '%s'", name);
    const Param& p = h->params[it->second];
    if (count != p.count()) return fail(hh, BYOLO_ERR_ARG, "byolo_get_param: size mismatch for '%s'", name);
    memcpy(data, p.data.data(), sizeof(float) * (size_t)count);
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// lowering
// ------------------------------------------------------------------------------------------------
// `self`: resolve the sources of view `idx` itself (for its STEP_GATHER), not the tensor it was materialised into
static bool resolve_view(const byolo_t* h, int idx, View& v, std::string& why, bool self = false) {
    if (idx < 0) { v.n = 1; v.s[0] = {-1, h->cfg.img_c, 0, false}; return true; }
    const Layer& l = h->layers[idx];
    if (!self && idx < (int)h->need_mat.size() && h->need_mat[idx]) {       // a view with a tensor of its own
        v.n = 1; v.s[0] = {idx, l.C, 0, false}; return true;
    }
    switch (l.op) {
        case OP_CONV: case OP_DETECTION: case OP_RESIDUAL:
            if (!l.materialized) { why = "layer " + std::to_string(idx) + " is fused away but referenced"; return false; }
            v.n = 1; v.s[0] = {idx, l.C, 0, false}; return true;
        case OP_ROUTE: {
            if (l.nref == 1) return resolve_view(h, l.ref[0], v, why);
            View a, b;
            if (!resolve_view(h, l.ref[0], a, why) || !resolve_view(h, l.ref[1], b, why)) return false;
            if (a.n != 1 || b.n != 1) { why = "nested concat"; h->want_mat = a.n != 1 ? l.ref[0] : l.ref[1]; return false; }
            v.n = 2; v.s[0] = a.s[0]; v.s[1] = b.s[0]; return true;
        }
        case OP_UPSAMPLE:
            if (!resolve_view(h, l.prev, v, why)) return false;
            for (int i = 0; i < v.n; ++i) { if (v.s[i].sh) { why = "double upsample"; h->want_mat = l.prev; return false; } v.s[i].sh = 1; }
            return true;
        case OP_STACK:
            if (!resolve_view(h, l.ref[0], v, why)) return false;
            for (int i = 0; i < v.n; ++i) v.s[i].tile = true;
            return true;
    }
    return false;
}

static bool is_view_op(Op op) { return op == OP_ROUTE || op == OP_UPSAMPLE || op == OP_STACK; }

static int32_t lower_once(byolo_t* h) {
    const int n = (int)h->layers.size();
    // reference counts
    std::vector<int> refs(n, 0);
    for (int i = 0; i < n; ++i) {
        const Layer& l = h->layers[i];
        if ((l.op == OP_CONV || l.op == OP_DETECTION || l.op == OP_RESIDUAL || l.op == OP_UPSAMPLE) && l.prev >= 0) refs[l.prev]++;
        for (int k = 0; k < l.nref; ++k) refs[l.ref[k]]++;
    }
    for (auto& l : h->layers) { l.materialized = false; l.out_tensor = -1; l.fused_residual = -1; l.standalone = false; l.add_a = -1; }
    for (int i = 0; i < n; ++i) {
        Layer& l = h->layers[i];
        if (l.op == OP_CONV || l.op == OP_DETECTION || (is_view_op(l.op) && h->need_mat[i])) { l.materialized = true; l.out_tensor = i; }
    }
    // an operand of a residual add is read as a dense [rows][C] tensor: the output of a convolution, of an earlier
    // residual, or a view that has been given a tensor; identity routes in between are looked through
    auto as_tensor = [&](int a) -> int {
        while (h->layers[a].op == OP_ROUTE && h->layers[a].nref == 1 && !h->need_mat[a]) a = h->layers[a].ref[0];
        if (h->layers[a].materialized && h->layers[a].out_tensor == a) return a;
        if (is_view_op(h->layers[a].op)) h->want_mat = a;       // lower() gives it a tensor and tries again
        return -1;
    };
    for (int i = 0; i < n; ++i) {
        Layer& l = h->layers[i];
        if (l.op != OP_RESIDUAL) continue;
        Layer& c = h->layers[l.prev];
        const int a = as_tensor(l.ref[0]);
        if (a < 0) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: residual layer %d: the shortcut is not a tensor", i);
        l.ref[0] = a;
        if (c.op == OP_CONV && refs[l.prev] == 1 && c.fused_residual < 0 && a != l.prev) {
            c.materialized = false; c.fused_residual = i; c.out_tensor = i; l.materialized = true; l.out_tensor = i;
        } else {                                            // its own element-wise step
            const int left = as_tensor(l.prev);
            if (left < 0) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: residual layer %d: the left operand is not a tensor", i);
            l.standalone = true; l.add_a = left; l.materialized = true; l.out_tensor = i;
        }
    }
    h->steps.clear();
    h->aux.clear();
    h->last_use.assign(n, -1);
    h->dedup = h->opts.dedup != 0;
    for (int i = 0; i < n; ++i) {
        Layer& l = h->layers[i];
        if (is_view_op(l.op) && h->need_mat[i]) {           // copy the view into its tensor
            Step g; g.layer = i; g.mode = STEP_GATHER; g.out_tensor = i;
            std::string why;
            if (!resolve_view(h, i, g.in, why, true)) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: layer %d: %s", i, why.c_str());
            const int gi = (int)h->steps.size();
            for (int k = 0; k < g.in.n; ++k) {
                const Src& s = g.in.s[k];
                if (s.layer >= 0) h->last_use[s.layer] = gi;
                if (l.stacked && !(s.layer >= 0 && h->layers[s.layer].stacked) && !s.tile)
                    return fail(h, BYOLO_ERR_ARG, "byolo_finalize: layer %d mixes stacked and unstacked inputs", i);
            }
            h->steps.push_back(g);
            continue;
        }
        if (l.op == OP_RESIDUAL && l.standalone) {
            Step a; a.layer = i; a.mode = STEP_ADD; a.out_tensor = i;
            a.in.n = 2; a.in.s[0] = {l.add_a, l.C, 0, false}; a.in.s[1] = {l.ref[0], l.C, 0, false};
            const int ai = (int)h->steps.size();
            h->last_use[l.add_a] = ai; h->last_use[l.ref[0]] = ai;
            h->steps.push_back(a);
            continue;
        }
        if (l.op != OP_CONV && l.op != OP_DETECTION) continue;
        Step st; st.layer = i; st.out_tensor = l.out_tensor; st.c_lo = 0; st.c_hi = l.Cin;
        std::string why;
        if (!resolve_view(h, l.prev, st.in, why)) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: layer %d: %s", i, why.c_str());
        int ctot = 0;
        for (int k = 0; k < st.in.n; ++k) {
            ctot += st.in.s[k].C;
            const Src& s = st.in.s[k];
            const bool src_stacked = s.layer >= 0 && h->layers[s.layer].stacked;
            if (l.stacked && !src_stacked && !s.tile) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: layer %d mixes stacked and unstacked inputs", i);
        }
        if (ctot != l.Cin) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: layer %d channel mismatch", i);
        // the implicit-GEMM loader walks the input channels in tiles of 32 (per source); everything else takes the
        // general direct kernel (conv_kernels.hip): slow, but the builder accepts what the reference's does
        l.direct = (l.Cin % 32) != 0 || (st.in.n == 2 && (st.in.s[0].C % 32) != 0);
        // ---- T-invariant de-duplication ----------------------------------------------------------
        if (h->dedup && l.op == OP_CONV && l.stacked && !l.direct) {
            bool all_tile = true, any_tile = false;
            for (int k = 0; k < st.in.n; ++k) { all_tile &= st.in.s[k].tile; any_tile |= st.in.s[k].tile; }
            if (all_tile && l.drop_ordinal >= 0 && l.fused_residual < 0) {
                st.mode = STEP_REP;                          // conv once per image, T masked epilogues
                for (int k = 0; k < st.in.n; ++k) st.in.s[k].tile = false;
            } else if (st.in.n == 2 && any_tile && !all_tile) {
                const int kt = st.in.s[0].tile ? 0 : 1;      // the tiled (T-invariant) source
                Step part; part.layer = i; part.mode = STEP_PARTIAL;
                part.in.n = 1; part.in.s[0] = st.in.s[kt]; part.in.s[0].tile = false;
                part.c_lo = kt == 0 ? 0 : st.in.s[0].C; part.c_hi = part.c_lo + st.in.s[kt].C;
                h->aux.push_back({l.H, l.W, l.filters});
                part.out_tensor = n + (int)h->aux.size() - 1;
                h->last_use.push_back(-1);
                const int pidx = (int)h->steps.size();
                if (part.in.s[0].layer >= 0) h->last_use[part.in.s[0].layer] = pidx;
                h->steps.push_back(part);
                Step main = st; main.mode = STEP_MAIN; main.addend_tensor = part.out_tensor;
                main.in.n = 1; main.in.s[0] = st.in.s[1 - kt];
                main.c_lo = kt == 0 ? st.in.s[0].C : 0; main.c_hi = main.c_lo + st.in.s[1 - kt].C;
                st = main;
                // the stacked half is an upsampled tensor and the convolution is 1x1: multiply at the source's resolution
                // (STEP_PARTIAL with `low`), finish element-wise at the output's (STEP_FINISH)
                if (h->opts.lowmain != 0 && l.ksize == 1 && l.stride == 1 && st.in.s[0].sh == 1 && st.in.s[0].layer >= 0 &&
                    !st.in.s[0].tile && (l.H & 1) == 0 && (l.W & 1) == 0 && (l.filters & 3) == 0 && l.fused_residual < 0 && (st.in.s[0].C % 32) == 0) {
                    Step lowst = st; lowst.mode = STEP_PARTIAL; lowst.low = true; lowst.addend_tensor = -1;
                    lowst.in.s[0].sh = 0;
                    h->aux.push_back({l.H / 2, l.W / 2, l.filters, true});
                    lowst.out_tensor = n + (int)h->aux.size() - 1;
                    h->last_use.push_back(-1);
                    const int lidx = (int)h->steps.size();
                    h->last_use[lowst.in.s[0].layer] = lidx;
                    h->steps.push_back(lowst);
                    Step fin; fin.layer = i; fin.mode = STEP_FINISH; fin.out_tensor = l.out_tensor;
                    fin.low_tensor = lowst.out_tensor; fin.addend_tensor = part.out_tensor;
                    fin.c_lo = st.c_lo; fin.c_hi = st.c_hi;
                    const int fidx = (int)h->steps.size();
                    h->last_use[fin.low_tensor] = fidx; h->last_use[fin.addend_tensor] = fidx;
                    h->steps.push_back(fin);
                    continue;
                }
            }
        }
        const int step_idx = (int)h->steps.size();
        for (int k = 0; k < st.in.n; ++k) if (st.in.s[k].layer >= 0) h->last_use[st.in.s[k].layer] = step_idx;
        if (st.addend_tensor >= 0) h->last_use[st.addend_tensor] = step_idx;
        if (l.fused_residual >= 0) h->last_use[h->layers[l.fused_residual].ref[0]] = step_idx;
        h->steps.push_back(st);
    }
    // launch geometry that depends on the graph only (byolo_workspace_bytes may plan before byolo_finalize)
    for (auto& st : h->steps) {
        if (!st.is_conv()) continue;
        const Layer& l = h->layers[st.layer];
        const int N = l.filters;
        if (l.direct) { st.tile = -1; st.Npad = N; }
        else { st.tile = conv_pick_tile(N); const int bn = conv_tile_bn(st.tile); st.Npad = (N + bn - 1) / bn * bn; }
        st.wino_ok = !l.direct && l.op == OP_CONV && l.ksize == 3 && l.stride == 1 && st.mode == STEP_NORMAL &&
                     st.in.n == 1 && st.in.s[0].sh == 0 && !st.in.s[0].tile && (l.Cin % 32) == 0 && (N % 4) == 0;
    }
    h->maxC = 1;
    for (const auto& l : h->layers) h->maxC = std::max(h->maxC, l.C);
    decide_loops(h);
    h->lowered = true;
    return BYOLO_OK;
}

// Lower the graph; where a view cannot be expressed inside a loader, give that view a tensor of its own and try again
// (at most once per layer).
int32_t byolo_lower(byolo_t* h) {
    const int n = (int)h->layers.size();
    if (!n) return fail(h, BYOLO_ERR_STATE, "byolo_finalize: empty graph");
    if (!h->n_det) return fail(h, BYOLO_ERR_STATE, "byolo_finalize: no detection layer (model.py:190: assert len(det_layers) > 0)");
    h->need_mat.assign(n, 0);
    for (int attempt = 0; ; ++attempt) {
        h->want_mat = -1;
        const int32_t rc = lower_once(h);
        if (rc == BYOLO_OK) return rc;
        if (h->want_mat < 0 || h->need_mat[h->want_mat] || attempt > n) return rc;
        h->need_mat[h->want_mat] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Position, in samples, of a call's first sample in the dropout stream of a tensor with T samples per image in this call:
// first_image * T -- or, with the T samples of an image sharded over ranks (byolo_set_tshard: one image per call), sample t0 of the
// tshard_T the image has in the whole job.
static uint64_t sample_base(const byolo_t* h, int T, bool stacked) {
    if (h->tshard_T > 0 && stacked) return (uint64_t)h->first_image * (uint64_t)h->tshard_T + (uint64_t)h->tshard_t0;
    return (uint64_t)h->first_image * (uint64_t)T;
}

static void fill_conv(const byolo_t* h, const Step& st, const float* d_img, char* ws, int B, int T, ConvParams& p) {
    const Layer& l = h->layers[st.layer];
    memset(&p, 0, sizeof p);
    const float* srcs[2] = {nullptr, nullptr};
    int Cs[2] = {0, 0}, Hs[2] = {1, 1}, Wsz[2] = {1, 1}, sh[2] = {0, 0}, sdiv[2] = {1, 1};
    int64_t nsrc[2] = {0, 0};                 // samples held by each source tensor
    for (int k = 0; k < st.in.n; ++k) {
        const Src& s = st.in.s[k];
        if (s.layer < 0) { srcs[k] = (h->precision == 1 && !l.direct) ? reinterpret_cast<const float*>(ws + h->plan.img_split_off) : d_img; Hs[k] = h->cfg.img_h; Wsz[k] = h->cfg.img_w; nsrc[k] = B; }
        else {
            srcs[k] = reinterpret_cast<const float*>(ws + h->plan.off[s.layer]); Hs[k] = h->layers[s.layer].H; Wsz[k] = h->layers[s.layer].W;
            nsrc[k] = h->layers[s.layer].stacked ? (int64_t)B * T : B;
        }
        Cs[k] = s.C; sh[k] = s.sh; sdiv[k] = s.tile ? T : 1;
    }
    p.src0 = srcs[0]; p.src1 = srcs[1] ? srcs[1] : srcs[0];
    p.C0 = Cs[0]; p.C1 = Cs[1];
    p.Hs0 = Hs[0]; p.Ws0 = Wsz[0]; p.Hs1 = Hs[1]; p.Ws1 = Wsz[1];
    p.sh0 = sh[0]; p.sh1 = sh[1]; p.sdiv0 = sdiv[0]; p.sdiv1 = sdiv[1];
    p.Hin = Hs[0] << sh[0]; p.Win = Wsz[0] << sh[0];
    const int lsh = st.low ? 1 : 0;                               // the `low` launch: output at the source's resolution
    p.Hout = l.H >> lsh; p.Wout = l.W >> lsh;
    p.ksize = l.ksize; p.stride = l.stride; p.pad = l.ksize == 3 ? 1 : 0;
    // rows of this launch: MC samples for stacked layers, IMAGES for the de-duplicated launches
    const bool per_image = st.mode == STEP_REP || (st.mode == STEP_PARTIAL && !st.low);
    const int64_t S = (l.stacked && !per_image) ? (int64_t)B * T : B;
    p.M = (int)(S * p.Hout * p.Wout);
    p.N = layer_pitch(l); p.Npad = st.Npad; p.ldc = layer_pitch(l);      // (a detection head: padded to a multiple of 4, <= Npad)
    p.cin_tiles = (st.c_hi - st.c_lo) / 32; p.KT = l.ksize * l.ksize * p.cin_tiles;
    p.wpk = dptr(h, st.w_off);
    if (st.mode == STEP_PARTIAL) { p.scale = h->d_ones; p.shift = h->d_zeros; }      // raw accumulators
    else { p.scale = dptr(h, l.scale_off); p.shift = dptr(h, l.shift_off); }
    p.dst = reinterpret_cast<float*>(ws + h->plan.off[st.out_tensor]);
    // buffer-descriptor extents (check_run bounds every tensor by CONV_MAX_SRC_BYTES) and the launch-constant divisors
    p.src0_bytes = (uint32_t)((uint64_t)nsrc[0] * Hs[0] * Wsz[0] * Cs[0] * 4);
    p.src1_bytes = Cs[1] ? (uint32_t)((uint64_t)nsrc[1] * Hs[1] * Wsz[1] * Cs[1] * 4) : p.src0_bytes;
    p.w_bytes = (uint32_t)((uint64_t)p.KT * p.Npad * 32 * 4);
    p.d_hw = make_fastdiv((uint32_t)(p.Hout * p.Wout)); p.d_wout = make_fastdiv((uint32_t)p.Wout);
    p.d_sdiv0 = make_fastdiv((uint32_t)sdiv[0]); p.d_sdiv1 = make_fastdiv((uint32_t)sdiv[1]);
    p.rep = st.mode == STEP_REP ? T : 1;
    // matrix-pipe launches: 1 = split-f16 operands; direct launches: bit 0 = the sources are hi/lo tensors, bit 1 = so is the output
    p.split = h->precision != 1 ? 0 : (!l.direct ? 1 : ((l.prev >= 0 ? 1 : 0) | (l.op == OP_DETECTION ? 0 : 2)));
    if (p.split && st.kx3) { p.kx3 = 1; p.KT = 3 * p.cin_tiles; }       // scheduling unit = stage (ky, chunk) = 3 K-tiles
    else if (p.split && st.p1) p.kx3 = 2;
    if (st.mode == STEP_MAIN) {
        p.addend = reinterpret_cast<const float*>(ws + h->plan.off[st.addend_tensor]);
        p.addend_T = l.stacked ? T : 1;
    } else { p.addend = nullptr; p.addend_T = 1; }
    p.d_addT = make_fastdiv((uint32_t)p.addend_T);
    p.status = h->precision == 1 ? h->d_status : nullptr; p.layer_idx = st.layer;
    p.no_plain = h->opts.plain_epilogue == 0;
}

// STEP_FINISH: mode 0 the raw sum (calibration), 1 / 2 the layer's epilogue with fp32 / hi-lo output
static void fill_finish(const byolo_t* h, const Step& st, char* ws, int B, int T, int mode, bool drop, const byolo_drop_keys& keys,
                        const uint32_t* mask_bits, bool inject, FinishParams& f) {
    const Layer& l = h->layers[st.layer];
    memset(&f, 0, sizeof f);
    f.low = reinterpret_cast<const float*>(ws + h->plan.off[st.low_tensor]);
    f.part = st.addend_tensor >= 0 ? reinterpret_cast<const float*>(ws + h->plan.off[st.addend_tensor]) : nullptr;
    f.dst = reinterpret_cast<float*>(ws + h->plan.off[st.out_tensor]);
    f.T = l.stacked ? T : 1; f.S = B * f.T; f.H = l.H; f.W = l.W; f.N = l.filters;
    f.mode = mode;
    if (mode == 0) { f.scale = h->d_ones; f.shift = h->d_zeros; }
    else {
        f.flags = EPI_LEAKY;
        f.scale = dptr(h, l.scale_off); f.shift = dptr(h, l.shift_off);
        if (drop) {
            f.flags |= EPI_DROPOUT; f.k0 = keys.k0; f.k1 = keys.k1; f.thr = keys.thr; f.mask_bits = mask_bits;
            f.idx_base = inject ? 0 : sample_base(h, f.T, l.stacked) * (uint64_t)l.H * l.W * l.filters;
            f.scale = dptr(h, l.scalek_off);
        }
    }
    f.status = h->precision == 1 ? h->d_status : nullptr; f.layer_idx = st.layer;
    f.d_hw = make_fastdiv((uint32_t)(l.H * l.W / 4)); f.d_w = make_fastdiv((uint32_t)(l.W / 2));      // of the source's grid
    f.d_n4 = make_fastdiv((uint32_t)(l.filters / 4)); f.d_T = make_fastdiv((uint32_t)f.T);
}

// STEP_GATHER / STEP_ADD (fill_conv has resolved the sources, the output extent and the destination)
static int32_t run_aux_step(byolo_t* h, const Step& s, const ConvParams& p, hipStream_t st) {
    const Layer& l = h->layers[s.layer];
    if (s.mode == STEP_GATHER) { HIPCHK(h, launch_view_gather(p, st)); }
    else { HIPCHK(h, launch_tensor_add(p.src0, p.src1, p.dst, (int64_t)p.M * l.C, h->precision == 1, st, p.status, s.layer)); }
    return BYOLO_OK;
}

// profiling level 2: event + bookkeeping entry before a launch of the convolution stack
static int32_t next_event(byolo_t* h, hipStream_t st, int* idx) {
    byolo::ProfSlot& ps = h->wslot();
    while ((int)ps.step_ev.size() <= ps.n_ev) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); ps.step_ev.push_back(e); }
    HIPCHK(h, hipEventRecord(ps.step_ev[ps.n_ev], st));
    *idx = ps.n_ev++;
    return BYOLO_OK;
}

// A launch on the caller's stream: its begin event is also the end event of the launch before it.
static int32_t mark_launch(byolo_t* h, int layer, int variant, int64_t m, int64_t n, int64_t k, double algo, hipStream_t st,
                           int ksplit = 1, int split_tiles = 0) {
    byolo::ProfSlot& ps = h->wslot();
    int e; int32_t rc = next_event(h, st, &e); if (rc) return rc;
    if (ps.last_main >= 0) ps.launches[ps.last_main].ev_end = e;
    ps.last_main = (int)ps.launches.size();
    ps.launches.push_back({layer, variant, m, n, k, algo, ksplit, split_tiles, e, -1});
    return BYOLO_OK;
}

// One 3x3 / stride-1 convolution as Winograd F(2x2,3x3): per chunk of samples, input transform -> ONE batched GEMM
// launch of the implicit-GEMM kernel (16 row blocks, one weight matrix each, raw accumulators out) -> output
// transform with the convolution's own epilogue.  `c` is the ConvParams of the direct launch (sources, dst, epilogue).
static int32_t run_winograd(byolo_t* h, const Step& s, const Layer& l, const ConvParams& c, const WinoPlan& wp, int tile,
                            double algo_flops, char* ws, hipStream_t st) {
    const bool prof = h->profiling >= 2;
    int32_t rc;
    const int S = c.M / (l.H * l.W), tt = wp.th * wp.tw;
    float* V = reinterpret_cast<float*>(ws + h->plan.wino_off);
    float* Mb = reinterpret_cast<float*>(ws + h->plan.wino_off + wp.v_bytes);
    for (int s0 = 0; s0 < S; s0 += wp.chunk) {
        const int ns = std::min(wp.chunk, S - s0);
        WinoParams w; memset(&w, 0, sizeof w);
        w.x = c.src0; w.v = V; w.m = Mb; w.y = c.dst;
        w.residual = (c.flags & EPI_RESIDUAL) ? c.residual : nullptr;
        w.scale = c.scale; w.shift = c.shift;
        w.H = l.H; w.W = l.W; w.C = c.C0; w.N = c.N; w.th = wp.th; w.tw = wp.tw;
        w.s0 = s0; w.P = ns * tt; w.P_pad = (int)align_up((size_t)w.P, 128);
        w.flags = c.flags; w.k0 = c.k0; w.k1 = c.k1; w.thr = c.thr; w.idx_base = c.idx_base;
        w.d_tt = make_fastdiv((uint32_t)tt); w.d_tw = make_fastdiv((uint32_t)wp.tw);
        w.d_c4 = make_fastdiv((uint32_t)(c.C0 / 4)); w.d_n4 = make_fastdiv((uint32_t)(c.N / 4));
        // variants of the profile entries: -2 input transform, BN of the GEMM tile, -3 output transform; the GEMM
        // entry carries the direct-convolution FLOPs its samples stand for, its m/n/k are the executed extents
        if (prof && (rc = mark_launch(h, s.layer, -2, w.P, c.C0, 0, 0.0, st))) return rc;
        HIPCHK(h, launch_wino_input(w, st));

        const int rows = 16 * w.P_pad;
        if (wp.fused) {                                         // GEMM + output transform + epilogue in one kernel: no M
            WinoFusedParams f; memset(&f, 0, sizeof f);
            f.v = V; f.v_bytes = (uint32_t)((uint64_t)rows * c.C0 * 4);
            f.w = dptr(h, s.wino_off); f.wstride = (uint32_t)((size_t)(c.C0 / 32) * c.N * 32 * 4); f.w_bytes = 16u * f.wstride;
            f.y = c.dst; f.residual = (c.flags & EPI_RESIDUAL) ? c.residual : nullptr; f.scale = c.scale; f.shift = c.shift;
            f.C = c.C0; f.N = c.N; f.KT = c.C0 / 32; f.n_tiles = c.N / 64;
            f.H = l.H; f.W = l.W; f.th = wp.th; f.tw = wp.tw; f.s0 = s0; f.P = w.P;
            const int RT = w.P_pad / 128;
            f.slots = 512 / f.n_tiles; f.q = RT / f.slots; f.rem = RT % f.slots;
            f.xi_stride = (uint32_t)((uint64_t)w.P_pad * c.C0 * 4);
            f.flags = c.flags; f.k0 = c.k0; f.k1 = c.k1; f.thr = c.thr; f.idx_base = c.idx_base;
            f.d_ntiles = make_fastdiv((uint32_t)f.n_tiles); f.d_tt = w.d_tt; f.d_tw = w.d_tw;
            if (prof && (rc = mark_launch(h, s.layer, 130, rows, c.N, c.C0, algo_flops * ns / S, st))) return rc;
            HIPCHK(h, launch_wino_fused(f, st));
            continue;
        }
        if (h->opts.gemm_stream != 0 && gemm_stream_ok(c.C0, c.N)) {           // persistent row-streaming GEMM (gemm_stream.hip)
            GemmStreamParams q; memset(&q, 0, sizeof q);
            q.a = V; q.a_bytes = (uint32_t)((uint64_t)rows * c.C0 * 4);
            q.w = dptr(h, s.wino_off); q.wstride = (uint32_t)((size_t)(c.C0 / 32) * c.N * 32 * 4); q.w_bytes = 16u * q.wstride;
            q.dst = Mb; q.C = c.C0; q.N = c.N; q.KT = c.C0 / 32; q.n_tiles = c.N / 128;
            q.RT = w.P_pad / 128;
            const int R = 16 * q.RT;
            q.slots = 512 / q.n_tiles; q.q = R / q.slots; q.rem = R % q.slots;
            q.d_ntiles = make_fastdiv((uint32_t)q.n_tiles); q.d_RT = make_fastdiv((uint32_t)q.RT);
            q.epi = 0; q.M = rows; q.Npad = c.N; q.ldc = c.N;
            if (prof && (rc = mark_launch(h, s.layer, 129, rows, c.N, c.C0, algo_flops * ns / S, st))) return rc;
            HIPCHK(h, launch_gemm_stream(q, st));
            if (prof && (rc = mark_launch(h, s.layer, -3, w.P, c.N, 0, 0.0, st))) return rc;
            HIPCHK(h, launch_wino_output(w, st));
            continue;
        }
        ConvParams g; memset(&g, 0, sizeof g);                  // a 1x1 convolution over a 1 x rows "image" of C0 channels
        g.src0 = V; g.src1 = V; g.C0 = c.C0; g.C1 = 0;
        g.src0_bytes = g.src1_bytes = (uint32_t)((uint64_t)rows * c.C0 * 4);
        g.Hs0 = g.Hs1 = 1; g.Ws0 = g.Ws1 = rows; g.sdiv0 = g.sdiv1 = 1;
        g.Hin = 1; g.Win = rows; g.Hout = 1; g.Wout = rows; g.ksize = 1; g.stride = 1; g.pad = 0;
        g.M = rows; g.N = c.N; g.Npad = c.Npad; g.ldc = c.N; g.cin_tiles = c.C0 / 32; g.KT = g.cin_tiles;
        g.wpk = dptr(h, s.wino_off);
        g.wino_rows = (uint32_t)w.P_pad; g.d_wino = make_fastdiv((uint32_t)w.P_pad);
        g.wino_wstride = (uint32_t)((size_t)g.cin_tiles * c.Npad * 32 * 4);
        g.w_bytes = 16u * g.wino_wstride;
        g.scale = h->d_ones; g.shift = h->d_zeros; g.flags = EPI_RAW; g.rep = 1; g.addend_T = 1;
        g.dst = Mb;
        g.d_hw = make_fastdiv((uint32_t)rows); g.d_wout = make_fastdiv((uint32_t)rows);
        g.d_sdiv0 = g.d_sdiv1 = g.d_addT = make_fastdiv(1u);
        const ConvSplit sp = conv_plan_split(rows, c.Npad, g.KT, tile, 1.0, h->opts.ksplit, h->opts.streamk);
        if (conv_split_slab_bytes(sp, tile) <= h->plan.slab_bytes) {
            g.full_tiles = sp.full_tiles; g.split_tiles = sp.split_tiles; g.split_blocks = sp.split_blocks; g.ksplit = sp.ksplit;
            g.sk_grid = sp.sk_grid;
            g.slabs = c.slabs; g.slab_bytes = (uint32_t)conv_split_slab_bytes(sp, tile); g.counters = c.counters;
        }
        if (prof && (rc = mark_launch(h, s.layer, conv_tile_bn(tile), rows, c.N, c.C0, algo_flops * ns / S, st, g.sk_grid > 0 ? -g.sk_grid : (g.ksplit > 1 ? g.ksplit : 1), g.split_tiles))) return rc;
        HIPCHK(h, launch_conv_igemm(g, tile, st));
        if (prof && (rc = mark_launch(h, s.layer, -3, w.P, c.N, 0, 0.0, st))) return rc;
        HIPCHK(h, launch_wino_output(w, st));
    }
    return BYOLO_OK;
}

// One 3x3 / stride-1 convolution as Winograd F(2x2,3x3) in split-f16 arithmetic (wino_split.hip): per chunk of samples the input
// transform and ONE launch of GEMM + output transform + epilogue.  `c` = the ConvParams of the direct launch.
// Profile variants: -4 the transform, 140 the fused launch (carries the direct-convolution FLOPs of its samples).
static int32_t run_wino_split(byolo_t* h, const Step& s, const Layer& l, const ConvParams& c, const WinoPlan& wp, double algo_flops,
                              char* ws, hipStream_t st) {
    const bool prof = h->profiling >= 2;
    int32_t rc;
    const int S = c.M / (l.H * l.W), tt = wp.th * wp.tw;
    float* V = reinterpret_cast<float*>(ws + h->plan.wino_off);
    const bool drop = c.flags & EPI_DROPOUT;
    for (int s0 = 0; s0 < S; s0 += wp.chunk) {
        const int ns = std::min(wp.chunk, S - s0);
        WinoParams w; memset(&w, 0, sizeof w);
        w.x = c.src0; w.v = V;
        w.H = l.H; w.W = l.W; w.C = c.C0; w.N = c.N; w.th = wp.th; w.tw = wp.tw;
        w.s0 = s0; w.P = ns * tt; w.P_pad = (int)align_up((size_t)w.P, 128);
        w.d_tt = make_fastdiv((uint32_t)tt); w.d_tw = make_fastdiv((uint32_t)wp.tw);
        w.d_c4 = make_fastdiv((uint32_t)(c.C0 / 4)); w.d_n4 = make_fastdiv((uint32_t)(c.N / 4));
        w.vmul = 1.f / ACT_SCALE;
        if (prof && (rc = mark_launch(h, s.layer, -4, w.P, c.C0, 0, 0.0, st))) return rc;
        HIPCHK(h, launch_wino_split_input(w, st));
        WinoSplitParams f; memset(&f, 0, sizeof f);
        const uint64_t rows = (uint64_t)16 * w.P_pad;
        f.v = V; f.v_bytes = (uint32_t)(rows * c.C0 * 4); f.xi_stride = (uint32_t)((uint64_t)w.P_pad * c.C0 * 4);
        f.w = dptr(h, s.wino_off); f.w_bytes = (uint32_t)((size_t)16 * c.C0 * c.N * 4);
        f.y = c.dst; f.residual = (c.flags & EPI_RESIDUAL) ? c.residual : nullptr; f.scale = dptr(h, drop ? l.wscalek_off : l.wscale_off); f.shift = c.shift;
        f.C = c.C0; f.N = c.N; f.KT = c.C0 / 32; f.n_tiles = c.N / wp.bn; f.bn = wp.bn;
        f.H = l.H; f.W = l.W; f.th = wp.th; f.tw = wp.tw; f.s0 = s0; f.P = w.P; f.P_pad = w.P_pad;
        f.bm = wp.bm; f.units = (w.P_pad / wp.bm) * f.n_tiles;
        // (persist 2 claims units from 8 words of this step's ticket area, zeroed by the forward's memset: one set per chunk)
        const int chunk_idx = s0 / wp.chunk;
        f.persist = (h->opts.wino_split_persist == 2 && c.counters && chunk_idx < CNT_PER_STEP / 8) ? 2 : (h->opts.wino_split_persist ? 1 : 0);
        f.claims = c.counters ? c.counters + 8 * chunk_idx : nullptr;
        f.flags = c.flags; f.k0 = c.k0; f.k1 = c.k1; f.thr = c.thr; f.idx_base = c.idx_base; f.mask_bits = c.mask_bits;
        f.status = c.status; f.layer_idx = c.layer_idx;
        f.d_ntiles = make_fastdiv((uint32_t)f.n_tiles); f.d_tt = w.d_tt; f.d_tw = w.d_tw;
        if (prof && (rc = mark_launch(h, s.layer, 140, (int64_t)rows, c.N, c.C0, algo_flops * ns / S, st, 1, wp.bn))) return rc;   // (split_tiles of a Winograd entry: its channels per workgroup)
        HIPCHK(h, launch_wino_split(f, st));
    }
    return BYOLO_OK;
}

static int32_t run_decode(byolo_t* h, char* ws, float* boxes, int B, int T, hipStream_t st, int mode = 0) {
    for (const auto& l : h->layers) {
        if (l.op != OP_DETECTION) continue;
        DecodeParams d; memset(&d, 0, sizeof d);
        d.mode = mode;
        if (mode && l.det_kind != BYOLO_DET_EPISTEMIC) return fail(h, BYOLO_ERR_ARG, "T sharding (byolo_set_tshard) is defined for epistemic detection layers");
        d.raw = mode == 2 ? nullptr : reinterpret_cast<const float*>(ws + h->plan.off[l.out_tensor]);
        d.boxes = boxes; d.lh = l.H; d.lw = l.W; d.C = h->cfg.cls_cnt;
        d.n_total = h->n_boxes; d.box_base = l.box_base; d.layer_id = l.det_id; d.ld = layer_pitch(l);
        d.status = h->precision == 1 ? h->d_status : nullptr;
        for (int k = 0; k < 3; ++k) { d.ph[k] = l.priors[2 * k]; d.pw[k] = l.priors[2 * k + 1]; }
        if (l.det_kind == BYOLO_DET_EPISTEMIC) { d.B = B; d.T = l.stacked ? T : 1; }
        else { d.B = l.stacked ? B * T : B; d.T = 1; }
        if (l.det_kind != BYOLO_DET_EPISTEMIC && l.stacked && T > 1)
            return fail(h, BYOLO_ERR_ARG, "byolo_forward: non-epistemic decode of a stacked (T>1) detection layer");
        HIPCHK(h, launch_decode(l.det_kind, d, st));
    }
    return BYOLO_OK;
}

// Injected dropout masks: dropout layer `ordinal` (creation order) owns elements_k = S*h*w*cout bits -- element i of its dropout
// input [S,h,w,cout], S = B*T in the stacked part of the graph -- starting at a 32-bit-aligned bit offset.
static int64_t mask_layout(const byolo_t* h, int B, int T, int ordinal, int64_t* elements) {
    int64_t off = 0;
    for (const auto& l : h->layers) {
        if (l.drop_ordinal < 0) continue;
        const int64_t n = (l.stacked ? (int64_t)B * T : B) * l.H * l.W * l.filters;
        if (l.drop_ordinal == ordinal) { if (elements) *elements = n; return off; }
        off += (n + 31) / 32 * 32;
    }
    if (elements) *elements = 0;
    return off;                                              // ordinal == n_dropout: the total
}

extern "C" int32_t byolo_num_dropout(const byolo_t* h) { return h ? h->n_dropout : BYOLO_ERR_ARG; }

static int32_t mask_layout_impl(byolo_t* h, int32_t B, int32_t T, int32_t ordinal, int64_t* bit_offset, int64_t* elements) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_mask_layout: null handle");
    if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }          // (layer shapes / dropout ordinals exist after lowering)
    if (B < 1 || T < 1 || ordinal < 0 || ordinal > h->n_dropout) return fail(h, BYOLO_ERR_ARG, "byolo_mask_layout: bad argument");
    int64_t n = 0;
    const int64_t off = mask_layout(h, B, T, ordinal, &n);
    if (bit_offset) *bit_offset = off;
    if (elements) *elements = n;
    return BYOLO_OK;
}
extern "C" int32_t byolo_mask_layout(byolo_t* h, int32_t B, int32_t T, int32_t ordinal, int64_t* bit_offset, int64_t* elements) {
    return guarded(h, "byolo_mask_layout", [&] { return mask_layout_impl(h, B, T, ordinal, bit_offset, elements); });
}

// the status words after everything enqueued on `st` so far; BLOCKS until the stream is idle
static int32_t read_status(byolo_t* h, hipStream_t st, unsigned* flags, unsigned* layer) {
    if (!h->d_status) { *flags = 0; *layer = 0xFFFFFFFFu; return BYOLO_OK; }
    HIPCHK(h, hipMemcpyAsync(h->h_status, h->d_status, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    *flags = h->h_status[0]; *layer = h->h_status[1];
    return BYOLO_OK;
}
static int32_t range_error(byolo_t* h, const char* what, unsigned flags, unsigned layer) {
    const char* scope = layer < h->layers.size() ? h->layers[layer].scope.c_str() : "?";
    if (flags & 1u)
        return fail(h, BYOLO_ERR_RANGE, "%s: an output of layer %u ('%s') exceeds the split-f16 range (|activation| > %.0f); the reference's float32 "
                    "tensor (lib_yolo/layers.py:550) holds it -- run this model with byolo_set_precision(BYOLO_PREC_F32)", what, layer, scope, 65504.0 / ACT_SCALE);
    return fail(h, BYOLO_ERR_RANGE, "%s: a raw detection output is inf / NaN (non-finite activations upstream, or weights the float32 reference overflows on as well)", what);
}

extern "C" int32_t byolo_set_async(byolo_t* h, int32_t on) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_set_async: null handle");
    h->async_status = on != 0;
    return BYOLO_OK;
}

extern "C" int32_t byolo_status(byolo_t* h, void* stream, uint32_t* flags, int32_t* layer) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_status: null handle");
    if (!h->finalized) return fail(h, BYOLO_ERR_STATE, "byolo_status: call byolo_finalize first");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned f = 0, l = 0xFFFFFFFFu;
    int32_t rc = read_status(h, reinterpret_cast<hipStream_t>(stream), &f, &l); if (rc) return rc;
    if (flags) *flags = f;
    if (layer) *layer = (f & 1u) ? (int32_t)l : -1;
    return f ? range_error(h, "byolo_status", f, l) : BYOLO_OK;
}

extern "C" int32_t byolo_clear_status(byolo_t* h, void* stream) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_clear_status: null handle");
    if (!h->d_status) return BYOLO_OK;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    HIPCHK(h, hipMemsetAsync(h->d_status, 0, sizeof(unsigned), st));
    HIPCHK(h, hipMemsetAsync(h->d_status + 1, 0xFF, sizeof(unsigned), st));
    return BYOLO_OK;
}

extern "C" int32_t byolo_copy_status(byolo_t* h, uint32_t* d_out, void* stream) {
    if (!h || !d_out) return fail(h, BYOLO_ERR_ARG, "byolo_copy_status: null argument");
    if (!h->finalized) return fail(h, BYOLO_ERR_STATE, "byolo_copy_status: call byolo_finalize first");
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (!h->d_status) { HIPCHK(h, hipMemsetAsync(d_out, 0, 2 * sizeof(uint32_t), st)); return BYOLO_OK; }       // fp32 mode: never raised
    HIPCHK(h, hipMemcpyAsync(d_out, h->d_status, 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    return BYOLO_OK;
}

extern "C" int32_t byolo_normalize_u8(byolo_t* h, const uint8_t* d_u8, int64_t n, float* d_f32, void* stream) {
    if (!h || n < 0 || (n > 0 && (!d_u8 || !d_f32))) return fail(h, BYOLO_ERR_ARG, "byolo_normalize_u8: bad argument");
    if ((reinterpret_cast<uintptr_t>(d_u8) & 3) || (reinterpret_cast<uintptr_t>(d_f32) & 15))
        return fail(h, BYOLO_ERR_ARG, "byolo_normalize_u8: d_u8 must be 4-byte and d_f32 16-byte aligned");
    if (n == 0) return BYOLO_OK;
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, launch_u8_to_f32(d_u8, d_f32, n, reinterpret_cast<hipStream_t>(stream)));
    return BYOLO_OK;
}

extern "C" const char* byolo_precision_note(const byolo_t* h) { return h ? h->prec_note.c_str() : ""; }

static int32_t forward_piece(byolo_t* h, const float* d_img, int32_t B, int32_t T, uint64_t seed, int32_t dropout_on,
                             const uint32_t* d_mask_bits, void* d_workspace, size_t workspace_bytes, float* d_boxes, float* d_rows,
                             int32_t* d_kept, int32_t* d_count, void* stream);
struct FwdArgs { const float* d_img; int32_t B, T; uint64_t seed; int32_t dropout_on; const uint32_t* d_mask_bits; void* d_workspace;
                 float* d_boxes; float* d_rows; int32_t* d_kept; int32_t* d_count; };
static int32_t enqueue_forward(byolo_t* h, const FwdArgs& a, hipStream_t st, bool capturing, bool wait_convs, bool heads_only = false);
static int32_t finish_forward(byolo_t* h, hipStream_t st, void* stream);
static int32_t forward_graph(byolo_t* h, const FwdArgs& a, hipStream_t st, bool* done);

// A batch beyond byolo_max_images(h, T) -- the convolutions address their sources with 32-bit byte offsets -- runs as consecutive
// pieces of at most that many images in the SAME workspace: images are independent end to end (the NMS is per image) and every
// piece draws the dropout masks of its position in the logical batch (first_image), so the result does not depend on the cut.
static int32_t forward_impl(byolo_t* h, const float* d_img, int32_t B, int32_t T, uint64_t seed, int32_t dropout_on,
                                 const uint32_t* d_mask_bits, void* d_workspace, size_t workspace_bytes, float* d_boxes, float* d_rows,
                                 int32_t* d_kept, int32_t* d_count, void* stream) {
    if (h && h->finalized && B >= 1 && T >= 1) {
        if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }
        const int64_t cap = piece_cap(h, T);
        if (cap >= 1 && B > cap) {
            if (d_mask_bits && dropout_on)
                return fail(h, BYOLO_ERR_ARG, "byolo_forward: injected masks describe ONE piece: keep B <= byolo_max_images (%lld at T = %d)", (long long)cap, T);
            const int64_t first = h->first_image;
            const size_t img_el = (size_t)h->cfg.img_h * h->cfg.img_w * h->cfg.img_c;
            const size_t out_cap = (size_t)h->cfg.max_out * (h->cfg.nms_mode == BYOLO_NMS_TWO_CLASS ? 2 : 1);
            int32_t rc = BYOLO_OK;
            for (int64_t lo = 0; lo < B && rc == BYOLO_OK; lo += cap) {
                const int32_t n = (int32_t)std::min<int64_t>(cap, B - lo);
                h->first_image = first + lo;
                rc = forward_piece(h, d_img ? d_img + lo * img_el : nullptr, n, T, seed, dropout_on, nullptr, d_workspace, workspace_bytes,
                                   d_boxes ? d_boxes + (size_t)lo * h->n_boxes * h->row_len : nullptr,
                                   d_rows ? d_rows + (size_t)lo * out_cap * h->row_len : nullptr, d_kept ? d_kept + (size_t)lo * out_cap : nullptr,
                                   d_count ? d_count + (size_t)lo * 2 : nullptr, stream);
            }
            h->first_image = first;
            return rc;
        }
    }
    return forward_piece(h, d_img, B, T, seed, dropout_on, d_mask_bits, d_workspace, workspace_bytes, d_boxes, d_rows, d_kept, d_count, stream);
}
extern "C" int32_t byolo_forward(byolo_t* h, const float* d_img, int32_t B, int32_t T, uint64_t seed, int32_t dropout_on,
                                 const uint32_t* d_mask_bits, void* d_workspace, size_t workspace_bytes, float* d_boxes, float* d_rows,
                                 int32_t* d_kept, int32_t* d_count, void* stream) {
    return guarded(h, "byolo_forward", [&] { return forward_impl(h, d_img, B, T, seed, dropout_on, d_mask_bits, d_workspace, workspace_bytes, d_boxes, d_rows, d_kept, d_count, stream); });
}

static int32_t forward_piece(byolo_t* h, const float* d_img, int32_t B, int32_t T, uint64_t seed, int32_t dropout_on,
                             const uint32_t* d_mask_bits, void* d_workspace, size_t workspace_bytes, float* d_boxes, float* d_rows,
                             int32_t* d_kept, int32_t* d_count, void* stream) {
    int32_t rc = check_run(h, B, T, "byolo_forward"); if (rc) return rc;
    if (!d_img || !d_workspace) return fail(h, BYOLO_ERR_ARG, "byolo_forward: null image or workspace");
    if ((d_rows || d_kept || d_count) && !(d_rows && d_kept && d_count))
        return fail(h, BYOLO_ERR_ARG, "byolo_forward: d_rows, d_kept and d_count go together");
    const bool inject = d_mask_bits != nullptr && dropout_on;
    // rate 0 (any rate whose 16-bit threshold is 2^16): the layer is the identity, as tf.layers.dropout(rate=0) is (byolo_rng.h);
    // injected bits are applied whatever the rate
    if (dropout_on && !inject && byolo_drop_is_identity((double)h->cfg.drop_prob)) dropout_on = 0;
    make_plan(h, B, T, inject);
    if (workspace_bytes < h->plan.total)
        return fail(h, BYOLO_ERR_NOMEM, "byolo_forward: workspace %zu < required %zu bytes", workspace_bytes, h->plan.total);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    h->last_ws = d_workspace;
    // (see ev_convs)  Only where a forward fills the chip by itself: a small one -- 8 images at 416 x 416 are 0.5 TFLOP in launches of
    // a few dozen tiles -- gains from running beside the next (config 2: 2480 img/s one after the other, 3110 side by side).
    // opts.serialize_convs: 0 never, 1 forwards of >= 1 TFLOP (default), 2 always.
    bool serialize = h->opts.serialize_convs >= 2;
    if (h->opts.serialize_convs == 1) { double f = 0; (void)byolo_flops(h, B, T, &f); serialize = f >= 1e12; }
    if (h->tshard_T > 0 && (B != 1 || h->tshard_t0 + T > h->tshard_T)) return fail(h, BYOLO_ERR_ARG, "byolo_forward: a T shard (byolo_set_tshard) is ONE image and t0 + T <= T_total");
    if (h->tshard_T > 0 && d_rows) return fail(h, BYOLO_ERR_ARG, "byolo_forward: a T shard (byolo_set_tshard) hands out per-box SUMS in d_boxes; the NMS runs after byolo_finish_tshard (byolo_sort_nms)");
    if (h->tshard_T > 0 && !d_boxes) return fail(h, BYOLO_ERR_ARG, "byolo_forward: a T shard (byolo_set_tshard) needs d_boxes (the per-box sums are its result)");
    if (h->tshard_T > 0)
        for (const auto& l : h->layers)
            if (l.op == OP_DETECTION && l.det_kind != BYOLO_DET_EPISTEMIC) return fail(h, BYOLO_ERR_ARG, "T sharding (byolo_set_tshard) is defined for epistemic detection layers");
    const FwdArgs a{d_img, B, T, seed, dropout_on, d_mask_bits, d_workspace, d_boxes, d_rows, d_kept, d_count};
    // Launch-graph replay (include/byolo.h byolo_plan_opts.graphs): a forward that does NOT fill the chip by itself (below the
    // `serialize` threshold: detect.py's batch-1 loop, BASELINE configs[0..1]) is ~85 dependent launches of 5 - 60 us; its whole
    // launch sequence is captured once per (arguments, plan) and replayed with ONE hipGraphLaunch.  Per-launch profiling, the
    // big forwards and a call that differs in any argument run eagerly / are captured anew.
    if (h->opts.graphs > 0 && !h->profiling && (!serialize || h->opts.graphs >= 2)) {
        bool done = false;
        if (serialize && h->ev_convs_valid && h->convs_stream != st) HIPCHK(h, hipStreamWaitEvent(st, h->ev_convs, 0));      // (graphs == 2 only)
        rc = forward_graph(h, a, st, &done); if (rc) return rc;
        if (done) return finish_forward(h, st, stream);
    }
    const bool wait = serialize && h->ev_convs_valid && h->convs_stream != st;
    // Per-launch profiling wants a quiet device: a forward recorded at level 2, and the one enqueued after it, wait for the other
    // stream's WHOLE convolution stack (otherwise the recorded forward's backbone launches sit between the previous forward's head
    // launches and its head launches between the next forward's backbone launches, and a launch's hipEvent time is no longer its own)
    const bool quiet = h->profiling >= 2 || h->quiet_next;
    h->quiet_next = h->profiling >= 2;
    rc = enqueue_forward(h, a, st, false, wait, h->opts.serialize_heads != 0 && !quiet); if (rc) return rc;
    return finish_forward(h, st, stream);
}

// One forward as ONE hipGraphLaunch.  The first call with a given argument set runs eagerly (function attributes are set, one-off
// shapes never pay for a capture); the second captures the launch sequence of enqueue_forward on the caller's stream (thread-local
// capture mode: the memset node + the kernel nodes, a linear chain), instantiates it and launches it; later calls replay.  A call that
// differs only in what the dropout masks are drawn from (seed, byolo_set_first_image) is captured anew and the executable graph is
// UPDATED in place (hipGraphExecUpdate: same topology, other kernel arguments).  *done = false: nothing was enqueued, run eagerly.
static int32_t forward_graph(byolo_t* h, const FwdArgs& a, hipStream_t st, bool* done) {
    *done = false;
    byolo::GraphKey key; memset(&key, 0, sizeof key);
    key.d_img = a.d_img; key.d_mask_bits = a.d_mask_bits; key.d_workspace = a.d_workspace; key.d_boxes = a.d_boxes; key.d_rows = a.d_rows;
    key.d_kept = a.d_kept; key.d_count = a.d_count; key.B = a.B; key.T = a.T; key.dropout_on = a.dropout_on; key.precision = h->precision;
    key.plan_epoch = h->plan_epoch; key.tshard_t0 = h->tshard_t0; key.tshard_T = h->tshard_T;
    const bool draws = a.dropout_on && h->n_dropout > 0;                   // seed / first_image reach a kernel argument
    const uint64_t seed = draws ? a.seed : 0; const int64_t first = draws ? h->first_image : 0;
    byolo::GraphEntry* e = nullptr;
    for (auto& g : h->graphs) if (g.key.same(key)) { e = &g; break; }
    if (!e) {
        if (h->graphs.size() >= 8) {                                       // drop the least recently used
            size_t lru = 0;
            for (size_t i = 1; i < h->graphs.size(); ++i) if (h->graphs[i].used < h->graphs[lru].used) lru = i;
            if (h->graphs[lru].exec) { (void)hipDeviceSynchronize(); (void)hipGraphExecDestroy(h->graphs[lru].exec); }      // (it may still be replaying)
            h->graphs.erase(h->graphs.begin() + lru);
        }
        h->graphs.emplace_back();
        e = &h->graphs.back(); e->key = key;
    }
    e->used = ++h->graph_clock;
    if (e->no_graph || e->seen++ == 0) return BYOLO_OK;                    // first sight (or not capturable here): eager
    if (e->exec && e->seed == seed && e->first_image == first) {
        HIPCHK(h, hipGraphLaunch(e->exec, st));
        ++h->graph_replays; *done = true;
    } else {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return BYOLO_OK; }     // the caller captures already: its graph, not ours
        // captured on a stream of the handle's own (the caller's may be the legacy default stream, which cannot capture); the
        // executable graph is launched on the caller's
        if (!h->cap_stream && hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); h->cap_stream = nullptr; e->no_graph = true; return BYOLO_OK; }
        if (hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); e->no_graph = true; return BYOLO_OK; }
        const int32_t rc = enqueue_forward(h, a, h->cap_stream, true, false);
        hipGraph_t g = nullptr;
        const hipError_t ce = hipStreamEndCapture(h->cap_stream, &g);
        if (rc != BYOLO_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
        if (ce != hipSuccess || !g) { (void)hipGetLastError(); if (g) (void)hipGraphDestroy(g); e->no_graph = true; return BYOLO_OK; }      // not capturable here: eager from now on
        bool ok = false;
        if (e->exec) {
            hipGraphNode_t bad = nullptr; hipGraphExecUpdateResult ur;
            ok = hipGraphExecUpdate(e->exec, g, &bad, &ur) == hipSuccess;
            if (ok) ++h->graph_updates; else { (void)hipGetLastError(); (void)hipDeviceSynchronize(); (void)hipGraphExecDestroy(e->exec); e->exec = nullptr; }
        }
        if (!ok) {
            ok = hipGraphInstantiate(&e->exec, g, nullptr, nullptr, 0) == hipSuccess;
            if (ok) ++h->graph_captures; else { (void)hipGetLastError(); e->exec = nullptr; e->no_graph = true; }
        }
        (void)hipGraphDestroy(g);
        if (!ok) return BYOLO_OK;
        e->seed = seed; e->first_image = first;
        HIPCHK(h, hipGraphLaunch(e->exec, st));
        *done = true;
    }
    // (the event a forward on another stream would wait for sits behind the whole graph: graphs are for forwards that are not serialised)
    if (!h->ev_convs) HIPCHK(h, hipEventCreateWithFlags(&h->ev_convs, hipEventDisableTiming));
    HIPCHK(h, hipEventRecord(h->ev_convs, st)); h->convs_stream = st; h->ev_convs_valid = true;
    return BYOLO_OK;
}

// Split precision: wait for the forward and read the status words, unless the caller does that itself (byolo_set_async +
// byolo_status).  A raised status is an ERROR here, not a row of inf / NaN: the words are cleared for the next call.
static int32_t finish_forward(byolo_t* h, hipStream_t st, void* stream) {
    if (h->precision == 1 && !h->async_status) {
        unsigned f = 0, ly = 0xFFFFFFFFu;
        int32_t rc = read_status(h, st, &f, &ly); if (rc) return rc;
        if (f) { (void)byolo_clear_status(h, stream); return range_error(h, "byolo_forward", f, ly); }
    }
    return BYOLO_OK;
}

// Everything one forward puts on the stream, in order: the split-K tickets' memset, the image's hi/lo copy, the convolution stack,
// decode, sort + NMS.  `capturing`: the stream is in capture mode (forward_graph) -- no event is recorded or waited for in here.
static int32_t enqueue_forward(byolo_t* h, const FwdArgs& a, hipStream_t st, bool capturing, bool wait_convs, bool heads_only) {
    const float* d_img = a.d_img; const int32_t B = a.B, T = a.T; const uint64_t seed = a.seed; const int32_t dropout_on = a.dropout_on;
    const uint32_t* d_mask_bits = a.d_mask_bits; float* d_boxes = a.d_boxes; float* d_rows = a.d_rows; int32_t* d_kept = a.d_kept; int32_t* d_count = a.d_count;
    const bool inject = d_mask_bits != nullptr && dropout_on;
    int32_t rc;
    char* ws = reinterpret_cast<char*>(a.d_workspace);
    if (h->profiling) {
        h->prof_w = (h->prof_w + 1) % (int)h->prof.size();
        byolo::ProfSlot& ps = h->wslot();
        ps.ev_valid = false; ps.step_valid = false; ps.launches.clear(); ps.n_ev = 0; ps.last_main = -1;
        for (auto& e : ps.ev) if (!e) HIPCHK(h, hipEventCreate(&e));
        HIPCHK(h, hipEventRecord(ps.ev[0], st));
    }
    const bool per_step = h->profiling >= 2;
    bool backbone_marked = false;
    // heads_only (opts.serialize_heads): the wait sits in front of the first HEAD launch instead -- this forward's backbone (52 launches
    // on B images that leave CUs idle: tile quantisation, fixed launch costs) runs beside the previous forward's heads and fills their
    // last rounds.  Measured at config 4 (same box, three interleaved runs each): 368.0 / 369.3 / 368.7 -> 375.1 / 375.0 / 375.7 img/s
    // (+1.8 %; profiles/r6_serialize_heads.md); the rows are the same bits (nothing but the order of independent launches changes).
    bool wait_pending = wait_convs && !capturing;
    if (wait_pending && !(heads_only && h->backbone_end >= 0)) { HIPCHK(h, hipStreamWaitEvent(st, h->ev_convs, 0)); wait_pending = false; }
    HIPCHK(h, launch_zero_words(ws + h->plan.cnt_off, (int64_t)(h->plan.cnt_bytes / 4), st));     // split-K arrival tickets, unit claims
    if (h->precision == 1 && h->img_split)
        HIPCHK(h, launch_f32_to_split(d_img, reinterpret_cast<float*>(ws + h->plan.img_split_off), (int64_t)B * h->cfg.img_h * h->cfg.img_w * h->cfg.img_c, ACT_SCALE, st, h->d_status));
    // everything a convolution step's launch needs, up to the launch itself: sources, epilogue flags, this call's dropout keys /
    // injected bits, tile configuration and split-K plan, the algorithmic FLOPs the step stands for
    auto prep = [&](size_t si, ConvParams& p, int& tile, double& algo) -> int32_t {
        const Step& s = h->steps[si];
        const Layer& l = h->layers[s.layer];
        fill_conv(h, s, d_img, ws, B, T, p);
        if (l.op == OP_CONV && s.mode != STEP_PARTIAL) {
            p.flags = EPI_LEAKY;
            if (l.drop_ordinal >= 0 && dropout_on) {
                const byolo_drop_keys k = byolo_layer_keys(seed, (uint32_t)l.drop_ordinal, (double)h->cfg.drop_prob);
                p.flags |= EPI_DROPOUT; p.k0 = k.k0; p.k1 = k.k1; p.thr = k.thr;
                if (inject) {                                // this layer's first word; bit i = element i of this call's tensor
                    int64_t n_el = 0;
                    const int64_t off = mask_layout(h, B, T, l.drop_ordinal, &n_el);
                    if ((l.filters & 3) && !l.direct) return fail(h, BYOLO_ERR_ARG, "byolo_forward: injected masks need cout %% 4 == 0 on a matrix-pipe dropout layer; '%s' has %d", l.scope.c_str(), l.filters);
                    if (n_el >= ((int64_t)1 << 32)) return fail(h, BYOLO_ERR_ARG, "byolo_forward: injected masks index a dropout tensor with 32 bits; layer '%s' has %lld elements", l.scope.c_str(), (long long)n_el);
                    p.mask_bits = d_mask_bits + off / 32;
                }
                // element index of this call's first output element in the logical batch's [S,h,w,c] tensor (the counter hash);
                // injected bits are indexed inside THIS call's tensor, whatever byolo_set_first_image says
                p.idx_base = inject ? 0 : sample_base(h, l.stacked ? T : 1, l.stacked) * (uint64_t)l.H * l.W * l.filters;
                p.scale = dptr(h, l.scalek_off);             // scale / (1 - p)
            }
            if (l.fused_residual >= 0) {
                p.flags |= EPI_RESIDUAL;
                p.residual = reinterpret_cast<const float*>(ws + h->plan.off[h->layers[l.fused_residual].ref[0]]);
            }
        }
        if (s.mode == STEP_PARTIAL) p.flags = EPI_RAW;          // raw partial sums for the STEP_MAIN launch
        if (h->precision == 1) {
            if (l.op == OP_DETECTION) p.flags |= EPI_F32OUT;    // the decode kernels read plain fp32
        }
        // tile configuration and split-K of the last partial round: decided per (B, T) in make_plan
        tile = h->plan.tile[si];
        const ConvSplit& sp = h->plan.split[si];
        p.full_tiles = sp.full_tiles; p.split_tiles = sp.split_tiles; p.split_blocks = sp.split_blocks; p.ksplit = sp.ksplit;
        p.sk_grid = sp.sk_grid;
        p.slabs = reinterpret_cast<float*>(ws + h->plan.slab_off);
        p.slab_bytes = (uint32_t)conv_split_slab_bytes(sp, tile);
        p.counters = reinterpret_cast<unsigned*>(ws + h->plan.cnt_off) + si * CNT_PER_STEP;
        // ALGORITHMIC FLOPs the step stands for (graph as written): the whole layer on all MC
        // samples for a de-duplicated launch, nothing for the auxiliary partial launch
        const int64_t S_all = l.stacked ? (int64_t)B * T : B;
        algo = (s.mode == STEP_PARTIAL && !s.low) ? 0.0 : 2.0 * (double)(S_all * l.H * l.W) * l.filters * (double)(l.ksize * l.ksize * l.Cin);      // (the `low` launch carries the layer's)
        return BYOLO_OK;
    };
    for (size_t si = 0; si < h->steps.size(); ++si) {
        const Step& s = h->steps[si];
        const Layer& l = h->layers[s.layer];
        if (wait_pending && s.layer >= h->backbone_end) { HIPCHK(h, hipStreamWaitEvent(st, h->ev_convs, 0)); wait_pending = false; }
        if (h->profiling && !backbone_marked && h->backbone_end >= 0 && s.layer >= h->backbone_end) {
            HIPCHK(h, hipEventRecord(h->wslot().ev[1], st)); backbone_marked = true;
        }
        ConvParams p;
        if (s.mode == STEP_FINISH) {
            FinishParams f;
            byolo_drop_keys keys{0, 0, 0};
            const bool drop = l.drop_ordinal >= 0 && dropout_on;
            const uint32_t* bits = nullptr;
            if (drop) {
                keys = byolo_layer_keys(seed, (uint32_t)l.drop_ordinal, (double)h->cfg.drop_prob);
                if (inject) {
                    int64_t n_el = 0;
                    const int64_t off = mask_layout(h, B, T, l.drop_ordinal, &n_el);
                    if (n_el >= ((int64_t)1 << 32)) return fail(h, BYOLO_ERR_ARG, "byolo_forward: injected masks index a dropout tensor with 32 bits; layer '%s' has %lld elements", l.scope.c_str(), (long long)n_el);
                    bits = d_mask_bits + off / 32;
                }
            }
            fill_finish(h, s, ws, B, T, h->precision == 1 ? 2 : 1, drop, keys, bits, inject, f);
            if (per_step) { rc = mark_launch(h, s.layer, -5, (int64_t)f.S * f.H * f.W, f.N, 0, 0.0, st); if (rc) return rc; }
            HIPCHK(h, launch_finish_upsampled(f, st));
            continue;
        }
        if (!s.is_conv()) { fill_conv(h, s, d_img, ws, B, T, p); rc = run_aux_step(h, s, p, st); if (rc) return rc; continue; }
        int tile = 0; double algo = 0.0;
        rc = prep(si, p, tile, algo); if (rc) return rc;
        const ConvSplit& sp = h->plan.split[si];
        if (h->plan.fuse[si]) {
            // back-to-back: the next step (the 1x1 convolution / detection head that alone reads this output) inside this launch
            ConvParams f; int ftile = 0; double falgo = 0.0;
            rc = prep(si + 1, f, ftile, falgo); if (rc) return rc;
            p.f_wpk = f.wpk; p.f_w_bytes = f.w_bytes; p.f_scale = f.scale; p.f_shift = f.shift; p.f_dst = f.dst;
            p.f_N = f.N; p.f_Npad = f.Npad; p.f_ldc = f.ldc; p.f_flags = f.flags; p.f_layer_idx = f.layer_idx;
            p.f_k0 = f.k0; p.f_k1 = f.k1; p.f_thr = f.thr; p.f_idx_base = f.idx_base; p.f_mask_bits = f.mask_bits;
            if (per_step) { rc = mark_launch(h, s.layer, 4256, p.M, l.filters, (int64_t)l.ksize * l.ksize * (s.c_hi - s.c_lo), algo + falgo, st, 1, 0); if (rc) return rc; }
            HIPCHK(h, launch_conv_igemm(p, tile, st));
            ++si;                                               // the follower is done
            continue;
        }
        if (h->plan.wino[si].chunk > 0 && h->precision == 1) { rc = run_wino_split(h, s, l, p, h->plan.wino[si], algo, ws, st); if (rc) return rc; continue; }
        if (h->plan.wino[si].chunk > 0) { rc = run_winograd(h, s, l, p, h->plan.wino[si], tile, algo, ws, st); if (rc) return rc; continue; }
        if (h->plan.stream1x1[si]) {                            // row-streaming 1x1 convolution / detection head
            const int bn = h->plan.stream1x1[si];
            GemmStreamParams q; memset(&q, 0, sizeof q);
            q.a = p.src0; q.a_bytes = p.src0_bytes;
            q.w = p.wpk; q.w_bytes = p.w_bytes; q.wstride = 0;
            q.dst = p.dst; q.C = p.C0; q.N = p.N; q.KT = p.KT; q.n_tiles = bn == 64 ? 1 : p.N / 128;
            const int R = (p.M + 127) / 128;
            q.RT = R + 1;                                       // one weight matrix for all rows
            q.slots = 512 / q.n_tiles; q.q = R / q.slots; q.rem = R % q.slots;
            q.d_ntiles = make_fastdiv((uint32_t)q.n_tiles); q.d_RT = make_fastdiv((uint32_t)q.RT);
            q.epi = l.op == OP_DETECTION ? 2 : 1; q.M = p.M; q.Npad = p.Npad; q.ldc = p.ldc;
            q.scale = p.scale; q.shift = p.shift; q.addend = p.addend; q.hw = l.H * l.W;
            q.flags = p.flags; q.k0 = p.k0; q.k1 = p.k1; q.thr = p.thr; q.idx_base = p.idx_base;
            q.d_hw = p.d_hw; q.d_addT = p.d_addT;
            if (per_step) { rc = mark_launch(h, s.layer, bn == 64 ? 132 : 131, p.M, l.filters, p.C0, algo, st); if (rc) return rc; }
            HIPCHK(h, launch_gemm_stream(q, st));
            continue;
        }
        if (per_step) { rc = mark_launch(h, s.layer, l.direct ? -1 : conv_tile_bn(tile) + (p.kx3 == 1 ? 3000 : (p.kx3 == 2 ? 2000 : (p.split ? 1000 : 0))), p.M, l.filters, (int64_t)l.ksize * l.ksize * (s.c_hi - s.c_lo), algo, st, l.direct ? 1 : (sp.sk_grid > 0 ? -sp.sk_grid : sp.ksplit), l.direct ? 0 : sp.split_tiles); if (rc) return rc; }
        HIPCHK(h, l.direct ? launch_conv_direct(p, st) : launch_conv_igemm(p, tile, st));
    }
    if (per_step) {
        byolo::ProfSlot& ps = h->wslot();
        int e; rc = next_event(h, st, &e); if (rc) return rc;       // end of the last launch on the caller's stream
        if (ps.last_main >= 0) ps.launches[ps.last_main].ev_end = e;
        ps.step_valid = true;
    }
    if (h->profiling) { if (!backbone_marked) HIPCHK(h, hipEventRecord(h->wslot().ev[1], st)); HIPCHK(h, hipEventRecord(h->wslot().ev[2], st)); }
    if (!capturing) {
        if (!h->ev_convs) HIPCHK(h, hipEventCreateWithFlags(&h->ev_convs, hipEventDisableTiming));
        HIPCHK(h, hipEventRecord(h->ev_convs, st)); h->convs_stream = st; h->ev_convs_valid = true;
    }
    float* boxes = d_boxes ? d_boxes : reinterpret_cast<float*>(ws + h->plan.boxes_off);
    if (d_boxes || d_rows) { rc = run_decode(h, ws, boxes, B, T, st, h->tshard_T > 0 ? 1 : 0); if (rc) return rc; }
    if (h->profiling) HIPCHK(h, hipEventRecord(h->wslot().ev[3], st));
    if (d_rows) {
        NmsParams n; memset(&n, 0, sizeof n);
        n.boxes = boxes; n.B = B; n.N = h->n_boxes; n.D = h->row_len; n.obj_idx = h->obj_idx; n.cls_start = h->cls_start;
        n.two_class = h->cfg.nms_mode == BYOLO_NMS_TWO_CLASS; n.max_out = h->cfg.max_out; n.iou_thr = h->cfg.iou_thresh;
        n.ws = ws + h->plan.nms_off; n.ws_bytes = nms_workspace_bytes(B, h->n_boxes);
        n.rows = d_rows; n.kept = d_kept; n.count = d_count; n.general_only = h->opts.nms_general != 0;
        if (n.two_class && h->cfg.cls_cnt != 2) return fail(h, BYOLO_ERR_ARG, "byolo_forward: 2-class NMS needs cls_cnt == 2");
        HIPCHK(h, launch_sort_nms(n, st));
    }
    if (h->profiling) { HIPCHK(h, hipEventRecord(h->wslot().ev[4], st)); h->wslot().ev_valid = true; }
    return BYOLO_OK;
}

extern "C" int32_t byolo_layer_output(const byolo_t* h, int32_t idx, const float** d_ptr, int64_t shape[4]) {
    byolo_t* hh = const_cast<byolo_t*>(h);
    if (!h || idx < 0 || idx >= (int)h->layers.size()) return fail(hh, BYOLO_ERR_ARG, "byolo_layer_output: bad index");
    const Layer& l = h->layers[idx];
    // (the raw outputs of the detection layers are never overwritten by the planner: readable on any handle)
    if (!h->cfg.keep_all_outputs && l.op != OP_DETECTION)
        return fail(hh, BYOLO_ERR_STATE, "byolo_layer_output: handle created without keep_all_outputs");
    if (!h->last_ws || h->plan.B < 0) return fail(hh, BYOLO_ERR_STATE, "byolo_layer_output: no forward has run");
    if (!l.materialized) return fail(hh, BYOLO_ERR_ARG, "byolo_layer_output: layer %d has no tensor of its own (fused or a view)", idx);
    const float* t = reinterpret_cast<const float*>(reinterpret_cast<const char*>(h->last_ws) + h->plan.off[idx]);
    if (d_ptr) *d_ptr = t;
    if (shape) { shape[0] = l.stacked ? (int64_t)h->plan.B * h->plan.T : h->plan.B; shape[1] = l.H; shape[2] = l.W; shape[3] = l.C; }
    return BYOLO_OK;
}

extern "C" int32_t byolo_copy_layer_output(const byolo_t* h, int32_t idx, float* d_dst, int64_t count, void* stream) {
    byolo_t* hh = const_cast<byolo_t*>(h);
    const float* src; int64_t shp[4];
    int32_t rc = byolo_layer_output(h, idx, &src, shp); if (rc) return rc;
    const int64_t n = shp[0] * shp[1] * shp[2] * shp[3];
    if (!d_dst || count != n) return fail(hh, BYOLO_ERR_ARG, "byolo_copy_layer_output: destination of %lld floats, the layer has %lld", (long long)count, (long long)n);
    HIPCHK(hh, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const Layer& l = h->layers[idx];
    if (h->precision == 1 && l.op != OP_DETECTION) HIPCHK(hh, launch_split_to_f32(src, d_dst, n, 1.f / ACT_SCALE, st));
    else if (layer_pitch(l) != l.C)                               // a detection head's rows are padded: dense copy
        HIPCHK(hh, hipMemcpy2DAsync(d_dst, (size_t)l.C * 4, src, (size_t)layer_pitch(l) * 4, (size_t)l.C * 4, (size_t)(n / l.C), hipMemcpyDeviceToDevice, st));
    else HIPCHK(hh, hipMemcpyAsync(d_dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, st));
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// staged tail entry points
// ------------------------------------------------------------------------------------------------
extern "C" int32_t byolo_decode(byolo_t* h, int32_t kind, const float* d_raw, int32_t B, int32_t T, int32_t lh, int32_t lw,
                                const float* priors_hw, int32_t layer_id, float* d_boxes, int64_t n_total,
                                int64_t box_base, void* stream) {
    if (!h || !d_raw || !d_boxes || !priors_hw) return fail(h, BYOLO_ERR_ARG, "byolo_decode: null argument");
    if (kind < 0 || kind > 2 || B < 1 || T < 1 || lh < 1 || lw < 1) return fail(h, BYOLO_ERR_ARG, "byolo_decode: bad argument");
    if (kind != BYOLO_DET_EPISTEMIC && T != 1) return fail(h, BYOLO_ERR_ARG, "byolo_decode: T > 1 only for the epistemic decode");
    HIPCHK(h, hipSetDevice(h->device));
    DecodeParams d; memset(&d, 0, sizeof d);
    d.raw = d_raw; d.boxes = d_boxes; d.B = B; d.T = T; d.lh = lh; d.lw = lw; d.C = h->cfg.cls_cnt;
    d.n_total = n_total; d.box_base = box_base; d.layer_id = layer_id;
    for (int k = 0; k < 3; ++k) { d.ph[k] = priors_hw[2 * k]; d.pw[k] = priors_hw[2 * k + 1]; }
    hipError_t e = launch_decode(kind, d, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? BYOLO_ERR_ARG : BYOLO_ERR_HIP,
                                     "byolo_decode: %s (cls_cnt %d)", hipGetErrorString(e), d.C);
    return BYOLO_OK;
}

extern "C" int32_t byolo_epistemic_stats(byolo_t* h, const float* d_raw, int32_t B, int32_t T, int32_t lh, int32_t lw,
                                         float* d_ev_loc, float* d_epi_covar, float* d_obj_samples, float* d_cls_samples,
                                         void* stream) {
    if (!h || !d_raw) return fail(h, BYOLO_ERR_ARG, "byolo_epistemic_stats: null argument");
    if (B < 1 || T < 1 || lh < 1 || lw < 1) return fail(h, BYOLO_ERR_ARG, "byolo_epistemic_stats: bad argument");
    HIPCHK(h, hipSetDevice(h->device));
    DecodeParams d; memset(&d, 0, sizeof d);
    d.raw = d_raw; d.B = B; d.T = T; d.lh = lh; d.lw = lw; d.C = h->cfg.cls_cnt;
    hipError_t e = launch_epi_stats(d, d_ev_loc, d_epi_covar, d_obj_samples, d_cls_samples, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, BYOLO_ERR_HIP, "byolo_epistemic_stats: %s", hipGetErrorString(e));
    return BYOLO_OK;
}

extern "C" size_t byolo_nms_workspace_bytes(int32_t B, int64_t N) { return nms_workspace_bytes(B, N); }

extern "C" int32_t byolo_sort_nms(byolo_t* h, const float* d_boxes, int32_t B, int64_t N, int32_t D, int32_t obj_idx,
                                  int32_t cls_start_idx, int32_t nms_mode, int32_t max_out, float iou_thresh,
                                  void* d_sort_ws, size_t ws_bytes, float* d_rows, int32_t* d_kept, int32_t* d_count,
                                  void* stream) {
    if (!h || !d_boxes || !d_sort_ws || !d_rows || !d_kept || !d_count) return fail(h, BYOLO_ERR_ARG, "byolo_sort_nms: null argument");
    if (B < 1 || N < 1 || D < 5 || obj_idx < 4 || obj_idx >= D) return fail(h, BYOLO_ERR_ARG, "byolo_sort_nms: bad shape");
    if (nms_mode == BYOLO_NMS_TWO_CLASS && (cls_start_idx < 0 || cls_start_idx + 1 >= D)) return fail(h, BYOLO_ERR_ARG, "byolo_sort_nms: bad cls_start_idx");
    if (max_out < 1 || max_out > 2048) return fail(h, BYOLO_ERR_ARG, "byolo_sort_nms: max_out out of [1,2048]");
    if (ws_bytes < nms_workspace_bytes(B, N)) return fail(h, BYOLO_ERR_NOMEM, "byolo_sort_nms: workspace too small");
    HIPCHK(h, hipSetDevice(h->device));
    NmsParams n; memset(&n, 0, sizeof n);
    n.boxes = d_boxes; n.B = B; n.N = N; n.D = D; n.obj_idx = obj_idx; n.cls_start = cls_start_idx;
    n.two_class = nms_mode == BYOLO_NMS_TWO_CLASS; n.max_out = max_out; n.iou_thr = iou_thresh;
    n.ws = d_sort_ws; n.ws_bytes = ws_bytes; n.rows = d_rows; n.kept = d_kept; n.count = d_count; n.general_only = h->opts.nms_general != 0;
    HIPCHK(h, launch_sort_nms(n, reinterpret_cast<hipStream_t>(stream)));
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// ground-truth encoding and training loss (train_kernels.hip)
// ------------------------------------------------------------------------------------------------
extern "C" int32_t byolo_encode_gt(byolo_t* h, int32_t n_layers, const int32_t* layer_hw, const double* priors_hw,
                                   const float* d_boxes, const int32_t* d_labels, const int32_t* d_counts, int32_t B,
                                   int32_t max_boxes, float ign_thresh, float* d_loc, float* d_obj, int32_t* d_cls,
                                   float* d_ign, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!layer_hw || !priors_hw || !d_loc || !d_obj || !d_cls || !d_ign) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: null argument");
    if (n_layers < 1 || n_layers > 4) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: 1 .. 4 detection layers");
    if (B < 1 || B > 65535 || max_boxes < 0 || (max_boxes > 0 && (!d_boxes || !d_labels))) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: bad batch / boxes");
    if (max_boxes > 0 && (!d_workspace || workspace_bytes < encode_gt_workspace_bytes(B, max_boxes)))
        return fail(h, BYOLO_ERR_NOMEM, "byolo_encode_gt: workspace below byolo_encode_gt_workspace_bytes(B, max_boxes)");
    if (max_boxes > 0 && (reinterpret_cast<uintptr_t>(d_boxes) & 15) != 0) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: d_boxes must be 16-byte aligned");
    if (h) HIPCHK(h, hipSetDevice(h->device));
    EncodeGtParams p; memset(&p, 0, sizeof p);
    int64_t n = 0;
    for (int l = 0; l < n_layers; ++l) {
        p.lh[l] = layer_hw[2 * l]; p.lw[l] = layer_hw[2 * l + 1];
        if (p.lh[l] < 1 || p.lw[l] < 1) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: empty detection layer %d", l);
        p.base[l] = (int)n;
        n += (int64_t)p.lh[l] * p.lw[l] * 3;
        for (int k = 0; k < 3; ++k) {
            p.ph[l][k] = priors_hw[(l * 3 + k) * 2]; p.pw[l][k] = priors_hw[(l * 3 + k) * 2 + 1];
            // lib_yolo/data.py:143-144
            if (!(p.ph[l][k] >= 0 && p.ph[l][k] <= 1 && p.pw[l][k] >= 0 && p.pw[l][k] <= 1))
                return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: prior height and width must be numbers between 0 and 1");
        }
    }
    if (n > (int64_t)1 << 28) return fail(h, BYOLO_ERR_ARG, "byolo_encode_gt: too many prior boxes");
    p.boxes = d_boxes; p.labels = d_labels; p.counts = d_counts; p.B = B; p.max_boxes = max_boxes; p.n_layers = n_layers; p.N = (int)n;
    p.ign_thresh = ign_thresh; p.loc = d_loc; p.obj = d_obj; p.cls = d_cls; p.ign = d_ign; p.best = reinterpret_cast<unsigned*>(d_workspace);
    hipError_t e = launch_encode_gt(p, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, BYOLO_ERR_HIP, "byolo_encode_gt: %s", hipGetErrorString(e));
    return BYOLO_OK;
}

extern "C" size_t byolo_encode_gt_workspace_bytes(int32_t B, int32_t max_boxes) { return encode_gt_workspace_bytes(B, max_boxes); }

extern "C" size_t byolo_loss_workspace_bytes(void) { return loss_workspace_bytes() + 64; }

extern "C" int32_t byolo_loss(byolo_t* h, int32_t kind, int32_t aleatoric_loss, int32_t cls_cnt, const float* d_raw, int32_t pitch,
                              int32_t S, int32_t lh, int32_t lw, const float* d_gt_loc, const float* d_gt_obj,
                              const int32_t* d_gt_cls, const float* d_gt_ign, int64_t gt_stride, double* d_loss,
                              float* d_grad, int32_t grad_pitch, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (!d_raw || !d_gt_loc || !d_gt_obj || !d_gt_cls || !d_gt_ign || !d_loss || !d_workspace) return fail(h, BYOLO_ERR_ARG, "byolo_loss: null argument");
    if (kind != BYOLO_DET_STANDARD && kind != BYOLO_DET_ALEATORIC)
        return fail(h, BYOLO_ERR_ARG, "byolo_loss: the loss exists for the standard and the aleatoric raw outputs (an epistemic layer "
                                      "is an aleatoric one outside inference mode, lib_yolo/model.py:166-173)");
    if (aleatoric_loss && kind != BYOLO_DET_ALEATORIC) return fail(h, BYOLO_ERR_ARG, "byolo_loss: aleatoric_loss needs log_loc_var (aleatoric output)");
    if (S < 1 || lh < 1 || lw < 1 || cls_cnt < 1 || cls_cnt > BYOLO_MAX_CLASSES) return fail(h, BYOLO_ERR_ARG, "byolo_loss: bad shape");
    const int F = 3 * (kind == BYOLO_DET_ALEATORIC ? 10 + 2 * cls_cnt : 5 + cls_cnt);
    if (pitch == 0) pitch = F;
    if (grad_pitch == 0) grad_pitch = F;
    if (pitch < F || (d_grad && grad_pitch < F)) return fail(h, BYOLO_ERR_ARG, "byolo_loss: pitch below the %d values of a cell", F);
    if (gt_stride < (int64_t)lh * lw * 3) return fail(h, BYOLO_ERR_ARG, "byolo_loss: gt_stride below the layer's prior boxes");
    if ((reinterpret_cast<uintptr_t>(d_gt_loc) & 15) != 0) return fail(h, BYOLO_ERR_ARG, "byolo_loss: d_gt_loc must be 16-byte aligned");
    if (workspace_bytes < byolo_loss_workspace_bytes()) return fail(h, BYOLO_ERR_NOMEM, "byolo_loss: workspace too small");
    if (h) HIPCHK(h, hipSetDevice(h->device));
    LossParams p; memset(&p, 0, sizeof p);
    p.raw = d_raw; p.pitch = pitch; p.grad = d_grad; p.grad_pitch = grad_pitch; p.S = S; p.lh = lh; p.lw = lw; p.C = cls_cnt;
    p.aleatoric = kind == BYOLO_DET_ALEATORIC; p.aleatoric_loss = aleatoric_loss != 0;
    p.gt_loc = d_gt_loc; p.gt_obj = d_gt_obj; p.gt_cls = d_gt_cls; p.gt_ign = d_gt_ign; p.gt_stride = gt_stride;
    p.partial = reinterpret_cast<double*>(align_up(reinterpret_cast<uintptr_t>(d_workspace), 64)); p.out = d_loss;
    hipError_t e = launch_loss(p, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail(h, e == hipErrorInvalidValue ? BYOLO_ERR_ARG : BYOLO_ERR_HIP, "byolo_loss: %s", hipGetErrorString(e));
    return BYOLO_OK;
}

// ------------------------------------------------------------------------------------------------
// data-dependent BN initialisation for synthetic weights
// ------------------------------------------------------------------------------------------------
static int32_t calibrate_bn_impl(byolo_t* h, const float* d_img, int32_t B, void* d_workspace, size_t workspace_bytes,
                                      void* stream) {
    int32_t rc = check_run(h, B, 1, "byolo_calibrate_bn"); if (rc) return rc;
    if (!d_img || !d_workspace) return fail(h, BYOLO_ERR_ARG, "byolo_calibrate_bn: null argument");
    make_plan(h, B, 1);
    if (workspace_bytes < h->plan.total) return fail(h, BYOLO_ERR_NOMEM, "byolo_calibrate_bn: workspace %zu < %zu", workspace_bytes, h->plan.total);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    char* ws = reinterpret_cast<char*>(d_workspace);
    double* d_tmp = reinterpret_cast<double*>(ws + h->plan.stats_off);
    float* d_mean = reinterpret_cast<float*>(ws + h->plan.stats_off + (size_t)1024 * 2 * h->maxC * sizeof(double));
    float* d_var = d_mean + h->maxC;
    std::vector<float> sc, sf;
    if (h->precision == 1 && h->img_split)
        HIPCHK(h, launch_f32_to_split(d_img, reinterpret_cast<float*>(ws + h->plan.img_split_off), (int64_t)B * h->cfg.img_h * h->cfg.img_w * h->cfg.img_c, ACT_SCALE, st));
    for (const Step& s : h->steps) {
        Layer& l = h->layers[s.layer];
        const bool split = h->precision == 1;
        ConvParams p;
        if (s.mode == STEP_FINISH) {                                       // the raw sum of the two halves, then the statistics below
            FinishParams f; fill_finish(h, s, ws, B, 1, 0, false, byolo_drop_keys{0, 0, 0}, nullptr, false, f);
            HIPCHK(h, launch_finish_upsampled(f, st));
            p.dst = f.dst; p.M = f.S * f.H * f.W;
        } else {
        fill_conv(h, s, d_img, ws, B, 1, p);
        if (!s.is_conv()) { int32_t rc = run_aux_step(h, s, p, st); if (rc) return rc; continue; }
        const int ctile = split ? conv_split_tile(s.tile, s.kx3 || s.p1) : s.tile;
        if (l.op == OP_DETECTION) { if (split) p.flags |= EPI_F32OUT; HIPCHK(h, launch_conv_igemm(p, ctile, st)); continue; }
        // raw conv output (+ addend for STEP_MAIN), fp32; split precision: the accumulators, ACT_SCALE * 2^wshift * conv
        p.scale = h->d_ones; p.shift = h->d_zeros; p.flags = split ? (s.mode == STEP_PARTIAL ? EPI_RAW : EPI_F32OUT) : 0;
        if (split && l.direct) p.split &= 1;                               // direct launch: plain fp32 output here
        HIPCHK(h, l.direct ? launch_conv_direct(p, st) : launch_conv_igemm(p, ctile, st));
        if (s.mode == STEP_PARTIAL) continue;                              // half of a split conv: statistics at STEP_MAIN / STEP_FINISH
        }
        const int N = l.filters;
        HIPCHK(h, launch_channel_stats(p.dst, p.M, N, d_mean, d_var, d_tmp, st));
        HIPCHK(h, hipMemcpyAsync(h->params[l.p_mean].data.data(), d_mean, sizeof(float) * N, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipMemcpyAsync(h->params[l.p_var].data.data(), d_var, sizeof(float) * N, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (split) {                                                       // statistics of the accumulators -> of the convolution
            for (int c = 0; c < N; ++c) { const float f = 1.f / acc_scale_of(l, c); h->params[l.p_mean].data[c] *= f; h->params[l.p_var].data[c] *= f * f; }
        }
        for (int c = 0; c < N; ++c)            // what byolo_finalize refuses in a checkpoint must not enter through calibration either
            if (!std::isfinite(h->params[l.p_mean].data[c]) || !std::isfinite(h->params[l.p_var].data[c]))
                return fail(h, BYOLO_ERR_ARG, "byolo_calibrate_bn: the batch statistics of layer '%s', channel %d are not finite (inf / NaN "
                            "activations on the calibration frames)", l.scope.c_str(), c);
        fold_layer(h, l, sc, sf);
        if (split) fold_split(l, sc, sf);
        HIPCHK(h, hipMemcpyAsync(dptr(h, l.scale_off), sc.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
        HIPCHK(h, hipMemcpyAsync(dptr(h, l.shift_off), sf.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
        HIPCHK(h, hipStreamSynchronize(st));
        if (l.drop_ordinal >= 0) {
            std::vector<float> sk = sc; scale_keep(h, sk);
            HIPCHK(h, hipMemcpyAsync(dptr(h, l.scalek_off), sk.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
            HIPCHK(h, hipStreamSynchronize(st));
        }
        if (split && s.wino_ok && !l.wshift_u.empty()) {
            std::vector<float> wsc, wsk;
            wino_scales(h, l, wsc, wsk);
            HIPCHK(h, hipMemcpyAsync(dptr(h, l.wscale_off), wsc.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
            HIPCHK(h, hipMemcpyAsync(dptr(h, l.wscalek_off), wsk.data(), sizeof(float) * N, hipMemcpyHostToDevice, st));
            HIPCHK(h, hipStreamSynchronize(st));
        }
        const float* res = l.fused_residual >= 0
            ? reinterpret_cast<const float*>(ws + h->plan.off[h->layers[l.fused_residual].ref[0]]) : nullptr;
        HIPCHK(h, launch_bn_act_inplace(p.dst, p.M, N, dptr(h, l.scale_off), dptr(h, l.shift_off), res, 1, split, st));
    }
    HIPCHK(h, hipStreamSynchronize(st));
    return BYOLO_OK;
}
extern "C" int32_t byolo_calibrate_bn(byolo_t* h, const float* d_img, int32_t B, void* d_workspace, size_t workspace_bytes,
                                      void* stream) {
    return guarded(h, "byolo_calibrate_bn", [&] { return calibrate_bn_impl(h, d_img, B, d_workspace, workspace_bytes, stream); });
}

// ------------------------------------------------------------------------------------------------
// profiling / cost model
// ------------------------------------------------------------------------------------------------
extern "C" int32_t byolo_set_profiling(byolo_t* h, int32_t on) {
    if (!h) return BYOLO_ERR_ARG;
    h->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    for (auto& ps : h->prof) { ps.ev_valid = false; ps.step_valid = false; }
    h->prof_age = 0;
    return BYOLO_OK;
}

extern "C" int32_t byolo_resume_profiling(byolo_t* h, int32_t on) {
    if (!h) return BYOLO_ERR_ARG;
    h->profiling = on < 0 ? 0 : (on > 2 ? 2 : on);
    return BYOLO_OK;
}

extern "C" int32_t byolo_set_profile_depth(byolo_t* h, int32_t depth) {
    if (!h) return BYOLO_ERR_ARG;
    if (depth < 1 || depth > 4096) return fail(h, BYOLO_ERR_ARG, "byolo_set_profile_depth: depth out of [1, 4096]");
    if ((int)h->prof.size() > depth) {
        HIPCHK(h, hipSetDevice(h->device));
        for (size_t i = depth; i < h->prof.size(); ++i) {
            for (auto& e : h->prof[i].ev) if (e) (void)hipEventDestroy(e);
            for (auto& e : h->prof[i].step_ev) (void)hipEventDestroy(e);
        }
    }
    // new slots get event pools as large as the last profiled forward's, so that a run that follows creates none
    const size_t pool = h->wslot().step_ev.size();
    const bool staged = h->wslot().ev[0] != nullptr;
    if (pool || staged) HIPCHK(h, hipSetDevice(h->device));
    h->prof.resize((size_t)depth);
    for (auto& ps : h->prof) {
        ps.ev_valid = false; ps.step_valid = false;
        while (ps.step_ev.size() < pool) { hipEvent_t e; HIPCHK(h, hipEventCreate(&e)); ps.step_ev.push_back(e); }
        if (staged) for (auto& e : ps.ev) if (!e) HIPCHK(h, hipEventCreate(&e));
    }
    h->prof_w = 0; h->prof_age = 0;
    return BYOLO_OK;
}

extern "C" int32_t byolo_select_profile(byolo_t* h, int32_t age) {
    if (!h) return BYOLO_ERR_ARG;
    if (age < 0 || age >= (int)h->prof.size()) return fail(h, BYOLO_ERR_ARG, "byolo_select_profile: age outside the profile depth");
    h->prof_age = age;
    return BYOLO_OK;
}

// T sharded over ranks (the latency alternative at the reference's batch_size = 1, inference_epistemic.py:193,220):
// every rank runs the backbone on the image and the heads on ITS T_local of the image's T samples -- drawing the masks of samples
// t0 .. t0 + T_local - 1 --, byolo_forward then writes the 21 + C per-box sums of layers.py:377-395's reductions into d_boxes
// instead of decoded rows; the caller adds the ranks' sums (ONE all-reduce) and byolo_finish_tshard turns them into rows.
extern "C" int32_t byolo_set_tshard(byolo_t* h, int32_t t0, int32_t T_total) {
    if (!h) return BYOLO_ERR_ARG;
    if (T_total < 0 || t0 < 0 || (T_total > 0 && t0 >= T_total)) return fail(h, BYOLO_ERR_ARG, "byolo_set_tshard: need 0 <= t0 < T_total (T_total = 0 switches it off)");
    h->tshard_t0 = T_total ? t0 : 0; h->tshard_T = T_total;
    return BYOLO_OK;
}
static int32_t finish_tshard_impl(byolo_t* h, float* d_sums, int32_t B, int32_t T_total, void* stream) {
    if (!h || !d_sums || B < 1 || T_total < 1) return fail(h, BYOLO_ERR_ARG, "byolo_finish_tshard: bad argument");
    if (!h->finalized) return fail(h, BYOLO_ERR_STATE, "byolo_finish_tshard: finalize first");
    HIPCHK(h, hipSetDevice(h->device));
    return run_decode(h, nullptr, d_sums, B, T_total, reinterpret_cast<hipStream_t>(stream), 2);
}
extern "C" int32_t byolo_finish_tshard(byolo_t* h, float* d_sums, int32_t B, int32_t T_total, void* stream) {
    return guarded(h, "byolo_finish_tshard", [&] { return finish_tshard_impl(h, d_sums, B, T_total, stream); });
}

extern "C" int32_t byolo_set_first_image(byolo_t* h, int64_t first_image) {
    if (!h) return BYOLO_ERR_ARG;
    if (first_image < 0) return fail(h, BYOLO_ERR_ARG, "byolo_set_first_image: negative index");
    h->first_image = first_image;
    return BYOLO_OK;
}

static int32_t max_images_impl(byolo_t* h, int32_t T, int32_t* max_images) {
    if (!h || !max_images) return BYOLO_ERR_ARG;
    if (T < 1) return fail(h, BYOLO_ERR_ARG, "byolo_max_images: T must be >= 1");
    if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }
    *max_images = (int32_t)piece_cap(h, T);
    return BYOLO_OK;
}
extern "C" int32_t byolo_max_images(byolo_t* h, int32_t T, int32_t* max_images) {
    return guarded(h, "byolo_max_images", [&] { return max_images_impl(h, T, max_images); });
}

extern "C" int32_t byolo_num_steps(const byolo_t* h) {
    if (!h) return BYOLO_ERR_ARG;
    byolo_t* hh = const_cast<byolo_t*>(h);
    if (!hh->rslot().step_valid) return fail(hh, BYOLO_ERR_STATE, "byolo_num_steps: no forward with profiling level 2");
    return (int32_t)hh->rslot().launches.size();
}

extern "C" int32_t byolo_step_profile(byolo_t* h, int32_t i, int32_t* layer, int32_t* variant, int64_t mnk[3], float* ms,
                                      double* algo_flops) {
    if (!h) return BYOLO_ERR_ARG;
    byolo::ProfSlot& ps = h->rslot();
    if (!ps.step_valid) return fail(h, BYOLO_ERR_STATE, "byolo_step_profile: no forward with profiling level 2");
    if (i < 0 || i >= (int)ps.launches.size()) return fail(h, BYOLO_ERR_ARG, "byolo_step_profile: bad index");
    const byolo::Launch& e = ps.launches[i];
    if (layer) *layer = e.layer;
    if (variant) *variant = e.variant;
    if (mnk) { mnk[0] = e.m; mnk[1] = e.n; mnk[2] = e.k; }       // EXECUTED extents of this launch
    if (algo_flops) *algo_flops = e.algo_flops;
    if (ms) {
        HIPCHK(h, hipSetDevice(h->device));
        if (e.ev_end < 0) return fail(h, BYOLO_ERR_STATE, "byolo_step_profile: launch %d has no end event", i);
        HIPCHK(h, hipEventSynchronize(ps.step_ev[e.ev_end]));
        HIPCHK(h, hipEventElapsedTime(ms, ps.step_ev[e.ev_begin], ps.step_ev[e.ev_end]));
    }
    return BYOLO_OK;
}

extern "C" int32_t byolo_step_split(byolo_t* h, int32_t i, int32_t* ksplit, int32_t* split_tiles) {
    if (!h) return BYOLO_ERR_ARG;
    byolo::ProfSlot& ps = h->rslot();
    if (!ps.step_valid) return fail(h, BYOLO_ERR_STATE, "byolo_step_split: no forward with profiling level 2");
    if (i < 0 || i >= (int)ps.launches.size()) return fail(h, BYOLO_ERR_ARG, "byolo_step_split: bad index");
    if (ksplit) *ksplit = ps.launches[i].ksplit;
    if (split_tiles) *split_tiles = ps.launches[i].split_tiles;
    return BYOLO_OK;
}

extern "C" int32_t byolo_stage_ms(byolo_t* h, float ms[4]) {
    if (!h || !ms) return BYOLO_ERR_ARG;
    byolo::ProfSlot& ps = h->rslot();
    if (!ps.ev_valid) return fail(h, BYOLO_ERR_STATE, "byolo_stage_ms: no profiled forward");
    HIPCHK(h, hipSetDevice(h->device));
    HIPCHK(h, hipEventSynchronize(ps.ev[4]));
    for (int i = 0; i < 4; ++i) HIPCHK(h, hipEventElapsedTime(&ms[i], ps.ev[i], ps.ev[i + 1]));
    return BYOLO_OK;
}

static int32_t flops_impl(byolo_t* h, int32_t B, int32_t T, double* flops) {
    if (!h || !flops || B < 1 || T < 1) return BYOLO_ERR_ARG;
    double f = 0;
    for (const auto& l : h->layers) {
        if (l.op != OP_CONV && l.op != OP_DETECTION) continue;
        const double S = l.stacked ? (double)B * T : B;
        f += 2.0 * S * l.H * l.W * (double)(l.ksize * l.ksize * l.Cin) * l.filters;
    }
    *flops = f;
    return BYOLO_OK;
}
extern "C" int32_t byolo_flops(byolo_t* h, int32_t B, int32_t T, double* flops) {
    return guarded(h, "byolo_flops", [&] { return flops_impl(h, B, T, flops); });
}

// ------------------------------------------------------------------------------------------------
// host utility for the input feed: CRC-32C (Castagnoli) as used by the TFRecord framing the
// reference's tf.data pipeline reads (lib_yolo/dataset_utils.py:188-199: TFRecordDataset).
// Slicing-by-8, tables built on first use.
// ------------------------------------------------------------------------------------------------
extern "C" uint32_t byolo_crc32c(const void* data, size_t n) {
    struct Tab { uint32_t t[8][256]; };
    static const Tab table = [] {                   // initialised once, thread-safe (C++11 function-local static)
        Tab x;
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            x.t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) x.t[t][i] = (x.t[t - 1][i] >> 8) ^ x.t[0][x.t[t - 1][i] & 0xFF];
        return x;
    }();
    const uint32_t (&tab)[8][256] = table.t;
    const uint8_t* p = static_cast<const uint8_t*>(data);
    uint32_t crc = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= crc;
        crc = tab[7][lo & 0xFF] ^ tab[6][(lo >> 8) & 0xFF] ^ tab[5][(lo >> 16) & 0xFF] ^ tab[4][lo >> 24] ^
              tab[3][hi & 0xFF] ^ tab[2][(hi >> 8) & 0xFF] ^ tab[1][(hi >> 16) & 0xFF] ^ tab[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) crc = (crc >> 8) ^ tab[0][(crc ^ *p++) & 0xFF];
    return crc ^ 0xFFFFFFFFu;
}

