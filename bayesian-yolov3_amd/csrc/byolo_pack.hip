// byolo_pack.hip -- byolo_finalize (include/byolo.h): BN folding (lib_yolo/layers.py:510-518 with the dropout scale between conv and
// BN, :560-574), weight packing for the MFMA tiles of both precisions (fragment order of the split-f16 kernels, Winograd U in
// double), one upload.  Split out of byolo_api.hip in round 5.
#include "byolo_internal.h"

void fold_layer(const byolo_t* h, const Layer& l, std::vector<float>& scale, std::vector<float>& shift) {
    const int N = l.filters;
    scale.resize(N); shift.resize(N);
    if (l.op == OP_DETECTION) {
        const float* b = h->params[l.p_bias].data.data();
        for (int c = 0; c < N; ++c) { scale[c] = 1.f; shift[c] = b[c]; }
        return;
    }
    const float* g = h->params[l.p_gamma].data.data();
    const float* be = h->params[l.p_beta].data.data();
    const float* m = h->params[l.p_mean].data.data();
    const float* v = h->params[l.p_var].data.data();
    for (int c = 0; c < N; ++c) {                              // layers.py:510-518, eps 1e-5
        const float inv = g[c] * (1.0f / sqrtf(v[c] + 1e-5f));
        scale[c] = inv; shift[c] = be[c] - m[c] * inv;
    }
}

// split precision: the accumulators hold ACT_SCALE * 2^wshift * conv (the stem: conv -- fp32 image, fp32 weights) and the
// output tensor holds ACT_SCALE * value (a detection head: the value itself, fp32) -- powers of two, folded exactly
float acc_scale_of(const Layer& l, int c) {                                    // the direct kernels keep fp32 weights
    return l.in_scale * ((l.direct || l.wshift.empty()) ? 1.f : ldexpf(1.f, l.wshift[c]));
}
void fold_split(const Layer& l, std::vector<float>& scale, std::vector<float>& shift) {
    const float out_scale = l.op == OP_DETECTION ? 1.f : ACT_SCALE;
    for (size_t c = 0; c < scale.size(); ++c) scale[c] *= out_scale / acc_scale_of(l, (int)c);
    for (float& v : shift) v *= out_scale;
}

// inverted dropout's 1 / (1 - p) (layers.py:520-527 via tf.layers.dropout) is folded into the per-channel scale
void scale_keep(const byolo_t* h, std::vector<float>& scale) {
    const float inv_keep = 1.0f / (1.0f - h->cfg.drop_prob);
    for (float& v : scale) v *= inv_keep;
}

// split precision, Winograd launches of layer l: V holds B^T d B of the VALUES (the transform multiplies the stored 4 * value by
// 1/4), U holds 2^wshift_u * (G g G^T): the accumulators are 2^wshift_u * conv, the output tensor holds ACT_SCALE * value
void wino_scales(const byolo_t* h, const Layer& l, std::vector<float>& sc, std::vector<float>& sk) {
    std::vector<float> sf;
    fold_layer(h, l, sc, sf);
    for (size_t c = 0; c < sc.size(); ++c) sc[c] *= ACT_SCALE / ldexpf(1.f, l.wshift_u[c]);
    sk = sc;
    scale_keep(h, sk);
}

// byolo_finalize packs ~62 M weights (hi/lo fragments, Winograd U in double): independent per launch, so on the host's cores.
// BYOLO_FINALIZE_THREADS: worker threads (default: the cores / LOCAL_WORLD_SIZE, at most 32; 1 = in the calling thread).  The packed bytes do not
// depend on it (every element is computed by the same expression; tasks write disjoint ranges).
template <class F>
static bool parallel_tasks(int n, F&& f) {
    const char* e = getenv("BYOLO_FINALIZE_THREADS");
    // one process per GPU: the N ranks of a node finalize at the same time -- each takes its share of the hardware threads
    // (LOCAL_WORLD_SIZE is what torchrun exports; 8 ranks x 32 packing threads on one host was round 4's default)
    const char* lw = getenv("LOCAL_WORLD_SIZE");
    const int ranks = std::max(1, lw ? atoi(lw) : 1);
    const int want = e ? atoi(e) : std::max(1, (int)std::thread::hardware_concurrency() / ranks);
    const int nt = std::min(n, std::max(1, std::min(want, 32)));
    std::atomic<int> next{0};
    std::atomic<bool> ok{true};
    auto work = [&] {
        try { for (int i; (i = next.fetch_add(1)) < n;) f(i); }
        catch (...) { ok = false; }
    };
    std::vector<std::thread> th;
    try { for (int t = 1; t < nt; ++t) th.emplace_back(work); } catch (...) {}      // (no more threads to be had: fewer workers)
    work();
    for (auto& t : th) t.join();
    return ok;
}

static int32_t finalize_impl(byolo_t* h) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "byolo_finalize: null handle");
    if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }
    // Every parameter must be a number.  The reference would carry an inf / NaN from a checkpoint (tf.train.Saver.restore,
    // inference_epistemic.py:58) silently into its float32 outputs; here it would also poison the per-channel weight scales.
    for (const auto& l : h->layers) {
        if (l.op != OP_CONV && l.op != OP_DETECTION) continue;
        for (int pi : {l.p_kernel, l.p_bias, l.p_gamma, l.p_beta, l.p_mean, l.p_var}) {
            if (pi < 0) continue;
            for (float v : h->params[pi].data)
                if (!std::isfinite(v)) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: variable '%s' holds a non-finite value", h->params[pi].name.c_str());
        }
        if (l.p_var >= 0)
            for (float v : h->params[l.p_var].data)
                if (!(v + 1e-5f > 0.f)) return fail(h, BYOLO_ERR_ARG, "byolo_finalize: variable '%s' holds a variance <= -eps (rsqrt of a negative number)", h->params[l.p_var].name.c_str());
    }
    // Split storage keeps activations in groups of 4 channels.  A graph it cannot express (no reference model has one) runs in
    // the fp32 mode instead of being refused: byolo_get_precision / byolo_precision_note tell.
    if (h->precision == 0 && !h->prec_note.empty() && h->prec_requested == 1) h->precision = 1;      // the request stands; decide again
    h->prec_note.clear();
    if (h->precision == 1)
        for (const auto& l : h->layers)
            if (l.op == OP_CONV && (l.filters % 4)) {
                char buf[256];
                snprintf(buf, sizeof buf, "fp32 mode: split-f16 storage needs output channels in groups of 4, layer '%s' has %d", l.scope.c_str(), l.filters);
                h->prec_note = buf; h->precision = 0; h->plan.B = -1; h->plan.T = -1; ++h->plan_epoch;
                static const bool quiet = [] { const char* e = getenv("BYOLO_QUIET"); return e && atoi(e); }();
                if (!quiet) fprintf(stderr, "byolo: %s\n", buf);
                break;
            }
    HIPCHK(h, hipSetDevice(h->device));
    if (!h->d_status) {
        HIPCHK(h, hipMalloc((void**)&h->d_status, 2 * sizeof(unsigned)));
        HIPCHK(h, hipHostMalloc((void**)&h->h_status, 2 * sizeof(unsigned), hipHostMallocDefault));
        const unsigned init[2] = {0u, 0xFFFFFFFFu};
        HIPCHK(h, hipMemcpy(h->d_status, init, sizeof init, hipMemcpyHostToDevice));
        h->h_status[0] = 0u; h->h_status[1] = 0xFFFFFFFFu;
    }
    // layout of the device blob
    size_t off = 0; const int maxC = h->maxC;
    for (auto& st : h->steps) {
        if (!st.is_conv()) continue;
        Layer& l = h->layers[st.layer];
        const int Cs = st.c_hi - st.c_lo, K = l.ksize * l.ksize * Cs, N = l.filters;
        st.w_off = off; off += align_up((size_t)K * st.Npad, 64);      // tile / Npad / wino_ok: set by lower()
        l.tile = st.tile; l.Npad = st.Npad;
        if (st.wino_ok) { st.wino_off = off; off += align_up((size_t)16 * Cs * st.Npad, 64); }
        if (st.mode == STEP_PARTIAL && !st.low) continue;       // raw accumulators: no scale / shift (the `low` launch owns the layer's, for its STEP_FINISH)
        l.scale_off = off; off += align_up((size_t)std::max(N, st.Npad), 64);   // readable (zeros) up to Npad
        l.shift_off = off; off += align_up((size_t)std::max(N, st.Npad), 64);
        if (l.drop_ordinal >= 0) { l.scalek_off = off; off += align_up((size_t)std::max(N, st.Npad), 64); }
        if (st.wino_ok) {                                       // split precision: the scales that go with U's own power-of-two shifts
            l.wscale_off = off; off += align_up((size_t)std::max(N, st.Npad), 64);
            l.wscalek_off = off; off += align_up((size_t)std::max(N, st.Npad), 64);
        }
    }
    std::vector<float> blob(off, 0.f);
    std::vector<float> sc, sf;
    h->img_split = false;
    if (h->precision == 1) {
        const bool per_layer = h->opts.wshift_per_layer != 0;
        const bool ok = parallel_tasks((int)h->layers.size(), [&](int li) {
            Layer& l = h->layers[li];
            if (l.op != OP_CONV && l.op != OP_DETECTION) return;
            // a direct convolution reads the image as it is (fp32); a matrix-pipe convolution reads a hi/lo copy of it
            l.in_scale = (l.direct && l.prev < 0) ? 1.f : ACT_SCALE;
            // One power of two PER OUTPUT CHANNEL (folded into scale[n], exactly): the column's largest |w'| lands in
            // [2^13, 2^14), so a filter whose weights are 2^-10 of its neighbours' -- a checkpoint whose BN gammas absorbed the
            // scale, e.g. -- keeps its 22 bits.  (One shift per layer gave such a column 12.)  Clamped: 2^shift stays finite.
            const Param& k = h->params[l.p_kernel];
            const int N = l.filters;
            const size_t rows = k.data.size() / (size_t)N;          // HWIO == [K][N]
            std::vector<float> mx((size_t)N, 0.f);
            for (size_t r = 0; r < rows; ++r) {
                const float* kr = k.data.data() + r * N;
                for (int n = 0; n < N; ++n) mx[n] = std::max(mx[n], std::fabs(kr[n]));
            }
            l.wshift.assign((size_t)N, 0);
            l.wshift_u.clear();
            for (int n = 0; n < N; ++n) {
                int e = 0;
                if (mx[n] > 0.f) (void)std::frexp(mx[n], &e);      // mx = m * 2^e, m in [0.5, 1)
                l.wshift[n] = mx[n] > 0.f ? std::min(126, std::max(-126, 14 - e)) : 0;      // 2^shift and 2^-shift are normal floats
            }
            // BYOLO_WSHIFT_PER_LAYER=1 (A/B in tests/test_robustness.py): one shift per layer, from the layer's largest weight
            if (per_layer) {
                int lo = 127;
                for (int n = 0; n < N; ++n) if (mx[n] > 0.f) lo = std::min(lo, l.wshift[n]);
                l.wshift.assign((size_t)N, lo == 127 ? 0 : lo);
            }
        });
        if (!ok) return fail(h, BYOLO_ERR_NOMEM, "byolo_finalize: out of host memory");
        for (const auto& l : h->layers)
            if ((l.op == OP_CONV || l.op == OP_DETECTION) && !l.direct && l.prev < 0) h->img_split = true;
    }
    // Tasks: (step, 0) packs the launch's weights, (step, 1) its Winograd-domain weights; the largest first.  Every task writes its
    // own range of the blob and its own step / layer fields (a Winograd launch is the only step of its layer).
    struct PackTask { int step, kind; double cost; };
    std::vector<PackTask> tasks;
    decide_loops(h);                                         // which split-f16 loop every launch runs on (and packs its weights for)
    for (size_t si = 0; si < h->steps.size(); ++si) {
        const Step& st = h->steps[si];
        if (!st.is_conv()) continue;
        const Layer& l = h->layers[st.layer];
        const double k = (double)l.ksize * l.ksize * (st.c_hi - st.c_lo) * l.filters;
        tasks.push_back({(int)si, 0, k});
        if (st.wino_ok) tasks.push_back({(int)si, 1, 2.0 * 16 / 9 * k});
    }
    std::stable_sort(tasks.begin(), tasks.end(), [](const PackTask& a, const PackTask& b) { return a.cost > b.cost; });
    const bool packed = parallel_tasks((int)tasks.size(), [&](int ti) {
        Step& st = h->steps[tasks[ti].step];
        const Layer& l = h->layers[st.layer];
        const int Cs = st.c_hi - st.c_lo, taps = l.ksize * l.ksize, N = l.filters;
        const float* w = h->params[l.p_kernel].data.data();     // HWIO == [K][N], k = (ky*ks + kx)*Cin + c
        if (tasks[ti].kind == 0) {
            float* dst = blob.data() + st.w_off;                   // (st.kx3 / st.p1: decide_loops, above -- they fix the K-tile order)
            if (l.direct) memcpy(dst, w, sizeof(float) * (size_t)taps * l.Cin * N);
            else if (h->precision == 1) {
                const int cts = Cs / 32;
                // split-f16 weights (mfma_pipe.h): w' = w * 2^wshift with the layer's largest |w'| in [2^13, 2^14), each
                // element as hi = RNE_f16(w'), lo = RNE_f16(w' - hi), in FRAGMENT ORDER: [K-tile][32-column block][step s]
                // [plane: hi, lo][lane = 32 * half + column][8 fp16: k = 32 kt + 16 s + 8 half + 0..7] -- the operand
                // registers of v_mfma_f32_32x32x16_f16 as one coalesced 1 KB load per (step, plane).
                // K-tile order: (tap, chunk); shared-tap launches: (ky, chunk, kx)
                _Float16* d16 = reinterpret_cast<_Float16*>(dst);
                std::vector<float> ws((size_t)N);
                for (int nn = 0; nn < N; ++nn) ws[nn] = ldexpf(1.f, l.wshift[nn]);
                const size_t blocks = st.Npad / 32;
                for (int tap = 0; tap < taps; ++tap)
                    for (int c = 0; c < Cs; ++c) {
                        const int kk = c & 31, step = kk >> 4, half = (kk >> 3) & 1, e = kk & 7;
                        const int kt = st.kx3 ? ((tap / 3) * cts + (c >> 5)) * 3 + tap % 3 : tap * cts + (c >> 5);
                        const float* wr = w + ((size_t)tap * l.Cin + st.c_lo + c) * N;
                        for (int nn = 0; nn < N; ++nn) {
                            const float v = wr[nn] * ws[nn];
                            const _Float16 hi = (_Float16)v;
                            _Float16* d = d16 + (((size_t)kt * blocks + (nn >> 5)) * 4 + step * 2) * 512 + (half * 32 + (nn & 31)) * 8 + e;
                            d[0] = hi; d[512] = (_Float16)(v - (float)hi);
                        }
                    }
            } else {
                for (int tap = 0; tap < taps; ++tap)
                    for (int c = 0; c < Cs; ++c) {                  // this launch's channel slice of every tap
                        const int k = tap * Cs + c, kt = k >> 5, kk = k & 31;
                        const float* wr = w + ((size_t)tap * l.Cin + st.c_lo + c) * N;
                        float* d = dst + ((size_t)kt * st.Npad) * 32 + kk;
                        for (int nn = 0; nn < N; ++nn) d[(size_t)nn * 32] = wr[nn];
                    }
            }
            return;
        }
        if (st.wino_ok && h->precision == 1 && wino_split_ok(Cs, N) && st.Npad == N) {
            // Winograd in split arithmetic (wino_split.hip): U[xi][c][n] = (G g G^T)[xi] in double, rounded once; one power of two per
            // output channel over all 16 points; hi/lo pairs in fragment order, K-tile order (point, chunk)
            const int NP = 16, KC = Cs;                                              // points; K rows per point
            std::vector<float> U((size_t)NP * KC * N);
            float g9[9], u16[16];
            for (int c = 0; c < Cs; ++c)
                for (int nn = 0; nn < N; ++nn) {
                    for (int tap = 0; tap < 9; ++tap) g9[tap] = w[((size_t)tap * l.Cin + st.c_lo + c) * N + nn];
                    wino_weight_transform(g9, u16);
                    for (int xi = 0; xi < 16; ++xi) U[((size_t)xi * Cs + c) * N + nn] = u16[xi];
                }
            Layer& lw = h->layers[st.layer];
            lw.wshift_u.assign((size_t)N, 0);
            std::vector<float> wsu((size_t)N), mxu((size_t)N, 0.f);
            for (size_t r = 0; r < (size_t)NP * KC; ++r) {
                const float* ur = U.data() + r * N;
                for (int nn = 0; nn < N; ++nn) mxu[nn] = std::max(mxu[nn], std::fabs(ur[nn]));
            }
            for (int nn = 0; nn < N; ++nn) {
                const float mx = mxu[nn];
                int e = 0;
                if (mx > 0.f) (void)std::frexp(mx, &e);
                lw.wshift_u[nn] = mx > 0.f ? std::min(126, std::max(-126, 14 - e)) : 0;
                wsu[nn] = ldexpf(1.f, lw.wshift_u[nn]);
            }
            _Float16* d16 = reinterpret_cast<_Float16*>(blob.data() + st.wino_off);
            const size_t blocks = N / 32;
            const int cts = KC / 32;
            for (int xi = 0; xi < NP; ++xi)
                for (int c = 0; c < KC; ++c) {
                    const int kk = c & 31, step = kk >> 4, half = (kk >> 3) & 1, e = kk & 7;
                    const int kt = xi * cts + (c >> 5);
                    const float* ur = U.data() + ((size_t)xi * KC + c) * N;
                    for (int nn = 0; nn < N; ++nn) {
                        const float v = ur[nn] * wsu[nn];
                        const _Float16 hi = (_Float16)v;
                        _Float16* d = d16 + (((size_t)kt * blocks + (nn >> 5)) * 4 + step * 2) * 512 + (half * 32 + (nn & 31)) * 8 + e;
                        d[0] = hi; d[512] = (_Float16)(v - (float)hi);
                    }
                }
        }
        if (st.wino_ok && h->precision == 0) {                  // U[xi][c][n] = (G g G^T)[xi], each xi packed like a 1x1 conv
            float* u = blob.data() + st.wino_off;
            const size_t xi_stride = (size_t)(Cs / 32) * st.Npad * 32;
            float g9[9], u16[16];
            for (int c = 0; c < Cs; ++c)
                for (int nn = 0; nn < N; ++nn) {
                    for (int tap = 0; tap < 9; ++tap) g9[tap] = w[((size_t)tap * l.Cin + c) * N + nn];
                    wino_weight_transform(g9, u16);
                    float* d = u + ((size_t)(c >> 5) * st.Npad + nn) * 32 + (c & 31);
                    for (int xi = 0; xi < 16; ++xi) d[(size_t)xi * xi_stride] = u16[xi];
                }
        }
    });
    if (!packed) return fail(h, BYOLO_ERR_NOMEM, "byolo_finalize: out of host memory");
    for (auto& st : h->steps) {
        if (!st.is_conv()) continue;
        const Layer& l = h->layers[st.layer];
        const int N = l.filters;
        if (st.mode == STEP_PARTIAL && !st.low) continue;
        fold_layer(h, l, sc, sf);
        if (h->precision == 1) fold_split(l, sc, sf);
        memcpy(blob.data() + l.scale_off, sc.data(), sizeof(float) * N);
        memcpy(blob.data() + l.shift_off, sf.data(), sizeof(float) * N);
        if (l.drop_ordinal >= 0) {
            scale_keep(h, sc);
            memcpy(blob.data() + l.scalek_off, sc.data(), sizeof(float) * N);
        }
        if (h->precision == 1 && st.wino_ok && !l.wshift_u.empty()) {
            std::vector<float> wsc, wsk;
            wino_scales(h, l, wsc, wsk);
            memcpy(blob.data() + l.wscale_off, wsc.data(), sizeof(float) * N);
            memcpy(blob.data() + l.wscalek_off, wsk.data(), sizeof(float) * N);
        }
    }
    if (h->d_blob && h->blob_floats != off) { HIPCHK(h, hipFree(h->d_blob)); h->d_blob = nullptr; }
    if (!h->d_blob) HIPCHK(h, hipMalloc((void**)&h->d_blob, sizeof(float) * off));
    h->blob_floats = off;
    HIPCHK(h, hipMemcpy(h->d_blob, blob.data(), sizeof(float) * off, hipMemcpyHostToDevice));
    if (!h->d_ones) {
        std::vector<float> ones((size_t)maxC, 1.f);
        HIPCHK(h, hipMalloc((void**)&h->d_ones, sizeof(float) * maxC));
        HIPCHK(h, hipMalloc((void**)&h->d_zeros, sizeof(float) * maxC));
        HIPCHK(h, hipMemcpy(h->d_ones, ones.data(), sizeof(float) * maxC, hipMemcpyHostToDevice));
        HIPCHK(h, hipMemset(h->d_zeros, 0, sizeof(float) * maxC));
    }
    h->finalized = true;
    h->plan.B = -1; h->plan.T = -1; ++h->plan_epoch;            // kernel choices depend on what was packed: plan again
    return BYOLO_OK;
}
extern "C" int32_t byolo_finalize(byolo_t* h) {
    return guarded(h, "byolo_finalize", [&] { return finalize_impl(h); });
}

