// byolo_plan.hip -- the workspace planner of the forward (host code only): which tensor lives where in the arena and until when,
// the launch geometry of every step, Winograd chunking, back-to-back fusion decisions.  Split out of byolo_api.hip in round 5
// (the one data-corruption bug of round 4 -- a fused launch whose output aliased its own input -- lived here
// and was caught by the range sentinel at full size, not by a planner test): byolo_plan_* below expose the plan to
// tests/test_planner.py, which checks on the CPU that no tensor is written while another one sharing its memory is still live.
#include "byolo_internal.h"

// ------------------------------------------------------------------------------------------------
// workspace planning (liveness-based first-fit; 288 GB HBM is not a reason to thrash the caches)
// ------------------------------------------------------------------------------------------------
// Floats per pixel of a layer's tensor.  A matrix-pipe detection head pads its 3 * (5 + C) or 3 * 2 * (5 + C) channels (21, 42, ..)
// to a multiple of 4: the epilogue then stores 16-byte vectors like every other convolution (the two padding channels come out
// as zeros: zero weight columns, zero bias); decode reads with that pitch, byolo_copy_layer_output hands out the dense tensor.
int layer_pitch(const Layer& l) { return (l.op == OP_DETECTION && !l.direct) ? (l.C + 3) / 4 * 4 : l.C; }

int64_t tensor_bytes(const byolo_t* h, int id, int B, int T) {
    const int n = (int)h->layers.size();
    if (id >= n) {                                                // auxiliary: one row per IMAGE pixel (stacked: per sample pixel)
        const AuxTensor& a = h->aux[id - n];
        return (int64_t)align_up((size_t)((int64_t)B * (a.stacked ? T : 1) * a.H * a.W * a.C) * sizeof(float), 256);
    }
    const Layer& l = h->layers[id];
    const int64_t S = l.stacked ? (int64_t)B * T : B;
    return (int64_t)align_up((size_t)(S * l.H * l.W * layer_pitch(l)) * sizeof(float), 256);
}

// rows (M) and K-tiles of one launch -- the same arithmetic as fill_conv
void step_geometry(const byolo_t* h, const Step& st, int B, int T, int* M, int* KT) {
    const Layer& l = h->layers[st.layer];
    const bool per_image = st.mode == STEP_REP || (st.mode == STEP_PARTIAL && !st.low);
    const int64_t S = (l.stacked && !per_image) ? (int64_t)B * T : B;
    *M = (int)(S * (l.H >> (st.low ? 1 : 0)) * (l.W >> (st.low ? 1 : 0)));
    *KT = l.ksize * l.ksize * ((st.c_hi - st.c_lo) / 32);
}

// Which K loop a split-f16 launch runs on -- the shared-tap stages of a 3x3 / stride-1 convolution over one plain source (kx3), the
// uniform loop of a 1x1 convolution (p1), or the general loop -- follows from the graph and the precision alone.  Decided when the
// graph is lowered (a plan made before byolo_finalize is then the plan made after it: byolo_workspace_bytes, tests/test_planner.py)
// and again by byolo_finalize once it has settled the precision; the weights are packed in the K-tile order of the loop.
void decide_loops(byolo_t* h) {
    const bool kx3_on = h->opts.kx3 != 0, p1_on = h->opts.p1 != 0;
    for (auto& st : h->steps) {
        st.kx3 = false; st.p1 = false;
        if (!st.is_conv()) continue;
        const Layer& l = h->layers[st.layer];
        if (l.direct || h->precision != 1) continue;
        st.kx3 = kx3_on && l.ksize == 3 && l.stride == 1 && st.in.n == 1 && st.in.s[0].sh == 0 && st.in.s[0].layer >= 0 && st.Npad >= 64 &&
                 (st.mode == STEP_NORMAL || st.mode == STEP_REP);
        st.p1 = p1_on && l.ksize == 1 && l.stride == 1 && st.in.n == 1 && st.in.s[0].layer >= 0 && st.Npad >= 64;
    }
}

void make_plan(byolo_t* h, int B, int T, bool inject) {
    Plan& p = h->plan;
    if (p.B == B && p.T == T && h->plan_inject == inject) return;
    h->plan_inject = inject;
    const int n = (int)h->layers.size() + (int)h->aux.size();     // tensor ids: layers, then auxiliaries
    p.B = B; p.T = T; p.off.assign(n, -1);
    struct Blk { int64_t off, size; };
    std::vector<Blk> free_list;
    int64_t end = 0;
    auto alloc = [&](int64_t sz) -> int64_t {
        for (size_t i = 0; i < free_list.size(); ++i) {
            if (free_list[i].size >= sz) {
                const int64_t o = free_list[i].off;
                free_list[i].off += sz; free_list[i].size -= sz;
                if (!free_list[i].size) free_list.erase(free_list.begin() + i);
                return o;
            }
        }
        if (!free_list.empty() && free_list.back().off + free_list.back().size == end) {   // grow the tail block
            const int64_t o = free_list.back().off; end = o + sz; free_list.pop_back(); return o;
        }
        const int64_t o = end; end += sz; return o;
    };
    auto release = [&](int64_t off, int64_t sz) {
        Blk b{off, sz};
        auto it = std::lower_bound(free_list.begin(), free_list.end(), b, [](const Blk& x, const Blk& y) { return x.off < y.off; });
        it = free_list.insert(it, b);
        if (it + 1 != free_list.end() && it->off + it->size == (it + 1)->off) { it->size += (it + 1)->size; free_list.erase(it + 1); }
        if (it != free_list.begin() && (it - 1)->off + (it - 1)->size == it->off) { (it - 1)->size += it->size; free_list.erase(it); }
    };
    // Back-to-back fusion (conv_igemm.hip fused_tail): a shared-tap 3x3 convolution with exactly 256 output channels (the 8-wave
    // 128 x 256 tile), followed by a 1x1 convolution / detection head of <= 128 output channels that is the ONLY reader of its
    // output: the follower runs inside the 3x3 launch from LDS, the 3x3 layer's output tensor is never written.  Decided here,
    // before the arena is laid out: the fused launch reads the 3x3 layer's INPUT while it writes the FOLLOWER's output, so those
    // two must not share memory (unfused, the follower's output may take the place of the 3x3 layer's dead input).
    // Measured at config 4 (round 4, three A/B runs on one box each, gpurun_out/r4f-r4h): the three pairs of the 76x76 head
    // 2.11 + 0.56 -> 2.60, 2.11 + 0.56 -> 2.59, 2.09 + 0.38 -> 2.32 ms; 342.6 -> 348.3, 344.5 -> 349.9, 347.1 -> 350.4 img/s.
    // opts.b2b: 0 never, 1 launches of >= 4 rounds of 256 workgroups (default), 2 every eligible pair (tests)
    p.fuse.assign(h->steps.size(), 0);
    { const int b2b = h->opts.b2b;
      for (size_t si = 0; b2b && h->precision == 1 && !inject && !h->cfg.keep_all_outputs && si + 1 < h->steps.size(); ++si) {
          const Step& s = h->steps[si];
          const Layer& l = h->layers[s.layer];
          if (!(s.is_conv() && s.kx3 && s.mode == STEP_NORMAL && l.op == OP_CONV && !l.direct && l.filters == 256 && s.Npad == 256 && l.fused_residual < 0)) continue;
          int M, KT; step_geometry(h, s, B, T, &M, &KT);
          if (b2b < 2 && (int64_t)((M + 127) / 128) < 4 * 256) continue;
          const Step& s2 = h->steps[si + 1];
          const Layer& l2 = h->layers[s2.layer];
          const int out_t = s.out_tensor;
          if (s2.is_conv() && s2.mode == STEP_NORMAL && s2.p1 && !l2.direct && l2.ksize == 1 && l2.stride == 1 && s2.in.n == 1 &&
              s2.in.s[0].layer == out_t && !s2.in.s[0].tile && s2.in.s[0].sh == 0 && s2.in.s[0].C == 256 && s2.c_lo == 0 && s2.c_hi == 256 &&
              l2.fused_residual < 0 && s2.Npad <= 128 && (layer_pitch(l2) % 4) == 0 && l2.H == l.H && l2.W == l.W && l2.stacked == l.stacked &&
              h->last_use[out_t] == (int)si + 1)
              p.fuse[si] = 1;
      }
    }
    for (int si = 0; si < (int)h->steps.size(); ++si) {
        const Layer& l = h->layers[h->steps[si].layer];
        const int t = h->steps[si].out_tensor;
        if (!(si > 0 && p.fuse[si - 1])) p.off[t] = alloc(tensor_bytes(h, t, B, T));      // (a fused follower's output exists since the step before)
        if (p.fuse[si]) { const int t2 = h->steps[si + 1].out_tensor; p.off[t2] = alloc(tensor_bytes(h, t2, B, T)); }
        if (h->cfg.keep_all_outputs) continue;
        for (int k = 0; k < n; ++k)
            if (p.off[k] >= 0 && h->last_use[k] == si) release(p.off[k], tensor_bytes(h, k, B, T));
        if (h->last_use[t] < 0 && l.op != OP_DETECTION) release(p.off[t], tensor_bytes(h, t, B, T));   // dead output
    }
    // detection raw outputs must survive until their decode (same step) -> they are released one step
    // late by construction (last_use == -1 handled below): keep them simple: never reuse det outputs.
    p.arena = align_up((size_t)end, 256);
    p.boxes_off = p.arena;
    size_t o = p.boxes_off + align_up((size_t)B * h->n_boxes * h->row_len * sizeof(float), 256);
    p.nms_off = o; o += align_up(nms_workspace_bytes(B, h->n_boxes), 256);
    p.stats_off = o; o += align_up((size_t)1024 * 2 * h->maxC * sizeof(double) + 2 * h->maxC * sizeof(float), 256);
    p.img_split_off = o; if (h->img_split) o += align_up((size_t)B * h->cfg.img_h * h->cfg.img_w * h->cfg.img_c * sizeof(float), 256);
    // launch geometry per step: tile configuration and the split-K of the last partial round (shape-only)
    p.split.assign(h->steps.size(), ConvSplit{0, 0, 0, 1});
    p.tile.assign(h->steps.size(), 0);
    size_t slab = 0;
    for (size_t si = 0; si < h->steps.size(); ++si) {
        const Step& s = h->steps[si];
        const Layer& l = h->layers[s.layer];
        if (!s.is_conv() || l.direct) continue;
        int M, KT; step_geometry(h, s, B, T, &M, &KT);
        // Grid fill: a 128x128 tiling of a small-M layer (deep backbone layers at small batch) leaves CUs
        // idle; the 128x64 tile doubles the block count (the packed weight layout [K/32][Npad][32] does not
        // depend on BN when N % 128 == 0).
        int tile = s.tile;
        if (tile == TILE_128x128 && (l.filters % 128) == 0 && (int64_t)((M + 127) / 128) * (l.filters / 128) < 512) tile = TILE_128x64;
        // split precision: only the shared-tap 3x3 kernel has a 128-wide tile (the plain kernel would need scratch memory there);
        // the 1x1 / stride-2 / concat convolutions run on the 64-wide tile, which also keeps twice the workgroups in flight per
        // byte streamed for the HBM-latency-bound 76x76 head layers (measured at config 4: 0.83 -> 0.56, 0.77 -> 0.67 ms)
        if (h->precision == 1) tile = conv_split_tile(tile, s.kx3 || s.p1);
        else if (inject && tile == TILE_128x128) tile = TILE_128x64;      // the fp32 128-wide build has no mask-injection path (conv_igemm.hip)
        // shared-tap 3x3 with cout % 256 == 0 and enough rows to fill the chip: ONE 8-wave workgroup owns all 256 output channels of
        // its 128 pixels, so an activation row is fetched and staged once per 256 columns instead of once per 128.  Measured at
        // config 4 (round 4, gpurun_out/r4b_*): the three 76x76 head convolutions 2.13 -> 2.22 ms each (-4 %): with ONE workgroup per
        // CU the epilogues of all eight waves coincide and nothing multiplies meanwhile, where two independent 4-wave workgroups
        // overlap one's epilogue with the other's K loop -- so it is NOT the default.
        // opts.kx3_wide: 0 never (default), 1 launches of >= 4 rounds of 256 workgroups, 2 every eligible launch (tests)
        const int kx3_wide = h->opts.kx3_wide;
        if (h->precision == 1 && s.kx3 && (s.Npad % 256) == 0 && kx3_wide && (l.filters % 128) == 0 &&
            (kx3_wide >= 2 ? (tile == TILE_128x128 || tile == TILE_128x64)       // (forced: also where the grid-fill rule above went narrow)
                           : (tile == TILE_128x128 && (int64_t)((M + 127) / 128) * (s.Npad / 256) >= 4 * 256)))
            tile = TILE_128x256;
        if (p.fuse[si]) tile = TILE_128x256;                  // (decided before the arena was laid out, above)
        p.tile[si] = tile;
        // split precision: a K-tile takes ~0.4 of the fp32 kernel's; a shared-tap launch is scheduled in stages of 3 K-tiles
        const bool sp = h->precision == 1, kx3 = sp && s.kx3;
        p.split[si] = conv_plan_split(M, s.Npad, kx3 ? KT / 3 : KT, tile, sp ? (kx3 ? 1.2 : 0.4) : 1.0, h->opts.ksplit, h->opts.streamk);
        if (s.low) p.split[si] = ConvSplit{((M + 127) / 128) * (s.Npad / conv_tile_bn(tile)), 0, 0, 1};      // whole tiles: the accumulation order of a STEP_MAIN tile
        if (tile == TILE_128x256) p.split[si] = ConvSplit{((M + 127) / 128) * (s.Npad / 256), 0, 0, 1};      // whole tiles only: its workgroups walk the tile list (conv_igemm.hip WALK); a follower needs a finished tile
        slab = std::max(slab, conv_split_slab_bytes(p.split[si], tile));
    }
    // Winograd F(2x2,3x3) for the large 3x3 / stride-1 convolutions (winograd.hip): samples per chunk such that
    // the transformed input V (4x the input) and the GEMM result M (4x the output) of a chunk fit the scratch.
    // BYOLO_WINOGRAD=0 keeps every convolution direct.
    p.wino.assign(h->steps.size(), WinoPlan{});
    size_t wino_scratch = 0;
    // Split precision: Winograd F(2x2,3x3) in split arithmetic (wino_split.hip) for the 3x3 / stride-1 convolutions it is faster on.
    // opts.wino_split: 0 never, 1 by the time model below (default), 2 every eligible layer (tests); opts.wino_split_bn: 0 the model
    // picks 256 or 128 output channels per workgroup (default), 256 | 128 forced; opts.wino_split_min_gflop: a floor under the model;
    // opts.wino_split_chunk_mb: V bytes of a chunk.
    //
    // The model (milliseconds; measured round 6 on 1024 x 1920 frames at batch 1 .. 11 and at config 4, profiles/r6_wino_small.md):
    // a launch is `units` = row tiles (64 output tiles) x column tiles (bn channels) on 256 CUs; a 256-channel unit owns its CU
    // (8 waves), two 128-channel units share one.  A round of 256-channel units takes a256(C); 128-channel units take s128(C) while
    // each has a CU to itself (<= 256 of them) and b128(C) when two share; beyond two full rounds the 128-channel form pays ~9 %
    // (measured 1.29 against 1.19 ms at config 4's 10.6 rounds).  The transform writes 4 x the input.  The direct shared-tap kernel
    // runs at 390 - 425 TFLOP/s of algorithmic FLOPs behind ~0.03 ms.  A layer is transformed when the model says >= 3 % faster.
    if (h->precision == 1) {
        const int on = h->opts.wino_split;
        const double min_flops = on >= 2 ? 0.0 : (double)h->opts.wino_split_min_gflop * 1e9, budget = (double)h->opts.wino_split_chunk_mb * 1e6;
        const int bm = 64;
        const int bn_pref = h->opts.wino_split_bn;
        auto model_ms = [](int64_t row_tiles, int N, int C, int bn) -> double {
            if (bn == 256) return (double)((row_tiles * (N / 256) + 255) / 256) * (0.030 + 0.000293 * C);
            const int64_t units = row_tiles * (N / 128), full = units / 512, rem = units % 512;
            const double t = (double)full * (0.012 + 0.000367 * C) + (rem == 0 ? 0.0 : rem <= 256 ? 0.024 + 0.00022 * C : 0.012 + 0.000367 * C);
            return units > 1024 ? 1.09 * t : t;
        };
        for (size_t si = 0; on && si < h->steps.size(); ++si) {
            const Step& s = h->steps[si];
            const Layer& l = h->layers[s.layer];
            if (!s.wino_ok || !s.kx3 || s.mode != STEP_NORMAL || l.wshift_u.empty() || (l.fused_residual >= 0 && l.drop_ordinal >= 0) || p.fuse[si]) continue;   // (the kernel's residual epilogue carries no dropout)
            int M, KT; step_geometry(h, s, B, T, &M, &KT);
            const double flops = 2.0 * M * l.filters * 9.0 * l.Cin;
            if (flops < min_flops) continue;
            // per transform point the K loop is only Cin / 32 tiles long, and the fold + the 5x input stream are paid per point:
            // measured at config 4 (direct -> transform + fused): Cin 512 2.00 -> 0.19 + 1.36 ms, 256 2.02 -> 0.35 + 1.37,
            // 128 2.15 -> 2 x (0.36 + 0.80) -- the 128-channel layers stay direct (BYOLO_WINO_SPLIT_MIN_C)
            if (on < 2 && l.Cin < h->opts.wino_split_min_c) continue;
            const int th = (l.H + 1) / 2, tw = (l.W + 1) / 2, S = M / (l.H * l.W);
            const double per_sample = 16.0 * th * tw * l.Cin * 4.0;
            const int nchunks = std::max(1, (int)std::ceil(S * per_sample / budget));          // equal chunks
            const int chunk = (S + nchunks - 1) / nchunks;
            const int64_t row_tiles = (int64_t)align_up((size_t)chunk * th * tw, 128) / bm;
            const bool can256 = (l.filters % 256) == 0;
            int bn = (bn_pref == 256 && can256) ? 256 : bn_pref == 128 ? 128 : 0;
            if (bn_pref == 256 && !can256) bn = 128;
            const double t256 = can256 ? model_ms(row_tiles, l.filters, l.Cin, 256) : 1e30, t128 = model_ms(row_tiles, l.filters, l.Cin, 128);
            if (bn == 0) bn = t256 <= t128 ? 256 : 128;
            if (on < 2) {
                const double t_res = l.fused_residual >= 0 ? 4.0 * M * l.filters / 2.5e9 : 0.0;      // the residual epilogue's blocking reads (wino_split.hip epilogue_res)
                const double t_wino = nchunks * ((bn == 256 ? t256 : t128) + 0.012 + (double)row_tiles * bm * per_sample / ((double)th * tw) / 4.5e9) + t_res;
                const double t_direct = 0.028 + flops / ((l.Cin >= 512 ? 390.0 : 425.0) * 1e9);
                if (t_wino > 0.97 * t_direct) continue;
            }
            WinoPlan& w = p.wino[si];
            w.th = th; w.tw = tw; w.bm = bm; w.fused = true;
            w.bn = bn;
            w.chunk = chunk;
            // BYOLO_WINO_SPLIT_ROUNDS=k (experiment): chunks whose fused launch is k whole rounds of resident workgroups, so that a
            // chunk's V (<= ~140 MB per round) is still in the Infinity Cache when the GEMM reads it
            const int rounds = h->opts.wino_split_rounds;
            if (rounds > 0) {
                const int slots = (w.bn == 256 ? 256 : 512), n_tiles = l.filters / w.bn;
                const int64_t row_tiles = (int64_t)rounds * slots / n_tiles;                // of w.bm rows each
                const int tt = w.th * w.tw;
                int c = (int)((row_tiles * w.bm) / tt);                                     // whole samples that fit
                while (c > 1 && (int64_t)align_up((size_t)c * tt, 128) / w.bm * n_tiles > (int64_t)rounds * slots) --c;
                w.chunk = std::max(1, std::min(S, c));
            }
            const size_t P_pad = align_up((size_t)w.chunk * w.th * w.tw, 128);
            w.v_bytes = align_up((size_t)16 * P_pad * l.Cin * 4, 256);
            if (w.v_bytes > CONV_MAX_SRC_BYTES) { w = WinoPlan{}; continue; }                 // 32-bit buffer offsets
            w.m_bytes = 0;
            wino_scratch = std::max(wino_scratch, w.v_bytes);
        }
    }
    { const int on = (h->precision == 1 || inject) ? 0 : h->opts.winograd;   // split precision: direct convolutions only (memory-bound transforms do not pay there); injected masks: conv_igemm's epilogue reads them
      // opts.wino_min_gflop: tuning knob: smallest layer (direct GFLOP) to transform
      // (measured at config 4: 100 -> 144.97, 20 -> 147.35, 5 -> 147.32 img/s; at config 2 (416x416, 8 images) the 52x52
      //  layers are 12.8 GFLOP: 20 -> 1375, 10 -> 1506, 5 -> 1504 img/s.  Default 10.)
      // opts.wino_chunk_mb: tuning knob: V + M bytes of one chunk
      // (chunk budget measured at config 4: 2600 MB 177.6, 600 MB 179.4, 300 MB 150.6 img/s -- below ~500 MB the fused
      //  kernel's slots run out of row tiles; 800 MB keeps the scratch small without costing rounds)
      const double min_flops = on >= 2 ? 0.0 : (double)h->opts.wino_min_gflop * 1e9, budget = (double)h->opts.wino_chunk_mb * 1e6;   // on == 2: every eligible layer (tests)
      for (size_t si = 0; on && si < h->steps.size(); ++si) {
        const Step& s = h->steps[si];
        const Layer& l = h->layers[s.layer];
        if (!s.wino_ok) continue;
        int M, KT; step_geometry(h, s, B, T, &M, &KT);
        if (2.0 * M * l.filters * 9.0 * l.Cin < min_flops) continue;
        // The transforms stream 4x the input + 4x the output through HBM (measured 5.2 TB/s); per output pixel the GEMM
        // saves 5/9 of 2*9*Cin*cout FLOPs.  That pays when Cin*cout/(Cin+cout) is large: measured at config 4
        // 512x1024 channels (19x19) -33 %, 256x512 (38x38) -24 %, 128x256 (76x76) +6 % -> direct below ~128.
        const double min_ratio = (double)h->opts.wino_min_ratio;
        if (on < 2 && (double)l.Cin * l.filters / (l.Cin + l.filters) < min_ratio) continue;
        WinoPlan& w = p.wino[si];
        w.th = (l.H + 1) / 2; w.tw = (l.W + 1) / 2;
        const int S = M / (l.H * l.W);
        const double per_sample = 16.0 * w.th * w.tw * (l.Cin + l.filters) * 4.0;
        const int nchunks = (int)std::ceil(S * per_sample / budget);                 // equal chunks
        w.chunk = (S + nchunks - 1) / nchunks;
        // Fused kernel (wino_fused.hip; no M): its work unit is a row tile of 128 output tiles through all 16 transform
        // points, dealt out statically to 512 / (cout/64) slots -- pick the chunk size (samples) whose row-tile count
        // wastes the fewest slot rounds, and use the fused kernel only when every slot gets >= 3 row tiles.
        const int fused_mode = h->opts.wino_fused;                             // 0 never, 2 always (tests)
        if (fused_mode && wino_fused_ok(l.Cin, l.filters)) {
            const int slots = 512 / (l.filters / 64), tt = w.th * w.tw;
            const int max_c = (int)std::max(1.0, std::min((double)S, std::floor(budget / (16.0 * tt * l.Cin * 4.0))));
            auto rt = [&](int c) { return (c * tt + 127) / 128; };
            auto rounds = [&](int c) { return (rt(c) + slots - 1) / slots; };
            int best_c = 0; double best_cost = 1e30;
            for (int c = std::max(1, max_c / 6); c <= max_c; ++c) {
                const int full = S / c, last = S % c;
                // cost in slot rounds (+ a little per chunk for the launches and the pipeline fill)
                const double cost = full * (rounds(c) + 0.15) + (last ? rounds(last) + 0.15 : 0.0);
                if (cost < best_cost - 1e-9 || (std::fabs(cost - best_cost) < 1e-9 && c > best_c)) { best_cost = cost; best_c = c; }
            }
            if (best_c > 0 && (fused_mode >= 2 || rt(best_c) / slots >= 3)) { w.fused = true; w.chunk = best_c; }
        }
        const size_t P_pad = align_up((size_t)w.chunk * w.th * w.tw, 128);
        w.v_bytes = align_up((size_t)16 * P_pad * l.Cin * 4, 256);
        w.m_bytes = w.fused ? 0 : align_up((size_t)16 * P_pad * l.filters * 4, 256);
        wino_scratch = std::max(wino_scratch, w.v_bytes + w.m_bytes);
        const int rows = (int)(16 * P_pad);
        p.split[si] = conv_plan_split(rows, s.Npad, l.Cin / 32, s.tile, 1.0, h->opts.ksplit, h->opts.streamk);
        p.tile[si] = s.tile;
        slab = std::max(slab, conv_split_slab_bytes(p.split[si], s.tile));
      }
    }
    // Row-streaming launch for the 1x1 / stride-1 convolutions over one plain source (gemm_stream.hip): the 1x1
    // convolutions of the heads, the concat convolutions' stacked half (STEP_MAIN) and the detection heads.
    // opts.stream1x1 = 0 keeps them on conv_igemm (A/B), 2 takes it for every shape the kernel can express (tests).
    p.stream1x1.assign(h->steps.size(), 0);
    { const bool on = h->precision == 0 && !inject && h->opts.stream1x1 != 0, force = h->opts.stream1x1 >= 2;
      for (size_t si = 0; on && si < h->steps.size(); ++si) {
        const Step& s = h->steps[si];
        const Layer& l = h->layers[s.layer];
        if (!s.is_conv() || l.direct || l.ksize != 1 || l.stride != 1) continue;
        if (s.mode != STEP_NORMAL && s.mode != STEP_MAIN) continue;
        if (s.in.n != 1 || s.in.s[0].sh || s.in.s[0].tile || s.in.s[0].layer < 0 || l.fused_residual >= 0) continue;
        if (l.op == OP_CONV && (l.filters % 4)) continue;
        int M, KT; step_geometry(h, s, B, T, &M, &KT);
        const int bn = conv1x1_stream_tile(M, s.c_hi - s.c_lo, l.filters, force);
        if (bn && (l.filters <= 64 ? s.Npad == 64 : s.Npad == l.filters)) p.stream1x1[si] = bn;
      }
    }
    p.wino_off = o; o += align_up(wino_scratch, 256);
    p.slab_off = o; p.slab_bytes = slab; o += align_up(slab, 256);
    p.cnt_bytes = h->steps.size() * CNT_PER_STEP * sizeof(unsigned);
    p.cnt_off = o; o += align_up(p.cnt_bytes, 256);
    p.total = o;
}

extern "C" int32_t byolo_num_layers(const byolo_t* h) { return h ? (int32_t)h->layers.size() : BYOLO_ERR_ARG; }

extern "C" int32_t byolo_num_boxes(const byolo_t* h, int64_t* n, int32_t* d) {
    if (!h) return BYOLO_ERR_ARG;
    if (n) *n = h->n_boxes;
    if (d) *d = h->row_len;
    return BYOLO_OK;
}

// images one launch sequence may carry at this T (byolo_max_images): 32-bit source offsets and pixel counts
int64_t piece_cap(const byolo_t* h, int32_t T) {
    uint64_t per_image = (uint64_t)h->cfg.img_h * h->cfg.img_w * h->cfg.img_c * 4;      // bytes per image of the largest tensor
    int64_t rows = 0;                                                                  // pixels per image of the largest layer
    for (const auto& l : h->layers) {
        const uint64_t s = l.stacked ? (uint64_t)T : 1;
        if (l.materialized) per_image = std::max(per_image, s * l.H * l.W * l.C * 4);
        rows = std::max<int64_t>(rows, (int64_t)s * l.H * l.W);
    }
    const uint64_t by_bytes = CONV_MAX_SRC_BYTES / per_image, by_rows = (((uint64_t)1 << 31) - 1) / (uint64_t)rows;
    return (int64_t)std::min<uint64_t>(std::min(by_bytes, by_rows), 1 << 20);
}

int32_t check_run(byolo_t* h, int32_t B, int32_t T, const char* what, bool need_device) {
    if (!h) return fail(nullptr, BYOLO_ERR_ARG, "%s: null handle", what);
    if (need_device && !h->finalized) return fail(h, BYOLO_ERR_STATE, "%s: call byolo_finalize first", what);
    if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }
    if (B < 1 || T < 1) return fail(h, BYOLO_ERR_ARG, "%s: B and T must be >= 1", what);
    for (const auto& l : h->layers) {
        const int64_t S = l.stacked ? (int64_t)B * T : B;
        if (S * l.H * l.W >= (int64_t)1 << 31) return fail(h, BYOLO_ERR_ARG, "%s: B*T*h*w exceeds 2^31 pixels", what);
        // the convolution addresses its sources with 32-bit byte offsets (buffer loads)
        if (l.materialized && (uint64_t)S * l.H * l.W * l.C * 4 > CONV_MAX_SRC_BYTES)
            return fail(h, BYOLO_ERR_ARG, "%s: a [%lld,%d,%d,%d] activation exceeds the 3 GiB a convolution source may span; "
                        "split the call into smaller image batches", what, (long long)S, l.H, l.W, l.C);
    }
    if ((uint64_t)B * h->cfg.img_h * h->cfg.img_w * h->cfg.img_c * 4 > CONV_MAX_SRC_BYTES)
        return fail(h, BYOLO_ERR_ARG, "%s: the image batch exceeds 3 GiB; split the call", what);
    return BYOLO_OK;
}

static int32_t workspace_bytes_impl(byolo_t* h, int32_t B, int32_t T, size_t* out) {
    if (h && B >= 1 && T >= 1) {                       // a batch beyond byolo_max_images runs in pieces (byolo_forward): the largest piece's arena
        if (!h->lowered) { int32_t rc = lower(h); if (rc) return rc; }
        const int64_t cap = piece_cap(h, T);
        if (cap >= 1 && B > cap) {
            if (!out) return fail(h, BYOLO_ERR_ARG, "byolo_workspace_bytes: null out");
            size_t a = 0, b = 0;
            int32_t rc = byolo_workspace_bytes(h, (int32_t)cap, T, &a); if (rc) return rc;
            if (B % cap) { rc = byolo_workspace_bytes(h, (int32_t)(B % cap), T, &b); if (rc) return rc; }
            *out = std::max(a, b);
            return BYOLO_OK;
        }
    }
    int32_t rc = check_run(h, B, T, "byolo_workspace_bytes", false); if (rc) return rc;
    if (!out) return fail(h, BYOLO_ERR_ARG, "byolo_workspace_bytes: null out");
    // the plan of a call with injected dropout masks (byolo_forward's d_mask_bits) differs in the fp32 mode (64-wide tiles, other
    // split-K slabs, no Winograd): the size returned covers BOTH, so a workspace sized here never fails either kind of call
    // (the plan in effect stays the unmasked one, made ONCE per (B, T): a caller asks for the size before every forward)
    make_plan(h, B, T, false);
    if (h->wsm_B != B || h->wsm_T != T || h->wsm_epoch != h->plan_epoch) {
        const Plan keep = h->plan;
        h->plan.B = -1;
        make_plan(h, B, T, true);
        h->wsm_total = h->plan.total; h->wsm_B = B; h->wsm_T = T; h->wsm_epoch = h->plan_epoch;
        h->plan = keep; h->plan_inject = false;
    }
    *out = std::max(h->plan.total, h->wsm_total);
    return BYOLO_OK;
}
extern "C" int32_t byolo_workspace_bytes(byolo_t* h, int32_t B, int32_t T, size_t* out) {
    return guarded(h, "byolo_workspace_bytes", [&] { return workspace_bytes_impl(h, B, T, out); });
}


// ---- introspection of the plan (tests/test_planner.py; host only, no device needed) ----------------------------------------
// byolo_plan_num makes the (B, T) plan -- `inject` != 0: the plan of a call with injected dropout masks -- and says how many steps
// and tensors it has; byolo_plan_step: the tensor a step writes, whether the NEXT step runs inside its launch (back-to-back
// fusion: that step's output is then written during THIS step and this step's own output never exists), and the tensors the step
// reads, taken from the step's operand description -- NOT from the liveness table the planner itself releases by, so that the test
// checks the planner against the launches rather than against its own bookkeeping; byolo_plan_tensor: where a tensor lives
// (offset < 0: it has no memory in this plan) and whether something outside the step list reads it afterwards (a detection
// layer's raw output: the decode launch, byolo_layer_output).
extern "C" int32_t byolo_plan_num(byolo_t* h, int32_t B, int32_t T, int32_t inject, int32_t* n_steps, int32_t* n_tensors, int64_t* arena_bytes) {
    return guarded(h, "byolo_plan_num", [&]() -> int32_t {
        int32_t rc = check_run(h, B, T, "byolo_plan_num", false); if (rc) return rc;
        h->plan.B = -1;                                                  // (always a fresh plan: the environment knobs may have changed)
        make_plan(h, B, T, inject != 0);
        if (n_steps) *n_steps = (int32_t)h->steps.size();
        if (n_tensors) *n_tensors = (int32_t)(h->layers.size() + h->aux.size());
        if (arena_bytes) *arena_bytes = (int64_t)h->plan.arena;
        return BYOLO_OK;
    });
}
extern "C" int32_t byolo_plan_step(byolo_t* h, int32_t step, int32_t* out_tensor, int32_t* fuses_next, int32_t reads[8], int32_t* n_reads) {
    if (!h || h->plan.B < 0 || step < 0 || step >= (int32_t)h->steps.size()) return fail(h, BYOLO_ERR_ARG, "byolo_plan_step: no plan (byolo_plan_num first) or bad step");
    const Step& st = h->steps[step];
    const Layer& l = h->layers[st.layer];
    int n = 0;
    for (int k = 0; k < st.in.n; ++k) if (st.in.s[k].layer >= 0) reads[n++] = st.in.s[k].layer;      // (< 0: the image)
    if (st.addend_tensor >= 0) reads[n++] = st.addend_tensor;
    if (st.low_tensor >= 0) reads[n++] = st.low_tensor;
    if (st.is_conv() && st.mode != STEP_PARTIAL && l.fused_residual >= 0) reads[n++] = h->layers[l.fused_residual].ref[0];
    if (out_tensor) *out_tensor = st.out_tensor;
    if (fuses_next) *fuses_next = (step < (int32_t)h->plan.fuse.size() && h->plan.fuse[step]) ? 1 : 0;
    if (n_reads) *n_reads = n;
    return BYOLO_OK;
}
extern "C" int32_t byolo_plan_tensor(byolo_t* h, int32_t tensor, int64_t* offset, int64_t* bytes, int32_t* read_after_the_steps) {
    const int32_t nt = h ? (int32_t)(h->layers.size() + h->aux.size()) : 0;
    if (!h || h->plan.B < 0 || tensor < 0 || tensor >= nt) return fail(h, BYOLO_ERR_ARG, "byolo_plan_tensor: no plan (byolo_plan_num first) or bad tensor");
    if (offset) *offset = h->plan.off[tensor];
    if (bytes) *bytes = tensor_bytes(h, tensor, h->plan.B, h->plan.T);
    if (read_after_the_steps) *read_after_the_steps = (tensor < (int32_t)h->layers.size() && (h->layers[tensor].op == OP_DETECTION || h->cfg.keep_all_outputs)) ? 1 : 0;
    return BYOLO_OK;
}
