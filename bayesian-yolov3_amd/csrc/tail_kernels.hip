// tail_kernels.hip -- anchor decode (K6), MC-sample aggregation + decode (K7), score sort (K8) and
// greedy NMS + gather (K9) of SURVEY.md section 2.1, as wavefront-level gfx950 kernels.
//
//   decode_std / decode_ale : lib_yolo/layers.py:11-84 (split) + :191-346 (decode) + :349-358 (entropies)
//   decode_epi              : lib_yolo/layers.py:361-411 (T-reduction) + :414-502 (decode)
//   all three write rows straight at their concat_bbox position (inference_epistemic.py:173-184,
//   inference_aleatoric.py:181-192): n = base(layer) + prior*lh*lw + row*lw + col
//   sort_keys + nms         : tf.image.non_max_suppression(boxes[:, :4], boxes[:, obj_idx], 1000) + tf.gather
//                             (inference_epistemic.py:99-128 incl. the commented 2-class variant)
// HBM-bound / dependency-bound byte work: no MFMA here.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <float.h>
#include "byolo_kernels.h"

namespace byk {

// ------------------------------------------------------------------------------------------------
// element-wise maths (IEEE semantics kept: 0*log(0) = NaN exactly like the reference, App. D.2)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float logistic_entropy_(float s) {          // layers.py:349-353
    const float no_obj = (1.0f - s) * logf(1.0f - s);
    const float obj = s * logf(s);
    return -(no_obj + obj);
}
// Class counts: the kernels are compiled for a capacity CM of class slots held in registers.  EXACT: the class count
// IS CM (the common counts: every loop bound is a constant); otherwise the runtime count C <= CM masks the slots.
template <int CM, bool EXACT>
__device__ __forceinline__ void softmax_(const float* x, float* p, int C) {   // tf.nn.softmax (max-subtracted)
    float mx = x[0];
#pragma unroll
    for (int c = 1; c < CM; ++c) if (EXACT || c < C) mx = fmaxf(mx, x[c]);
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (EXACT || c < C) { p[c] = expf(x[c] - mx); sum += p[c]; }
#pragma unroll
    for (int c = 0; c < CM; ++c) if (EXACT || c < C) p[c] = p[c] / sum;
}
template <int CM, bool EXACT>
__device__ __forceinline__ float softmax_entropy_(const float* p, int C) {    // layers.py:356-358
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CM; ++c) if (EXACT || c < C) s += p[c] * logf(p[c]);
    return -s;
}
__device__ __forceinline__ void corners_(float tx, float ty, float tw, float th, int col, int row, int lw, int lh,
                                         float pw, float ph, float* o /*y0,x0,y1,x1*/) {
    // layers.py:237-249 / :316-328 / :471-483
    const float x = ((float)col + sigmoidf_(tx)) / (float)lw;
    const float y = ((float)row + sigmoidf_(ty)) / (float)lh;
    const float w = expf(tw) * pw, h = expf(th) * ph;
    const float w2 = w / 2, h2 = h / 2;
    o[0] = y - h2; o[1] = x - w2; o[2] = y + h2; o[3] = x + w2;
}

// thread <-> (image b, cell, prior p), prior fastest: a wave reads 64 * blk contiguous floats.
template <int CM, bool EXACT>
__global__ __launch_bounds__(256) void decode_std_kernel(const DecodeParams p) {
    const int C = EXACT ? CM : p.C;
    const int BLK = 5 + C, D = 5 + C;
    const int LDC = p.ld ? p.ld : 3 * BLK;
    const int cells = p.lh * p.lw;
    const int64_t total = (int64_t)p.B * cells * 3;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(gid % 3);
        const int64_t bc = gid / 3;
        const int cell = (int)(bc % cells), b = (int)(bc / cells);
        const float* d = p.raw + (size_t)bc * LDC + pr * BLK;
        float v[5 + CM];
        float chk = 0.f;                                        // stays 0 unless a raw value is inf / NaN (x * 0 is NaN then)
#pragma unroll
        for (int i = 0; i < 5 + CM; ++i) if (EXACT || i < BLK) { v[i] = d[i]; chk = fmaf(v[i], 0.f, chk); }
        if (chk != 0.f && p.status) atomicOr(p.status, 2u);
        float out[5 + CM];
        corners_(v[0], v[1], v[2], v[3], cell % p.lw, cell / p.lw, p.lw, p.lh, p.pw[pr], p.ph[pr], out);
        out[4] = sigmoidf_(v[4]);
        softmax_<CM, EXACT>(v + 5, out + 5, C);
        float* o = p.boxes + ((size_t)b * p.n_total + p.box_base + (size_t)pr * cells + cell) * D;
#pragma unroll
        for (int i = 0; i < 5 + CM; ++i) if (EXACT || i < D) o[i] = out[i];
    }
}

template <int CM, bool EXACT>
__global__ __launch_bounds__(256) void decode_ale_kernel(const DecodeParams p) {
    const int C = EXACT ? CM : p.C;
    const int BLK = 2 * (5 + C), D = 14 + C;
    const int LDC = p.ld ? p.ld : 3 * BLK;
    const int cells = p.lh * p.lw;
    const int64_t total = (int64_t)p.B * cells * 3;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(gid % 3);
        const int64_t bc = gid / 3;
        const int cell = (int)(bc % cells), b = (int)(bc / cells);
        const float* d = p.raw + (size_t)bc * LDC + pr * BLK;
        // [x,y,w,h, logvar x4, obj, log_obj_std, cls xC, log_cls_std xC]   (layers.py:41-84); the stds are not decoded
        float v[10 + CM];
        float chk = 0.f;
#pragma unroll
        for (int i = 0; i < 10 + CM; ++i) if (EXACT || i < 10 + C) { v[i] = d[i]; chk = fmaf(v[i], 0.f, chk); }
        if (chk != 0.f && p.status) atomicOr(p.status, 2u);
        float out[11 + CM];
        corners_(v[0], v[1], v[2], v[3], cell % p.lw, cell / p.lw, p.lw, p.lh, p.pw[pr], p.ph[pr], out);
        float prod = 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { out[4 + i] = expf(v[4 + i]); }
        prod = ((out[4] * out[5]) * out[6]) * out[7];                   // tf.reduce_prod
        out[8] = prod;
        const float obj = sigmoidf_(v[8]);
        out[9] = obj;
        out[10] = logistic_entropy_(obj);
        softmax_<CM, EXACT>(v + 10, out + 11, C);
        const float clsH = softmax_entropy_<CM, EXACT>(out + 11, C);
        float* o = p.boxes + ((size_t)b * p.n_total + p.box_base + (size_t)pr * cells + cell) * D;
#pragma unroll
        for (int i = 0; i < 11 + CM; ++i) if (EXACT || i < 11 + C) o[i] = out[i];
        o[11 + C] = clsH;
        o[12 + C] = (float)p.layer_id;
        o[13 + C] = (float)pr;
    }
}

// 4x4 determinant by LU with partial pivoting (what tf.linalg.det does via Eigen PartialPivLU).
// The pivot row is swapped in with SELECTS over compile-time indices: a run-time row index (a[piv][c]) sends the matrix
// to scratch memory (20 bytes of private segment per lane in the first version), and this library's kernels stay out
// of scratch altogether -- kernels with private segments from several HIP streams at once disturbed each other's
// spilled values on this stack (tests/test_gpu_parity.py::test_engines_on_concurrent_streams: this determinant, whose
// value is rounding noise for T <= 4 samples, was the one visible symptom).
__device__ __forceinline__ float det4_(float a[4][4]) {
    float det = 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int piv = k; float best = fabsf(a[k][k]);
#pragma unroll
        for (int r = k + 1; r < 4; ++r) { const float v = fabsf(a[r][k]); if (v > best) { best = v; piv = r; } }
#pragma unroll
        for (int r = k + 1; r < 4; ++r) {
            const bool sw = piv == r;                              // (same first-maximum choice, same swap as before)
#pragma unroll
            for (int c = 0; c < 4; ++c) { const float x = a[k][c], y = a[r][c]; a[k][c] = sw ? y : x; a[r][c] = sw ? x : y; }
        }
        if (piv != k) det = -det;
        const float d = a[k][k];
        det *= d;
        if (d != 0.f) {
#pragma unroll
            for (int r = k + 1; r < 4; ++r) {
                const float f = a[r][k] / d;
#pragma unroll
                for (int c = k + 1; c < 4; ++c) a[r][c] -= f * a[k][c];
            }
        }
    }
    return det;
}

// One lane per (image, cell, prior): a single pass over the image's T samples keeps
// 4 + 10 + 4 + 1 + 1 + C + 1 running sums in registers.
template <int CM, bool EXACT>
__global__ __launch_bounds__(256) void decode_epi_kernel(const DecodeParams p) {
    const int C = EXACT ? CM : p.C;
    const int BLK = 2 * (5 + C), D = 21 + C;
    const int LDC = p.ld ? p.ld : 3 * BLK;
    const int cells = p.lh * p.lw;
    const int64_t total = (int64_t)p.B * cells * 3;
    const size_t sample_stride = (size_t)cells * LDC;
    const float invT = 1.0f / (float)p.T;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(gid % 3);
        const int64_t bc = gid / 3;
        const int cell = (int)(bc % cells), b = (int)(bc / cells);
        const float* d0 = p.raw + ((size_t)b * p.T) * sample_stride + (size_t)cell * LDC + pr * BLK;
        float s_loc[4] = {0, 0, 0, 0}, s_ll[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s_var[4] = {0, 0, 0, 0};
        float s_obj = 0.f, s_objH = 0.f, s_cls[CM], s_clsH = 0.f;
        float chk = 0.f;                                        // stays 0 unless a raw value is inf / NaN
#pragma unroll
        for (int c = 0; c < CM; ++c) s_cls[c] = 0.f;
        float* o = p.boxes + ((size_t)b * p.n_total + p.box_base + (size_t)pr * cells + cell) * D;
        if (p.mode == 2) {                                      // the sums of ALL samples, added up across the ranks (byolo_finish_tshard)
#pragma unroll
            for (int i = 0; i < 4; ++i) { s_loc[i] = o[i]; s_var[i] = o[14 + i]; }
#pragma unroll
            for (int i = 0; i < 10; ++i) s_ll[i] = o[4 + i];
            s_obj = o[18]; s_objH = o[19];
#pragma unroll
            for (int c = 0; c < CM; ++c) if (EXACT || c < C) s_cls[c] = o[20 + c];
            s_clsH = o[20 + C];
        }
        for (int t = 0; t < (p.mode == 2 ? 0 : p.T); ++t) {
            const float* d = d0 + (size_t)t * sample_stride;
            float v[10 + CM];                                   // the two std logit groups are not decoded
#pragma unroll
            for (int i = 0; i < 10 + CM; ++i) if (EXACT || i < 10 + C) { v[i] = d[i]; chk = fmaf(v[i], 0.f, chk); }
            int q = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_loc[i] += v[i];
                s_var[i] += expf(v[4 + i]);
#pragma unroll
                for (int j = i; j < 4; ++j) s_ll[q++] += v[i] * v[j];
            }
            const float obj = sigmoidf_(v[8]);
            s_obj += obj;
            s_objH += logistic_entropy_(obj);
            float pc[CM];
            softmax_<CM, EXACT>(v + 10, pc, C);
#pragma unroll
            for (int c = 0; c < CM; ++c) if (EXACT || c < C) s_cls[c] += pc[c];
            s_clsH += softmax_entropy_<CM, EXACT>(pc, C);
        }
        if (chk != 0.f && p.status) atomicOr(p.status, 2u);
        if (p.mode == 1) {                                      // this rank's share of the T samples: hand out the sums
#pragma unroll
            for (int i = 0; i < 4; ++i) { o[i] = s_loc[i]; o[14 + i] = s_var[i]; }
#pragma unroll
            for (int i = 0; i < 10; ++i) o[4 + i] = s_ll[i];
            o[18] = s_obj; o[19] = s_objH;
#pragma unroll
            for (int c = 0; c < CM; ++c) if (EXACT || c < C) o[20 + c] = s_cls[c];
            o[20 + C] = s_clsH;
            continue;
        }
        float ev[4], cov[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ev[i] = s_loc[i] * invT;
        {
            int q = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i; j < 4; ++j) {
                    const float c = s_ll[q++] * invT - ev[i] * ev[j];   // E[l l^T] - E[l]E[l]^T (layers.py:383)
                    cov[i][j] = c; cov[j][i] = c;
                }
        }
        float out[17 + CM];
        corners_(ev[0], ev[1], ev[2], ev[3], cell % p.lw, cell / p.lw, p.lw, p.lh, p.pw[pr], p.ph[pr], out);
        float ale_sum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[4 + i] = cov[i][i];
            const float a = s_var[i] * invT;
            out[8 + i] = a;
            ale_sum += a;
        }
        out[12] = det4_(cov);
        out[13] = ale_sum;
        const float obj_mean = s_obj * invT;
        const float objH = logistic_entropy_(obj_mean);
        out[14] = obj_mean;
        out[15] = objH - s_objH * invT;
        out[16] = objH;
#pragma unroll
        for (int c = 0; c < CM; ++c) if (EXACT || c < C) out[17 + c] = s_cls[c] * invT;
        const float clsH = softmax_entropy_<CM, EXACT>(out + 17, C);
#pragma unroll
        for (int i = 0; i < 17 + CM; ++i) if (EXACT || i < 17 + C) o[i] = out[i];
        o[17 + C] = clsH - s_clsH * invT;
        o[18 + C] = clsH;
        o[19 + C] = (float)p.layer_id;
        o[20 + C] = (float)pr;
    }
}

// The entries of decode_epistemic's dict (lib_yolo/layers.py:397-411) that are not columns of the box row: the mean
// raw location logits `ev_loc`, the full 4x4 `epi_covar_loc` (same one-pass sums as decode_epi_kernel: its diagonal
// equals the row's columns 4..7 bit for bit) and the per-sample `obj_samples` / `cls_samples`.  One lane per
// (image, cell, prior); any output may be null.
template <int CM, bool EXACT>
__global__ __launch_bounds__(256) void epi_stats_kernel(const DecodeParams p, float* ev_loc, float* covar, float* obj_s, float* cls_s) {
    const int C = EXACT ? CM : p.C;
    const int BLK = 2 * (5 + C);
    const int LDC = p.ld ? p.ld : 3 * BLK;
    const int cells = p.lh * p.lw;
    const int64_t total = (int64_t)p.B * cells * 3;
    const size_t sample_stride = (size_t)cells * LDC;
    const float invT = 1.0f / (float)p.T;
    for (int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (int64_t)gridDim.x * blockDim.x) {
        const int pr = (int)(gid % 3);
        const int64_t bc = gid / 3;
        const int cell = (int)(bc % cells), b = (int)(bc / cells);
        const float* d0 = p.raw + ((size_t)b * p.T) * sample_stride + (size_t)cell * LDC + pr * BLK;
        float s_loc[4] = {0, 0, 0, 0}, s_ll[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int t = 0; t < p.T; ++t) {
            const float* d = d0 + (size_t)t * sample_stride;
            float v[10 + CM];
#pragma unroll
            for (int i = 0; i < 10 + CM; ++i) if (EXACT || i < 10 + C) v[i] = d[i];
            int q = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s_loc[i] += v[i];
#pragma unroll
                for (int j = i; j < 4; ++j) s_ll[q++] += v[i] * v[j];
            }
            const size_t so = (((size_t)b * p.T + t) * cells + cell) * 3 + pr;       // [S, lh, lw, 3]
            if (obj_s) obj_s[so] = sigmoidf_(v[8]);
            if (cls_s) {
                float pc[CM];
                softmax_<CM, EXACT>(v + 10, pc, C);
#pragma unroll
                for (int c = 0; c < CM; ++c) if (EXACT || c < C) cls_s[so * C + c] = pc[c];
            }
        }
        float ev[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) ev[i] = s_loc[i] * invT;
        const size_t o = ((size_t)b * cells + cell) * 3 + pr;                        // [B, lh, lw, 3]
        if (ev_loc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ev_loc[o * 4 + i] = ev[i];
        }
        if (covar) {
            int q = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = i; j < 4; ++j) {
                    const float c = s_ll[q++] * invT - ev[i] * ev[j];                // E[l l^T] - E[l]E[l]^T (layers.py:383)
                    covar[o * 16 + i * 4 + j] = c; covar[o * 16 + j * 4 + i] = c;
                }
        }
    }
}

template <int CM, bool EXACT>
static hipError_t launch_epi_stats_c(const DecodeParams& p, float* ev, float* cov, float* os, float* cs, hipStream_t st) {
    const int64_t total = (int64_t)p.B * p.lh * p.lw * 3;
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(4096, (total + 255) / 256));
    hipLaunchKernelGGL((epi_stats_kernel<CM, EXACT>), dim3((unsigned)blocks), dim3(256), 0, st, p, ev, cov, os, cs);
    return hipGetLastError();
}

hipError_t launch_epi_stats(const DecodeParams& p, float* ev, float* cov, float* os, float* cs, hipStream_t st) {
    if (p.C < 1 || p.C > BYOLO_MAX_CLASSES) return hipErrorInvalidValue;
    if (p.C == 2) return launch_epi_stats_c<2, true>(p, ev, cov, os, cs, st);
    if (p.C <= 8) return launch_epi_stats_c<8, false>(p, ev, cov, os, cs, st);
    if (p.C <= 24) return launch_epi_stats_c<24, false>(p, ev, cov, os, cs, st);
    if (p.C <= 48) return launch_epi_stats_c<48, false>(p, ev, cov, os, cs, st);
    return launch_epi_stats_c<BYOLO_MAX_CLASSES, false>(p, ev, cov, os, cs, st);
}

template <int CM, bool EXACT>
static hipError_t launch_decode_c(int kind, const DecodeParams& p, hipStream_t st) {
    const int64_t total = (int64_t)p.B * p.lh * p.lw * 3;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    if (kind == 0) hipLaunchKernelGGL((decode_std_kernel<CM, EXACT>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else if (kind == 1) hipLaunchKernelGGL((decode_ale_kernel<CM, EXACT>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((decode_epi_kernel<CM, EXACT>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hipGetLastError();
}

// The reference takes any cls_cnt (lib_yolo/yolov3.py:180): exact builds for the usual counts, capacity builds
// (slots masked by the runtime count; the large ones spill to scratch -- correct, not fast) for everything else.
hipError_t launch_decode(int kind, const DecodeParams& p, hipStream_t st) {
    switch (p.C) {
        case 1: return launch_decode_c<1, true>(kind, p, st);
        case 2: return launch_decode_c<2, true>(kind, p, st);
        case 3: return launch_decode_c<3, true>(kind, p, st);
        case 4: return launch_decode_c<4, true>(kind, p, st);
        case 8: return launch_decode_c<8, true>(kind, p, st);
        case 80: return launch_decode_c<80, true>(kind, p, st);
        default: break;
    }
    if (p.C < 1 || p.C > BYOLO_MAX_CLASSES) return hipErrorInvalidValue;
    if (p.C <= 8) return launch_decode_c<8, false>(kind, p, st);
    if (p.C <= 24) return launch_decode_c<24, false>(kind, p, st);
    if (p.C <= 48) return launch_decode_c<48, false>(kind, p, st);
    return launch_decode_c<BYOLO_MAX_CLASSES, false>(kind, p, st);
}

// ------------------------------------------------------------------------------------------------
// K8: per-image sort of (score desc, index asc) as 64-bit keys, bitonic, one 1024-thread
// workgroup per image; sub-sequences of SORT_CH keys are sorted in LDS, only strides >= SORT_CH go
// through global memory (L2-resident).
// ------------------------------------------------------------------------------------------------
static constexpr int SORT_THREADS = 1024;
static constexpr int SORT_CH = 4096;                        // u64 keys per LDS chunk (32 KiB)

static inline int64_t next_pow2(int64_t n) { int64_t p = 1; while (p < n) p <<= 1; return p; }
static inline int64_t sort_np(int64_t N) { const int64_t p = next_pow2(N); return p < SORT_CH ? SORT_CH : p; }


__device__ __forceinline__ unsigned int score_key(float f) {
    // ascending key == descending score; NaN and scores <= -FLT_MAX are not candidates
    // (TF: `score > std::numeric_limits<float>::lowest()`), they sort last.
    if (!(f > -FLT_MAX)) return 0xFFFFFFFFu;
    unsigned int u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ~u;
}

__device__ __forceinline__ void ce(unsigned long long& a, unsigned long long& b, bool up) {
    if ((a > b) == up) { const unsigned long long t = a; a = b; b = t; }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_keys_kernel(const float* boxes, int64_t N, int D, int obj_idx,
                                                                 int64_t NP, unsigned long long* keys_all,
                                                                 int* n_valid_all, const int* need) {
    __shared__ unsigned long long sk[SORT_CH];
    __shared__ int s_valid;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (need && !need[b]) return;                            // the fast path already finished this image
    unsigned long long* keys = keys_all + (size_t)b * NP;
    const float* bx = boxes + (size_t)b * N * D;
    if (tid == 0) s_valid = 0;
    __syncthreads();
    int local_valid = 0;
    const int64_t nch = NP / SORT_CH;
    // phase 1: build keys and fully sort each chunk in LDS (direction by global index)
    for (int64_t ch = 0; ch < nch; ++ch) {
        for (int i = tid; i < SORT_CH; i += SORT_THREADS) {
            const int64_t g = ch * SORT_CH + i;
            unsigned long long k = ~0ull;
            if (g < N) {
                const unsigned int sk32 = score_key(bx[(size_t)g * D + obj_idx]);
                if (sk32 != 0xFFFFFFFFu) { k = ((unsigned long long)sk32 << 32) | (unsigned int)g; ++local_valid; }
            }
            sk[i] = k;
        }
        __syncthreads();
        for (int k = 2; k <= SORT_CH; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < SORT_CH / 2; t += SORT_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    const bool up = (((ch * SORT_CH + i) & k) == 0);
                    ce(sk[i], sk[i + j], up);
                }
                __syncthreads();
            }
        }
        for (int i = tid; i < SORT_CH; i += SORT_THREADS) keys[ch * SORT_CH + i] = sk[i];
        __syncthreads();
    }
    // phase 2: merge across chunks
    for (int64_t k = 2 * (int64_t)SORT_CH; k <= NP; k <<= 1) {
        for (int64_t j = k >> 1; j >= SORT_CH; j >>= 1) {
            for (int64_t t = tid; t < NP / 2; t += SORT_THREADS) {
                const int64_t i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const bool up = ((i & k) == 0);
                unsigned long long a = keys[i], c = keys[i + j];
                if ((a > c) == up) { keys[i] = c; keys[i + j] = a; }
            }
            __syncthreads();
        }
        for (int64_t ch = 0; ch < nch; ++ch) {
            for (int i = tid; i < SORT_CH; i += SORT_THREADS) sk[i] = keys[ch * SORT_CH + i];
            __syncthreads();
            const bool up = (((ch * SORT_CH) & k) == 0);
            for (int j = SORT_CH >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < SORT_CH / 2; t += SORT_THREADS) {
                    const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                    ce(sk[i], sk[i + j], up);
                }
                __syncthreads();
            }
            for (int i = tid; i < SORT_CH; i += SORT_THREADS) keys[ch * SORT_CH + i] = sk[i];
            __syncthreads();
        }
    }
    atomicAdd(&s_valid, local_valid);
    __syncthreads();
    if (tid == 0) n_valid_all[b] = s_valid;
}

// ------------------------------------------------------------------------------------------------
// K9: greedy NMS, one 1024-thread workgroup per image, candidates in sorted order, 1024 per round:
//   phase A  every lane tests its candidate against all boxes kept in earlier rounds (LDS broadcast)
//   phase B  the 16 waves take turns: intra-wave 64x64 suppression bit-matrix, serial resolve over
//            the 64 candidates with readlane, kept boxes appended to the LDS list, the later waves
//            then test against just those.
// IoU is evaluated operation-for-operation as TensorFlow's IOU() in float32 with single rounding
// (__f*_rn: no FMA contraction), std::min/std::max NaN semantics -> kept indices bit-exact.
// ------------------------------------------------------------------------------------------------
static constexpr int NMS_THREADS = 1024;
static constexpr int NMS_MAXK = 2048;                        // max_out limit per class pass

struct NBox { float y0, x0, y1, x1, area; };

__device__ __forceinline__ float smin_(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float smax_(float a, float b) { return (a < b) ? b : a; }

__device__ __forceinline__ NBox make_box(float b0, float b1, float b2, float b3) {
    NBox r;
    r.y0 = smin_(b0, b2); r.x0 = smin_(b1, b3);
    r.y1 = smax_(b0, b2); r.x1 = smax_(b1, b3);
    r.area = __fmul_rn(__fsub_rn(r.y1, r.y0), __fsub_rn(r.x1, r.x0));
    return r;
}
__device__ __forceinline__ bool iou_gt(const NBox& i, const NBox& j, float thr) {
    if (i.area <= 0.f || j.area <= 0.f) return false;        // IoU = 0
    const float iy0 = smax_(i.y0, j.y0), ix0 = smax_(i.x0, j.x0);
    const float iy1 = smin_(i.y1, j.y1), ix1 = smin_(i.x1, j.x1);
    const float inter = __fmul_rn(smax_(__fsub_rn(iy1, iy0), 0.f), smax_(__fsub_rn(ix1, ix0), 0.f));
    const float iou = __fdiv_rn(inter, __fsub_rn(__fadd_rn(i.area, j.area), inter));
    return iou > thr;
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const float* boxes, int64_t N, int D, int obj_idx,
                                                          int cls_start, int two_class, int max_out, float thr,
                                                          int64_t NP, const unsigned long long* keys_all,
                                                          const int* n_valid_all, float* rows, int32_t* kept,
                                                          int32_t* count, const int* need) {
    __shared__ float k_y0[NMS_MAXK], k_x0[NMS_MAXK], k_y1[NMS_MAXK], k_x1[NMS_MAXK], k_ar[NMS_MAXK];
    __shared__ int k_idx[2 * NMS_MAXK];
    __shared__ float w_y0[64], w_x0[64], w_y1[64], w_x1[64], w_ar[64];
    __shared__ int s_nk;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (need && !need[b]) return;                            // the fast path already finished this image
    const float* bx = boxes + (size_t)b * N * D;
    const unsigned long long* keys = keys_all + (size_t)b * NP;
    const int n_valid = n_valid_all[b];
    const int npass = two_class ? 2 : 1;
    int out_base = 0, first_cnt = 0;

    for (int pass = 0; pass < npass; ++pass) {
        if (tid == 0) s_nk = 0;
        __syncthreads();
        int nk_cur = 0;                                      // register copy of s_nk, uniform across the block
        for (int base = 0; base < n_valid && nk_cur < max_out; base += NMS_THREADS) {
            const int i = base + tid;
            bool alive = i < n_valid;
            int idx = -1;
            NBox me = {0, 0, 0, 0, 0};
            if (alive) {
                idx = (int)(keys[i] & 0xFFFFFFFFull);
                const float* r = bx + (size_t)idx * D;
                me = make_box(r[0], r[1], r[2], r[3]);
                if (two_class) {
                    const float c0 = r[cls_start], c1 = r[cls_start + 1];
                    alive = (pass == 0) ? (c0 > c1) : (c1 > c0);      // strict; ties dropped (:108-110)
                }
            }
            // phase A: against everything kept in earlier rounds
            for (int k = 0; k < nk_cur; ++k) {
                if (alive) {
                    const NBox kb = {k_y0[k], k_x0[k], k_y1[k], k_x1[k], k_ar[k]};
                    if (iou_gt(me, kb, thr)) alive = false;
                }
            }
            // phase B: the 16 waves take turns
            for (int sub = 0; sub < NMS_THREADS / 64 && nk_cur < max_out; ++sub) {
                if (wave == sub) {
                    const unsigned long long alive_mask = __ballot(alive);
                    if (alive_mask) {
                        w_y0[lane] = me.y0; w_x0[lane] = me.x0; w_y1[lane] = me.y1; w_x1[lane] = me.x1; w_ar[lane] = me.area;
                        unsigned long long supp = 0ull;      // bit e: earlier live candidate e overlaps me
                        for (int e = 0; e < 64; ++e) {
                            if (e < lane && ((alive_mask >> e) & 1ull) && alive) {
                                const NBox ob = {w_y0[e], w_x0[e], w_y1[e], w_x1[e], w_ar[e]};
                                if (iou_gt(me, ob, thr)) supp |= (1ull << e);
                            }
                        }
                        const unsigned int supp_lo = (unsigned int)supp, supp_hi = (unsigned int)(supp >> 32);
                        unsigned long long keptmask = 0ull;
                        int room = max_out - nk_cur;
                        for (int e = 0; e < 64 && room > 0; ++e) {       // serial greedy resolve (uniform)
                            if ((alive_mask >> e) & 1ull) {
                                const unsigned long long se =
                                    ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)supp_hi, e) << 32) |
                                    (unsigned int)__builtin_amdgcn_readlane((int)supp_lo, e);
                                if ((se & keptmask) == 0ull) { keptmask |= (1ull << e); --room; }
                            }
                        }
                        if ((keptmask >> lane) & 1ull) {
                            const int pos = nk_cur + __popcll(keptmask & ((1ull << lane) - 1ull));
                            k_y0[pos] = me.y0; k_x0[pos] = me.x0; k_y1[pos] = me.y1; k_x1[pos] = me.x1; k_ar[pos] = me.area;
                            k_idx[out_base + pos] = idx;
                        }
                        if (lane == 0) s_nk = nk_cur + __popcll(keptmask);
                    }
                }
                __syncthreads();
                const int nk_new = s_nk;
                if (wave > sub && alive) {
                    for (int k = nk_cur; k < nk_new; ++k) {
                        const NBox kb = {k_y0[k], k_x0[k], k_y1[k], k_x1[k], k_ar[k]};
                        if (iou_gt(me, kb, thr)) { alive = false; break; }
                    }
                }
                nk_cur = nk_new;
                __syncthreads();                             // nobody still reads s_nk / w_* when the next wave writes
            }
        }
        if (pass == 0) first_cnt = nk_cur;
        out_base += nk_cur;
        __syncthreads();
    }
    // gather rows (tf.gather) + zero fill
    const int cap = max_out * npass;
    float* ro = rows + (size_t)b * cap * D;
    int32_t* ko = kept + (size_t)b * cap;
    for (int e = tid; e < cap * D; e += NMS_THREADS) {
        const int k = e / D, c = e - k * D;
        ro[e] = (k < out_base) ? bx[(size_t)k_idx[k] * D + c] : 0.f;
    }
    for (int k = tid; k < cap; k += NMS_THREADS) ko[k] = (k < out_base) ? k_idx[k] : -1;
    if (tid == 0) { count[2 * b] = out_base; count[2 * b + 1] = first_cnt; }
}

// ------------------------------------------------------------------------------------------------
// Fast path of K8/K9: greedy NMS only ever looks at a PREFIX of the score order (until max_out boxes
// are kept), and on that prefix it is a bit-matrix problem that the whole chip can work on:
//   topk_select   per image, one workgroup: 3-level radix select (11+11+10 bits of the score key) of
//                 the best >= NMS_TOPK candidates, compaction, bitonic sort in LDS  -> sorted prefix
//   nms_matrix    suppression bit matrix of the prefix (row r, bit c: c later than r and IoU > thr),
//                 64x64 tiles on ALL CUs (upper triangle only)
//   nms_scan      per image: serial greedy scan over the prefix, 64 candidates per step from LDS,
//                 the 4096-bit "removed" set lives in one wave's registers (lane w = word w)
//   nms_finish    gather rows (tf.gather) + counts
// Same result as the sequential definition.  If the prefix is exhausted before max_out boxes are kept
// while more candidates exist (pathological clustering), or scores tie en masse, the image is flagged
// and the exact general kernels above (full sort + nms_kernel) redo it -- decided on the device, no
// host round trip.
// ------------------------------------------------------------------------------------------------
static constexpr int NMS_TOPK = 4096;                        // prefix length / matrix dimension
static constexpr int NMS_CAP = 8192;                         // LDS sort capacity (prefix + score ties)
static constexpr int NMS_WORDS = NMS_TOPK / 64;

struct NmsWs {                                               // carve-up of the workspace
    unsigned long long* keys; int* n_valid;                  // general path
    unsigned long long* cand; int* n_cand; int* more; int* need; int* cnt;   // cnt [B][2]
    unsigned long long* mask; int* kidx;
};
static size_t nms_ws_layout(int B, int64_t N, char* base, NmsWs* w) {
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + o : nullptr; o += (bytes + 255) / 256 * 256; return p; };
    char* a = take((size_t)B * sort_np(N) * 8); char* b = take((size_t)B * 4);
    char* c = take((size_t)B * NMS_CAP * 8); char* d = take((size_t)B * 4); char* e = take((size_t)B * 4);
    char* f = take((size_t)B * 4); char* g = take((size_t)B * 8);
    char* hh = take((size_t)B * NMS_TOPK * NMS_WORDS * 8); char* i = take((size_t)B * 2 * NMS_MAXK * 4);
    if (w) { w->keys = (unsigned long long*)a; w->n_valid = (int*)b; w->cand = (unsigned long long*)c; w->n_cand = (int*)d;
             w->more = (int*)e; w->need = (int*)f; w->cnt = (int*)g; w->mask = (unsigned long long*)hh; w->kidx = (int*)i; }
    return o;
}
size_t nms_workspace_bytes(int B, int64_t N) { return nms_ws_layout(B, N, nullptr, nullptr); }

__global__ __launch_bounds__(1024) void topk_select_kernel(const float* boxes, int64_t N, int D, int obj_idx, int cls_start,
                                                           int two_class, int pass, unsigned long long* cand_all,
                                                           int* n_cand_all, int* more_all, int* need) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long sbuf[];      // NMS_CAP keys
    __shared__ int hist[2048];
    __shared__ int part[32];
    __shared__ int s_digit, s_below, s_le, s_cnt;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* bx = boxes + (size_t)b * N * D;
    if (pass == 0 && tid == 0) need[b] = 0;
    auto key_of = [&](int64_t i) -> unsigned int {
        const float* r = bx + (size_t)i * D;
        if (two_class) {
            const float c0 = r[cls_start], c1 = r[cls_start + 1];
            if (!((pass == 0) ? (c0 > c1) : (c1 > c0))) return 0xFFFFFFFFu;       // strict; ties dropped
        }
        return score_key(r[obj_idx]);
    };
    unsigned int prefix = 0, pmask = 0;
    int below = 0, n_valid = 0, C = 0;
    bool ties_overflow = false;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int level = 0; level < 3; ++level) {
        const int shift = shifts[level], nd = 1 << widths[level];
        for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
        __syncthreads();
        for (int64_t i = tid; i < N; i += 1024) {
            const unsigned int k = key_of(i);
            if (k != 0xFFFFFFFFu && (k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & (nd - 1)], 1);
        }
        __syncthreads();
        if (tid < 32) { int s = 0; for (int d = tid * 64; d < tid * 64 + 64; ++d) s += hist[d]; part[tid] = s; }
        __syncthreads();
        if (tid == 0) {
            int total = 0;
            for (int q = 0; q < 32; ++q) total += part[q];
            s_cnt = total;                                   // level 0: nothing filtered yet = #valid scores
        }
        __syncthreads();
        if (level == 0) n_valid = s_cnt;
        const int target = n_valid < NMS_TOPK ? n_valid : NMS_TOPK;
        if (tid == 0) {                                      // smallest digit d with below + #(digit <= d) >= target
            int cum = below, q = 0;
            while (q < 31 && cum + part[q] < target) { cum += part[q]; ++q; }
            int d = q * 64;
            while (d < q * 64 + 63 && cum + hist[d] < target) { cum += hist[d]; ++d; }
            s_digit = d; s_below = cum; s_le = cum + hist[d];
        }
        __syncthreads();
        prefix |= (unsigned int)s_digit << shift;
        pmask |= (unsigned int)(nd - 1) << shift;
        below = s_below;
        C = s_le;
        __syncthreads();
        if (C <= NMS_CAP) break;                             // every key <= this (partial) threshold fits
        if (level == 2) ties_overflow = true;                // > NMS_CAP boxes share one exact score
    }
    if (n_valid == 0) C = 0;
    if (ties_overflow) { C = 0; if (tid == 0) need[b] = 1; }
    // compaction of the keys <= threshold, then sort
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    if (C > 0) {
        for (int64_t i = tid; i < N; i += 1024) {
            const unsigned int k = key_of(i);
            if (k != 0xFFFFFFFFu && (k & pmask) <= prefix) {
                const int pos = atomicAdd(&s_cnt, 1);
                sbuf[pos] = ((unsigned long long)k << 32) | (unsigned int)i;
            }
        }
    }
    __syncthreads();
    int P2 = 64;
    while (P2 < C) P2 <<= 1;
    for (int i = C + tid; i < P2; i += 1024) sbuf[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < P2 / 2; t += 1024) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                ce(sbuf[i], sbuf[i + j], (i & k) == 0);
            }
            __syncthreads();
        }
    unsigned long long* cand = cand_all + (size_t)b * NMS_CAP;
    for (int i = tid; i < C; i += 1024) cand[i] = sbuf[i];
    if (tid == 0) { n_cand_all[b] = C; more_all[b] = n_valid > C ? 1 : 0; }
}

__global__ __launch_bounds__(64) void nms_matrix_kernel(const float* boxes, int64_t N, int D, float thr,
                                                        const unsigned long long* cand_all, const int* n_cand_all,
                                                        unsigned long long* mask_all) {
    __shared__ float c_y0[64], c_x0[64], c_y1[64], c_x1[64], c_ar[64];
    const int w = blockIdx.x, rb = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    if (w < rb) return;                                      // lower triangle is never read
    int Kc = n_cand_all[b];
    if (Kc > NMS_TOPK) Kc = NMS_TOPK;
    if (rb * 64 >= Kc) return;
    const float* bx = boxes + (size_t)b * N * D;
    const unsigned long long* cand = cand_all + (size_t)b * NMS_CAP;
    const int r = rb * 64 + lane, c = w * 64 + lane;
    NBox me = {0, 0, 0, 0, 0};
    if (r < Kc) { const float* q = bx + (size_t)(cand[r] & 0xFFFFFFFFull) * D; me = make_box(q[0], q[1], q[2], q[3]); }
    if (c < Kc) {
        const float* q = bx + (size_t)(cand[c] & 0xFFFFFFFFull) * D;
        const NBox o = make_box(q[0], q[1], q[2], q[3]);
        c_y0[lane] = o.y0; c_x0[lane] = o.x0; c_y1[lane] = o.y1; c_x1[lane] = o.x1; c_ar[lane] = o.area;
    }
    __syncthreads();
    unsigned long long bits = 0ull;
    if (r < Kc) {
        for (int e = 0; e < 64; ++e) {
            const int col = w * 64 + e;
            if (col > r && col < Kc) {
                const NBox o = {c_y0[e], c_x0[e], c_y1[e], c_x1[e], c_ar[e]};
                if (iou_gt(me, o, thr)) bits |= 1ull << e;
            }
        }
        mask_all[((size_t)b * NMS_TOPK + r) * NMS_WORDS + w] = bits;
    }
}

__global__ __launch_bounds__(256) void nms_scan_kernel(int max_out, int pass, const unsigned long long* cand_all,
                                                       const int* n_cand_all, const int* more_all,
                                                       const unsigned long long* mask_all, int* kidx_all, int* cnt_all,
                                                       int* need) {
    // The 64 matrix rows of a step, double-buffered (2 x 32 KiB, dynamic LDS): waves 1..3 fetch the rows of step c+1
    // while wave 0 resolves step c -- the fetch latency (L2) leaves the serial chain.
    extern __shared__ __attribute__((aligned(16))) unsigned long long rows_all[];   // [2][64][NMS_WORDS]
    __shared__ int s_nk, s_done;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_cand = n_cand_all[b];
    const int Kc = n_cand > NMS_TOPK ? NMS_TOPK : n_cand;
    const int nwords = (Kc + 63) / 64;
    const unsigned long long* cand = cand_all + (size_t)b * NMS_CAP;
    const unsigned long long* mask = mask_all + (size_t)b * NMS_TOPK * NMS_WORDS;
    const int base = pass ? cnt_all[2 * b] : 0;
    int* kidx = kidx_all + (size_t)b * 2 * NMS_MAXK + base;
    if (tid == 0) { s_nk = 0; s_done = 0; }
    auto fetch = [&](int c, int first, int nthreads) {       // rows of step c -> buffer c & 1, by `nthreads` threads from `first`
        unsigned long long (*rows)[NMS_WORDS] = reinterpret_cast<unsigned long long (*)[NMS_WORDS]>(rows_all + (size_t)(c & 1) * 64 * NMS_WORDS);
        for (int t = tid - first; t < 64 * NMS_WORDS; t += nthreads) {
            const int r = t / NMS_WORDS, w = t - r * NMS_WORDS;
            const int gi = c * 64 + r;
            rows[r][w] = (w >= c && w < nwords && gi < Kc) ? mask[(size_t)gi * NMS_WORDS + w] : 0ull;
        }
    };
    unsigned long long rem = 0ull;                           // wave 0, lane w: word w of the removed set
    if (nwords > 0) fetch(0, 0, 256);
    __syncthreads();
    for (int c = 0; c < nwords; ++c) {
        if (wave != 0) {
            if (c + 1 < nwords) fetch(c + 1, 64, 192);
        } else {
            const unsigned long long (*rows)[NMS_WORDS] = reinterpret_cast<const unsigned long long (*)[NMS_WORDS]>(rows_all + (size_t)(c & 1) * 64 * NMS_WORDS);
            // The only serial dependency is the removed-word of THIS step (`cur`); everything on its
            // chain stays in registers: lane i holds word c of row i (suppression inside the step) and
            // candidate i's box index.  Folding the kept rows into the other 63 words of the removed set
            // and writing the kept indices happen after the loop, fully parallel.
            const int nk0 = s_nk;
            const int gi_l = c * 64 + lane;
            const unsigned long long own = rows[lane][c];
            const int my_idx = gi_l < Kc ? (int)(cand[gi_l] & 0xFFFFFFFFull) : -1;
            const unsigned int own_lo = (unsigned int)own, own_hi = (unsigned int)(own >> 32);
            unsigned long long cur =
                ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)(rem >> 32), c) << 32) |
                (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)rem, c);
            unsigned long long keptmask = 0ull;
            int nk = nk0;
            const int lim = Kc - c * 64 < 64 ? Kc - c * 64 : 64;
            if (lim < 64) cur |= ~0ull << lim;               // positions past the last candidate are never kept
            // visit only the survivors: the next candidate that is not suppressed = the lowest clear bit of `cur`
            // at or above the last kept one (a suppressed run costs nothing)
            unsigned long long done = 0ull;                  // bits below the scan position
            while (nk < max_out) {
                const unsigned long long open = ~(cur | done);
                if (!open) break;
                const int i = __builtin_ctzll(open);
                keptmask |= 1ull << i;
                ++nk;
                cur |= ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)own_hi, i) << 32) |
                       (unsigned int)__builtin_amdgcn_readlane((int)own_lo, i);
                done = (i == 63) ? ~0ull : ((2ull << i) - 1ull);
            }
            if ((keptmask >> lane) & 1ull) kidx[nk0 + __popcll(keptmask & ((1ull << lane) - 1ull))] = my_idx;
            // fold the kept rows into the removed set: 64 independent LDS reads per lane, selected by the kept bit
            unsigned long long acc = 0ull;
#pragma unroll 16
            for (int i = 0; i < 64; ++i) acc |= ((keptmask >> i) & 1ull) ? rows[i][lane] : 0ull;
            rem |= acc;
            if (lane == 0) { s_nk = nk; s_done = nk >= max_out ? 1 : 0; }
        }
        __syncthreads();
        if (s_done) break;
    }
    if (tid == 0) {
        const int nk = s_nk;
        cnt_all[2 * b + pass] = nk;
        // prefix exhausted without filling max_out although more candidates exist -> general path
        if (nk < max_out && (more_all[b] || n_cand > Kc)) need[b] = 1;
    }
}

__global__ __launch_bounds__(1024) void nms_finish_kernel(const float* boxes, int64_t N, int D, int max_out, int npass,
                                                         const int* kidx_all, const int* cnt_all, const int* need,
                                                         float* rows, int32_t* kept, int32_t* count) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (need[b]) return;                                     // the general kernels write this image
    const float* bx = boxes + (size_t)b * N * D;
    const int* kidx = kidx_all + (size_t)b * 2 * NMS_MAXK;
    const int first = cnt_all[2 * b], total = first + (npass > 1 ? cnt_all[2 * b + 1] : 0);
    const int cap = max_out * npass;
    float* ro = rows + (size_t)b * cap * D;
    int32_t* ko = kept + (size_t)b * cap;
    for (int e = tid; e < cap * D; e += 1024) {
        const int k = e / D, c = e - k * D;
        ro[e] = (k < total) ? bx[(size_t)kidx[k] * D + c] : 0.f;
    }
    for (int k = tid; k < cap; k += 1024) ko[k] = (k < total) ? kidx[k] : -1;
    if (tid == 0) { count[2 * b] = total; count[2 * b + 1] = first; }
}

hipError_t launch_sort_nms(const NmsParams& p, hipStream_t st) {
    if (p.max_out > NMS_MAXK || p.max_out < 1) return hipErrorInvalidValue;
    const int64_t NP = sort_np(p.N);
    if (p.ws_bytes < nms_workspace_bytes(p.B, p.N)) return hipErrorInvalidValue;
    NmsWs w;
    nms_ws_layout(p.B, p.N, reinterpret_cast<char*>(p.ws), &w);
    const bool general_only = p.general_only != 0;            // byolo_plan_opts.nms_general
    const int* need = nullptr;
    if (!general_only) {
        static std::atomic<uint64_t> attr_done{0};
        const size_t lds = (size_t)NMS_CAP * sizeof(unsigned long long);
        if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(topk_select_kernel), lds, attr_done); e != hipSuccess) return e;
        static std::atomic<uint64_t> scan_attr_done{0};
        const size_t scan_lds = (size_t)2 * 64 * NMS_WORDS * sizeof(unsigned long long);      // 64 KiB
        if (hipError_t e = set_dynamic_lds_once(reinterpret_cast<const void*>(nms_scan_kernel), scan_lds, scan_attr_done); e != hipSuccess) return e;
        const int npass = p.two_class ? 2 : 1;
        for (int pass = 0; pass < npass; ++pass) {
            hipLaunchKernelGGL(topk_select_kernel, dim3(p.B), dim3(1024), lds, st, p.boxes, p.N, p.D, p.obj_idx, p.cls_start,
                               p.two_class, pass, w.cand, w.n_cand, w.more, w.need);
            hipLaunchKernelGGL(nms_matrix_kernel, dim3(NMS_WORDS, NMS_WORDS, p.B), dim3(64), 0, st, p.boxes, p.N, p.D,
                               p.iou_thr, w.cand, w.n_cand, w.mask);
            hipLaunchKernelGGL(nms_scan_kernel, dim3(p.B), dim3(256), scan_lds, st, p.max_out, pass, w.cand, w.n_cand, w.more,
                               w.mask, w.kidx, w.cnt, w.need);
        }
        hipLaunchKernelGGL(nms_finish_kernel, dim3(p.B), dim3(1024), 0, st, p.boxes, p.N, p.D, p.max_out, npass, w.kidx,
                           w.cnt, w.need, p.rows, p.kept, p.count);
        need = w.need;
    }
    // general path (exact for every input); a no-op per image unless flagged by the fast path
    hipLaunchKernelGGL(sort_keys_kernel, dim3(p.B), dim3(SORT_THREADS), 0, st, p.boxes, p.N, p.D, p.obj_idx, NP, w.keys,
                       w.n_valid, need);
    hipLaunchKernelGGL(nms_kernel, dim3(p.B), dim3(NMS_THREADS), 0, st, p.boxes, p.N, p.D, p.obj_idx, p.cls_start,
                       p.two_class, p.max_out, p.iou_thr, NP, w.keys, w.n_valid, p.rows, p.kept, p.count, need);
    return hipGetLastError();
}

}  // namespace byk
