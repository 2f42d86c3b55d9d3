// train_kernels.hip -- ground-truth encoding and the training loss
//
//   encode_gt_*_kernel lib_yolo/tfdata.py:77-171 `encode_boxes` (with `create_prior_data` :16-75 over lib_yolo/data.py:125-166
//                      and `calc_iou` :174-189): the sequential `tf.while_loop` over an image's boxes, one thread per prior box
//   loss_kernel        lib_yolo/layers.py:126-188 `loss_tf` on the tensors of `split_detection(_aleatoric)` (:11-84), one
//                      detection layer per launch; optionally d(loc + obj + cls) / d(raw output)
//
// Both are small, HBM- / latency-shaped launches (22 743 prior boxes per image at 608 x 608; a loss launch reads the raw
// detection tensor once).  What matters here is the arithmetic contract: every float32 operation below is the reference's
// operation in the reference's order -- no fused multiply-add (the file is compiled with contraction off), IEEE division --,
// so the DISCRETE outputs (object / class / ignore masks) are bit-identical to the float32 oracle and the transcendental
// ones (logit, log) differ by the libm's last bits only.  Sums are accumulated in double in a fixed order (deterministic).
#include <hip/hip_runtime.h>
#include "byolo_kernels.h"

#pragma clang fp contract(off)

namespace byk {

// ---------------------------------------------------------------------------------------------------------------------
// prior box n of layer L at (row, col, box): float32 views of the doubles lib_yolo/data.py:125-166 computes
// ---------------------------------------------------------------------------------------------------------------------
struct PriorBox { float ymin, xmin, ymax, xmax, area, cx, cy, pw, ph, lw, lh; };

__device__ __forceinline__ PriorBox prior_box(const EncodeGtParams& p, int n) {
    int l = 0;
    while (l + 1 < p.n_layers && n >= p.base[l + 1]) ++l;
    const int r = n - p.base[l];
    const int lw = p.lw[l], lh = p.lh[l];
    const int box = r % 3, cell = r / 3, col = cell % lw, row = cell / lw;
    const double ph = p.ph[l][box], pw = p.pw[l][box];
    const double y_center = (row + 0.5) / (double)lh, x_center = (col + 0.5) / (double)lw;
    const double h2 = ph / 2., w2 = pw / 2.;
    PriorBox b;
    b.ymin = (float)(y_center - h2); b.xmin = (float)(x_center - w2);
    b.ymax = (float)(y_center + h2); b.xmax = (float)(x_center + w2);
    b.area = (float)(ph * pw);
    b.cx = (float)(col / (double)lw); b.cy = (float)(row / (double)lh);
    b.pw = (float)pw; b.ph = (float)ph; b.lw = (float)lw; b.lh = (float)lh;
    return b;
}

// lib_yolo/tfdata.py:174-189
__device__ __forceinline__ float calc_iou(const PriorBox& b, float r0, float r1, float r2, float r3) {
    const float int_ymin = fmaxf(b.ymin, r0), int_xmin = fmaxf(b.xmin, r1);
    const float int_ymax = fminf(b.ymax, r2), int_xmax = fminf(b.xmax, r3);
    const float h = fmaxf(int_ymax - int_ymin, 0.f), w = fmaxf(int_xmax - int_xmin, 0.f);
    const float inter = h * w;
    const float uni = (b.area - inter) + ((r2 - r0) * (r3 - r1));
    return inter / uni;
}

// lib_yolo/tfdata.py:7-11
__device__ __forceinline__ float logit_tf(float x) { return -logf((1.f / x) - 1.f); }

// Two launches, one thread per (image, prior box) in both; the prior box is computed once per thread (double arithmetic) and
// kept in registers while the thread walks the image's boxes in order, like the reference's sequential tf.while_loop:
//   encode_gt_max_kernel   tf.reduce_max(iou) of every box over ALL prior boxes of all layers -> best[image][box] (wave maximum,
//                          then one atomic maximum per wave and box on the IoU's bit pattern: IoUs are >= 0, so unsigned order =
//                          float order; NaN never wins, like a maximum over numbers)
//   encode_gt_assign_kernel the loop body (tfdata.py:109-149) with its loop variables in registers, one store at the end
constexpr int ENC_THREADS = 256;

__global__ __launch_bounds__(ENC_THREADS) void encode_gt_max_kernel(const EncodeGtParams p) {
    const int img = blockIdx.y, n = blockIdx.x * ENC_THREADS + threadIdx.x;
    const int cnt = min(max(p.counts ? p.counts[img] : p.max_boxes, 0), p.max_boxes);
    const float* bb = p.boxes + (size_t)img * p.max_boxes * 4;
    const bool live = n < p.N;
    const PriorBox b = prior_box(p, live ? n : 0);
    for (int g = 0; g < cnt; ++g) {
        const float4 r = reinterpret_cast<const float4*>(bb)[g];
        float m = live ? calc_iou(b, r.x, r.y, r.z, r.w) : -1.f;
        m = fmaxf(m, -1.f);                                            // NaN -> -1
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if ((threadIdx.x & 63) == 0 && m >= 0.f) atomicMax(p.best + (size_t)img * p.max_boxes + g, __float_as_uint(m));
    }
}

__global__ __launch_bounds__(ENC_THREADS) void encode_gt_assign_kernel(const EncodeGtParams p) {
    const int img = blockIdx.y, n = blockIdx.x * ENC_THREADS + threadIdx.x;
    if (n >= p.N) return;
    const int cnt = min(max(p.counts ? p.counts[img] : p.max_boxes, 0), p.max_boxes);
    const float* bb = p.boxes + (size_t)img * p.max_boxes * 4;
    const int32_t* lab = p.labels + (size_t)img * p.max_boxes;
    const unsigned* best = p.best + (size_t)img * p.max_boxes;
    const PriorBox b = prior_box(p, n);
    // the loop variables of the tf.while_loop (tfdata.py:87-93)
    float4 loc = make_float4(0.f, 0.f, 0.f, 0.f);
    float obj = 0.f, ign = 1.f;
    int32_t cls = 0;
    const float eps = 1e-7f, hi = (float)(1 - 1e-7);
    for (int g = 0; g < cnt; ++g) {
        const float4 r = reinterpret_cast<const float4*>(bb)[g];
        const float r0 = r.x, r1 = r.y, r2 = r.z, r3 = r.w;
        const float w = r3 - r1, h = r2 - r0, x = (r3 + r1) / 2.f, y = (r2 + r0) / 2.f;
        const float dx = b.lw * (x - b.cx), dy = b.lh * (y - b.cy);
        const float iou = calc_iou(b, r0, r1, r2, r3);
        const float mx = __uint_as_float(best[g]);
        const bool om = (iou >= mx) && dx >= 0.f && dx <= 1.f && dy >= 0.f && dy <= 1.f;
        if (om) {
            loc.x = logit_tf(fminf(fmaxf(dx, eps), hi));
            loc.y = logit_tf(fminf(fmaxf(dy, eps), hi));
            loc.z = logf(fmaxf(w / b.pw, eps));
            loc.w = logf(fmaxf(h / b.ph, eps));
            cls = lab[g];
            obj = 1.f;
        }
        if (iou >= p.ign_thresh) ign = 0.f;
    }
    const size_t o = (size_t)img * p.N + n;
    reinterpret_cast<float4*>(p.loc)[o] = loc;
    p.obj[o] = obj; p.cls[o] = cls;
    p.ign[o] = fmaxf(ign, obj);                                        // tfdata.py:156
}

size_t encode_gt_workspace_bytes(int B, int max_boxes) { return (size_t)std::max(B, 1) * std::max(max_boxes, 1) * sizeof(unsigned); }

hipError_t launch_encode_gt(const EncodeGtParams& p, hipStream_t st) {
    if (p.B < 1 || p.N < 1 || p.n_layers < 1 || p.n_layers > 4 || p.max_boxes < 0 || p.B > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((p.N + ENC_THREADS - 1) / ENC_THREADS), (unsigned)p.B);
    if (p.max_boxes > 0) {
        if (!p.best) return hipErrorInvalidValue;
        if (hipError_t e = hipMemsetAsync(p.best, 0, encode_gt_workspace_bytes(p.B, p.max_boxes), st); e != hipSuccess) return e;
        hipLaunchKernelGGL(encode_gt_max_kernel, grid, dim3(ENC_THREADS), 0, st, p);
    }
    hipLaunchKernelGGL(encode_gt_assign_kernel, grid, dim3(ENC_THREADS), 0, st, p);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// loss of one detection layer: thread = (sample, cell, prior box); partial sums per workgroup in double, summed in a
// fixed order by loss_finish_kernel
// ---------------------------------------------------------------------------------------------------------------------
constexpr int LOSS_THREADS = 256;

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.;
    for (int i = 0; i < LOSS_THREADS / 64; ++i) s += red[i];
    return s;
}

__global__ __launch_bounds__(LOSS_THREADS) void loss_kernel(const LossParams p) {
    __shared__ double red[LOSS_THREADS / 64];
    const int64_t per_img = (int64_t)p.lh * p.lw * 3;
    const int64_t total = (int64_t)p.S * per_img;
    const int blk = p.aleatoric ? 10 + 2 * p.C : 5 + p.C;
    const int o_obj = p.aleatoric ? 8 : 4, o_cls = p.aleatoric ? 10 : 5;
    const float bs = (float)p.S;
    double a_loc = 0., a_obj = 0., a_cls = 0.;
    for (int64_t i = (int64_t)blockIdx.x * LOSS_THREADS + threadIdx.x; i < total; i += (int64_t)gridDim.x * LOSS_THREADS) {
        const int64_t s = i / per_img, r = i - s * per_img;          // r = (row * lw + col) * 3 + box
        const int64_t cell = r / 3; const int box = (int)(r - cell * 3);
        const float* x = p.raw + ((size_t)s * p.lh * p.lw + cell) * p.pitch + box * blk;
        float* gx = p.grad ? p.grad + ((size_t)s * p.lh * p.lw + cell) * p.grad_pitch + box * blk : nullptr;
        const size_t gi = (size_t)s * p.gt_stride + r;
        const float z = p.gt_obj[gi], ig = p.gt_ign[gi];
        const float4 gl = reinterpret_cast<const float4*>(p.gt_loc)[gi];
        const float g4[4] = {gl.x, gl.y, gl.z, gl.w};
        // localization (layers.py:146-158)
        float t_loc = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = g4[k] - x[k];
            float term = d * d;
            float e = 1.f, lv = 0.f; bool inside = true;
            if (p.aleatoric_loss) {
                const float raw_lv = x[4 + k];
                lv = fminf(fmaxf(raw_lv, -40.f), 40.f);
                inside = raw_lv >= -40.f && raw_lv <= 40.f;
                e = expf(-lv);
                term = term * e;
                term = term + lv;
            }
            term = term * z;
            t_loc += term;
            if (gx) {
                gx[k] = (-d * e) * z / bs;
                if (p.aleatoric_loss) gx[4 + k] = inside ? (1.f - d * d * e) * z / (2.f * bs) : 0.f;
                else if (p.aleatoric) gx[4 + k] = 0.f;
            }
        }
        a_loc += (double)t_loc;
        // objectness: tf.nn.sigmoid_cross_entropy_with_logits = max(x, 0) - x z + log(1 + exp(-|x|))  (layers.py:164-166)
        const float xo = x[o_obj];
        const float t_obj = (fmaxf(xo, 0.f) - xo * z + log1pf(expf(-fabsf(xo)))) * ig;
        a_obj += (double)t_obj;
        // classes: tf.nn.sparse_softmax_cross_entropy_with_logits = logsumexp(x) - x[label]  (layers.py:175-177)
        const int lab = p.gt_cls[gi];
        float m = x[o_cls];
        for (int c = 1; c < p.C; ++c) m = fmaxf(m, x[o_cls + c]);
        float se = 0.f;
        for (int c = 0; c < p.C; ++c) se += expf(x[o_cls + c] - m);
        const float lse = logf(se) + m;
        const float picked = (lab >= 0 && lab < p.C) ? x[o_cls + lab] : NAN;
        const float t_cls = (lse - picked) * z;
        a_cls += (double)t_cls;
        if (gx) {
            gx[o_obj] = (1.f / (1.f + expf(-xo)) - z) * ig / bs;
            if (p.aleatoric) gx[o_obj + 1] = 0.f;                      // log_obj_stddev: not in the loss (layers.py:160-163)
            for (int c = 0; c < p.C; ++c) {
                gx[o_cls + c] = (expf(x[o_cls + c] - lse) - (c == lab ? 1.f : 0.f)) * z / bs;
                if (p.aleatoric) gx[o_cls + p.C + c] = 0.f;            // log_cls_stddev
            }
        }
    }
    const double s_loc = block_sum(a_loc, red), s_obj = block_sum(a_obj, red), s_cls = block_sum(a_cls, red);
    if (threadIdx.x == 0) { p.partial[3 * blockIdx.x] = s_loc; p.partial[3 * blockIdx.x + 1] = s_obj; p.partial[3 * blockIdx.x + 2] = s_cls; }
}

__global__ __launch_bounds__(64) void loss_finish_kernel(const double* partial, int n, double bs, double* out) {
    const int k = threadIdx.x;
    if (k >= 3) return;
    double s = 0.;
    for (int i = 0; i < n; ++i) s += partial[3 * i + k];
    out[k] = k == 0 ? s / (2. * bs) : s / bs;                          // layers.py:157, :168, :180
}

size_t loss_workspace_bytes() { return (size_t)LOSS_MAX_BLOCKS * 3 * sizeof(double); }

hipError_t launch_loss(const LossParams& p, hipStream_t st) {
    if (p.S < 1 || p.lh < 1 || p.lw < 1 || p.C < 1 || p.C > BYOLO_MAX_CLASSES) return hipErrorInvalidValue;
    const int64_t total = (int64_t)p.S * p.lh * p.lw * 3;
    const int blocks = (int)std::min<int64_t>((total + LOSS_THREADS - 1) / LOSS_THREADS, LOSS_MAX_BLOCKS);
    hipLaunchKernelGGL(loss_kernel, dim3((unsigned)blocks), dim3(LOSS_THREADS), 0, st, p);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, p.partial, blocks, (double)p.S, p.out);
    return hipGetLastError();
}

}  // namespace byk
