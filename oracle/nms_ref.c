/* Oracle (TEST INFRASTRUCTURE, not product): plain-C restatement of
 * tf.image.non_max_suppression as the reference calls it
 *   (/root/reference/inference_epistemic.py:101, inference_aleatoric.py:107,
 *    inference_standard_yolov3.py:107)
 * following TensorFlow's published algorithm (tensorflow/core/kernels/non_max_suppression_op.cc,
 * TF 1.x; third-party dependency of the reference, version unpinned => "parity unpinned").
 * Same semantics as oracle/nms_ref.py (see there).  Compile with -ffp-contract=off so every
 * float32 operation rounds once, which the HIP kernel reproduces.
 */
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float smin(float a, float b) { return (b < a) ? b : a; }
static inline float smax(float a, float b) { return (a < b) ? b : a; }

static float iou(const float* bi, const float* bj) {
    const float ymin_i = smin(bi[0], bi[2]), xmin_i = smin(bi[1], bi[3]);
    const float ymax_i = smax(bi[0], bi[2]), xmax_i = smax(bi[1], bi[3]);
    const float ymin_j = smin(bj[0], bj[2]), xmin_j = smin(bj[1], bj[3]);
    const float ymax_j = smax(bj[0], bj[2]), xmax_j = smax(bj[1], bj[3]);
    const float area_i = (ymax_i - ymin_i) * (xmax_i - xmin_i);
    const float area_j = (ymax_j - ymin_j) * (xmax_j - xmin_j);
    if (area_i <= 0 || area_j <= 0) return 0.0f;
    const float iy0 = smax(ymin_i, ymin_j), ix0 = smax(xmin_i, xmin_j);
    const float iy1 = smin(ymax_i, ymax_j), ix1 = smin(xmax_i, xmax_j);
    const float inter = smax(iy1 - iy0, 0.0f) * smax(ix1 - ix0, 0.0f);
    return inter / (area_i + area_j - inter);
}

typedef struct { float s; int32_t i; } cand_t;

static int cmp_cand(const void* a, const void* b) {
    const cand_t* x = (const cand_t*)a; const cand_t* y = (const cand_t*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return (x->i > y->i) - (x->i < y->i);        /* lower index first */
}

/* boxes: [n][stride] float32, first 4 columns y0,x0,y1,x1.  cand: optional u8 [n] filter.
 * Returns the number of kept indices written to out (<= max_out). */
int oracle_nms(const float* boxes, int stride, const float* scores, const uint8_t* cand, int n,
               int max_out, float iou_thr, int32_t* out) {
    cand_t* c = (cand_t*)malloc(sizeof(cand_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        if (cand && !cand[i]) continue;
        if (scores[i] > -FLT_MAX) { c[m].s = scores[i]; c[m].i = i; ++m; }   /* NaN, -inf excluded */
    }
    qsort(c, (size_t)m, sizeof(cand_t), cmp_cand);
    int k = 0;
    for (int q = 0; q < m && k < max_out; ++q) {
        const float* bq = boxes + (size_t)c[q].i * stride;
        int ok = 1;
        for (int j = k - 1; j >= 0; --j) {
            if (iou(bq, boxes + (size_t)out[j] * stride) > iou_thr) { ok = 0; break; }
        }
        if (ok) out[k++] = c[q].i;
    }
    free(c);
    return k;
}
