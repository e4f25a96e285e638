// train.hip -- the non-GEMM kernels of the train step (SURVEY.md section 8 rows f2 + f3).
//
//   label_proposals_kernel   models/faster_rcnn.py:421-510 _label_proposals (IoU of proposals vs GT boxes,
//                            class labels, box-delta regression targets, mask/target map)
//   rpn_loss_kernel          models/rpn.py:176-272 class_loss + regression_loss, and their gradient with
//                            respect to the RPN head's pre-activation output (what autograd derives)
//   detector_loss_kernel     models/detector.py:83-155 class_loss + regression_loss and the gradient with
//                            respect to the stacked head logits
//   relu_backward / maxpool2x2_backward / roi_pool_backward / transpose / gather_rows / pack_conv3x3_dgrad
//                            the pieces autograd's backward of vgg16.py:76-96, detector.py:65-78 needs
//   sgd_kernel               torch.optim.SGD.step as configured at __main__.py:98-105 (momentum, weight
//                            decay folded into the gradient, no dampening, no Nesterov)
// Loss sums are accumulated in float64 in a fixed order (single block), then rounded once.
#include "common.h"
#include <cfloat>
#include <cmath>

namespace frcnn {

// ---- proposal labelling -------------------------------------------------------------------------
// One block of 1024 threads; row i < n_props is a proposal, row n_props + j is GT box j (the "fake
// proposals" of faster_rcnn.py:433).  All arithmetic is float32 exactly as math_utils.py:39-63.
__device__ __forceinline__ float iou_f32(const f32x4 a, const f32x4 g)
{
    const float ty = fmaxf(a[0], g[0]), tx = fmaxf(a[1], g[1]);
    const float by = fminf(a[2], g[2]), bx = fminf(a[3], g[3]);
    const float ok = (ty < by && tx < bx) ? 1.0f : 0.0f;
    const float inter = __fmul_rn(ok, __fmul_rn(by - ty, bx - tx));
    const float a1 = __fmul_rn(a[2] - a[0], a[3] - a[1]);
    const float a2 = __fmul_rn(g[2] - g[0], g[3] - g[1]);
    const float uni = __fadd_rn(__fadd_rn(a1, a2), -inter);
    return inter / __fadd_rn(uni, 1e-7f);
}

__global__ __launch_bounds__(1024)
void label_proposals_kernel(const f32x4* __restrict__ props, const int32_t* __restrict__ n_props_p, int max_props,
                            const f32x4* __restrict__ gt, const int32_t* __restrict__ gt_cls, int M, int ncls,
                            float bg_thr, float obj_thr, f32x4 means, f32x4 stds,
                            f32x4* __restrict__ out_props, int32_t* __restrict__ out_cls,
                            float* __restrict__ out_onehot, float* __restrict__ out_deltas,
                            int32_t* __restrict__ out_count)
{
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int n_props = *n_props_p;
    if (n_props > max_props) n_props = max_props;
    const int total = n_props + M;
    const int nd = 4 * (ncls - 1);
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += 1024) {
        const int i = i0 + tid;
        bool keep = false;
        f32x4 box = {0.f, 0.f, 0.f, 0.f};
        float best = 0.f;
        int best_j = 0;
        if (i < total) {
            box = i < n_props ? props[i] : gt[i - n_props];
            best = iou_f32(box, gt[0]);
            for (int j = 1; j < M; ++j) {
                const float v = iou_f32(box, gt[j]);
                if (v > best) { best = v; best_j = j; }       // first maximum wins (torch.argmax on ties)
            }
            keep = best >= bg_thr;
        }
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) wave_cnt[wave] = __popcll(bal);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
        if (keep) {
            const int cls = best < obj_thr ? 0 : gt_cls[best_j];
            const f32x4 g = gt[best_j];
            out_props[pos] = box;
            out_cls[pos] = cls;
            float* oh = out_onehot + (size_t)pos * ncls;
            for (int c = 0; c < ncls; ++c) oh[c] = c == cls ? 1.0f : 0.0f;
            // faster_rcnn.py:486-503
            const float pcy = __fmul_rn(0.5f, __fadd_rn(box[0], box[2])), pcx = __fmul_rn(0.5f, __fadd_rn(box[1], box[3]));
            const float ph = box[2] - box[0], pw = box[3] - box[1];
            const float gcy = __fmul_rn(0.5f, __fadd_rn(g[0], g[2])), gcx = __fmul_rn(0.5f, __fadd_rn(g[1], g[3]));
            const float gh = g[2] - g[0], gw = g[3] - g[1];
            float tgt[4];
            tgt[0] = (gcy - pcy) / ph;
            tgt[1] = (gcx - pcx) / pw;
            tgt[2] = (float)log((double)(gh / ph));
            tgt[3] = (float)log((double)(gw / pw));
#pragma unroll
            for (int k = 0; k < 4; ++k) tgt[k] = __fadd_rn(tgt[k], -means[k]) / stds[k];
            float* dm = out_deltas + (size_t)pos * 2 * nd;     // [2][nd]: mask row, target row
            for (int q = 0; q < nd; ++q) {
                dm[q] = (q >> 2) + 1 == cls ? 1.0f : 0.0f;
                dm[nd + q] = tgt[q & 3];
            }
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int w = 0; w < 16; ++w) s += wave_cnt[w]; base_s += s; }
        __syncthreads();
    }
    if (tid == 0) *out_count = base_s;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int n,
                                   int row_floats, float* __restrict__ dst)
{
    const int r = blockIdx.x;
    if (r >= n) return;
    const float* s = src + (size_t)idx[r] * row_floats;
    float* d = dst + (size_t)r * row_floats;
    for (int i = threadIdx.x; i < row_floats; i += blockDim.x) d[i] = s[i];
}

// ---- block-wide float64 sum (fixed order: lane tree, then waves ascending) ------------------------
__device__ __forceinline__ double block_sum_f64(double v, double* sh /* >= 16 */)
{
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wave] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
    return s;
}

// robust-L1 of rpn.py:258-262 / detector.py:143-147 and its derivative with respect to x
__device__ __forceinline__ float smooth_l1(float x, float sigma_sq, float* d)
{
    const float ax = fabsf(x);
    if (ax < 1.0f / sigma_sq) { *d = __fmul_rn(sigma_sq, x); return __fmul_rn(__fmul_rn(__fmul_rn(0.5f, x), x), sigma_sq); }
    *d = x > 0.f ? 1.0f : -1.0f;
    return ax - 0.5f / sigma_sq;
}

// ---- RPN losses -----------------------------------------------------------------------------------
// head: [P][ld] rows = [9 objectness logits | 36 box deltas | pad]; sample: flat anchor indices
// n = (y*W + x)*9 + k of the mini-batch (faster_rcnn.py:364-419 marks exactly these as trainable);
// rpn_map: [A][6] = (trainable, object, ty, tx, th, tw).  d_head must be zero on entry.
__global__ __launch_bounds__(256)
void rpn_loss_kernel(const float* __restrict__ head, int ld, const int32_t* __restrict__ sample, int n_sample,
                     const float* __restrict__ rpn_map, float* __restrict__ losses, float* __restrict__ d_head)
{
    __shared__ double sh[16];
    const float n_cls = (float)n_sample + 1e-7f;       // rpn.py:207: count_nonzero(mask) + epsilon
    double cls_sum = 0.0, reg_sum = 0.0;
    for (int i = threadIdx.x; i < n_sample; i += 256) {
        const int a = sample[i];
        const int cell = a / 9, k = a - cell * 9;
        const float* hr = head + (size_t)cell * ld;
        const float* gt = rpn_map + (size_t)a * 6;
        const float y = gt[1];
        const float p = 1.0f / (1.0f + expf(-hr[k]));              // rpn.py:89 t.sigmoid
        // F.binary_cross_entropy (log terms clamped to >= -100)
        const float lp = fmaxf(logf(p), -100.0f), lq = fmaxf(logf(1.0f - p), -100.0f);
        cls_sum += (double)(-(y * lp + (1.0f - y) * lq));
        // d/dp = (p - y) / max(p (1-p), 1e-12); sigmoid backward multiplies by p (1-p)
        const float pq = p * (1.0f - p);
        const float dp = (p - y) / fmaxf(pq, 1e-12f) / n_cls;
        if (d_head) d_head[(size_t)cell * ld + k] = dp * pq;
        if (y != 0.f) {                                              // rpn.py:236-238 included * positive
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float x = gt[2 + c] - hr[9 + 4 * k + c];
                float d;
                reg_sum += (double)(y * smooth_l1(x, 9.0f, &d));
                if (d_head) d_head[(size_t)cell * ld + 9 + 4 * k + c] = -(y * d) / n_cls;
            }
        }
    }
    cls_sum = block_sum_f64(cls_sum, sh);
    reg_sum = block_sum_f64(reg_sum, sh);
    if (threadIdx.x == 0) {
        losses[0] = (float)cls_sum / n_cls;
        losses[1] = (float)reg_sum / n_cls;
    }
}

// ---- detector losses ------------------------------------------------------------------------------
// classes [S][ncls] softmax outputs, deltas [S][nd]; gt_onehot [S][ncls]; gt_deltas [S][2][nd].
// d_logits [S][ld] = gradient w.r.t. [class logits | regressor outputs | pad] (pad written as 0).
__global__ __launch_bounds__(256)
void detector_loss_kernel(const float* __restrict__ classes, const float* __restrict__ deltas,
                          const float* __restrict__ gt_onehot, const float* __restrict__ gt_deltas,
                          int S, int ncls, float* __restrict__ losses, float* __restrict__ d_logits, int ld)
{
    __shared__ double sh[16];
    const int nd = 4 * (ncls - 1);
    const float n_f = (float)((double)S + 1e-7);        // detector.py:102,151: python float, then f32 division
    double cls_sum = 0.0, reg_sum = 0.0;
    for (int r = threadIdx.x; r < S; r += 256) {
        const float* p = classes + (size_t)r * ncls;
        const float* y = gt_onehot + (size_t)r * ncls;
        // -(y * log(p + eps)).sum(); gradient wrt p_j = -y_j / (p_j + eps) / N, then softmax backward
        float row = 0.f, dot = 0.f;
        for (int j = 0; j < ncls; ++j) {
            const float pe = p[j] + 1e-7f;
            row += y[j] * logf(pe);
            dot += (-(y[j] / pe) / n_f) * p[j];
        }
        cls_sum += (double)(-row);
        float* dl = d_logits ? d_logits + (size_t)r * ld : nullptr;
        if (dl)
            for (int j = 0; j < ncls; ++j) {
                const float g = -(y[j] / (p[j] + 1e-7f)) / n_f;
                dl[j] = p[j] * (g - dot);
            }
        const float* mask = gt_deltas + (size_t)r * 2 * nd;
        const float* tgt = mask + nd;
        const float* pd = deltas + (size_t)r * nd;
        for (int q = 0; q < nd; ++q) {
            const float x = tgt[q] - pd[q];
            float d;
            const float l = smooth_l1(x, 1.0f, &d);
            reg_sum += (double)(mask[q] * l);
            if (dl) dl[ncls + q] = -(mask[q] * d) / n_f;
        }
        if (dl) for (int q = ncls + nd; q < ld; ++q) dl[q] = 0.f;
    }
    cls_sum = block_sum_f64(cls_sum, sh);
    reg_sum = block_sum_f64(reg_sum, sh);
    if (threadIdx.x == 0) {
        losses[0] = (float)cls_sum / n_f;
        losses[1] = (float)reg_sum / n_f;
    }
}

// ---- elementwise backward pieces --------------------------------------------------------------------
__global__ __launch_bounds__(256)
void relu_backward_kernel(float* __restrict__ dy, const float* __restrict__ y, size_t n4, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        f32x4 g = reinterpret_cast<f32x4*>(dy)[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = v[j] > 0.f ? g[j] : 0.f;
        reinterpret_cast<f32x4*>(dy)[i] = g;
    }
    if (blockIdx.x == 0)
        for (size_t i = 4 * n4 + threadIdx.x; i < n; i += 256) dy[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ __launch_bounds__(256)
void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] += b[i];
}

// x [H][W][C] pre-pool, dy [H/2][W/2][C]; dx zero-filled by the launcher (odd remainders stay 0).
// The gradient goes to the first maximum of the window in (row, column) scan order, as
// F.max_pool2d's backward does with its saved indices.
__global__ __launch_bounds__(256)
void maxpool2x2_backward_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                int H, int W, int C)
{
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const size_t total = (size_t)Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % Wo), oy = (int)(p / Wo);
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        f32x4 v[4];
        size_t off[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            off[q] = (((size_t)(2 * oy + (q >> 1)) * W + 2 * ox + (q & 1)) * C) / 4 + c4;
            v[q] = reinterpret_cast<const f32x4*>(x)[off[q]];
        }
        f32x4 o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int am = 0; float m = v[0][j];
#pragma unroll
            for (int q = 1; q < 4; ++q) if (v[q][j] > m) { m = v[q][j]; am = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][j] = q == am ? g[j] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) reinterpret_cast<f32x4*>(dx)[off[q]] = o[q];
    }
}

// ---- RoI pool backward (deterministic gather form) ------------------------------------------------
struct RoiGeom { int rs_h, rs_w; float bin_h, bin_w; };
__device__ __forceinline__ RoiGeom roi_geom(const f32x4 roi, float scale, int pooled)
{
    RoiGeom g;
    g.rs_h = (int)roundf(roi[0] * scale); g.rs_w = (int)roundf(roi[1] * scale);
    const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
    const int roi_h = max(re_h - g.rs_h + 1, 1), roi_w = max(re_w - g.rs_w + 1, 1);
    g.bin_h = (float)roi_h / (float)pooled; g.bin_w = (float)roi_w / (float)pooled;
    return g;
}
__device__ __forceinline__ void bin_range(int p, float bin, int rs, int limit, int* s, int* e)
{
    int a = (int)floorf((float)p * bin) + rs, b = (int)ceilf((float)(p + 1) * bin) + rs;
    *s = min(max(a, 0), limit); *e = min(max(b, 0), limit);
}

// phase 1: argmax cell (h*fw + w, or -1 for an empty bin) per (roi, ph, pw, channel); first maximum in
// (h, w) scan order with a strict '>' -- the index torchvision's RoIPool forward saves for its backward.
__global__ __launch_bounds__(256)
void roi_pool_argmax_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                            int pooled, float scale, int32_t* __restrict__ argmax)
{
    const int r = blockIdx.x, ph = blockIdx.y, pw = blockIdx.z;      // one block per bin: the gather is latency bound (csrc/roipool.hip)
    const RoiGeom g = roi_geom(reinterpret_cast<const f32x4*>(rois)[r], scale, pooled);
    int hs, he, ws, we;
    bin_range(ph, g.bin_h, g.rs_h, fh, &hs, &he);
    bin_range(pw, g.bin_w, g.rs_w, fw, &ws, &we);
    for (int c = threadIdx.x; c < C; c += 256) {
        float m = -FLT_MAX; int am = -1;
        for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
                const float v = fm[((size_t)h * fw + w) * C + c];
                if (v > m) { m = v; am = h * fw + w; }
            }
        argmax[((size_t)(r * pooled + ph) * pooled + pw) * C + c] = am;
    }
}

// phase 2: one block per feature-map cell; RoIs ascending, bins in (ph, pw) order -> fixed summation order.
// Which bins of which RoIs hold the cell is pure geometry, the same for every thread of the block: 64 RoIs at a time, one lane
// per RoI works it out (a cell can sit in every bin of a RoI smaller than the 7 x 7 grid), an ordered compaction puts the
// (roi, ph, pw) hits in LDS, and then all threads walk only the hits.  (Every thread evaluating all n_rois x 49 bin ranges
// itself: 437 us for 128 RoIs on the 37 x 62 map.)
__global__ __launch_bounds__(256)
void roi_pool_scatter_kernel(const float* __restrict__ rois, int n_rois, int fh, int fw, int C, int pooled, float scale,
                             const int32_t* __restrict__ argmax, const float* __restrict__ dout,
                             float* __restrict__ dfm, int accumulate)
{
    __shared__ int hits[64 * 49];
    __shared__ int n_hits;
    const int cell = blockIdx.x;
    const int h = cell / fw, w = cell - h * fw;
    const int tid = threadIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};          // channels tid, tid+256, ... (C <= 1024)
    for (int r0 = 0; r0 < n_rois; r0 += 64) {
        if (tid < 64) {
            const int r = r0 + tid;
            unsigned phm = 0u, pwm = 0u;           // bins (pooled <= 7) whose range holds the cell's row / column
            if (r < n_rois) {
                const RoiGeom g = roi_geom(reinterpret_cast<const f32x4*>(rois)[r], scale, pooled);
                for (int p = 0; p < pooled; ++p) {
                    int s0, e0;
                    bin_range(p, g.bin_h, g.rs_h, fh, &s0, &e0);
                    if (h >= s0 && h < e0) phm |= 1u << p;
                    bin_range(p, g.bin_w, g.rs_w, fw, &s0, &e0);
                    if (w >= s0 && w < e0) pwm |= 1u << p;
                }
            }
            const int mine = __popc(phm) * __popc(pwm);
            int incl = mine;
            for (int o = 1; o < 64; o <<= 1) {
                const int v = __shfl_up(incl, o);
                if (tid >= o) incl += v;
            }
            int at = incl - mine;
            for (unsigned a = phm; a != 0u; a &= a - 1u) {
                const int ph = __ffs((int)a) - 1;
                for (unsigned b2 = pwm; b2 != 0u; b2 &= b2 - 1u) hits[at++] = (r * pooled + ph) * pooled + (__ffs((int)b2) - 1);
            }
            if (tid == 63) n_hits = incl;
        }
        __syncthreads();
        const int nh = n_hits;
#pragma unroll 2
        for (int i = 0; i < nh; ++i) {
            const size_t base = (size_t)hits[i] * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = tid + 256 * k;
                if (c < C && argmax[base + c] == cell) acc[k] += dout[base + c];
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            float* o = dfm + (size_t)cell * C + c;
            *o = accumulate ? *o + acc[k] : acc[k];
        }
    }
}

// y[c][r] (row stride ldo) = x[r][c] (row stride ldi); columns rows..ldo-1 of y are zero filled
__global__ __launch_bounds__(256)
void transpose_kernel(const float* __restrict__ x, int ldi, float* __restrict__ y, int ldo, int rows, int cols)
{
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + tx;
        tile[j][tx] = (r < rows && c < cols) ? x[(size_t)r * ldi + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + tx;
        if (c < cols && r < ldo) y[(size_t)c * ldo + r] = tile[tx][j];
    }
}

// forward pack [tap][co][ci] -> data-gradient pack [8 - tap][ci][co]: dX = conv3x3(dZ, this)
__global__ __launch_bounds__(256)
void pack_conv3x3_dgrad_kernel(const float* __restrict__ wp, float* __restrict__ wd, int cout, int cin)
{
    const size_t per = (size_t)cout * cin, total = 9 * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int tap = (int)(i / per);
        const size_t rem = i - tap * per;
        const int ci = (int)(rem / cout), co = (int)(rem % cout);        // destination [tap'][ci][co]
        wd[i] = wp[(size_t)(8 - tap) * per + (size_t)co * cin + ci];
    }
}

// wd[tap][ci][co] = wp[tap][co][ci]: the weight pack of the gather kernel's transposed (data-gradient) mode
__global__ __launch_bounds__(256)
void pack_conv_dgrad_kernel(const float* __restrict__ wp, float* __restrict__ wd, int taps, int cout, int cin)
{
    const size_t per = (size_t)cout * cin, total = (size_t)taps * per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int tap = (int)(i / per);
        const size_t rem = i - tap * per;
        const int ci = (int)(rem / cout), co = (int)(rem % cout);
        wd[i] = wp[(size_t)tap * per + (size_t)co * cin + ci];
    }
}

// dst[tap][co][ci] = src[tap][co][ci] * scale[co]: folding a frozen BatchNorm's scale into a weight pack, and the
// chain rule back from the folded weight's gradient to the raw weight's (same factor)
__global__ __launch_bounds__(256)
void scale_rows_kernel(const float* __restrict__ src, const float* __restrict__ scale, float* __restrict__ dst,
                       int taps, int cout, int cin)
{
    const size_t total = (size_t)taps * cout * cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)((i / cin) % cout);
        dst[i] = __fmul_rn(src[i], scale[co]);
    }
}

// frozen BatchNorm (eval mode) as y = x * scale + shift: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
__global__ void bn_scale_shift_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps, int c,
                                      float* __restrict__ scale, float* __restrict__ shift)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= c) return;
    const float sc = gamma[i] / sqrtf(var[i] + eps);
    scale[i] = sc;
    shift[i] = beta[i] - mean[i] * sc;
}

// backward of y = x.mean(-1).mean(-1) (models/resnet.py:117): dx[n][y][x][c] = (dy[n][c] / H) / W
__global__ __launch_bounds__(256)
void spatial_mean_backward_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C)
{
    const size_t total = (size_t)N * H * W * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t n = i / ((size_t)H * W * C);
        dx[i] = (dy[n * C + c] / (float)H) / (float)W;
    }
}

// torch.optim.SGD.step: g += wd * w; buf = first ? g : momentum * buf + g; w -= lr * buf
__global__ __launch_bounds__(256)
void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ buf, size_t n,
                float lr, float momentum, float weight_decay, int first)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float wi = w[i];
        float gi = g[i];
        if (weight_decay != 0.f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, wi));
        float b = gi;
        if (momentum != 0.f) {
            b = first ? gi : __fadd_rn(__fmul_rn(momentum, buf[i]), gi);
            buf[i] = b;
        }
        w[i] = __fadd_rn(wi, -__fmul_rn(lr, b));
    }
}

// The same update of a conv master whose frozen BatchNorm is folded into the convolution (training.py _TrainConv), with the folded pack
// rebuilt in the same pass: folded[tap][co][ci] = w_new * scale[co] (scale_rows_kernel's product) -- one launch per convolution and step
// instead of two (ResNet-101: 91 trainable convolutions).
__global__ __launch_bounds__(256)
void sgd_fold_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ buf, size_t n,
                     float lr, float momentum, float weight_decay, int first, const float* __restrict__ scale,
                     float* __restrict__ folded, int cout, int cin)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float wi = w[i];
        float gi = g[i];
        if (weight_decay != 0.f) gi = __fadd_rn(gi, __fmul_rn(weight_decay, wi));
        float b = gi;
        if (momentum != 0.f) {
            b = first ? gi : __fadd_rn(__fmul_rn(momentum, buf[i]), gi);
            buf[i] = b;
        }
        const float wn = __fadd_rn(wi, -__fmul_rn(lr, b));
        w[i] = wn;
        folded[i] = __fmul_rn(wn, scale[(int)((i / cin) % cout)]);
    }
}

// ---- launchers -----------------------------------------------------------------------------------
static int grid_for(size_t n, int cap = 8192)
{
    size_t b = (n + 255) / 256;
    if (b < 1) b = 1;
    if (b > (size_t)cap) b = cap;
    return (int)b;
}

int launch_label_proposals(const float* props, const int32_t* n_props, int max_props, const float* gt,
                           const int32_t* gt_cls, int M, int ncls, float bg_thr, float obj_thr,
                           const float means[4], const float stds[4], float* out_props, int32_t* out_cls,
                           float* out_onehot, float* out_deltas, int32_t* out_count, hipStream_t s)
{
    if (M < 1 || ncls < 2 || max_props < 0) return FRCNN_EINVAL;
    const f32x4 mn = {means[0], means[1], means[2], means[3]}, sd = {stds[0], stds[1], stds[2], stds[3]};
    hipLaunchKernelGGL(label_proposals_kernel, dim3(1), dim3(1024), 0, s, reinterpret_cast<const f32x4*>(props), n_props,
                       max_props, reinterpret_cast<const f32x4*>(gt), gt_cls, M, ncls, bg_thr, obj_thr, mn, sd,
                       reinterpret_cast<f32x4*>(out_props), out_cls, out_onehot, out_deltas, out_count);
    return check_launch();
}

int launch_gather_rows(const float* src, const int32_t* idx, int n, int row_floats, float* dst, hipStream_t s)
{
    if (n < 0 || row_floats < 1) return FRCNN_EINVAL;
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(row_floats >= 256 ? 256 : 64), 0, s, src, idx, n, row_floats, dst);
    return check_launch();
}

int launch_rpn_loss(const float* head, int ld, int cells, const int32_t* sample, int n_sample, const float* rpn_map,
                    float* losses, float* d_head, hipStream_t s)
{
    if (ld < 45 || cells < 1 || n_sample < 0) return FRCNN_EINVAL;
    if (d_head) FRCNN_HIP_TRY(hipMemsetAsync(d_head, 0, (size_t)cells * ld * sizeof(float), s));
    hipLaunchKernelGGL(rpn_loss_kernel, dim3(1), dim3(256), 0, s, head, ld, sample, n_sample, rpn_map, losses, d_head);
    return check_launch();
}

int launch_detector_loss(const float* classes, const float* deltas, const float* gt_onehot, const float* gt_deltas,
                         int S, int ncls, float* losses, float* d_logits, int ld, hipStream_t s)
{
    if (S < 0 || ncls < 2 || (d_logits && ld < ncls + 4 * (ncls - 1))) return FRCNN_EINVAL;
    hipLaunchKernelGGL(detector_loss_kernel, dim3(1), dim3(256), 0, s, classes, deltas, gt_onehot, gt_deltas, S, ncls,
                       losses, d_logits, ld);
    return check_launch();
}

int launch_relu_backward(float* dy, const float* y, size_t n, hipStream_t s)
{
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(y)) & 15) return FRCNN_EINVAL;
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(relu_backward_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, dy, y, n / 4, n);
    return check_launch();
}

int launch_add_inplace(float* a, const float* b, size_t n, hipStream_t s)
{
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, n);
    return check_launch();
}

int launch_maxpool2x2_backward(const float* x, const float* dy, float* dx, int H, int W, int C, hipStream_t s)
{
    if (H < 2 || W < 2 || C < 4 || (C & 3)) return FRCNN_EINVAL;
    FRCNN_HIP_TRY(hipMemsetAsync(dx, 0, (size_t)H * W * C * sizeof(float), s));
    hipLaunchKernelGGL(maxpool2x2_backward_kernel, dim3(grid_for((size_t)(H / 2) * (W / 2) * (C / 4))), dim3(256), 0, s,
                       x, dy, dx, H, W, C);
    return check_launch();
}

size_t roi_pool_backward_workspace_bytes(int n_rois, int pooled, int C)
{
    return (size_t)n_rois * pooled * pooled * C * sizeof(int32_t);
}

int launch_roi_pool_backward(const float* fm, int fh, int fw, int C, const float* rois, int n_rois, int pooled,
                             float scale, const float* dout, float* dfm, int accumulate, void* ws, size_t ws_bytes,
                             hipStream_t s)
{
    if (fh < 1 || fw < 1 || C < 1 || C > 1024 || n_rois < 0 || pooled < 1 || pooled > 7) return FRCNN_EINVAL;   // hit list: 64 RoIs x 49 bins
    if (n_rois > 0 && (!ws || ws_bytes < roi_pool_backward_workspace_bytes(n_rois, pooled, C))) return FRCNN_EINVAL;
    int32_t* argmax = static_cast<int32_t*>(ws);
    if (n_rois > 0) {
        hipLaunchKernelGGL(roi_pool_argmax_kernel, dim3(n_rois, pooled, pooled), dim3(256), 0, s, fm, fh, fw, C, rois, pooled,
                           scale, argmax);
        int rc = check_launch();
        if (rc) return rc;
    }
    hipLaunchKernelGGL(roi_pool_scatter_kernel, dim3(fh * fw), dim3(256), 0, s, rois, n_rois, fh, fw, C, pooled, scale,
                       argmax, dout, dfm, accumulate);
    return check_launch();
}

int launch_transpose(const float* x, int ldi, float* y, int ldo, int rows, int cols, hipStream_t s)
{
    if (rows < 1 || cols < 1 || ldi < cols || ldo < rows) return FRCNN_EINVAL;
    hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(cols, 32), cdiv(ldo, 32)), dim3(256), 0, s, x, ldi, y, ldo, rows, cols);
    return check_launch();
}

int launch_pack_conv3x3_dgrad(const float* wp, float* wd, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pack_conv3x3_dgrad_kernel, dim3(grid_for((size_t)9 * cout * cin)), dim3(256), 0, s, wp, wd, cout, cin);
    return check_launch();
}

int launch_pack_conv_dgrad(const float* wp, float* wd, int taps, int cout, int cin, hipStream_t s)
{
    if (taps < 1 || cout < 1 || cin < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pack_conv_dgrad_kernel, dim3(grid_for((size_t)taps * cout * cin)), dim3(256), 0, s, wp, wd, taps, cout, cin);
    return check_launch();
}

int launch_scale_rows(const float* src, const float* scale, float* dst, int taps, int cout, int cin, hipStream_t s)
{
    if (taps < 1 || cout < 1 || cin < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((size_t)taps * cout * cin)), dim3(256), 0, s, src, scale, dst, taps, cout, cin);
    return check_launch();
}

int launch_bn_scale_shift(const float* gamma, const float* beta, const float* mean, const float* var, float eps, int c,
                          float* scale, float* shift, hipStream_t s)
{
    if (c < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(bn_scale_shift_kernel, dim3(cdiv(c, 256)), dim3(256), 0, s, gamma, beta, mean, var, eps, c, scale, shift);
    return check_launch();
}

int launch_spatial_mean_backward(const float* dy, float* dx, int N, int H, int W, int c, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || c < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(spatial_mean_backward_kernel, dim3(grid_for((size_t)N * H * W * c)), dim3(256), 0, s, dy, dx, N, H, W, c);
    return check_launch();
}

int launch_sgd(float* w, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, int first,
               hipStream_t s)
{
    if (momentum != 0.f && !buf) return FRCNN_EINVAL;
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, w, g, buf, n, lr, momentum, weight_decay, first);
    return check_launch();
}

int launch_sgd_fold(float* w, const float* g, float* buf, size_t n, float lr, float momentum, float weight_decay, int first,
                    const float* scale, float* folded, int cout, int cin, hipStream_t s)
{
    if (momentum != 0.f && !buf) return FRCNN_EINVAL;
    if (cout < 1 || cin < 1 || n % ((size_t)cout * cin) != 0) return FRCNN_EINVAL;
    if (n == 0) return FRCNN_OK;
    hipLaunchKernelGGL(sgd_fold_kernel, dim3(grid_for(n, 16384)), dim3(256), 0, s, w, g, buf, n, lr, momentum, weight_decay, first, scale, folded,
                       cout, cin);
    return check_launch();
}

}  // namespace frcnn
