// targets.hip -- RPN ground-truth labelling ("anchor <-> GT IoU matching"), SURVEY.md section 8 row f2.
// Replaces models/anchors.py:137-262 generate_rpn_map:
//   :183-190 anchor corners from (cy,cx,h,w) in float32, widened to float64
//   :199     IoU (N x M) in float64 via math_utils.intersection_over_union (:13-37, eps 1e-7)
//   :204     invalid anchors -> IoU -1
//   :215-218 max/argmax per anchor, max per GT box, anchors attaining a GT box's max
//   :221-227 labels: < 0.3 background, >= 0.7 object, best-per-GT object, else ignored
//   :243-246 regression targets in float32: (gt_c - a_c)/a_hw, log(gt_hw/a_hw)
//   :249-257 map (A,6) = (trainable, object, ty, tx, th, tw); ordered lists of object / background anchors
// Three launches: per-GT max (wave-reduced, one ordered-u64 atomicMax per wave), per-anchor
// labels + targets (IoUs recomputed with the identical instruction sequence, so the float64
// equality test of :218 is exact), single-block order-preserving compaction of the two lists.
#include "common.h"

namespace frcnn {

typedef unsigned long long u64;

__device__ __forceinline__ u64 ordered_f64(double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double from_ordered_f64(u64 o)
{
    const u64 b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    return __longlong_as_double((long long)b);
}

struct AnchorBox { double y1, x1, y2, x2, area; };

__device__ __forceinline__ AnchorBox anchor_corners(const f32x4 a)
{
    AnchorBox b;
    b.y1 = (double)(a[0] - 0.5f * a[2]);      // float32 arithmetic, then widened (anchors.py:188-189)
    b.x1 = (double)(a[1] - 0.5f * a[3]);
    b.y2 = (double)(a[0] + 0.5f * a[2]);
    b.x2 = (double)(a[1] + 0.5f * a[3]);
    b.area = __dmul_rn(b.y2 - b.y1, b.x2 - b.x1);
    return b;
}

// math_utils.py:29-37 for one (anchor, gt) pair; gt corners are float32 values widened to float64.
__device__ __forceinline__ double iou_f64(const AnchorBox& a, const f32x4 g)
{
    const double gy1 = (double)g[0], gx1 = (double)g[1], gy2 = (double)g[2], gx2 = (double)g[3];
    const double ty = fmax(a.y1, gy1), tx = fmax(a.x1, gx1);
    const double by = fmin(a.y2, gy2), bx = fmin(a.x2, gx2);
    const bool ok = (ty < by) && (tx < bx);
    const double inter = ok ? __dmul_rn(by - ty, bx - tx) : 0.0;
    const double garea = __dmul_rn(gy2 - gy1, gx2 - gx1);
    const double uni = __dadd_rn(__dadd_rn(a.area, garea), -inter);
    return inter / __dadd_rn(uni, 1e-7);
}

__global__ __launch_bounds__(256)
void rpn_gt_max_kernel(const f32x4* __restrict__ anchors, const float* __restrict__ valid, int A,
                       const f32x4* __restrict__ gt, int M, u64* __restrict__ gt_max)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const bool live = n < A;
    const bool ok = live && valid[n] != 0.f;
    AnchorBox a = {};
    if (live) a = anchor_corners(anchors[n]);
    for (int m = 0; m < M; ++m) {
        const double v = live ? (ok ? iou_f64(a, gt[m]) : -1.0) : -2.0;
        u64 o = ordered_f64(v);
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)(o & 0xFFFFFFFFull), s);
            const unsigned hi = __shfl_xor((unsigned)(o >> 32), s);
            const u64 other = ((u64)hi << 32) | lo;
            o = other > o ? other : o;
        }
        if ((threadIdx.x & 63) == 0) atomicMax(&gt_max[m], o);
    }
}

__global__ __launch_bounds__(256)
void rpn_targets_kernel(const f32x4* __restrict__ anchors, const float* __restrict__ valid, int A,
                        const f32x4* __restrict__ gt, int M, const u64* __restrict__ gt_max,
                        double obj_thr, double bg_thr, float* __restrict__ rpn_map)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= A) return;
    const f32x4 am = anchors[n];
    const AnchorBox a = anchor_corners(am);
    const bool ok = valid[n] != 0.f;
    double best = -INFINITY;
    int best_m = 0;
    bool highest = false;
    for (int m = 0; m < M; ++m) {
        const double v = ok ? iou_f64(a, gt[m]) : -1.0;
        if (v > best) { best = v; best_m = m; }          // np.argmax: first maximum
        if (v == from_ordered_f64(gt_max[m])) highest = true;
    }
    int objectness = -1;
    if (best < bg_thr) objectness = 0;
    if (best >= obj_thr) objectness = 1;
    if (highest) objectness = 1;
    const float enable = objectness >= 0 ? 1.f : 0.f;
    if (objectness < 0) objectness = 0;
    const f32x4 g = gt[best_m];
    const float gcy = 0.5f * (g[0] + g[2]), gcx = 0.5f * (g[1] + g[3]);      // float32 (anchors.py:180-181)
    const float gh = g[2] - g[0], gw = g[3] - g[1];
    float* o = rpn_map + (size_t)n * 6;
    o[0] = valid[n] * enable;
    o[1] = (float)objectness;
    o[2] = (gcy - am[0]) / am[2];
    o[3] = (gcx - am[1]) / am[3];
    o[4] = logf(gh / am[2]);
    o[5] = logf(gw / am[3]);
}

// single block, 1024 threads: ascending lists of object / background anchors
__global__ __launch_bounds__(1024)
void rpn_lists_kernel(const float* __restrict__ rpn_map, int A, int32_t* __restrict__ obj_idx,
                      int32_t* __restrict__ bg_idx, int32_t* __restrict__ counts)
{
    __shared__ int wave_tot[2][16];
    const int tid = threadIdx.x;
    const int per = (A + 1023) / 1024;
    const int beg = tid * per, end = min(beg + per, A);
    int c_obj = 0, c_bg = 0;
    for (int n = beg; n < end; ++n) {
        const float tr = rpn_map[(size_t)n * 6], ob = rpn_map[(size_t)n * 6 + 1];
        c_obj += (ob > 0.f && tr > 0.f);
        c_bg += (ob == 0.f && tr > 0.f);
    }
    int i_obj = c_obj, i_bg = c_bg;
    for (int o = 1; o < 64; o <<= 1) {
        const int v1 = __shfl_up(i_obj, o), v2 = __shfl_up(i_bg, o);
        if ((tid & 63) >= o) { i_obj += v1; i_bg += v2; }
    }
    if ((tid & 63) == 63) { wave_tot[0][tid >> 6] = i_obj; wave_tot[1][tid >> 6] = i_bg; }
    __syncthreads();
    int off_obj = 0, off_bg = 0;
    for (int w = 0; w < (tid >> 6); ++w) { off_obj += wave_tot[0][w]; off_bg += wave_tot[1][w]; }
    int p_obj = off_obj + i_obj - c_obj, p_bg = off_bg + i_bg - c_bg;
    for (int n = beg; n < end; ++n) {
        const float tr = rpn_map[(size_t)n * 6], ob = rpn_map[(size_t)n * 6 + 1];
        if (ob > 0.f && tr > 0.f) obj_idx[p_obj++] = n;
        if (ob == 0.f && tr > 0.f) bg_idx[p_bg++] = n;
    }
    if (tid == 1023) { counts[0] = p_obj; counts[1] = p_bg; }
}

int launch_rpn_targets(const float* anchor_map, const float* valid_map, int A, const float* gt, int M,
                       double obj_thr, double bg_thr, float* rpn_map, int32_t* obj_idx, int32_t* bg_idx,
                       int32_t* counts, void* ws, hipStream_t s)
{
    if (A < 1 || M < 1) return FRCNN_EINVAL;
    u64* gt_max = static_cast<u64*>(ws);
    FRCNN_HIP_TRY(hipMemsetAsync(gt_max, 0, (size_t)M * sizeof(u64), s));       // ordered(anything) > 0
    hipLaunchKernelGGL(rpn_gt_max_kernel, dim3(cdiv(A, 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(anchor_map),
                       valid_map, A, reinterpret_cast<const f32x4*>(gt), M, gt_max);
    int rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(rpn_targets_kernel, dim3(cdiv(A, 256)), dim3(256), 0, s, reinterpret_cast<const f32x4*>(anchor_map),
                       valid_map, A, reinterpret_cast<const f32x4*>(gt), M, (const u64*)gt_max, obj_thr, bg_thr, rpn_map);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(rpn_lists_kernel, dim3(1), dim3(1024), 0, s, (const float*)rpn_map, A, obj_idx, bg_idx, counts);
    return check_launch();
}

}  // namespace frcnn
