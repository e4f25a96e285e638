// roipool.hip -- RoI max pooling + anchor generation (the two small "geometry" kernels).
//
// roi_pool_kernel replaces torchvision.ops.RoIPool((7,7), 1/16) as used at
// models/detector.py:27,72, including the (y1,x1,y2,x2) -> (b,x1,y1,x2,y2) shuffle of :65-69.
// torchvision semantics (restated; torchvision itself is not vendored in the reference):
//   rs = round(coord * scale) (C round(): half away from zero), re likewise;
//   roi_w = max(re_w - rs_w + 1, 1); bin = roi_w / pooled (float);
//   bin window [floor(p*bin) + rs, ceil((p+1)*bin) + rs) clipped to the map; empty -> 0, else max.
// The feature map is NHWC, so a bin-window pixel is one contiguous C-float run: the block
// (roi, ph) sweeps it with float4 lanes (coalesced 2 KB rows for C = 512), output is
// [roi][ph][pw][C] which the repacked fc1 weight consumes directly (frcnn_pack_fc_chw_to_hwc).
//
// anchors_kernel replaces models/anchors.py:43-135 bit-exactly: float64 arithmetic on a
// float32-rounded cell centre, one final cast to float32.
#include "common.h"
#include <cfloat>
#include <cmath>

namespace frcnn {

__global__ __launch_bounds__(256)
void roi_pool_kernel(const float* __restrict__ fm, int fh, int fw, int C,
                     const float* __restrict__ rois, const int32_t* __restrict__ n_rois,
                     int pooled, float scale, float* __restrict__ out)
{
    const int r = blockIdx.x, ph = blockIdx.y;
    const int C4 = C >> 2;
    f32x4* orow = reinterpret_cast<f32x4*>(out + ((size_t)(r * pooled + ph) * pooled) * C);
    if (r >= *n_rois) {
        for (int i = threadIdx.x; i < pooled * C4; i += 256) orow[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];     // y1, x1, y2, x2
    const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
    const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
    const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
    const float bin_h = (float)roi_h / (float)pooled, bin_w = (float)roi_w / (float)pooled;
    int hs = (int)floorf((float)ph * bin_h) + rs_h;
    int he = (int)ceilf((float)(ph + 1) * bin_h) + rs_h;
    hs = min(max(hs, 0), fh); he = min(max(he, 0), fh);
    for (int pw = 0; pw < pooled; ++pw) {
        int ws = (int)floorf((float)pw * bin_w) + rs_w;
        int we = (int)ceilf((float)(pw + 1) * bin_w) + rs_w;
        ws = min(max(ws, 0), fw); we = min(max(we, 0), fw);
        const bool empty = (he <= hs) || (we <= ws);
        for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
            f32x4 m = empty ? f32x4{0.f, 0.f, 0.f, 0.f} : f32x4{-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
            for (int h = hs; h < he; ++h) {
                const f32x4* p = reinterpret_cast<const f32x4*>(fm + ((size_t)h * fw + ws) * C) + c4;
                for (int w = ws; w < we; ++w, p += C4) {
                    const f32x4 v = *p;
#pragma unroll
                    for (int j = 0; j < 4; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];
                }
            }
            orow[pw * C4 + c4] = m;
        }
    }
}

struct AnchorSizes { double h[9]; double w[9]; };

__global__ __launch_bounds__(256)
void anchors_kernel(AnchorSizes sz, int image_h, int image_w, int fh, int fw, int feature_pixels,
                    float* __restrict__ anchor_map, float* __restrict__ valid_map)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int A = fh * fw * 9;
    if (n >= A) return;
    const int cell = n / 9, k = n - cell * 9;
    const int cy_i = cell / fw, cx_i = cell - cy_i * fw;
    // anchors.py:105 cell*feature_pixels + 0.5*feature_pixels (float64), :118 .astype(float32)
    const float cyf = (float)((double)(cy_i * feature_pixels) + 0.5 * (double)feature_pixels);
    const float cxf = (float)((double)(cx_i * feature_pixels) + 0.5 * (double)feature_pixels);
    // anchors.py:92-93,118: float32 centre + float64 template -> float64 corners
    const double y1 = (double)cyf + (-0.5 * sz.h[k]);
    const double x1 = (double)cxf + (-0.5 * sz.w[k]);
    const double y2 = (double)cyf + (0.5 * sz.h[k]);
    const double x2 = (double)cxf + (0.5 * sz.w[k]);
    const bool valid = (y1 >= 0.0) && (x1 >= 0.0) && (y2 <= (double)image_h) && (x2 <= (double)image_w);
    f32x4 a;
    a[0] = (float)(0.5 * (y1 + y2));
    a[1] = (float)(0.5 * (x1 + x2));
    a[2] = (float)(y2 - y1);
    a[3] = (float)(x2 - x1);
    reinterpret_cast<f32x4*>(anchor_map)[n] = a;
    valid_map[n] = valid ? 1.0f : 0.0f;
}

int launch_roi_pool(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois,
                    int max_rois, int pooled, float scale, float* out, hipStream_t s)
{
    if (fh < 1 || fw < 1 || c < 4 || c % 4 != 0 || max_rois < 1 || pooled < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(roi_pool_kernel, dim3(max_rois, pooled), dim3(256), 0, s, fm, fh, fw, c, rois,
                       n_rois, pooled, scale, out);
    return check_launch();
}

int launch_anchors(int image_h, int image_w, int fh, int fw, int feature_pixels,
                   float* anchor_map, float* valid_map, hipStream_t s)
{
    if (fh < 1 || fw < 1 || feature_pixels < 1) return FRCNN_EINVAL;
    // anchors.py:25-41: areas x aspect ratios, k = area-major, aspect-minor; math.sqrt in float64.
    static const double areas[3] = {128.0 * 128.0, 256.0 * 256.0, 512.0 * 512.0};
    static const double aspects[3] = {0.5, 1.0, 2.0};
    AnchorSizes sz;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            const double root = std::sqrt(areas[i] / aspects[j]);
            sz.h[i * 3 + j] = aspects[j] * root;
            sz.w[i * 3 + j] = root;
        }
    const int A = fh * fw * 9;
    hipLaunchKernelGGL(anchors_kernel, dim3(cdiv(A, 256)), dim3(256), 0, s, sz, image_h, image_w, fh, fw,
                       feature_pixels, anchor_map, valid_map);
    return check_launch();
}

}  // namespace frcnn
