// roialign.hip -- RoIAlign forward and backward on an NHWC feature map (torchvision.ops.roi_align semantics).
//
// BASELINE.json's north_star / configs[4] name RoIAlign; the reference itself pools with torchvision.ops.RoIPool
// (pytorch/FasterRCNN/models/detector.py:16,27,72), so this is the "beyond the reference" pooling option of
// DetectorNetwork(pooling="align").  torchvision 0.15 is a third-party dependency that is not in /root/reference: the algorithm
// is restated from its published kernel (roi_align_kernel.cpp / .cu) -- oracle/frcnn_oracle.py: roi_align_weights states it --
// and the parity of these kernels is against that restatement (unpinned, like nms and RoIPool).
//
//   offset = aligned ? 0.5 : 0;  start = coord * scale - offset;  size = end - start (>= 1 unless aligned);  bin = size / P;
//   grid = sampling_ratio > 0 ? sampling_ratio : ceil(size / P);  count = max(grid_h * grid_w, 1);
//   out[r][ph][pw][c] = (1 / count) * sum_{iy, ix} bilinear(fm[.][.][c], y(ph, iy), x(pw, ix))
//   bilinear: 0 outside [-1, H] x [-1, W]; coordinates clamped to >= 0; at the last row / column low = high = H - 1.
//
// Forward: block = (roi, ph), lanes over channel quads (a pixel's channels are contiguous in NHWC: every sample is four
// coalesced 16-byte loads per lane).  float32, the operation order of torchvision's CPU kernel (separately rounded
// multiplies and adds: the library is built with -ffp-contract=off).
// Backward: torchvision scatters with atomicAdd (run-to-run different sums); here the gradient is GATHERED, one block per
// feature-map cell: the block walks the RoIs in ascending order, finds the samples whose bilinear footprint contains its
// cell from the RoI's geometry (uniform across the block: scalar work) and sums g * w / count in that fixed order --
// deterministic, no atomics, no scratch.
#include "common.h"

namespace frcnn {

struct RoiGeom { float start_h, start_w, bin_h, bin_w; int grid_h, grid_w; float count; };

__device__ __forceinline__ RoiGeom roi_geom(const f32x4 roi /* y1, x1, y2, x2 */, float scale, int pooled, int sampling_ratio, int aligned)
{
    RoiGeom g;
    const float offset = aligned ? 0.5f : 0.0f;
    g.start_w = roi[1] * scale - offset;
    g.start_h = roi[0] * scale - offset;
    const float end_w = roi[3] * scale - offset, end_h = roi[2] * scale - offset;
    float roi_w = end_w - g.start_w, roi_h = end_h - g.start_h;
    if (!aligned) { roi_w = fmaxf(roi_w, 1.0f); roi_h = fmaxf(roi_h, 1.0f); }
    g.bin_h = roi_h / (float)pooled;
    g.bin_w = roi_w / (float)pooled;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_h / (float)pooled);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(roi_w / (float)pooled);
    const int cnt = g.grid_h * g.grid_w;
    g.count = (float)(cnt > 1 ? cnt : 1);
    return g;
}

// one coordinate of bilinear_interpolate: returns false when the sample contributes nothing
__device__ __forceinline__ bool axis_weights(float v, int n, int& low, int& high, float& wl, float& wh)
{
    if (v < -1.0f || v > (float)n) return false;
    if (v <= 0.f) v = 0.f;
    low = (int)v;
    if (low >= n - 1) { high = low = n - 1; v = (float)low; }
    else high = low + 1;
    wh = v - (float)low;
    wl = 1.0f - wh;
    return true;
}

// rois: [max_rois][4] (y1, x1, y2, x2) as `forward` produces them; rows >= *n_rois are written as zeros.  out: [max_rois][P][P][C].
__global__ __launch_bounds__(256)
void roi_align_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                      const int32_t* __restrict__ n_rois, int pooled, float scale, int sampling_ratio, int aligned,
                      float* __restrict__ out)
{
    const int r = blockIdx.x, ph = blockIdx.y;
    const int C4 = C >> 2;
    f32x4* orow = reinterpret_cast<f32x4*>(out + ((size_t)r * pooled + ph) * pooled * C);
    if (r >= *n_rois) {
        for (int i = threadIdx.x; i < pooled * C4; i += 256) orow[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        return;
    }
    const RoiGeom g = roi_geom(reinterpret_cast<const f32x4*>(rois)[r], scale, pooled, sampling_ratio, aligned);
    for (int pw = 0; pw < pooled; ++pw) {
        for (int c4 = threadIdx.x; c4 < C4; c4 += 256) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int iy = 0; iy < g.grid_h; ++iy) {
                const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
                int yl, yh; float hy, ly;
                const bool yok = axis_weights(y, fh, yl, yh, hy, ly);
                for (int ix = 0; ix < g.grid_w; ++ix) {
                    const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
                    int xl, xh; float hx, lx;
                    if (!yok || !axis_weights(x, fw, xl, xh, hx, lx)) continue;
                    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                    const f32x4 v1 = reinterpret_cast<const f32x4*>(fm + ((size_t)yl * fw + xl) * C)[c4];
                    const f32x4 v2 = reinterpret_cast<const f32x4*>(fm + ((size_t)yl * fw + xh) * C)[c4];
                    const f32x4 v3 = reinterpret_cast<const f32x4*>(fm + ((size_t)yh * fw + xl) * C)[c4];
                    const f32x4 v4 = reinterpret_cast<const f32x4*>(fm + ((size_t)yh * fw + xh) * C)[c4];
                    acc = acc + (((v1 * w1 + v2 * w2) + v3 * w3) + v4 * w4);
                }
            }
            orow[pw * C4 + c4] = acc / g.count;
        }
    }
}

// dfm[y][x][c] (+)= sum over rois (ascending), bins and samples whose footprint holds (y, x) of dout[r][ph][pw][c] * w / count.
// One block per feature-map cell; the search over rois / sample rows / sample columns is uniform across the block.
__global__ __launch_bounds__(256)
void roi_align_backward_kernel(const float* __restrict__ rois, int n_rois, int fh, int fw, int C, int pooled, float scale,
                               int sampling_ratio, int aligned, const float* __restrict__ dout, float* __restrict__ dfm, int accumulate)
{
    const int cell = blockIdx.x;
    const int cy = cell / fw, cx = cell - cy * fw;
    const int C4 = C >> 2;
    constexpr int MAXH = 64;                     // sample rows / columns of one roi that can touch one cell: <= 2 * pooled * grid; capped
    __shared__ int s_p[2][MAXH];                 // [0]: ph of the hit rows, [1]: pw of the hit columns
    __shared__ float s_w[2][MAXH];
    f32x4* const drow = reinterpret_cast<f32x4*>(dfm + (size_t)cell * C);
    for (int c4base = 0; c4base < C4; c4base += 256) {
        const int c4 = c4base + threadIdx.x;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (accumulate && c4 < C4) acc = drow[c4];
        for (int r = 0; r < n_rois; ++r) {
            const RoiGeom g = roi_geom(reinterpret_cast<const f32x4*>(rois)[r], scale, pooled, sampling_ratio, aligned);
            // quick reject: the footprints of all samples lie within [start - 1, start + size + 1]
            const float end_h = g.start_h + g.bin_h * (float)pooled, end_w = g.start_w + g.bin_w * (float)pooled;
            if ((float)cy < g.start_h - 2.f || (float)cy > end_h + 2.f || (float)cx < g.start_w - 2.f || (float)cx > end_w + 2.f) continue;
            // hit lists (identical in every thread; thread 0 publishes them so that the inner loop reads LDS, not registers)
            int nh = 0, nw = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                for (int ph = 0; ph < pooled; ++ph)
                    for (int iy = 0; iy < g.grid_h; ++iy) {
                        const float y = g.start_h + (float)ph * g.bin_h + ((float)iy + 0.5f) * g.bin_h / (float)g.grid_h;
                        int lo, hi; float wl, wh;
                        if (!axis_weights(y, fh, lo, hi, wl, wh)) continue;
                        float wgt = 0.f;
                        bool hit = false;
                        if (lo == cy) { wgt = wl; hit = true; }
                        if (hi == cy) { wgt = hit ? wgt + wh : wh; hit = true; }
                        if (hit && nh < MAXH - 1) { s_p[0][nh] = ph; s_w[0][nh] = wgt; ++nh; }
                    }
                for (int pw = 0; pw < pooled; ++pw)
                    for (int ix = 0; ix < g.grid_w; ++ix) {
                        const float x = g.start_w + (float)pw * g.bin_w + ((float)ix + 0.5f) * g.bin_w / (float)g.grid_w;
                        int lo, hi; float wl, wh;
                        if (!axis_weights(x, fw, lo, hi, wl, wh)) continue;
                        float wgt = 0.f;
                        bool hit = false;
                        if (lo == cx) { wgt = wl; hit = true; }
                        if (hi == cx) { wgt = hit ? wgt + wh : wh; hit = true; }
                        if (hit && nw < MAXH - 1) { s_p[1][nw] = pw; s_w[1][nw] = wgt; ++nw; }
                    }
                s_p[0][MAXH - 1] = nh;           // counts ride in the last slots (nh, nw < MAXH by the cap above)
                s_p[1][MAXH - 1] = nw;
            }
            __syncthreads();
            nh = s_p[0][MAXH - 1];
            nw = s_p[1][MAXH - 1];
            if (nh == 0 || nw == 0 || c4 >= C4) continue;
            const float* dr = dout + (size_t)r * pooled * pooled * C;
            for (int a = 0; a < nh; ++a) {
                const int ph = s_p[0][a];
                const float wy = s_w[0][a];
                for (int b = 0; b < nw; ++b) {
                    const float w = wy * s_w[1][b];
                    const f32x4 gv = reinterpret_cast<const f32x4*>(dr + ((size_t)ph * pooled + s_p[1][b]) * C)[c4];
                    acc = acc + (gv * w) / g.count;
                }
            }
        }
        if (c4 < C4) drow[c4] = acc;
    }
}

int launch_roi_align(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois, int max_rois, int pooled,
                     float scale, int sampling_ratio, int aligned, float* out, hipStream_t s)
{
    if (fh < 1 || fw < 1 || c < 4 || c % 4 != 0 || max_rois < 1 || pooled < 1 || pooled > 14 || sampling_ratio > 2) return FRCNN_EINVAL;
    hipLaunchKernelGGL(roi_align_kernel, dim3(max_rois, pooled), dim3(256), 0, s, fm, fh, fw, c, rois, n_rois, pooled, scale,
                       sampling_ratio, aligned ? 1 : 0, out);
    return check_launch();
}

int launch_roi_align_backward(const float* rois, int n_rois, int fh, int fw, int c, int pooled, float scale, int sampling_ratio,
                              int aligned, const float* dout, float* dfm, int accumulate, hipStream_t s)
{
    // sampling_ratio <= 2 and pooled <= 14 keep the per-cell hit lists within their capacity (2 * 14 * 2 + slack < 64)
    if (fh < 1 || fw < 1 || c < 4 || c % 4 != 0 || n_rois < 0 || pooled < 1 || pooled > 14 || sampling_ratio > 2)
        return FRCNN_EINVAL;
    hipLaunchKernelGGL(roi_align_backward_kernel, dim3(fh * fw), dim3(256), 0, s, rois, n_rois, fh, fw, c, pooled, scale,
                       sampling_ratio, aligned ? 1 : 0, dout, dfm, accumulate ? 1 : 0);
    return check_launch();
}

}  // namespace frcnn
