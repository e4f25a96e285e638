// preprocess.hip -- image preprocessing on the device, SURVEY.md section 8 row f1.
// Replaces datasets/image.py:92-100 + :43-57 of the reference: PIL `Image.resize((w, h), BILINEAR)`
// of the decoded 8-bit RGB image followed by channel re-ordering, scaling and mean/std
// normalisation into the float32 (3, h, w) tensor the model consumes.
//
// PIL's 8-bit resampler (libImaging/Resample.c) is an integer algorithm and is reproduced bit for
// bit: per output coordinate a triangle filter of support max(scale, 1) (antialiasing when
// shrinking) is evaluated in double precision, normalised, converted to 22-bit fixed point with
// round-half-away, applied as sum(pixel*k) + 2^21 >> 22 and clipped to 0..255; horizontal pass
// first (8-bit intermediate), then vertical.  The normalisation is the reference's float32
// sequence: x *= scaling; x = (x - mean) / std.
#include "common.h"

namespace frcnn {

static constexpr int PRECISION_BITS = 32 - 8 - 2;

// One thread per output coordinate: bounds[xx] = (first input index, tap count), kk[xx][ksize].
__global__ void resample_coeffs_kernel(int in_size, int out_size, int ksize, int* __restrict__ bounds,
                                       int* __restrict__ kk)
{
    const int xx = blockIdx.x * 256 + threadIdx.x;
    if (xx >= out_size) return;
    const double scale = (double)in_size / (double)out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 1.0 * filterscale;                 // bilinear: support 1
    const double center = (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double a = (x + xmin - center + 0.5) * ss;
        if (a < 0.0) a = -a;
        ww += a < 1.0 ? 1.0 - a : 0.0;
    }
    int* k = kk + (size_t)xx * ksize;
    for (int x = 0; x < ksize; ++x) {
        double w = 0.0;
        if (x < xmax) {
            double a = (x + xmin - center + 0.5) * ss;
            if (a < 0.0) a = -a;
            w = a < 1.0 ? 1.0 - a : 0.0;
            if (ww != 0.0) w /= ww;
        }
        k[x] = w < 0.0 ? (int)(-0.5 + w * (double)(1 << PRECISION_BITS)) : (int)(0.5 + w * (double)(1 << PRECISION_BITS));
    }
    bounds[xx * 2 + 0] = xmin;
    bounds[xx * 2 + 1] = xmax;
}

__device__ __forceinline__ unsigned char clip8(int v)
{
    v >>= PRECISION_BITS;
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: src u8 [H][W][3] -> tmp u8 [H][Wo][3]; `flip` mirrors the source columns first
// (Image.transpose(FLIP_LEFT_RIGHT), image.py:90-91)
__global__ __launch_bounds__(256)
void resample_h_kernel(const unsigned char* __restrict__ src, int H, int W, int Wo, int ksize,
                       const int* __restrict__ bounds, const int* __restrict__ kk, int flip,
                       unsigned char* __restrict__ tmp)
{
    const size_t total = (size_t)H * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % Wo);
        const int y = (int)(i / Wo);
        const int xmin = bounds[xx * 2], n = bounds[xx * 2 + 1];
        const int* k = kk + (size_t)xx * ksize;
        int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < n; ++x) {
            int sx = xmin + x;
            if (flip) sx = W - 1 - sx;
            const unsigned char* p = src + ((size_t)y * W + sx) * 3;
            s0 += p[0] * k[x]; s1 += p[1] * k[x]; s2 += p[2] * k[x];
        }
        unsigned char* o = tmp + i * 3;
        o[0] = clip8(s0); o[1] = clip8(s1); o[2] = clip8(s2);
    }
}

struct Norm { float scaling; float mean[3]; float stdv[3]; int src_channel[3]; };

// vertical pass + normalisation: tmp u8 [H][Wo][3] -> out f32 [3][Ho][Wo]
// (optionally also the resized 8-bit image [Ho][Wo][3], what PIL returns)
__global__ __launch_bounds__(256)
void resample_v_norm_kernel(const unsigned char* __restrict__ tmp, int H, int Wo, int Ho, int ksize,
                            const int* __restrict__ bounds, const int* __restrict__ kk, Norm nm,
                            float* __restrict__ out, unsigned char* __restrict__ out_u8)
{
    const size_t total = (size_t)Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int xx = (int)(i % Wo);
        const int yy = (int)(i / Wo);
        const int ymin = bounds[yy * 2], n = bounds[yy * 2 + 1];
        const int* k = kk + (size_t)yy * ksize;
        int s[3] = {1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1), 1 << (PRECISION_BITS - 1)};
        for (int y = 0; y < n; ++y) {
            const unsigned char* p = tmp + ((size_t)(ymin + y) * Wo + xx) * 3;
            s[0] += p[0] * k[y]; s[1] += p[1] * k[y]; s[2] += p[2] * k[y];
        }
        unsigned char px[3] = {clip8(s[0]), clip8(s[1]), clip8(s[2])};
        if (out_u8) { out_u8[i * 3 + 0] = px[0]; out_u8[i * 3 + 1] = px[1]; out_u8[i * 3 + 2] = px[2]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = (float)px[nm.src_channel[c]];
            v = v * nm.scaling;
            v = (v - nm.mean[c]) / nm.stdv[c];
            out[((size_t)c * Ho + yy) * Wo + xx] = v;
        }
    }
}

static int ksize_for(int in_size, int out_size)
{
    double scale = (double)in_size / (double)out_size;
    if (scale < 1.0) scale = 1.0;
    return (int)ceil(1.0 * scale) * 2 + 1;
}

size_t preprocess_workspace_bytes(int H, int W, int Ho, int Wo)
{
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1) return 0;
    size_t b = (size_t)H * Wo * 3;                                   // intermediate 8-bit image
    b = (b + 255) / 256 * 256;
    b += ((size_t)Wo * (2 + ksize_for(W, Wo)) + (size_t)Ho * (2 + ksize_for(H, Ho))) * sizeof(int) + 1024;
    return b;
}

int launch_preprocess(const unsigned char* rgb, int H, int W, int Ho, int Wo, int bgr, int flip, float scaling,
                      const float* means, const float* stds, float* out, unsigned char* out_u8, void* ws,
                      size_t ws_bytes, hipStream_t s)
{
    if (H < 1 || W < 1 || Ho < 1 || Wo < 1 || !means || !stds) return FRCNN_EINVAL;
    if (ws_bytes < preprocess_workspace_bytes(H, W, Ho, Wo)) return FRCNN_EINVAL;
    unsigned char* tmp = static_cast<unsigned char*>(ws);
    size_t off = ((size_t)H * Wo * 3 + 255) / 256 * 256;
    const int kx = ksize_for(W, Wo), ky = ksize_for(H, Ho);
    int* bx = reinterpret_cast<int*>(tmp + off);
    int* kkx = bx + (size_t)Wo * 2;
    int* by = kkx + (size_t)Wo * kx;
    int* kky = by + (size_t)Ho * 2;
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(Wo, 256)), dim3(256), 0, s, W, Wo, kx, bx, kkx);
    hipLaunchKernelGGL(resample_coeffs_kernel, dim3(cdiv(Ho, 256)), dim3(256), 0, s, H, Ho, ky, by, kky);
    int rc = check_launch();
    if (rc) return rc;
    size_t total = (size_t)H * Wo;
    int blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resample_h_kernel, dim3(blocks), dim3(256), 0, s, rgb, H, W, Wo, kx, (const int*)bx, (const int*)kkx,
                       flip, tmp);
    rc = check_launch();
    if (rc) return rc;
    Norm nm;
    nm.scaling = scaling;
    for (int c = 0; c < 3; ++c) {
        nm.mean[c] = means[c]; nm.stdv[c] = stds[c];
        nm.src_channel[c] = bgr ? 2 - c : c;                         // image.py:47: RGB -> BGR
    }
    total = (size_t)Ho * Wo;
    blocks = (int)((total + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(resample_v_norm_kernel, dim3(blocks), dim3(256), 0, s, (const unsigned char*)tmp, H, Wo, Ho, ky,
                       (const int*)by, (const int*)kky, nm, out, out_u8);
    return check_launch();
}

}  // namespace frcnn
