// roipool.hip -- RoI max pooling + anchor generation (the two small "geometry" kernels).
//
// roi_pool_kernel replaces torchvision.ops.RoIPool((7,7), 1/16) as used at
// models/detector.py:27,72, including the (y1,x1,y2,x2) -> (b,x1,y1,x2,y2) shuffle of :65-69.
// torchvision semantics (restated; torchvision itself is not vendored in the reference):
//   rs = round(coord * scale) (C round(): half away from zero), re likewise;
//   roi_w = max(re_w - rs_w + 1, 1); bin = roi_w / pooled (float);
//   bin window [floor(p*bin) + rs, ceil((p+1)*bin) + rs) clipped to the map; empty -> 0, else max.
// The feature map is NHWC, so a bin-window pixel is one contiguous C-float run: the block
// (roi, ph, pw) sweeps it with float4 lanes (coalesced 2 KB rows for C = 512), output is
// [roi][ph][pw][C] which the repacked fc1 weight consumes directly (frcnn_pack_fc_chw_to_hwc).
//
// anchors_kernel replaces models/anchors.py:43-135 bit-exactly: float64 arithmetic on a
// float32-rounded cell centre, one final cast to float32.
#include "x3t.h"
#include <cfloat>
#include <cmath>

namespace frcnn {

// One block per output bin (roi, ph, pw), one thread per 4 channels: 14,700 blocks of two waves for 300 RoIs.  The kernel is a
// gather with a dependent max chain per thread (up to ~60 window pixels), i.e. latency bound: what pays is many waves in flight and
// independent loads inside a thread (two partial maxima), not wide blocks -- one block per (roi, ph) that walked the 7 bins in
// sequence with half of its 256 threads idle took 54 us for the 30 MB it writes.
__global__ __launch_bounds__(128)
void roi_pool_kernel(const float* __restrict__ fm, int fh, int fw, int C,
                     const float* __restrict__ rois, const int32_t* __restrict__ n_rois,
                     int pooled, float scale, float* __restrict__ out)
{
    const int r = blockIdx.x, ph = blockIdx.y, pw = blockIdx.z;
    const int C4 = C >> 2;
    f32x4* obin = reinterpret_cast<f32x4*>(out + (((size_t)(r * pooled + ph) * pooled) + pw) * C);
    if (r >= *n_rois) {
        for (int i = threadIdx.x; i < C4; i += 128) obin[i] = f32x4{0.f, 0.f, 0.f, 0.f};
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
    int ws = (int)floorf((float)pw * bin_w) + rs_w;
    int we = (int)ceilf((float)(pw + 1) * bin_w) + rs_w;
    ws = min(max(ws, 0), fw); we = min(max(we, 0), fw);
    const bool empty = (he <= hs) || (we <= ws);
    // The window as ONE run of (he - hs) (we - ws) cells, four independent loads in flight (round 6: a wave that keeps one or two loads in
    // flight is bound by their latency, not by bytes -- what roi_pool_x3t_rows_kernel measured: 46 -> 26.6 us); the cell coordinates are
    // block-uniform (scalar registers).  max is exact: the association does not matter.
    const int ww = we - ws, n_cells = empty ? 0 : (he - hs) * ww;
    const int c4a = threadIdx.x, c4b = threadIdx.x + 128;                          // C <= 1024: one pass of two channel quads per thread, else a loop
    for (int c40 = 0; c40 < C4; c40 += 256) {
        const f32x4 lowest = {-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
        const bool has_a = c40 + c4a < C4, has_b = c40 + c4b < C4;
        f32x4 ma = lowest, mb = lowest;
        const f32x4* base = reinterpret_cast<const f32x4*>(fm) + c40;
        int ch = hs, cw = ws;
        for (int i = 0; i < n_cells; i += 2) {
            const f32x4* p0 = base + ((size_t)ch * fw + cw) * C4;
            if (++cw == we) { cw = ws; ++ch; }
            const bool two = i + 1 < n_cells;
            const f32x4* p1 = two ? base + ((size_t)ch * fw + cw) * C4 : p0;
            if (two && ++cw == we) { cw = ws; ++ch; }
            f32x4 a0 = lowest, a1 = lowest, b0 = lowest, b1 = lowest;
            if (has_a) { a0 = p0[c4a]; a1 = p1[c4a]; }
            if (has_b) { b0 = p0[c4b]; b1 = p1[c4b]; }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = a0[j] > a1[j] ? a0[j] : a1[j], b = b0[j] > b1[j] ? b0[j] : b1[j];
                ma[j] = a > ma[j] ? a : ma[j]; mb[j] = b > mb[j] ? b : mb[j];
            }
        }
        if (empty) { ma = f32x4{0.f, 0.f, 0.f, 0.f}; mb = ma; }
        if (has_a) obin[c40 + c4a] = ma;
        if (has_b) obin[c40 + c4b] = mb;
    }
}

__device__ __forceinline__ unsigned short rp_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float rp_bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// RoIPool straight into x6t tile records (csrc/gemm_x6t.hip: fc1's A operand), written as WHOLE 1 KB pieces: one wave = (bin, block
// of 32 RoIs, 16-channel chunk), lane l = RoI 32 rb + (l & 31), channels 8 (l >> 5) .. + 7 of the chunk -- exactly the record image, so
// the wave's three stores are three contiguous kilobytes.  (A kernel that kept roi_pool_kernel's lanes on channels had 8-byte record stores that land
// in 5.6 million different cache lines and the kernel took 56 us instead of the float32 version's 32.)  Each lane walks its own RoI's
// window (32-byte reads of an L2-resident 4.7 MB map); max is exact, so the association does not matter and the values are
// bit-identical to roi_pool_kernel's.
__global__ __launch_bounds__(256)
void roi_pool_x6t_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                         const int32_t* __restrict__ n_rois, int max_rois, int pooled, float scale, unsigned char* __restrict__ rec, int rbt)
{
    const int lane = threadIdx.x & 63;
    const int K16c = C >> 4;                                           // chunks per bin
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)pooled * pooled * rbt * K16c;
    if (wave >= total) return;
    const int cc = (int)(wave % K16c);
    long long t = wave / K16c;
    const int rb = (int)(t % rbt);
    const int bin = (int)(t / rbt);
    const int ph = bin / pooled, pw = bin - ph * pooled;
    const int r = rb * 32 + (lane & 31);
    const int c0 = cc * 16 + 8 * (lane >> 5);
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    if (r < n) {
        const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];     // y1, x1, y2, x2
        const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
        const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
        const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
        const float bin_h = (float)roi_h / (float)pooled, bin_w = (float)roi_w / (float)pooled;
        int hs = (int)floorf((float)ph * bin_h) + rs_h;
        int he = (int)ceilf((float)(ph + 1) * bin_h) + rs_h;
        hs = min(max(hs, 0), fh); he = min(max(he, 0), fh);
        int ws = (int)floorf((float)pw * bin_w) + rs_w;
        int we = (int)ceilf((float)(pw + 1) * bin_w) + rs_w;
        ws = min(max(ws, 0), fw); we = min(max(we, 0), fw);
        if (he > hs && we > ws) {
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -FLT_MAX;
            for (int h = hs; h < he; ++h) {
                const float* p = fm + ((size_t)h * fw + ws) * C + c0;
                for (int w = ws; w < we; ++w, p += C) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { m[e] = v0[e] > m[e] ? v0[e] : m[e]; m[4 + e] = v1[e] > m[4 + e] ? v1[e] : m[4 + e]; }
                }
            }
        }
    }
    unsigned hi[8], mid[8], lo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = rp_bf16_rne(m[e]);
        const float r1 = m[e] - rp_bf16_f32((unsigned short)hi[e]);
        mid[e] = rp_bf16_rne(r1);
        lo[e] = rp_bf16_rne(r1 - rp_bf16_f32((unsigned short)mid[e]));
    }
    const int chunk = bin * K16c + cc;                                  // k = (ph * pooled + pw) * C + c
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * 3072 + lane * 16;
    *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    *reinterpret_cast<uint4*>(dst + 1024) = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
    *reinterpret_cast<uint4*>(dst + 2048) = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

// ---- f32x3 form (csrc/gemm_x3t.hip): the record array of the pooled matrix in two fp16 terms per value, rows scaled per RoI ----------
// (the same pooling as roi_pool_kernel = torchvision.ops.RoIPool at models/detector.py:65-72; only the output format differs)
// inv[r] = 2^-e of RoI r: every pooled value of the RoI is a maximum over cells of its window, so max over the window of the per-cell
// channel maximum `cmax` (launch_pixel_absmax of the feature map) bounds the row.  One wave per RoI.
__global__ __launch_bounds__(256)
void roi_scale_x3t_kernel(const float* __restrict__ cmax, int fh, int fw, const float* __restrict__ rois, const int32_t* __restrict__ n_rois,
                          int max_rois, int rec_rows, int pooled, float scale, float* __restrict__ inv)
{
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rec_rows) return;
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    float mx = 0.f;
    if (r < n) {
        const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];
        const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
        const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
        const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
        // the union of the bins, from roi_pool_x3t_kernel's OWN float32 edge expressions: the first bin starts at floor(0 * bin) + rs
        // = rs and the last one ends at ceil(pooled * (roi / pooled)) + rs, which in float32 is roi + 1 for some sizes (roi = 57, 114,
        // 121 with 7 bins: ADVICE r3) -- the bins CAN reach one cell past [rs, rs + roi), and a row scale taken from the shorter range
        // would not bound that cell.
        const float bin_h = (float)roi_h / (float)pooled, bin_w = (float)roi_w / (float)pooled;
        const int hs = min(max(rs_h, 0), fh), he = min(max((int)ceilf((float)pooled * bin_h) + rs_h, 0), fh);
        const int ws = min(max(rs_w, 0), fw), we = min(max((int)ceilf((float)pooled * bin_w) + rs_w, 0), fw);
        const int ww = we - ws, cells = (he - hs) * ww;
        for (int i = lane; i < cells; i += 64) {
            const int h = hs + i / ww, w = ws + i % ww;
            mx = fmaxf(mx, cmax[(size_t)h * fw + w]);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float mult, iv;
    hx_row_scale(mx, mult, iv);
    if (lane == 0) inv[r] = iv;
}

// roi_pool_x6t_kernel's work split (lane = RoI, wave = (bin, 32 RoIs, 16-channel chunk)); values scaled by 1 / inv[roi], two pieces.
__global__ __launch_bounds__(256)
void roi_pool_x3t_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                         const int32_t* __restrict__ n_rois, int max_rois, int pooled, float scale, const float* __restrict__ inv,
                         unsigned char* __restrict__ rec, int rbt)
{
    const int lane = threadIdx.x & 63;
    const int K16c = C >> 4;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)pooled * pooled * rbt * K16c;
    if (wave >= total) return;
    const int cc = (int)(wave % K16c);
    long long t = wave / K16c;
    const int rb = (int)(t % rbt);
    const int bin = (int)(t / rbt);
    const int ph = bin / pooled, pw = bin - ph * pooled;
    const int r = rb * 32 + (lane & 31);
    const int c0 = cc * 16 + 8 * (lane >> 5);
    float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    if (r < n) {
        const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];     // y1, x1, y2, x2
        const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
        const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
        const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
        const float bin_h = (float)roi_h / (float)pooled, bin_w = (float)roi_w / (float)pooled;
        int hs = (int)floorf((float)ph * bin_h) + rs_h;
        int he = (int)ceilf((float)(ph + 1) * bin_h) + rs_h;
        hs = min(max(hs, 0), fh); he = min(max(he, 0), fh);
        int ws = (int)floorf((float)pw * bin_w) + rs_w;
        int we = (int)ceilf((float)(pw + 1) * bin_w) + rs_w;
        ws = min(max(ws, 0), fw); we = min(max(we, 0), fw);
        if (he > hs && we > ws) {
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = -FLT_MAX;
            for (int h = hs; h < he; ++h) {
                const float* p = fm + ((size_t)h * fw + ws) * C + c0;
                for (int w = ws; w < we; ++w, p += C) {
                    const f32x4 v0 = *reinterpret_cast<const f32x4*>(p), v1 = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { m[e] = v0[e] > m[e] ? v0[e] : m[e]; m[4 + e] = v1[e] > m[4 + e] ? v1[e] : m[4 + e]; }
                }
            }
        }
        const float mult = hx_mult_of_inv(inv[r]);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] *= mult;
    }
    uint4 phh, pll;
    hx_split8(m, phh, pll);
    const int chunk = bin * K16c + cc;                                  // k = (ph * pooled + pw) * C + c
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = phh;
    *reinterpret_cast<uint4*>(dst + HX_PIECE) = pll;
}

// Round 6: the same records from a block per (bin, block of 32 RoIs).  The kernel above gives a LANE its own RoI: the 64 lanes of a load read
// 32 bytes each of 64 different cells -- 64 cache lines per instruction, and a bin window of a large RoI is dozens of cells: ~350 MB of
// scattered reads per image out of the 4.7 MB map, 47 us with the chip full (5 % of an image's CU time).  Here a WAVE walks the cells of one
// (RoI, bin) with its 64 lanes on 8 consecutive channels each (one 2 KB cell = one fully coalesced load), eight RoIs per wave, the pooled
// and scaled values of the block's 32 RoIs go through LDS ([32][C + 4] floats), and the four waves then write whole 1 KB pieces.  Same maxima,
// same scale, same split: the same bits.  C <= 512 per pass (64 lanes x 8 channels); wider maps loop.
#ifndef FRCNN_ROI_ROWS_WAVES
#define FRCNN_ROI_ROWS_WAVES 8      // measured (one image at a time): 4 waves 34.9 us, 8 waves 28.6 us, 16 waves 28.5 us; the lane-per-RoI kernel 47
#endif
template <int NW>
__global__ __launch_bounds__(64 * NW)
void roi_pool_x3t_rows_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                              const int32_t* __restrict__ n_rois, int max_rois, int pooled, float scale, const float* __restrict__ inv,
                              unsigned char* __restrict__ rec, int rbt)
{
    extern __shared__ __attribute__((aligned(16))) float rp_pool[];           // [32][C + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int bin = blockIdx.x / rbt, rb = blockIdx.x - bin * rbt;
    const int ph = bin / pooled, pw = bin - ph * pooled;
    const int LD = C + 4;
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    for (int q = 0; q < 32 / NW; ++q) {                                      // (NW waves, 32 / NW RoIs each: the more waves, the more loads in flight per CU)
        const int rl = wave * (32 / NW) + q, r = rb * 32 + rl;                       // (wave-uniform)
        int hs = 0, he = 0, ws = 0, we = 0;
        float mult = 0.f;
        if (r < n) {
            const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];     // y1, x1, y2, x2
            const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
            const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
            const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
            const float bin_h = (float)roi_h / (float)pooled, bin_w = (float)roi_w / (float)pooled;
            hs = (int)floorf((float)ph * bin_h) + rs_h;
            he = (int)ceilf((float)(ph + 1) * bin_h) + rs_h;
            hs = min(max(hs, 0), fh); he = min(max(he, 0), fh);
            ws = (int)floorf((float)pw * bin_w) + rs_w;
            we = (int)ceilf((float)(pw + 1) * bin_w) + rs_w;
            ws = min(max(ws, 0), fw); we = min(max(we, 0), fw);
            mult = hx_mult_of_inv(inv[r]);
        }
        const bool any = r < n && he > hs && we > ws;
        for (int c0 = 8 * lane; c0 < C; c0 += 512) {
            float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (any) {
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = -FLT_MAX;
                // the window as ONE run of cells, four loads in flight (the pooling was bound by the ONE load a wave kept in flight: 5.8 TB/s
                // of the L2s' ~34); past the end the last cell is read again (a maximum does not mind)
                const int ww = we - ws, ncell = (he - hs) * ww;
                const float rww = __builtin_amdgcn_rcpf((float)ww);
                constexpr int UN = 4;
                for (int i0 = 0; i0 < ncell; i0 += UN) {
                    f32x4 v[UN][2];
#pragma unroll
                    for (int u = 0; u < UN; ++u) {
                        const int i = min(i0 + u, ncell - 1);
                        int dh = (int)((float)i * rww);                        // i / ww (ww <= fw: exact after one correction)
                        int dw = i - dh * ww;
                        if (dw < 0) { --dh; dw += ww; } else if (dw >= ww) { ++dh; dw -= ww; }
                        const float* p = fm + ((size_t)(hs + dh) * fw + ws + dw) * C + c0;
                        v[u][0] = *reinterpret_cast<const f32x4*>(p); v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
                    }
#pragma unroll
                    for (int u = 0; u < UN; ++u)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {                       // (one v_max_f32 per value: `v > m ? v : m` is a compare and a select; finite inputs: the same value)
                            asm("v_max_f32 %0, %1, %2" : "=v"(m[e]) : "v"(v[u][0][e]), "v"(m[e]));
                            asm("v_max_f32 %0, %1, %2" : "=v"(m[4 + e]) : "v"(v[u][1][e]), "v"(m[4 + e]));
                        }
                }
            }
            if (r < n) {
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] *= mult;
            }
            float* d = rp_pool + rl * LD + c0;
            *reinterpret_cast<f32x4*>(d) = f32x4{m[0], m[1], m[2], m[3]};
            *reinterpret_cast<f32x4*>(d + 4) = f32x4{m[4], m[5], m[6], m[7]};
        }
    }
    __syncthreads();
    // pieces: wave w writes the chunks w, w + 8, ...; lane = (row lane & 31, k-half lane >> 5) as in the record layout
    const int K16c = C >> 4;
    for (int cc = wave; cc < K16c; cc += NW) {
        const float* sp = rp_pool + (lane & 31) * LD + cc * 16 + 8 * (lane >> 5);
        const f32x4 a = *reinterpret_cast<const f32x4*>(sp), b = *reinterpret_cast<const f32x4*>(sp + 4);
        const float m[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        uint4 phh, pll;
        hx_split8(m, phh, pll);
        const int chunk = bin * K16c + cc;                              // k = (ph * pooled + pw) * C + c
        unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
        *reinterpret_cast<uint4*>(dst) = phh;
        *reinterpret_cast<uint4*>(dst + HX_PIECE) = pll;
    }
}

// Round 6, second form: a WAVE per (RoI, bin ROW ph) pools the seven bins of that row in one sweep over the row's cells -- every cell of the
// band is loaded ONCE (64 lanes x 8 channels = one coalesced 2 KB load) and folded into the one or two bins whose column range holds it
// (adjacent bins overlap by the floor / ceil of their edges: a sweep per bin, the kernels above, reads the workload's proposals' cells 1.45x
// as often: 257 MB per image against 177 MB, tools/exp_roi_cells.py; the pooling runs at the L2s' bandwidth).  The column ranges are
// wave-uniform (scalar compares, uniform branches).  No LDS: a lane writes its 8 channels of a bin as one 16-byte hi and one 16-byte lo
// store into the record pieces.  Same maxima, scale and split: the same bits.  pooled <= 8, C % 8 == 0; C > 512 loops.
template <int P>
__global__ __launch_bounds__(256)
void roi_pool_x3t_bands_kernel(const float* __restrict__ fm, int fh, int fw, int C, const float* __restrict__ rois,
                               const int32_t* __restrict__ n_rois, int max_rois, float scale, const float* __restrict__ inv,
                               unsigned char* __restrict__ rec, int rbt)
{
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + (int)(threadIdx.x >> 6));
    if (wid >= rbt * 32 * P) return;
    const int ph = wid % P, r = wid / P;
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    int hs = 0, he = 0, ws[P], we[P], wlo = 0, whi = 0;
    float mult = 0.f;
#pragma unroll
    for (int q = 0; q < P; ++q) { ws[q] = 0; we[q] = 0; }
    if (r < n) {
        const f32x4 roi = reinterpret_cast<const f32x4*>(rois)[r];     // y1, x1, y2, x2 (uniform address: every lane reads the same box)
        const int rs_h = (int)roundf(roi[0] * scale), rs_w = (int)roundf(roi[1] * scale);
        const int re_h = (int)roundf(roi[2] * scale), re_w = (int)roundf(roi[3] * scale);
        const int roi_h = max(re_h - rs_h + 1, 1), roi_w = max(re_w - rs_w + 1, 1);
        const float bin_h = (float)roi_h / (float)P, bin_w = (float)roi_w / (float)P;
        hs = (int)floorf((float)ph * bin_h) + rs_h;
        he = (int)ceilf((float)(ph + 1) * bin_h) + rs_h;
        hs = min(max(hs, 0), fh); he = min(max(he, 0), fh);
        wlo = fw; whi = 0;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            int a = (int)floorf((float)q * bin_w) + rs_w;
            int b = (int)ceilf((float)(q + 1) * bin_w) + rs_w;
            a = min(max(a, 0), fw); b = min(max(b, 0), fw);
            ws[q] = __builtin_amdgcn_readfirstlane(a); we[q] = __builtin_amdgcn_readfirstlane(b);
            wlo = min(wlo, ws[q]); whi = max(whi, we[q]);
        }
        hs = __builtin_amdgcn_readfirstlane(hs); he = __builtin_amdgcn_readfirstlane(he);
        wlo = __builtin_amdgcn_readfirstlane(wlo); whi = __builtin_amdgcn_readfirstlane(whi);
        mult = hx_mult_of_inv(inv[r]);
    }
    const int K16c = C >> 4, rb = r >> 5, row = r & 31;
    for (int c0 = 8 * lane; c0 < C; c0 += 512) {
        float acc[P][8];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const float init = (r < n && he > hs && we[q] > ws[q]) ? -FLT_MAX : 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[q][e] = init;
        }
        for (int h = hs; h < he; ++h) {
            const float* prow = fm + ((size_t)h * fw) * C + c0;
            for (int w0 = wlo; w0 < whi; w0 += 4) {
                // four cells in flight (the tail re-reads the band's last cell: it is folded twice, a maximum does not mind)
                f32x4 v[4][2];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int w = min(w0 + u, whi - 1);
                    const float* p = prow + (size_t)w * C;
                    v[u][0] = *reinterpret_cast<const f32x4*>(p); v[u][1] = *reinterpret_cast<const f32x4*>(p + 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int w = min(w0 + u, whi - 1);
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        if (w >= ws[q] && w < we[q]) {                      // (wave-uniform)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                acc[q][e] = v[u][0][e] > acc[q][e] ? v[u][0][e] : acc[q][e];
                                acc[q][4 + e] = v[u][1][e] > acc[q][4 + e] ? v[u][1][e] : acc[q][4 + e];
                            }
                        }
                    }
                }
            }
        }
        const int cc = c0 >> 4, kh = (c0 >> 3) & 1;
#pragma unroll
        for (int q = 0; q < P; ++q) {
            float m[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = r < n ? acc[q][e] * mult : 0.f;
            uint4 phh, pll;
            hx_split8(m, phh, pll);
            const int chunk = (ph * P + q) * K16c + cc;                     // k = (ph * pooled + pw) * C + c
            unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + kh * 512 + row * 16;
            *reinterpret_cast<uint4*>(dst) = phh;
            *reinterpret_cast<uint4*>(dst + HX_PIECE) = pll;
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
    hipLaunchKernelGGL(roi_pool_kernel, dim3(max_rois, pooled, pooled), dim3(128), 0, s, fm, fh, fw, c, rois,
                       n_rois, pooled, scale, out);
    return check_launch();
}

// the same pooling, output = the x6t record array (csrc/gemm_x6t.hip) of the [max_rois][pooled * pooled * c] matrix; rec_rows % 32 == 0
// rows allocated, the rows max_rois .. rec_rows-1 are the caller's to zero once; c % 16 == 0
int launch_roi_pool_x6t(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois,
                        int max_rois, int pooled, float scale, void* rec, int rec_rows, hipStream_t s)
{
    if (fh < 1 || fw < 1 || c < 16 || c % 16 != 0 || max_rois < 1 || pooled < 1 || rec_rows < max_rois || rec_rows % 32 != 0) return FRCNN_EINVAL;
    // only the row blocks that hold RoIs are written (the rest of the array stays as the caller zeroed it)
    const int rbt = rec_rows / 32, rb_live = cdiv(max_rois, 32);
    (void)rb_live;
    const long long waves = (long long)pooled * pooled * rbt * (c / 16);
    hipLaunchKernelGGL(roi_pool_x6t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, fm, fh, fw, c, rois, n_rois, max_rois,
                       pooled, scale, static_cast<unsigned char*>(rec), rbt);
    return check_launch();
}

// RoI pooling into x3t records + the per-RoI scales (csrc/gemm_x3t.hip).  cmax: fh * fw floats of scratch; inv: rec_rows floats.
int launch_roi_pool_x3t(const float* fm, int fh, int fw, int c, const float* rois, const int32_t* n_rois, int max_rois, int pooled,
                        float scale, float* cmax, float* inv, void* rec, int rec_rows, hipStream_t s, bool cmax_ready)
{
    if (fh < 1 || fw < 1 || c < 16 || c % 16 != 0 || max_rois < 1 || pooled < 1 || rec_rows < max_rois || rec_rows % 32 != 0 || !cmax || !inv)
        return FRCNN_EINVAL;
    int rc = cmax_ready ? FRCNN_OK : launch_pixel_absmax(fm, cmax, (long long)fh * fw, c, s);      // (ready: conv5_3's epilogue wrote them)
    if (rc) return rc;
    hipLaunchKernelGGL(roi_scale_x3t_kernel, dim3(cdiv(rec_rows, 4)), dim3(256), 0, s, cmax, fh, fw, rois, n_rois, max_rois, rec_rows, pooled, scale, inv);
    if ((rc = check_launch()) != FRCNN_OK) return rc;
    const int rbt = rec_rows / 32;
#ifdef FRCNN_ROI_BANDS
    if (pooled == 7 && c % 8 == 0) {
        // a wave per (RoI, bin row): every cell of the band read once (round 6, second form; the same bits as the kernels below).  MEASURED
        // 68.7 us against the rows kernel's 44-46: the pooling is not bound by the bytes it reads (5.8 TB/s of the L2s' ~34) but by the loads
        // it keeps in flight and by its compare-and-select work; not the shipped path
        const long long waves = (long long)rec_rows * pooled;
        hipLaunchKernelGGL(roi_pool_x3t_bands_kernel<7>, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, fm, fh, fw, c, rois, n_rois, max_rois, scale,
                           inv, static_cast<unsigned char*>(rec), rbt);
        return check_launch();
    }
#endif
    const size_t lds = (size_t)32 * (c + 4) * sizeof(float);
    if (c % 8 == 0 && lds <= 160 * 1024) {
        // a block per (bin, 32 RoIs): coalesced cell reads, whole-piece writes (round 6; the same bits as the kernel below)
        constexpr int NW = FRCNN_ROI_ROWS_WAVES;
        auto kern = roi_pool_x3t_rows_kernel<NW>;
        FRCNN_MAX_LDS_ONCE(kern, 160 * 1024);                           // (once per device: the largest size any c may ask for)
        hipLaunchKernelGGL(kern, dim3((unsigned)(pooled * pooled * rbt)), dim3(64 * NW), lds, s, fm, fh, fw, c, rois, n_rois, max_rois, pooled, scale,
                           inv, static_cast<unsigned char*>(rec), rbt);
        return check_launch();
    }
    const long long waves = (long long)pooled * pooled * rbt * (c / 16);
    hipLaunchKernelGGL(roi_pool_x3t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, fm, fh, fw, c, rois, n_rois, max_rois,
                       pooled, scale, inv, static_cast<unsigned char*>(rec), rbt);
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
