// detect.hip -- final detections on the device: one block per foreground class.
// Replaces models/faster_rcnn.py:179-224, which on the reference is 3 D2H copies, a numpy
// float64 decode per class, and 20 x (2 H2D + torchvision nms + 1 D2H).
//
// Arithmetic follows the reference's numpy promotion rules exactly:
//   :181-183  anchor centre/size from the float32 proposals IN FLOAT32, then widened to float64
//   :192-197  convert_deltas_to_boxes (math_utils.py:91-96) in float64 with stds [.1,.1,.2,.2],
//             means 0: d*std+mean, a_hw*d_yx + a_cyx (two roundings), a_hw*exp(d_hw), +-0.5*size
//   :200-201  clip y to [0,H-1], x to [0,W-1]
//   :208      keep score > threshold (float32 compare)
//   :216-220  torchvision nms on float64 boxes: stable score-descending order, suppress iff
//             inter/(area_i+area_j-inter) > 0.3 evaluated in float64
//   :221-224  rows (y1,x1,y2,x2,score) float64, NMS order
// Sorting is an LDS bitonic sort of (score bits, ~index) keys; the 512x512 IoU bit matrix lives
// in LDS; the greedy pass is one wave holding the 8 "removed" words in lanes 0-7.
#include "common.h"

namespace frcnn {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned ordered_bits32(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}

static constexpr int DET_MAX = 512;

__global__ __launch_bounds__(1024)
void detections_kernel(const float* __restrict__ props, const float* __restrict__ classes,
                       const float* __restrict__ deltas, const int32_t* __restrict__ n_rois,
                       int max_rois, int ncls, double clip_h, double clip_w, float score_thr,
                       double nms_thr, double* __restrict__ out, int32_t* __restrict__ out_cnt)
{
    __shared__ u64 keys[DET_MAX];
    __shared__ double sbox[DET_MAX][4];      // boxes in sorted order
    __shared__ f32x4 fbox[DET_MAX];          // the same boxes rounded to float32: the IoU pre-filter of the bit matrix
    __shared__ unsigned char zero_area[DET_MAX];   // the float64 box has exactly zero height or width (clipped onto an image edge)
    __shared__ u64 mask[DET_MAX][DET_MAX / 64];
    __shared__ int keep_list[DET_MAX];
    __shared__ int counters[2];

    const int cls = blockIdx.x + 1;
    const int t = threadIdx.x;
    int n = *n_rois;
    if (n > max_rois) n = max_rois;
    if (t == 0) { counters[0] = 0; counters[1] = 0; }
    __syncthreads();

#ifdef DET_CLOCKS
    unsigned long long tk[7]; tk[0] = __builtin_readcyclecounter();
#endif
    u64 key = 0ull;
    if (t < n && t < DET_MAX) {
        const float score = classes[(size_t)t * ncls + cls];
        if (score > score_thr) {
            key = ((u64)ordered_bits32(score) << 32) | (u64)(0xFFFFFFFFu - (unsigned)t);
            atomicAdd(&counters[0], 1);
        }
    }
    if (t < DET_MAX) keys[t] = key;
    __syncthreads();
    const int m = counters[0];

#ifdef DET_CLOCKS
    tk[1] = __builtin_readcyclecounter();
#endif
    // bitonic sort, descending, 512 keys / 512 threads (256 compare-exchanges per pass)
    for (int k = 2; k <= DET_MAX; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (t < DET_MAX / 2) {
                const int i = 2 * t - (t & (j - 1));
                const int l = i + j;
                const bool desc = (i & k) == 0;
                const u64 a = keys[i], b = keys[l];
                if ((a < b) == desc) { keys[i] = b; keys[l] = a; }
            }
            __syncthreads();
        }
    }

#ifdef DET_CLOCKS
    tk[2] = __builtin_readcyclecounter();
#endif
    if (t < m) {
        // decode the box of the proposal that landed on rank t
        const int idx = (int)(0xFFFFFFFFu - (unsigned)(keys[t] & 0xFFFFFFFFull));
        const f32x4 p = reinterpret_cast<const f32x4*>(props)[idx];
        const double acy = (double)(0.5f * (p[0] + p[2]));
        const double acx = (double)(0.5f * (p[1] + p[3]));
        const double ah = (double)(p[2] - p[0]);
        const double aw = (double)(p[3] - p[1]);
        const float* d = deltas + (size_t)idx * (ncls - 1) * 4 + (cls - 1) * 4;
        const double dy = __dadd_rn(__dmul_rn((double)d[0], 0.1), 0.0);
        const double dx = __dadd_rn(__dmul_rn((double)d[1], 0.1), 0.0);
        const double dh = __dadd_rn(__dmul_rn((double)d[2], 0.2), 0.0);
        const double dw = __dadd_rn(__dmul_rn((double)d[3], 0.2), 0.0);
        const double cy = __dadd_rn(__dmul_rn(ah, dy), acy);
        const double cx = __dadd_rn(__dmul_rn(aw, dx), acx);
        const double h = __dmul_rn(ah, exp(dh));
        const double w = __dmul_rn(aw, exp(dw));
        double y1 = cy - 0.5 * h, x1 = cx - 0.5 * w, y2 = cy + 0.5 * h, x2 = cx + 0.5 * w;
        y1 = fmin(fmax(y1, 0.0), clip_h); y2 = fmin(fmax(y2, 0.0), clip_h);
        x1 = fmin(fmax(x1, 0.0), clip_w); x2 = fmin(fmax(x2, 0.0), clip_w);
        sbox[t][0] = y1; sbox[t][1] = x1; sbox[t][2] = y2; sbox[t][3] = x2;
        fbox[t] = f32x4{(float)y1, (float)x1, (float)y2, (float)x2};
        zero_area[t] = (y2 - y1 == 0.0 || x2 - x1 == 0.0) ? 1 : 0;
    }
    __syncthreads();

#ifdef DET_CLOCKS
    tk[3] = __builtin_readcyclecounter();
#endif
    // IoU bit matrix (row i, columns > i).  One work item = (row, 64-column word) on or above the diagonal, dealt round-robin to the 8
    // WAVES; lane l of the wave decides column 64 wq + l and a ballot assembles the word (round 2: one thread per ROW walked all its
    // columns in float64 at ~450 cycles per pair -- 131k cycles for a 289-candidate class, the kernel's longest phase by far).
    // Each pair is first decided in float32 with a margin that covers the rounding of the boxes and of the float32 arithmetic
    // (coordinates <= a few thousand pixels: 1e-3 px per box side is > 8 ulp); only lanes inside that band -- a handful of pairs per
    // image -- evaluate the reference's float64 expression inter / (area_i + area_j - inter) > thr.  Same bits as before.
    // Four items per trip with every LDS operand of the four loaded up front and no short-circuit evaluation: the first version of
    // this loop spent ~700 cycles per item in four DEPENDENT LDS round trips (flag -> branch -> boxes -> flag).
    {
        const int nw = (m + 63) >> 6;
        const float thr_f = (float)nms_thr;
        const int nwaves = blockDim.x >> 6;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        // rows 64 g .. 64 g + 63 need the words g .. nw - 1: items are numbered row-major over those (row, word) pairs only; the words left
        // of a row's diagonal word are zero by construction of the greedy pass (it never reads them: a kept row only removes LATER boxes)
        auto decode_item = [&](int item, int& i, int& wq) {
            // closed form: rows of group g start at item S(g) = 64 * (g * nw - g * (g - 1) / 2)
            int g = 0, start = 0;
            while (g + 1 < nw && item >= start + 64 * (nw - g)) { start += 64 * (nw - g); ++g; }
            const int per = nw - g, r = (item - start) / per;
            i = 64 * g + r; wq = g + (item - start) - r * per;
        };
        int total = 0;
        for (int g = 0; g < nw; ++g) { const int rows = (m - 64 * g) < 64 ? (m - 64 * g) : 64; total += rows * (nw - g); }
        for (int base = wave; base < total; base += 4 * nwaves) {
            f32x4 a[4], b[4];
            unsigned z[4];
            int ii[4], wqs[4], jc[4];
            bool valid[4], live[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int item = base + nwaves * q;
                live[q] = item < total;
                decode_item(live[q] ? item : 0, ii[q], wqs[q]);
                const int jj = wqs[q] * 64 + lane;
                valid[q] = live[q] && jj > ii[q] && jj < m;
                jc[q] = jj < m ? jj : m - 1;
                a[q] = fbox[ii[q]]; b[q] = fbox[jc[q]];
                z[q] = (unsigned)zero_area[ii[q]] | (unsigned)zero_area[jc[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float oy = fminf(a[q][2], b[q][2]) - fmaxf(a[q][0], b[q][0]), ox = fminf(a[q][3], b[q][3]) - fmaxf(a[q][1], b[q][1]);
                const float inter = fmaxf(oy, 0.f) * fmaxf(ox, 0.f);
                const float ha = a[q][2] - a[q][0], wa = a[q][3] - a[q][1], hb = b[q][2] - b[q][0], wb = b[q][3] - b[q][1];
                const float lhs = inter - thr_f * (ha * wa + hb * wb - inter);
                const float margin = 1e-3f * (ha + wa + hb + wb + 1.0f);
                // exact shortcuts that keep the float64 path rare: a box of exactly zero area intersects nothing (inter == 0 -> IoU 0 or NaN:
                // never > thr), and boxes the float32 coordinates separate by more than 1e-3 px are separate in float64 too
                const bool never = (z[q] != 0u) | (oy < -1e-3f) | (ox < -1e-3f);
                bool sup = (lhs > margin) & !never;
                if (valid[q] & !never & (lhs <= margin) & (lhs >= -margin)) {
                    const int i = ii[q], j = jc[q];
                    const double a0 = sbox[i][0], a1 = sbox[i][1], a2 = sbox[i][2], a3 = sbox[i][3];
                    const double b0 = sbox[j][0], b1 = sbox[j][1], b2 = sbox[j][2], b3 = sbox[j][3];
                    const double e0 = fmax(fmin(a2, b2) - fmax(a0, b0), 0.0);
                    const double e1 = fmax(fmin(a3, b3) - fmax(a1, b1), 0.0);
                    const double it = e0 * e1;
                    const double u = (a2 - a0) * (a3 - a1) + (b2 - b0) * (b3 - b1) - it, rhs = nms_thr * u;
                    if (it > rhs * (1.0 + 1e-12)) sup = true;                   // the divide only inside a 1e-12 band around the threshold
                    else if (it < rhs * (1.0 - 1e-12)) sup = false;
                    else sup = it / u > nms_thr;
                }
                const u64 bits = __ballot(valid[q] & sup);
                if (live[q] && lane == 0) mask[ii[q]][wqs[q]] = bits;
            }
        }
    }
    __syncthreads();

#ifdef DET_CLOCKS
    tk[4] = __builtin_readcyclecounter();
#endif
    // greedy pass: wave 0, lane w (< 8) owns removed word w
    if (t < 64) {
        u64 rem = 0ull;
        int kept = 0;
        const int nw = (m + 63) >> 6;
        for (int p = 0; p < m; ++p) {
            const int src = __builtin_amdgcn_readfirstlane(p >> 6);
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem & 0xFFFFFFFFull), src);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem >> 32), src);
            const u64 word = ((u64)hi << 32) | lo;
            if (!((word >> (p & 63)) & 1ull)) {
                if (t == 0) keep_list[kept] = p;
                ++kept;
                // words left of the diagonal word of row p are never written (not needed: they could only remove EARLIER boxes)
                if (t < nw && t >= (p >> 6)) rem |= mask[p][t];
            }
        }
        if (t == 0) counters[1] = kept;
    }
    __syncthreads();
#ifdef DET_CLOCKS
    tk[5] = __builtin_readcyclecounter();
    if (t == 0) printf("det cls %d m %d kept %d cycles: score %llu sort %llu decode %llu mask %llu greedy %llu\n", cls, m, counters[1], tk[1] - tk[0],
                       tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
#endif
    const int kept = counters[1];
    if (t < kept) {
        const int p = keep_list[t];
        const u64 k = keys[p];
        const unsigned ob = (unsigned)(k >> 32);
        const unsigned fb = (ob & 0x80000000u) ? (ob & 0x7fffffffu) : ~ob;
        double* o = out + ((size_t)(cls - 1) * max_rois + t) * 5;
        o[0] = sbox[p][0]; o[1] = sbox[p][1]; o[2] = sbox[p][2]; o[3] = sbox[p][3];
        o[4] = (double)__uint_as_float(fb);
    }
    if (t == 0) out_cnt[cls - 1] = kept;
}

int launch_detections(const float* props, const float* classes, const float* deltas,
                      const int32_t* n_rois, int max_rois, int ncls, int image_h, int image_w,
                      float score_thr, float nms_thr, double* out, int32_t* out_cnt, hipStream_t s)
{
    if (max_rois < 1 || max_rois > DET_MAX || ncls < 2 || ncls > 128) return FRCNN_EINVAL;
    hipLaunchKernelGGL(detections_kernel, dim3(ncls - 1), dim3(1024), 0, s, props, classes, deltas, n_rois,
                       max_rois, ncls, (double)(image_h - 1), (double)(image_w - 1), score_thr,
                       (double)nms_thr, out, out_cnt);
    return check_launch();
}

}  // namespace frcnn
