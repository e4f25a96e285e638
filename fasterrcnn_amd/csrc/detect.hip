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
// Sorting is a rank sort of the unique (score bits, ~index) keys; the 512x512 IoU bit matrix lives in LDS (a lane per row, a wave per
// (64 rows, 64 columns) unit); the greedy pass is one wave holding the 8 "removed" words in lanes 0-7.
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
    __shared__ u64 keys[DET_MAX];            // (score bits, ~proposal index) of the candidates (0 = below the threshold); after the sort: ranks 0 .. m - 1
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
    // Rank sort, descending (round 5; the LDS bitonic sort it replaces was 45 passes with a 1024-thread barrier each): the keys are unique
    // (the proposal index rides in the low word), so the rank of a key is the number of larger keys.  Threads t and t + 512 count over one
    // half of the keys each (every lane of a wave reads the same key: an LDS broadcast), the halves meet in an LDS counter.
    {
        const int nk = n < DET_MAX ? n : DET_MAX;
        const int half = (nk + 1) >> 1;
        const int kt = t & (DET_MAX - 1);
        const u64 mine = keys[kt];
        int* const rank_of = keep_list;                                     // (the greedy pass's list: not in use yet)
        if (t < DET_MAX) rank_of[t] = 0;
        __syncthreads();
        if (mine != 0ull) {
            // eight keys per trip (four 16-byte broadcast reads issued together: an un-unrolled loop paid the LDS latency per key);
            // slots past nk hold 0 and count nothing, so the ranges are rounded to whole trips
            const int h8 = (half + 7) & ~7, n8 = (nk + 7) & ~7;
            const int j0 = t < DET_MAX ? 0 : h8, j1 = t < DET_MAX ? h8 : n8;
            int cnt = 0;
            for (int j = j0; j < j1; j += 8) {
                u64 k[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) k[q] = keys[j + q];
#pragma unroll
                for (int q = 0; q < 8; ++q) cnt += k[q] > mine ? 1 : 0;
            }
            atomicAdd(&rank_of[kt], cnt);
        }
        __syncthreads();                                                    // every read of keys[] is done: the sorted keys go in place
        if (t < DET_MAX && mine != 0ull) keys[rank_of[t]] = mine;
        __syncthreads();
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
    // IoU bit matrix (row i, columns > i).  One work unit = (group g of 64 rows, 64-column word wq >= g), dealt round-robin to the 16 waves;
    // LANE l OWNS ROW 64 g + l and walks the word's 64 columns (every lane reads the same column box: an LDS broadcast), collecting its
    // row's bits in a register -- no ballot, no per-item index arithmetic (round 3's form dealt (row, word) items to the waves with lane =
    // column and a ballot per item: ~930 items of ~500 instructions per four for a 289-candidate class, 50 of the kernel's 76 us; round 2's
    // lane-per-row form walked the columns in float64 at ~450 cycles per pair).
    // Each pair is first decided in float32 with a margin that covers the rounding of the boxes and of the float32 arithmetic
    // (coordinates <= a few thousand pixels: 1e-3 px per box side is > 8 ulp); only lanes inside that band -- a handful of pairs per
    // image -- evaluate the reference's float64 expression inter / (area_i + area_j - inter) > thr.  Same decisions, same bits.
    {
        const int nw = (m + 63) >> 6;
        const float thr_f = (float)nms_thr;
        const int nwaves = blockDim.x >> 6;
        const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
        const int units = nw * (nw + 1) / 2;
        for (int u = wave; u < units; u += nwaves) {
            int g = 0, start = 0;                                            // unit -> (g, wq): row group g has the words g .. nw - 1
            while (u >= start + (nw - g)) { start += nw - g; ++g; }
            const int wq = g + (u - start);
            const int i = 64 * g + lane;
            const bool row_ok = i < m;
            const f32x4 a = fbox[row_ok ? i : 0];
            const unsigned za = zero_area[row_ok ? i : 0];
            const float ha = a[2] - a[0], wa = a[3] - a[1];
            const int jn = (m - 64 * wq) < 64 ? (m - 64 * wq) : 64;
            u64 bits = 0ull;
            for (int jj = 0; jj < jn; ++jj) {
                const int j = 64 * wq + jj;
                const f32x4 b = fbox[j];
                const unsigned z = za | (unsigned)zero_area[j];
                const float oy = fminf(a[2], b[2]) - fmaxf(a[0], b[0]), ox = fminf(a[3], b[3]) - fmaxf(a[1], b[1]);
                const float inter = fmaxf(oy, 0.f) * fmaxf(ox, 0.f);
                const float hb = b[2] - b[0], wb = b[3] - b[1];
                const float lhs = inter - thr_f * (ha * wa + hb * wb - inter);
                const float margin = 1e-3f * (ha + wa + hb + wb + 1.0f);
                // exact shortcuts that keep the float64 path rare: a box of exactly zero area intersects nothing (inter == 0 -> IoU 0 or NaN:
                // never > thr), and boxes the float32 coordinates separate by more than 1e-3 px are separate in float64 too
                const bool never = (z != 0u) | (oy < -1e-3f) | (ox < -1e-3f);
                const bool valid = row_ok && j > i;
                bool sup = (lhs > margin) & !never;
                if (valid & !never & (lhs <= margin) & (lhs >= -margin)) {
                    const double a0 = sbox[i][0], a1 = sbox[i][1], a2 = sbox[i][2], a3 = sbox[i][3];
                    const double b0 = sbox[j][0], b1 = sbox[j][1], b2 = sbox[j][2], b3 = sbox[j][3];
                    const double e0 = fmax(fmin(a2, b2) - fmax(a0, b0), 0.0);
                    const double e1 = fmax(fmin(a3, b3) - fmax(a1, b1), 0.0);
                    const double it = e0 * e1;
                    const double un = (a2 - a0) * (a3 - a1) + (b2 - b0) * (b3 - b1) - it, rhs = nms_thr * un;
                    if (it > rhs * (1.0 + 1e-12)) sup = true;                   // the divide only inside a 1e-12 band around the threshold
                    else if (it < rhs * (1.0 - 1e-12)) sup = false;
                    else sup = it / un > nms_thr;
                }
                if (valid & sup) bits |= 1ull << jj;
            }
            if (row_ok) mask[i][wq] = bits;
        }
    }
    __syncthreads();

#ifdef DET_CLOCKS
    tk[4] = __builtin_readcyclecounter();
#endif
    // greedy pass: wave 0, lane w (< 8) owns removed word w.  Per 64-row chunk (round 5; the row-by-row loop it replaces spent ~120 cycles on
    // every one of the m rows in a readlane -> scalar test -> branch chain): the chunk's removed word is read ONCE, lane l fetches the diagonal
    // word of row 64 c + l, the chunk is resolved on a wave-uniform 64-bit "alive" word (s_ff1 + v_readlane per KEPT row), then the kept
    // rows are OR-ed into the later words four rows at a time.
    if (t < 64) {
        u64 rem = 0ull;
        int kept = 0;
        const int nw = (m + 63) >> 6;
        for (int c = 0; c < nw; ++c) {
            const int row = 64 * c + t;
            const u64 diag = row < m ? mask[row][c] : 0ull;
            const int src = __builtin_amdgcn_readfirstlane(c);
            const unsigned clo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem & 0xFFFFFFFFull), src);
            const unsigned chi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(rem >> 32), src);
            const int left = m - 64 * c;
            const u64 validm = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
            u64 alive = ~(((u64)chi << 32) | clo) & validm;
            u64 keepbits = 0ull;
            while (alive != 0ull) {
                const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)alive) - 1);
                keepbits |= 1ull << b;
                if (t == 0) keep_list[kept] = 64 * c + b;
                ++kept;
                const unsigned dlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag & 0xFFFFFFFFull), b);
                const unsigned dhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(diag >> 32), b);
                alive &= ~(((u64)dhi << 32) | dlo);
                alive &= ~(1ull << b);
            }
            // words left of the diagonal word of a row are never written (not needed: they could only remove EARLIER boxes)
            const bool mine_w = t < nw && t > c;
            u64 kb = keepbits;
            while (kb != 0ull) {
                u64 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q] = 0ull;
                    if (kb != 0ull) {
                        const int b = __ffsll((long long)kb) - 1;
                        kb &= kb - 1ull;
                        if (mine_w) v[q] = mask[64 * c + b][t];
                    }
                }
                rem |= (v[0] | v[1]) | (v[2] | v[3]);
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
