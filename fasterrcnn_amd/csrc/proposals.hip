// proposals.hip -- RPN proposal generation without host round trips.
// Replaces, in order (reference: models/rpn.py):
//   :89  t.sigmoid                          \
//   :98-104,158-173 _extract_valid            >  rpn_decode_kernel (one thread per anchor)
//   :118-123 t_convert_deltas_to_boxes      /   (models/math_utils.py:122-127, fp32, no FMA)
//   :129-132 argsort ascending, flip, [0:N]    topk_sort_kernel: 8-pass radix select of the
//                                              N-th largest 64-bit key, LDS bitonic sort of the
//                                              survivors.  key = (score bits, anchor index+1):
//                                              unique keys -> one deterministic order; ties go
//                                              to the HIGHER anchor index, which is what a stable
//                                              ascending sort followed by flip() produces.
//   :135-144 clamp, >= 16 px filter            same kernel, order-preserving scan compaction
//   :147-153 torchvision.ops.nms(0.7)[0:N]     nms_mask_kernel (64x64 IoU bit tiles) +
//                                              nms_reduce_kernel (one wave: 64 boxes per step,
//                                              readlane over the diagonal word, early exit)
#include "common.h"
#include <type_traits>

namespace frcnn {

typedef unsigned long long u64;

__device__ __forceinline__ unsigned ordered_bits(float f)
{
    const unsigned b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_bits(unsigned o)
{
    const unsigned b = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(b);
}

// One thread per anchor n = (y*fw + x)*9 + k  (the reference's flat order, rpn.py:162-165).
__global__ __launch_bounds__(256)
void rpn_decode_kernel(const float* __restrict__ head, int ld, const float* __restrict__ anchors,
                       const float* __restrict__ valid, int A, float* __restrict__ scores,
                       f32x4* __restrict__ boxes_all, u64* __restrict__ keys)
{
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= A) return;
    const int pix = n / 9, k = n - pix * 9;
    const float* row = head + (size_t)pix * ld;
    const float logit = row[k];
    const float score = 1.0f / (1.0f + expf(-logit));
    const float dy = row[9 + 4 * k + 0], dx = row[9 + 4 * k + 1];
    const float dh = row[9 + 4 * k + 2], dw = row[9 + 4 * k + 3];
    const f32x4 a = reinterpret_cast<const f32x4*>(anchors)[n];   // cy, cx, h, w
    // center = anchors[:,2:4] * deltas[:,0:2] + anchors[:,0:2]  (two roundings, as torch does)
    const float cy = __fadd_rn(__fmul_rn(a[2], dy), a[0]);
    const float cx = __fadd_rn(__fmul_rn(a[3], dx), a[1]);
    const float h = __fmul_rn(a[2], expf(dh));
    const float w = __fmul_rn(a[3], expf(dw));
    const float hh = 0.5f * h, hw = 0.5f * w;
    f32x4 b;
    b[0] = cy - hh; b[1] = cx - hw; b[2] = cy + hh; b[3] = cx + hw;
    scores[n] = score;
    boxes_all[n] = b;
    const bool ok = (valid == nullptr) || (valid[n] > 0.f);
    keys[n] = ok ? (((u64)ordered_bits(score) << 32) | (u64)(unsigned)(n + 1)) : 0ull;
}

// keys for the stand-alone NMS entry: stable descending (ties -> LOWER index first).
__global__ __launch_bounds__(256)
void nms_keys_kernel(const float* __restrict__ scores, int n, u64* __restrict__ keys)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) keys[i] = ((u64)ordered_bits(scores[i]) << 32) | (u64)(0xFFFFFFFFu - (unsigned)i);
}

// ---- register-resident bitonic sort (descending) of 1024*PER keys held PER-per-thread ---------------
// thread t owns global positions [PER*t, PER*t+PER).  Sub-passes with partner distance j < PER are
// in-thread, PER <= j < 64*PER go through __shfl_xor (same wave), j >= 64*PER through LDS.
__device__ __forceinline__ u64 shfl_xor_u64(u64 v, int lane_mask)
{
    const unsigned lo = __shfl_xor((unsigned)(v & 0xFFFFFFFFull), lane_mask);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), lane_mask);
    return ((u64)hi << 32) | lo;
}

// element at global index g, partner at g ^ j, run length k: keep max iff (descending run) == (lower index)
__device__ __forceinline__ u64 bitonic_pick(u64 mine, u64 other, int g, int j, int k)
{
    const bool keep_max = (((g & k) == 0) == ((g & j) == 0));
    const bool other_bigger = other > mine;
    return (keep_max == other_bigger) ? other : mine;
}

template <int PER, int J>
__device__ __forceinline__ void bitonic_inthread(u64 (&e)[PER], int base, int k)
{
#pragma unroll
    for (int r = 0; r < PER; ++r) {
        if ((r & J) == 0) {
            const u64 a = e[r], b = e[r | J];
            e[r] = bitonic_pick(a, b, base + r, J, k);
            e[r | J] = bitonic_pick(b, a, base + (r | J), J, k);
        }
    }
}

template <int PER>
__device__ void bitonic_sort_regs(u64* buf, int tid)
{
    constexpr int N = 1024 * PER;
    u64 e[PER];
    const int base = tid * PER;
#pragma unroll
    for (int r = 0; r < PER; ++r) e[r] = buf[base + r];
    for (int k = 2; k <= N; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64 * PER) {
#pragma unroll
                for (int r = 0; r < PER; ++r) buf[base + r] = e[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < PER; ++r) e[r] = bitonic_pick(e[r], buf[(base + r) ^ j], base + r, j, k);
                __syncthreads();
            } else if (j >= PER) {
                const int lane_mask = j / PER;
#pragma unroll
                for (int r = 0; r < PER; ++r) e[r] = bitonic_pick(e[r], shfl_xor_u64(e[r], lane_mask), base + r, j, k);
            } else {
                if (PER > 8 && j == 8) bitonic_inthread<PER, (PER > 8 ? 8 : 1)>(e, base, k);
                else if (PER > 4 && j == 4) bitonic_inthread<PER, (PER > 4 ? 4 : 1)>(e, base, k);
                else if (PER > 2 && j == 2) bitonic_inthread<PER, (PER > 2 ? 2 : 1)>(e, base, k);
                else if (PER > 1 && j == 1) bitonic_inthread<PER, 1>(e, base, k);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < PER; ++r) buf[base + r] = e[r];
    __syncthreads();
}

// Single block, 1024 threads.  Selects the K largest of n_keys 64-bit keys (zero keys are
// "absent"), sorts them descending in LDS, then emits them.
//   MODE 0 (RPN): idx = low-1; writes sorted_idx; clips to the image, drops boxes with a side
//                 < min_side, compacts in order into cand_boxes/cand_scores.
//   MODE 1 (NMS): idx = 0xFFFFFFFF-low; gathers boxes in sorted order, no clip/filter.
// counts[0] = number selected (<= K), counts[1] = number emitted to cand_*.
//   SPLIT (round 3, MODE 0 only): the kernel stops after the radix select: the survivors go, unsorted, to `sel_out` (sort_n keys, zero
//   padded), counts[0] = their number; topk_rank_kernel (many blocks) and topk_emit_kernel finish the job.  The one-block register
//   bitonic sort of 8192 keys was 63 of this kernel's 103 us with the rest of the chip idle.
template <int MODE, bool SPLIT = false>
__global__ __launch_bounds__(1024)
void topk_sort_kernel(const u64* __restrict__ keys, int n_keys, int K, int sort_n,
                      const f32x4* __restrict__ boxes_src, float image_h, float image_w, float min_side,
                      int32_t* __restrict__ sorted_idx, f32x4* __restrict__ cand_boxes,
                      float* __restrict__ cand_scores, int32_t* __restrict__ counts, u64* __restrict__ sel_out = nullptr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64* buf = reinterpret_cast<u64*>(smem_raw);                 // [sort_n]
    int* hist = reinterpret_cast<int*>(buf + sort_n);            // [4096]
    int* misc = hist + 4096;                                     // [64]
    const int tid = threadIdx.x;

    // each thread keeps its keys (i = tid + 1024*it) in registers when they fit: the select reads them 8 times
    constexpr int KPT = 24;
    const bool in_regs = n_keys <= 1024 * KPT;
    u64 kreg[KPT];
#pragma unroll
    for (int it = 0; it < KPT; ++it) {
        const int i = tid + 1024 * it;
        kreg[it] = (in_regs && i < n_keys) ? keys[i] : 0ull;
    }

#ifdef TOPK_CLOCKS
    unsigned long long tk[6]; tk[0] = __builtin_readcyclecounter();
#endif
    // ---- how many keys are present ----------------------------------------------------------
    if (tid == 0) { misc[0] = 0; misc[1] = 0; }
    __syncthreads();
    {
        int c = 0;
        if (in_regs) {
#pragma unroll
            for (int it = 0; it < KPT; ++it) c += kreg[it] != 0ull;
        } else {
            for (int i = tid; i < n_keys; i += 1024) c += keys[i] != 0ull;
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if ((tid & 63) == 0 && c) atomicAdd(&misc[0], c);
    }
    __syncthreads();
    const int present = misc[0];
    const int want = present < K ? present : K;
    __syncthreads();

#ifdef TOPK_CLOCKS
    tk[1] = __builtin_readcyclecounter();
#endif
    // ---- radix select: threshold = want-th largest key --------------------------------------
    // 12-bit digits (4096 bins): four histogram passes over the 48 significant bits of an RPN key instead of six 8-bit ones, and
    // the scores' clustered high bits spread over 16x more bins -- the LDS atomics of a pass serialise per address.
    u64 thr = 1ull;            // all present keys
    if (present > K) {
        u64 prefix = 0ull, pmask = 0ull;
        int remaining = K;
        const bool short_idx = MODE == 0 && n_keys < 65535;        // index + 1 < 2^16: bits 31..16 are 0 in every key
        // digit d covers bits [sh, sh + bits)
        const int nd = short_idx ? 5 : 6;
        for (int d = 0; d < nd; ++d) {
            int sh, bits;
            if (short_idx) { sh = d == 0 ? 52 : d == 1 ? 40 : d == 2 ? 32 : d == 3 ? 4 : 0; bits = d == 2 ? 8 : d == 4 ? 4 : 12; }
            else { sh = d == 0 ? 52 : d == 1 ? 40 : d == 2 ? 28 : d == 3 ? 16 : d == 4 ? 4 : 0; bits = d == 5 ? 4 : 12; }
            const int nb = 1 << bits;
            const u64 dmask = (u64)(nb - 1);
            for (int i = tid; i < nb; i += 1024) hist[i] = 0;
            __syncthreads();
            if (in_regs) {
#pragma unroll
                for (int it = 0; it < KPT; ++it) {
                    const u64 k = kreg[it];
                    if (k != 0ull && (k & pmask) == prefix) atomicAdd(&hist[(int)((k >> sh) & dmask)], 1);
                }
            } else {
                for (int i = tid; i < n_keys; i += 1024) {
                    const u64 k = keys[i];
                    if (k != 0ull && (k & pmask) == prefix) atomicAdd(&hist[(int)((k >> sh) & dmask)], 1);
                }
            }
            __syncthreads();
            {
                // thread t owns bins top .. top - 3 (descending); block-wide inclusive scan of the counts from the top bin down
                const int top = nb - 1 - 4 * tid;
                const bool own = top >= 3;
                const int h0 = own ? hist[top] : 0, h1 = own ? hist[top - 1] : 0, h2 = own ? hist[top - 2] : 0, h3 = own ? hist[top - 3] : 0;
                const int sum = h0 + h1 + h2 + h3;
                int incl = sum;
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if ((tid & 63) >= o) incl += v;
                }
                if ((tid & 63) == 63) misc[8 + (tid >> 6)] = incl;
                __syncthreads();
                int off = 0;
                for (int wv = 0; wv < (tid >> 6); ++wv) off += misc[8 + wv];
                incl += off;
                const int excl = incl - sum;
                if (excl < remaining && incl >= remaining) {       // exactly one thread
                    int c = excl, dg = top;
                    if (c + h0 < remaining) { c += h0; dg = top - 1;
                        if (c + h1 < remaining) { c += h1; dg = top - 2;
                            if (c + h2 < remaining) { c += h2; dg = top - 3; } } }
                    misc[2] = dg; misc[3] = remaining - c;
                }
            }
            __syncthreads();
            prefix |= (u64)misc[2] << sh;
            pmask |= dmask << sh;
            remaining = misc[3];
            __syncthreads();
        }
        if (short_idx) pmask |= 0xFFFF0000ull;                      // (all zero)
        thr = prefix;
    }

#ifdef TOPK_CLOCKS
    tk[2] = __builtin_readcyclecounter();
#endif
    // ---- gather survivors into LDS, pad, bitonic sort descending -----------------------------
    for (int i = tid; i < sort_n; i += 1024) buf[i] = 0ull;
    __syncthreads();
    if (in_regs) {
        // one returning atomic per wave and round instead of one per survivor (6000 on one address); the order inside buf is
        // irrelevant, it is sorted next
#pragma unroll
        for (int it = 0; it < KPT; ++it) {
            const u64 k = kreg[it];
            const bool take = k != 0ull && k >= thr;
            const unsigned long long m = __ballot(take);
            if (m != 0ull) {
                const int lead = __ffsll((long long)m) - 1;
                int base = 0;
                if ((tid & 63) == lead) base = atomicAdd(&misc[1], __popcll(m));
                base = __shfl(base, lead);
                const int pos = base + __popcll(m & ((1ull << (tid & 63)) - 1ull));
                if (take && pos < sort_n) buf[pos] = k;
            }
        }
    } else {
        for (int i = tid; i < n_keys; i += 1024) {
            const u64 k = keys[i];
            if (k != 0ull && k >= thr) {
                const int pos = atomicAdd(&misc[1], 1);
                if (pos < sort_n) buf[pos] = k;
            }
        }
    }
    __syncthreads();
#ifdef TOPK_CLOCKS
    tk[3] = __builtin_readcyclecounter();
#endif
    if (SPLIT) {
        for (int i = tid; i < sort_n; i += 1024) sel_out[i] = buf[i];
        if (tid == 0) counts[0] = want;
        return;
    }
    if (sort_n == 8192) {
        bitonic_sort_regs<8>(buf, tid);          // the 6000-of-20646 case: 81 of 91 sub-passes barrier-free
    } else if (sort_n == 16384) {
        bitonic_sort_regs<16>(buf, tid);
    } else {
        for (int k = 2; k <= sort_n; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (sort_n >> 1); t += 1024) {
                    const int i = 2 * t - (t & (j - 1));
                    const int l = i + j;
                    const bool desc = (i & k) == 0;
                    const u64 a = buf[i], b = buf[l];
                    if ((a < b) == desc) { buf[i] = b; buf[l] = a; }
                }
                __syncthreads();
            }
        }
    }

#ifdef TOPK_CLOCKS
    tk[4] = __builtin_readcyclecounter();
#endif
    // ---- emit in rank order ------------------------------------------------------------------
    // thread t owns ranks [t*per, (t+1)*per).  With `per` a compile-time constant the box gathers of a thread are issued together
    // and kept for the second pass (they were 2 x per dependent global round trips).
    const int per = sort_n >> 10 ? sort_n >> 10 : 1;
    int* wave_tot = misc + 8;     // [16]
    int pos = 0;
    auto emit = [&](auto perc) {
        constexpr int PERC = decltype(perc)::value;
        u64 kk[PERC];
        f32x4 bb[PERC];
        bool kp[PERC];
        int local = 0;
#pragma unroll
        for (int q = 0; q < PERC; ++q) {
            const int p = tid * PERC + q;
            kp[q] = p < want && p < sort_n;
            kk[q] = kp[q] ? buf[p] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < PERC; ++q) {
            const unsigned low = (unsigned)(kk[q] & 0xFFFFFFFFull);
            const int idx = MODE == 0 ? (int)(low - 1u) : (int)(0xFFFFFFFFu - low);
            bb[q] = kp[q] ? boxes_src[idx] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (kp[q] && sorted_idx) sorted_idx[tid * PERC + q] = idx;
        }
#pragma unroll
        for (int q = 0; q < PERC; ++q) {
            if (MODE == 0) {
                f32x4 b = bb[q];
                b[0] = fmaxf(b[0], 0.f); b[1] = fmaxf(b[1], 0.f);
                b[2] = fminf(b[2], image_h); b[3] = fminf(b[3], image_w);
                bb[q] = b;
                kp[q] = kp[q] && ((b[2] - b[0]) >= min_side) && ((b[3] - b[1]) >= min_side);
            }
            local += kp[q];
        }
        // block exclusive scan of `local`
        int incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) wave_off += wave_tot[wv];
        pos = wave_off + incl - local;
#pragma unroll
        for (int q = 0; q < PERC; ++q) {
            if (kp[q]) {
                cand_boxes[pos] = bb[q];
                cand_scores[pos] = from_ordered_bits((unsigned)(kk[q] >> 32));
                ++pos;
            }
        }
    };
    if (per == 8) {
        emit(std::integral_constant<int, 8>());
    } else if (per == 16) {
        emit(std::integral_constant<int, 16>());
    } else {
        int local = 0;
        for (int q = 0; q < per; ++q) {
            const int p = tid * per + q;
            if (p < want && p < sort_n) {
                const u64 k = buf[p];
                const unsigned low = (unsigned)(k & 0xFFFFFFFFull);
                const int idx = MODE == 0 ? (int)(low - 1u) : (int)(0xFFFFFFFFu - low);
                if (sorted_idx) sorted_idx[p] = idx;
                if (MODE == 0) {
                    f32x4 b = boxes_src[idx];
                    b[0] = fmaxf(b[0], 0.f); b[1] = fmaxf(b[1], 0.f);
                    b[2] = fminf(b[2], image_h); b[3] = fminf(b[3], image_w);
                    const bool keep = ((b[2] - b[0]) >= min_side) && ((b[3] - b[1]) >= min_side);
                    local += keep;
                } else {
                    local += 1;
                }
            }
        }
        int incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) wave_off += wave_tot[wv];
        pos = wave_off + incl - local;
        for (int q = 0; q < per; ++q) {
            const int p = tid * per + q;
            if (p < want && p < sort_n) {
                const u64 k = buf[p];
                const unsigned low = (unsigned)(k & 0xFFFFFFFFull);
                const int idx = MODE == 0 ? (int)(low - 1u) : (int)(0xFFFFFFFFu - low);
                f32x4 b = boxes_src[idx];
                bool keep = true;
                if (MODE == 0) {
                    b[0] = fmaxf(b[0], 0.f); b[1] = fmaxf(b[1], 0.f);
                    b[2] = fminf(b[2], image_h); b[3] = fminf(b[3], image_w);
                    keep = ((b[2] - b[0]) >= min_side) && ((b[3] - b[1]) >= min_side);
                }
                if (keep) {
                    cand_boxes[pos] = b;
                    cand_scores[pos] = from_ordered_bits((unsigned)(k >> 32));
                    ++pos;
                }
            }
        }
    }
    if (tid == 1023) { counts[0] = want; counts[1] = pos; }
#ifdef TOPK_CLOCKS
    tk[5] = __builtin_readcyclecounter();
    if (tid == 0 && n_keys > 20000) printf("topk cycles: load %llu  select %llu  gather %llu  sort %llu  emit %llu\n", tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3], tk[5] - tk[4]);
#endif
}

// Rank sort of the selected keys (all different: the low word is the anchor index): rank = number of keys greater.  Block = 32 keys x 32
// segments of the list (1024 threads): wave w holds the keys 32 b .. 32 b + 31 twice (lane halves) against the segments 2 w and 2 w + 1, so
// that every LDS read of the inner loop is a broadcast; the 32 partial counts of a key meet in LDS.  The thread that owns a key then
// writes everything the old kernel's emit phase produced for its rank: sorted_idx, the clipped box, the score and the keep flag of the
// 16-pixel filter (models/rpn.py:135-144).  ~190 blocks for 6000 keys: the whole chip for a few microseconds.
__global__ __launch_bounds__(1024)
void topk_rank_kernel(const u64* __restrict__ sel, const int32_t* __restrict__ counts, int sort_n, const f32x4* __restrict__ boxes_src,
                      float image_h, float image_w, float min_side, int32_t* __restrict__ sorted_idx, f32x4* __restrict__ tmp_boxes,
                      float* __restrict__ tmp_scores, int32_t* __restrict__ keep_flag)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_rank[];
    u64* lk = reinterpret_cast<u64*>(smem_rank);                    // [sort_n]
    int* part = reinterpret_cast<int*>(lk + sort_n);               // [32 keys][32 segments]
    const int n = counts[0];
    const int k0 = blockIdx.x * 32;
    if (k0 >= n) return;
    const int tid = threadIdx.x;
    const int npad = (n + 31) & ~31;
    for (int i = tid; i < npad; i += 1024) lk[i] = i < n ? sel[i] : 0ull;
    __syncthreads();
    const int kb = tid & 31, seg = tid >> 5;                         // 32 segments
    const int seg_len = npad >> 5;
    const u64 mine = k0 + kb < n ? lk[k0 + kb] : ~0ull;
    const u64* p = lk + seg * seg_len;
    int cnt = 0;
    int j = 0;
    for (; j + 4 <= seg_len; j += 4) {
        const u64 a = p[j], b = p[j + 1], c = p[j + 2], d = p[j + 3];
        cnt += (a > mine) + (b > mine) + (c > mine) + (d > mine);
    }
    for (; j < seg_len; ++j) cnt += p[j] > mine;
    part[kb * 32 + seg] = cnt;
    __syncthreads();
    if (tid < 32 && k0 + tid < n) {
        int rank = 0;
#pragma unroll
        for (int q = 0; q < 32; ++q) rank += part[tid * 32 + ((q + tid) & 31)];        // rotated: conflict-free column walk
        const u64 k = lk[k0 + tid];
        const int idx = (int)((unsigned)(k & 0xFFFFFFFFull) - 1u);
        f32x4 b = boxes_src[idx];
        b[0] = fmaxf(b[0], 0.f); b[1] = fmaxf(b[1], 0.f);
        b[2] = fminf(b[2], image_h); b[3] = fminf(b[3], image_w);
        if (sorted_idx) sorted_idx[rank] = idx;
        tmp_boxes[rank] = b;
        tmp_scores[rank] = from_ordered_bits((unsigned)(k >> 32));
        keep_flag[rank] = (((b[2] - b[0]) >= min_side) && ((b[3] - b[1]) >= min_side)) ? 1 : 0;
    }
}

// Order-preserving compaction of the ranked candidates that passed the size filter: one block, thread t owns the ranks
// [t per, (t + 1) per).  counts[1] = number emitted.
__global__ __launch_bounds__(1024)
void topk_emit_kernel(const f32x4* __restrict__ tmp_boxes, const float* __restrict__ tmp_scores, const int32_t* __restrict__ keep_flag,
                      int sort_n, f32x4* __restrict__ cand_boxes, float* __restrict__ cand_scores, int32_t* __restrict__ counts)
{
    __shared__ int wave_tot[16];
    const int tid = threadIdx.x;
    const int n = counts[0];
    // flags, boxes and scores of the thread's ranks are all requested before anything is used: ONE global round trip
    auto body = [&](auto perc) {
        constexpr int PER = decltype(perc)::value;
        int fl[PER];
        f32x4 bx[PER];
        float sc[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int r = tid * PER + q;
            const bool in = r < n;
            fl[q] = in ? keep_flag[r] : 0;
            bx[q] = in ? tmp_boxes[r] : f32x4{0.f, 0.f, 0.f, 0.f};
            sc[q] = in ? tmp_scores[r] : 0.f;
        }
        int local = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) local += fl[q] ? 1 : 0;
        int incl = local;
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
        __syncthreads();
        int wave_off = 0;
        for (int wv = 0; wv < (tid >> 6); ++wv) wave_off += wave_tot[wv];
        int pos = wave_off + incl - local;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            if (fl[q]) {
                cand_boxes[pos] = bx[q];
                cand_scores[pos] = sc[q];
                ++pos;
            }
        }
        if (tid == 1023) counts[1] = pos;
    };
    const int per = sort_n >> 10 ? sort_n >> 10 : 1;
    if (per == 8) body(std::integral_constant<int, 8>());
    else if (per == 16) body(std::integral_constant<int, 16>());
    else if (per == 4) body(std::integral_constant<int, 4>());
    else if (per == 2) body(std::integral_constant<int, 2>());
    else body(std::integral_constant<int, 1>());
}

// IoU exactly as torchvision's nms kernels compute it (fp32, no +1, no epsilon):
//   inter / (area_a + area_b - inter), suppression iff iou > thr.
__device__ __forceinline__ bool iou_gt(const f32x4 a, const f32x4 b, float thr)
{
    const float l0 = fmaxf(a[0], b[0]), l1 = fmaxf(a[1], b[1]);
    const float r0 = fminf(a[2], b[2]), r1 = fminf(a[3], b[3]);
    const float d0 = fmaxf(r0 - l0, 0.f), d1 = fmaxf(r1 - l1, 0.f);
    const float inter = d0 * d1;
    const float sa = (a[2] - a[0]) * (a[3] - a[1]);
    const float sb = (b[2] - b[0]) * (b[3] - b[1]);
    return (inter / (sa + sb - inter)) > thr;
}

// grid (nw, nw), one wave per 64x64 tile; only tiles on or above the diagonal are written.
__global__ __launch_bounds__(64)
void nms_mask_kernel(const f32x4* __restrict__ boxes, const int32_t* __restrict__ n_ptr, float thr,
                     int nw_stride, u64* __restrict__ mask)
{
    const int n = *n_ptr;
    const int by = blockIdx.y, bx = blockIdx.x;
    if (bx < by || by * 64 >= n || bx * 64 >= n) return;
    __shared__ f32x4 colb[64];
    const int t = threadIdx.x;
    const int jn = bx * 64 + t;
    colb[t] = jn < n ? boxes[jn] : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const int i = by * 64 + t;
    if (i >= n) return;
    const f32x4 a = boxes[i];
    u64 bits = 0ull;
    const int jmax = (n - bx * 64) < 64 ? (n - bx * 64) : 64;
    for (int j = 0; j < jmax; ++j) {
        const int jj = bx * 64 + j;
        if (jj > i && iou_gt(a, colb[j], thr)) bits |= 1ull << j;
    }
    mask[(size_t)i * nw_stride + bx] = bits;
}

__device__ __forceinline__ u64 readlane64(u64 v, int lane_uniform)
{
    const int l = __builtin_amdgcn_readfirstlane(lane_uniform);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xFFFFFFFFull), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((u64)hi << 32) | lo;
}

// One wave.  removed[] lives in registers: lane l holds words l, l+64 (capacity 8192 boxes) and
// l+128, l+192 (16384).  Per 64-box chunk: lane l holds the diagonal word of box 64c+l (prefetched
// one chunk ahead), the chunk is resolved serially on a wave-uniform 64-bit "alive" word in
// SGPRs (s_ff1 + v_readlane per kept box, no LDS), then the kept rows are OR-ed into removed[]
// with independent row loads issued four rows at a time.
__global__ __launch_bounds__(64)
void nms_reduce_kernel(const u64* __restrict__ mask, int nw_stride, const int32_t* __restrict__ n_ptr,
                       int max_keep, const f32x4* __restrict__ cand_boxes,
                       const int32_t* __restrict__ order,      // optional map to input indices
                       int32_t* __restrict__ keep, f32x4* __restrict__ props,
                       int32_t* __restrict__ n_keep_out)
{
    const int n = *n_ptr;
    const int lane = threadIdx.x;
    const int nw = (n + 63) >> 6;
    u64 rem[4] = {0ull, 0ull, 0ull, 0ull};
    __shared__ int32_t kept_list[2048];
    int kept = 0;
    u64 diag_next = (lane < n) ? mask[(size_t)lane * nw_stride] : 0ull;
    for (int c = 0; c < nw && kept < max_keep; ++c) {
        const u64 diag = diag_next;
        {
            const int nrow = (c + 1) * 64 + lane;
            diag_next = (c + 1 < nw && nrow < n) ? mask[(size_t)nrow * nw_stride + (c + 1)] : 0ull;
        }
        const int cq = c >> 6;
        const u64 sel = cq == 0 ? rem[0] : cq == 1 ? rem[1] : cq == 2 ? rem[2] : rem[3];
        const u64 cur = readlane64(sel, c & 63);
        const int left = n - c * 64;
        const u64 validm = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        u64 alive = ~cur & validm;
        u64 keepbits = 0ull;
        while (alive != 0ull && kept < max_keep) {
            const int b = __ffsll((long long)alive) - 1;
            keepbits |= 1ull << b;
            if (lane == 0) kept_list[kept] = c * 64 + b;
            ++kept;
            alive &= ~readlane64(diag, b);
            alive &= ~(1ull << b);
        }
        if (kept >= max_keep) break;
        // OR the kept rows into removed[] for words > c.  The next chunk cannot be resolved before these loads are back, so as many
        // of them as registers allow are in flight at once: 16 rows per step (one row per step was the kernel's latency chain:
        // 300 kept boxes x ~1 us; four rows per step 84 us)
        u64 kb = keepbits;
        const bool wide = nw > 128;
        constexpr int NR = 16;
        while (kb != 0ull) {
            const u64* rows[NR];
#pragma unroll
            for (int u = 0; u < NR; ++u) {
                if (kb != 0ull) {
                    const int b = __ffsll((long long)kb) - 1;
                    kb &= kb - 1ull;
                    rows[u] = mask + (size_t)(c * 64 + b) * nw_stride;
                } else {
                    rows[u] = nullptr;
                }
            }
            u64 v[NR][2];
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int w = lane + 64 * q;
                    const bool need = rows[u] != nullptr && w > c && w < nw;
                    v[u][q] = need ? rows[u][w] : 0ull;
                }
#pragma unroll
            for (int u = 0; u < NR; ++u)
#pragma unroll
                for (int q = 0; q < 2; ++q) rem[q] |= v[u][q];
            if (wide) {                                  // more than 8192 candidates: words 128 .. 255
#pragma unroll
                for (int u = 0; u < NR; ++u)
#pragma unroll
                    for (int q = 2; q < 4; ++q) {
                        const int w = lane + 64 * q;
                        const bool need = rows[u] != nullptr && w > c && w < nw;
                        v[u][q - 2] = need ? rows[u][w] : 0ull;
                    }
#pragma unroll
                for (int u = 0; u < NR; ++u)
#pragma unroll
                    for (int q = 2; q < 4; ++q) rem[q] |= v[u][q - 2];
            }
        }
    }
    __syncthreads();
    for (int k = lane; k < max_keep; k += 64) {
        if (k < kept) {
            const int ci = kept_list[k];
            if (keep) keep[k] = order ? order[ci] : ci;
            if (props) props[k] = cand_boxes[ci];
        } else {
            if (props) props[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if (lane == 0) *n_keep_out = kept;
}

// ---- host side ------------------------------------------------------------------------------
static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t proposal_scratch_bytes(int a_cap, int pre_cap, int post_cap)
{
    size_t b = 0;
    b += align_up((size_t)a_cap * 8, 256);                       // keys
    b += align_up((size_t)a_cap * 16, 256);                      // boxes_all
    b += align_up((size_t)pre_cap * 16, 256);                    // cand_boxes
    b += align_up((size_t)pre_cap * 4, 256);                     // cand_scores
    b += align_up((size_t)pre_cap * (pre_cap / 64) * 8, 256);    // mask
    b += align_up((size_t)post_cap * 4, 256);                    // keep
    return b;
}

void proposal_scratch_carve(ProposalScratch& ps, void* base, int a_cap, int pre_cap, int post_cap)
{
    unsigned char* p = static_cast<unsigned char*>(base);
    ps.keys = reinterpret_cast<u64*>(p);         p += align_up((size_t)a_cap * 8, 256);
    ps.boxes_all = reinterpret_cast<float*>(p);  p += align_up((size_t)a_cap * 16, 256);
    ps.cand_boxes = reinterpret_cast<float*>(p); p += align_up((size_t)pre_cap * 16, 256);
    ps.cand_scores = reinterpret_cast<float*>(p); p += align_up((size_t)pre_cap * 4, 256);
    ps.mask = reinterpret_cast<u64*>(p);         p += align_up((size_t)pre_cap * (pre_cap / 64) * 8, 256);
    ps.keep = reinterpret_cast<int32_t*>(p);
    ps.a_cap = a_cap; ps.pre_cap = pre_cap; ps.post_cap = post_cap;
}

static int pow2_at_least(int v) { int p = 1024; while (p < v) p <<= 1; return p; }

// tmp != NULL (MODE 0): the three-launch form (select -> rank on the whole chip -> emit); tmp holds >= 32 sort_n bytes
template <int MODE>
static int launch_topk(const u64* keys, int n_keys, int K, const float* boxes_src, float ih, float iw,
                       float min_side, int32_t* sorted_idx, float* cand_boxes, float* cand_scores,
                       int32_t* counts, hipStream_t s, void* tmp = nullptr)
{
    const int sort_n = pow2_at_least(K);
    if (sort_n > 16384) return FRCNN_EUNSUPPORTED;
    const size_t lds = (size_t)sort_n * 8 + 4096 * 4 + 64 * 4;
    static const bool no_split = frcnn_knob("FRCNN_TOPK_ONE_BLOCK") != nullptr;          // experiments / A-B tests
    if (MODE == 0 && tmp != nullptr && !no_split) {
        unsigned char* tb = static_cast<unsigned char*>(tmp);
        u64* sel = reinterpret_cast<u64*>(tb);
        f32x4* tboxes = reinterpret_cast<f32x4*>(tb + (size_t)sort_n * 8);
        float* tscores = reinterpret_cast<float*>(tb + (size_t)sort_n * 24);
        int32_t* flags = reinterpret_cast<int32_t*>(tb + (size_t)sort_n * 28);
        auto k1 = topk_sort_kernel<0, true>;
        FRCNN_MAX_LDS_ONCE(k1, 16384 * 8 + 4096 * 4 + 64 * 4);
        hipLaunchKernelGGL(k1, dim3(1), dim3(1024), lds, s, keys, n_keys, K, sort_n, reinterpret_cast<const f32x4*>(boxes_src), ih, iw,
                           min_side, sorted_idx, reinterpret_cast<f32x4*>(cand_boxes), cand_scores, counts, sel);
        int rc = check_launch();
        if (rc) return rc;
        const size_t lds2 = (size_t)sort_n * 8 + 32 * 32 * 4;
        FRCNN_MAX_LDS_ONCE(topk_rank_kernel, 16384 * 8 + 32 * 32 * 4);
        hipLaunchKernelGGL(topk_rank_kernel, dim3(cdiv(K, 32)), dim3(1024), lds2, s, sel, counts, sort_n,
                           reinterpret_cast<const f32x4*>(boxes_src), ih, iw, min_side, sorted_idx, tboxes, tscores, flags);
        rc = check_launch();
        if (rc) return rc;
        hipLaunchKernelGGL(topk_emit_kernel, dim3(1), dim3(1024), 0, s, tboxes, tscores, flags, sort_n, reinterpret_cast<f32x4*>(cand_boxes),
                           cand_scores, counts);
        return check_launch();
    }
    auto kern = topk_sort_kernel<MODE>;
    FRCNN_MAX_LDS_ONCE(kern, 16384 * 8 + 4096 * 4 + 64 * 4);
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, s, keys, n_keys, K, sort_n,
                       reinterpret_cast<const f32x4*>(boxes_src), ih, iw, min_side, sorted_idx,
                       reinterpret_cast<f32x4*>(cand_boxes), cand_scores, counts, (u64*)nullptr);
    return check_launch();
}

int launch_rpn_proposals(const ProposalScratch& ps, const float* head, int ld_head,
                         const float* anchor_map, const float* valid_map, int fh, int fw,
                         int image_h, int image_w, int pre_nms, int post_nms, float nms_thr,
                         float min_side, float* scores, int32_t* sorted_idx, float* props,
                         int32_t* counts, hipStream_t s)
{
    const int A = fh * fw * 9;
    if (A < 1 || A > ps.a_cap || pre_nms < 1 || pre_nms > ps.pre_cap || post_nms < 1 ||
        post_nms > ps.post_cap || post_nms > 2048 || ld_head < 45)
        return FRCNN_EINVAL;
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(cdiv(A, 256)), dim3(256), 0, s, head, ld_head, anchor_map,
                       valid_map, A, scores, reinterpret_cast<f32x4*>(ps.boxes_all), ps.keys);
    int rc = check_launch();
    if (rc) return rc;
    // (the NMS bit matrix is written after the top-N: its storage doubles as the top-N's scratch)
    rc = launch_topk<0>(ps.keys, A, pre_nms, ps.boxes_all, (float)image_h, (float)image_w, min_side,
                        sorted_idx, ps.cand_boxes, ps.cand_scores, counts, s, ps.mask);
    if (rc) return rc;
    const int nw = cdiv(pre_nms, 64);
    const int nw_stride = ps.pre_cap / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, s,
                       reinterpret_cast<const f32x4*>(ps.cand_boxes), counts + 1, nms_thr, nw_stride, ps.mask);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, ps.mask, nw_stride, counts + 1, post_nms,
                       reinterpret_cast<const f32x4*>(ps.cand_boxes), (const int32_t*)nullptr,
                       (int32_t*)nullptr, reinterpret_cast<f32x4*>(props), counts + 2);
    return check_launch();
}

int launch_nms(const ProposalScratch& ps, const float* boxes, const float* scores, int n, float thr,
               int max_keep, int32_t* keep, int32_t* n_keep, hipStream_t s)
{
    if (n < 0 || n > ps.pre_cap || n > ps.a_cap || max_keep < 1 || max_keep > 2048) return FRCNN_EINVAL;
    if (n == 0) {
        FRCNN_HIP_TRY(hipMemsetAsync(n_keep, 0, sizeof(int32_t), s));
        return FRCNN_OK;
    }
    hipLaunchKernelGGL(nms_keys_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, scores, n, ps.keys);
    int rc = check_launch();
    if (rc) return rc;
    // sorted order -> ps.keep is too small for n indices; reuse boxes_all's tail as int scratch
    int32_t* order = reinterpret_cast<int32_t*>(ps.boxes_all);
    int32_t* counts = order + ps.a_cap;          // boxes_all holds 4*a_cap floats; use [a_cap, a_cap+4)
    rc = launch_topk<1>(ps.keys, n, n, boxes, 0.f, 0.f, 0.f, order, ps.cand_boxes, ps.cand_scores, counts, s);
    if (rc) return rc;
    const int nw = cdiv(n, 64);
    const int nw_stride = ps.pre_cap / 64;
    hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, s,
                       reinterpret_cast<const f32x4*>(ps.cand_boxes), counts + 1, thr, nw_stride, ps.mask);
    rc = check_launch();
    if (rc) return rc;
    hipLaunchKernelGGL(nms_reduce_kernel, dim3(1), dim3(64), 0, s, ps.mask, nw_stride, counts + 1, max_keep,
                       reinterpret_cast<const f32x4*>(ps.cand_boxes), order, keep, (f32x4*)nullptr, n_keep);
    return check_launch();
}

}  // namespace frcnn
