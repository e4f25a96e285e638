// conv_x6.hip -- the 3x3 halo-tile convolution of conv.hip on the bf16 matrix pipe with fp32-class
// accuracy ("f32x6" math mode).
//
// Every fp32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (8 + 8 + 8 significand
// bits; each residual is formed in fp32 and is exact).  A product x*y is then the sum of nine
// bf16 x bf16 products, each exact in fp32; the six largest are evaluated
//     hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid
// with v_mfma_f32_32x32x16_bf16 (f32 accumulate); the three dropped terms are <= 2^-24 |x*y| each,
// the size of one fp32 rounding.  Six bf16 MFMAs (32 cycles, K = 16) replace eight f32 MFMAs
// (64 cycles, K = 2): 2.67x the matrix-pipe rate of the exact-f32 kernel at the same accuracy class
// (measured against fp64 truth in tests/test_kernels_gpu.py).
//
// Weights are pre-split at pack time into [tap][cout][chunk][plane][16] bf16 (96 contiguous bytes
// per (cout, 16-channel chunk)); activations stay fp32 in HBM and are split while the halo tile is
// staged into LDS.  LDS rows are [plane][16] bf16 = 96 B padded to 112 B: a ds_read_b128 lane
// group then covers 16 distinct 16-byte slots (28*i mod 64 is injective on i mod 16).
// conv3x3_x6_kernel (cout = 64 tile): the halo tile is single-buffered (one extra barrier per 16-channel chunk) so that a
// block needs 51.5 KB of LDS and three blocks fit a CU.  conv3x3_x6p_kernel (cout % 128 == 0, all other layers) is the
// software-pipelined form (see its comment).  Where the mode stands: 171 TFLOP/s fp32-equivalent over the VGG-16 layers,
// 215 on a long-K layer = 1.29 PFLOP/s of bf16 MFMA work, 51 % of the 2.5 PFLOP/s dense peak.
// Tiling, weight double buffering, split-K and the fused bias/ReLU/2x2-pool epilogue are those of
// conv3x3_mfma_kernel (the 32x32 accumulator layout is dtype independent).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace frcnn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

static constexpr int XROW = 112;   // bytes per LDS row: 3 planes x 32 B + 16 B pad
static constexpr int XHC = 34;

__device__ __forceinline__ u16 bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (u16)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(u16 h) { return __uint_as_float((unsigned)h << 16); }

// x = hi + mid + lo exactly (barring overflow / subnormal tails, irrelevant for activations)
__device__ __forceinline__ void split3(float x, u16& hi, u16& mid, u16& lo)
{
    hi = bf16_rne(x);
    const float r1 = x - bf16_to_f32(hi);
    mid = bf16_rne(r1);
    const float r2 = r1 - bf16_to_f32(mid);
    lo = bf16_rne(r2);
}

template <int WM, int WN, int MT>
struct X6Cfg {
    static constexpr int TR = MT * WM;                   // image rows per block (MT rows per wave)
    static constexpr int BN = 64 * WN;
    static constexpr int HR = TR + 2;
    static constexpr int HALO_B = HR * XHC * XROW;       // bytes
    static constexpr int WT_B = BN * XROW;
    static constexpr int NHP = HR * XHC * 4;             // fp32x4 pieces per chunk
    static constexpr int NH = (NHP + 255) / 256;
    static constexpr int NWP = BN * 6;                   // 16-B weight pieces per stage
    static constexpr int NW = (NWP + 255) / 256;         // per thread
    static constexpr size_t LDS_BYTES = (size_t)HALO_B + 2 * WT_B;     // ONE halo buffer: 3 blocks per CU
};

template <int WM, int WN, int MT, bool POOL>
__global__ __launch_bounds__(256, (MT >= 4 ? 2 : 3))      // 8 accumulators: hold the kernel to 256 registers
void conv3x3_x6_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wq,
                       const float* __restrict__ bias, float* __restrict__ y,
                       int H, int W, int Cin, int Cout, int relu, int cout_tiles, int chunks_per_split,
                       float* __restrict__ ws)
{
    using C = X6Cfg<WM, WN, MT>;
    static_assert(MT % 2 == 0, "the fused 2x2 pool pairs rows inside a wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x6[];
    unsigned char* const halo0 = smem_x6;
    unsigned char* const wts0 = smem_x6 + C::HALO_B;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int x0 = blockIdx.x * 32;
    const int y0 = blockIdx.y * C::TR;
    const int ksplit_idx = blockIdx.z / cout_tiles;
    const int n0 = (blockIdx.z - ksplit_idx * cout_tiles) * C::BN;
    const int nchunks = Cin >> 4;

    int h_src[C::NH], h_dst[C::NH];
#pragma unroll
    for (int it = 0; it < C::NH; ++it) {
        const int q = tid + 256 * it;
        const int pix = q >> 2, p = q & 3;
        const int hy = pix / XHC, hx = pix - hy * XHC;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool have = q < C::NHP;
        const bool inb = have && gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_dst[it] = have ? pix * XROW + p * 8 : -1;
        h_src[it] = inb ? (gy * W + gx) * Cin + 4 * p : -1;
    }
    // weights: piece q -> row q/6, 16-byte slot q%6 of the 96-byte (cout, chunk) record
    size_t w_src[C::NW];
    int w_dst[C::NW];
#pragma unroll
    for (int it = 0; it < C::NW; ++it) {
        const int q = tid + 256 * it;
        const bool have = q < C::NWP;
        const int qq = have ? q : tid;                  // surplus threads re-load a valid piece, never store it
        const int row = qq / 6, j = qq - row * 6;
        w_src[it] = (size_t)(n0 + row) * nchunks * 96 + j * 16;
        w_dst[it] = have ? row * XROW + j * 16 : -1;
    }
    const size_t tap_stride = (size_t)Cout * nchunks * 96;

    f32x4 hreg[C::NH];
    f32x4 wregA[C::NW];

    auto load_halo = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < C::NH; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (h_src[it] >= 0) v = *reinterpret_cast<const f32x4*>(x + h_src[it] + chunk * 16);
            hreg[it] = v;
        }
    };
    auto store_halo = [&](unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < C::NH; ++it) {
            if (h_dst[it] < 0) continue;
            u16 hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(hreg[it][j], hi[j], mid[j], lo[j]);
            uint2 ph, pm, pl;
            ph.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);   ph.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
            pm.x = (unsigned)mid[0] | ((unsigned)mid[1] << 16); pm.y = (unsigned)mid[2] | ((unsigned)mid[3] << 16);
            pl.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16);   pl.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
            *reinterpret_cast<uint2*>(buf + h_dst[it]) = ph;
            *reinterpret_cast<uint2*>(buf + h_dst[it] + 32) = pm;
            *reinterpret_cast<uint2*>(buf + h_dst[it] + 64) = pl;
        }
    };
    auto load_w = [&](f32x4 (&wreg)[C::NW], int stage) {       // stage = absolute (chunk*9 + tap)
        const int chunk = stage / 9, tap = stage - chunk * 9;
#pragma unroll
        for (int it = 0; it < C::NW; ++it)      // unconditional: a predicated load makes hipcc wait vmcnt(0) on the spot
            wreg[it] = *reinterpret_cast<const f32x4*>(wq + tap * tap_stride + w_src[it] + (size_t)chunk * 96);
    };
    auto store_w = [&](const f32x4 (&wreg)[C::NW], unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < C::NW; ++it)
            if (w_dst[it] >= 0) *reinterpret_cast<f32x4*>(buf + w_dst[it]) = wreg[it];
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int chunk_begin = ksplit_idx * chunks_per_split;
    int chunk_end = chunk_begin + chunks_per_split;
    if (chunk_end > nchunks) chunk_end = nchunks;
    const int st0 = chunk_begin * 9;
    const int nstages = (chunk_end - chunk_begin) * 9;

    // prologue: stage 0 into LDS
    load_halo(chunk_begin);
    load_w(wregA, st0);
    store_halo(halo0);
    store_w(wregA, wts0);
    __syncthreads();

    const int a_base = ((MT * wm) * XHC + li) * XROW + lh * 16;
    const int b_base = (64 * wn + li) * XROW + lh * 16;

    int chunk = chunk_begin, tap = 0, tr = 0, ts = 0;
    for (int s = 0; s < nstages; ++s) {
        const bool has_next = (s + 1) < nstages;
        int nchunk = chunk, ntap = tap + 1;
        if (ntap == 9) { ntap = 0; nchunk = chunk + 1; }
        // next stage's weights: always issued (the last stage re-loads its own, harmlessly)
        load_w(wregA, has_next ? nchunk * 9 + ntap : chunk * 9 + tap);
        if (has_next && ntap == 0) load_halo(nchunk);
        const unsigned char* hal = halo0 + a_base + (tr * XHC + ts) * XROW;
        const unsigned char* wt = wts0 + (s & 1) * C::WT_B + b_base;
        bf16x8 ah[MT], am[MT], al[MT], bh[2], bm[2], bl[2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ah[mt] = *reinterpret_cast<const bf16x8*>(hal + mt * XHC * XROW);
            am[mt] = *reinterpret_cast<const bf16x8*>(hal + mt * XHC * XROW + 32);
            al[mt] = *reinterpret_cast<const bf16x8*>(hal + mt * XHC * XROW + 64);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bh[nt] = *reinterpret_cast<const bf16x8*>(wt + nt * 32 * XROW);
            bm[nt] = *reinterpret_cast<const bf16x8*>(wt + nt * 32 * XROW + 32);
            bl[nt] = *reinterpret_cast<const bf16x8*>(wt + nt * 32 * XROW + 64);
        }
        // term-major, tile-minor: consecutive MFMAs hit different accumulators (a dependent accumulate
        // needs ~2x the issue interval); smallest terms first
#define X6_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define X6_TERM(A, B)                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
                acc[mt][nt] = X6_MFMA(A[mt], B[nt], acc[mt][nt]);
        X6_TERM(al, bh)
        X6_TERM(ah, bl)
        X6_TERM(am, bm)
        X6_TERM(am, bh)
        X6_TERM(ah, bm)
        X6_TERM(ah, bh)
#undef X6_TERM
#undef X6_MFMA
        if (has_next) {
            store_w(wregA, wts0 + ((s + 1) & 1) * C::WT_B);
            if (ntap == 0) {                 // chunk seam: everyone must be done reading the old halo
                __syncthreads();
                store_halo(halo0);
            }
        }
        __syncthreads();
        chunk = nchunk; tap = ntap;
        ts += 1; if (ts == 3) { ts = 0; tr += 1; if (tr == 3) tr = 0; }
    }

    // ---- epilogue (as conv3x3_mfma_kernel; the wave owns MT consecutive rows) --------------------
    const int orow = y0 + MT * wm;
    if (ws != nullptr) {
        float* part = ws + (size_t)ksplit_idx * H * W * Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + 64 * wn + 32 * nt + li;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = orow + mt;
                if (yy >= H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (xx < W) part[((size_t)yy * W + xx) * Cout + co] = acc[mt][nt][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + 64 * wn + 32 * nt + li;
        const float bv = bias[co];
        if (!POOL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = orow + mt;
                if (yy >= H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (xx < W) {
                        float v = acc[mt][nt][r] + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        y[((size_t)yy * W + xx) * Cout + co] = v;
                    }
                }
            }
        } else {
            const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
            for (int mp = 0; mp < MT; mp += 2) {
                const int py = (orow + mp) >> 1;
                if (py >= Hp) continue;
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int px = (x0 + (r & 3) + 8 * (r >> 2) + 4 * lh) >> 1;
                    if (px < Wp) {
                        float v = fmaxf(fmaxf(acc[mp][nt][r], acc[mp][nt][r + 1]),
                                        fmaxf(acc[mp + 1][nt][r], acc[mp + 1][nt][r + 1])) + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        y[((size_t)py * Wp + px) * Cout + co] = v;
                    }
                }
            }
        }
    }
}

// ---- software-pipelined variant (cout % 128 == 0: block = 4 rows x 32 cols x 128 couts, wave = 2 rows x 64 couts) ----
// Same tiles, same six-term product, same LDS row layout; the K loop follows the schedule that took the exact-f32 kernel
// from 85 % to 99 % of the pipe inside the loop (csrc/conv.hip): every staging instruction is issued in the shadow of the
// wave's own MFMAs, one barrier per stage, in the MIDDLE of the stage.  A stage here is only 24 MFMAs x 32 cycles, so
//   * the tile of stage s+4 is loaded (global -> registers) during stage s, written to LDS during stage s+2 and read as
//     fragments during stage s+3: three rotating register sets (the 9 unrolled taps are a multiple of 3);
//   * the fragment registers are single-buffered and refilled as soon as the last term that needs them has been issued:
//     term order mm, lh, hl | barrier | (read al', bl') mh (read am') hm (read bm') hh (read ah', bh'), so the next stage
//     starts with the term whose operands arrived first;
//   * the halo is double-buffered (2 blocks per CU): chunk c+1 is loaded in tap 4, split and written in tap 7, first read in tap 8.
template <bool POOL>
__global__ __launch_bounds__(256, 2)
void conv3x3_x6p_kernel(const float* __restrict__ x, const unsigned char* __restrict__ wq,
                        const float* __restrict__ bias, float* __restrict__ y,
                        int H, int W, int Cin, int Cout, int relu, int cout_tiles, int chunks_per_split,
                        float* __restrict__ ws)
{
    using C = X6Cfg<2, 2, 2>;
    constexpr int MT = 2, WN = 2;
    static_assert(C::NWP == 3 * 256, "three 16-byte weight pieces per thread per stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_x6[];
    unsigned char* const halo0 = smem_x6;                          // 2 halo buffers
    unsigned char* const wts0 = smem_x6 + 2 * C::HALO_B;           // 2 weight buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int x0 = blockIdx.x * 32;
    const int y0 = blockIdx.y * C::TR;
    const int ksplit_idx = blockIdx.z / cout_tiles;
    const int n0 = (blockIdx.z - ksplit_idx * cout_tiles) * C::BN;
    const int nchunks = Cin >> 4;

    // staging addresses: byte offsets from uniform bases; out-of-image halo pieces are loaded from offset 0 and zeroed
    // by a select before the split (no predicated loads in the loop)
    unsigned h_src[C::NH];
    int h_dst[C::NH];
    unsigned h_inb = 0;
#pragma unroll
    for (int it = 0; it < C::NH; ++it) {
        int q = tid + 256 * it;
        if (q >= C::NHP) q = C::NHP - 1;                            // surplus threads duplicate the last piece
        const int pix = q >> 2, p = q & 3;
        const int hy = pix / XHC, hx = pix - hy * XHC;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_dst[it] = pix * XROW + p * 8;
        h_src[it] = inb ? (unsigned)(((gy * W + gx) * Cin + 4 * p) * 4) : 0u;
        h_inb |= (inb ? 1u : 0u) << it;
    }
    unsigned w_src[3];
    int w_dst[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int q = tid + 256 * it;
        const int row = q / 6, j = q - row * 6;
        w_src[it] = (unsigned)((size_t)(n0 + row) * nchunks * 96 + j * 16);
        w_dst[it] = row * XROW + j * 16;
    }
    const size_t tap_stride = (size_t)Cout * nchunks * 96;

    f32x4 hreg[C::NH];
    f32x4 wset[3][3];                                               // [register set][piece]

    auto load_halo = [&](int chunk) {
        const char* base = reinterpret_cast<const char*>(x + chunk * 16);
#pragma unroll
        for (int it = 0; it < C::NH; ++it) hreg[it] = *reinterpret_cast<const f32x4*>(base + h_src[it]);
    };
    auto store_halo = [&](unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < C::NH; ++it) {
            f32x4 v = hreg[it];
            if (!((h_inb >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            u16 hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) split3(v[j], hi[j], mid[j], lo[j]);
            uint2 ph, pm, pl;
            ph.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);   ph.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
            pm.x = (unsigned)mid[0] | ((unsigned)mid[1] << 16); pm.y = (unsigned)mid[2] | ((unsigned)mid[3] << 16);
            pl.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16);   pl.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
            *reinterpret_cast<uint2*>(buf + h_dst[it]) = ph;
            *reinterpret_cast<uint2*>(buf + h_dst[it] + 32) = pm;
            *reinterpret_cast<uint2*>(buf + h_dst[it] + 64) = pl;
        }
    };
    auto load_w = [&](f32x4 (&wr)[3], int chunk, int tap) {
        const unsigned char* base = wq + (size_t)tap * tap_stride + (size_t)chunk * 96;
#pragma unroll
        for (int it = 0; it < 3; ++it) wr[it] = *reinterpret_cast<const f32x4*>(base + w_src[it]);
    };
    auto store_w = [&](const f32x4 (&wr)[3], unsigned char* buf) {
#pragma unroll
        for (int it = 0; it < 3; ++it) *reinterpret_cast<f32x4*>(buf + w_dst[it]) = wr[it];
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int chunk_begin = ksplit_idx * chunks_per_split;
    int chunk_end = chunk_begin + chunks_per_split;
    if (chunk_end > nchunks) chunk_end = nchunks;
    const int last_chunk = chunk_end - 1;

    const int a_base = ((MT * wm) * XHC + li) * XROW + lh * 16;
    const int b_base = (64 * wn + li) * XROW + lh * 16;

    bf16x8 ah[MT], am[MT], al[MT], bh[2], bm[2], bl[2];
    auto read_a = [&](bf16x8 (&dst)[MT], const unsigned char* hal, int plane) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) dst[mt] = *reinterpret_cast<const bf16x8*>(hal + mt * XHC * XROW + 32 * plane);
    };
    auto read_b = [&](bf16x8 (&dst)[2], const unsigned char* wt, int plane) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) dst[nt] = *reinterpret_cast<const bf16x8*>(wt + nt * 32 * XROW + 32 * plane);
    };

    // prologue: tiles of stages 0 and 1 in LDS, tiles of stages 2 and 3 in register sets 2 and 0, F(0) in registers
    {
        unsigned char* hb = halo0 + (chunk_begin & 1) * C::HALO_B;
        load_halo(chunk_begin);
        load_w(wset[0], chunk_begin, 0);
        load_w(wset[1], chunk_begin, 1);
        store_halo(hb);
        store_w(wset[0], wts0);
        store_w(wset[1], wts0 + C::WT_B);
        load_w(wset[2], chunk_begin, 2);
        load_w(wset[0], chunk_begin, 3);
        __syncthreads();
        read_a(ah, hb + a_base, 0); read_a(am, hb + a_base, 1); read_a(al, hb + a_base, 2);
        read_b(bh, wts0 + b_base, 0); read_b(bm, wts0 + b_base, 1); read_b(bl, wts0 + b_base, 2);
    }

#define X6_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define X6_TERM(A, B)                                                                                   \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt)                                               \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                            \
                acc[mt][nt] = X6_MFMA(A[mt], B[nt], acc[mt][nt]);

    int wpar = 0;                                                   // weight buffer of the current stage
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const int nchunk = chunk < last_chunk ? chunk + 1 : last_chunk;       // clamped: surplus prefetches are harmless
        unsigned char* const hal_cur = halo0 + (chunk & 1) * C::HALO_B;
        unsigned char* const hal_nxt = halo0 + ((chunk + 1) & 1) * C::HALO_B;
        auto stage = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            // stage s = (chunk, tap).  Tile s+1 sits in weight buffer wpar^1 (+ halo), tile s+2 goes to buffer wpar.
            unsigned char* const wt_s2 = wts0 + wpar * C::WT_B;
            const unsigned char* const wt_s1 = wts0 + (wpar ^ 1) * C::WT_B + b_base;
            constexpr int t1 = (tap + 1) % 9;
            const unsigned char* const hal_s1 = (tap == 8 ? hal_nxt : hal_cur) + a_base + ((t1 / 3) * XHC + (t1 % 3)) * XROW;
            // ---- first three terms; the tile of stage s+4 starts its way from global memory -----------------
            if (tap <= 4) load_w(wset[(tap + 4) % 3], chunk, tap + 4); else load_w(wset[(tap + 4) % 3], nchunk, tap - 5);
            if (tap == 4) load_halo(nchunk);
            X6_TERM(am, bm)
            X6_TERM(al, bh)
            X6_TERM(ah, bl)
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();            // tile s+1 complete; every wave has issued (and received) all of F(s)
            __builtin_amdgcn_sched_barrier(0);
            // ---- refill the fragment registers as their last use passes; tile s+2 goes to LDS ----------------
            read_a(al, hal_s1, 2);
            read_b(bl, wt_s1, 2);
            store_w(wset[(tap + 2) % 3], wt_s2);
            if (tap == 7) store_halo(hal_nxt);
            X6_TERM(am, bh)
            read_a(am, hal_s1, 1);
            X6_TERM(ah, bm)
            read_b(bm, wt_s1, 1);
            X6_TERM(ah, bh)
            read_a(ah, hal_s1, 0);
            read_b(bh, wt_s1, 0);
            __builtin_amdgcn_sched_barrier(0);
            wpar ^= 1;
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
    }
#undef X6_TERM
#undef X6_MFMA

    // ---- epilogue (as conv3x3_x6_kernel) -------------------------------------------------------------------------
    const int orow = y0 + MT * wm;
    if (ws != nullptr) {
        float* part = ws + (size_t)ksplit_idx * H * W * Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + 64 * wn + 32 * nt + li;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = orow + mt;
                if (yy >= H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (xx < W) part[((size_t)yy * W + xx) * Cout + co] = acc[mt][nt][r];
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int co = n0 + 64 * wn + 32 * nt + li;
        const float bv = bias[co];
        if (!POOL) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int yy = orow + mt;
                if (yy >= H) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int xx = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (xx < W) {
                        float v = acc[mt][nt][r] + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        y[((size_t)yy * W + xx) * Cout + co] = v;
                    }
                }
            }
        } else {
            const int Hp = H >> 1, Wp = W >> 1;
            const int py = orow >> 1;
            if (py < Hp) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const int px = (x0 + (r & 3) + 8 * (r >> 2) + 4 * lh) >> 1;
                    if (px < Wp) {
                        float v = fmaxf(fmaxf(acc[0][nt][r], acc[0][nt][r + 1]),
                                        fmaxf(acc[1][nt][r], acc[1][nt][r + 1])) + bv;
                        if (relu) v = fmaxf(v, 0.f);
                        y[((size_t)py * Wp + px) * Cout + co] = v;
                    }
                }
            }
        }
    }
}

// OIHW fp32 [cout][cin][3][3] -> [tap][cout][cin/16][plane(hi,mid,lo)][16] bf16
__global__ void pack_conv3x3_x6_kernel(const float* __restrict__ w, u16* __restrict__ wq, int cout, int cin)
{
    const size_t total = (size_t)9 * cout * cin;
    const int nchunks = cin >> 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % cin);
        const size_t t = i / cin;
        const int o = (int)(t % cout);
        const int tap = (int)(t / cout);
        u16 hi, mid, lo;
        split3(w[((size_t)o * cin + ci) * 9 + tap], hi, mid, lo);
        const size_t rec = (((size_t)tap * cout + o) * nchunks + (ci >> 4)) * 48;     // 48 bf16 = 96 B
        wq[rec + (ci & 15)] = hi;
        wq[rec + 16 + (ci & 15)] = mid;
        wq[rec + 32 + (ci & 15)] = lo;
    }
}

template <int WM, int WN, int MT, bool POOL>
static int launch_x6_cfg(const float* x, const void* wq, const float* b, float* y, int H, int W,
                         int cin, int cout, int relu, int ksplit, float* ws, hipStream_t s)
{
    using C = X6Cfg<WM, WN, MT>;
    auto kern = conv3x3_x6_kernel<WM, WN, MT, POOL>;
    FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
    const int cout_tiles = cout / C::BN;
    const int nchunks = cin / 16;
    dim3 grid(cdiv(W, 32), cdiv(H, C::TR), cout_tiles * ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, x, static_cast<const unsigned char*>(wq), b, y, H, W, cin,
                       cout, relu, cout_tiles, cdiv(nchunks, ksplit), ksplit > 1 ? ws : (float*)nullptr);
    return check_launch();
}

template <bool POOL>
static int launch_x6p(const float* x, const void* wq, const float* b, float* y, int H, int W,
                      int cin, int cout, int relu, int ksplit, float* ws, hipStream_t s)
{
    using C = X6Cfg<2, 2, 2>;
    constexpr size_t lds = (size_t)2 * C::HALO_B + 2 * C::WT_B;
    auto kern = conv3x3_x6p_kernel<POOL>;
    FRCNN_MAX_LDS_ONCE(kern, lds);
    const int cout_tiles = cout / C::BN;
    const int nchunks = cin / 16;
    dim3 grid(cdiv(W, 32), cdiv(H, C::TR), cout_tiles * ksplit);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, static_cast<const unsigned char*>(wq), b, y, H, W, cin,
                       cout, relu, cout_tiles, cdiv(nchunks, ksplit), ksplit > 1 ? ws : (float*)nullptr);
    return check_launch();
}

int launch_conv_splitk_finish(const float* ws, int ksplit, const float* b, float* y, int H, int W, int cout, int relu,
                              int pool, hipStream_t s);               // conv.hip

// Tiles as in conv.hip: cout % 128 == 0 -> block 4 rows x 32 cols x 128 couts, cout == 64 -> 8 rows x 32 x 64
// (wave = 2 rows x 32 cols x 64 couts, 24 MFMAs per stage).  A 4-rows-per-wave variant (MT = 4, 48 MFMAs
// per stage) measured no faster: the kernel is bound by per-block latency.
static int x6_ksplit(int H, int W, int cin, int cout)
{
    const bool narrow = (cout % 128) != 0;
    const int bn = narrow ? 64 : 128;
    const int rows = narrow ? 8 : 4;
    const int blocks = cdiv(W, 32) * cdiv(H, rows) * (cout / bn);
    const int nchunks = cin / 16;
    static int env_target = -1;
    if (env_target < 0) {
        const char* e = frcnn_knob("FRCNN_X6_BLOCKS_TARGET");       // tuning knob
        env_target = e ? atoi(e) : 0;
        if (env_target < 0) env_target = 0;
    }
    // otherwise the regime-dependent target of conv.hip (frcnn_forward_params.conv_blocks_target; 1280 by default)
    const int target = env_target > 0 ? env_target : conv3x3_blocks_target();
    if (blocks * 2 > target) return 1;
    if ((size_t)H * W * cout * sizeof(float) > ((size_t)40 << 20)) return 1;
    int k = 1;
    while (k * 2 * blocks <= target && nchunks % (k * 2) == 0 && nchunks / (k * 2) >= 2) k *= 2;
    return k;
}

size_t conv3x3_x6_workspace_bytes(int H, int W, int cin, int cout)
{
    if (cin % 16 != 0 || cout % 64 != 0 || H < 1 || W < 1) return 0;
    const int k = x6_ksplit(H, W, cin, cout);
    return k > 1 ? (size_t)k * H * W * cout * sizeof(float) : 0;
}

int launch_conv3x3_x6(const float* x, const void* wq, const float* b, float* y, int H, int W, int cin, int cout,
                      unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (cin % 16 != 0 || cout % 64 != 0 || H < 1 || W < 1) return FRCNN_EINVAL;
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    const bool pool = (flags & FRCNN_POOL2) != 0;
    if (pool && (H < 2 || W < 2)) return FRCNN_EINVAL;
    int ksplit = x6_ksplit(H, W, cin, cout);
    if (ksplit > 1 && (ws == nullptr || ws_bytes < (size_t)ksplit * H * W * cout * sizeof(float))) ksplit = 1;
    float* wsf = static_cast<float*>(ws);
    const bool piped = (cout % 128 == 0);       // conv3x3_x6p_kernel; the cout = 64 layer (conv1_2) keeps the 8-row x 64 tile
    if (ksplit > 1) {
        int rc = (cout % 128 != 0) ? launch_x6_cfg<4, 1, 2, false>(x, wq, b, y, H, W, cin, cout, relu, ksplit, wsf, s)
                 : piped           ? launch_x6p<false>(x, wq, b, y, H, W, cin, cout, relu, ksplit, wsf, s)
                                   : launch_x6_cfg<2, 2, 2, false>(x, wq, b, y, H, W, cin, cout, relu, ksplit, wsf, s);
        if (rc) return rc;
        return launch_conv_splitk_finish(wsf, ksplit, b, y, H, W, cout, relu, pool ? 1 : 0, s);
    }
    if (piped)
        return pool ? launch_x6p<true>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s)
                    : launch_x6p<false>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s);
    if (cout % 128 != 0) {
        return pool ? launch_x6_cfg<4, 1, 2, true>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s)
                    : launch_x6_cfg<4, 1, 2, false>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s);
    }
    return pool ? launch_x6_cfg<2, 2, 2, true>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s)
                : launch_x6_cfg<2, 2, 2, false>(x, wq, b, y, H, W, cin, cout, relu, 1, nullptr, s);
}

int launch_pack_conv3x3_x6(const float* w, void* wq, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 16 || cin % 16 != 0) return FRCNN_EINVAL;
    const size_t total = (size_t)9 * cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv3x3_x6_kernel, dim3(blocks), dim3(256), 0, s, w, static_cast<u16*>(wq), cout, cin);
    return check_launch();
}

}  // namespace frcnn
