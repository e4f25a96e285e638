// conv.hip -- 3x3 "same" convolutions of the VGG-16 feature extractor and the RPN trunk
// (reference: models/vgg16.py:27-47,76-96 and models/rpn.py:39,88 -- there nn.Conv2d on cuDNN).
//
// conv3x3_mfma_kernel: implicit GEMM on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32):
//   M = output pixels (32 consecutive x of one image row per MFMA tile)
//   N = output channels (32 per MFMA tile)
//   K = 9 taps x Cin, walked as (16-channel chunk) x (tap)
// Activations are NHWC so a pixel's 16-channel chunk is one 64-B run: the block stages the
// (rows+2) x 34 pixel halo of the chunk in LDS once and re-reads it for all 9 taps; the
// tap's [BN][16] weight slice is staged per (chunk, tap) stage.  Both tiles are double buffered
// so there is ONE barrier per stage (32 MFMAs = 2048 matrix-pipe cycles per wave).
// LDS rows are padded 16 -> 20 floats: a ds_read_b128 lane group then touches 16 distinct
// 16-B slots (5*i mod 16 is a bijection), i.e. conflict-free for both operands.
// K-order trick: MFMA 32x32x2 takes k = lane>>5.  The lane half h owns channels
// [8g+4h, 8g+4h+4) of every 8-channel group g, so one ds_read_b128 feeds four MFMAs for A and
// one for B (a dot product does not care about the order of k as long as A and B agree).
// Epilogue: bias + ReLU (+ the 2x2/stride-2 max-pool of vgg16.py:78,82,87,92 done in
// registers: the wave owns two image rows, and the accumulator layout puts x, x+1 in
// adjacent registers of one lane), then NHWC stores of 128 B per half-wave.
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace frcnn {

static constexpr int LDK = 20;   // padded LDS row (floats) of a 16-channel chunk
static constexpr int HC  = 34;   // halo columns = 32 + 2

template <int WM, int WN>
struct ConvCfg {
    static constexpr int TR  = 2 * WM;        // image rows per block
    static constexpr int BN  = 64 * WN;       // output channels per block
    static constexpr int HR  = TR + 2;        // halo rows
    static constexpr int HALO_F = HR * HC * LDK;
    static constexpr int WT_F   = BN * LDK;
    static constexpr int NHP = HR * HC * 4;   // 16-B halo pieces per chunk
    static constexpr int NH  = (NHP + 255) / 256;
    static constexpr int NW  = BN * 4 / 256;  // 16-B weight pieces per thread per stage
    static constexpr size_t LDS_BYTES = (size_t)(2 * HALO_F + 2 * WT_F) * sizeof(float);
};

// ---- the K loop is software pipelined ------------------------------------------------------------
// Measured on the skeletons in tools/micro/: a wave that has MFMAs ready starves the younger waves of
// its SIMD, so the staging instructions of a wave are NOT hidden by the other resident waves -- they
// must be issued in the shadow of the wave's own MFMAs (a loop that does staging, then 32 MFMAs, then
// a barrier tops out at 85 % of the pipe however many waves are resident; this order reaches 99 % in
// the loop, 146 TFLOP/s on a long-K layer).  Per stage (chunk c, tap t):
//   F0(s) is in registers | read F1(s) | LDS-write the tile of stage s+1 | global-load the tile of stage s+2 |
//   16 MFMAs on F0 with those instructions interleaved one per MFMA | barrier |
//   read F0(s+1) interleaved with the 16 MFMAs on F1
// One barrier per stage still suffices: every read of a buffer happens before the barrier of the stage
// that owns it, every write to it after the barrier of the previous stage.  The 9 taps are unrolled so each
// stage is one basic block (the halo is loaded in tap 7 and written in tap 8 of the previous chunk).
template <int WM, int WN, bool POOL>
__global__ __launch_bounds__(256)
void conv3x3_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                         const float* __restrict__ bias, float* __restrict__ y,
                         int H, int W, int Cin, int Cout, int relu, int cout_tiles, int chunks_per_split,
                         float* __restrict__ ws)
{
    using C = ConvCfg<WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const halo0 = smem;
    float* const wts0  = smem + 2 * C::HALO_F;

    const int tid  = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int x0 = blockIdx.x * 32;
    const int y0 = blockIdx.y * C::TR;
    const int ksplit_idx = blockIdx.z / cout_tiles;
    const int n0 = (blockIdx.z - ksplit_idx * cout_tiles) * C::BN;

    // staging addresses (32-bit element offsets, as in conv3x3_mfma_kernel).  Out-of-image halo pieces are loaded from
    // offset 0 (always valid) and zeroed by a select before the LDS write: no predicated loads in the pipelined loop.
    int h_src[C::NH], h_dst[C::NH];
    unsigned h_inb = 0;
#pragma unroll
    for (int it = 0; it < C::NH; ++it) {
        int q = tid + 256 * it;
        if (q >= C::NHP) q = C::NHP - 1;                       // surplus threads duplicate the last piece (same value, same slot)
        const int pix = q >> 2, p = q & 3;
        const int hy = pix / HC, hx = pix - hy * HC;
        const int gy = y0 - 1 + hy, gx = x0 - 1 + hx;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_dst[it] = pix * LDK + 4 * p;
        h_src[it] = inb ? ((gy * W + gx) * Cin + 4 * p) * 4 : 0;       // BYTE offset from the (uniform) chunk base
        h_inb |= (inb ? 1u : 0u) << it;
    }
    int w_src[C::NW], w_dst[C::NW];
#pragma unroll
    for (int it = 0; it < C::NW; ++it) {
        const int q = tid + 256 * it;
        const int o = q >> 2, p = q & 3;
        w_src[it] = ((n0 + o) * Cin + 4 * p) * 4;                       // BYTE offset from the (uniform) tap/chunk base
        w_dst[it] = o * LDK + 4 * p;
    }
    const int tap_stride = Cout * Cin;

    f32x4 hreg[C::NH];
    f32x4 wreg[C::NW];
    auto load_halo = [&](int chunk) {
        // uniform base in SGPRs + one 32-bit per-lane byte offset (global_load saddr form: no 64-bit address VGPRs)
        const char* base = reinterpret_cast<const char*>(x + chunk * 16);
#pragma unroll
        for (int it = 0; it < C::NH; ++it) hreg[it] = *reinterpret_cast<const f32x4*>(base + (unsigned)h_src[it]);
    };
    auto store_halo = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < C::NH; ++it) {
            f32x4 v = hreg[it];
            if (!((h_inb >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(buf + h_dst[it]) = v;
        }
    };
    auto load_w = [&](int chunk, int tap) {
        const char* base = reinterpret_cast<const char*>(wp + (size_t)tap * tap_stride + chunk * 16);
#pragma unroll
        for (int it = 0; it < C::NW; ++it) wreg[it] = *reinterpret_cast<const f32x4*>(base + (unsigned)w_src[it]);
    };
    auto store_w = [&](float* buf) {
#pragma unroll
        for (int it = 0; it < C::NW; ++it) *reinterpret_cast<f32x4*>(buf + w_dst[it]) = wreg[it];
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int chunk_begin = ksplit_idx * chunks_per_split;
    int chunk_end = chunk_begin + chunks_per_split;
    if (chunk_end > (Cin >> 4)) chunk_end = Cin >> 4;
    const int last_chunk = chunk_end - 1;

    const int a_base = ((2 * wm) * HC + li) * LDK + 4 * lh;
    const int b_base = (64 * wn + li) * LDK + 4 * lh;

    // prologue: tile of stage 0 in LDS, weights of stage 1 in registers, F0 of stage 0 in registers
    load_halo(chunk_begin);
    load_w(chunk_begin, 0);
    store_halo(halo0 + (chunk_begin & 1) * C::HALO_F);
    store_w(wts0);
    load_w(chunk_begin, 1);
    __syncthreads();
    f32x4 a0[2], b0[2], a1[2], b1[2];
    {
        const float* hal = halo0 + (chunk_begin & 1) * C::HALO_F + a_base;
        const float* wt = wts0 + b_base;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a0[mt] = *reinterpret_cast<const f32x4*>(hal + mt * HC * LDK);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b0[nt] = *reinterpret_cast<const f32x4*>(wt + nt * 32 * LDK);
    }

    int wpar = 0;                                    // weight buffer of the current stage
    for (int chunk = chunk_begin; chunk < chunk_end; ++chunk) {
        const int nchunk = chunk < last_chunk ? chunk + 1 : last_chunk;      // clamped: surplus prefetches are harmless
        float* const hal_cur = halo0 + (chunk & 1) * C::HALO_F;
        float* const hal_nxt = halo0 + ((chunk + 1) & 1) * C::HALO_F;
        auto stage = [&](auto tap_c) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int tr = tap / 3, ts = tap % 3;
            const float* hal = hal_cur + a_base + (tr * HC + ts) * LDK;
            const float* wt = wts0 + wpar * C::WT_F + b_base;
            float* const wt_nxt = wts0 + (wpar ^ 1) * C::WT_F;
            // b. second-half fragments of this stage
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a1[mt] = *reinterpret_cast<const f32x4*>(hal + mt * HC * LDK + 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) b1[nt] = *reinterpret_cast<const f32x4*>(wt + nt * 32 * LDK + 8);
            // c. tile of stage s+1 -> LDS (weights every stage; the next chunk's halo in tap 8)
            store_w(wt_nxt);
            if (tap == 8) store_halo(hal_nxt);
            // d. global loads for stage s+2
            if (tap <= 6) load_w(chunk, tap + 2); else load_w(nchunk, tap - 7);
            if (tap == 7) load_halo(nchunk);
            // e. first half: 16 MFMAs on F0
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[mt][kk], b0[nt][kk], acc[mt][nt], 0, 0, 0);
            // one staging instruction in the shadow of each MFMA
            {
                constexpr int n_wr = C::NW + (tap == 8 ? C::NH : 0);
                constexpr int n_ld = C::NW + (tap == 7 ? C::NH : 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
                for (int q = 0; q < n_wr; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
                for (int q = 0; q < n_ld; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
            // f. the tile of stage s+1 is complete; nobody reads this stage's buffers any more
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            // g. first-half fragments of stage s+1
            {
                constexpr int ntr = (tap + 1) % 9 / 3, nts = (tap + 1) % 3;
                const float* nhal = (tap == 8 ? hal_nxt : hal_cur) + a_base + (ntr * HC + nts) * LDK;
                const float* nwt = wt_nxt + b_base;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) a0[mt] = *reinterpret_cast<const f32x4*>(nhal + mt * HC * LDK);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) b0[nt] = *reinterpret_cast<const f32x4*>(nwt + nt * 32 * LDK);
            }
            // h. second half: 16 MFMAs on F1
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[mt][kk], b1[nt][kk], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            wpar ^= 1;
        };
        stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{}); stage(std::integral_constant<int, 2>{});
        stage(std::integral_constant<int, 3>{}); stage(std::integral_constant<int, 4>{}); stage(std::integral_constant<int, 5>{});
        stage(std::integral_constant<int, 6>{}); stage(std::integral_constant<int, 7>{}); stage(std::integral_constant<int, 8>{});
    }

    // ---- epilogue ------------------------------------------------------------------------------
    // acc[mt][nt][r] = out[row y0+2wm+mt][col x0 + (r&3)+8(r>>2)+4lh][cout n0+64wn+32nt+li]
    const int orow = y0 + 2 * wm;
    if (ws != nullptr) {
        float* part = ws + (size_t)ksplit_idx * H * W * Cout;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int co = n0 + 64 * wn + 32 * nt + li;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
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
            for (int mt = 0; mt < 2; ++mt) {
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

// Split-K finish: y = act(bias + sum_k part[k]) (+ 2x2 max-pool), fixed summation order.
__global__ __launch_bounds__(256)
void conv_splitk_finish_kernel(const float* __restrict__ ws, int ksplit, const float* __restrict__ bias,
                               float* __restrict__ y, int H, int W, int Cout, int relu, int pool)
{
    const int C4 = Cout >> 2;
    const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
    const size_t total = (size_t)Ho * Wo * C4;
    const size_t plane = (size_t)H * W * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % Wo), oy = (int)(p / Wo);
        const f32x4 bv = reinterpret_cast<const f32x4*>(bias)[c4];
        f32x4 best;
        const int np = pool ? 4 : 1;
        for (int q = 0; q < np; ++q) {
            const int iy = pool ? 2 * oy + (q >> 1) : oy, ix = pool ? 2 * ox + (q & 1) : ox;
            const size_t off = ((size_t)iy * W + ix) * Cout + 4 * c4;
            f32x4 v = *reinterpret_cast<const f32x4*>(ws + off);
            for (int k = 1; k < ksplit; ++k) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(ws + k * plane + off);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += t[j];
            }
            if (q == 0) best = v;
            else {
#pragma unroll
                for (int j = 0; j < 4; ++j) best[j] = fmaxf(best[j], v[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = best[j] + bv[j];
            best[j] = relu ? fmaxf(t, 0.f) : t;
        }
        reinterpret_cast<f32x4*>(y)[i] = best;
    }
}

// First layer (models/vgg16.py:27,76): Cin = 3, K = 27 is too thin for the matrix pipe and the
// layer is bound by its 4*H*W*cout-byte output write.  One thread = one pixel, 16 output channels
// at a time; the 27 x 16 weights of a channel group are wave-uniform (scalar loads).
__global__ __launch_bounds__(256)
void conv3x3_c3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                       const float* __restrict__ bias, float* __restrict__ y,
                       int H, int W, int Cout, int relu)
{
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int yy = pix / W, xx = pix - yy * W;
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int gy = yy + r - 1, gx = xx + s - 1;
                const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
                in[ci * 9 + r * 3 + s] = inb ? x[((size_t)ci * H + gy) * W + gx] : 0.f;
            }
    // all channel groups of the pixel from ONE thread: the image is read once and the pixel's
    // Cout*4 output bytes leave the wave together (the 4-blocks-per-pixel version measured
    // 2.35x write amplification: each 128-B line was completed by two blocks at different times)
    for (int og = 0; og < (Cout >> 4); ++og) {
        float acc[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[o] = 0.f;
        const float* wg = wp + og * 16;
#pragma unroll
        for (int k = 0; k < 27; ++k)
#pragma unroll
            for (int o = 0; o < 16; ++o) acc[o] = fmaf(in[k], wg[k * Cout + o], acc[o]);
        float* out = y + (size_t)pix * Cout + og * 16;
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float t = acc[o4 * 4 + j] + bias[og * 16 + o4 * 4 + j];
                v[j] = relu ? fmaxf(t, 0.f) : t;
            }
            *reinterpret_cast<f32x4*>(out + 4 * o4) = v;
        }
    }
}

// The same layer for Cout = 64 (conv1_1).  Thread = (four adjacent pixels of a row, cout quad q = t & 15): the 16 lanes of a pixel write its
// 256 output bytes together (the pixel-per-thread kernel above stores 16 bytes at a 256-byte lane stride: 81 us for the 153.6 MB of a
// 600x1000 image); the input patch of a tile is staged in LDS (zero padding folded in), the quad's 27 x 4 weights live in registers.
// cmax_out (optional): the per-pixel maximum over the 64 output channels, [H][W] -- the scale source of an f32x3 layer that consumes this
// tensor (csrc/wino_x3f.hip).  The 16 lanes of a pixel hold all of its channels, so it is four DPP row rotations and ONE plain store per
// pixel: no atomics, no zeroed buffer, and the consumer does not read the 153.6 MB tensor once more (pixel_absmax_kernel).
// Round 5: a PERSISTENT kernel (two blocks per CU walk the 2-row x 64-pixel tiles of the image).  What round 2's kernel (block = 8 rows x
// 64 pixels, one block per tile) paid on a 600x1000 image (51-53 us = 2.9 TB/s of output; issue-bound estimate of its instruction stream: 28 us):
// 1200 blocks on 768 block slots = two rounds with the second half empty, a prologue per block (27 weight quads, the input patch with two
// integer divisions per element) that nothing overlapped, and 15 LDS reads per 54 packed FMAs.  Here the weights are loaded once per block, the next tile's
// input patch is fetched into registers before the current tile is computed (its element -> (channel, row, column) split is tile
// independent: done once), 4800 tiles over 512 blocks leave a 7 % tail, and a thread owns FOUR ADJACENT pixels of a row so that the 3 x 6
// input values of a (channel, row) are one ds_read_b128 + one ds_read_b64 for 24 packed FMAs.  The fmaf order over k = ci*9 + r*3 + s per
// output is the one of conv3x3_c3_kernel: bit-identical results (tests/test_kernels_gpu.py compares the two).  Measured: 36.6 us alone
// (4.2 TB/s), two blocks per CU (three: 38.0); one image at a time +0.5 %; with three images in flight the headline does not move
// (846 / 846 images/sec, A/B on one box) -- conv1_1 runs in the gaps of the Winograd launches there.
typedef float c3_f32x2 __attribute__((ext_vector_type(2)));
template <bool RELU, bool CMAX, int BPC>
__global__ __launch_bounds__(256, BPC)
void conv3x3_c3_p_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                         float* __restrict__ y, int H, int W, float* __restrict__ cmax_out, int tiles_x, int ntiles)
{
    constexpr int COUT = 64, SEG = 64, ROWS = 2, LW = 68;      // LW: floats per staged row (66 used; 272 B keeps the 16-byte reads aligned)
    constexpr int NEL = 3 * (ROWS + 2) * (SEG + 2);             // 792 input values per tile
    __shared__ __attribute__((aligned(16))) float in_s[2][3][ROWS + 2][LW];
    const int tid = threadIdx.x, q = tid & 15, g = tid >> 4;
    // this thread's (up to four) elements of a tile's input patch, once: the offset from the tile's origin in x, and (row, column, LDS
    // offset) packed into one register (bits 0-6 column, 8-10 row, 12-.. LDS offset; -1: no element)
    int e_off[4], e_rc[4];
    const int HW = H * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = tid + 256 * k;
        const int ci = e / ((ROWS + 2) * (SEG + 2)), rem = e - ci * (ROWS + 2) * (SEG + 2);
        const int r = rem / (SEG + 2), c = rem - r * (SEG + 2);
        e_off[k] = ci * HW + (r - 1) * W + c - 1;
        e_rc[k] = e < NEL ? (c | (r << 8) | (((ci * (ROWS + 2) + r) * LW + c) << 12)) : -1;
    }
    auto fetch = [&](int tile, float (&v)[4]) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * ROWS, x0 = tx * SEG;
        const int base = y0 * W + x0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gy = y0 + ((e_rc[k] >> 8) & 7) - 1, gx = x0 + (e_rc[k] & 127) - 1;
            const bool inb = e_rc[k] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
            v[k] = inb ? x[base + e_off[k]] : 0.f;
        }
    };
    auto stage = [&](int buf, const float (&v)[4]) {
        float* dst = &in_s[buf][0][0][0];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e_rc[k] >= 0) dst[e_rc[k] >> 12] = v[k];
    };
    f32x4 wq[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wq[k] = *reinterpret_cast<const f32x4*>(wp + k * COUT + 4 * q);
    const f32x4 bq = *reinterpret_cast<const f32x4*>(bias + 4 * q);
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    {
        float v[4];
        fetch(tile, v);
        stage(0, v);
    }
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        float nv_[4];
        if (next < ntiles) fetch(next, nv_);                    // (in flight under the tile's arithmetic)
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y0 = ty * ROWS, x0 = tx * SEG;
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
            const int yy = y0 + row;
            if (yy >= H) break;
            // One v_pk_fma_f32 per (pixel, channel pair, tap), the input value broadcast from its half of a register pair by op_sel.  Inline
            // assembly: left to the compiler the nest was packed across pixels and spilled (SLP), or -- written with two-float vectors --
            // got a v_mov per odd-positioned input value to build (v, v) pairs.
            c3_f32x2 acc[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = c3_f32x2{0.f, 0.f};
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float* src = &in_s[buf][ci][row + r][4 * g];
                    const f32x4 a = *reinterpret_cast<const f32x4*>(src);
                    c3_f32x2 pr[3];
                    pr[0] = c3_f32x2{a[0], a[1]}; pr[1] = c3_f32x2{a[2], a[3]};
                    pr[2] = *reinterpret_cast<const c3_f32x2*>(src + 4);
#pragma unroll
                    for (int s_ = 0; s_ < 3; ++s_) {
                        const f32x4 w4 = wq[ci * 9 + r * 3 + s_];
                        const c3_f32x2 wlo = {w4[0], w4[1]}, whi = {w4[2], w4[3]};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = i + s_;
                            if ((k & 1) == 0) {
                                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i][0]) : "v"(pr[k >> 1]), "v"(wlo));
                                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i][1]) : "v"(pr[k >> 1]), "v"(whi));
                            } else {
                                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[i][0]) : "v"(pr[k >> 1]), "v"(wlo));
                                asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc[i][1]) : "v"(pr[k >> 1]), "v"(whi));
                            }
                        }
                    }
                }
            unsigned pm[4];                                        // the four pixels' channel maxima (after the row reduction: in every lane of the 16)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int xx = x0 + 4 * g + i;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float t = acc[i][j >> 1][j & 1] + bq[j];
                    o[j] = RELU ? fmaxf(t, 0.f) : t;
                }
                if (xx < W) *reinterpret_cast<f32x4*>(y + ((size_t)yy * W + xx) * COUT + 4 * q) = o;
                if (CMAX) {
                    // (post-ReLU values are >= 0; without ReLU the consumer wants the maximum MAGNITUDE)
                    const float mf = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
                    // (magnitudes: non-negative floats order like their bit patterns -- the row reduction as unsigned maxima, one
                    //  instruction per rotation instead of the four a float maximum with its NaN canonicalisation takes)
                    pm[i] = __builtin_bit_cast(unsigned, mf);
                }
            }
            if (CMAX) {
                // the four pixels' rotations side by side: a DPP instruction may not read its predecessor's result without wait states
#pragma unroll
                for (int i = 0; i < 4; ++i) pm[i] = max(pm[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm[i], 0x128, 0xf, 0xf, false));   // row_ror:8
#pragma unroll
                for (int i = 0; i < 4; ++i) pm[i] = max(pm[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm[i], 0x124, 0xf, 0xf, false));   // row_ror:4
#pragma unroll
                for (int i = 0; i < 4; ++i) pm[i] = max(pm[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm[i], 0x122, 0xf, 0xf, false));   // row_ror:2
#pragma unroll
                for (int i = 0; i < 4; ++i) pm[i] = max(pm[i], (unsigned)__builtin_amdgcn_update_dpp(0, (int)pm[i], 0x121, 0xf, 0xf, false));   // row_ror:1
                // lanes q = 0 .. 3 of the 16 store pixels 0 .. 3: ONE 16-byte run per thread group and row
                const unsigned mine = q == 0 ? pm[0] : q == 1 ? pm[1] : q == 2 ? pm[2] : pm[3];
                const int xq = x0 + 4 * g + q;
                if (q < 4 && xq < W) cmax_out[(size_t)yy * W + xq] = __builtin_bit_cast(float, mine);
            }
        }
        if (next < ntiles) stage(buf ^ 1, nv_);
        __syncthreads();
        buf ^= 1;
    }
}

__global__ __launch_bounds__(256)
void maxpool2x2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C)
{
    const int Hp = H >> 1, Wp = W >> 1, C4 = C >> 2;
    const size_t total = (size_t)Hp * Wp * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int px = (int)(p % Wp), py = (int)(p / Wp);
        const f32x4* r0 = reinterpret_cast<const f32x4*>(x + ((size_t)(2 * py) * W + 2 * px) * C) + c4;
        const f32x4* r1 = reinterpret_cast<const f32x4*>(x + ((size_t)(2 * py + 1) * W + 2 * px) * C) + c4;
        const f32x4 a = r0[0], b = r0[C4], c = r1[0], d = r1[C4];
        f32x4 m;
#pragma unroll
        for (int j = 0; j < 4; ++j) m[j] = fmaxf(fmaxf(a[j], b[j]), fmaxf(c[j], d[j]));
        reinterpret_cast<f32x4*>(y)[i] = m;
    }
}

// ---- weight repacking ----------------------------------------------------------------------
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout, int cin)
{
    const size_t total = (size_t)9 * cout * cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % cin);
        const size_t t = i / cin;
        const int o = (int)(t % cout);
        const int tap = (int)(t / cout);
        wp[i] = w[((size_t)o * cin + ci) * 9 + tap];
    }
}

__global__ void pack_conv3x3_c3_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout)
{
    const int total = 27 * cout;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int o = i % cout, k = i / cout;     // k = ci*9 + r*3 + s
        wp[i] = w[o * 27 + k];
    }
}

template <int WM, int WN, bool POOL>
static int launch_cfg(const float* x, const float* wp, const float* b, float* y, int H, int W,
                      int cin, int cout, int relu, int ksplit, float* ws, hipStream_t s)
{
    using C = ConvCfg<WM, WN>;
    const int cout_tiles = cout / C::BN;
    const int nchunks = cin / 16;
    dim3 grid(cdiv(W, 32), cdiv(H, C::TR), cout_tiles * ksplit);
    auto kern = conv3x3_mfma_kernel<WM, WN, POOL>;
    FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, x, wp, b, y, H, W, cin, cout, relu, cout_tiles,
                       cdiv(nchunks, ksplit), ksplit > 1 ? ws : (float*)nullptr);
    return check_launch();
}

// Split-K factor.  Two reasons to split: (1) a layer whose (rows x 32-col segments x cout tiles)
// grid cannot fill the chip -- the 37x62 maps of block 5 / the RPN trunk give 80 blocks for 256
// CUs; (2) block-count quantisation -- the dispatcher packs the tail of a grid 3 blocks per CU
// onto a subset of CUs (measured: 1200 blocks on 768 slots run as two full rounds = 78 % MFMA
// busy), so work units must be small against blocks/slots.  Power of two, at least two 16-channel
// chunks per split; the target of ~5 blocks per CU was the measured optimum (profiles/r01).
// That optimum is for ONE image on the chip (latency).  With many images in flight on separate streams the
// other images' kernels fill the tail, and fewer, longer work units win (measured, 24 in flight: target 320 ->
// 293 img/s, 1280 -> 286, 2560 -> 281): frcnn_forward_params.conv_blocks_target lets the caller say which regime
// it is in; 0 = this default.  FRCNN_CONV_BLOCKS_TARGET overrides both (tuning knob).
static thread_local int g_blocks_target_override = 0;
void conv3x3_set_blocks_target(int target) { g_blocks_target_override = target > 0 ? target : 0; }

static int conv_blocks_target()
{
    static int env_target = -1;
    if (env_target < 0) {
        const char* e = frcnn_knob("FRCNN_CONV_BLOCKS_TARGET");
        env_target = e ? atoi(e) : 0;
        if (env_target < 0) env_target = 0;
    }
    if (env_target > 0) return env_target;
    return g_blocks_target_override > 0 ? g_blocks_target_override : 1280;
}

static int choose_ksplit(int H, int W, int cin, int cout)
{
    const bool narrow = (cout % 128) != 0;
    const int tr = narrow ? 8 : 4, bn = narrow ? 64 : 128;
    const int blocks = cdiv(W, 32) * cdiv(H, tr) * (cout / bn);
    const int nchunks = cin / 16;
    const int target = conv_blocks_target();
    if (blocks * 2 > target) return 1;
    // partial sums cost 2 x ksplit x output bytes of HBM traffic: not worth it on the big early maps
    if ((size_t)H * W * cout * sizeof(float) > ((size_t)40 << 20)) return 1;
    int k = 1;
    while (k * 2 * blocks <= target && nchunks % (k * 2) == 0 && nchunks / (k * 2) >= 4) k *= 2;
    return k;
}

int conv3x3_choose_ksplit(int H, int W, int cin, int cout) { return choose_ksplit(H, W, cin, cout); }
int conv3x3_blocks_target() { return conv_blocks_target(); }

int launch_conv_splitk_finish(const float* ws, int ksplit, const float* b, float* y, int H, int W, int cout, int relu,
                              int pool, hipStream_t s)
{
    const size_t total = (size_t)(pool ? H / 2 : H) * (pool ? W / 2 : W) * (cout / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, ws, ksplit, b, y, H, W, cout, relu, pool);
    return check_launch();
}

size_t conv3x3_workspace_bytes(int H, int W, int cin, int cout)
{
    if (cin % 16 != 0 || cout % 64 != 0 || H < 1 || W < 1) return 0;
    const int k = choose_ksplit(H, W, cin, cout);
    return k > 1 ? (size_t)k * H * W * cout * sizeof(float) : 0;
}

int launch_conv3x3_nhwc(const float* x, const float* wp, const float* b, float* y, int H, int W,
                        int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (cin % 16 != 0 || cout % 64 != 0 || H < 1 || W < 1) return FRCNN_EINVAL;
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    const bool pool = (flags & FRCNN_POOL2) != 0;
    if (pool && (H < 2 || W < 2)) return FRCNN_EINVAL;
    int ksplit = choose_ksplit(H, W, cin, cout);
    if (ksplit > 1 && (ws == nullptr || ws_bytes < (size_t)ksplit * H * W * cout * sizeof(float))) ksplit = 1;
    float* wsf = static_cast<float*>(ws);
    int rc;
    // Tile choice: cout = 64 -> 8 rows x 32 cols x 64 ch; otherwise 4 rows x 32 cols x 128 ch.
    if (ksplit > 1) {
        rc = (cout % 128 != 0) ? launch_cfg<4, 1, false>(x, wp, b, y, H, W, cin, cout, relu, ksplit, wsf, s)
                               : launch_cfg<2, 2, false>(x, wp, b, y, H, W, cin, cout, relu, ksplit, wsf, s);
        if (rc) return rc;
        return launch_conv_splitk_finish(wsf, ksplit, b, y, H, W, cout, relu, pool ? 1 : 0, s);
    }
    if (cout % 128 != 0) {
        return pool ? launch_cfg<4, 1, true>(x, wp, b, y, H, W, cin, cout, relu, 1, nullptr, s)
                    : launch_cfg<4, 1, false>(x, wp, b, y, H, W, cin, cout, relu, 1, nullptr, s);
    }
    return pool ? launch_cfg<2, 2, true>(x, wp, b, y, H, W, cin, cout, relu, 1, nullptr, s)
                : launch_cfg<2, 2, false>(x, wp, b, y, H, W, cin, cout, relu, 1, nullptr, s);
}

int launch_conv3x3_c3(const float* x, const float* wp, const float* b, float* y, int H, int W,
                      int cout, unsigned flags, hipStream_t s, float* cmax_out)
{
    if (cout % 16 != 0 || H < 1 || W < 1) return FRCNN_EINVAL;
    if (cmax_out && cout != 64) return FRCNN_EINVAL;                        // only the 64-channel kernel holds a pixel's channels in one DPP row
    if (cout == 64) {
        const int tiles_x = cdiv(W, 64), tiles_y = cdiv(H, 2);
        if ((long long)tiles_x * tiles_y > 0x7fffffffLL) return FRCNN_EINVAL;
        const int ntiles = tiles_x * tiles_y;
        static const int cus = []() { hipDeviceProp_t p; int d = 0; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) ? p.multiProcessorCount : 256; }();
        static const int bpc = []() { const char* e = frcnn_knob("FRCNN_C3_BPC"); return e && atoi(e) == 3 ? 3 : 2; }();      // blocks per CU (A/B runs)
        const int grid = ntiles < bpc * cus ? ntiles : bpc * cus;             // (launch bounds: the 27 weight quads stay in registers), each block walking its tiles
        const bool relu = (flags & FRCNN_RELU) != 0;
#define C3P_LAUNCH(R_, C_) do { if (bpc == 3) hipLaunchKernelGGL((conv3x3_c3_p_kernel<R_, C_, 3>), dim3(grid), dim3(256), 0, s, x, wp, b, y, H, W, cmax_out, tiles_x, ntiles); \
                                else hipLaunchKernelGGL((conv3x3_c3_p_kernel<R_, C_, 2>), dim3(grid), dim3(256), 0, s, x, wp, b, y, H, W, cmax_out, tiles_x, ntiles); } while (0)
        if (relu) { if (cmax_out) C3P_LAUNCH(true, true); else C3P_LAUNCH(true, false); }
        else { if (cmax_out) C3P_LAUNCH(false, true); else C3P_LAUNCH(false, false); }
#undef C3P_LAUNCH
        return check_launch();
    }
    dim3 grid(cdiv(H * W, 256), 1);
    hipLaunchKernelGGL(conv3x3_c3_kernel, grid, dim3(256), 0, s, x, wp, b, y, H, W, cout,
                       (flags & FRCNN_RELU) ? 1 : 0);
    return check_launch();
}

int launch_maxpool2x2(const float* x, float* y, int H, int W, int c, hipStream_t s)
{
    if (c % 4 != 0 || H < 2 || W < 2) return FRCNN_EINVAL;
    const size_t total = (size_t)(H / 2) * (W / 2) * (c / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxpool2x2_kernel, dim3(blocks), dim3(256), 0, s, x, y, H, W, c);
    return check_launch();
}

int launch_pack_conv3x3(const float* w, float* wp, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)9 * cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(blocks), dim3(256), 0, s, w, wp, cout, cin);
    return check_launch();
}

int launch_pack_conv3x3_c3(const float* w, float* wp, int cout, hipStream_t s)
{
    if (cout < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pack_conv3x3_c3_kernel, dim3(cdiv(27 * cout, 256)), dim3(256), 0, s, w, wp, cout);
    return check_launch();
}

}  // namespace frcnn
