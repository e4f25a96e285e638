// wino_x3f.hip -- EXPERIMENTAL, first version (end of round 3): the Winograd F(2x2,3x3) layer in the f32x3 arithmetic as ONE launch, the
// form the layers conv1_2 ... conv3_3 (pytorch/FasterRCNN/models/vgg16.py:77-87: 3x3 convolution + ReLU, MaxPool2d after each block) need (their V + M scratch is 600 MB per layer in the three-launch form of csrc/wino_x3.hip:
// DESIGN.md 7.1).  Validated on the MI355X bit for bit against the three-launch layer (tests/test_gemm_x3t_gpu.py); NOT yet tuned and
// not used by any forward: 0.5-0.8x the speed of the float32 one-launch kernel on the six VGG-16 layers (tools/x3f_bench.py).
//
// By construction the result equals launch_conv3x3_winograd_x3's BIT FOR BIT (same per-tile scale, same fp16 split, the same sequence of
// float32 accumulations per (position, tile, output channel) -- hi*lo, hi*hi, lo*hi per 16-channel chunk, chunks in order -- and
// wino_output_kernel's operation order in the epilogue), which is how it is meant to be validated.
//
// Block = 4 waves = 32 tiles (2 tile rows x 16) x 64 output channels x all 16 positions.  Wave w owns position row i = w (positions
// 4 w .. 4 w + 3) for both 32-channel output tiles: 8 accumulator tiles of 32 x 32.  Per 16-channel chunk:
//   * the (4 + 2) x (32 + 2)-pixel input halo of the chunk sits in LDS (float32, 20-float pixel stride);
//   * a lane (tile l & 31, channels 8 (l >> 5) .. + 7) reads the two patch rows B^T combines for its wave's position row, forms
//     V[4 w + j] = (B^T d B)[w][j] in csrc/winograd.hip's float32 operation order, multiplies by the tile's 2^e and splits into the two
//     fp16 MFMA operand fragments -- in registers, no V in LDS;
//   * the filter bank's records of the chunk (16 positions x 2 row blocks x 2 terms x 1 KB = 64 KB: the x3t blob of
//     launch_pack_conv3x3_winograd_x3) arrive by LDS-DMA, double buffered;
//   * 24 MFMAs per wave (4 positions x 2 output tiles x 3 products).
// Epilogue: every wave scales its accumulators by 2^-e(tile) 2^-e(position, channel) and writes them to LDS ([16][32][64] float32 =
// 128 KB = the two filter buffers); a thread then owns (tile, 4 channels) x 2 and applies A^T M A + bias + ReLU (+ 2x2 max-pool) exactly as
// wino_output_kernel does.
#include "x3t.h"

namespace frcnn {

typedef _Float16 xf_f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* xf_lds_ptr;

static constexpr int XF_TR = 2, XF_TC = 16;                  // tile rows / columns of a block
static constexpr int XF_HR = 2 * XF_TR + 2, XF_HC = 2 * XF_TC + 2;   // halo rows / columns: 6 x 34 pixels
static constexpr int XF_PS = 20;                              // floats per halo pixel (16 + 4 padding)
static constexpr int XF_HALO_BYTES = XF_HR * XF_HC * XF_PS * 4;      // 16,320
static constexpr int XF_U_BYTES = 16 * 2 * 2 * HX_PIECE;     // one chunk of the bank for 64 output channels: 65,536
static constexpr size_t XF_LDS_BYTES = (size_t)XF_HALO_BYTES + 2 * XF_U_BYTES;     // 147,392 (version 1)
static constexpr size_t XF_LDS_BYTES2 = 2 * (size_t)XF_HALO_BYTES + 2 * XF_U_BYTES;   // 163,712 (version 2: two halo buffers) <= 163,840

struct XfGeom { int tbx, tby, ncb, tw, th; };

// VER 1: the first version.  VER 2 (FRCNN_X3F_VER=2; passes the same bit-for-bit test): V of chunk c+1 is formed under the MFMAs of chunk c
// (two halo buffers), the filter DMA runs two chunks ahead, ONE barrier per chunk, and the three products of a chunk are issued
// term-major so that consecutive MFMAs hit different accumulators (the per-accumulator order, hence every bit, is unchanged).  Measured:
// no faster than VER 1 (conv3_2 232 vs 213 us) -- the kernel is bound by the L2 -> LDS staging of the filter records (64 KB per chunk and
// block for 96 MFMAs), not by barriers or vector work: DESIGN.md 7.1.  The next version needs 64 tiles per block.
template <bool POOL, int VER>
__global__ __launch_bounds__(256, 1)
void wino_x3f_kernel(const float* __restrict__ x_maps, const float* __restrict__ cmax_maps, const unsigned char* __restrict__ ublob,
                     const float* __restrict__ bias, float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int u_rbt, int relu,
                     XfGeom gm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_xf[];
    float* const halo = reinterpret_cast<float*>(smem_xf);
    unsigned char* const ubuf = smem_xf + (VER == 1 ? 1 : 2) * XF_HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // position row i
    const int K16 = Cin >> 4;

    int b = blockIdx.x;
    const int cb = b % gm.ncb;
    b /= gm.ncb;
    const int bx = b % gm.tbx;
    b /= gm.tbx;
    const int by = b % gm.tby;
    const int map = b / gm.tby;
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    const float* __restrict__ const cmax = cmax_maps + (size_t)map * H * W;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout;

    // ---- this lane's tile and its scale (the same bound as wino_input_x3t_kernel: 4 x the largest channel maximum of the 4 x 4 patch) ----
    const int tl = lane & 31, tyl = tl >> 4, txl = tl & 15, kh = lane >> 5;
    const int ty = XF_TR * by + tyl, tx = XF_TC * bx + txl;
    float mult, vinv;
    {
        float dmax = 0.f;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = y0 + a, xx = x0 + c;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) dmax = fmaxf(dmax, cmax[(size_t)yy * W + xx]);
            }
        hx_row_scale(4.0f * dmax, mult, vinv);
    }

    // ---- halo staging: 204 pixels x 4 quads of 16 B = 816 pieces, 4 per thread (the last ones idle) ------------------------------------
    const int hy0 = 2 * XF_TR * by - 1, hx0 = 2 * XF_TC * bx - 1;
    int h_src[4], h_dst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int q = tid + 256 * it;
        const int px = q >> 2, quad = q & 3;
        const int hr = px / XF_HC, hc = px - hr * XF_HC;
        const int gy = hy0 + hr, gx = hx0 + hc;
        const bool live = q < XF_HR * XF_HC * 4;
        const bool inb = live && gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_src[it] = inb ? (gy * W + gx) * Cin + 4 * quad : -1;         // float offset (H * W * Cin < 2^31 / 4: checked by the launcher)
        h_dst[it] = live ? (hr * XF_HC + hc) * XF_PS + 4 * quad : -1;
    }
    f32x4 hreg[4];
    auto load_halo = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < 4; ++it)
            hreg[it] = h_src[it] >= 0 ? *reinterpret_cast<const f32x4*>(x + h_src[it] + 16 * chunk) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&](int hb = 0) {
        float* hbuf = halo + hb * (XF_HALO_BYTES / 4);
#pragma unroll
        for (int it = 0; it < 4; ++it)
            if (h_dst[it] >= 0) *reinterpret_cast<f32x4*>(hbuf + h_dst[it]) = hreg[it];
    };
    // ---- filter records: piece (position p, row block r, term t) of chunk c = ublob + ((p K16 + c) u_rbt + 2 cb + r) 2 KB + t 1 KB ------
    auto issue_u = [&](int chunk, int buf) {
        unsigned char* dst = ubuf + buf * XF_U_BYTES;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int q = wave + 4 * it;                                    // 64 pieces, 16 per wave
            const int p = q >> 2, r = (q >> 1) & 1, t = q & 1;
            const unsigned char* src = ublob + (((size_t)p * K16 + chunk) * u_rbt + 2 * cb + r) * HX_RB + t * HX_PIECE + lane * 16;
            __builtin_amdgcn_global_load_lds(src, (xf_lds_ptr)(dst + q * HX_PIECE), 16, 0, 0);
        }
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;

    // patch rows that B^T combines for position row i = wave: (a1, a2, subtract)
    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int a2 = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
    const bool rsub = wave != 1;                                            // i = 1: d1 + d2; i = 0, 2, 3: differences
    const int d_off = ((2 * tyl) * XF_HC + 2 * txl) * XF_PS + 8 * kh;       // + (a XF_HC + b) XF_PS

    // V of this wave's four positions for its lane's tile and 8 channels, from halo buffer `hb`, as two fp16 fragments per position
    auto form_v = [&](int hb, xf_f16x8 (&vh)[4], xf_f16x8 (&vl)[4]) {
        const float* hbuf = halo + hb * (XF_HALO_BYTES / 4);
        float r[4][8];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float* p1 = hbuf + d_off + (a1 * XF_HC + bb) * XF_PS;
            const float* p2 = hbuf + d_off + (a2 * XF_HC + bb) * XF_PS;
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(p1), u1 = *reinterpret_cast<const f32x4*>(p1 + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p2), w1 = *reinterpret_cast<const f32x4*>(p2 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[bb][e] = rsub ? u0[e] - w0[e] : u0[e] + w0[e];
                r[bb][4 + e] = rsub ? u1[e] - w1[e] : u1[e] + w1[e];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = j == 0 ? r[0][e] - r[2][e] : j == 1 ? r[1][e] + r[2][e] : j == 2 ? r[2][e] - r[1][e] : r[1][e] - r[3][e];
                v[e] = t * mult;
            }
            uint4 ph, pl;
            hx_split8(v, ph, pl);
            vh[j] = __builtin_bit_cast(xf_f16x8, ph);
            vl[j] = __builtin_bit_cast(xf_f16x8, pl);
        }
    };

    if (VER == 1) {
        load_halo(0);
        issue_u(0, 0);
        store_halo();
        __syncthreads();                                                     // (the compiler's fence waits for the DMA: vmcnt(0))
        for (int c = 0; c < K16; ++c) {
            xf_f16x8 vh[4], vl[4];
            form_v(0, vh, vl);
            __syncthreads();                                                 // everybody has read halo(c)
            const bool more = c + 1 < K16;
            if (more) { load_halo(c + 1); issue_u(c + 1, (c + 1) & 1); }
            // ---- 24 MFMAs: per (position, output tile) hi*lo, hi*hi, lo*hi -- the accumulation order of gemm_x3t_kernel --------------------
            const unsigned char* ub = ubuf + (c & 1) * XF_U_BYTES + lane * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * wave + j;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const xf_f16x8 uh = *reinterpret_cast<const xf_f16x8*>(ub + ((p * 2 + ct) * 2 + 0) * HX_PIECE);
                    const xf_f16x8 ul = *reinterpret_cast<const xf_f16x8*>(ub + ((p * 2 + ct) * 2 + 1) * HX_PIECE);
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh[j], acc[j][ct], 0, 0, 0);      // filter lo x V hi
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh[j], acc[j][ct], 0, 0, 0);      // filter hi x V hi
                    acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl[j], acc[j][ct], 0, 0, 0);      // filter hi x V lo
                }
            }
            if (more) store_halo();
            __syncthreads();                                                 // halo(c+1) and U(c+1) have landed; U(c) is no longer read
        }
    } else {
        // ---- version 2: V(c+1) under the MFMAs of chunk c; halo(c) lives in halo buffer c & 1, U(c) in filter buffer c & 1 ------------------
        xf_f16x8 vh[4], vl[4], nh[4], nl[4];
        load_halo(0);
        issue_u(0, 0);
        store_halo(0);
        __syncthreads();
        form_v(0, vh, vl);
        if (K16 > 1) { load_halo(1); issue_u(1, 1); store_halo(1); }
        __syncthreads();                                                     // halo(1), U(0), U(1) have landed
        for (int c = 0; c < K16; ++c) {
            const bool more1 = c + 1 < K16, more2 = c + 2 < K16;
            if (more2) load_halo(c + 2);                                     // registers; written to halo buffer c & 1 below (halo(c) is spent)
            const unsigned char* ub = ubuf + (c & 1) * XF_U_BYTES + lane * 16;
            xf_f16x8 uh[4][2], ul[4][2];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    uh[j][ct] = *reinterpret_cast<const xf_f16x8*>(ub + (((4 * wave + j) * 2 + ct) * 2 + 0) * HX_PIECE);
                    ul[j][ct] = *reinterpret_cast<const xf_f16x8*>(ub + (((4 * wave + j) * 2 + ct) * 2 + 1) * HX_PIECE);
                }
            // term-major: eight independent accumulators per term; per accumulator still lo*hi, hi*hi, hi*lo in this order
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul[j][ct], vh[j], acc[j][ct], 0, 0, 0);
            if (more1) form_v((c + 1) & 1, nh, nl);                          // vector work for the next chunk while the matrix pipe runs
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[j][ct], vh[j], acc[j][ct], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh[j][ct], vl[j], acc[j][ct], 0, 0, 0);
            if (more2) store_halo(c & 1);
            __syncthreads();                                                 // U(c) and halo(c+1) are spent, halo(c+2) is visible, U(c+1) has landed
            if (more2) issue_u(c + 2, c & 1);
            if (more1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { vh[j] = nh[j]; vl[j] = nl[j]; }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: scaled M to LDS [16 positions][32 tiles][64 channels], then wino_output_kernel's arithmetic ----------------------------
    float* const mbuf = reinterpret_cast<float*>(ubuf);
    {
        const int Np = u_rbt * 32;
        const float* uinv = reinterpret_cast<const float*>(ublob + (size_t)16 * K16 * u_rbt * HX_RB);       // [16][Np]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = 4 * wave + j;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * ct + 8 * g + 4 * kh;                 // the MFMA's row operand was the filter: rows = channels
                    const f32x4 sb = *reinterpret_cast<const f32x4*>(uinv + (size_t)p * Np + 64 * cb + co);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[j][ct][4 * g + e] * vinv) * sb[e];
                    *reinterpret_cast<f32x4*>(mbuf + ((size_t)p * 32 + tl) * 64 + co) = v;
                }
        }
    }
    __syncthreads();
    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + 256 * it;                                     // (tile, channel quad): 32 x 16
        const int t = item >> 4, k = (item & 15) * 4;
        const int oty = XF_TR * by + (t >> 4), otx = XF_TC * bx + (t & 15);
        if (oty >= gm.th || otx >= gm.tw) continue;
        if (POOL && (oty >= Ho || otx >= Wo)) continue;
        const int kg = 64 * cb + k;
        if (kg >= Cout) continue;
        const float* mp = mbuf + (size_t)t * 64 + k;
        f32x4 s[2][4];                                                      // A^T M
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(mp + (size_t)(0 + j) * 2048);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(mp + (size_t)(4 + j) * 2048);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(mp + (size_t)(8 + j) * 2048);
            const f32x4 m3 = *reinterpret_cast<const f32x4*>(mp + (size_t)(12 + j) * 2048);
            s[0][j] = (m0 + m1) + m2;
            s[1][j] = (m1 - m2) - m3;
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + kg);
        f32x4 o[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {                                       // (A^T M) A
            o[a][0] = ((s[a][0] + s[a][1]) + s[a][2]) + bv;
            o[a][1] = ((s[a][1] - s[a][2]) - s[a][3]) + bv;
        }
        if (relu) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[a][bb][e] = fmaxf(o[a][bb][e], 0.f);
        }
        if (POOL) {
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
            *reinterpret_cast<f32x4*>(y + ((size_t)oty * Wo + otx) * Cout + kg) = m;
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int yy = 2 * oty + a;
                if (yy >= H) continue;
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const int xx = 2 * otx + bb;
                    if (xx < W) *reinterpret_cast<f32x4*>(y + ((size_t)yy * W + xx) * Cout + kg) = o[a][bb];
                }
            }
        }
    }
}

// ---- VER 3: 64 tiles per block (4 tile rows x 16; FRCNN_X3F_VER=3) -----------------------------------------------------------------------
// The same arithmetic with twice the tiles per staged filter chunk (75 instead of 43 FLOP per staged byte).  A lane owns the tiles
// (tile row 2 h + ((l & 31) >> 4), column l & 15) for h = 0, 1: 16 accumulator tiles per wave = 256 accumulator registers.  The loop
// has VER 1's structure; the epilogue goes through LDS one half (32 tiles) at a time.  Bit-identical too; measured 3 % faster than VER 1
// (conv3_2 208 us): with VER 1's loop the filter DMA of a chunk is exposed.  Next: this tile with VER 2's DMA distance.
static constexpr int X3_HR = 10;                                              // halo rows: 4 tile rows x 2 + 2
static constexpr int X3_HALO_BYTES = X3_HR * XF_HC * XF_PS * 4;               // 27,200
static constexpr int X3_NPC = (X3_HR * XF_HC * 4 + 255) / 256;                // halo pieces per thread: 6
static constexpr size_t XF_LDS_BYTES3 = (size_t)X3_HALO_BYTES + 2 * XF_U_BYTES;   // 158,272

template <bool POOL>
__global__ __launch_bounds__(256, 1)
void wino_x3f64_kernel(const float* __restrict__ x_maps, const float* __restrict__ cmax_maps, const unsigned char* __restrict__ ublob,
                       const float* __restrict__ bias, float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int u_rbt, int relu,
                       XfGeom gm)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_xf[];
    float* const halo = reinterpret_cast<float*>(smem_xf);
    unsigned char* const ubuf = smem_xf + X3_HALO_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K16 = Cin >> 4;

    int b = blockIdx.x;
    const int cb = b % gm.ncb;
    b /= gm.ncb;
    const int bx = b % gm.tbx;
    b /= gm.tbx;
    const int by = b % gm.tby;                                               // gm.tby counts blocks of FOUR tile rows here
    const int map = b / gm.tby;
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    const float* __restrict__ const cmax = cmax_maps + (size_t)map * H * W;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout;

    const int tl = lane & 31, tyl = tl >> 4, txl = tl & 15, kh = lane >> 5;
    const int tx = XF_TC * bx + txl;
    float mult[2], vinv[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int ty = 4 * by + 2 * h + tyl;
        float dmax = 0.f;
        const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int yy = y0 + a, xx = x0 + c;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) dmax = fmaxf(dmax, cmax[(size_t)yy * W + xx]);
            }
        hx_row_scale(4.0f * dmax, mult[h], vinv[h]);
    }

    const int hy0 = 8 * by - 1, hx0 = 2 * XF_TC * bx - 1;
    int h_src[X3_NPC], h_dst[X3_NPC];
#pragma unroll
    for (int it = 0; it < X3_NPC; ++it) {
        const int q = tid + 256 * it;
        const int px = q >> 2, quad = q & 3;
        const int hr = px / XF_HC, hc = px - hr * XF_HC;
        const int gy = hy0 + hr, gx = hx0 + hc;
        const bool live = q < X3_HR * XF_HC * 4;
        const bool inb = live && gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_src[it] = inb ? (gy * W + gx) * Cin + 4 * quad : -1;
        h_dst[it] = live ? (hr * XF_HC + hc) * XF_PS + 4 * quad : -1;
    }
    f32x4 hreg[X3_NPC];
    auto load_halo = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < X3_NPC; ++it)
            hreg[it] = h_src[it] >= 0 ? *reinterpret_cast<const f32x4*>(x + h_src[it] + 16 * chunk) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int it = 0; it < X3_NPC; ++it)
            if (h_dst[it] >= 0) *reinterpret_cast<f32x4*>(halo + h_dst[it]) = hreg[it];
    };
    auto issue_u = [&](int chunk, int buf) {
        unsigned char* dst = ubuf + buf * XF_U_BYTES;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int q = wave + 4 * it;
            const int p = q >> 2, r = (q >> 1) & 1, t = q & 1;
            const unsigned char* src = ublob + (((size_t)p * K16 + chunk) * u_rbt + 2 * cb + r) * HX_RB + t * HX_PIECE + lane * 16;
            __builtin_amdgcn_global_load_lds(src, (xf_lds_ptr)(dst + q * HX_PIECE), 16, 0, 0);
        }
    };

    f32x16 acc[2][4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][j][ct][r] = 0.f;

    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int a2 = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
    const bool rsub = wave != 1;

    auto form_v = [&](int h, xf_f16x8 (&vh)[4], xf_f16x8 (&vl)[4]) {
        const int d_off = ((2 * (2 * h + tyl)) * XF_HC + 2 * txl) * XF_PS + 8 * kh;
        float r[4][8];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            const float* p1 = halo + d_off + (a1 * XF_HC + bb) * XF_PS;
            const float* p2 = halo + d_off + (a2 * XF_HC + bb) * XF_PS;
            const f32x4 u0 = *reinterpret_cast<const f32x4*>(p1), u1 = *reinterpret_cast<const f32x4*>(p1 + 4);
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(p2), w1 = *reinterpret_cast<const f32x4*>(p2 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                r[bb][e] = rsub ? u0[e] - w0[e] : u0[e] + w0[e];
                r[bb][4 + e] = rsub ? u1[e] - w1[e] : u1[e] + w1[e];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float t = j == 0 ? r[0][e] - r[2][e] : j == 1 ? r[1][e] + r[2][e] : j == 2 ? r[2][e] - r[1][e] : r[1][e] - r[3][e];
                v[e] = t * mult[h];
            }
            uint4 ph, pl;
            hx_split8(v, ph, pl);
            vh[j] = __builtin_bit_cast(xf_f16x8, ph);
            vl[j] = __builtin_bit_cast(xf_f16x8, pl);
        }
    };

    load_halo(0);
    issue_u(0, 0);
    store_halo();
    __syncthreads();
    for (int c = 0; c < K16; ++c) {
        xf_f16x8 vh[2][4], vl[2][4];
        form_v(0, vh[0], vl[0]);
        form_v(1, vh[1], vl[1]);
        __syncthreads();                                                     // everybody has read halo(c)
        const bool more = c + 1 < K16;
        if (more) { load_halo(c + 1); issue_u(c + 1, (c + 1) & 1); }
        const unsigned char* ub = ubuf + (c & 1) * XF_U_BYTES + lane * 16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = 4 * wave + j;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const xf_f16x8 uh = *reinterpret_cast<const xf_f16x8*>(ub + ((p * 2 + ct) * 2 + 0) * HX_PIECE);
                const xf_f16x8 ul = *reinterpret_cast<const xf_f16x8*>(ub + ((p * 2 + ct) * 2 + 1) * HX_PIECE);
                // per accumulator: filter lo x V hi, filter hi x V hi, filter hi x V lo; the two halves alternate (independent accumulators)
                acc[0][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh[0][j], acc[0][j][ct], 0, 0, 0);
                acc[1][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ul, vh[1][j], acc[1][j][ct], 0, 0, 0);
                acc[0][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh[0][j], acc[0][j][ct], 0, 0, 0);
                acc[1][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vh[1][j], acc[1][j][ct], 0, 0, 0);
                acc[0][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl[0][j], acc[0][j][ct], 0, 0, 0);
                acc[1][j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(uh, vl[1][j], acc[1][j][ct], 0, 0, 0);
            }
        }
        if (more) store_halo();
        __syncthreads();
    }

    float* const mbuf = reinterpret_cast<float*>(ubuf);
    const int Np = u_rbt * 32;
    const float* uinv = reinterpret_cast<const float*>(ublob + (size_t)16 * K16 * u_rbt * HX_RB);
    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();                                              // the first half's transform has read mbuf
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int p = 4 * wave + j;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int co = 32 * ct + 8 * g + 4 * kh;
                    const f32x4 sb = *reinterpret_cast<const f32x4*>(uinv + (size_t)p * Np + 64 * cb + co);
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[h][j][ct][4 * g + e] * vinv[h]) * sb[e];
                    *reinterpret_cast<f32x4*>(mbuf + ((size_t)p * 32 + tl) * 64 + co) = v;
                }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int item = tid + 256 * it;
            const int t = item >> 4, k = (item & 15) * 4;
            const int oty = 4 * by + 2 * h + (t >> 4), otx = XF_TC * bx + (t & 15);
            if (oty >= gm.th || otx >= gm.tw) continue;
            if (POOL && (oty >= Ho || otx >= Wo)) continue;
            const int kg = 64 * cb + k;
            const float* mp = mbuf + (size_t)t * 64 + k;
            f32x4 s[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 m0 = *reinterpret_cast<const f32x4*>(mp + (size_t)(0 + j) * 2048);
                const f32x4 m1 = *reinterpret_cast<const f32x4*>(mp + (size_t)(4 + j) * 2048);
                const f32x4 m2 = *reinterpret_cast<const f32x4*>(mp + (size_t)(8 + j) * 2048);
                const f32x4 m3 = *reinterpret_cast<const f32x4*>(mp + (size_t)(12 + j) * 2048);
                s[0][j] = (m0 + m1) + m2;
                s[1][j] = (m1 - m2) - m3;
            }
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + kg);
            f32x4 o[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                o[a][0] = ((s[a][0] + s[a][1]) + s[a][2]) + bv;
                o[a][1] = ((s[a][1] - s[a][2]) - s[a][3]) + bv;
            }
            if (relu) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[a][bb][e] = fmaxf(o[a][bb][e], 0.f);
            }
            if (POOL) {
                f32x4 m;
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
                *reinterpret_cast<f32x4*>(y + ((size_t)oty * Wo + otx) * Cout + kg) = m;
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int yy = 2 * oty + a;
                    if (yy >= H) continue;
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int xx = 2 * otx + bb;
                        if (xx < W) *reinterpret_cast<f32x4*>(y + ((size_t)yy * W + xx) * Cout + kg) = o[a][bb];
                    }
                }
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// cmax scratch: n_maps * H * W floats (the channel maxima of the layer input, computed here)
size_t conv3x3_winograd_x3_fused_workspace_bytes(int N, int H, int W) { return (size_t)N * H * W * sizeof(float); }

int launch_conv3x3_winograd_x3_fused(const float* x, const void* ublob, const float* b, float* y, int N, int H, int W, int cin, int cout,
                                     unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || cin < 16 || cin % 16 != 0 || cout < 64 || cout % 64 != 0) return FRCNN_EUNSUPPORTED;
    if ((size_t)H * W * cin >= ((size_t)1 << 29)) return FRCNN_EUNSUPPORTED;          // 32-bit float offsets inside one map
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    if (!ws || ws_bytes < conv3x3_winograd_x3_fused_workspace_bytes(N, H, W)) return FRCNN_EINVAL;
    float* cmax = static_cast<float*>(ws);
    int rc = launch_pixel_absmax(x, cmax, (long long)N * H * W, cin, s);
    if (rc) return rc;
    XfGeom gm;
    gm.tw = cdiv(W, 2); gm.th = cdiv(H, 2);
    static const int ver = []() { const char* e = frcnn_knob("FRCNN_X3F_VER"); return e ? atoi(e) : 1; }();       // 2, 3: the experimental variants
    gm.tbx = cdiv(gm.tw, XF_TC); gm.tby = cdiv(gm.th, ver == 3 ? 4 : XF_TR);
    gm.ncb = cout / 64;
    const long long total = (long long)gm.tbx * gm.tby * gm.ncb * N;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    const int u_rbt = cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout) / 32;
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    const unsigned char* ub = static_cast<const unsigned char*>(ublob);
#define XF_LAUNCH(P, V, LDS)                                                                                        \
    do {                                                                                                           \
        auto kern = wino_x3f_kernel<P, V>;                                                                          \
        FRCNN_MAX_LDS_ONCE(kern, LDS);                                                                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), LDS, s, x, cmax, ub, b, y, H, W, cin, cout, u_rbt, relu, gm);   \
    } while (0)
    const bool pool = (flags & FRCNN_POOL2) != 0;
    if (ver == 3) {
        if (pool) {
            auto kern = wino_x3f64_kernel<true>;
            FRCNN_MAX_LDS_ONCE(kern, XF_LDS_BYTES3);
            hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), XF_LDS_BYTES3, s, x, cmax, ub, b, y, H, W, cin, cout, u_rbt, relu, gm);
        } else {
            auto kern = wino_x3f64_kernel<false>;
            FRCNN_MAX_LDS_ONCE(kern, XF_LDS_BYTES3);
            hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), XF_LDS_BYTES3, s, x, cmax, ub, b, y, H, W, cin, cout, u_rbt, relu, gm);
        }
    } else
    if (ver == 2) { if (pool) XF_LAUNCH(true, 2, XF_LDS_BYTES2); else XF_LAUNCH(false, 2, XF_LDS_BYTES2); }
    else          { if (pool) XF_LAUNCH(true, 1, XF_LDS_BYTES); else XF_LAUNCH(false, 1, XF_LDS_BYTES); }
#undef XF_LAUNCH
    return check_launch();
}

}  // namespace frcnn
