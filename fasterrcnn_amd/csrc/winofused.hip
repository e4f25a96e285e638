// winofused.hip -- 3x3 stride-1 "same" convolution as Winograd F(2x2,3x3) in float32, ONE launch per layer.
// Replaces the three launches of csrc/winograd.hip (input transform -> 16 batched GEMMs -> output transform) for a single
// map: models/vgg16.py:77-96 (conv1_2 ... conv5_3), models/rpn.py:88 (RPN trunk), the stride-1 3x3 convolutions of ResNet's
// layer2 / layer3 (models/resnet.py:38-46 over torchvision's Bottleneck).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// No V / M scratch: HBM traffic of a layer is its input, its filter bank and its output.
//
// Work decomposition.  A block owns 4 x 16 Winograd tiles (8 x 32 output pixels) x 32 output channels and ALL 16 Winograd
// positions of that slab: wave w owns tile row w, its accumulators are acc[position][cout half] = 16 x 2 MFMA 16x16 tiles
// (v_mfma_f32_16x16x4_f32, exact f32, 4 registers each = 128 registers).  Because every position's accumulator is live, the
// output transform A^T M A (+ bias, ReLU, 2x2 max-pool: a Winograd tile IS a pooling window) runs on registers in the epilogue.
//
// K loop over 16-channel chunks; per chunk
//   * the (8+2) x (32+2) pixel input halo of the block's tiles is staged ONCE in LDS (zero padding folded in).  The B^T d B
//     input transform is evaluated when the MFMA operand is formed: for position row i the wave reads the two patch rows
//     that B^T combines (8 ds_read_b128), r_i[b] = d[a1][b] +- d[a2][b], and the four operands of the row are
//     V[i][j] = r_i[b1] +- r_i[b2] -- the same float32 operation order as wino_input_kernel;
//   * the filter bank arrives in four slabs (one per position row i: 4 positions x 32 couts x 16 channels = 8 KB, contiguous
//     in the [chunk][cout block][position][32][16] layout of wino_pack_fused_kernel), double buffered;
//   * MFMA roles: A = U (rows = couts), B = V (columns = tiles), so a lane's four accumulator registers are four CONSECUTIVE
//     output channels of one tile -> 16-byte stores.
// A "stage" is (chunk, i): 32 MFMAs per wave (1024 matrix-pipe cycles), 16 ds_read_b128, 32 VALU adds, 2 global loads +
// 2 LDS writes for the next filter slab, 2 global loads of the next chunk's halo; one barrier per stage, a second one at the
// chunk seam (the halo is single buffered: 32.6 + 2 x 8 KB of LDS per block, two blocks per CU).
//
// LDS layouts (both conflict-free for ds_read_b128 with lane = row + 16 * k-quad, checked against the lane groups of
// MI355X_MICROARCH.md): halo pixel = 16 channels padded to 24 floats, pixels de-interleaved by column parity so that the 16
// tiles of a wave read 16 consecutive pixels; filter rows = 16 floats, the k-quad slot XOR-swizzled by {0,2,3,1}[(row >> 2) & 3].
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace frcnn {

static constexpr int WF_TR = 4, WF_TC = 16;              // tile rows / columns per block (one tile row per wave)
static constexpr int WF_HR = 2 * WF_TR + 2;              // 10 halo rows
static constexpr int WF_HC = 2 * WF_TC + 2;              // 34 halo columns
static constexpr int WF_HP = WF_HC / 2;                  // 17 pixels per parity plane of a halo row
static constexpr int WF_PS = 24;                         // floats per halo pixel in LDS
static constexpr int WF_NPIX = WF_HR * WF_HC;            // 340
static constexpr int WF_HALO_F = WF_NPIX * WF_PS;        // 8160 floats
#ifndef WF_NT
#define WF_NT 2                                          // 16-cout MFMA tiles per wave: 2 = 128 accumulator registers, two blocks per CU
#endif
static constexpr int WF_BN = 16 * WF_NT;                 // output channels per block
static constexpr int WF_U_F = 4 * WF_BN * 16;            // floats per filter slab (4 positions x WF_BN couts x 16 channels)
static constexpr int WF_NU = WF_U_F / 4 / 256;           // 16-byte filter pieces per thread per slab
static constexpr int WF_NHP = WF_NPIX * 4;               // 16-byte halo pieces per chunk
static constexpr int WF_NH = (WF_NHP + 255) / 256;       // 6 per thread
#ifndef WF_ABLATE
#define WF_ABLATE 0                                      // timing experiments only (tools/build_ablate.sh), results wrong.  Inside the K loop:
#endif                                                   // 1 no operand adds, 2 no patch-row reads, 4 no filter fragment reads, 8 no LDS writes,
                                                         // 16 no filter loads, 32 no halo loads, 64 no barrier, 128 half of the operand adds
#ifndef WF_SCALAR_ADDS
#define WF_SCALAR_ADDS 0
#endif
#ifndef WF_PRIO
#define WF_PRIO 1
#endif
#define WF_HALO_BUFS 2                                   // no barrier at the chunk seam; 81,664 B of LDS per block (two blocks fill a CU's 160 KB)
static constexpr size_t WF_LDS_BYTES = (size_t)(WF_HALO_BUFS * WF_HALO_F + 2 * WF_U_F) * sizeof(float);
static_assert(WF_NH == 6, "halo pieces are spread over the first three stages of a chunk, two per stage");

// B^T rows: r_i = d[A1] (-|+) d[A2];  the same table gives the column combination V[i][j] = r_i[A1_j] (-|+) r_i[A2_j]
__device__ __forceinline__ constexpr int wf_a1(int i) { return i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 2 : 1)); }
__device__ __forceinline__ constexpr int wf_a2(int i) { return i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3)); }
__device__ __forceinline__ constexpr bool wf_sub(int i) { return i != 1; }

// U'[chunk][cout block][p = 4 i + j][32][16] = (G g G^T)[i][j] of filter (cout, cin);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
// g: OIHW [cout][cin][3][3]; `scale` (per cout, may be NULL) = the frozen-BatchNorm fold of the ResNet layers, multiplied in
// float32 first exactly as fold_bn_pack_kernel does.  Values are identical to wino_pack_kernel's bank (float64, rounded once).
__global__ __launch_bounds__(256)
void wino_pack_fused_kernel(const float* __restrict__ g, const float* __restrict__ scale, float* __restrict__ u, int cout, int cin)
{
    const size_t total = (size_t)cout * cin;
    const int ncb = cout / WF_BN;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int co = (int)(idx / cin), ci = (int)(idx % cin);
        const float* gp = g + idx * 9;
        const float sc = scale ? scale[co] : 1.0f;
        double w[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) w[a][b] = (double)(scale ? gp[a * 3 + b] * sc : gp[a * 3 + b]);
        double r[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            r[0][b] = w[0][b];
            r[1][b] = 0.5 * (w[0][b] + w[1][b] + w[2][b]);
            r[2][b] = 0.5 * (w[0][b] - w[1][b] + w[2][b]);
            r[3][b] = w[2][b];
        }
        const size_t base = ((size_t)(ci >> 4) * ncb + (co / WF_BN)) * 16 * (WF_BN * 16) + (size_t)(co % WF_BN) * 16 + (ci & 15);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double q[4] = {r[a][0], 0.5 * (r[a][0] + r[a][1] + r[a][2]), 0.5 * (r[a][0] - r[a][1] + r[a][2]), r[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) u[base + (size_t)(4 * a + b) * (WF_BN * 16)] = (float)q[b];
        }
    }
}

// The same bank from the direct kernels' tap-major pack wp[tap][cout][cin] (the train step's master weights):
//   data_gradient == 0: the layer's own filter;  == 1: the data-gradient convolution dz (cout channels) -> dx (cin channels),
//   filter g'[ci][co][tap] = wp[8 - tap][co][ci] (180-degree rotation, channels transposed); output channels = cin then.
__global__ __launch_bounds__(256)
void wino_pack_fused_taps_kernel(const float* __restrict__ wp, float* __restrict__ u, int cout, int cin, int data_gradient)
{
    const size_t total = (size_t)cout * cin;
    const int oc = data_gradient ? cin : cout, ic = data_gradient ? cout : cin;     // channels of the convolution that is packed
    const int ncb = oc / WF_BN;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int o = (int)(idx / ic), c = (int)(idx % ic);
        double w[3][3];
        if (!data_gradient) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) w[tp / 3][tp % 3] = (double)wp[(size_t)tp * total + idx];
        } else {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) w[tp / 3][tp % 3] = (double)wp[(size_t)(8 - tp) * total + (size_t)c * cin + o];
        }
        double r[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            r[0][b] = w[0][b];
            r[1][b] = 0.5 * (w[0][b] + w[1][b] + w[2][b]);
            r[2][b] = 0.5 * (w[0][b] - w[1][b] + w[2][b]);
            r[3][b] = w[2][b];
        }
        const size_t base = ((size_t)(c >> 4) * ncb + (o / WF_BN)) * 16 * (WF_BN * 16) + (size_t)(o % WF_BN) * 16 + (c & 15);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const double q[4] = {r[a][0], 0.5 * (r[a][0] + r[a][1] + r[a][2]), 0.5 * (r[a][0] - r[a][1] + r[a][2]), r[a][2]};
#pragma unroll
            for (int b = 0; b < 4; ++b) u[base + (size_t)(4 * a + b) * (WF_BN * 16)] = (float)q[b];
        }
    }
}

struct WfGeom { int tbx, tby, ncb, total, xcl; };        // xcl: log2 of the number of XCD groups the cout blocks are split over

#define WF_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define WF_MFMA 0x008
#define WF_VALU 0x002
#define WF_DSR  0x100
#define WF_DSW  0x200

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 wf_lo(const f32x4& a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ f32x2 wf_hi(const f32x4& a) { return __builtin_shufflevector(a, a, 2, 3); }
// a -+ b on a register pair in one instruction (the neg modifiers make it a - b: the same IEEE result as v_sub_f32)
__device__ __forceinline__ f32x2 wf_pk(bool sub, f32x2 a, f32x2 b)
{
    f32x2 o;
#if WF_SCALAR_ADDS
    float o0, o1;
    if (sub) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o0) : "v"(a[0]), "v"(b[0])); asm volatile("v_sub_f32 %0, %1, %2" : "=v"(o1) : "v"(a[1]), "v"(b[1])); }
    else { asm volatile("v_add_f32 %0, %1, %2" : "=v"(o0) : "v"(a[0]), "v"(b[0])); asm volatile("v_add_f32 %0, %1, %2" : "=v"(o1) : "v"(a[1]), "v"(b[1])); }
    o[0] = o0; o[1] = o1;
#else
    if (sub) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
#endif
    return o;
}

template <bool POOL>
__global__ __launch_bounds__(256, WF_NT <= 2 ? 2 : 1)
void wino_fused_kernel(const float* __restrict__ x, const float* __restrict__ u, const float* __restrict__ bias,
                       float* __restrict__ y, int H, int W, int Cin, int Cout, int relu, WfGeom gm)
{
#ifdef WF_CLOCKS
    const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
#if WF_PRIO
    __builtin_amdgcn_s_setprio(3);                       // prologue and epilogue ahead of the co-resident block's K loop
#endif
    extern __shared__ __attribute__((aligned(16))) float smem_wf[];
    float* const halo0 = smem_wf;                           // WF_HALO_BUFS halo buffers
    float* const ub0 = smem_wf + WF_HALO_BUFS * WF_HALO_F;  // 2 filter slab buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;

    // XCD-aware block order.  Hardware block b runs on XCD b % 8 and every XCD has its own 4 MB L2.  The 8 XCDs form a
    // (8 >> xcl) x (1 << xcl) grid: XCD (xti, xci) owns the cout blocks of group xci and a contiguous range of that group's
    // (tile block, cout block) pairs, cout block fastest -- the blocks resident on an XCD at one time share halos (same tile
    // block) and filter slabs (same cout block).  xcl = 0: every XCD streams the whole filter bank once per tile-block
    // generation (fine while the bank fits the L2); larger xcl keeps a bank slice of 1 / 2^xcl resident at the price of
    // 2^xcl XCDs reading every input tile (the launcher chooses per layer).
    int cb, tb;
    {
        const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
        const int xt = 8 >> gm.xcl, xti = xcd >> gm.xcl, xci = xcd & ((1 << gm.xcl) - 1);
        const int ncbx = gm.ncb >> gm.xcl, totx = gm.total >> gm.xcl;
        const int per = totx / xt, rem = totx - per * xt;
        if (j >= per + (xti < rem ? 1 : 0)) return;      // the grid is padded to 8 x the largest range
        const int L = (xti < rem ? xti * (per + 1) : rem * (per + 1) + (xti - rem) * per) + j;
        tb = L / ncbx;
        cb = xci * ncbx + (L - tb * ncbx);
    }
    const int bx = tb % gm.tbx, by = tb / gm.tbx;
    const int n0 = cb * WF_BN;
    const int y0 = 2 * WF_TR * by - 1, x0 = 2 * WF_TC * bx - 1;
    const int nchunks = Cin >> 4;

    // ---- halo staging: piece = (pixel, k-quad).  Buffer loads: an out-of-image piece carries an offset past the descriptor's
    // size and the hardware returns zeros for it (the "same" padding costs no instruction) -------------------------------
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, H * W * Cin * (int)sizeof(float), 0x00020000);
    int h_src[WF_NH];
    int h_dst[WF_NH];
#pragma unroll
    for (int it = 0; it < WF_NH; ++it) {
        const int q = tid + 256 * it;
        const int qq = q < WF_NHP ? q : q - WF_NHP;       // surplus threads duplicate the first pieces (same data, same address): no branch in the loop
        const int px = qq >> 2, pk = qq & 3;
        const int hr = px / WF_HC, hc = px - hr * WF_HC;
        const int gy = y0 + hr, gx = x0 + hc;
        const bool inb = gy >= 0 && gy < H && gx >= 0 && gx < W;
        h_src[it] = inb ? (int)((((unsigned)gy * W + gx) * Cin + 4 * pk) * sizeof(float)) : (int)0xFFFFFFF0u;
        h_dst[it] = ((hr * 2 + (hc & 1)) * WF_HP + (hc >> 1)) * WF_PS + 4 * pk;
    }
    // ---- filter slab staging: 512 pieces of 16 B per slab, two per thread ------------------------------------------------
    int u_dst[WF_NU];
#pragma unroll
    for (int it = 0; it < WF_NU; ++it) {
        const int q = tid + 256 * it;
        const int row = q >> 2, pk = q & 3;
        u_dst[it] = row * 16 + 4 * (pk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3));
    }
    // filter loads: buffer loads too -- lane offset 16 tid, everything else (cout block, chunk, slab, piece) in the scalar offset
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u), 0, 16 * Cin * Cout * (int)sizeof(float), 0x00020000);
    const int u_voff = 16 * tid;
    const int u_blk = cb * 16 * (WF_BN * 16) * (int)sizeof(float);                  // + chunk * u_chunk_stride + (i * 2048 + 1024 * it) * 4
    const int u_chunk_stride = gm.ncb * 16 * (WF_BN * 16) * (int)sizeof(float);

    f32x4 hreg[2][2];            // two pieces in flight + two waiting for their LDS write
    f32x4 ureg[2][WF_NU];        // filter slab t travels in set t & 1: loaded two stages ahead, written to LDS one stage ahead
    bool in_loop = false;
    auto load_halo_piece = [&](f32x4& dst, int it, int chunk) {
        if ((WF_ABLATE & 32) && in_loop) return;
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, h_src[it], chunk * 64, 0));
    };
    auto store_halo_piece = [&](float* hb, const f32x4& src, int it) {
        if ((WF_ABLATE & 8) && in_loop) { asm volatile("" :: "v"(src)); return; }
        *reinterpret_cast<f32x4*>(hb + h_dst[it]) = src;
    };
    auto load_u_piece = [&](f32x4& dst, int chunk, int i, int it) {
        if ((WF_ABLATE & 16) && in_loop) return;
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, u_blk + chunk * u_chunk_stride + (i * WF_U_F + 1024 * it) * (int)sizeof(float), 0));
    };
    auto store_u_piece = [&](int buf, const f32x4& src, int it) {
        if ((WF_ABLATE & 8) && in_loop) { asm volatile("" :: "v"(src)); return; }
        *reinterpret_cast<f32x4*>(ub0 + buf * WF_U_F + u_dst[it]) = src;
    };

    f32x4 acc[16][WF_NT];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int c = 0; c < WF_NT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand addresses: V from patch pixel (a, b) of tile (wave, l16): halo row 2 wave + a, plane b & 1, index l16 + (b >> 1)
    const int v_off = ((4 * wave) * WF_HP + l16) * WF_PS + 4 * kq;
    // U of position j (within the stage's row), cout half c: row = 32 j + 16 c + l16
    const int u_off = l16 * 16 + 4 * (kq ^ ((0x78 >> (2 * ((l16 >> 2) & 3))) & 3));

    // state carried from stage to stage: r_i of the current stage (as register pairs: the transform adds are v_pk_add_f32),
    // the fragments of its first position
    f32x2 r[4][2], v[2][2];
    f32x4 uf[2][WF_NT], d[8];
    // the 8 patch-row reads of position row i: n = 2 b + (0: row a1, 1: row a2)
    auto read_d = [&](const float* hb, int i, int n) {
        const int b = n >> 1, a = (n & 1) ? wf_a2(i) : wf_a1(i);
        if ((WF_ABLATE & 2) && in_loop) return;
        d[n] = *reinterpret_cast<const f32x4*>(hb + v_off + ((2 * a + (b & 1)) * WF_HP + (b >> 1)) * WF_PS);
    };
    // r_i[b] = d[a1][b] -+ d[a2][b], one register pair (m = 2 b + half) per instruction
    auto make_r = [&](int i, int m) {
        const int b = m >> 1;
        if ((WF_ABLATE & 1) && in_loop) return;
        if ((WF_ABLATE & 128) && in_loop && (i & 1)) return;
        r[b][m & 1] = (m & 1) ? wf_pk(wf_sub(i), wf_hi(d[2 * b]), wf_hi(d[2 * b + 1])) : wf_pk(wf_sub(i), wf_lo(d[2 * b]), wf_lo(d[2 * b + 1]));
    };
    // V[i][j] = r_i[a1(j)] -+ r_i[a2(j)]
    auto make_v = [&](int j, int slot, int h) { if ((WF_ABLATE & 1) && in_loop) return; if ((WF_ABLATE & 128) && in_loop && (j & 1)) return; v[slot][h] = wf_pk(wf_sub(j), r[wf_a1(j)][h], r[wf_a2(j)][h]); };
    auto read_u = [&](int buf, int j, int slot, int cc) {
        if ((WF_ABLATE & 4) && in_loop) return;
        uf[slot][cc] = *reinterpret_cast<const f32x4*>(ub0 + buf * WF_U_F + u_off + (WF_BN * j + 16 * cc) * 16);
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------
    {
        f32x4 h0[WF_NH];
#pragma unroll
        for (int it = 0; it < WF_NH; ++it) load_halo_piece(h0[it], it, 0);
#pragma unroll
        for (int it = 0; it < WF_NU; ++it) load_u_piece(ureg[0][it], 0, 0, it);
#pragma unroll
        for (int it = 0; it < WF_NH; ++it) store_halo_piece(halo0, h0[it], it);
#pragma unroll
        for (int it = 0; it < WF_NU; ++it) store_u_piece(0, ureg[0][it], it);
#pragma unroll
        for (int it = 0; it < WF_NU; ++it) load_u_piece(ureg[1][it], 0, 1, it);       // slab 1: written to LDS in stage 0
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 8; ++n) read_d(halo0, 0, n);
#pragma unroll
    for (int m = 0; m < 8; ++m) make_r(0, m);
    read_u(0, 0, 0, 0);
    read_u(0, 0, 0, 1);
    make_v(0, 0, 0);
    make_v(0, 0, 1);

    // the first third of chunk 1's halo (the slot of "stage 3 of the previous chunk")
    {
        const int c1 = nchunks > 1 ? 1 : 0;
        load_halo_piece(hreg[1][0], 0, c1);
        load_halo_piece(hreg[1][1], 1, c1);
    }

    // The K loop is scheduled BY HAND: every statement group below is one MFMA plus at most two other instructions, fenced
    // with sched_barrier(0) so that hipcc keeps the order.  A batch of non-MFMA instructions between two MFMAs holds the
    // wave's issue slot while the matrix pipe drains (measured: ~5 pipe cycles per instruction); one or two of them right
    // behind an MFMA issue are hidden under its 32 cycles.
#define WF_GAP() __builtin_amdgcn_sched_barrier(0)
    if (WF_ABLATE) {                                     // ablated loops still multiply real data (MFMA timing of junk / zero operands differs)
        v[1][0] = v[0][0]; v[1][1] = v[0][1]; uf[1][0] = uf[0][1]; uf[1][1] = uf[0][0];
#pragma unroll
        for (int n = 0; n < 8; ++n) d[n] = uf[n & 1][n >> 2 & 1];
    }
    in_loop = true;
#if WF_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef WF_CLOCKS
    const unsigned long long clk0 = __builtin_readcyclecounter(), real0 = __builtin_amdgcn_s_memrealtime();
#endif
    // one chunk; PAR = chunk parity = halo buffer, a compile-time constant so that every LDS address of the loop is an
    // instruction immediate (address arithmetic on the vector ALU costs matrix-pipe time like any other vector instruction)
    auto chunk_body = [&](const int c, auto par) {
        constexpr int PAR = decltype(par)::value;
        const int cn = (c + 1) < nchunks ? c + 1 : c;    // clamped: past the last chunk the loads re-read it, harmlessly
        const int cnn = (c + 2) < nchunks ? c + 2 : cn;
        float* const hcur = halo0 + PAR * WF_HALO_F;
        float* const hnxt = halo0 + (PAR ^ 1) * WF_HALO_F;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int buf = i & 1;                       // 4 stages per chunk: the slab buffer parity repeats every chunk
            const int ls = i & 1, ss = (i + 1) & 1;      // register sets: loaded in this stage / written to LDS in this stage
            const float* const hrow = i == 3 ? hnxt : hcur;
            const int inext = (i + 1) & 3;
            auto mfma = [&](int j, int slot, int k) {
                const int s = k >> 1, cc = k & 1;
                acc[4 * i + j][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[slot][cc][s], v[slot][s >> 1][s & 1], acc[4 * i + j][cc], 0, 0, 0);
            };
            // Data movement of the stage.  Filter slab s+2 is loaded (one stage of prefetch is shorter than the L2 latency when
            // the block runs alone on its CU), slab s+1 is written to LDS.  Halo of chunk c+1 in thirds: pieces (0,1) were loaded
            // in stage 3 of the previous chunk, (2,3) and (4,5) are loaded in stages 0 and 1, each third is written to LDS one
            // stage after its load -- the whole halo is in LDS before the barrier of stage 2.
            // phase 0: position 0 | fragments + operand of position 1, the stage's global loads and LDS writes
            mfma(0, 0, 0); read_u(buf, 1, 1, 0); if (i < 2) load_u_piece(ureg[ls][0], c, i + 2, 0); else load_u_piece(ureg[ls][0], cn, i - 2, 0); WF_GAP();
            mfma(0, 0, 1); read_u(buf, 1, 1, 1); if (i < 2) load_u_piece(ureg[ls][1], c, i + 2, 1); else load_u_piece(ureg[ls][1], cn, i - 2, 1); WF_GAP();
            mfma(0, 0, 2); make_v(1, 1, 0); if (i < 2) load_halo_piece(hreg[ls][0], 2 * i + 2, cn); if (i == 3) load_halo_piece(hreg[1][0], 0, cnn); WF_GAP();
            mfma(0, 0, 3); make_v(1, 1, 1); if (i < 2) load_halo_piece(hreg[ls][1], 2 * i + 3, cn); if (i == 3) load_halo_piece(hreg[1][1], 1, cnn); WF_GAP();
            mfma(0, 0, 4); store_u_piece(buf ^ 1, ureg[ss][0], 0); WF_GAP();
            mfma(0, 0, 5); store_u_piece(buf ^ 1, ureg[ss][1], 1); WF_GAP();
            mfma(0, 0, 6); if (i < 3) store_halo_piece(hnxt, hreg[ss][0], 2 * i); WF_GAP();
            mfma(0, 0, 7); if (i < 3) store_halo_piece(hnxt, hreg[ss][1], 2 * i + 1); WF_GAP();
            // phase 1: position 1 | position 2's fragments + operand, the patch rows of the NEXT position row (long landed by the barrier)
            mfma(1, 1, 0); read_u(buf, 2, 0, 0); WF_GAP();
            mfma(1, 1, 1); read_u(buf, 2, 0, 1); WF_GAP();
            mfma(1, 1, 2); make_v(2, 0, 0); WF_GAP();
            mfma(1, 1, 3); make_v(2, 0, 1); WF_GAP();
            mfma(1, 1, 4); read_d(hrow, inext, 0); read_d(hrow, inext, 1); WF_GAP();
            mfma(1, 1, 5); read_d(hrow, inext, 2); read_d(hrow, inext, 3); WF_GAP();
            mfma(1, 1, 6); read_d(hrow, inext, 4); read_d(hrow, inext, 5); WF_GAP();
            mfma(1, 1, 7); read_d(hrow, inext, 6); read_d(hrow, inext, 7); WF_GAP();
            // phase 2: position 2 | position 3's fragments + operand (r is dead after it), then the next row's r
            mfma(2, 0, 0); read_u(buf, 3, 1, 0); WF_GAP();
            mfma(2, 0, 1); read_u(buf, 3, 1, 1); WF_GAP();
            mfma(2, 0, 2); make_v(3, 1, 0); WF_GAP();
            mfma(2, 0, 3); make_v(3, 1, 1); WF_GAP();
            mfma(2, 0, 4); make_r(inext, 0); make_r(inext, 1); WF_GAP();
            mfma(2, 0, 5); make_r(inext, 2); make_r(inext, 3); WF_GAP();
            mfma(2, 0, 6); make_r(inext, 4); make_r(inext, 5); WF_GAP();
            mfma(2, 0, 7); make_r(inext, 6); make_r(inext, 7); WF_GAP();
            if (!(WF_ABLATE & 64)) __syncthreads();
            WF_GAP();
            // phase 3: position 3 | the next stage's first fragments (next slab) and first operand
            mfma(3, 1, 0); read_u(buf ^ 1, 0, 0, 0); WF_GAP();
            mfma(3, 1, 1); read_u(buf ^ 1, 0, 0, 1); WF_GAP();
            mfma(3, 1, 2); make_v(0, 0, 0); WF_GAP();
            mfma(3, 1, 3); make_v(0, 0, 1); WF_GAP();
            mfma(3, 1, 4); WF_GAP();
            mfma(3, 1, 5); WF_GAP();
            mfma(3, 1, 6); WF_GAP();
            mfma(3, 1, 7); WF_GAP();
        }
    };
    for (int c = 0; c < nchunks; c += 2) {
        chunk_body(c, std::integral_constant<int, 0>());
        if (c + 1 < nchunks) chunk_body(c + 1, std::integral_constant<int, 1>());
    }
#undef WF_GAP
#if WF_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
#ifdef WF_CLOCKS
    const unsigned long long clk1 = __builtin_readcyclecounter(), real1 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: Y = A^T M A on registers; lane = tile (wave, l16) x couts n0 + 16 c + 4 kq .. + 3 -----------------------
    const int ty = WF_TR * by + wave, tx = WF_TC * bx + l16;
    const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
#ifndef WF_CLOCKS
    if (ty >= th || tx >= tw) return;
#endif
    const int Ho = H >> 1, Wo = W >> 1;
    if (POOL && (ty >= Ho || tx >= Wo)) return;          // floor pooling drops the odd last row / column
#pragma unroll
    for (int c = 0; c < WF_NT; ++c) {
        const int co = n0 + 16 * c + 4 * kq;
        f32x4 s[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[0][j] = (acc[0 + j][c] + acc[4 + j][c]) + acc[8 + j][c];
            s[1][j] = (acc[4 + j][c] - acc[8 + j][c]) - acc[12 + j][c];
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
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
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[a][b][e] = fmaxf(o[a][b][e], 0.f);
        }
        if (POOL) {
            f32x4 m;
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
            *reinterpret_cast<f32x4*>(y + ((size_t)ty * Wo + tx) * Cout + co) = m;
        } else {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int yy = 2 * ty + a;
                if (yy >= H) continue;
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int xx = 2 * tx + b;
                    if (xx < W) *reinterpret_cast<f32x4*>(y + ((size_t)yy * W + xx) * Cout + co) = o[a][b];
                }
            }
        }
    }
#ifdef WF_CLOCKS
    // timing experiment: per wave, written BEHIND the output map (the caller of this build allocates H * W * Cout + 32 * blocks
    // floats: tools/wf_clocks.py): shader cycles and 100 MHz ticks of the K loop, ticks before and after it, entry time
    if (lane == 0) {
        const unsigned long long real_exit = __builtin_amdgcn_s_memrealtime();
        float* o = y + (size_t)(POOL ? (H >> 1) * (W >> 1) : H * W) * Cout + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = (float)(clk1 - clk0); o[1] = (float)(real1 - real0); o[2] = (float)(real0 - real_entry); o[3] = (float)(real_exit - real1);
        o[4] = (float)(real_entry & 0xFFFFFF); o[5] = (float)(real_exit & 0xFFFFFF); o[6] = (float)nchunks;
        unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid)); o[7] = (float)(hwid & 0xFFFFFF);
    }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------
bool conv3x3_winograd_fused_ok(int H, int W, int cin, int cout)
{
    return H >= 1 && W >= 1 && cin >= 16 && cin % 16 == 0 && cout >= WF_BN && cout % WF_BN == 0 &&
           (size_t)H * W * cin * sizeof(float) < ((size_t)1 << 32);
}

int launch_pack_conv3x3_winograd_fused(const float* w, const float* scale, float* u, int cout, int cin, hipStream_t s)
{
    if (cout < WF_BN || cout % WF_BN != 0 || cin < 16 || cin % 16 != 0) return FRCNN_EUNSUPPORTED;
    const size_t total = (size_t)cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_pack_fused_kernel, dim3(blocks), dim3(256), 0, s, w, scale, u, cout, cin);
    return check_launch();
}

int launch_pack_conv3x3_winograd_fused_taps(const float* wp, float* u, int cout, int cin, int data_gradient, hipStream_t s)
{
    const int oc = data_gradient ? cin : cout, ic = data_gradient ? cout : cin;
    if (oc < WF_BN || oc % WF_BN != 0 || ic < 16 || ic % 16 != 0) return FRCNN_EUNSUPPORTED;
    const size_t total = (size_t)cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_pack_fused_taps_kernel, dim3(blocks), dim3(256), 0, s, wp, u, cout, cin, data_gradient ? 1 : 0);
    return check_launch();
}

// How many XCD groups share the cout blocks (log2), by the size of the filter bank.  Measured with FETCH_SIZE on the VGG-16 layers
// (tools/wf_traffic.py, MB fetched through the fabric per launch for xcl = 0 / 1 / 2 / 3): conv2_2 (1 MB bank) 108 / 182 / 348 / 348,
// conv3_2 (4.2 MB) 207 / 166 / 210 / 345, conv4_1 (8.4 MB) 202 / 110 / 84 / 101, conv4_2 (16.8 MB) 416 / 239 / 181 / 209,
// conv5_x (16.8 MB, 37x62) 144 / 80 / 54 / 55.  The layer times do not move (+-0.5 %: the Infinity Cache serves the re-fetches and
// the loads are hidden), the traffic of one image's 13 layers falls from 2.4 GB to 1.4 GB.  Experiments: FRCNN_WF_XCL=0..3 overrides.
static int wf_choose_xcl(int cin, int cout, int ncb)
{
    int maxl = 0;
    while (maxl < 3 && (ncb % (2 << maxl)) == 0) ++maxl;
    static const char* env = getenv("FRCNN_WF_XCL");
    int want;
    if (env) {
        want = atoi(env);
    } else {
        const size_t bank = (size_t)64 * cin * cout;     // 16 positions x 4 bytes
        want = bank <= (size_t)5 << 19 ? 0 : (bank <= (size_t)5 << 20 ? 1 : 2);
    }
    return want < 0 ? 0 : (want > maxl ? maxl : want);
}

int launch_conv3x3_winograd_fused(const float* x, const float* u, const float* b, float* y, int H, int W, int cin, int cout,
                                  unsigned flags, hipStream_t s)
{
    if (!conv3x3_winograd_fused_ok(H, W, cin, cout)) return FRCNN_EUNSUPPORTED;
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    const int th = cdiv(H, 2), tw = cdiv(W, 2);
    WfGeom gm;
    gm.tbx = cdiv(tw, WF_TC);
    gm.tby = cdiv(th, WF_TR);
    gm.ncb = cout / WF_BN;
    const long long total = (long long)gm.tbx * gm.tby * gm.ncb;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    gm.total = (int)total;
    gm.xcl = wf_choose_xcl(cin, cout, gm.ncb);
    const int xt = 8 >> gm.xcl, totx = gm.total >> gm.xcl;
    const unsigned grid = 8u * (unsigned)cdiv(totx, xt);
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    static bool attr_set = false;
    if (!attr_set) {
        FRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)WF_LDS_BYTES));
        FRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fused_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)WF_LDS_BYTES));
        attr_set = true;
        if (getenv("FRCNN_DEBUG_OCCUPANCY")) {
            int nb = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(wino_fused_kernel<false>), 256, WF_LDS_BYTES);
            fprintf(stderr, "wino_fused_kernel: %d blocks per CU at %zu B of LDS\n", nb, WF_LDS_BYTES);
        }
    }
    if (flags & FRCNN_POOL2)
        hipLaunchKernelGGL(wino_fused_kernel<true>, dim3(grid), dim3(256), WF_LDS_BYTES, s, x, u, b, y, H, W, cin, cout, relu, gm);
    else
        hipLaunchKernelGGL(wino_fused_kernel<false>, dim3(grid), dim3(256), WF_LDS_BYTES, s, x, u, b, y, H, W, cin, cout, relu, gm);
    return check_launch();
}

}  // namespace frcnn
