// winofused.hip -- 3x3 stride-1 "same" convolution as Winograd F(2x2,3x3) in float32, ONE launch per layer.
// Replaces the three launches of csrc/winograd.hip (input transform -> 16 batched GEMMs -> output transform) for a single
// map: models/vgg16.py:77-96 (conv1_2 ... conv5_3), models/rpn.py:88 (RPN trunk), the stride-1 3x3 convolutions of ResNet's
// layer2 / layer3 (models/resnet.py:38-46 over torchvision's Bottleneck).
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// No V / M scratch: HBM traffic of a layer is its input, its filter bank and its output.
//
// Work decomposition.  A block owns 2 x 16 Winograd tiles (4 x 32 output pixels) x 64 output channels and ALL 16 Winograd
// positions of that slab.  Its four waves are (tile row tr, position-row pair h): wave (tr, h) owns the tiles of row tr, the
// position rows i = 2 h, 2 h + 1 and all 64 couts: accumulators acc[8 positions][4 cout tiles] = 32 MFMA 16x16 tiles
// (v_mfma_f32_16x16x4_f32, exact f32, 4 registers each = 128 registers).  The position rows are split over the wave PAIR rather
// than the cout tiles so that each wave forms the transformed operand of only its own two rows: every vector instruction costs
// matrix-pipe time (measured: ~9 cycles per packed add) and a block of 64 couts needs half the adds per
// MFMA of the 32-cout block it replaces.  Because every position's accumulator is live, the output transform A^T M A (+ bias,
// ReLU, 2x2 max-pool: a Winograd tile IS a pooling window) runs on registers in the epilogue: column pass per wave, the
// partner's two rows arrive through LDS, row pass, store.
//
// K loop over 16-channel chunks; per chunk
//   * the (4+2) x (32+2) pixel input halo of the block's tiles is staged ONCE in LDS (zero padding = buffer loads past the
//     descriptor).  The B^T d B input transform is evaluated when the MFMA operand is formed: for position row i the wave reads
//     the two patch rows that B^T combines (8 ds_read_b128), r_i[b] = d[a1][b] +- d[a2][b], and the four operands of the row are
//     V[i][j] = r_i[b1] +- r_i[b2] -- the same float32 operation order as wino_input_kernel;
//   * the filter bank arrives in four slabs of 16 KB (half a position row of BOTH wave groups: 4 positions x 64 couts x 16
//     channels, contiguous in the [chunk][cout block][slab][h][jj][64][16] layout of wino_pack_fused_kernel), double buffered;
//   * MFMA roles: A = U (rows = couts), B = V (columns = tiles), so a lane's four accumulator registers are four CONSECUTIVE
//     output channels of one tile -> 16-byte stores.
// A "stage" is (chunk, slab): 32 MFMAs per wave (2 positions x 4 cout tiles x 4 k-steps = 1024 matrix-pipe cycles), 8 fragment
// reads, on average 8 packed adds and 4 patch-row reads, 4 global loads + 4 LDS writes of the next filter slab, one barrier.
// The loop is ordered by hand (one MFMA + at most two other instructions between sched_barrier fences).
//
// LDS layouts (both conflict-free for ds_read_b128 with lane = row + 16 * k-quad, checked against the lane groups of
// MI355X_MICROARCH.md): halo pixel = 16 channels padded to 24 floats, pixels de-interleaved by column parity so that the 16
// tiles of a wave read 16 consecutive pixels; filter rows = 16 floats, the k-quad slot XOR-swizzled by {0,2,3,1}[(row >> 2) & 3].
// 71,936 B per block: two blocks per CU.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace frcnn {

static constexpr int WF_TR = 2, WF_TC = 16;              // tile rows / columns per block (a wave pair per tile row)
static constexpr int WF_HR = 2 * WF_TR + 2;              // 6 halo rows
static constexpr int WF_HC = 2 * WF_TC + 2;              // 34 halo columns
static constexpr int WF_HP = WF_HC / 2;                  // 17 pixels per parity plane of a halo row
static constexpr int WF_PS = 24;                         // floats per halo pixel in LDS
static constexpr int WF_NPIX = WF_HR * WF_HC;            // 204
static constexpr int WF_HALO_F = WF_NPIX * WF_PS;        // 4896 floats
static constexpr int WF_BN = 64;                         // output channels per block (4 MFMA tiles of 16)
static constexpr int WF_U_F = 4 * WF_BN * 16;            // floats per filter slab (4 positions x 64 couts x 16 channels)
static constexpr int WF_NU = WF_U_F / 4 / 256;           // 4 16-byte filter pieces per thread per slab
static constexpr int WF_NHP = WF_NPIX * 4;               // 816 16-byte halo pieces per chunk
static constexpr int WF_NH = (WF_NHP + 255) / 256;       // 4 per thread
#ifndef WF_PRIO
#define WF_PRIO 1                                        // s_setprio 3 outside the K loop (prologue 4.4 -> 2.3 us; the total does not move)
#endif
static constexpr size_t WF_LDS_BYTES = (size_t)(2 * WF_HALO_F + 2 * WF_U_F) * sizeof(float);   // 71,936
static_assert(WF_NH == 4 && WF_NU == 4, "the stage schedule places exactly these pieces");

// B^T rows: r_i = d[A1] (-|+) d[A2];  the same table gives the column combination V[i][j] = r_i[A1_j] (-|+) r_i[A2_j]
__device__ __forceinline__ constexpr int wf_a1(int i) { return i == 0 ? 0 : (i == 1 ? 1 : (i == 2 ? 2 : 1)); }
__device__ __forceinline__ constexpr int wf_a2(int i) { return i == 0 ? 2 : (i == 1 ? 2 : (i == 2 ? 1 : 3)); }
__device__ __forceinline__ constexpr bool wf_sub(int i) { return i != 1; }

// storage slot of Winograd position (i, j) inside a (chunk, cout block): slab = (i & 1) * 2 + (j >> 1) [the stage that uses it],
// then the wave group h = i >> 1, then jj = j & 1
__device__ __forceinline__ constexpr int wf_slot(int i, int j) { return (((i & 1) * 2 + (j >> 1)) * 2 + (i >> 1)) * 2 + (j & 1); }

// U'[chunk][cout block][wf_slot(i, j)][64][16] = (G g G^T)[i][j] of filter (cout, cin);  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
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
            for (int b = 0; b < 4; ++b) u[base + (size_t)wf_slot(a, b) * (WF_BN * 16)] = (float)q[b];
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
            for (int b = 0; b < 4; ++b) u[base + (size_t)wf_slot(a, b) * (WF_BN * 16)] = (float)q[b];
        }
    }
}

struct WfGeom { int tbx, tby, ncb, total, xcl, tbm; };   // xcl: log2 of the number of XCD groups the cout blocks are split over;
                                                         // tbm = tbx * tby: tile blocks per map (the maps of a launch follow each other)

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 wf_lo(const f32x4& a) { return __builtin_shufflevector(a, a, 0, 1); }
__device__ __forceinline__ f32x2 wf_hi(const f32x4& a) { return __builtin_shufflevector(a, a, 2, 3); }
// a -+ b on a register pair in one instruction (the neg modifiers make it a - b: the same IEEE result as v_sub_f32).
// Two scalar adds instead cost the same matrix-pipe time (measured).
__device__ __forceinline__ f32x2 wf_pk(bool sub, f32x2 a, f32x2 b)
{
    f32x2 o;
    if (sub) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    else asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

template <bool POOL>
__global__ __launch_bounds__(256, 2)
void wino_fused_kernel(const float* __restrict__ x_maps, const float* __restrict__ u, const float* __restrict__ bias,
                       float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int relu, WfGeom gm)
{
#ifdef WF_CLOCKS
    const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
#if WF_PRIO
    __builtin_amdgcn_s_setprio(3);                       // prologue and epilogue ahead of the co-resident block's K loop
#endif
    extern __shared__ __attribute__((aligned(16))) float smem_wf[];
    float* const halo0 = smem_wf;                           // 2 halo buffers
    float* const ub0 = smem_wf + 2 * WF_HALO_F;             // 2 filter slab buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int tr = wave >> 1;                               // tile row of the wave; its position rows: 2 h, 2 h + 1 with h = wave & 1

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
    // several maps in one launch (a batch of images through a ResNet bottleneck): tile blocks are numbered map-major, the map index
    // only shifts the two base pointers (scalar arithmetic: the block is uniform)
    const int map = tb / gm.tbm;
    tb -= map * gm.tbm;
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (H >> 1) * (W >> 1) : H * W) * Cout;
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
    // ---- filter slab staging: 1024 pieces of 16 B per slab, four per thread; buffer loads too -- lane offset 16 tid, everything
    // else (cout block, chunk, slab, piece) in the scalar offset ------------------------------------------------------------
    int u_dst[WF_NU];
#pragma unroll
    for (int it = 0; it < WF_NU; ++it) {
        const int q = tid + 256 * it;
        const int row = q >> 2, pk = q & 3;
        u_dst[it] = row * 16 + 4 * (pk ^ ((0x78 >> (2 * ((row >> 2) & 3))) & 3));
    }
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(u), 0, 16 * Cin * Cout * (int)sizeof(float), 0x00020000);
    const int u_voff = 16 * tid;
    const int u_blk = cb * 16 * (WF_BN * 16) * (int)sizeof(float);                  // + chunk * u_chunk_stride + (slab * 4096 + 1024 * it) * 4
    const int u_chunk_stride = gm.ncb * 16 * (WF_BN * 16) * (int)sizeof(float);

    f32x4 hreg[2][2];            // halo pieces (0,1) and (2,3) of the next chunk on their way to LDS
    f32x4 ureg[2][WF_NU];        // filter slab s travels in set s & 1: loaded two stages ahead, written to LDS one stage ahead
    auto load_halo_piece = [&](f32x4& dst, int it, int chunk) {
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, h_src[it], chunk * 64, 0));
    };
    auto store_halo_piece = [&](float* hb, const f32x4& src, int it) { *reinterpret_cast<f32x4*>(hb + h_dst[it]) = src; };
    auto load_u_piece = [&](f32x4& dst, int chunk, int slab, int it) {
        dst = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, u_voff, u_blk + chunk * u_chunk_stride + (slab * WF_U_F + 1024 * it) * (int)sizeof(float), 0));
    };
    auto store_u_piece = [&](int buf, const f32x4& src, int it) { *reinterpret_cast<f32x4*>(ub0 + buf * WF_U_F + u_dst[it]) = src; };

    f32x4 acc[8][4];             // [2 t + ... position (t, j) = 4 t + j][cout tile]
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand addresses: V from patch pixel (a, b) of tile (tr, l16): halo row 2 tr + a, plane b & 1, index l16 + (b >> 1)
    const int v_off = ((4 * tr) * WF_HP + l16) * WF_PS + 4 * kq;
    // U of slab slot sl (= 2 h + jj), cout tile c: row = 64 sl + 16 c + l16
    const int u_off = l16 * 16 + 4 * (kq ^ ((0x78 >> (2 * ((l16 >> 2) & 3))) & 3));

#ifdef WF_CLOCKS
    unsigned long long clk0 = 0, real0 = 0, clk1 = 0, real1 = 0;
#endif
    f32x4 M[16][2];              // epilogue: all 16 positions of the two cout tiles this wave finishes
    // Everything that depends on the wave group h (which patch rows B^T combines, with which sign, which filter slots) must be a
    // compile-time constant of the loop: the K loop exists once per h.
    auto run = [&](auto hsel) {
        constexpr int HG = decltype(hsel)::value;
        // state carried from stage to stage: r of the current position row (as register pairs: the adds are v_pk_add_f32), the two
        // operands in flight, two fragment sets of two cout tiles each
        f32x2 r[4][2], v[2][2];
        f32x4 uf[2][2], d[8];
        // the 8 patch-row reads of position row i: n = 2 b + (0: row a1, 1: row a2)
        auto read_d = [&](const float* hb, int i, int n) {
            const int b = n >> 1, a = (n & 1) ? wf_a2(i) : wf_a1(i);
            d[n] = *reinterpret_cast<const f32x4*>(hb + v_off + ((2 * a + (b & 1)) * WF_HP + (b >> 1)) * WF_PS);
        };
        // r_i[b] = d[a1][b] -+ d[a2][b], one register pair (m = 2 b + half) per instruction
        auto make_r = [&](int i, int m) {
            const int b = m >> 1;
            r[b][m & 1] = (m & 1) ? wf_pk(wf_sub(i), wf_hi(d[2 * b]), wf_hi(d[2 * b + 1])) : wf_pk(wf_sub(i), wf_lo(d[2 * b]), wf_lo(d[2 * b + 1]));
        };
        // V[i][j] = r_i[a1(j)] -+ r_i[a2(j)]
        auto make_v = [&](int j, int slot, int hf) { v[slot][hf] = wf_pk(wf_sub(j), r[wf_a1(j)][hf], r[wf_a2(j)][hf]); };
        // fragment of filter slot (HG, jj), cout tile c of the slab in buffer `buf`
        auto read_u = [&](int buf, int jj, int c, int set, int e) {
            uf[set][e] = *reinterpret_cast<const f32x4*>(ub0 + buf * WF_U_F + u_off + ((2 * HG + jj) * WF_BN + 16 * c) * 16);
        };

        // ---- prologue of the group: row 2 HG of chunk 0, its first operand, the first fragments ---------------------------------
#pragma unroll
        for (int n = 0; n < 8; ++n) read_d(halo0, 2 * HG, n);
#pragma unroll
        for (int m = 0; m < 8; ++m) make_r(2 * HG, m);
        read_u(0, 0, 0, 0, 0);
        read_u(0, 0, 1, 0, 1);
        make_v(0, 0, 0);
        make_v(0, 0, 1);

#define WF_GAP() __builtin_amdgcn_sched_barrier(0)
#if WF_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
#ifdef WF_CLOCKS
        clk0 = __builtin_readcyclecounter(); real0 = __builtin_amdgcn_s_memrealtime();
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
            for (int q = 0; q < 4; ++q) {
                // stage (chunk c, slab q): position row i = 2 HG + t of the wave, positions j = 2 hf and 2 hf + 1
                const int t = q >> 1, hf = q & 1;
                const int buf = q & 1;                       // 4 slabs per chunk: the slab buffer parity repeats every chunk
                const int ls = q & 1, ss = (q + 1) & 1;      // filter register sets: loaded in this stage (slab s + 2) / written to LDS (slab s + 1)
                const int pa = 4 * t + 2 * hf, pb = pa + 1;  // accumulator rows of the two positions
                const int jn = (2 * hf + 2) & 3;             // the next stage's first position
                const int in_ = 2 * HG + (t ^ 1);            // the position row after this one (next chunk's when t == 1)
                const float* const hrow = t == 1 ? hnxt : hcur;
                auto mfma = [&](int p, int cbase, int set, int vs, int k) {
                    const int s = k >> 1, e = k & 1;
                    acc[p][cbase + e] = __builtin_amdgcn_mfma_f32_16x16x4f32(uf[set][e][s], v[vs][s >> 1][s & 1], acc[p][cbase + e], 0, 0, 0);
                };
                // group 0: position a, cout tiles 0,1 | fragments of (a, tiles 2,3), operand of position b, the stage's filter loads
                mfma(pa, 0, 0, 0, 0); read_u(buf, 0, 2, 1, 0); WF_GAP();
                mfma(pa, 0, 0, 0, 1); read_u(buf, 0, 3, 1, 1); WF_GAP();
                mfma(pa, 0, 0, 0, 2); make_v(2 * hf + 1, 1, 0); WF_GAP();
                mfma(pa, 0, 0, 0, 3); make_v(2 * hf + 1, 1, 1); WF_GAP();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    mfma(pa, 0, 0, 0, 4 + it);
                    if (q < 2) load_u_piece(ureg[ls][it], c, q + 2, it); else load_u_piece(ureg[ls][it], cn, q - 2, it);
                    WF_GAP();
                }
                // group 1: position a, cout tiles 2,3 | fragments of (b, tiles 0,1); second half of a row: the next row's r
                // (r is dead once the row's last operand exists); first stage / last stage of a chunk: halo loads of the next chunk
                mfma(pa, 2, 1, 0, 0); read_u(buf, 1, 0, 0, 0); WF_GAP();
                mfma(pa, 2, 1, 0, 1); read_u(buf, 1, 1, 0, 1); WF_GAP();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mfma(pa, 2, 1, 0, 2 + g);
                    if (hf == 1) { make_r(in_, 2 * g); make_r(in_, 2 * g + 1); }
                    WF_GAP();
                }
                mfma(pa, 2, 1, 0, 6); if (q == 0) load_halo_piece(hreg[1][0], 2, cn); if (q == 3) load_halo_piece(hreg[0][0], 0, cnn); WF_GAP();
                mfma(pa, 2, 1, 0, 7); if (q == 0) load_halo_piece(hreg[1][1], 3, cn); if (q == 3) load_halo_piece(hreg[0][1], 1, cnn); WF_GAP();
                // group 2: position b, cout tiles 0,1 | fragments of (b, tiles 2,3), the LDS writes (long done by the barrier)
                mfma(pb, 0, 0, 1, 0); read_u(buf, 1, 2, 1, 0); WF_GAP();
                mfma(pb, 0, 0, 1, 1); read_u(buf, 1, 3, 1, 1); WF_GAP();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    mfma(pb, 0, 0, 1, 2 + it);
                    store_u_piece(buf ^ 1, ureg[ss][it], it);
                    WF_GAP();
                }
                mfma(pb, 0, 0, 1, 6); if (q < 2) store_halo_piece(hnxt, hreg[q][0], 2 * q); WF_GAP();
                mfma(pb, 0, 0, 1, 7); if (q < 2) store_halo_piece(hnxt, hreg[q][1], 2 * q + 1); WF_GAP();
                __syncthreads();
                WF_GAP();
                // group 3: position b, cout tiles 2,3 | the next stage's first fragments (next slab) and first operand; first half
                // of a row: the patch rows of the NEXT position row
                mfma(pb, 2, 1, 1, 0); read_u(buf ^ 1, 0, 0, 0, 0); WF_GAP();
                mfma(pb, 2, 1, 1, 1); read_u(buf ^ 1, 0, 1, 0, 1); WF_GAP();
                mfma(pb, 2, 1, 1, 2); make_v(jn, 0, 0); WF_GAP();
                mfma(pb, 2, 1, 1, 3); make_v(jn, 0, 1); WF_GAP();
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    mfma(pb, 2, 1, 1, 4 + g);
                    if (hf == 0) { read_d(hrow, in_, 2 * g); read_d(hrow, in_, 2 * g + 1); }
                    WF_GAP();
                }
            }
        };
        for (int c = 0; c < nchunks; c += 2) {
            chunk_body(c, std::integral_constant<int, 0>());
            if (c + 1 < nchunks) chunk_body(c + 1, std::integral_constant<int, 1>());
        }
#undef WF_GAP
#ifdef WF_CLOCKS
        clk1 = __builtin_readcyclecounter(); real1 = __builtin_amdgcn_s_memrealtime();
#endif
#if WF_PRIO
        __builtin_amdgcn_s_setprio(3);
#endif
        // The pair (tr, 0) / (tr, 1) holds the position rows {0, 1} / {2, 3} of all four cout tiles.  Wave h finishes the tiles 2 h and
        // 2 h + 1 and needs the partner's two rows of them: 16 accumulators per lane travel through LDS (16 KB per wave, in the
        // staging buffers).  Static indices only: a select between accumulator registers by a run-time h puts them in scratch.
        __syncthreads();                                 // every wave has left the K loop: the staging buffers are free
        {
            float* const mine = smem_wf + (size_t)wave * 16 * 64 * 4 + lane * 4;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    *reinterpret_cast<f32x4*>(mine + (p * 2 + e) * 64 * 4) = acc[p][2 * (1 - HG) + e];      // the partner's tiles
        }
        __syncthreads();
        {
            const float* const theirs = smem_wf + (size_t)(wave ^ 1) * 16 * 64 * 4 + lane * 4;
#pragma unroll
            for (int p = 0; p < 8; ++p)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    M[8 * HG + p][e] = acc[p][2 * HG + e];
                    M[8 * (1 - HG) + p][e] = *reinterpret_cast<const f32x4*>(theirs + (p * 2 + e) * 64 * 4);
                }
        }
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
        // the first half of chunk 1's halo (the slot of "stage 3 of the previous chunk")
        const int c1 = nchunks > 1 ? 1 : 0;
        load_halo_piece(hreg[0][0], 0, c1);
        load_halo_piece(hreg[0][1], 1, c1);
    }
    __syncthreads();
    if (wave & 1) run(std::integral_constant<int, 1>());
    else run(std::integral_constant<int, 0>());

    // ---- epilogue: Y = A^T M A; lane = tile (tr, l16) x couts n0 + 16 c + 4 kq .. + 3 ----------------------------------------
    // M (filled at the end of `run`) holds all 16 positions of the wave's two cout tiles; the transform runs in the row-then-column
    // order of wino_output_kernel.
    const int h = wave & 1;
#ifdef WF_CLOCKS
    // timing experiment: per wave, written BEHIND the output map (the caller of this build allocates H * W * Cout + 32 * gridDim.x
    // floats: tools/wf_clocks.py): shader cycles and 100 MHz ticks of the K loop, ticks before and after it, entry time
    if (lane == 0) {
        const unsigned long long real_exit = __builtin_amdgcn_s_memrealtime();
        float* o = y + (size_t)(POOL ? (H >> 1) * (W >> 1) : H * W) * Cout + ((size_t)blockIdx.x * 4 + wave) * 8;
        o[0] = (float)(clk1 - clk0); o[1] = (float)(real1 - real0); o[2] = (float)(real0 - real_entry); o[3] = (float)(real_exit - real1);
        o[4] = (float)(real_entry & 0xFFFFFF); o[5] = (float)(real_exit & 0xFFFFFF); o[6] = (float)nchunks; o[7] = 1.f;
    }
#endif
    const int ty = WF_TR * by + tr, tx = WF_TC * bx + l16;
    const int th = (H + 1) >> 1, tw = (W + 1) >> 1;
    if (ty >= th || tx >= tw) return;
    const int Ho = H >> 1, Wo = W >> 1;
    if (POOL && (ty >= Ho || tx >= Wo)) return;          // floor pooling drops the odd last row / column
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int co = n0 + 16 * (2 * h + e) + 4 * kq;
        f32x4 sr[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sr[0][j] = (M[0 + j][e] + M[4 + j][e]) + M[8 + j][e];
            sr[1][j] = (M[4 + j][e] - M[8 + j][e]) - M[12 + j][e];
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + co);
        f32x4 o[2][2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            o[a][0] = ((sr[a][0] + sr[a][1]) + sr[a][2]) + bv;
            o[a][1] = ((sr[a][1] - sr[a][2]) - sr[a][3]) + bv;
        }
        if (relu) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int q = 0; q < 4; ++q) o[a][b][q] = fmaxf(o[a][b][q], 0.f);
        }
        if (POOL) {
            f32x4 m;
#pragma unroll
            for (int q = 0; q < 4; ++q) m[q] = fmaxf(fmaxf(o[0][0][q], o[0][1][q]), fmaxf(o[1][0][q], o[1][1][q]));
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
}

// ---- host side ------------------------------------------------------------------------------------------------------------
bool conv3x3_winograd_fused_ok(int H, int W, int cin, int cout)
{
    // the buffer descriptors' sizes and every byte offset inside them are 32-bit quantities computed in int
    return H >= 1 && W >= 1 && cin >= 16 && cin % 16 == 0 && cout >= WF_BN && cout % WF_BN == 0 &&
           (size_t)H * W * cin * sizeof(float) < ((size_t)1 << 31) && (size_t)16 * cin * cout * sizeof(float) < ((size_t)1 << 31);
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
    static const char* env = frcnn_knob("FRCNN_WF_XCL");
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
                                  unsigned flags, hipStream_t s, int n_maps)
{
    if (n_maps < 1) return FRCNN_EINVAL;
    if (!conv3x3_winograd_fused_ok(H, W, cin, cout)) return FRCNN_EUNSUPPORTED;
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    const int th = cdiv(H, 2), tw = cdiv(W, 2);
    WfGeom gm;
    gm.tbx = cdiv(tw, WF_TC);
    gm.tby = cdiv(th, WF_TR);
    gm.ncb = cout / WF_BN;
    gm.tbm = gm.tbx * gm.tby;
    const long long total = (long long)gm.tbm * n_maps * gm.ncb;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    gm.total = (int)total;
    gm.xcl = wf_choose_xcl(cin, cout, gm.ncb);
    const int xt = 8 >> gm.xcl, totx = gm.total >> gm.xcl;
    const unsigned grid = 8u * (unsigned)cdiv(totx, xt);
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    FRCNN_MAX_LDS_ONCE(wino_fused_kernel<true>, WF_LDS_BYTES);
    FRCNN_MAX_LDS_ONCE(wino_fused_kernel<false>, WF_LDS_BYTES);
    static bool occ_printed = false;
    if (!occ_printed) {
        occ_printed = true;
        if (frcnn_knob("FRCNN_DEBUG_OCCUPANCY")) {
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
