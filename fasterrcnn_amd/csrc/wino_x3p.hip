// wino_x3p.hip -- the one-launch f32x3 Winograd F(2x2,3x3) layer with 128 OUTPUT CHANNELS per block, in TWO PASSES over the input
// channels (round 6).  Same layer as csrc/wino_x3f.hip (pytorch/FasterRCNN/models/vgg16.py:77-96: 3x3 convolution + ReLU, MaxPool2d after
// each block; models/rpn.py:88: the RPN trunk), same arithmetic bit for bit: per-tile scale 2^e, two-term fp16 split, float32
// accumulation per (position, tile, output channel) in the order filter lo x V hi, filter hi x V hi, filter hi x V lo per 16-channel
// chunk, chunks in order; the output transform's operation order is wino_x3d_kernel's.
//
// Why: wino_x3d_kernel (64 tiles x 64 channels x 16 positions, a wave owns a position ROW) forms every operand V(h, j) for SIX MFMAs:
// ~6.3 vector / LDS / load instructions per MFMA, and a SIMD issues one instruction per ~4.5-8 cycles whoever it belongs to -- 2470-2545
// cycles per chunk against the 1536 of the MFMAs (profiles/r06/xd_clocks_D.txt).  An operand's cost is independent of the number of output
// channels it meets, so the lever is output channels per block; the 256 accumulator registers of a wave are 16 tiles of 32 x 32, and
//   (tile halves T, channel tiles C, positions P) = (2, 2, 4)   today:   8 operands + 16 filter pieces (1 KB each) per 48 MFMAs
//                                                   (1, 4, 4)   32 tiles x 128 channels: 4 operands but 32 filter pieces -- the L1 delivers
//                                                               ~57 B / clock / CU (tools/micro/split_fill.hip part 3): 2245 cycles
//                                                   (2, 4, 2)   HERE: 4 operands + 16 filter pieces per 48 MFMAs, a wave owns HALF a position
//                                                               row; the other half is a second pass over the input channels.
// Pass q in {0, 1}: wave w holds positions (i = w, j = 2 q + jj), jj in {0, 1}, of 64 tiles x 128 channels.  After pass 0 the 256
// accumulators go to a block-private scratch (256 KB, written and read once by the same lanes: a spill), pass 1 runs the same loop on the
// other position pair, and the epilogue is wino_x3d_kernel's with j = 0, 1 coming back from the scratch: column combination in
// registers, rows through LDS, in two rounds of 64 channels.
// Per chunk and pass a wave forms FOUR operands (h, jj) for TWELVE MFMAs each (4 channel tiles x 3 products) from THREE r columns:
//   pass 0: V0 2^e = fma(r0, 2^e, -r2 2^e), V1 2^e = fma(r1, 2^e, r2 2^e)        (columns a = 0, b = 1, s = 2; s is the scaled one)
//   pass 1: V2 2^e = fma(r2, 2^e, -r1 2^e), V3 2^e = fma(r3, -2^e, r1 2^e)       (columns a = 2, b = 3, s = 1)
// -- the values of wino_x3d_kernel (r 2^e is exact, every sum is rounded once).  The pass is a RUN-TIME parameter of one loop body: it
// selects three LDS column offsets, the sign of one multiplier and the filter base.
#include "wino_x3_shared.h"

namespace frcnn {

static constexpr int XP_CB = 128;                                               // output channels per block
static constexpr int XP_SC_OFFSET = XD_M_BYTES;                                 // filter scales [16][128], bias [128]
static constexpr int XP_CM_OFFSET = XP_SC_OFFSET + 16 * XP_CB * 4 + XP_CB * 4;  // channel maxima of the 10 x 34 halo pixels (512 floats)
static constexpr size_t XP_LDS_BYTES = XP_CM_OFFSET + 512 * 4;                  // 150,016
static constexpr size_t XP_SPILL_FLOATS = 256 * 256;                            // per block: 256 lanes x 256 accumulator registers
typedef unsigned xp_u32x4 __attribute__((ext_vector_type(4)));

template <bool POOL>
__global__ __launch_bounds__(256, 1)
void wino_x3p_kernel(const float* __restrict__ x_maps, const float* __restrict__ cmax_maps, const unsigned char* __restrict__ ublob,
                     const float* __restrict__ bias, float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int u_rbt, int relu,
                     XfGeom gm, float* __restrict__ cmax_out_maps, float* __restrict__ spill_all)
{
#ifdef XD_CLOCKS
    const unsigned long long xd_t_in = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_xf[];
    float* const hbuf0 = reinterpret_cast<float*>(smem_xf);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K16 = Cin >> 4;

    int cb, bx, by, map;
    if (!xd_block_to_tile(gm, blockIdx.x, cb, bx, by, map)) return;          // cb: block of 128 output channels (gm.ncb = Cout / 128)
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    const float* __restrict__ const cmax = cmax_maps + (size_t)map * H * W;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout;
    float* __restrict__ const cmax_out = cmax_out_maps ? cmax_out_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) : nullptr;
    // the block's scratch behind a buffer descriptor: slot k (0 .. 63) of lane tid at byte 4096 k + 16 tid -- 4096 k travels in the SCALAR offset
    // (as 64-bit global addresses the compiler keeps 64 address pairs alive across the loop)
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(spill_all + (size_t)blockIdx.x * XP_SPILL_FLOATS, 0, (int)(XP_SPILL_FLOATS * sizeof(float)), 0x00020000);
    const int tid16 = tid * 16;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, H * W * Cin * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cmax), 0, H * W * (int)sizeof(float), 0x00020000);

    const int tl = lane & 31, kh = lane >> 5;
    int tyl, txl;
    xd_slot_tile(tl, tyl, txl);                                              // (a ds_read_b128 lane group = one tile row: conflict-free patch reads)
    float mult[2], vinv[2];

    // ---- halo staging by LDS-DMA (wino_x3d_kernel's: ring of three buffers, two chunks ahead) ------------------------------------------------
    const int hy0 = 8 * by - 1, hx0 = 2 * XF_TC * bx - 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    int h_src[XD_NDMA];
    auto halo_sources = [&]() {
#pragma unroll
        for (int it = 0; it < XD_NDMA; ++it) {
            const unsigned P = (unsigned)((it * 4 + wave) * 64 + lane);
            const unsigned slot = __umul24(P, 52429u) >> 18, part = P - 5u * slot;            // P / 5, P % 5
            const unsigned hr = __umul24(slot, 1928u) >> 16, rem = slot - (unsigned)XF_HC * hr;   // slot / 34, slot % 34
            const unsigned par = rem >= (unsigned)XD_HP ? 1u : 0u, hc = 2u * (rem - par * (unsigned)XD_HP) + par;
            const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
            const bool inb = part < 4u && hr < (unsigned)X3_HR && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned off = (__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * (unsigned)(Cin * 4) + 16u * part;
            h_src[it] = inb ? (int)off : (int)0xFFFFFFF0u;
        }
    };
    auto dma_halo1 = [&](float* hb, int chunk_off, auto IT) {                // ONE piece (instruction) of a chunk's halo
        if (XD_ABLATE & 8) return;
        constexpr int it = decltype(IT)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (xd_lds_ptr)(reinterpret_cast<unsigned char*>(hb) + (it * 4 + wave_u) * 1024), 16, h_src[it],
                                                 chunk_off, 0, 0);
    };
    auto dma_halo = [&](float* hb, int chunk_off) {
        dma_halo1(hb, chunk_off, XdInt<0>{}); dma_halo1(hb, chunk_off, XdInt<1>{}); dma_halo1(hb, chunk_off, XdInt<2>{}); dma_halo1(hb, chunk_off, XdInt<3>{});
        dma_halo1(hb, chunk_off, XdInt<4>{}); dma_halo1(hb, chunk_off, XdInt<5>{}); dma_halo1(hb, chunk_off, XdInt<6>{});
    };

    // ---- filter fragments: piece (position p, chunk c, 32-channel row block rb, term t) = ublob + ((p K16 + c) u_rbt + rb) 2 KB + t 1 KB ------
    xf_f16x8 U[2][2][4][2];                                                  // [register set][jj][channel tile][0 = hi, 1 = lo]
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(ublob), 0, 16 * K16 * u_rbt * HX_RB, 0x00020000);
    const int chunk_stride = u_rbt * HX_RB, pos_stride = K16 * chunk_stride;
    const int ub_row = 4 * wave * pos_stride + 4 * cb * HX_RB;               // position (i = wave, j = 0), the block's first row block
    const int lane16 = lane * 16;
    // piece (jj, ct, t) of the chunk at scalar offset so (= ub_row + (2 q + jj) pos_stride + c chunk_stride): the constants in the SCALAR offset
    auto load_piece = [&](int so, auto SET, auto JJ, auto CT, auto T) {
        if (XD_ABLATE & 4) return;
        constexpr int set = decltype(SET)::value, jj = decltype(JJ)::value, ct = decltype(CT)::value, t = decltype(T)::value;
        U[set][jj][ct][t] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16, so + ct * HX_RB + t * HX_PIECE, 0));
    };

    f32x16 acc[2][2][4];                                                     // [tile half][jj][channel tile]; never zeroed (first MFMA: C = 0)

    // ---- operand formation -------------------------------------------------------------------------------------------------------------
    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int a2 = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
    const float rsgn = wave != 1 ? -1.0f : 1.0f;
    // LDS halo layout [row][column parity][17 slots][20 floats] (wino_x3_shared.h); float index of (patch row a, column b, half h) for this lane:
    //   d_lane + ((4 h + a) 2 + (b & 1)) XD_HP XF_PS + (b >> 1) XF_PS
    const int d_lane = (4 * tyl * XD_HP + txl) * XF_PS + 8 * kh;
    constexpr int ROW_F = 2 * XD_HP * XF_PS, HALF_F = 4 * ROW_F;             // floats between patch rows / between the tile halves
    const int rowb1 = d_lane + a1 * ROW_F, rowb2 = d_lane + a2 * ROW_F;
    auto coff = [](int b) { return ((b & 1) * XD_HP + (b >> 1)) * XF_PS; };
    int ad_a1, ad_a2, ad_b1, ad_b2, ad_s1, ad_s2;                            // float indices of the pass's three columns, patch rows a1 / a2
    auto set_columns = [&](int pass) {
        const int ca = pass ? coff(2) : coff(0), cbb = pass ? coff(3) : coff(1), cs = pass ? coff(1) : coff(2);
        ad_a1 = rowb1 + ca; ad_a2 = rowb2 + ca; ad_b1 = rowb1 + cbb; ad_b2 = rowb2 + cbb; ad_s1 = rowb1 + cs; ad_s2 = rowb2 + cs;
    };
    f32x4 xu0, xu1, xw0, xw1, yu0, yu1, yw0, yw1;                            // staging X (columns a, b), Y (column s): rows a1 (u) / a2 (w), channel halves 0 / 1
    float ra[8], rb[8], rs[8];
    float m1[2];                                                             // the jj = 1 multiplier: +2^e (pass 0), -2^e (pass 1)
#define XP_RD(DST, HB, AD, H_, HALF) DST = *reinterpret_cast<const f32x4*>((HB) + (AD) + (H_) * HALF_F + 4 * (HALF))
    auto fma4 = [&](float* dst, const f32x4& w, const f32x4& u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) dst[e] = __builtin_fmaf(w[e], rsgn, u[e]);
    };
    auto mul4 = [&](float* v, float m) {                                     // (volatile asm: the compiler otherwise sinks the multiply to its first use)
#pragma unroll
        for (int e = 0; e < 4; ++e) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(m));
    };
    unsigned vhi[2][4], vlo[2][4];                                           // [slot][channel pair]
    float tt[8];
    auto adds_jj0 = [&](int hf, int q0) {                                    // tt[q0 .. q0 + 3] = fma(ra, 2^e, -rs)
#pragma unroll
        for (int q = q0; q < q0 + 4; ++q) tt[q] = __builtin_fmaf(ra[q], mult[hf], -rs[q]);
    };
    auto adds_jj1 = [&](int hf, int q0) {                                    // tt[q0 .. q0 + 3] = fma(rb, +-2^e, rs)
#pragma unroll
        for (int q = q0; q < q0 + 4; ++q) tt[q] = __builtin_fmaf(rb[q], m1[hf], rs[q]);
    };
    auto v_hi = [&](int slot, int e2) {                                      // channel pairs e2, e2 + 1 from tt[2 e2 .. 2 e2 + 3]
        unsigned ha, hb;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ha) : "v"(tt[2 * e2]), "v"(tt[2 * e2 + 1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(tt[2 * e2 + 2]), "v"(tt[2 * e2 + 3]));
        vhi[slot][e2] = ha;
        vhi[slot][e2 + 1] = hb;
    };
    // lo = fp16(ts - hi) (exact difference, one rounding): the low halves of two registers, then (one MFMA later) their high halves
    auto v_lo_a = [&](int slot, int e2) {
        unsigned la, lb;
        asm("v_fma_mixlo_f16 %0, %2, 1.0, -%4 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %1, %3, 1.0, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]"
            : "=&v"(la), "=&v"(lb) : "v"(tt[2 * e2]), "v"(tt[2 * e2 + 2]), "v"(vhi[slot][e2]), "v"(vhi[slot][e2 + 1]));
        vlo[slot][e2] = la;
        vlo[slot][e2 + 1] = lb;
    };
    auto v_lo_b = [&](int slot, int e2) {
        asm("v_fma_mixhi_f16 %0, %2, 1.0, -%4 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %3, 1.0, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "+v"(vlo[slot][e2]), "+v"(vlo[slot][e2 + 1]) : "v"(tt[2 * e2 + 1]), "v"(tt[2 * e2 + 3]), "v"(vhi[slot][e2]), "v"(vhi[slot][e2 + 1]));
    };
    auto frag = [&](const unsigned (&q)[4]) { return __builtin_bit_cast(xf_f16x8, uint4{q[0], q[1], q[2], q[3]}); };
#define XP_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XP_IF(BIT, STMT) do { if (!(XD_ABLATE & (BIT))) { STMT; } } while (0)

    // One chunk of one pass = four steps S = (h, jj) of TWELVE MFMAs: g = 0..3 filter lo x V hi, 4..7 filter hi x V hi, 8..11 filter hi x V lo
    // (channel tile g & 3).  Vector work per step, one slice per MFMA gap (sched_barrier after every slice), ~200 VALU cycles per step:
    //   S even, E(h) = (h, 0), forms V(h, 1):  rb(h) | adds | hi | rs(h') fma, mul (rs(h) is dead after the adds) | lo of pairs 0, 1
    //   S odd,  O(h) = (h, 1), forms V(h', 0): lo of pairs 2, 3 of ITS OWN operand (needed from g = 8) | ra(h') | adds | hi | lo
    //   with h' the next half (S = 3: half 0 of the next chunk, out of the next ring buffer).
    // Patch reads: E(h) reads columns s(h') (gaps 0-3, consumed from gap 6) and a(h') (gaps 4-7, consumed in O(h)); O(h) reads column b(h')
    // (gaps 4-7, consumed in E(h')).  Filter pieces of the next chunk: eight in S = 0 (jj = 0's), eight in S = 1 (jj = 1's) -- a full chunk
    // ahead of their use; halo(c + 3): S = 2, 3, after the block barrier at the end of S = 1 (the last read of halo(c) is O(h0)'s).
    auto step = [&](int un0, int un1, int hso, float* hcur, float* hnxt, auto PAR, auto S_, auto FIRST) {
        constexpr int par = decltype(PAR)::value, S = decltype(S_)::value;
        constexpr bool first = decltype(FIRST)::value != 0;
        const f32x16 xp_zero16 = {};
        constexpr int h = S >> 1, jj = S & 1, slot = S & 1, nslot = slot ^ 1, nh = h ^ 1;
        const float* rsrc = S < 2 ? hcur : hnxt;                             // the buffer of half h' = nh
        const xf_f16x8 vh = frag(vhi[slot]);
        auto mf = [&](auto G) {
            constexpr int g = decltype(G)::value, ct = g & 3;
            if (XD_ABLATE & 16) return;
            if (g < 4) {
                if (first) acc[h][jj][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[par][jj][ct][1], vh, xp_zero16, 0, 0, 0);
                else acc[h][jj][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[par][jj][ct][1], vh, acc[h][jj][ct], 0, 0, 0);
            } else if (g < 8) {
                acc[h][jj][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[par][jj][ct][0], vh, acc[h][jj][ct], 0, 0, 0);
            } else {
                acc[h][jj][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[par][jj][ct][0], frag(vlo[slot]), acc[h][jj][ct], 0, 0, 0);
            }
        };
        // the memory instruction of gap g: S = 0, 1 one filter piece of the next chunk in gaps 2 .. 9; S = 2: halo pieces 0-3, S = 3: 4-6
        auto mem = [&](auto G) {
            constexpr int g = decltype(G)::value;
            if constexpr (S < 2) {
                if constexpr (g >= 2 && g < 10) load_piece(S == 0 ? un0 : un1, XdInt<par ^ 1>{}, XdInt<S & 1>{}, XdInt<((g - 2) >> 1)>{}, XdInt<((g - 2) & 1)>{});
            } else if constexpr (S == 2) {
                if (g == 8) dma_halo1(hcur, hso, XdInt<0>{});
                if (g == 9) dma_halo1(hcur, hso, XdInt<1>{});
                if (g == 10) dma_halo1(hcur, hso, XdInt<2>{});
                if (g == 11) dma_halo1(hcur, hso, XdInt<3>{});
            } else {
                if (g == 0) dma_halo1(hcur, hso, XdInt<4>{});
                if (g == 2) dma_halo1(hcur, hso, XdInt<5>{});
                if (g == 8) dma_halo1(hcur, hso, XdInt<6>{});
            }
        };
        if (jj == 0) {
            mf(XdInt<0>{}); XP_IF(2, fma4(rb, xw0, xu0)); XP_IF(2, XP_RD(yu0, rsrc, ad_s1, nh, 0)); mem(XdInt<0>{}); XP_FENCE();
            mf(XdInt<1>{}); XP_IF(2, fma4(rb + 4, xw1, xu1)); XP_IF(2, XP_RD(yw0, rsrc, ad_s2, nh, 0)); mem(XdInt<1>{}); XP_FENCE();
            mf(XdInt<2>{}); XP_IF(1, adds_jj1(h, 0)); XP_IF(2, XP_RD(yu1, rsrc, ad_s1, nh, 1)); mem(XdInt<2>{}); XP_FENCE();
            mf(XdInt<3>{}); XP_IF(1, adds_jj1(h, 4)); XP_IF(2, XP_RD(yw1, rsrc, ad_s2, nh, 1)); mem(XdInt<3>{}); XP_FENCE();
            mf(XdInt<4>{}); XP_IF(1, v_hi(nslot, 0)); XP_IF(2, XP_RD(xu0, rsrc, ad_a1, nh, 0)); mem(XdInt<4>{}); XP_FENCE();
            mf(XdInt<5>{}); XP_IF(1, v_hi(nslot, 2)); XP_IF(2, XP_RD(xw0, rsrc, ad_a2, nh, 0)); mem(XdInt<5>{}); XP_FENCE();
            mf(XdInt<6>{}); XP_IF(2, fma4(rs, yw0, yu0)); XP_IF(2, XP_RD(xu1, rsrc, ad_a1, nh, 1)); mem(XdInt<6>{}); XP_FENCE();
            mf(XdInt<7>{}); XP_IF(2, mul4(rs, mult[nh])); XP_IF(2, XP_RD(xw1, rsrc, ad_a2, nh, 1)); mem(XdInt<7>{}); XP_FENCE();
            mf(XdInt<8>{}); XP_IF(2, fma4(rs + 4, yw1, yu1)); mem(XdInt<8>{}); XP_FENCE();
            mf(XdInt<9>{}); XP_IF(2, mul4(rs + 4, mult[nh])); mem(XdInt<9>{}); XP_FENCE();
            mf(XdInt<10>{}); XP_IF(1, v_lo_a(nslot, 0)); mem(XdInt<10>{}); XP_FENCE();
            mf(XdInt<11>{}); XP_IF(1, v_lo_b(nslot, 0)); mem(XdInt<11>{}); XP_FENCE();
        } else {
            mf(XdInt<0>{}); XP_IF(1, v_lo_a(slot, 2)); mem(XdInt<0>{}); XP_FENCE();
            mf(XdInt<1>{}); XP_IF(1, v_lo_b(slot, 2)); mem(XdInt<1>{}); XP_FENCE();
            mf(XdInt<2>{}); XP_IF(2, fma4(ra, xw0, xu0)); mem(XdInt<2>{}); XP_FENCE();
            mf(XdInt<3>{}); XP_IF(2, fma4(ra + 4, xw1, xu1)); mem(XdInt<3>{}); XP_FENCE();
            mf(XdInt<4>{}); XP_IF(1, adds_jj0(nh, 0)); XP_IF(2, XP_RD(xu0, rsrc, ad_b1, nh, 0)); mem(XdInt<4>{}); XP_FENCE();
            mf(XdInt<5>{}); XP_IF(1, adds_jj0(nh, 4)); XP_IF(2, XP_RD(xw0, rsrc, ad_b2, nh, 0)); mem(XdInt<5>{}); XP_FENCE();
            mf(XdInt<6>{}); XP_IF(1, v_hi(nslot, 0)); XP_IF(2, XP_RD(xu1, rsrc, ad_b1, nh, 1)); mem(XdInt<6>{}); XP_FENCE();
            mf(XdInt<7>{}); XP_IF(1, v_hi(nslot, 2)); XP_IF(2, XP_RD(xw1, rsrc, ad_b2, nh, 1)); mem(XdInt<7>{}); XP_FENCE();
            mf(XdInt<8>{}); XP_IF(1, v_lo_a(nslot, 0)); mem(XdInt<8>{}); XP_FENCE();
            mf(XdInt<9>{}); XP_IF(1, v_lo_b(nslot, 0)); mem(XdInt<9>{}); XP_FENCE();
            mf(XdInt<10>{}); XP_IF(1, v_lo_a(nslot, 2)); mem(XdInt<10>{}); XP_FENCE();
            mf(XdInt<11>{}); XP_IF(1, v_lo_b(nslot, 2)); mem(XdInt<11>{}); XP_FENCE();
        }
        if (S == 1) {
            // halo(c + 1) has landed (its DMA left two chunks ago; vector memory operations complete in issue order, and at most the 7 pieces
            // of halo(c + 2) and this chunk's 16 filter pieces are younger), then the block barrier: halo(c + 1) visible, halo(c)'s buffer spent
            asm volatile("s_waitcnt vmcnt(23)" ::: "memory");
            xd_lds_barrier();
            XP_FENCE();
        }
    };
    float *hcur = hbuf0, *hnxt = hbuf0 + XD_HBUF_FLOATS, *hthird = hbuf0 + 2 * XD_HBUF_FLOATS;
    // un: scalar offset of the NEXT chunk's pieces of (row, jj = 0) -- jj = 1 is pos_stride further; hso: byte offset of chunk c + 3 in a pixel;
    // next_pass >= 0: the chunk is the last of its pass -- its steps 2, 3 read the first operand columns of pass `next_pass`
    auto chunk = [&](int un, int hso, int next_pass, auto PAR, auto FIRST) {
        step(un, un + pos_stride, hso, hcur, hnxt, PAR, XdInt<0>{}, FIRST);
        step(un, un + pos_stride, hso, hcur, hnxt, PAR, XdInt<1>{}, FIRST);
        if (next_pass >= 0) set_columns(next_pass);
        step(un, un + pos_stride, hso, hcur, hnxt, PAR, XdInt<2>{}, FIRST);
        step(un, un + pos_stride, hso, hcur, hnxt, PAR, XdInt<3>{}, FIRST);
        float* const t = hcur; hcur = hnxt; hnxt = hthird; hthird = t;
    };

    // ---- prologue (wino_x3d_kernel's: ONE round trip to memory) ------------------------------------------------------------------------------
    float* const sc_lds = reinterpret_cast<float*>(smem_xf + XP_SC_OFFSET);
    // the block's 16 x 128 filter scales (two DMA instructions: 32 lanes x 16 bytes per position) and 128 biases
    const float* const uinv0 = reinterpret_cast<const float*>(ublob + (size_t)16 * K16 * u_rbt * HX_RB) + (size_t)(tid >> 5) * (u_rbt * 32) + XP_CB * cb + (tid & 31) * 4;
    int cm_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned P = (unsigned)(tid + 256 * q);
        const unsigned hr = __umul24(P, 1928u) >> 16, hc = P - (unsigned)XF_HC * hr;              // P / 34, P % 34
        const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
        const bool inb = hr < (unsigned)X3_HR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cm_src[q] = inb ? (int)((__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * 4u) : (int)0xFFFFFFF0u;
    }
    XP_FENCE();
#pragma unroll
    for (int q = 0; q < 2; ++q)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(mrs, (xd_lds_ptr)(smem_xf + XP_CM_OFFSET + (q * 4 + wave_u) * 256), 4, cm_src[q], 0, 0, 0);
    {
        const int so = ub_row;                                                // pass 0, chunk 0
        load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<0>{}, XdInt<0>{}); load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<0>{}, XdInt<1>{});
        load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<1>{}, XdInt<0>{}); load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<1>{}, XdInt<1>{});
        load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<2>{}, XdInt<0>{}); load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<2>{}, XdInt<1>{});
        load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<3>{}, XdInt<0>{}); load_piece(so, XdInt<0>{}, XdInt<0>{}, XdInt<3>{}, XdInt<1>{});
        const int s1 = ub_row + pos_stride;
        load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<0>{}, XdInt<0>{}); load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<0>{}, XdInt<1>{});
        load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<1>{}, XdInt<0>{}); load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<1>{}, XdInt<1>{});
        load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<2>{}, XdInt<0>{}); load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<2>{}, XdInt<1>{});
        load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<3>{}, XdInt<0>{}); load_piece(s1, XdInt<0>{}, XdInt<1>{}, XdInt<3>{}, XdInt<1>{});
    }
    __builtin_amdgcn_global_load_lds(uinv0, (xd_lds_ptr)(smem_xf + XP_SC_OFFSET + wave_u * 1024), 16, 0, 0);
    __builtin_amdgcn_global_load_lds(uinv0 + (size_t)8 * (u_rbt * 32), (xd_lds_ptr)(smem_xf + XP_SC_OFFSET + 4096 + wave_u * 1024), 16, 0, 0);
    if (wave_u == 0 && lane < 32) __builtin_amdgcn_global_load_lds(bias + XP_CB * cb + 4 * lane, (xd_lds_ptr)(smem_xf + XP_SC_OFFSET + 16 * XP_CB * 4), 16, 0, 0);
    XP_FENCE();
    halo_sources();
    XP_FENCE();
    dma_halo(hcur, 0);
    dma_halo(hnxt, (K16 > 1 ? 1 : 0) * 64);
    dma_halo(hthird, (K16 > 2 ? 2 : K16 - 1) * 64);
    XP_FENCE();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_issued = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XD_NDMA) : "memory");       // everything but halo(1) and halo(2)
    xd_lds_barrier();
    {   // the lane's two tile scales from the halo pixels' channel maxima (rows 4 h + 2 tyl + a, columns 2 txl + c)
        const float* cm = reinterpret_cast<const float*>(smem_xf + XP_CM_OFFSET) + (2 * tyl) * XF_HC + 2 * txl;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float dmax = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const xd_f32x2 u = *reinterpret_cast<const xd_f32x2*>(cm + (4 * h + a) * XF_HC);
                const xd_f32x2 v = *reinterpret_cast<const xd_f32x2*>(cm + (4 * h + a) * XF_HC + 2);
                dmax = fmaxf(fmaxf(dmax, fmaxf(u[0], u[1])), fmaxf(v[0], v[1]));
            }
            hx_row_scale(4.0f * dmax, mult[h], vinv[h]);
        }
    }
#ifdef XD_CLOCKS
    const unsigned long long xd_t_landed = __builtin_amdgcn_s_memrealtime();
#endif
    // loop entry state (top of E(h0) of chunk 0, pass 0): V(h0, jj 0) formed in slot 0, rs(h0) made, column b(h0) on its way into X
    set_columns(0);
    m1[0] = mult[0]; m1[1] = mult[1];
    XP_RD(xu0, hcur, ad_a1, 0, 0); XP_RD(xw0, hcur, ad_a2, 0, 0); XP_RD(xu1, hcur, ad_a1, 0, 1); XP_RD(xw1, hcur, ad_a2, 0, 1);
    XP_RD(yu0, hcur, ad_s1, 0, 0); XP_RD(yw0, hcur, ad_s2, 0, 0); XP_RD(yu1, hcur, ad_s1, 0, 1); XP_RD(yw1, hcur, ad_s2, 0, 1);
    fma4(ra, xw0, xu0); fma4(ra + 4, xw1, xu1);
    fma4(rs, yw0, yu0); fma4(rs + 4, yw1, yu1); mul4(rs, mult[0]); mul4(rs + 4, mult[0]);
    XP_RD(xu0, hcur, ad_b1, 0, 0); XP_RD(xw0, hcur, ad_b2, 0, 0); XP_RD(xu1, hcur, ad_b1, 0, 1); XP_RD(xw1, hcur, ad_b2, 0, 1);
    adds_jj0(0, 0); adds_jj0(0, 4);
    v_hi(0, 0); v_hi(0, 2); v_lo_a(0, 0); v_lo_b(0, 0); v_lo_a(0, 2); v_lo_b(0, 2);
#ifdef XD_CLOCKS
    const unsigned long long xd_t_loop = __builtin_amdgcn_s_memrealtime(), xd_c_loop = __builtin_readcyclecounter();
    unsigned long long xd_t_spill0 = 0, xd_t_spill1 = 0;
#endif

#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        // scalar offset of (this pass, chunk c, jj = 0): pb + c chunk_stride; the chunk after the pass's last one: pass 0 -> (pass 1, chunk 0),
        // pass 1 -> the last chunk again (nobody consumes those loads)
        const int pb = ub_row + 2 * pass * pos_stride;
        const int after = pass == 0 ? ub_row + 2 * pos_stride : pb + (K16 - 1) * chunk_stride;
        // halo(c + 3): pass 0 wraps around into the next pass's first chunks, pass 1 re-reads its last chunk
        auto hso_of = [&](int c) { const int n = c + 3; return (n < K16 ? n : (pass == 0 ? n - K16 : K16 - 1)) * 64; };
        chunk(pb + chunk_stride, hso_of(0), -1, XdInt<0>{}, XdInt<1>{});
        chunk(pb + 2 * chunk_stride, hso_of(1), -1, XdInt<1>{}, XdInt<0>{});
        for (int c = 2; c < K16; c += 2) {                                   // K16 is even and >= 4 (checked by the launcher)
            chunk(pb + (c + 1) * chunk_stride, hso_of(c), -1, XdInt<0>{}, XdInt<0>{});
            const bool last = c + 2 >= K16;
            chunk(last ? after : pb + (c + 2) * chunk_stride, hso_of(c + 1), last && pass == 0 ? 1 : -1, XdInt<1>{}, XdInt<0>{});
        }
        if (pass == 0) {
#ifdef XD_CLOCKS
            xd_t_spill0 = __builtin_amdgcn_s_memrealtime();
#endif
            // the pass's 256 accumulators -> the block's scratch, lane-private (k = ((h 2 + jj) 4 + ct) 4 + register quad)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 v = {acc[h][jj][ct][4 * g], acc[h][jj][ct][4 * g + 1], acc[h][jj][ct][4 * g + 2], acc[h][jj][ct][4 * g + 3]};
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xp_u32x4, v), srs, tid16, 4096 * ((((h * 2 + jj) * 4 + ct) * 4) + g), 0);
                        }
            m1[0] = -mult[0]; m1[1] = -mult[1];
#ifdef XD_CLOCKS
            xd_t_spill1 = __builtin_amdgcn_s_memrealtime();
#endif
        }
    }
#undef XP_RD
#undef XP_IF
#ifdef XD_CLOCKS
    const unsigned long long xd_t_done = __builtin_amdgcn_s_memrealtime(), xd_c_done = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the clamped re-loads of the last chunks have landed (and the spill is out) ...
    __syncthreads();                                                         // ... and every wave is past its last halo read: the Y buffer may overwrite the ring
#ifdef XD_CLOCKS
    const unsigned long long xd_t_e0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long xd_t_e1 = 0;
#endif

    // ---- epilogue: A^T M A + bias + ReLU (+ 2x2 max-pool), wino_x3d_kernel's order of operations, in two rounds of 64 output channels.
    // The wave owns position row i = wave: columns j = 0, 1 come back from the scratch (this lane's own stores), j = 2, 3 are the accumulators.
    float* const ybuf = reinterpret_cast<float*>(smem_xf);                   // [half 2][row i 4][b 2][tile 32][68]
    const int Ho = H >> 1, Wo = W >> 1;
    auto round = [&](auto RD) {
        constexpr int rd = decltype(RD)::value;
        f32x4 fill[2][4][2][2];                                              // [ct2][g][h][jj]
#pragma unroll
        for (int ct2 = 0; ct2 < 2; ++ct2)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        fill[ct2][g][h][jj] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, tid16, 4096 * ((((h * 2 + jj) * 4 + 2 * rd + ct2) * 4) + g), 0));
        if (rd == 1) __syncthreads();                                        // round 0's row pass has read the Y buffer
#pragma unroll
        for (int ct2 = 0; ct2 < 2; ++ct2)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int col = 8 * g + 4 * kh;                              // channel within the 32-channel tile
                const int co = 64 * rd + 32 * ct2 + col;                    // channel within the block's 128
                f32x4 sb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) sb[j] = *reinterpret_cast<const f32x4*>(sc_lds + (4 * wave + j) * XP_CB + co);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 m[4];
                    m[0] = fill[ct2][g][h][0] * sb[0];
                    m[1] = fill[ct2][g][h][1] * sb[1];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int ct = 2 * rd + ct2;
                        const f32x4 a = {acc[h][jj][ct][4 * g], acc[h][jj][ct][4 * g + 1], acc[h][jj][ct][4 * g + 2], acc[h][jj][ct][4 * g + 3]};
                        m[2 + jj] = a * sb[2 + jj];
                    }
                    const f32x4 vi = {vinv[h], vinv[h], vinv[h], vinv[h]};
                    const f32x4 y0 = ((m[0] + m[1]) + m[2]) * vi;
                    const f32x4 y1 = xd_sub4(xd_sub4(m[1], m[2]), m[3]) * vi;
                    float* dst = ybuf + ((((h * 4 + wave) * 2) * 32 + tl) * XD_MS) + 32 * ct2 + col;
                    *reinterpret_cast<f32x4*>(dst) = y0;
                    *reinterpret_cast<f32x4*>(dst + 32 * XD_MS) = y1;
                }
            }
        __syncthreads();
#ifdef XD_CLOCKS
        if (rd == 0) xd_t_e1 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int item = tid + 256 * it;                                 // (tile of 64, channel quad of 16)
            const int t = item >> 4, k = (item & 15) * 4;
            const int h = t >> 5, tt_ = t & 31;
            int sty, stx;
            xd_slot_tile(tt_, sty, stx);
            const int oty = 4 * by + 2 * h + sty, otx = XF_TC * bx + stx;
            const bool live = oty < gm.th && otx < gm.tw && !(POOL && (oty >= Ho || otx >= Wo));
            if (!live && !cmax_out) continue;
            const int kg = XP_CB * cb + 64 * rd + k;
            const float* yp = ybuf + ((h * 4) * 2 * 32 + tt_) * XD_MS + k;  // + (i 2 + b) 32 XD_MS
            f32x4 Y[4][2];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) Y[i][bb] = *reinterpret_cast<const f32x4*>(yp + (i * 2 + bb) * (32 * XD_MS));
            const f32x4 bv = *reinterpret_cast<const f32x4*>(sc_lds + 16 * XP_CB + 64 * rd + k);
            f32x4 o[2][2];
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) {
                o[0][bb] = ((Y[0][bb] + Y[1][bb]) + Y[2][bb]) + bv;
                o[1][bb] = xd_sub4(xd_sub4(Y[1][bb], Y[2][bb]), Y[3][bb]) + bv;
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
                f32x4 mx;
#pragma unroll
                for (int e = 0; e < 4; ++e) mx[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
                if (live) *reinterpret_cast<f32x4*>(y + ((size_t)oty * Wo + otx) * Cout + kg) = mx;
                if (cmax_out) {
                    const float pm = xd_rowmax16(fmaxf(fmaxf(mx[0], mx[1]), fmaxf(mx[2], mx[3])));
                    if (live && (item & 15) == 0) atomicMax(reinterpret_cast<unsigned*>(cmax_out + (size_t)oty * Wo + otx), __float_as_uint(pm));
                }
            } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int yy = 2 * oty + a;
#pragma unroll
                    for (int bb = 0; bb < 2; ++bb) {
                        const int xx = 2 * otx + bb;
                        const bool ok = live && yy < H && xx < W;
                        if (ok) *reinterpret_cast<f32x4*>(y + ((size_t)yy * W + xx) * Cout + kg) = o[a][bb];
                        if (cmax_out) {
                            const float pm = xd_rowmax16(fmaxf(fmaxf(o[a][bb][0], o[a][bb][1]), fmaxf(o[a][bb][2], o[a][bb][3])));
                            if (ok && (item & 15) == 0) atomicMax(reinterpret_cast<unsigned*>(cmax_out + (size_t)yy * W + xx), __float_as_uint(pm));
                        }
                    }
                }
            }
        }
    };
    round(XdInt<0>{});
    round(XdInt<1>{});
#ifdef XD_CLOCKS
    // timing build (tools/xd_clocks.py): wave 0 / lane 0 of every block leaves its stamps behind the (single-map) output
    if (tid == 0) {
        const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();
        float* rec = y_maps + (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout + (size_t)blockIdx.x * 16;
        rec[0] = (float)(xd_t_loop - xd_t_in); rec[1] = (float)(xd_t_done - xd_t_loop); rec[2] = (float)(t_out - xd_t_done);
        rec[3] = (float)(xd_c_done - xd_c_loop); rec[4] = (float)(xd_t_in & 0xFFFFFF); rec[5] = (float)(t_out & 0xFFFFFF);
        rec[6] = (float)(2 * K16); rec[7] = 1.0f;
        rec[8] = (float)(xd_t_issued - xd_t_in); rec[9] = (float)(xd_t_landed - xd_t_issued); rec[10] = (float)(xd_t_loop - xd_t_landed);
        rec[11] = (float)(xd_t_e0 - xd_t_done); rec[12] = (float)(xd_t_e1 - xd_t_e0); rec[13] = (float)(t_out - xd_t_e1);
        rec[14] = (float)(xd_t_spill1 - xd_t_spill0); rec[15] = 0.f;
    }
#endif
#undef XP_FENCE
}

// ---- host side ------------------------------------------------------------------------------------------------------------
size_t conv3x3_winograd_x3_pair_spill_bytes(int N, int H, int W, int cout)
{
    if (N < 1 || H < 1 || W < 1 || cout < XP_CB) return 0;
    const long long tbx = cdiv(cdiv(W, 2), XF_TC), tby = cdiv(cdiv(H, 2), 4);
    // (the grid may carry up to 7 surplus blocks: xd_block_to_tile's XCD groups)
    return (size_t)(tbx * tby * (cout / XP_CB) * N + 8) * XP_SPILL_FLOATS * sizeof(float);
}

int launch_wino_x3p(bool pool, const float* x, const float* cmax, const unsigned char* ublob, const float* bias, float* y, int N, int H, int W,
                    int cin, int cout, int relu, float* cmax_out, float* spill, size_t spill_bytes, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || cin < 64 || cin % 32 != 0 || cout < XP_CB || cout % XP_CB != 0) return FRCNN_EUNSUPPORTED;
    if ((size_t)H * W * cin >= ((size_t)1 << 29)) return FRCNN_EUNSUPPORTED;
    if (!spill || spill_bytes < conv3x3_winograd_x3_pair_spill_bytes(N, H, W, cout)) return FRCNN_EINVAL;
    XfGeom gm;
    gm.tw = cdiv(W, 2); gm.th = cdiv(H, 2);
    gm.tbx = cdiv(gm.tw, XF_TC); gm.tby = cdiv(gm.th, 4);
    gm.ncb = cout / XP_CB;
    const long long total = (long long)gm.tbx * gm.tby * gm.ncb * N;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    gm.xg = ((long long)gm.tbx * gm.tby * N) % 8 == 0 ? 1 : 0;
    gm.ntb = gm.tbx * gm.tby * N;
    long long grid_blocks = total;
    if (cin >= 256 && (gm.ncb % 8 == 0 || 8 % gm.ncb == 0)) {
        gm.xg = 2;
        if (gm.ncb < 8) grid_blocks = 8LL * cdiv(gm.ntb, 8 / gm.ncb);
    }
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    gm.m_tbx = magic(gm.tbx); gm.m_tby = magic(gm.tby); gm.m_ncb = magic(gm.ncb); gm.m_ntb = magic(gm.ntb);
    gm.g8 = gm.ncb < 8 ? 8 / gm.ncb : 1;
    if (total * std::max(std::max(gm.ncb, gm.ntb), std::max(gm.tbx, gm.tby)) >= 0x100000000ll) return FRCNN_EUNSUPPORTED;
    if ((size_t)grid_blocks * XP_SPILL_FLOATS * sizeof(float) > spill_bytes) return FRCNN_EINVAL;
    const int u_rbt = cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout) / 32;
    if ((size_t)16 * (cin / 16) * u_rbt * HX_RB >= ((size_t)1 << 31)) return FRCNN_EUNSUPPORTED;
    if (pool) {
        auto kern = wino_x3p_kernel<true>;
        FRCNN_MAX_LDS_ONCE(kern, XP_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid_blocks), dim3(256), XP_LDS_BYTES, s, x, cmax, ublob, bias, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out, spill);
    } else {
        auto kern = wino_x3p_kernel<false>;
        FRCNN_MAX_LDS_ONCE(kern, XP_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)grid_blocks), dim3(256), XP_LDS_BYTES, s, x, cmax, ublob, bias, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out, spill);
    }
    return check_launch();
}

}  // namespace frcnn
