// wino_x3e.hip -- the one-launch f32x3 Winograd F(2x2,3x3) layer with TWO waves per SIMD (round 5).  Same layer, same arguments and the same
// results BIT FOR BIT as wino_x3d_kernel (csrc/wino_x3f.hip): every 3x3 convolution + ReLU (+ MaxPool2d) of pytorch/FasterRCNN/models/vgg16.py:77-96
// from conv2_2 on and the RPN trunk (models/rpn.py:88).
//
// Why: wino_x3d_kernel is ONE wave per SIMD (256 accumulator + ~210 other registers per lane), and an in-order wave pays its 48 MFMAs
// (1536 cycles per 16-channel chunk) PLUS its ~375 operand-forming / LDS / load instructions (2800-2850 cycles per chunk measured, DESIGN.md
// section 5).  The register file gives two waves per SIMD 256 registers each (VGPR + AGPR, one allocation per kernel), so no wave can keep 256
// accumulators: the block's 64 tiles x 64 output channels x 16 positions are split over EIGHT waves instead -- wave (i, jp) owns position row
// i and the position columns j = 2 jp, 2 jp + 1 of every tile: 128 accumulator registers -- and the two waves of a SIMD overlap each other's
// MFMAs and vector work (tools/micro/mfma_fill2.hip: 8 fillers per MFMA cost one wave 46 cycles per gap, two waves 43; the kernel's own mix
// with an LDS read per gap 80 against 66).  What the split costs: V(., j) = r[b1] +- r[b2] needs the columns {0, 1, 2} of r = B^T d for
// j in {0, 1} and {1, 2, 3} for j in {2, 3} -- three column reads and row combinations per wave where the four-wave kernel has four for
// twice the positions (+12 % vector instructions per SIMD) -- and the output transform's column pass crosses the wave pair: one vector per
// (tile, channel quad) is exchanged through LDS before the row pass (same summation order: same bits).
// The filter fragments are single-buffered (8 pieces = 32 registers) and re-loaded just in time: the pieces of position column jj are dead
// after the column's second step and needed again two steps later.
#include "wino_x3_shared.h"

namespace frcnn {

static constexpr int XE_NDMA = 4;                                             // halo DMA instructions per thread and chunk: 4 x 512 x 16 B = 32,768 B >= 27,200
static constexpr int XE_HBUF_BYTES = XE_NDMA * 512 * 16;
static constexpr int XE_HBUF_FLOATS = XE_HBUF_BYTES / 4;
static_assert(3 * XE_HBUF_BYTES <= XD_M_BYTES, "the epilogue's buffer overlays the halo ring");

template <bool POOL>
__global__ __launch_bounds__(512, 1)
void wino_x3e_kernel(const float* __restrict__ x_maps, const float* __restrict__ cmax_maps, const unsigned char* __restrict__ ublob,
                     const float* __restrict__ bias, float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int u_rbt, int relu,
                     XfGeom gm, float* __restrict__ cmax_out_maps)
{
#ifdef XD_CLOCKS
    const unsigned long long xd_t_in = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_xe[];
    float* const hbuf0 = reinterpret_cast<float*>(smem_xe);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wi = wave & 3, jp = wave >> 2;                                 // position row i, position column pair (j = 2 jp, 2 jp + 1)
    const int K16 = Cin >> 4;

    int cb, bx, by, map;
    if (!xd_block_to_tile(gm, blockIdx.x, cb, bx, by, map)) return;
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    const float* __restrict__ const cmax = cmax_maps + (size_t)map * H * W;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout;
    float* __restrict__ const cmax_out = cmax_out_maps ? cmax_out_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) : nullptr;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, H * W * Cin * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cmax), 0, H * W * (int)sizeof(float), 0x00020000);

    const int tl = lane & 31, tyl = tl >> 4, txl = tl & 15, kh = lane >> 5;
    float mult[2], vinv[2];

    // ---- halo staging by LDS-DMA (wino_x3d_kernel's layout: slot order [row][column parity][17], 80-byte pixels): piece P = (it 8 + wave) 64 + lane
    const int hy0 = 8 * by - 1, hx0 = 2 * XF_TC * bx - 1;
    int h_src[XE_NDMA];
    auto halo_sources = [&]() {
#pragma unroll
        for (int it = 0; it < XE_NDMA; ++it) {
            const unsigned P = (unsigned)((it * 8 + wave) * 64 + lane);                       // < 2048: the reciprocal constants are exact
            const unsigned slot = __umul24(P, 52429u) >> 18, part = P - 5u * slot;            // P / 5, P % 5
            const unsigned hr = __umul24(slot, 1928u) >> 16, rem = slot - (unsigned)XF_HC * hr;   // slot / 34, slot % 34
            const unsigned par = rem >= (unsigned)XD_HP ? 1u : 0u, hc = 2u * (rem - par * (unsigned)XD_HP) + par;
            const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
            const bool inb = part < 4u && hr < (unsigned)X3_HR && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const unsigned off = (__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * (unsigned)(Cin * 4) + 16u * part;
            h_src[it] = inb ? (int)off : (int)0xFFFFFFF0u;
        }
    };
    auto dma_halo = [&](float* hb, int chunk_off, auto IT0, auto IT1) {
#pragma unroll
        for (int it = decltype(IT0)::value; it < decltype(IT1)::value; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (xd_lds_ptr)(reinterpret_cast<unsigned char*>(hb) + (it * 8 + wave) * 1024), 16, h_src[it],
                                                     chunk_off, 0, 0);
    };

    // ---- filter fragments: piece (position p, row block r, term t) of chunk c = ublob + ((p K16 + c) u_rbt + 2 cb + r) 2 KB + t 1 KB
    xf_f16x8 U[2][2][2];                                                     // [position column jj][output tile ct][0 = hi, 1 = lo]: ONE set
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(ublob), 0, 16 * K16 * u_rbt * HX_RB, 0x00020000);
    const int chunk_stride = u_rbt * HX_RB;
    int ubase[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) ubase[jj] = ((4 * wi + 2 * jp + jj) * K16 * u_rbt + 2 * cb) * HX_RB;
    const int lane16 = lane * 16;
    // both output tiles of term t of position column jj (chunk byte offset co): the constants land in the instruction offset
    auto load_u = [&](int co, auto JJ, auto T) {
        constexpr int jj = decltype(JJ)::value, t = decltype(T)::value;
        const int so = ubase[jj] + co;
        U[jj][0][t] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16 + t * HX_PIECE, so, 0));
        U[jj][1][t] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16 + HX_RB + t * HX_PIECE, so, 0));
    };

    f32x16 acc[2][2][2];                                                     // [tile half h][position column jj][output tile ct]; never zeroed (chunk 0: C = 0)

    // ---- operand formation: position row i = wi: r[b] = d[a1][b] +- d[a2][b]; V[i][j] = r[b1] +- r[b2] (wino_x3d_kernel's operations)
    const int a1 = wi == 0 ? 0 : (wi == 2 ? 2 : 1);
    const int a2 = wi == 0 ? 2 : (wi == 2 ? 1 : (wi == 1 ? 2 : 3));
    const float rsgn = wi != 1 ? -1.0f : 1.0f;
    // The wave's three patch columns: A (the minuend of its first position column), B (its second column's own), S (shared):
    //   jp = 0: V0 = r0 - r2, V1 = r1 + r2: A = 0, B = 1, S = 2;   jp = 1: V2 = r2 - r1, V3 = r1 - r3: A = 2, B = 3, S = 1
    //   first = A - S;  second = S + sB B  (sB = +1 / -1: an FMA by +-1 rounds like the add / subtract it replaces)
    const int cA = jp ? 2 : 0, cB = jp ? 3 : 1, cS = jp ? 1 : 2;
    const float sB = jp ? -1.0f : 1.0f;
    auto col_off = [&](int b) { return ((b & 1) * XD_HP + (b >> 1)) * XF_PS; };
    const int d_lane = (4 * tyl * XD_HP + txl) * XF_PS + 8 * kh;
    const int r1off = a1 * 2 * XD_HP * XF_PS, r2off = a2 * 2 * XD_HP * XF_PS;   // (wave-uniform: scalar registers)
    const int oA = col_off(cA), oB = col_off(cB), oS = col_off(cS);
    constexpr int H_OFF = 4 * 2 * XD_HP * XF_PS;                               // tile half h: four halo rows further

    float rS[2][8], rA[8], rB[8];                                            // the live columns of r: S of both halves, the current A and B
    struct Pend { f32x4 u, w; };                                             // one half-column (4 channels) of two patch rows on its way from LDS
    auto rd = [&](const float* hb, int col, int h, int half) {
        Pend p;
        const float* q = hb + d_lane + col + h * H_OFF + 4 * half;
        p.u = *reinterpret_cast<const f32x4*>(q + r1off);
        p.w = *reinterpret_cast<const f32x4*>(q + r2off);
        return p;
    };
    auto mk = [&](float (&r)[8], const Pend& p, int half) {                  // u + sgn w as ONE fused operation == u +- w rounded once
#pragma unroll
        for (int e = 0; e < 4; ++e) r[4 * half + e] = __builtin_fmaf(p.w[e], rsgn, p.u[e]);
    };
    unsigned vhi[2][4], vlo[2][4];
    float tt[4];
    auto t_first = [&](const float (&A)[8], const float (&S)[8], int e2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tt[q] = A[2 * e2 + q] - S[2 * e2 + q];
    };
    auto t_second = [&](const float (&B)[8], const float (&S)[8], int e2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) tt[q] = __builtin_fmaf(B[2 * e2 + q], sB, S[2 * e2 + q]);
    };
    auto v_hi = [&](int h, int slot, int e2) {
        // ONE asm statement per pair of registers: hipcc pads a v_fma_mixhi that follows inline-asm partial writes with an s_nop it cannot
        // prove unnecessary (an issue slot like any other); the other register's instruction between a register's two halves is the wait state
        unsigned ha, hb;
        asm("v_fma_mixlo_f16 %0, %2, %6, 0\n\tv_fma_mixlo_f16 %1, %4, %6, 0\n\tv_fma_mixhi_f16 %0, %3, %6, 0\n\tv_fma_mixhi_f16 %1, %5, %6, 0"
            : "=&v"(ha), "=&v"(hb) : "v"(tt[0]), "v"(tt[1]), "v"(tt[2]), "v"(tt[3]), "v"(mult[h]));
        vhi[slot][e2] = ha;
        vhi[slot][e2 + 1] = hb;
    };
    auto v_lo = [&](int h, int slot, int e2) {
        unsigned la, lb;
        asm("v_fma_mixlo_f16 %0, %2, %6, -%7 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %1, %4, %6, -%8 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %3, %6, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %6, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(la), "=&v"(lb) : "v"(tt[0]), "v"(tt[1]), "v"(tt[2]), "v"(tt[3]), "v"(mult[h]), "v"(vhi[slot][e2]), "v"(vhi[slot][e2 + 1]));
        vlo[slot][e2] = la;
        vlo[slot][e2 + 1] = lb;
    };
    auto frag = [&](const unsigned (&q)[4]) { return __builtin_bit_cast(xf_f16x8, uint4{q[0], q[1], q[2], q[3]}); };
#define XE_MFMA(H_, JJ, CT, UT, VV) \
    do { acc[H_][JJ][CT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[JJ][CT][UT], VV, acc[H_][JJ][CT], 0, 0, 0); } while (0)
#define XE_MFMA0(H_, JJ, CT, UT, VV) \
    do { if (first) acc[H_][JJ][CT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[JJ][CT][UT], VV, xe_zero16, 0, 0, 0); else XE_MFMA(H_, JJ, CT, UT, VV); } while (0)
#define XE_FENCE() __builtin_amdgcn_sched_barrier(0)

    // ---- one chunk = four steps (jj, h) of six MFMAs: (0, 0), (0, 1), (1, 0), (1, 1); per accumulator filter lo x V hi, filter hi x V hi,
    // filter hi x V lo (wino_x3d_kernel's order).  Step s forms the operand of step s + 1 in the other slot; the columns of r it needs
    // were read one step earlier:
    //   step 0 forms V(0, 1) = A1 - S1          reads B0                      loads U[1] hi of this chunk
    //   step 1 forms V(1, 0) = S0 +- B0         reads B1                      loads U[0] lo of the next chunk; barrier
    //   step 2 forms V(1, 1) = S1 +- B1         reads A0', S0' (next chunk)   loads U[0] hi of the next chunk, halo DMA
    //   step 3 forms V(0, 0)' = A0' - S0'       reads A1', S1'                loads U[1] lo of the next chunk, halo DMA
    float *hcur = hbuf0, *hnxt = hbuf0 + XE_HBUF_FLOATS, *hthird = hbuf0 + 2 * XE_HBUF_FLOATS;
    Pend pX, pY, pZ;                                                         // half-columns in flight (at most three)
    auto chunk = [&](int c, auto FIRST) {
        constexpr bool first = decltype(FIRST)::value != 0;
        const f32x16 xe_zero16 = {};
        const int cu_this = c * chunk_stride, cu_next = (c + 1 < K16 ? c + 1 : K16 - 1) * chunk_stride, hso = (c + 3 < K16 ? c + 3 : K16 - 1) * 64;
        {   // ---- step 0: MFMAs (jj 0, h 0) from slot 0; forms V(jj 0, h 1) = A1 - S1 into slot 1; column B0
            const xf_f16x8 vh = frag(vhi[0]), vl = frag(vlo[0]);
            XE_MFMA0(0, 0, 0, 1, vh);
            mk(rS[1], pY, 0);                                                // S1 lo (read in the previous step 3, gap 5)
            t_first(rA, rS[1], 0);
            pX = rd(hcur, oB, 0, 0);                                         // B0 lo
            XE_FENCE();
            XE_MFMA0(0, 0, 1, 1, vh);
            v_hi(1, 1, 0);
            mk(rS[1], pZ, 1);                                                // S1 hi (previous step 3, gap 6)
            load_u(cu_this, XdInt<1>{}, XdInt<0>{});                         // U[1] hi of this chunk (its last reader: the previous step 3)
            XE_FENCE();
            XE_MFMA(0, 0, 0, 0, vh);
            v_lo(1, 1, 0);
            pY = rd(hcur, oB, 0, 1);                                         // B0 hi
            XE_FENCE();
            XE_MFMA(0, 0, 1, 0, vh);
            t_first(rA, rS[1], 2);
            v_hi(1, 1, 2);
            XE_FENCE();
            XE_MFMA(0, 0, 0, 0, vl);
            v_lo(1, 1, 2);
            XE_FENCE();
            XE_MFMA(0, 0, 1, 0, vl);
            mk(rB, pX, 0);                                                   // B0 lo
            XE_FENCE();
        }
        {   // ---- step 1: MFMAs (jj 0, h 1) from slot 1; forms V(jj 1, h 0) = S0 +- B0 into slot 0; column B1; the chunk's barrier
            const xf_f16x8 vh = frag(vhi[1]), vl = frag(vlo[1]);
            XE_MFMA0(1, 0, 0, 1, vh);
            t_second(rB, rS[0], 0);
            pX = rd(hcur, oB, 1, 0);                                         // B1 lo
            XE_FENCE();
            XE_MFMA0(1, 0, 1, 1, vh);
            v_hi(0, 0, 0);
            mk(rB, pY, 1);                                                   // B0 hi
            load_u(cu_next, XdInt<0>{}, XdInt<1>{});                         // U[0] lo of the next chunk (dead since the MFMA above)
            XE_FENCE();
            XE_MFMA(1, 0, 0, 0, vh);
            v_lo(0, 0, 0);
            pY = rd(hcur, oB, 1, 1);                                         // B1 hi
            XE_FENCE();
            XE_MFMA(1, 0, 1, 0, vh);
            t_second(rB, rS[0], 2);
            v_hi(0, 0, 2);
            XE_FENCE();
            XE_MFMA(1, 0, 0, 0, vl);
            v_lo(0, 0, 2);
            XE_FENCE();
            XE_MFMA(1, 0, 1, 0, vl);
            mk(rB, pX, 0);                                                   // B1 lo (B0 lo was consumed in gap 1)
            // halo(c + 1) has landed: its DMA left in chunk c - 2 (the prologue for c < 2), and at most the 8 youngest vector memory
            // operations may be outstanding here -- chunk 0: halo(2)'s 4 pieces + the 4 filter pieces of steps 0-1; later chunks have 12
            // more in between.  Then the block barrier: halo(c + 1) visible to every wave, halo(c)'s buffer spent (its last read: B1 hi above).
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            xd_lds_barrier();
            XE_FENCE();
        }
        {   // ---- step 2: MFMAs (jj 1, h 0) from slot 0; forms V(jj 1, h 1) = S1 +- B1 into slot 1; columns A0', S0' of the next chunk
            const xf_f16x8 vh = frag(vhi[0]), vl = frag(vlo[0]);
            XE_MFMA0(0, 1, 0, 1, vh);
            t_second(rB, rS[1], 0);
            pX = rd(hnxt, oA, 0, 0);                                         // A0' lo
            XE_FENCE();
            XE_MFMA0(0, 1, 1, 1, vh);
            v_hi(1, 1, 0);
            mk(rB, pY, 1);                                                   // B1 hi
            load_u(cu_next, XdInt<0>{}, XdInt<0>{});                         // U[0] hi of the next chunk (its last reader: step 1)
            XE_FENCE();
            XE_MFMA(0, 1, 0, 0, vh);
            v_lo(1, 1, 0);
            pY = rd(hnxt, oA, 0, 1);                                         // A0' hi
            XE_FENCE();
            XE_MFMA(0, 1, 1, 0, vh);
            t_second(rB, rS[1], 2);
            v_hi(1, 1, 2);
            mk(rA, pX, 0);                                                   // A0' lo (A1 lo was consumed in step 0)
            XE_FENCE();
            XE_MFMA(0, 1, 0, 0, vl);
            v_lo(1, 1, 2);
            pX = rd(hnxt, oS, 0, 0);                                         // S0' lo
            dma_halo(hcur, hso, XdInt<0>{}, XdInt<1>{});                     // halo(c + 3) -> the buffer halo(c) was read from
            XE_FENCE();
            XE_MFMA(0, 1, 1, 0, vl);
            mk(rA, pY, 1);                                                   // A0' hi
            pZ = rd(hnxt, oS, 0, 1);                                         // S0' hi
            dma_halo(hcur, hso, XdInt<1>{}, XdInt<2>{});
            XE_FENCE();
        }
        {   // ---- step 3: MFMAs (jj 1, h 1) from slot 1; forms V(jj 0, h 0)' = A0' - S0' into slot 0; columns A1', S1' of the next chunk
            const xf_f16x8 vh = frag(vhi[1]), vl = frag(vlo[1]);
            XE_MFMA0(1, 1, 0, 1, vh);
            mk(rS[0], pX, 0);                                                // S0' lo
            t_first(rA, rS[0], 0);
            pX = rd(hnxt, oA, 1, 0);                                         // A1' lo
            XE_FENCE();
            XE_MFMA0(1, 1, 1, 1, vh);
            v_hi(0, 0, 0);
            mk(rS[0], pZ, 1);                                                // S0' hi
            load_u(cu_next, XdInt<1>{}, XdInt<1>{});                         // U[1] lo of the next chunk (dead since the MFMA above)
            XE_FENCE();
            XE_MFMA(1, 1, 0, 0, vh);
            v_lo(0, 0, 0);
            pZ = rd(hnxt, oA, 1, 1);                                         // A1' hi
            XE_FENCE();
            XE_MFMA(1, 1, 1, 0, vh);
            t_first(rA, rS[0], 2);                                           // (A0' hi: made in step 2, gap 6)
            v_hi(0, 0, 2);
            mk(rA, pX, 0);                                                   // A1' lo (A0' lo was consumed in gap 1)
            XE_FENCE();
            XE_MFMA(1, 1, 0, 0, vl);
            v_lo(0, 0, 2);
            pY = rd(hnxt, oS, 1, 0);                                         // S1' lo -> step 0, gap 1
            dma_halo(hcur, hso, XdInt<2>{}, XdInt<3>{});
            XE_FENCE();
            XE_MFMA(1, 1, 1, 0, vl);
            mk(rA, pZ, 1);                                                   // A1' hi (A0' hi was consumed in gap 4)
            pZ = rd(hnxt, oS, 1, 1);                                         // S1' hi -> step 0, gap 2
            dma_halo(hcur, hso, XdInt<3>{}, XdInt<XE_NDMA>{});
            XE_FENCE();
        }
        float* const t = hcur; hcur = hnxt; hnxt = hthird; hthird = t;
    };

    // ---- prologue: every load the block needs before its first MFMA leaves here, back to back (wino_x3d_kernel's order)
    float* const sc_lds = reinterpret_cast<float*>(smem_xe + XD_SC_OFFSET);
    int cm_src;
    {
        const unsigned P = (unsigned)tid;                                    // 512 >= 10 x 34 halo pixels
        const unsigned hr = __umul24(P, 1928u) >> 16, hc = P - (unsigned)XF_HC * hr;
        const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
        const bool inb = hr < (unsigned)X3_HR && gy >= 0 && gy < H && gx >= 0 && gx < W;
        cm_src = inb ? (int)((__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * 4u) : (int)0xFFFFFFF0u;
    }
    XE_FENCE();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(mrs, (xd_lds_ptr)(smem_xe + XD_CM_OFFSET + wave * 256), 4, cm_src, 0, 0, 0);
    load_u(0, XdInt<0>{}, XdInt<1>{}); load_u(0, XdInt<0>{}, XdInt<0>{}); load_u(0, XdInt<1>{}, XdInt<1>{});   // (U[1] hi: step 0 of chunk 0)
    if (wave < 4) {                                                          // the block's 16 x 64 filter scales: 16 bytes per thread of waves 0-3
        const float* const uinv0 = reinterpret_cast<const float*>(ublob + (size_t)16 * K16 * u_rbt * HX_RB) + (size_t)(tid >> 4) * (u_rbt * 32) + 64 * cb + (tid & 15) * 4;
        __builtin_amdgcn_global_load_lds(uinv0, (xd_lds_ptr)(smem_xe + XD_SC_OFFSET + wave * 1024), 16, 0, 0);
    }
    if (wave == 4 && lane < 16) __builtin_amdgcn_global_load_lds(bias + 64 * cb + 4 * lane, (xd_lds_ptr)(smem_xe + XD_SC_OFFSET + 4096), 16, 0, 0);
    XE_FENCE();
    halo_sources();
    XE_FENCE();
    dma_halo(hcur, 0, XdInt<0>{}, XdInt<XE_NDMA>{});
    dma_halo(hnxt, (K16 > 1 ? 1 : 0) * 64, XdInt<0>{}, XdInt<XE_NDMA>{});
    dma_halo(hthird, (K16 > 2 ? 2 : K16 - 1) * 64, XdInt<0>{}, XdInt<XE_NDMA>{});
    XE_FENCE();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_issued = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * XE_NDMA) : "memory");       // everything but halo(1) and halo(2)
    xd_lds_barrier();
    {   // the lane's two tile scales from the halo pixels' channel maxima (rows 4 h + 2 tyl + a, columns 2 txl + c)
        const float* cm = reinterpret_cast<const float*>(smem_xe + XD_CM_OFFSET) + (2 * tyl) * XF_HC + 2 * txl;
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
    // the loop's entry state: V(jj 0, h 0) of chunk 0 in slot 0; S0 and A1 made; S1 in flight in (pY, pZ)
    pX = rd(hcur, oA, 0, 0); pY = rd(hcur, oA, 0, 1); mk(rA, pX, 0); mk(rA, pY, 1);
    pX = rd(hcur, oS, 0, 0); pY = rd(hcur, oS, 0, 1); mk(rS[0], pX, 0); mk(rS[0], pY, 1);
    t_first(rA, rS[0], 0); v_hi(0, 0, 0); v_lo(0, 0, 0);
    t_first(rA, rS[0], 2); v_hi(0, 0, 2); v_lo(0, 0, 2);
    pX = rd(hcur, oA, 1, 0); pY = rd(hcur, oA, 1, 1); mk(rA, pX, 0); mk(rA, pY, 1);
    pY = rd(hcur, oS, 1, 0); pZ = rd(hcur, oS, 1, 1);
#ifdef XD_CLOCKS
    const unsigned long long xd_t_loop = __builtin_amdgcn_s_memrealtime(), xd_c_loop = __builtin_readcyclecounter();
#endif
    chunk(0, XdInt<1>{});                                                    // chunk 0 starts every accumulator from a zero C operand
    for (int c = 1; c < K16; ++c) chunk(c, XdInt<0>{});
#undef XE_MFMA
#undef XE_MFMA0
#ifdef XD_CLOCKS
    const unsigned long long xd_t_done = __builtin_amdgcn_s_memrealtime(), xd_c_done = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // the clamped re-loads of the last chunk have landed too ...
    __syncthreads();                                                         // ... and every wave is past its last halo read: the buffer below may overwrite the ring
#ifdef XD_CLOCKS
    const unsigned long long xd_t_e0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: A^T M A + bias + ReLU (+ 2x2 max-pool), wino_x3d_kernel's summation order:
    //   Y[i][0] = ((m_i0 + m_i1) + m_i2) 2^-e,  Y[i][1] = ((m_i1 - m_i2) - m_i3) 2^-e   (m = accumulator x filter scale), then the row pass.
    // Wave (i, 0) holds m_i0, m_i1 and forms Y[i][0]: it needs m_i2; wave (i, 1) holds m_i2, m_i3 and forms Y[i][1]: it needs m_i1.  Each wave
    // writes the vector its partner needs into the slot of the Y the PARTNER forms, the block synchronises, and each wave reads its own
    // slot and overwrites it with its Y (same lane, same address: no second buffer) -- one more LDS round trip than the four-wave kernel.
    float* const ybuf = reinterpret_cast<float*>(smem_xe);                   // [half 2][row i 4][b 2][tile 32][68]
    auto slot_of = [&](int h, int b) { return ybuf + (((h * 4 + wi) * 2 + b) * 32 + tl) * XD_MS; };
    auto scaled = [&](int h, int jj, int ct, int g, const f32x4& sbv) {
        const f32x4 a = {acc[h][jj][ct][4 * g], acc[h][jj][ct][4 * g + 1], acc[h][jj][ct][4 * g + 2], acc[h][jj][ct][4 * g + 3]};
        return a * sbv;
    };
    // phase A: the vector the partner needs (jp 0: m_i1 -> slot b = 1; jp 1: m_i2 -> slot b = 0)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * ct + 8 * g + 4 * kh;
            const f32x4 sbs = *reinterpret_cast<const f32x4*>(sc_lds + (4 * wi + 2 * jp + (jp ? 0 : 1)) * 64 + co);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x4 send = jp ? scaled(h, 0, ct, g, sbs) : scaled(h, 1, ct, g, sbs);
                *reinterpret_cast<f32x4*>(slot_of(h, jp ? 0 : 1) + co) = send;
            }
        }
    __syncthreads();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_ea = __builtin_amdgcn_s_memrealtime();
#endif
    // phase B: own Y from the two own vectors and the partner's, in place
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * ct + 8 * g + 4 * kh;
            const f32x4 sb0 = *reinterpret_cast<const f32x4*>(sc_lds + (4 * wi + 2 * jp) * 64 + co);
            const f32x4 sb1 = *reinterpret_cast<const f32x4*>(sc_lds + (4 * wi + 2 * jp + 1) * 64 + co);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float* const dst = slot_of(h, jp) + co;
                const f32x4 pv = *reinterpret_cast<const f32x4*>(dst);
                const f32x4 m0 = scaled(h, 0, ct, g, sb0), m1 = scaled(h, 1, ct, g, sb1);
                const f32x4 vi = {vinv[h], vinv[h], vinv[h], vinv[h]};
                // jp 0: m0 = m_i0, m1 = m_i1, pv = m_i2;  jp 1: m0 = m_i2, m1 = m_i3, pv = m_i1
                const f32x4 yv = jp ? xd_sub4(xd_sub4(pv, m0), m1) * vi : ((m0 + m1) + pv) * vi;
                *reinterpret_cast<f32x4*>(dst) = yv;
            }
        }
    __syncthreads();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_e1 = __builtin_amdgcn_s_memrealtime();
#endif
    const int Ho = H >> 1, Wo = W >> 1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int item = tid + 512 * it;                                     // (tile of 64, channel quad of 16)
        const int t = item >> 4, k = (item & 15) * 4;
        const int h = t >> 5, tt_ = t & 31;
        const int oty = 4 * by + 2 * h + (tt_ >> 4), otx = XF_TC * bx + (tt_ & 15);
        const bool live = oty < gm.th && otx < gm.tw && !(POOL && (oty >= Ho || otx >= Wo));
        if (!live && !cmax_out) continue;
        const int kg = 64 * cb + k;
        const float* yp = ybuf + ((h * 4) * 2 * 32 + tt_) * XD_MS + k;      // + (i 2 + b) 32 XD_MS
        f32x4 Y[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int bb = 0; bb < 2; ++bb) Y[i][bb] = *reinterpret_cast<const f32x4*>(yp + (i * 2 + bb) * (32 * XD_MS));
        const f32x4 bv = *reinterpret_cast<const f32x4*>(sc_lds + 1024 + k);
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
#ifdef XD_CLOCKS
    // timing build (tools/xd_clocks.py): wave 0 / lane 0 of every block leaves its stamps behind the (single-map) output
    if (tid == 0) {
        const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();
        float* rec = y_maps + (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout + (size_t)blockIdx.x * 16;
        rec[0] = (float)(xd_t_loop - xd_t_in); rec[1] = (float)(xd_t_done - xd_t_loop); rec[2] = (float)(t_out - xd_t_done);
        rec[3] = (float)(xd_c_done - xd_c_loop); rec[4] = (float)(xd_t_in & 0xFFFFFF); rec[5] = (float)(t_out & 0xFFFFFF);
        rec[6] = (float)K16; rec[7] = 1.0f;
        rec[8] = (float)(xd_t_issued - xd_t_in); rec[9] = (float)(xd_t_landed - xd_t_issued); rec[10] = (float)(xd_t_loop - xd_t_landed);
        rec[11] = (float)(xd_t_e0 - xd_t_done); rec[12] = (float)(xd_t_e1 - xd_t_e0); rec[13] = (float)(t_out - xd_t_e1);
        rec[14] = (float)(xd_t_ea - xd_t_e0);
    }
#endif
#undef XE_FENCE
}

int launch_wino_x3e(bool pool, unsigned grid_blocks, const float* x, const float* cmax, const unsigned char* ublob, const float* bias, float* y,
                    int H, int W, int cin, int cout, int u_rbt, int relu, const XfGeom& gm, float* cmax_out, hipStream_t s)
{
    if (pool) {
        auto kern = wino_x3e_kernel<true>;
        FRCNN_MAX_LDS_ONCE(kern, XD_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(512), XD_LDS_BYTES, s, x, cmax, ublob, bias, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out);
    } else {
        auto kern = wino_x3e_kernel<false>;
        FRCNN_MAX_LDS_ONCE(kern, XD_LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3(grid_blocks), dim3(512), XD_LDS_BYTES, s, x, cmax, ublob, bias, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out);
    }
    return check_launch();
}

}  // namespace frcnn
