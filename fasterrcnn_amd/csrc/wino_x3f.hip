// wino_x3f.hip -- the Winograd F(2x2,3x3) layer in the f32x3 arithmetic as ONE launch: the form of every 3x3 layer of VGG-16 from conv1_2
// on (pytorch/FasterRCNN/models/vgg16.py:77-96: 3x3 convolution + ReLU, MaxPool2d after each block) and of the RPN trunk (models/rpn.py:88).
// The V + M scratch of conv1_2 ... conv3_3 would be 600 MB per layer in the three-launch form of csrc/wino_x3.hip: those layers run here in
// every slot; the 512-channel layers run here when several images are in flight (frcnn_forward_params.winograd_x3f_mask, round 4).
//
// Arithmetic: the per-tile scale, the fp16 split and the float32 accumulation order per (position, tile, output channel) -- hi*lo, hi*hi,
// lo*hi per 16-channel chunk, chunks in order -- are the three-launch layer's (launch_conv3x3_winograd_x3): the accumulators hold the same
// bits.  The output transform combines the position columns before the rows (wino_output_kernel: rows first); the results differ by
// float32 rounding of the 2 x 2 output sums (tests/test_gemm_x3t_gpu.py: <= 2e-6 of max|y| apart, same error against float64).
//
// Block = 4 waves = 64 tiles (4 tile rows x 16) x 64 output channels x all 16 positions; wave w owns position row i = w (positions
// 4 w .. 4 w + 3) of every tile: 16 accumulator tiles of 32 x 32 = 256 accumulator registers, one block per CU.  History (rounds 3-4,
// DESIGN.md section 5): round 3's three variants staged the filter records through LDS by LDS-DMA and were bound by that staging
// (conv3_2: 208-232 us against the float32 kernel's 174); the kernel below loads them straight into registers (conv3_2: 119 us), then moved
// the halo to an LDS-DMA ring and the prologue's small loads to DMA (105 us).
#include "wino_x3_shared.h"

namespace frcnn {

typedef unsigned xp_u32x4_t __attribute__((ext_vector_type(4)));

// ---- the kernel: 64 tiles per block, the filter fragments straight from L2 into registers ------------------------------------------------------
// What bounded round 3's versions was the L2 -> LDS staging of the filter records (64 KB per chunk and block by LDS-DMA: ~17 B per clock and CU).
// Every filter piece is read by exactly ONE wave (a wave owns a position row), and a piece IS the register image of an MFMA operand
// (lane l's fragment at byte 16 l), so the wave loads its 16 pieces of a chunk with plain 1 KB buffer loads into registers, one chunk
// ahead (two register sets), and the LDS holds nothing but the input halo (a ring of three buffers filled by LDS-DMA two chunks ahead: ONE
// barrier per chunk, no staging registers).  The vector work of the operand formation (B^T d B, scale, fp16 split) is spread over the MFMAs
// of the chunk in eight steps of six MFMAs; step (h, j) forms the operand of step (h, j + 1).
// Measured with the shader clock inside the kernel (tools/xd_clocks.py, ablation builds XD_ABLATE; the chip runs this kernel at 1.55-1.8 GHz,
// the MFMA-free ablation at 2.3): 2800-2880 cycles per chunk against the 1536 of its 48 MFMAs.  The MFMA stream alone runs at 1581, the
// vector instructions alone (no MFMA) at 1211, and together they ADD rather than overlap: ~370 non-MFMA instructions per chunk are 7.7 per
// MFMA, and one wave per SIMD issues in order.  History of the loop: 3150-3260 cycles with the halo staged through registers (the
// ds_write of a chunk's halo waited ~570 cycles for loads issued six steps earlier).  Per block around the loop: 3.0 us before it (ONE memory
// round trip: the halo pixels' channel maxima, the scales and the bias go to LDS by DMA, the first filter pieces to registers, halo(0..2)
// to the ring, all issued back to back after the address arithmetic -- the first order was four round trips, 4.4-4.9 us) and 3.7-4.3 us
// after it (2,070 instructions: 256 accumulator reads, the scales, the column pass through LDS, the row pass, the output's channel maxima).
// A PERSISTENT form of the kernel (one block per CU walking its items, the next item's loads issued at the start of the epilogue, the Y
// buffer next to the ring instead of over it) was built and measured: the loads hide completely (0.02 us of wait) -- and the item costs
// the same, because prologue and epilogue are bound by instruction issue (~5 cycles each), not by the round trip; not in the tree.
// Round 5 (profiles/r05/xd_clocks_prologue_ablation.txt, XD_ABLATE 32 / 64): WITHOUT the prologue's filter and halo loads a block spends
// 1.9-2.0 us before its loop instead of 3.1-4.4 -- the 164 one-kilobyte loads of a block are 16 address cycles each on the CU's one
// texture-address unit, wherever they are issued (that is what the persistent form moved into its epilogue).  A form that carried the
// next item's pieces in the loop's own, today clamped and wasted, load slots could win that 1.1-2.5 us (3-7 % of a launch); the address
// arithmetic (1.0 us), the first operand (0.35 us) and the 3.8-4.5 us after the loop would remain.  Not built.

// ---- the epilogue's channel maxima: sixteen (four) row reductions as ONE butterfly -----------------------------------------------------------
// A thread of the row pass holds, per output pixel, the maximum of its four channels; the 16 lanes of a DPP row hold the 64 channels of that
// pixel.  Rounds 4-5 reduced every pixel on its own (four row rotations each: v_mov_b32_dpp + v_max_f32, every lane ending with the same
// value) and lane 0 of the row issued one atomic per pixel: 16 reductions and 16 atomic instructions per thread for a layer without pooling --
// and a vector-memory instruction is 16 cycles of the CU's address unit whatever its active lanes: the 16 stores + 16 atomics of a wave were
// the row pass's time (2.2-2.5 us against 1.3 of the pooled layers).  Here the reductions share a reduce-scatter: v_max_f32 with a DPP operand
// and a bank mask keeps, per step, the half of the values whose index bit equals the lane's bit (partner lane ^ 8, then lane ^ 4: banks of
// four lanes), two quad permutations finish it, and lane i of the row ends with the maximum of value i: 32 instructions instead of 128, and
// ONE atomic instruction per thread.  (VALU write -> DPP read of the same register needs two wait states: the order below keeps three
// instructions between them; the s_nop covers the compiler's code in front.)
// v[16] -> the row's maximum of v[lane & 15]
__device__ __forceinline__ float xd_rowmax16_scatter16(const float (&v)[16])
{
    float n0, n1, n2, n3, n4, n5, n6, n7;
    asm volatile("s_nop 1\n\t"
        "v_max_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_max_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_max_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_max_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_max_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_max_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_max_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_max_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_max_f32_dpp %0, %16, %16 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_max_f32_dpp %1, %17, %17 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_f32_dpp %2, %18, %18 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_max_f32_dpp %3, %19, %19 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_f32_dpp %4, %20, %20 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_max_f32_dpp %5, %21, %21 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "v_max_f32_dpp %6, %22, %22 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_max_f32_dpp %7, %23, %23 row_ror:8 row_mask:0xf bank_mask:0xc"
        : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3), "=&v"(n4), "=&v"(n5), "=&v"(n6), "=&v"(n7)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]),
          "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
    float m0, m1, m2, m3;
    // banks 0, 2 (lane bit 2 clear) keep n0..n3 and read lane + 4 (row_ror:12: lane i reads lane i - 12 = i + 4 of its row); banks 1, 3 keep n4..n7 and read lane - 4
    asm volatile(
        "v_max_f32_dpp %0, %4, %4 row_ror:12 row_mask:0xf bank_mask:0x5\n\tv_max_f32_dpp %1, %5, %5 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_max_f32_dpp %2, %6, %6 row_ror:12 row_mask:0xf bank_mask:0x5\n\tv_max_f32_dpp %3, %7, %7 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
        "v_max_f32_dpp %0, %8, %8 row_ror:4 row_mask:0xf bank_mask:0xa\n\tv_max_f32_dpp %1, %9, %9 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_max_f32_dpp %2, %10, %10 row_ror:4 row_mask:0xf bank_mask:0xa\n\tv_max_f32_dpp %3, %11, %11 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_max_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
        : "v"(n0), "v"(n1), "v"(n2), "v"(n3), "v"(n4), "v"(n5), "v"(n6), "v"(n7));
    const int q = (int)(threadIdx.x & 3);
    return q == 0 ? m0 : q == 1 ? m1 : q == 2 ? m2 : m3;
}
// v[4] -> the row's maximum of v[(lane & 15) >> 2] (every lane of a bank of four ends with its bank's value)
__device__ __forceinline__ float xd_rowmax16_scatter4(const float (&v)[4])
{
    float n0, n1, m;
    asm volatile("s_nop 1\n\t"
        "v_max_f32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\tv_max_f32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
        "v_max_f32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\tv_max_f32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %2, %0, %0 row_ror:12 row_mask:0xf bank_mask:0x5\n\tv_max_f32_dpp %2, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "=&v"(n0), "=&v"(n1), "=&v"(m)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return m;
}

#ifndef XD_EARLY_HALO
#define XD_EARLY_HALO 1      // 1 (shipped): halo(1) / halo(2) in chunks 0 / 1; 2: halo(1) at the end of the prologue, halo(2) in chunk 0; 0: both at the end of the prologue.
                             // Same-box A/B (tools/run_ab_x3f.sh): the three forms are within 1 % of each other on every layer and on the headline (920 vs 920 for 1 vs 2) --
                             // the 14 DMA pieces that fill the ring cost the address unit the same wherever they are issued
#endif
static constexpr int XD_FIRST = 1, XD_EARLY = XD_EARLY_HALO ? 2 : 0, XD_LAST = 4;             // chunk flags (the step lambda below)

// TWO: the layer has exactly two 16-channel chunks (32 input channels)
template <bool POOL, bool TWO>
__global__ __launch_bounds__(256, 1)
void wino_x3d_kernel(const float* __restrict__ x_maps, const float* __restrict__ cmax_maps, const unsigned char* __restrict__ ublob,
                     const float* __restrict__ bias, float* __restrict__ y_maps, int H, int W, int Cin, int Cout, int u_rbt, int relu,
                     XfGeom gm, float* __restrict__ cmax_out_maps)
{
#ifdef XD_CLOCKS
    const unsigned long long xd_t_in = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_xf[];
    float* const hbuf0 = reinterpret_cast<float*>(smem_xf);

#ifndef XD_PERSIST
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const int K16 = Cin >> 4;

#ifdef XD_PERSIST
    // experiment (tools/build_ablate.sh persist -DXD_PERSIST): one block per CU walks the items item, item + gridDim.x, ... (gridDim.x a multiple of 8:
    // the XCD an item runs on is unchanged) -- no dispatch gap between the blocks of a CU (~0.45 us of every block).  MEASURED, round 6
    // (same-box A/B): layers 105 / 58 / 81 / 50 / 88 / 84 / 55 / 97 / 96 / 42 us -> 147 / 82 / 106 / 68 / 107 / 102 / 67 / 113 / 109 / 50, 927 -> 796
    // images/sec: with the item loop around it the compiler hoists ~50 lane constants of the prologue out of the loop into SCRATCH (53-60
    // spilled registers, reloaded per item) and the row pass spills -- even a launch whose blocks walk ONE item each is 19 % slower.  Not built
    // further (the prefetch of the next item's filter pieces between the epilogue's instructions, DESIGN.md section 7, needs this form).
    for (int item = blockIdx.x; item < gm.n_items; item += gridDim.x) {
    // (XD_PERSIST=2, second half of round 6: the thread index laundered per item, so that no lane constant derived from it is loop-invariant to
    //  the compiler: 53-60 spilled registers -> 4-12, all of them outside the chunk loop; bit-identical (40 x3 tests).  MEASURED, same box
    //  (profiles/r06/ab_persist2.txt): layers 117 / 64 / 89 / 54 / 92 / 89 / 55 / 96.5 / 95.5 / 42 us -> 119 / 66 / 91 / 56.5 / 96 / 92.5 / 58 / 100.5 / 99 / 45,
    //  947-951 -> 919-924 images/sec in the driver's form: the 0.45 us dispatch gap a persistent block saves per item is what the item loop costs
    //  it (scalar registers of the kernel arguments spilled to lanes, the barrier behind the row pass, a static item order that no longer
    //  follows the CUs' finishing order) -- conv5_x, ONE item per block, is 7 % slower.  Only the prefetch of the next item's filter pieces
    //  under the epilogue (~1 us of a block) could pay for that, and by this measurement it would about break even: not built.)
    int tid_l = threadIdx.x;
#if XD_PERSIST >= 2
    asm volatile("" : "+v"(tid_l));
#endif
    const int tid = tid_l, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
#else
    {
    const int item = blockIdx.x;
#endif
    int cb, bx, by, map;
    if (!xd_block_to_tile(gm, item, cb, bx, by, map)) {                      // (block -> XCD mapping: wino_x3_shared.h)
#ifdef XD_PERSIST
        continue;
#else
        return;
#endif
    }
    const float* __restrict__ const x = x_maps + (size_t)map * H * W * Cin;
    const float* __restrict__ const cmax = cmax_maps + (size_t)map * H * W;
    float* __restrict__ const y = y_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) * Cout;
    // optional: the per-pixel channel maximum of the OUTPUT (atomic maxima into a buffer the caller zeroed; outputs are post-ReLU, and
    // non-negative floats order like their bit patterns) -- the next f32x3 layer's scale source
    float* __restrict__ const cmax_out = cmax_out_maps ? cmax_out_maps + (size_t)map * (POOL ? (size_t)(H >> 1) * (W >> 1) : (size_t)H * W) : nullptr;

    // Buffer loads: a piece outside the image carries an offset past the descriptor's size and the hardware returns zeros for it (the
    // "same" padding and the channel maxima of absent pixels cost no branch)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, H * W * Cin * (int)sizeof(float), 0x00020000);
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cmax), 0, H * W * (int)sizeof(float), 0x00020000);

    const int tl = lane & 31, kh = lane >> 5;
    int tyl, txl;
    xd_slot_tile(tl, tyl, txl);                                              // (a ds_read_b128 lane group = one tile row: conflict-free patch reads)
    // The channel maxima of the block's 10 x 34 halo pixels reach LDS by DMA with everything else the prologue fetches (ONE round trip to
    // memory for all of it; the address arithmetic of the whole prologue comes first so that nothing touches a register with a load in
    // flight); every lane then reduces the 4 x 4 of each of its two tiles to the tile's scale.  Absent pixels read as 0 (offset past the
    // descriptor's size).  (As 2 x 16 register loads per lane this was 32 loads and ~100 address instructions of every block's prologue.)
    float mult[2], vinv[2];

    // ---- halo staging by LDS-DMA: lane piece P = (7 it + ... ) -> LDS byte 16 P of the buffer = pixel slot P / 5, part P % 5 (4 = padding) ------
    // slot order [row][column parity][17]: the de-interleaved columns of the patch reads below
    const int hy0 = 8 * by - 1, hx0 = 2 * XF_TC * bx - 1;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);                 // (wave-uniform: the DMA's LDS base travels in M0)
    // (24-bit multiplies and reciprocal constants instead of integer divisions: this arithmetic stands between the block's entry and its first
    //  halo load; P < 1792 and slot < 359 keep the constants exact)
    int h_src[XD_NDMA];
    auto halo_source = [&](auto IT) {
        constexpr int it = decltype(IT)::value;
        const unsigned P = (unsigned)((it * 4 + wave) * 64 + lane);
        const unsigned slot = __umul24(P, 52429u) >> 18, part = P - 5u * slot;            // P / 5, P % 5
        const unsigned hr = __umul24(slot, 1928u) >> 16, rem = slot - (unsigned)XF_HC * hr;   // slot / 34, slot % 34
        const unsigned par = rem >= (unsigned)XD_HP ? 1u : 0u, hc = 2u * (rem - par * (unsigned)XD_HP) + par;
        const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
        // (one unsigned comparison per coordinate, no short circuit: as && the compiler made every piece a nest of execution-mask branches;
        //  only the last piece reaches halo rows >= X3_HR)
        bool inb = (part < 4u) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        if (it == XD_NDMA - 1) inb &= hr < (unsigned)X3_HR;
        const unsigned off = (__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * (unsigned)(Cin * 4) + 16u * part;
        h_src[it] = inb ? (int)off : (int)0xFFFFFFF0u;
    };
    // pieces [it0, it1) of chunk `chunk_off / 64` into the buffer at hb
    __amdgpu_buffer_rsrc_t xrs_ring = xrs;                                   // the descriptor of the ring's steady-state DMA (zero records past the last chunk)
    auto dma_halo = [&](float* hb, int chunk_off, auto IT0, auto IT1, const __amdgpu_buffer_rsrc_t& rs) {
        if (XD_ABLATE & 8) return;
#pragma unroll
        for (int it = decltype(IT0)::value; it < decltype(IT1)::value; ++it)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (xd_lds_ptr)(reinterpret_cast<unsigned char*>(hb) + (it * 4 + wave_u) * 1024), 16, h_src[it],
                                                     chunk_off, 0, 0);
    };

    // ---- filter fragments: piece (position p, row block r, term t) of chunk c = ublob + ((p K16 + c) u_rbt + 2 cb + r) 2 KB + t 1 KB -------
    // lane offset 16 l, everything else in the scalar offset
    xf_f16x8 U[2][4][2][2];                                                  // [register set][position j][output tile][0 = hi, 1 = lo]
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(ublob), 0, 16 * K16 * u_rbt * HX_RB, 0x00020000);
    const int chunk_stride = u_rbt * HX_RB;
    int ubase[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ubase[j] = ((4 * wave + j) * K16 * u_rbt + 2 * cb) * HX_RB;
    const int lane16 = lane * 16;
    auto load_u = [&](int chunk, auto SET, auto J) {
        constexpr int set = decltype(SET)::value, j = decltype(J)::value;
        const int so = ubase[j] + chunk * chunk_stride;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                U[set][j][ct][t] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16, so + ct * HX_RB + t * HX_PIECE, 0));
    };

    f32x16 acc[2][4][2];                 // (never zeroed: the first MFMA of every accumulator, in chunk 0, takes a zero C operand instead -- 512 instructions
                                         //  and ~1 us of every block's prologue)

    // ---- operand formation -------------------------------------------------------------------------------------------------------------
    // position row i = wave: r[b] = d[a1][b] +- d[a2][b] (B^T), V[i][j] = r[b1] +- r[b2] (B), csrc/winograd.hip's float32 operation order.
    // Scalar float32 instructions on purpose: beside MFMAs a packed-f32 instruction costs more than the two scalar ones it replaces
    // (MI355X_MICROARCH.md), and the file is compiled with -fno-slp-vectorize for the same reason.
    const int a1 = wave == 0 ? 0 : (wave == 2 ? 2 : 1);
    const int a2 = wave == 0 ? 2 : (wave == 2 ? 1 : (wave == 1 ? 2 : 3));
    const float rsgn = wave != 1 ? -1.0f : 1.0f;                             // i = 1: d1 + d2; i = 0, 2, 3: differences
    // LDS halo layout [row][column parity][17 slots][20 floats]: the 16 lanes of a tile row read patch column b of their tiles at pixel
    // columns 2 t + b, i.e. at CONSECUTIVE slots t + (b >> 1) of parity b & 1 -- 80 bytes apart, and 16 x 80 B covers every bank group once
    // (with a plain [row][column] layout the lanes are 160 B apart: lanes t and t + 8 collide, every read takes twice)
    const int d_lane = (4 * tyl * XD_HP + txl) * XF_PS + 8 * kh;             // + ((4 h + a) 2 + (b & 1)) XD_HP XF_PS + (b >> 1) XF_PS
    float r[4][8];                                                           // [patch column b][channel]
    f32x4 du0, du1, dw0, dw1;                                                // the two patch rows of one column on their way from LDS
    auto read_d = [&](const float* hb, int h, int bb) {
        const float* p1 = hb + d_lane + (((4 * h + a1) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        const float* p2 = hb + d_lane + (((4 * h + a2) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        du0 = *reinterpret_cast<const f32x4*>(p1); du1 = *reinterpret_cast<const f32x4*>(p1 + 4);
        dw0 = *reinterpret_cast<const f32x4*>(p2); dw1 = *reinterpret_cast<const f32x4*>(p2 + 4);
    };
    // the same column in two halves (channels 0..3, channels 4..7): the loop issues them two MFMAs apart and consumes each five MFMAs after
    // its issue -- with all four reads of a column in one burst the four waves' sixteen 1 KB reads queue up behind each other and the last
    // one returns after the r it feeds is due (measured: ~85 stall cycles per step on s_waitcnt lgkmcnt)
    auto read_d_lo = [&](const float* hb, int h, int bb) {
        const float* p1 = hb + d_lane + (((4 * h + a1) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        const float* p2 = hb + d_lane + (((4 * h + a2) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        du0 = *reinterpret_cast<const f32x4*>(p1);
        dw0 = *reinterpret_cast<const f32x4*>(p2);
    };
    auto read_d_hi = [&](const float* hb, int h, int bb) {
        const float* p1 = hb + d_lane + (((4 * h + a1) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        const float* p2 = hb + d_lane + (((4 * h + a2) * 2 + (bb & 1)) * XD_HP + (bb >> 1)) * XF_PS;
        du1 = *reinterpret_cast<const f32x4*>(p1 + 4);
        dw1 = *reinterpret_cast<const f32x4*>(p2 + 4);
    };
    // u + sgn w as ONE fused operation == u +- w rounded once (sgn w is exact).  Round 6: columns 1 and 2 are kept SCALED by their half's
    // tile scale 2^e (exact: a power of two), because every V below is then the scaled t in ONE rounding without a multiply of its own --
    // V0 2^e = fma(r0, 2^e, -r2s), V1 2^e = r1s + r2s, V2 2^e = r2s - r1s, V3 2^e = fma(r3, -2^e, r1s) -- and the fp16 hi term of a channel
    // PAIR is one v_cvt_pk_f16_f32 of the scaled t instead of two v_fma_mix (tools/micro/split_fill.hip, profiles/r06/micro_split_fill.txt:
    // every f32 <-> f16 converting instruction is half rate, ~8 cycles against 4.5 for a plain one, beside MFMAs or not; a step of six
    // MFMAs with this split measures 267 cycles against 310).  Same bits: the scaling commutes with every rounding.
    // (the multiply as volatile asm: the compiler otherwise sinks it from the gap it was placed in to the column's first use, four steps later)
    auto make_r = [&](int bb, int half, int hs) {
        const bool scaled = bb == 1 || bb == 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = half == 0 ? __builtin_fmaf(dw0[e], rsgn, du0[e]) : __builtin_fmaf(dw1[e], rsgn, du1[e]);
            if (scaled) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(v) : "v"(v), "v"(mult[hs]));
            r[bb][4 * half + e] = v;
        }
    };
    unsigned vhi[2][4], vlo[2][4];                                           // [slot][channel pair]: the operand in use and the one being formed
    // Channel pairs (e2, e2 + 1) of V(h, j): ts = (r[b1] +- r[b2]) 2^e in one rounding (above); hi = fp16(ts) by v_cvt_pk_f16_f32 (two channels
    // per instruction), lo = fp16(ts - hi) by v_fma_mix{lo,hi}_f16 (the difference is exact, so ONE rounding -- the bits of hx_split8).
    // Two pairs travel together so that no partial register write is read by the very next instruction (hipcc pads such pairs with
    // s_nop, and an s_nop costs an issue slot beside the MFMAs like any other instruction).
    float tt[4];
    auto v_adds = [&](int h, int j, int e2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 2 * e2 + q;
            tt[q] = j == 0 ? __builtin_fmaf(r[0][e], mult[h], -r[2][e]) : j == 1 ? r[1][e] + r[2][e] : j == 2 ? r[2][e] - r[1][e]
                                                                                                   : __builtin_fmaf(r[3][e], -mult[h], r[1][e]);
        }
    };
    auto v_hi = [&](int slot, int e2) {
        unsigned ha, hb;
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ha) : "v"(tt[0]), "v"(tt[1]));
        asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(tt[2]), "v"(tt[3]));
        vhi[slot][e2] = ha;
        vhi[slot][e2 + 1] = hb;
    };
    auto v_lo = [&](int slot, int e2) {
        // ONE asm statement per pair of registers: hipcc pads a v_fma_mixhi that follows inline-asm partial writes with an s_nop it cannot
        // prove unnecessary (an issue slot like any other); the other register's instruction between a register's two halves is the wait state
        unsigned la, lb;
        asm("v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(la), "=&v"(lb) : "v"(tt[0]), "v"(tt[1]), "v"(tt[2]), "v"(tt[3]), "v"(vhi[slot][e2]), "v"(vhi[slot][e2 + 1]));
        vlo[slot][e2] = la;
        vlo[slot][e2 + 1] = lb;
    };
    auto frag = [&](const unsigned (&q)[4]) { return __builtin_bit_cast(xf_f16x8, uint4{q[0], q[1], q[2], q[3]}); };
    // the six MFMAs of step (h, j): per accumulator filter lo x V hi, filter hi x V hi, filter hi x V lo (gemm_x3t_kernel's order)
#define XD_MFMA(SET, H_, J_, CT, UT, VV) \
    do { if (!(XD_ABLATE & 16)) acc[H_][J_][CT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[SET][J_][CT][UT], VV, acc[H_][J_][CT], 0, 0, 0); } while (0)
    // the first product of an accumulator (filter lo x V hi of chunk 0): C = 0
#define XD_MFMA0(SET, H_, J_, CT, UT, VV) \
    do { if (first) acc[H_][J_][CT] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[SET][J_][CT][UT], VV, xd_zero16, 0, 0, 0); else XD_MFMA(SET, H_, J_, CT, UT, VV); } while (0)
#define XD_FENCE() __builtin_amdgcn_sched_barrier(0)
#define XD_IF(BIT, STMT) do { if (!(XD_ABLATE & (BIT))) { STMT; } } while (0)

    // One chunk = eight steps (h, j) of six MFMAs.  Between the MFMAs of step s: the operand of step s + 1 (four channel pairs), one patch
    // column of the r the step after next needs, two of the next chunk's sixteen filter pieces, and the halo traffic -- about seven
    // instructions per MFMA, placed by hand (sched_barrier after every slice).  Order of the r columns: b = 0, 2, 1, 3, each into the slot
    // whose old value died in the previous step (V(., 1) = r1 + r2, V(., 2) = r2 - r1, V(., 3) = r1 - r3, V(., 0) = r0 - r2).
    // A column read in step s is first used by the operand formed in step s + 2, so its two channel halves are read in gaps 1 and 3 of step
    // s and turned into r in gap 6 of step s and gap 2 of step s + 1: five MFMAs (160 cycles) between every ds_read and its use.
#ifdef XD_STEPS
    // timing experiment on top of XD_CLOCKS (tools/xd_steps.py): shader cycles per STEP of the chunk, summed over the chunks (the s_memtime
    // at a step's entry waits for the wave's outstanding LDS reads: ~10 % more cycles than the un-instrumented loop)
    unsigned long long xd_step_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xd_step_last = 0;
#endif
    // FLAGS of a chunk (round 6: no load is issued twice, and none that nobody consumes):
    //   XD_FIRST  chunk 0: every accumulator starts from a zero C operand
    //   XD_EARLY  chunks 0 and 1: halo(c + 1) has NOT been issued yet (the prologue fetches halo(0) alone) -- its seven pieces leave in steps 0
    //             and 1, beside the MFMAs, instead of standing between the block's entry and its first MFMA (each 1 KB load is 16 cycles of the
    //             CU's one address unit, 64 with four waves: the prologue's 41 loads per wave were 2,600 cycles before the loop could start)
    //   XD_LAST   the last chunk: no filter pieces for a next chunk, no patch reads / r / operand for it (they were clamped re-reads and
    //             discarded work: ~23 loads and ~80 vector instructions of every block)
    // Chunks K16 - 3 and K16 - 2 still re-read the last halo (14 pieces nobody consumes): as run-time branches around the two DMA groups
    // of the generic chunk the skip cost the whole loop 250 cycles per chunk (measured; the branch ends the basic block the steps are
    // placed in), and as more instantiations it costs code size.
    auto step = [&](int ucb, int hso, int hso_next, float* hcur, float* hnxt, float* hthird_, auto PAR, auto S, auto FLAGS) {     // hso: byte offset of chunk c + 3 in a pixel
        constexpr int par = decltype(PAR)::value, s = decltype(S)::value;
#ifdef XD_STEPS
        {
            const unsigned long long now = __builtin_readcyclecounter();
            if (xd_step_last != 0) xd_step_acc[(s + 7) & 7] += now - xd_step_last;
            xd_step_last = now;
        }
#endif
        constexpr bool first = (decltype(FLAGS)::value & XD_FIRST) != 0, early = (decltype(FLAGS)::value & XD_EARLY) != 0,
                       last = (decltype(FLAGS)::value & XD_LAST) != 0;
        const f32x16 xd_zero16 = {};
        constexpr int h = s >> 2, j = s & 3, slot = s & 1, nslot = slot ^ 1;
        constexpr int nh = s == 3 ? 1 : s == 7 ? 0 : h, nj = (j + 1) & 3;                 // the operand formed in this step: V(nh, nj)
        constexpr int rb = j == 0 ? 0 : j == 1 ? 2 : j == 2 ? 1 : 3;                        // the r column read and formed in this step ...
        constexpr int prb = j == 0 ? 3 : j == 1 ? 0 : j == 2 ? 2 : 1;                       // ... and the previous step's, whose upper channels are still due
        constexpr int rh = h ^ 1;                                                           // ... belongs to the other half (h 0: this chunk's, h 1: the next chunk's)
        constexpr int prh = j == 0 ? h : rh;                                                // the half of the previous step's column (step (h, 0) finishes column 3 of half h)
        // the last chunk has no successor: its steps 4-7 read and form nothing for one (step 4 still finishes column 3 of ITS half 1)
        constexpr bool col = !(last && h == 1), pcol = !(last && s >= 5), form = !(last && s == 7);
        const float* rsrc = h == 0 ? hcur : hnxt;
        const xf_f16x8 vh = frag(vhi[slot]), vl = frag(vlo[slot]);
        XD_MFMA0(par, h, j, 0, 1, vh);
        if (col) XD_IF(2, read_d_lo(rsrc, rh, rb));
        if (form) XD_IF(1, v_adds(nh, nj, 0));
        XD_FENCE();
        XD_MFMA0(par, h, j, 1, 1, vh);
        if (form) XD_IF(1, v_hi(nslot, 0));
        if (pcol) XD_IF(2, make_r(prb, 1, prh));
        XD_FENCE();
        XD_MFMA(par, h, j, 0, 0, vh);
        if (form) XD_IF(1, v_lo(nslot, 0));
        if (col) XD_IF(2, read_d_hi(rsrc, rh, rb));
        // halo(c + 3) -> the buffer halo(c) was read from, free since the barrier of step 3 (its last patch read is that step's)
        if (s == 4 && !last) dma_halo(hcur, hso, XdInt<0>{}, XdInt<4>{}, xrs_ring);
        // chunks 0 and 1: halo(c + 1) -> the buffer steps 4-7 will read it from (landed by the barrier of step 3: the wait below)
        // (XD_EARLY_HALO 2: the piece that leaves early is halo(c + 2), into the third buffer, and only chunk 0 carries the flag: a DMA issued
        //  in steps 0-1 for the barrier of step 3 of the SAME chunk made that barrier wait for its round trip -- chunk 0 2,600-3,170 cycles,
        //  chunk 1 2,880-3,600 against 2,500; issued a chunk earlier it has landed when it is due)
        if (early && !last && !(XD_EARLY_HALO == 2 && TWO) && s == 0) dma_halo(XD_EARLY_HALO == 2 ? hthird_ : hnxt, hso_next + (XD_EARLY_HALO == 2 ? 64 : 0), XdInt<0>{}, XdInt<4>{}, xrs);
        if (early && !last && !(XD_EARLY_HALO == 2 && TWO) && s == 1) dma_halo(XD_EARLY_HALO == 2 ? hthird_ : hnxt, hso_next + (XD_EARLY_HALO == 2 ? 64 : 0), XdInt<4>{}, XdInt<XD_NDMA>{}, xrs);
        XD_FENCE();
        XD_MFMA(par, h, j, 1, 0, vh);
        if (form) XD_IF(1, v_adds(nh, nj, 2));
        if (form) XD_IF(1, v_hi(nslot, 2));
        XD_FENCE();
        XD_MFMA(par, h, j, 0, 0, vl);
        if (form) XD_IF(1, v_lo(nslot, 2));
        if (!last && !(XD_ABLATE & 4)) {   // two of the next chunk's filter pieces per step: U[par ^ 1][s >> 1][s & 1][hi, lo]; the constants land in the instruction offset
            const int so = ubase[s >> 1] + ucb;
            U[par ^ 1][s >> 1][s & 1][0] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16 + (s & 1) * HX_RB, so, 0));
            U[par ^ 1][s >> 1][s & 1][1] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16 + (s & 1) * HX_RB + HX_PIECE, so, 0));
        }
        XD_FENCE();
        XD_MFMA(par, h, j, 1, 0, vl);
        if (col) XD_IF(2, make_r(rb, 0, rh));
        if (s == 3) {
            // halo(c + 1) has landed.  Steady state: its DMA left in chunk c - 2, and LDS-DMA completes in issue order like any vector
            // memory load -- at most the 15 youngest may be outstanding: the last 3 pieces of halo(c + 2) (chunk c - 1, step 5), the 4
            // filter pieces of that chunk's steps 6 and 7 and the 8 of this chunk's steps 0-3 (the tail chunks, which issue fewer
            // loads, only make the count stricter).  Chunks 0 and 1 issued halo(c + 1) themselves, in steps 0 and 1: younger than its
            // last piece are the 6 filter pieces of steps 1-3.  Then the block barrier: halo(c + 1) visible to every wave, halo(c)'s
            // buffer spent.
            // (XD_EARLY_HALO 2, chunk 0: halo(1) left last in the prologue; younger are halo(2)'s 7 pieces and the 8 filter pieces of steps 0-3: 15 again
            //  -- 8 in the two-chunk kernel, which has no halo(2))
            if (early) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XD_EARLY_HALO == 2 ? (last ? 0 : TWO ? 8 : 15) : (last ? 0 : 6)) : "memory");
            else asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
            xd_lds_barrier();
        }
        if (s == 5 && !last) dma_halo(hcur, hso, XdInt<4>{}, XdInt<XD_NDMA>{}, xrs_ring);
        XD_FENCE();
    };
    // the ring: chunk c reads halo(c) from hcur (steps 0-3) and halo(c + 1) from hnxt (steps 4-7)
    float *hcur = hbuf0, *hnxt = hbuf0 + XD_HBUF_FLOATS, *hthird = hbuf0 + 2 * XD_HBUF_FLOATS;
    auto chunk = [&](int c, auto PAR, auto FLAGS) {
        // (past the last chunk the halo DMA of the two generic tail chunks goes through a descriptor of ZERO records -- the hardware fetches
        //  nothing and writes zeros nobody reads -- instead of branching; the filter offset of a chunk past the end is never used: XD_LAST)
        const int ucb = (c + 1) * chunk_stride, hso = (c + 3 < K16 ? c + 3 : K16 - 1) * 64, hson = (c + 1) * 64;
        xrs_ring = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, c + 3 < K16 ? H * W * Cin * (int)sizeof(float) : 0, 0x00020000);
        step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<0>{}, FLAGS); step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<1>{}, FLAGS);
        step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<2>{}, FLAGS); step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<3>{}, FLAGS);
        step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<4>{}, FLAGS); step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<5>{}, FLAGS);
        step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<6>{}, FLAGS); step(ucb, hso, hson, hcur, hnxt, hthird, PAR, XdInt<7>{}, FLAGS);
        float* const t = hcur; hcur = hnxt; hnxt = hthird; hthird = t;
    };

    // ---- prologue ---------------------------------------------------------------------------------------------------------------------
    // What the first MFMA needs leaves HERE, in order of need -- the halo pixels' channel maxima, the filter pieces of position column 0,
    // halo(0) -- with the address arithmetic of the halo pieces BETWEEN those loads (a wave that has just issued a 1 KB load waits ~64
    // cycles for the address unit's next slot whatever it does: the arithmetic used to stand in front of them); then, behind halo(0), the
    // block's 16 x 64 filter scales and 64 biases (on their way to LDS: the epilogue has no register to prefetch them into) and the
    // filter pieces of position columns 1-3, which land while the block forms its tile scales and its first operand.  halo(1) and
    // halo(2) leave in chunks 0 and 1 (XD_EARLY).  Rounds 4-5 issued all 41 loads of a wave here, back to back: 2.2-3.5 us before the
    // first could be consumed; the earliest order (maxima, reduce, scales -> LDS, bias -> LDS, then the halo) was four round trips.
    float* const sc_lds = reinterpret_cast<float*>(smem_xf + XD_SC_OFFSET);
    // the block's 16 x 64 filter scales and 64 biases -> LDS by DMA (16 bytes per thread = the [16][64] layout; the bias: 16 lanes): the
    // epilogue has no register to prefetch them into and would otherwise wait for each of its 64 scale vectors in turn
    const float* const uinv0 = reinterpret_cast<const float*>(ublob + (size_t)16 * K16 * u_rbt * HX_RB) + (size_t)(tid >> 4) * (u_rbt * 32) + 64 * cb + (tid & 15) * 4;
    int cm_src[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned P = (unsigned)(tid + 256 * q);
        const unsigned hr = __umul24(P, 1928u) >> 16, hc = P - (unsigned)XF_HC * hr;              // P / 34, P % 34
        const int gy = hy0 + (int)hr, gx = hx0 + (int)hc;
        const bool inb = (hr < (unsigned)X3_HR) & ((unsigned)gy < (unsigned)H) & ((unsigned)gx < (unsigned)W);
        cm_src[q] = inb ? (int)((__umul24((unsigned)gy, (unsigned)W) + (unsigned)gx) * 4u) : (int)0xFFFFFFF0u;
    }
    XD_FENCE();
    auto load_u1 = [&](auto J, auto CT, auto T) {                            // one filter piece of chunk 0
        constexpr int j = decltype(J)::value, ct = decltype(CT)::value, tt_ = decltype(T)::value;
        U[0][j][ct][tt_] = __builtin_bit_cast(xf_f16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, lane16, ubase[j] + ct * HX_RB + tt_ * HX_PIECE, 0));
    };
    __builtin_amdgcn_raw_ptr_buffer_load_lds(mrs, (xd_lds_ptr)(smem_xf + XD_CM_OFFSET + (0 * 4 + wave_u) * 256), 4, cm_src[0], 0, 0, 0);
    halo_source(XdInt<0>{}); XD_FENCE();
    __builtin_amdgcn_raw_ptr_buffer_load_lds(mrs, (xd_lds_ptr)(smem_xf + XD_CM_OFFSET + (1 * 4 + wave_u) * 256), 4, cm_src[1], 0, 0, 0);
    halo_source(XdInt<1>{}); XD_FENCE();
    if (!(XD_ABLATE & 32)) load_u1(XdInt<0>{}, XdInt<0>{}, XdInt<0>{});
    halo_source(XdInt<2>{}); XD_FENCE();
    if (!(XD_ABLATE & 32)) load_u1(XdInt<0>{}, XdInt<0>{}, XdInt<1>{});
    halo_source(XdInt<3>{}); XD_FENCE();
    if (!(XD_ABLATE & 32)) load_u1(XdInt<0>{}, XdInt<1>{}, XdInt<0>{});
    halo_source(XdInt<4>{}); XD_FENCE();
    if (!(XD_ABLATE & 32)) load_u1(XdInt<0>{}, XdInt<1>{}, XdInt<1>{});
    halo_source(XdInt<5>{}); halo_source(XdInt<6>{}); XD_FENCE();
    // (XD_ABLATE 32, timing experiment: what the block pays for the prologue's loads -- no filter pieces, no halo)
    if (!(XD_ABLATE & 32)) dma_halo(hcur, 0, XdInt<0>{}, XdInt<XD_NDMA>{}, xrs);
    __builtin_amdgcn_global_load_lds(uinv0, (xd_lds_ptr)(smem_xf + XD_SC_OFFSET + wave_u * 1024), 16, 0, 0);
    const xd_lds_ptr bias_lds = (xd_lds_ptr)(smem_xf + XD_SC_OFFSET + 4096);   // (the cast outside the divergent branch)
    if (wave_u == 0 && lane < 16) __builtin_amdgcn_global_load_lds(bias + 64 * cb + 4 * lane, bias_lds, 16, 0, 0);
    XD_FENCE();
    if (!(XD_ABLATE & 32)) { load_u(0, XdInt<0>{}, XdInt<1>{}); load_u(0, XdInt<0>{}, XdInt<2>{}); load_u(0, XdInt<0>{}, XdInt<3>{}); }
    XD_FENCE();
    if (XD_EARLY_HALO != 1 && !(XD_ABLATE & 32)) {
        dma_halo(hnxt, 64, XdInt<0>{}, XdInt<XD_NDMA>{}, xrs);
        if (XD_EARLY_HALO == 0 && !TWO) dma_halo(hthird, 128, XdInt<0>{}, XdInt<XD_NDMA>{}, xrs);
    }
    XD_FENCE();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_issued = __builtin_amdgcn_s_memrealtime();
#endif
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((XD_ABLATE & 32) ? 0 : 12 + (XD_EARLY_HALO == 1 ? 0 : XD_EARLY_HALO == 2 ? XD_NDMA : (TWO ? 1 : 2) * XD_NDMA)) : "memory");       // everything but the 12 filter pieces of position columns 1-3: the maxima, column 0's pieces, halo(0), the scales and the bias are in
    xd_lds_barrier();
    {   // the lane's two tile scales from the halo pixels' channel maxima (rows 4 h + 2 tyl + a, columns 2 txl + c)
        const float* cm = reinterpret_cast<const float*>(smem_xf + XD_CM_OFFSET) + (2 * tyl) * XF_HC + 2 * txl;
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
    // state at the top of a chunk: r = the r of (chunk, half 0) with column 0 already replaced ... the loop's steady state is entered with
    // V(0, 0) formed and r[1..3] of half 0 live; r[0] is free (the loop's first step writes half 1's column 0 there)
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) { read_d(hcur, 0, bb); make_r(bb, 0, 0); make_r(bb, 1, 0); }
    v_adds(0, 0, 0); v_hi(0, 0); v_lo(0, 0);
    v_adds(0, 0, 2); v_hi(0, 2); v_lo(0, 2);
#ifdef XD_CLOCKS
    const unsigned long long xd_t_loop = __builtin_amdgcn_s_memrealtime(), xd_c_loop = __builtin_readcyclecounter();
#endif
    // K16 is even (cin % 32 == 0: checked by the launcher).  Chunk 0 starts every accumulator from a zero C operand; chunks 0 and 1 fetch
    // halo(1) and halo(2); the last chunk fetches and forms nothing for a successor
    chunk(0, XdInt<0>{}, XdInt<XD_FIRST | XD_EARLY>{});
#ifdef XD_CHUNK_CLOCKS
    const unsigned long long xd_c_k0 = __builtin_readcyclecounter();
    unsigned long long xd_c_k1 = xd_c_k0, xd_c_kl = xd_c_k0, xd_c_kl2 = xd_c_k0;
#endif
    if constexpr (TWO) {                                                     // (its own instantiation: as a run-time branch the two paths cost the register allocation 318 spills)
        int one = 1;
        asm volatile("" : "+s"(one));
        for (int rep = 0; rep < one; ++rep) chunk(1, XdInt<1>{}, XdInt<(XD_EARLY_HALO == 1 ? XD_EARLY : 0) | XD_LAST>{});     // (one trip: see below)
    } else {
        chunk(1, XdInt<1>{}, XdInt<(XD_EARLY_HALO == 1 ? XD_EARLY : 0)>{});
#ifdef XD_CHUNK_CLOCKS
        xd_c_k1 = __builtin_readcyclecounter();
#endif
        for (int c = 2; c < K16 - 2; c += 2) {
            chunk(c, XdInt<0>{}, XdInt<0>{});
            chunk(c + 1, XdInt<1>{}, XdInt<0>{});
        }
#ifdef XD_CHUNK_CLOCKS
        xd_c_kl2 = __builtin_readcyclecounter();
#endif
        chunk(K16 - 2, XdInt<0>{}, XdInt<0>{});
#ifdef XD_CHUNK_CLOCKS
        xd_c_kl = __builtin_readcyclecounter();
#endif
        // (a loop of ONE trip the compiler cannot count: as straight-line code in front of the epilogue the register allocator moved the
        //  epilogue's 176 accumulator reads up behind the last MFMA of each accumulator -- v_accvgpr_read_b32 behind s_nop 11, a full
        //  MFMA latency of stall per accumulator: the last chunk measured 3,600 cycles against the 2,480 of a steady-state chunk)
        int one = 1;
        asm volatile("" : "+s"(one));
        for (int rep = 0; rep < one; ++rep) chunk(K16 - 1, XdInt<1>{}, XdInt<XD_LAST>{});
    }
#undef XD_MFMA
#undef XD_MFMA0
#undef XD_FENCE
#undef XD_IF
#ifdef XD_CLOCKS
    const unsigned long long xd_t_done = __builtin_amdgcn_s_memrealtime(), xd_c_done = __builtin_readcyclecounter();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                         // (nothing is outstanding: the last chunks issue no load nobody consumes) ...
    __syncthreads();                                                         // ... and every wave is past its last halo read: the M buffer may overwrite the ring
#ifdef XD_CLOCKS
    const unsigned long long xd_t_e0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue: A^T M A + bias + ReLU (+ 2x2 max-pool).  A wave owns position ROW i = wave of every tile, i.e. all four columns j of
    // that row: the column combination M A (Y[i][0] = (m_i0 + m_i1) + m_i2, Y[i][1] = (m_i1 - m_i2) - m_i3) happens in registers, and only
    // the two Y per row go through LDS for the row combination A^T Y across the four waves -- half the LDS traffic of combining rows first
    // (what wino_output_kernel does for the three-launch layers: the two orders differ by float32 rounding only), both tile halves in ONE
    // pass.  A tile's 64 channels are 68 floats apart: with 64 the 32 lanes of a ds_write_b128 would hit the same four banks.
    float* const ybuf = reinterpret_cast<float*>(smem_xf);                   // [half 2][row i 4][b 2][tile 32][68]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * ct + 8 * g + 4 * kh;                     // the MFMA's row operand was the filter: accumulator rows = channels
            f32x4 sb[4];                                                 // one read of the four scale vectors serves both tile halves
#pragma unroll
            for (int j = 0; j < 4; ++j) sb[j] = *reinterpret_cast<const f32x4*>(sc_lds + (4 * wave + j) * 64 + co);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                // (whole-vector expressions: no MFMA runs beside the epilogue, so the packed float32 instructions they become are the cheap form here)
                // m_j = a_j x (filter scale 2^-e(j, channel)) is EXACT (a power of two), so every sum below is ONE fused multiply-add with the
                // bits of the separately rounded form it replaces (round 6: 16 packed instructions per eight outputs instead of 20):
                //   Y0 = ((m0 + m1) + m2) 2^-e(tile),   Y1 = ((m1 - m2) - m3) 2^-e(tile)
                f32x4 a[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = f32x4{acc[h][j][ct][4 * g], acc[h][j][ct][4 * g + 1], acc[h][j][ct][4 * g + 2], acc[h][j][ct][4 * g + 3]};
                const f32x4 vi = {vinv[h], vinv[h], vinv[h], vinv[h]};       // the tile's 2^-e: exact, commutes with every rounding above
                const f32x4 y0 = __builtin_elementwise_fma(a[2], sb[2], __builtin_elementwise_fma(a[1], sb[1], a[0] * sb[0])) * vi;
                const f32x4 y1 = __builtin_elementwise_fma(a[3], -sb[3], __builtin_elementwise_fma(a[2], -sb[2], a[1] * sb[1])) * vi;
                float* dst = ybuf + ((((h * 4 + wave) * 2) * 32 + tl) * XD_MS) + co;
                *reinterpret_cast<f32x4*>(dst) = y0;
                *reinterpret_cast<f32x4*>(dst + 32 * XD_MS) = y1;
            }
        }
    __syncthreads();
#ifdef XD_CLOCKS
    const unsigned long long xd_t_e1 = __builtin_amdgcn_s_memrealtime();
#endif
    const int Ho = H >> 1, Wo = W >> 1;
    // ReLU as ONE v_max_f32 against a uniform lower bound (0 or -inf): fmaxf(x, 0) is two instructions (the compiler quiets a possible
    // signalling NaN first) and `if (relu)` duplicated the row pass behind a branch
    const float lowb = relu ? 0.f : -__builtin_inff();
    auto xmax = [](float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
    // Stores through a buffer descriptor: a pixel outside the image carries an offset past the descriptor's size and the hardware drops the
    // store -- no execution-mask branch per pixel, 32-bit offsets (the launcher bounds a map's bytes by 2^31).  Every thread computes all
    // four of its items (a tile outside the image is garbage nobody stores: its halo was zeros), so the row's lanes stay together for
    // the reduce-scatter below.
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(y, 0, (POOL ? Ho * Wo : H * W) * Cout * (int)sizeof(float), 0x00020000);
    float pmv[POOL ? 4 : 16];                                                // per output pixel of the thread's four items: the maximum of its four channels
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int item = tid + 256 * it;                                     // (tile of 64, channel quad of 16)
        const int t = item >> 4, k = (item & 15) * 4;
        const int h = t >> 5, tt_ = t & 31;
        int sty, stx;
        xd_slot_tile(tt_, sty, stx);
        const int oty = 4 * by + 2 * h + sty, otx = XF_TC * bx + stx;
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
        if constexpr (POOL) {
            // max-pool of the ReLUs = ReLU of the maximum: two v_max3_f32 per channel instead of four ReLUs and three maxima
            f32x4 mx;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m3;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(o[0][0][e]), "v"(o[0][1][e]), "v"(o[1][0][e]));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(mx[e]) : "v"(m3), "v"(o[1][1][e]), "v"(lowb));
            }
            const bool live = (oty < Ho) & (otx < Wo);                     // (Ho <= th, Wo <= tw)
            const int off = live ? ((oty * Wo + otx) * Cout + kg) * 4 : (int)0xFFFFFFF0u;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xp_u32x4_t, mx), yrs, off, 0, 0);
            { float m3; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(mx[0]), "v"(mx[1]), "v"(mx[2])); pmv[it] = xmax(m3, mx[3]); }
        } else {
            const int base = ((2 * oty * W + 2 * otx) * Cout + kg) * 4;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    const bool ok = (2 * oty + a < H) & (2 * otx + bb < W);  // (th = ceil(H / 2): a tile row past it has 2 oty >= H)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[a][bb][e] = xmax(o[a][bb][e], lowb);
                    const int off = ok ? base + (a * W + bb) * Cout * 4 : (int)0xFFFFFFF0u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(xp_u32x4_t, o[a][bb]), yrs, off, 0, 0);
                    { float m3; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m3) : "v"(o[a][bb][0]), "v"(o[a][bb][1]), "v"(o[a][bb][2])); pmv[it * 4 + a * 2 + bb] = xmax(m3, o[a][bb][3]); }
                }
            }
        }
    }
    if (cmax_out) {
        // the per-pixel channel maxima of the block's 64 channels: one reduce-scatter over the row, ONE atomic per thread (the buffer was
        // zeroed by the caller; outputs are post-ReLU, and non-negative floats order like their bit patterns).  Lane l of a row ends
        // with pixel (item l >> 2, position l & 3) of the row's items (pooled: item l >> 2, lanes l & 3 == 0 write).
        const int li = tid & 15, itl = li >> 2;
        const int t = (tid >> 4) + 16 * itl, h = t >> 5, tt_ = t & 31;
        int sty, stx;
        xd_slot_tile(tt_, sty, stx);
        const int oty = 4 * by + 2 * h + sty, otx = XF_TC * bx + stx;
        if constexpr (POOL) {
            const float pm = xd_rowmax16_scatter4(reinterpret_cast<const float (&)[4]>(pmv));
            const bool live = oty < gm.th && otx < gm.tw && oty < Ho && otx < Wo;
            if (live && (li & 3) == 0) atomicMax(reinterpret_cast<unsigned*>(cmax_out + (size_t)oty * Wo + otx), __float_as_uint(pm));
        } else {
            const float pm = xd_rowmax16_scatter16(reinterpret_cast<const float (&)[16]>(pmv));
            const int yy = 2 * oty + ((li >> 1) & 1), xx = 2 * otx + (li & 1);
            const bool ok = oty < gm.th && otx < gm.tw && yy < H && xx < W;
            if (ok) atomicMax(reinterpret_cast<unsigned*>(cmax_out + (size_t)yy * W + xx), __float_as_uint(pm));
        }
    }
#ifdef XD_PERSIST
    __syncthreads();                                                         // the Y buffer is read: the next item's DMA may write the ring it lies over
#endif
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
#ifdef XD_CHUNK_CLOCKS
        // (tools/xd_clocks.py chunks: shader cycles of chunk 0, chunk 1, the steady-state chunks, chunk K16 - 2, the last chunk)
        rec[8] = (float)(xd_c_k0 - xd_c_loop); rec[9] = (float)(xd_c_k1 - xd_c_k0); rec[10] = (float)(xd_c_kl2 - xd_c_k1);
        rec[11] = (float)(xd_c_kl - xd_c_kl2); rec[12] = (float)(xd_c_done - xd_c_kl);
#endif
#ifdef XD_STEPS
        for (int q = 0; q < 8; ++q) rec[8 + q] = (float)xd_step_acc[q];
#endif
    }
#endif
    }   // (the item: one per block, or the persistent walk)
}

// ---- host side ------------------------------------------------------------------------------------------------------------
// Measured (tools/x3f_bench.py, tools/xd_clocks.py, profiles/r05): the two forms cost the same -- 2790-2870 against 2820-2900 cycles per chunk --
// so the default stays the four-wave form; FRCNN_X3F_WAVES8 selects the other.
static constexpr int X3F_EIGHT_WAVES_MIN_CIN = 1 << 30;
// cmax scratch: n_maps * H * W floats (the channel maxima of the layer input, computed here)
size_t conv3x3_winograd_x3_fused_workspace_bytes(int N, int H, int W) { return (size_t)N * H * W * sizeof(float); }

static inline size_t x3f_align256(size_t v) { return (v + 255) / 256 * 256; }
#ifndef FRCNN_EXPERIMENTS
// csrc/wino_x3p.hip ships in `make EXPERIMENTS=1` builds only (measured round 6: its loop is 12 % shorter, the forward is not -- DESIGN.md section 5)
size_t conv3x3_winograd_x3_pair_spill_bytes(int, int, int, int) { return 0; }
#endif
size_t conv3x3_winograd_x3_pair_workspace_bytes(int N, int H, int W, int cout)
{
    const size_t sp = conv3x3_winograd_x3_pair_spill_bytes(N, H, W, cout);
    return sp ? x3f_align256(conv3x3_winograd_x3_fused_workspace_bytes(N, H, W)) + sp : 0;
}

int launch_conv3x3_winograd_x3_fused(const float* x, const void* ublob, const float* b, float* y, int N, int H, int W, int cin, int cout,
                                     unsigned flags, void* ws, size_t ws_bytes, hipStream_t s, const float* cmax_ready, float* cmax_out,
                                     float* pair_spill, size_t pair_spill_bytes)
{
    if (flags & FRCNN_X3F_PAIR) {
#ifndef FRCNN_EXPERIMENTS
        return FRCNN_EUNSUPPORTED;
#else
        // the two-pass form, 128 output channels per block (csrc/wino_x3p.hip): ws = [channel maxima | spill scratch] unless the caller brings the scratch
        if (N < 1 || H < 1 || W < 1) return FRCNN_EUNSUPPORTED;
        if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
        if (cmax_out && !(flags & FRCNN_RELU)) return FRCNN_EINVAL;
        const size_t cm_bytes = x3f_align256(conv3x3_winograd_x3_fused_workspace_bytes(N, H, W));
        if (!cmax_ready && (!ws || ws_bytes < cm_bytes)) return FRCNN_EINVAL;
        if (!pair_spill) {
            if (!ws || ws_bytes < cm_bytes) return FRCNN_EINVAL;
            pair_spill = reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + cm_bytes);
            pair_spill_bytes = ws_bytes - cm_bytes;
        }
        const float* cm = cmax_ready;
        if (!cm) {
            int rc = launch_pixel_absmax(x, static_cast<float*>(ws), (long long)N * H * W, cin, s);
            if (rc) return rc;
            cm = static_cast<const float*>(ws);
        }
        return launch_wino_x3p((flags & FRCNN_POOL2) != 0, x, cm, static_cast<const unsigned char*>(ublob), b, y, N, H, W, cin, cout,
                               (flags & FRCNN_RELU) ? 1 : 0, cmax_out, pair_spill, pair_spill_bytes, s);
#endif
    }
    // the kernel walks the 16-channel chunks in pairs: cin % 32 == 0 (other widths: the three-launch layer, csrc/wino_x3.hip)
    if (N < 1 || H < 1 || W < 1 || cin < 32 || cin % 32 != 0 || cout < 64 || cout % 64 != 0) return FRCNN_EUNSUPPORTED;
    if ((size_t)H * W * cin >= ((size_t)1 << 29) || (size_t)H * W * cout >= ((size_t)1 << 29)) return FRCNN_EUNSUPPORTED;          // 32-bit byte offsets inside one map (input and output)
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    if (!cmax_ready && (!ws || ws_bytes < conv3x3_winograd_x3_fused_workspace_bytes(N, H, W))) return FRCNN_EINVAL;
    if (cmax_out && !(flags & FRCNN_RELU)) return FRCNN_EINVAL;          // the emitted maxima are those of non-negative outputs
    const float* cmax = cmax_ready;
    if (!cmax) {                                                          // nobody left the input's channel maxima behind: one pass over x
        int rc = launch_pixel_absmax(x, static_cast<float*>(ws), (long long)N * H * W, cin, s);
        if (rc) return rc;
        cmax = static_cast<const float*>(ws);
    }
    XfGeom gm;
    gm.tw = cdiv(W, 2); gm.th = cdiv(H, 2);
    gm.tbx = cdiv(gm.tw, XF_TC); gm.tby = cdiv(gm.th, 4);
    gm.ncb = cout / 64;
    const long long total = (long long)gm.tbx * gm.tby * gm.ncb * N;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    gm.xg = ((long long)gm.tbx * gm.tby * N) % 8 == 0 ? 1 : 0;
    gm.ntb = gm.tbx * gm.tby * N;
    long long grid_blocks = total;
#ifndef XD_NO_FILTER_RESIDENT
    if (cin >= 256 && (gm.ncb % 8 == 0 || 8 % gm.ncb == 0)) {
        gm.xg = 2;
        if (gm.ncb < 8) grid_blocks = 8LL * cdiv(gm.ntb, 8 / gm.ncb);
    }
#endif
    auto magic = [](int d) { return d == 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    gm.m_tbx = magic(gm.tbx); gm.m_tby = magic(gm.tby); gm.m_ncb = magic(gm.ncb); gm.m_ntb = magic(gm.ntb);
    gm.g8 = gm.ncb < 8 ? 8 / gm.ncb : 1;
    gm.n_items = (int)grid_blocks;
#ifdef XD_PERSIST
    if (grid_blocks > 256) grid_blocks = 256;                              // one block per CU walks the items
#endif
    if (total * std::max(std::max(gm.ncb, gm.ntb), std::max(gm.tbx, gm.tby)) >= 0x100000000ll) return FRCNN_EUNSUPPORTED;
    const int u_rbt = cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout) / 32;
    if ((size_t)16 * (cin / 16) * u_rbt * HX_RB >= ((size_t)1 << 31)) return FRCNN_EUNSUPPORTED;   // the record bank behind one buffer descriptor
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    const unsigned char* ub = static_cast<const unsigned char*>(ublob);
    // Two forms, same results bit for bit: eight waves (two per SIMD: the waves overlap each other's MFMAs and vector work) where the
    // loop dominates the block -- >= 8 chunks of 16 input channels -- and four waves (one LDS round trip less in the epilogue) for the
    // 64-channel layers, whose blocks are half prologue and epilogue.  FRCNN_X3F_WAVES4 / _WAVES8 force one (tests, tools).
    const bool eight = (flags & FRCNN_X3F_WAVES8) ? true : (flags & FRCNN_X3F_WAVES4) ? false : cin >= X3F_EIGHT_WAVES_MIN_CIN;
    if (eight)
#ifndef FRCNN_EXPERIMENTS
        return FRCNN_EUNSUPPORTED;     // csrc/wino_x3e.hip ships in `make EXPERIMENTS=1` builds only (measured round 5: no gain over the four-wave form)
#else
        return launch_wino_x3e((flags & FRCNN_POOL2) != 0, (unsigned)grid_blocks, x, cmax, ub, b, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out, s);
#endif
#define XD_LAUNCH(P_, T_)                                                                                                                    \
    do {                                                                                                                                    \
        auto kern = wino_x3d_kernel<P_, T_>;                                                                                                \
        FRCNN_MAX_LDS_ONCE(kern, XD_LDS_BYTES);                                                                                             \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid_blocks), dim3(256), XD_LDS_BYTES, s, x, cmax, ub, b, y, H, W, cin, cout, u_rbt, relu, gm, cmax_out); \
    } while (0)
    if (flags & FRCNN_POOL2) { if (cin == 32) XD_LAUNCH(true, true); else XD_LAUNCH(true, false); }
    else                     { if (cin == 32) XD_LAUNCH(false, true); else XD_LAUNCH(false, false); }
#undef XD_LAUNCH
    return check_launch();
}

}  // namespace frcnn
