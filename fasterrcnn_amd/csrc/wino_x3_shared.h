// wino_x3_shared.h -- what the two forms of the one-launch f32x3 Winograd layer share (csrc/wino_x3f.hip: wino_x3d_kernel, four waves;
// csrc/wino_x3e.hip: wino_x3e_kernel, eight waves): tile / halo geometry, the LDS layout, the block -> (tile block, cout block) mapping and
// the small device helpers.  Reference: the 3x3 convolution + ReLU (+ MaxPool2d) layers of pytorch/FasterRCNN/models/vgg16.py:77-96 and
// the RPN trunk (models/rpn.py:88).
#pragma once
#include "x3t.h"

namespace frcnn {

typedef _Float16 xf_f16x8 __attribute__((ext_vector_type(8)));

static constexpr int XF_TC = 16;                             // tile columns of a block
static constexpr int XF_HC = 2 * XF_TC + 2;                  // halo columns: 34 pixels
static constexpr int XF_PS = 20;                             // floats per halo pixel (16 + 4 padding)
static constexpr int X3_HR = 10;                             // halo rows: 4 tile rows x 2 + 2


// m_*: ceil(2^32 / d) of the three divisors of the block index (0 for d = 1): the quotient is ONE scalar multiply-high on the device instead of
// a division sequence per divisor between the block's entry and its first load (exact while block index x d < 2^32: checked by the launcher)
struct XfGeom { int tbx, tby, ncb, tw, th, xg; unsigned m_tbx, m_tby, m_ncb; int ntb; unsigned m_ntb; int g8; int n_items; };   // ntb: tile blocks of all maps; g8 = 8 / ncb (xg == 2, ncb < 8: a division sequence on the device otherwise); n_items = work items of the launch (= the grid of the one-item-per-block form)
__device__ __forceinline__ int xd_div(int n, int d, unsigned m) { return d == 1 ? n : (int)__umulhi((unsigned)n, m); }

typedef float xd_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 xd_f16x2 __attribute__((ext_vector_type(2)));

static constexpr int XD_HP = XF_HC / 2;                                       // 17 pixel slots per (halo row, column parity)
// halo: X3_HR x XF_HC pixels x XF_PS floats = 6,800 floats = 27,200 B per chunk
// The halo reaches LDS by DMA (buffer_load_dwordx4 ... lds: lane l of a wave instruction writes 16 B at base + 16 l, no staging registers, no
// ds_write) into a ring of THREE buffers, two chunks ahead.  A pixel's 80 bytes are five consecutive lanes: four channel quads and one
// padding lane whose load is out of range (nothing fetched) -- the 80-byte pixel stride is what keeps the patch reads conflict-free.
static constexpr int XD_NDMA = 7;                                              // DMA instructions per thread and chunk: 7 x 256 x 16 B = 28,672 B
static constexpr int XD_HBUF_BYTES = XD_NDMA * 256 * 16;                      // >= 27,200: the surplus lanes' zeros land in the buffer's tail
static constexpr int XD_HBUF_FLOATS = XD_HBUF_BYTES / 4;
typedef __attribute__((address_space(3))) void* xd_lds_ptr;
static constexpr int XD_MS = 68;                                              // floats between two tiles of the epilogue's M buffer (64 + 4: conflict-free)
static constexpr int XD_M_BYTES = 16 * 32 * XD_MS * 4;                       // 139,264: the epilogue's Y buffer [2 halves][4 rows][2][32 tiles][68] (>= the three halo buffers' 86,016)
static constexpr int XD_SC_OFFSET = XD_M_BYTES;                               // the block's filter scales [16][64] and bias [64] (4,352 B)
static constexpr int XD_CM_OFFSET = XD_SC_OFFSET + 16 * 64 * 4 + 64 * 4;      // the channel maxima of the block's 10 x 34 halo pixels (DMA: 512 floats)
static constexpr size_t XD_LDS_BYTES = XD_CM_OFFSET + 512 * 4;                // 145,664

template <int N> struct XdInt { static constexpr int value = N; };
#ifndef XD_ABLATE
#define XD_ABLATE 0          // timing experiments (tools/build_ablate.sh): 1 no operand VALU, 2 no patch reads / r, 4 no filter loads, 8 no halo traffic, 16 no MFMAs, 32 / 64 the prologue's loads (wino_x3f.hip)
#endif

// maximum over the aligned group of 16 lanes a lane belongs to (a DPP row): four row rotations, no LDS (a __shfl_xor is a ds_bpermute)
__device__ __forceinline__ float xd_rowmax16(float v)
{
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false)));   // row_ror:8
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false)));   // row_ror:4
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false)));   // row_ror:2
    v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false)));   // row_ror:1
    return v;
}

// a - b on four floats as two v_pk_add_f32 with the negate modifier (the compiler scalarises a float32 vector subtraction into four v_sub_f32:
// there is no v_pk_sub_f32 and it does not fold the negation into the packed add).  Epilogue only: beside MFMAs packed float32 is the slow form.
__device__ __forceinline__ f32x4 xd_sub4(f32x4 a, f32x4 b)
{
    xd_f32x2 a0 = {a[0], a[1]}, a1 = {a[2], a[3]}, b0 = {b[0], b[1]}, b1 = {b[2], b[3]}, r0, r1;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r0) : "v"(a0), "v"(b0));
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r1) : "v"(a1), "v"(b1));
    return f32x4{r0[0], r0[1], r1[0], r1[1]};
}

// Tile slot s (0 .. 31: the MFMA column a lane holds, lane & 31) -> tile (row ty, column tx) of the half's 2 x 16 tiles.  NOT s >> 4, s & 15:
// a ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} (+ 32 for the upper half wave; MI355X_MICROARCH.md,
// LDS), one LDS cycle per group when its 16 lanes hit 16 different bank quads.  The 16 tiles of ONE tile row are 80 bytes apart = 16
// different quads; with the plain numbering every group mixes the two tile rows, whose addresses differ by 5,440 bytes = 16 banks mod 64,
// and four of its lane pairs collide: every patch read took two cycles per group (round 5's unexplained SQ_LDS_BANK_CONFLICT /
// SQ_LDS_IDX_ACTIVE = 0.45 on wino_x3d_kernel).  Here a lane group IS a tile row.
__device__ __forceinline__ void xd_slot_tile(int s, int& ty, int& tx)
{
    ty = (int)((0xF00F0FF0u >> s) & 1u);          // s: 0-3 -> 0, 4-11 -> 1, 12-15 -> 0, 16-19 -> 1, 20-27 -> 0, 28-31 -> 1
    tx = s - 4 * ((s + 4) >> 3);                  // s: 0-3 -> s, 4-11 -> s - 4, 12-19 -> s - 8, 20-27 -> s - 12, 28-31 -> s - 16 (as a chain of ?: the compiler made it divergent branches)
}

__device__ __forceinline__ void xd_lds_barrier()
{
    // LDS writes of this wave done, then the block barrier; the outstanding GLOBAL loads (next chunk's filter fragments) stay in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}


// Block -> XCD mapping (hardware block b runs on XCD b % 8, every XCD has its own 4 MB L2).  gm.xg != 0 (the number of tile blocks is a
// multiple of 8): the ncb output-channel blocks of one tile block run on the SAME XCD, back to back -- they share the input halo through
// that XCD's L2 instead of fetching it ncb times through the fabric (measured: fabric traffic 203 MB per launch against 72 MB algorithmic
// with the plain order, and 1-2 % of the launch time).  Otherwise: output-channel block fastest, round-robin over the XCDs.
// gm.xg == 2, FILTER-RESIDENT (layers with >= 256 input channels: a cout block's filter records are 1-2 MB, and with the order above
// every XCD streams the records of ALL cout blocks at once -- 16 MB against a 4 MB L2 for a 512-channel layer: each block fetched its
// 2 MB through the fabric, 640 MB per launch): an XCD OWNS cout blocks -- XCD x works on cout blocks x, x + 8, ... one after the other
// over all tile blocks (ncb >= 8), or 8 / ncb XCDs share a cout block and split the tile blocks (ncb = 1, 2, 4) -- so the records of
// the cout block in progress stay in that XCD's L2 and leave the fabric once per XCD; the input halo is what crosses it per cout block.
// Returns false for the surplus blocks of the last group (the grid is 8 x ceil(ntb / g) blocks when ncb < 8).
__device__ __forceinline__ bool xd_block_to_tile(const XfGeom& gm, int b, int& cb, int& bx, int& by, int& map)
{
    if (gm.xg == 2) {
        const int xcd = b & 7, s = b >> 3;
        if (gm.ncb >= 8) {
            const int k = xd_div(s, gm.ntb, gm.m_ntb);
            cb = xcd + 8 * k;
            b = s - k * gm.ntb;
        } else {
            const int g = gm.g8, sub = xd_div(xcd, gm.ncb, gm.m_ncb);
            cb = xcd - sub * gm.ncb;
            b = s * g + sub;
            if (b >= gm.ntb) return false;
        }
    } else if (gm.xg) {
        const int xcd = b & 7, q = b >> 3, qq = xd_div(q, gm.ncb, gm.m_ncb);
        cb = q - qq * gm.ncb;
        b = qq * 8 + xcd;
    } else {
        const int qq = xd_div(b, gm.ncb, gm.m_ncb);
        cb = b - qq * gm.ncb;
        b = qq;
    }
    const int b1 = xd_div(b, gm.tbx, gm.m_tbx);
    bx = b - b1 * gm.tbx;
    map = xd_div(b1, gm.tby, gm.m_tby);
    by = b1 - map * gm.tby;                                                  // blocks of FOUR tile rows
    return true;
}

// the eight-wave form of the same layer (csrc/wino_x3e.hip); same arguments and results (bit for bit) as launch_conv3x3_winograd_x3_fused's kernel
int launch_wino_x3e(bool pool, unsigned grid_blocks, const float* x, const float* cmax, const unsigned char* ublob, const float* bias, float* y,
                    int H, int W, int cin, int cout, int u_rbt, int relu, const XfGeom& gm, float* cmax_out, hipStream_t s);

// the two-pass form with 128 output channels per block (csrc/wino_x3p.hip, FRCNN_X3F_PAIR); same results bit for bit.  spill: block-private
// scratch for the first pass's accumulators, >= conv3x3_winograd_x3_pair_spill_bytes (256 KB per block of the grid)
size_t conv3x3_winograd_x3_pair_spill_bytes(int N, int H, int W, int cout);
int launch_wino_x3p(bool pool, const float* x, const float* cmax, const unsigned char* ublob, const float* bias, float* y, int N, int H, int W,
                    int cin, int cout, int relu, float* cmax_out, float* spill, size_t spill_bytes, hipStream_t s);

}  // namespace frcnn
