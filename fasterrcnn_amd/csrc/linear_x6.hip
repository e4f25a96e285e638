// linear_x6.hip -- the dense layers fc1 / fc2 (models/vgg16.py:129-133: nn.Linear + ReLU on one image's RoIs) on the bf16 matrix
// pipe with fp32-class accuracy, the "f32x6" arithmetic of csrc/conv_x6.hip applied to a row-major GEMM
//
//   y[m][n] = act( bias[n] + sum_k a[m][k] * w[n][k] )
//
// Every fp32 operand is split EXACTLY into three bf16 terms, x = hi + mid + lo (each residual formed in fp32); a product is the sum
// of the six largest bf16 x bf16 partial products (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid) on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulation.  The three dropped terms are <= 2^-24 |a*w| each -- the size of one fp32 rounding of the product.
//
// Both operands arrive PRE-SPLIT as "x6 records": for a row-major matrix [R][K] the record of (row, 16-k chunk) is 96 contiguous
// bytes [hi 16 | mid 16 | lo 16] bf16 (split_rows_x6_kernel), stored CHUNK-MAJOR -- [K/16][rows][96 B] -- so that the tile a
// block stages per 16-k step (all 320 activation rows; its 128 weight rows) is ONE contiguous run of bytes: a wave's
// global_load_dwordx4 covers 8 full 128-byte lines.  (Row-major records made it 11 row segments of 96 B = ~16 lines per
// instruction and kept the CU's L1 / address path 70 % busy: fc1 355 us.)  Weights are split once at pack time (fc1: 411 MB fp32 -> 617 MB of
// records, streamed once per image); activations are split by the kernel that produces them (the RoI-pool output by
// split_rows_x6_kernel, fc1's output by the split-K reduction's epilogue), so the GEMM's staging is pure copying.
//
// Tiling = linear_mfma_kernel<5,1,2,4>: the block tile spans ALL rows (320 x 128, 8 waves = 2 x 4 of 5 x 1 MFMA tiles), so the
// weight records stream exactly once; deterministic split-K fills the chip (partials to scratch, fixed-order reduce fused with
// bias + ReLU + the split of the next layer's operand).  LDS rows are 96 B padded to 112 B (conflict-free ds_read_b128 for
// lane = row + 32 * k-half), double buffered (98 KB: one 8-wave block per CU); one barrier per 16-k stage of 30 MFMAs per wave.
#include "common.h"
#include <cstdlib>

namespace frcnn {

typedef __bf16 lx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short lx_u16;

#ifndef LX_ABLATE
#define LX_ABLATE 0      // timing experiments only (tools/build_ablate.sh): 1 no A loads, 2 no B loads, 4 no LDS writes (after the prologue)
#endif
static constexpr int LX_ROW = 112;                       // LDS bytes per record row (96 + 16 pad)
static constexpr int LX_BM = 320, LX_BN = 128;
static constexpr int LX_THREADS = 512;
static constexpr int LX_NA = (LX_BM * 6 + LX_THREADS - 1) / LX_THREADS;     // 4 pieces of 16 B per thread per stage (last partially)
static constexpr int LX_NB = (LX_BN * 6 + LX_THREADS - 1) / LX_THREADS;     // 2
static constexpr int LX_BUF = (LX_BM + LX_BN) * LX_ROW;                     // 50,176 B
static constexpr size_t LX_LDS_BYTES = 2 * (size_t)LX_BUF;

__device__ __forceinline__ lx_u16 lx_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (lx_u16)(u >> 16);
}
__device__ __forceinline__ float lx_bf16_f32(lx_u16 h) { return __uint_as_float((unsigned)h << 16); }

// x = hi + mid + lo exactly (barring overflow / subnormal tails)
__device__ __forceinline__ void lx_split3(float x, lx_u16& hi, lx_u16& mid, lx_u16& lo)
{
    hi = lx_bf16_rne(x);
    const float r1 = x - lx_bf16_f32(hi);
    mid = lx_bf16_rne(r1);
    const float r2 = r1 - lx_bf16_f32(mid);
    lo = lx_bf16_rne(r2);
}

// [R][ld] fp32 (K used columns, K % 16 == 0) -> records [K/16][rows_out][3][16] bf16.  One thread = 4 consecutive floats.
// rows_out >= R is the row count of the record array: the rows R .. rows_out-1 are zero filled (weight matrices padded to the
// column tile, activations to the 320-row tile).
__global__ __launch_bounds__(256)
void split_rows_x6_kernel(const float* __restrict__ a, int lda, unsigned char* __restrict__ rec, int R, int rows_out, int K)
{
    const int q4 = K >> 2;
    const size_t total = (size_t)rows_out * q4;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int row = (int)(idx / q4), k = (int)(idx % q4) * 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (row < R) v = *reinterpret_cast<const f32x4*>(a + (size_t)row * lda + k);
        lx_u16 hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lx_split3(v[j], hi[j], mid[j], lo[j]);
        unsigned char* p = rec + ((size_t)(k >> 4) * rows_out + row) * 96 + (k & 15) * 2;
        uint2 ph, pm, pl;
        ph.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);   ph.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
        pm.x = (unsigned)mid[0] | ((unsigned)mid[1] << 16); pm.y = (unsigned)mid[2] | ((unsigned)mid[3] << 16);
        pl.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16);   pl.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
        *reinterpret_cast<uint2*>(p) = ph;
        *reinterpret_cast<uint2*>(p + 32) = pm;
        *reinterpret_cast<uint2*>(p + 64) = pl;
    }
}

__global__ __launch_bounds__(LX_THREADS, 1)
void linear_x6_kernel(const unsigned char* __restrict__ a_rec, const unsigned char* __restrict__ w_rec,
                      const float* __restrict__ bias, float* __restrict__ y, int ldy, float* __restrict__ ws,
                      int M, int N, int nchunks, int chunks_per_split, int relu, int w_rows)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_lx[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * LX_BN;

    const int c_begin = blockIdx.z * chunks_per_split;
    int c_end = c_begin + chunks_per_split;
    if (c_end > nchunks) c_end = nchunks;
    const int nst = c_end - c_begin;

    // staging: the tile of a 16-k chunk is contiguous in the chunk-major record arrays (LX_BM activation rows: the caller pads;
    // this block's LX_BN weight rows); piece q = 16 bytes -> LDS (row q / 6, slot q % 6)
    const size_t a_chunk_bytes = (size_t)LX_BM * 96, b_chunk_bytes = (size_t)w_rows * 96;
    int a_src[LX_NA], b_src[LX_NB];
    int a_dst[LX_NA], b_dst[LX_NB];
#pragma unroll
    for (int it = 0; it < LX_NA; ++it) {
        int q = tid + LX_THREADS * it;
        if (q >= LX_BM * 6) q -= LX_BM * 6 / 2;           // surplus threads duplicate an earlier piece (same data, same address)
        const int row = q / 6, j = q - row * 6;
        a_src[it] = q * 16;
        a_dst[it] = row * LX_ROW + j * 16;
    }
#pragma unroll
    for (int it = 0; it < LX_NB; ++it) {
        int q = tid + LX_THREADS * it;
        if (q >= LX_BN * 6) q -= LX_BN * 6 / 2;
        const int row = q / 6, j = q - row * 6;
        b_src[it] = n0 * 96 + q * 16;
        b_dst[it] = LX_BM * LX_ROW + row * LX_ROW + j * 16;
    }
    // Register staging.  A (the activations' records, L2 resident) one tile ahead; B (the weight records, streamed from HBM exactly
    // once) THREE tiles ahead in a ring of three register sets: one 16-k stage of an 8-wave block is ~0.8 us, less than the loaded
    // HBM latency, and with a single tile in flight per CU the layer ran at 1.6 TB/s (measured) instead of at the matrix pipe's rate.
    f32x4 areg[LX_NA], breg[3][LX_NB];
    bool past_prologue = false;
    auto load_a = [&](int chunk) {
        if ((LX_ABLATE & 1) && past_prologue) return;
        const unsigned char* ab = a_rec + (size_t)chunk * a_chunk_bytes;
#pragma unroll
        for (int it = 0; it < LX_NA; ++it) areg[it] = *reinterpret_cast<const f32x4*>(ab + a_src[it]);
    };
    auto load_b = [&](int chunk, f32x4 (&br)[LX_NB]) {
        if ((LX_ABLATE & 2) && past_prologue) return;
        const unsigned char* wb = w_rec + (size_t)chunk * b_chunk_bytes;
#pragma unroll
        for (int it = 0; it < LX_NB; ++it) br[it] = *reinterpret_cast<const f32x4*>(wb + b_src[it]);
    };
    auto store_tiles = [&](int buf, const f32x4 (&br)[LX_NB]) {
        if ((LX_ABLATE & 4) && past_prologue) return;
        unsigned char* base = smem_lx + buf * LX_BUF;
#pragma unroll
        for (int it = 0; it < LX_NA; ++it) *reinterpret_cast<f32x4*>(base + a_dst[it]) = areg[it];
#pragma unroll
        for (int it = 0; it < LX_NB; ++it) *reinterpret_cast<f32x4*>(base + b_dst[it]) = br[it];
    };

    f32x16 acc[5];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    const int a_base = (160 * wm + li) * LX_ROW + lh * 16;
    const int b_base = (LX_BM + 32 * wn + li) * LX_ROW + lh * 16;

    if (nst > 0) {
        const int last = c_end - 1;
        auto clampc = [&](int c) { return c < c_end ? c : last; };       // prefetches past the end re-load the last tile, harmlessly
        load_a(c_begin);
        load_b(c_begin, breg[0]);
        store_tiles(0, breg[0]);
        load_a(clampc(c_begin + 1));
        load_b(clampc(c_begin + 1), breg[1]);
        load_b(clampc(c_begin + 2), breg[2]);
        load_b(clampc(c_begin + 3), breg[0]);
        __syncthreads();
        past_prologue = true;
        // fragments of tile 0 (single-buffered registers, refilled inside the stage as soon as their last term has been issued)
        lx_bf16x8 ah[5], am[5], al[5], bh, bm, bl;
        {
            const unsigned char* at = smem_lx + a_base;
            const unsigned char* bt = smem_lx + b_base;
            bh = *reinterpret_cast<const lx_bf16x8*>(bt);
            bm = *reinterpret_cast<const lx_bf16x8*>(bt + 32);
            bl = *reinterpret_cast<const lx_bf16x8*>(bt + 64);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                ah[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW);
                am[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW + 32);
                al[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW + 64);
            }
        }
#define LX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define LX_TERM(A, B)                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 5; ++i)                                                         \
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], B, acc[i], 0, 0, 0);
        // Software-pipelined stage (the schedule of conv3x3_x6p_kernel): the eight waves of the CU's only block run in lock step,
        // so every staging instruction is issued in the shadow of the wave's own MFMAs:
        //   terms mm, lh, hl (15 MFMAs) || LDS writes of tile s+1, then the global loads that refill those registers
        //   barrier (tile s+1 visible; nobody reads tile s from LDS after it: its fragments are in registers)
        //   term mh || read al', bl' ; term hm || read am' ; term hh || read bm' ; then ah', bh' (tile s+1's fragments)
        auto stage = [&](int s, f32x4 (&bset)[LX_NB]) {
            const int cur = s & 1;
            LX_TERM(am, bm)
            LX_TERM(al, bh)
            LX_TERM(ah, bl)
            store_tiles(cur ^ 1, bset);
            load_a(clampc(c_begin + s + 2));
            load_b(clampc(c_begin + s + 4), bset);
#pragma unroll
            for (int q = 0; q < LX_NA + LX_NB; ++q) { LX_SGB(0x008, 1); LX_SGB(0x200, 1); }
#pragma unroll
            for (int q = 0; q < LX_NA + LX_NB; ++q) { LX_SGB(0x008, 1); LX_SGB(0x020, 1); }
            LX_SGB(0x008, 15 - 2 * (LX_NA + LX_NB));
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            const unsigned char* at = smem_lx + (cur ^ 1) * LX_BUF + a_base;
            const unsigned char* bt = smem_lx + (cur ^ 1) * LX_BUF + b_base;
            LX_TERM(am, bh)
#pragma unroll
            for (int i = 0; i < 5; ++i) al[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW + 64);
            bl = *reinterpret_cast<const lx_bf16x8*>(bt + 64);
            LX_SGB(0x008, 1); LX_SGB(0x100, 2); LX_SGB(0x008, 1); LX_SGB(0x100, 2); LX_SGB(0x008, 1); LX_SGB(0x100, 2); LX_SGB(0x008, 2);
            __builtin_amdgcn_sched_barrier(0);
            LX_TERM(ah, bm)
#pragma unroll
            for (int i = 0; i < 5; ++i) am[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW + 32);
            LX_SGB(0x008, 1); LX_SGB(0x100, 2); LX_SGB(0x008, 1); LX_SGB(0x100, 2); LX_SGB(0x008, 1); LX_SGB(0x100, 1); LX_SGB(0x008, 2);
            __builtin_amdgcn_sched_barrier(0);
            LX_TERM(ah, bh)
            bm = *reinterpret_cast<const lx_bf16x8*>(bt + 32);
            LX_SGB(0x008, 1); LX_SGB(0x100, 1); LX_SGB(0x008, 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 5; ++i) ah[i] = *reinterpret_cast<const lx_bf16x8*>(at + i * 32 * LX_ROW);
            bh = *reinterpret_cast<const lx_bf16x8*>(bt);
        };
        int s = 0;
        for (; s + 2 < nst; s += 3) { stage(s, breg[1]); stage(s + 1, breg[2]); stage(s + 2, breg[0]); }
        if (s < nst) { stage(s, breg[1]); ++s; }
        if (s < nst) { stage(s, breg[2]); ++s; }
#undef LX_TERM
#undef LX_SGB
    }

    // epilogue: acc[i][r] = out[32 (5 wm + i) + (r&3) + 8 (r>>2) + 4 lh][n0 + 32 wn + li]
    const bool direct = (gridDim.z == 1);
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * N;
    const int ldd = direct ? ldy : N;
    const int n = n0 + 32 * wn + li;
    if (n < N) {
        const float bv = direct ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * (5 * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float v = acc[i][r] + bv;
                    if (direct && relu) v = fmaxf(v, 0.f);
                    dst[(size_t)m * ldd + n] = v;
                }
            }
        }
    }
}

// y[m][n] = act(bias[n] + sum_z ws[z][m][n]) in fixed z order (deterministic); y_rec (optional): the x6 records of y for the next
// layer's GEMM ([N/16][LX_BM][3][16] bf16, N % 16 == 0; rows M .. LX_BM-1 zero).  One thread = 4 consecutive n.
__global__ __launch_bounds__(256)
void splitk_reduce_x6_kernel(const float* __restrict__ ws, const float* __restrict__ bias, float* __restrict__ y, int ldy,
                             unsigned char* __restrict__ y_rec, int M, int N, int splits, int relu)
{
    const int q4 = N >> 2;
    const size_t total = (size_t)(y_rec != nullptr ? LX_BM : M) * q4, plane = (size_t)M * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int m = (int)(idx / q4), n = (int)(idx % q4) * 4;
        if (m >= M) {                                    // padding rows of the record array
            unsigned char* p = y_rec + ((size_t)(n >> 4) * LX_BM + m) * 96 + (n & 15) * 2;
            const uint2 z = {0u, 0u};
            *reinterpret_cast<uint2*>(p) = z;
            *reinterpret_cast<uint2*>(p + 32) = z;
            *reinterpret_cast<uint2*>(p + 64) = z;
            continue;
        }
        const size_t off = (size_t)m * N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + off);
        for (int z = 1; z < splits; ++z) v = v + *reinterpret_cast<const f32x4*>(ws + (size_t)z * plane + off);
        v = v + *reinterpret_cast<const f32x4*>(bias + n);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (y != nullptr) *reinterpret_cast<f32x4*>(y + (size_t)m * ldy + n) = v;
        if (y_rec != nullptr) {
            lx_u16 hi[4], mid[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) lx_split3(v[j], hi[j], mid[j], lo[j]);
            unsigned char* p = y_rec + ((size_t)(n >> 4) * LX_BM + m) * 96 + (n & 15) * 2;
            uint2 ph, pm, pl;
            ph.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);   ph.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
            pm.x = (unsigned)mid[0] | ((unsigned)mid[1] << 16); pm.y = (unsigned)mid[2] | ((unsigned)mid[3] << 16);
            pl.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16);   pl.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
            *reinterpret_cast<uint2*>(p) = ph;
            *reinterpret_cast<uint2*>(p + 32) = pm;
            *reinterpret_cast<uint2*>(p + 64) = pl;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
struct LxPlan { int nblocks, splits, chunks_per_split; };

static LxPlan plan_linear_x6(int N, int K)
{
    LxPlan p;
    p.nblocks = cdiv(N, LX_BN);
    const int chunks = K / 16;
    int want = 256 / p.nblocks;                      // one 8-wave block per CU
    int cap = chunks / 16;                           // >= 16 stages (256 k) per split
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 2) want = 2;                          // the fused reduce (bias + ReLU + next layer's records) always runs
    if (want > chunks) want = chunks;
    p.chunks_per_split = cdiv(chunks, want);
    p.splits = cdiv(chunks, p.chunks_per_split);
    return p;
}

bool linear_x6_shape_ok(int M, int N, int K) { return M >= 1 && M <= LX_BM && N >= 4 && N % 4 == 0 && K >= 32 && K % 16 == 0; }

size_t linear_x6_workspace_bytes(int M, int N, int K)
{
    if (!linear_x6_shape_ok(M, N, K)) return 0;
    const LxPlan p = plan_linear_x6(N, K);
    return (size_t)p.splits * M * N * sizeof(float);
}

int launch_split_rows_x6(const float* a, int lda, void* rec, int R, int rows_out, int K, hipStream_t s)
{
    if (R < 1 || rows_out < R || K < 16 || K % 16 != 0 || lda < K || lda % 4 != 0) return FRCNN_EINVAL;
    const size_t total = (size_t)rows_out * (K / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_rows_x6_kernel, dim3(blocks), dim3(256), 0, s, a, lda, static_cast<unsigned char*>(rec), R, rows_out, K);
    return check_launch();
}

// a_rec: records of [M][K] in a 320-row record array (split_rows_x6 with rows_out = 320); w_rec: records of a
// ceil(N / 128) * 128-row array (rows beyond N zero);  y fp32 [M][ldy] and / or y_rec = the records of [M][N] in a 320-row array
// (either may be NULL, not both); ws >= linear_x6_workspace_bytes.
int launch_linear_x6(const void* a_rec, const void* w_rec, const float* bias, float* y, int ldy, void* y_rec, int M, int N, int K,
                     unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (!linear_x6_shape_ok(M, N, K)) return FRCNN_EUNSUPPORTED;
    if ((y == nullptr && y_rec == nullptr) || (y != nullptr && (ldy < N || ldy % 4 != 0)) || (y_rec != nullptr && N % 16 != 0))
        return FRCNN_EINVAL;
    const LxPlan p = plan_linear_x6(N, K);
    if (ws == nullptr || ws_bytes < (size_t)p.splits * M * N * sizeof(float)) return FRCNN_EINVAL;
    FRCNN_MAX_LDS_ONCE(linear_x6_kernel, LX_LDS_BYTES);
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    hipLaunchKernelGGL(linear_x6_kernel, dim3(p.nblocks, 1, p.splits), dim3(LX_THREADS), LX_LDS_BYTES, s,
                       static_cast<const unsigned char*>(a_rec), static_cast<const unsigned char*>(w_rec), bias, y, ldy,
                       static_cast<float*>(ws), M, N, K / 16, p.chunks_per_split, relu, cdiv(N, LX_BN) * LX_BN);
    int rc = check_launch();
    if (rc) return rc;
    const size_t total = (size_t)(y_rec != nullptr ? LX_BM : M) * (N / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_x6_kernel, dim3(blocks), dim3(256), 0, s, static_cast<const float*>(ws), bias, y, ldy,
                       static_cast<unsigned char*>(y_rec), M, N, p.splits, relu);
    return check_launch();
}

}  // namespace frcnn
