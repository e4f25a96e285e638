// gemm_x6t.hip -- batched  C_b[m][n] = sum_k A_b[m][k] * B_b[n][k]  in the "f32x6" arithmetic on the bf16 matrix pipe, operands as
// TILE records ("x6t"), staged by LDS-DMA.  Round 3: the GEMM of the x6 Winograd layers (csrc/wino_x6.hip: the 512-channel
// convolutions of models/vgg16.py:89-96 and the RPN trunk models/rpn.py:88) and of fc1 / fc2 (models/vgg16.py:129-133).
//
// Arithmetic: every float32 operand is split EXACTLY into three bfloat16 terms
// x = hi + mid + lo (residuals formed in float32); a product is the sum of the six largest bf16 x bf16 partial products
// (hh, hl, lh, hm, mh, mm) on v_mfma_f32_32x32x16_bf16 with float32 accumulation; the three dropped terms are <= 2^-24 |a b| each.
//
// x6t record layout of a row-major matrix X[R][K] (K % 16 == 0), rows padded to RBT row blocks of 32:
//     [K/16 chunks][RBT row blocks][3 terms hi, mid, lo][1024 B],   1024 B = [k-half 2][row 32][8 bf16]
// i.e. the 1 KB piece of (chunk, row block, term) is EXACTLY the register image of one MFMA operand fragment of a 32-row tile: lane l
// holds row (l & 31), k = 8 (l >> 5) .. + 7, at byte 16 l.  Consequences, all by construction:
//   * global -> LDS staging is a pure linear copy of contiguous bytes (a block's A tile per 16-k step: RB_A x 3 KB in one run), so
//     it is done by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no staging registers, no ds_write, no VALU);
//   * a fragment read is ds_read_b128 at base + 16 * lane: bank-conflict free without padding or swizzle;
//   * the producers (csrc/wino_x6.hip's input transform, split_rows_x6t_kernel) write whole 1 KB pieces per wave store.
//
// Tiling <WTM, WTN, WVM, WVN>: block = WVM x WVN waves, wave tile = WTM x WTN MFMA tiles of 32 x 32.  The main shape is
// <5, 2, 2, 4>: block tile 320 x 256, 8 waves (two per SIMD), 10 accumulator tiles = 160 AGPRs + 21 fragments (84 VGPRs) per wave.
// (A 4-wave block with 5 x 4 tiles per wave needs 320 accumulator registers: more than the 256 AGPRs an instruction can address, and
// hipcc spills rather than mixing AGPR- and VGPR-resident accumulators.)  Per 16-k stage a wave issues 60 MFMAs (6 products x 10
// tiles) against 21 ds_read_b128 and 7 LDS-DMA pieces: 0.35 LDS reads per MFMA (round 2's 320 x 128 tile: 0.6), and the
// block reads 54 KB from L2 per 3840 matrix-pipe cycles = 14 B/clk/CU (round 2's kernel: 22).
// 320 rows = the 300 RoIs of fc1 / fc2 in one tile; 8 x 320 rows = the 2394 Winograd tiles of a 75 x 125 map, so that
// 8 m-tiles x 2 n-tiles x 16 positions = 256 blocks = ONE block per CU.
// Two LDS stage buffers (2 x 54 KB); the DMA of stage s+1 is issued at the top of stage s and waited for (vmcnt(0)) at the
// stage's single barrier.  The mid x mid product of a stage is issued AFTER that barrier, in the shadow of the next stage's first
// fragment reads, so the matrix pipe does not idle while the first ds_reads of a stage are in flight.
#include "common.h"

namespace frcnn {

typedef __bf16 gx_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short gx_u16;

#ifndef GX_ABLATE
#define GX_ABLATE 0     // timing experiments only (tools/build_ablate.sh, results wrong): 1 no LDS-DMA after the prologue, 2 no MFMAs / fragment reads, 4 no epilogue stores, 8 barrier without the DMA wait
#endif
static constexpr int GX_PIECE = 1024;                 // bytes of one (chunk, row block, term) piece
static constexpr int GX_RB = 3 * GX_PIECE;            // bytes of one (chunk, row block)

__device__ __forceinline__ gx_u16 gx_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (gx_u16)(u >> 16);
}
__device__ __forceinline__ float gx_bf16_f32(gx_u16 h) { return __uint_as_float((unsigned)h << 16); }

// x = hi + mid + lo exactly (barring overflow / subnormal tails)
__device__ __forceinline__ void gx_split3(float x, gx_u16& hi, gx_u16& mid, gx_u16& lo)
{
    hi = gx_bf16_rne(x);
    const float r1 = x - gx_bf16_f32(hi);
    mid = gx_bf16_rne(r1);
    const float r2 = r1 - gx_bf16_f32(mid);
    lo = gx_bf16_rne(r2);
}

// 8 consecutive k of one row -> the three 16-byte pieces of its record slot
__device__ __forceinline__ void gx_split8(const float (&v)[8], uint4& ph, uint4& pm, uint4& pl)
{
    gx_u16 hi[8], mid[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) gx_split3(v[j], hi[j], mid[j], lo[j]);
    ph.x = (unsigned)hi[0] | ((unsigned)hi[1] << 16);   ph.y = (unsigned)hi[2] | ((unsigned)hi[3] << 16);
    ph.z = (unsigned)hi[4] | ((unsigned)hi[5] << 16);   ph.w = (unsigned)hi[6] | ((unsigned)hi[7] << 16);
    pm.x = (unsigned)mid[0] | ((unsigned)mid[1] << 16); pm.y = (unsigned)mid[2] | ((unsigned)mid[3] << 16);
    pm.z = (unsigned)mid[4] | ((unsigned)mid[5] << 16); pm.w = (unsigned)mid[6] | ((unsigned)mid[7] << 16);
    pl.x = (unsigned)lo[0] | ((unsigned)lo[1] << 16);   pl.y = (unsigned)lo[2] | ((unsigned)lo[3] << 16);
    pl.z = (unsigned)lo[4] | ((unsigned)lo[5] << 16);   pl.w = (unsigned)lo[6] | ((unsigned)lo[7] << 16);
}

// [batch][R][ld] float32 (K used columns) -> x6t records [batch][K/16][rbt][3][1 KB]; rows R .. 32 rbt - 1 are zero.
// One wave = one (batch, chunk, row block): lane l = row (l & 31), k-half (l >> 5); its three stores are whole 1 KB pieces.
__global__ __launch_bounds__(256)
void split_rows_x6t_kernel(const float* __restrict__ a, int lda, size_t a_batch, unsigned char* __restrict__ rec, int R, int rbt,
                           int K16, int batches)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long total = (long long)batches * K16 * rbt;
    if (wave >= total) return;
    const int rb = (int)(wave % rbt);
    const long long t = wave / rbt;
    const int chunk = (int)(t % K16), batch = (int)(t / K16);
    const int row = rb * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < R) {
        const float* src = a + (size_t)batch * a_batch + (size_t)row * lda + k;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    }
    uint4 ph, pm, pl;
    gx_split8(v, ph, pm, pl);
    unsigned char* dst = rec + (size_t)wave * GX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + GX_PIECE) = pm;
    *reinterpret_cast<uint4*>(dst + 2 * GX_PIECE) = pl;
}

// NHWC [N][H][W][C] -> x6t records of the [N Ho Wo][C] matrix of its pixels taken with `stride` in y and x (a 1x1 convolution's A
// operand; stride 2 = the downsample / strided 1x1 convolutions of models/resnet.py's bottlenecks).  Waves as split_rows_x6t_kernel;
// consecutive waves = consecutive chunks of one row block, so a pixel's channels are read in 64-byte runs.
__global__ __launch_bounds__(256)
void split_pixels_x6t_kernel(const float* __restrict__ x, unsigned char* __restrict__ rec, int H, int W, int Ho, int Wo, int C, int stride,
                             int R, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long long)K16 * rbt) return;
    const int chunk = (int)(wave % K16), rb = (int)(wave / K16);
    const int row = rb * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < R) {
        const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const float* src = x + (((size_t)n * H + (size_t)oy * stride) * W + (size_t)ox * stride) * C + k;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
    }
    uint4 ph, pm, pl;
    gx_split8(v, ph, pm, pl);
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * GX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + GX_PIECE) = pm;
    *reinterpret_cast<uint4*>(dst + 2 * GX_PIECE) = pl;
}

// im2col + split for a 3x3 convolution with padding 1 and stride 1 / 2: row = output pixel (n, oy, ox), column k = tap * C + c with
// tap = 3 r + s reading x[n][oy * stride - 1 + r][ox * stride - 1 + s][c] (zero outside the map).  C % 16 == 0, so a 16-k chunk
// never straddles two taps.  Waves as split_pixels_x6t_kernel.
__global__ __launch_bounds__(256)
void split_patches3x3_x6t_kernel(const float* __restrict__ x, unsigned char* __restrict__ rec, int H, int W, int Ho, int Wo, int C, int stride,
                                 int R, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long long)K16 * rbt) return;
    const int chunk = (int)(wave % K16), rb = (int)(wave / K16);
    const int row = rb * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5);
    const int tap = k / C, c = k - tap * C;
    const int tr = tap / 3, ts = tap - 3 * tr;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (row < R) {
        const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const int iy = oy * stride - 1 + tr, ix = ox * stride - 1 + ts;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float* src = x + (((size_t)n * H + iy) * W + ix) * C + c;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = v0[j]; v[4 + j] = v1[j]; }
        }
    }
    uint4 ph, pm, pl;
    gx_split8(v, ph, pm, pl);
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * GX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + GX_PIECE) = pm;
    *reinterpret_cast<uint4*>(dst + 2 * GX_PIECE) = pl;
}

struct GxParams {
    const unsigned char* a;     // A records: [batch][chunk][a_rbt][3][1 KB]
    const unsigned char* b;     // B records: [batch][chunk][b_rbt][3][1 KB]
    float* c;                   // C [batch][M][ldc]            (splits == 1)
    float* ws;                  // partials [split][batch][M][N] (splits > 1)
    const float* bias;          // per n, may be NULL (splits == 1 only)
    const float* residual;      // [batch][M][ldc] like C, added before the activation, may be NULL (splits == 1 only)
    size_t a_batch, b_batch, c_batch;       // bytes, bytes, floats
    int a_rbt, b_rbt;           // row blocks per chunk of the record arrays
    int M, N, ldc;              // valid rows / columns, C row stride in floats
    int nchunks, chunks_per_split, splits, batches;
    int mtiles, ntiles;         // block tiles per batch
    int relu;
    int total;                  // mtiles * ntiles * batches * splits
};

template <int WTM, int WTN, int WVM, int WVN>
struct GxCfg {
    static constexpr int NW = WVM * WVN, THREADS = 64 * NW;
    static constexpr int ARB = WVM * WTM, BRB = WVN * WTN;             // row blocks of the block tile
    static constexpr int BM = 32 * ARB, BN = 32 * BRB;
    static constexpr int A_BYTES = ARB * GX_RB, B_BYTES = BRB * GX_RB;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = 2 * (size_t)STAGE;
    static constexpr int NPA = ARB * 3, NPB = BRB * 3, NP = NPA + NPB;   // 1 KB pieces per stage
    static constexpr int PPW = (NP + NW - 1) / NW;                     // pieces per wave (the last one clamped: a duplicate copy)
};

typedef __attribute__((address_space(3))) void* gx_lds_ptr;

template <int WTM, int WTN, int WVM, int WVN>
__global__ __launch_bounds__(64 * WVM * WVN, 2)          // two waves per SIMD: one 8-wave block or two 4-wave blocks per CU
void gemm_x6t_kernel(const GxParams p)
{
    using C = GxCfg<WTM, WTN, WVM, WVN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gx[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    // hardware block b runs on XCD b % 8: the logical index is laid out XCD-major, n-tile fastest, so that the n-tiles that share an A
    // tile and the m-tiles that share a batch's B operand run on ONE XCD's L2
    int bid = blockIdx.x;
    {
        const int q = p.total >> 3, r = p.total & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int nt = bid % p.ntiles;
    int t = bid / p.ntiles;
    const int mt = t % p.mtiles;
    t /= p.mtiles;
    const int split = t % p.splits;
    const int batch = t / p.splits;

    const int c_begin = split * p.chunks_per_split;
    int c_end = c_begin + p.chunks_per_split;
    if (c_end > p.nchunks) c_end = p.nchunks;
    const int nst = c_end - c_begin;

    const size_t a_chunk = (size_t)p.a_rbt * GX_RB, b_chunk = (size_t)p.b_rbt * GX_RB;
    const unsigned char* ag = p.a + (size_t)batch * p.a_batch + (size_t)mt * C::A_BYTES + (size_t)c_begin * a_chunk + lane * 16;
    const unsigned char* bg = p.b + (size_t)batch * p.b_batch + (size_t)nt * C::B_BYTES + (size_t)c_begin * b_chunk + lane * 16;

    // LDS-DMA of one stage: the A run and the B run are NP pieces of 1 KB (one wave instruction each); wave w copies the pieces
    // w, w + NW, ...; a wave whose last index is past the end repeats piece NP - 1 (same bytes to the same place: no branch)
    auto issue_stage = [&](int s, int buf) {
        const unsigned char* as = ag + (size_t)s * a_chunk;
        const unsigned char* bs = bg + (size_t)s * b_chunk;
        unsigned char* ldsb = smem_gx + buf * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
            int q = wave + C::NW * i;
            q = q < C::NP ? q : C::NP - 1;
            const unsigned char* src = q < C::NPA ? as + q * GX_PIECE : bs + (q - C::NPA) * GX_PIECE;
            __builtin_amdgcn_global_load_lds(src, (gx_lds_ptr)(ldsb + q * GX_PIECE), 16, 0, 0);
        }
    };
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gx_bf16x8 ah[WTM], am[WTM], al[WTM], bh[WTN], bm[WTN], bl[WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) am[i][e] = (__bf16)0.f;
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) bm[j][e] = (__bf16)0.f;

#ifndef GX_NO_SCHED
#define GX_NO_SCHED 0   // 1: leave the stage's instruction order to the compiler (experiments)
#endif
#define GX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define GX_TERM(A, B)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < WTN; ++j)                                                 \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(B[j], A[i], acc[i][j], 0, 0, 0);

#ifdef GX_CLOCKS
    unsigned long long clk0 = 0, real0 = 0, clk1 = 0, real1 = 0;
    const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
    if (nst > 0) {
        issue_stage(0, 0);
        __syncthreads();                       // the compiler's barrier fence waits for the DMA (vmcnt(0))
        const int a_off = wm * WTM * GX_RB + lane * 16;
        const int b_off = C::A_BYTES + wn * WTN * GX_RB + lane * 16;
#ifdef GX_CLOCKS
        clk0 = __builtin_readcyclecounter(); real0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int s = 0; s < nst; ++s) {
            const int cur = s & 1;
            // The DMA of stage s+1 is issued UNCONDITIONALLY (the last stage re-loads itself into the idle buffer: nobody reads it) so that
            // the stage is one basic block and the instruction order below can be pinned: a wave that issues its 1 KB pieces back to
            // back stalls ~100 cycles on each while its SIMD partner does the same, and the matrix pipe idles (measured: 4725 shader
            // cycles per stage against 4189 without the DMA); one piece per MFMA keeps the pipe fed from the partner wave.
            if (GX_ABLATE & 2) { if (!(GX_ABLATE & 1)) issue_stage(s + 1 < nst ? s + 1 : s, cur ^ 1); }
            if (!(GX_ABLATE & 2)) {
                const unsigned char* at = smem_gx + cur * C::STAGE + a_off;
                const unsigned char* bt = smem_gx + cur * C::STAGE + b_off;
#pragma unroll
                for (int i = 0; i < WTM; ++i) ah[i] = *reinterpret_cast<const gx_bf16x8*>(at + i * GX_RB);
#pragma unroll
                for (int j = 0; j < WTN; ++j) bh[j] = *reinterpret_cast<const gx_bf16x8*>(bt + j * GX_RB);
                // (in program order AFTER the first fragment reads: an LDS-DMA is an LDS write the scheduler will not move a ds_read across)
                if (!(GX_ABLATE & 1)) issue_stage(s + 1 < nst ? s + 1 : s, cur ^ 1);
                GX_TERM(am, bm)                    // mid x mid of the PREVIOUS stage (zeros before the first), operands still in registers
#pragma unroll
                for (int j = 0; j < WTN; ++j) bl[j] = *reinterpret_cast<const gx_bf16x8*>(bt + j * GX_RB + 2 * GX_PIECE);
#pragma unroll
                for (int i = 0; i < WTM; ++i) al[i] = *reinterpret_cast<const gx_bf16x8*>(at + i * GX_RB + 2 * GX_PIECE);
                GX_TERM(ah, bh)
#pragma unroll
                for (int j = 0; j < WTN; ++j) bm[j] = *reinterpret_cast<const gx_bf16x8*>(bt + j * GX_RB + GX_PIECE);
                GX_TERM(ah, bl)
#pragma unroll
                for (int i = 0; i < WTM; ++i) am[i] = *reinterpret_cast<const gx_bf16x8*>(at + i * GX_RB + GX_PIECE);
                GX_TERM(al, bh)
                GX_TERM(ah, bm)
                GX_TERM(am, bh)
#if !GX_NO_SCHED
                // pinned order (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x010 VMEM): NT = MFMAs of one product
                constexpr int NT = WTM * WTN, NF = WTM + WTN, PW = C::PPW;
                GX_SGB(0x100, NF);                                                        // ah, bh
                _Pragma("unroll") for (int q = 0; q < (PW < NT ? PW : NT); ++q) { GX_SGB(0x008, 1); GX_SGB(0x010, 1); }   // mm || the DMA pieces
                if (PW > NT) GX_SGB(0x010, PW - NT);
                if (NT > PW) GX_SGB(0x008, NT - PW);
                _Pragma("unroll") for (int q = 0; q < (NF < NT ? NF : NT); ++q) { GX_SGB(0x008, 1); GX_SGB(0x100, 1); }   // hh || bl, al
                if (NF > NT) GX_SGB(0x100, NF - NT);
                if (NT > NF) GX_SGB(0x008, NT - NF);
                _Pragma("unroll") for (int q = 0; q < (WTN < NT ? WTN : NT); ++q) { GX_SGB(0x008, 1); GX_SGB(0x100, 1); } // hl || bm
                if (NT > WTN) GX_SGB(0x008, NT - WTN);
                _Pragma("unroll") for (int q = 0; q < (WTM < NT ? WTM : NT); ++q) { GX_SGB(0x008, 1); GX_SGB(0x100, 1); } // lh || am
                if (NT > WTM) GX_SGB(0x008, NT - WTM);
                GX_SGB(0x008, 2 * NT);                                                    // hm, mh
#endif
            }
            __builtin_amdgcn_sched_barrier(0);   // the stage's MFMAs stay ABOVE the barrier: the DMA gets the whole stage to land
            if (GX_ABLATE & 8) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }   // never waits for the DMA
            else __syncthreads();              // stage s+1 has landed (vmcnt(0)) and nobody reads stage s from LDS any more
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef GX_CLOCKS
        clk1 = __builtin_readcyclecounter(); real1 = __builtin_amdgcn_s_memrealtime();
#endif
        GX_TERM(am, bm)
    }
#undef GX_TERM
#undef GX_SGB

    // epilogue.  The MFMA was issued with the operands swapped (B as the row operand), so a lane holds, per tile, ONE row m of C and
    // four groups of four CONSECUTIVE columns: acc[i][j][4 g + e] = C[32 (WTM wm + i) + (lane & 31)][32 (WTN wn + j) + 8 g + 4 (lane >> 5) + e]
    // -> 16-byte stores, four per tile instead of sixteen 4-byte ones (the store tail of a block is issue bound)
    const bool direct = p.splits == 1;
    float* dst;
    int ldd;
    if (direct) { dst = p.c + (size_t)batch * p.c_batch; ldd = p.ldc; }
    else        { dst = p.ws + ((size_t)split * p.batches + batch) * (size_t)p.M * p.N; ldd = p.N; }
    const int m_base = mt * C::BM + 32 * WTM * wm + (lane & 31);
    const int n_base = nt * C::BN + 32 * WTN * wn + 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        const int m = m_base + 32 * i;
        if (m >= p.M) continue;
        float* row = dst + (size_t)m * ldd;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + 32 * j + 8 * g;
                if (n >= p.N) continue;                     // N % 4 == 0: a group of four is inside or outside as a whole
                f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                if (direct && p.bias != nullptr) v = v + *reinterpret_cast<const f32x4*>(p.bias + n);
                if (direct && p.residual != nullptr)
                    v = v + *reinterpret_cast<const f32x4*>(p.residual + (size_t)batch * p.c_batch + (size_t)m * ldd + n);
                if (direct && p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (GX_ABLATE & 4) asm volatile("" ::"v"(v));      // keeps the accumulators live without the store
                else *reinterpret_cast<f32x4*>(row + n) = v;
            }
        }
    }
#ifdef GX_CLOCKS
    // timing experiment (tools/gx_clocks.py; splits == 1 and a caller-provided ws of 32 B per wave): shader cycles and 100 MHz ticks of the
    // K loop, ticks before and after it
    if (lane == 0 && p.ws != nullptr) {
        const unsigned long long real_exit = __builtin_amdgcn_s_memrealtime();
        float* o = p.ws + ((size_t)blockIdx.x * C::NW + wave) * 8;
        o[0] = (float)(clk1 - clk0); o[1] = (float)(real1 - real0); o[2] = (float)(real0 - real_entry); o[3] = (float)(real_exit - real1);
        o[4] = (float)nst; o[5] = (float)(real_entry & 0xFFFFFF); o[6] = (float)(real_exit & 0xFFFFFF); o[7] = 1.0f;
    }
#endif
}

// c[b][m][n] = act(bias[n] + residual[b][m][n] + sum_z ws[z][b][m][n]) in fixed z order (deterministic).  One thread = 4 consecutive n.
__global__ __launch_bounds__(256)
void gemm_x6t_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias, const float* __restrict__ residual,
                            float* __restrict__ c, int ldc, size_t c_batch, int M, int N, int batches, int splits, int relu)
{
    const int q4 = N >> 2;
    const size_t per_batch = (size_t)M * q4, total = per_batch * batches, plane = (size_t)batches * M * N;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int b = (int)(idx / per_batch);
        const size_t rem = idx - (size_t)b * per_batch;
        const int m = (int)(rem / q4), n = (int)(rem % q4) * 4;
        const size_t off = ((size_t)b * M + m) * N + n;
        f32x4 v = *reinterpret_cast<const f32x4*>(ws + off);
        for (int z = 1; z < splits; ++z) v = v + *reinterpret_cast<const f32x4*>(ws + (size_t)z * plane + off);
        if (bias != nullptr) v = v + *reinterpret_cast<const f32x4*>(bias + n);
        if (residual != nullptr) v = v + *reinterpret_cast<const f32x4*>(residual + (size_t)b * c_batch + (size_t)m * ldc + n);
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        *reinterpret_cast<f32x4*>(c + (size_t)b * c_batch + (size_t)m * ldc + n) = v;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
int gemm_x6t_row_tile(int /*M*/) { return 320; }          // rows of an A record array are padded to a multiple of this
int gemm_x6t_col_tile(int /*N*/) { return 256; }          // rows of a B record array (output columns) likewise

size_t x6t_record_bytes(int rows_padded, int K) { return (size_t)(K / 16) * (rows_padded / 32) * GX_RB; }

int launch_split_rows_x6t(const float* a, int lda, size_t a_batch_floats, void* rec, int R, int rows_padded, int K, int batches,
                          hipStream_t s)
{
    if (R < 1 || rows_padded < R || rows_padded % 32 != 0 || K < 16 || K % 16 != 0 || lda < K || lda % 4 != 0 || batches < 1)
        return FRCNN_EINVAL;
    const long long waves = (long long)batches * (K / 16) * (rows_padded / 32);
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(split_rows_x6t_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, lda, a_batch_floats,
                       static_cast<unsigned char*>(rec), R, rows_padded / 32, K / 16, batches);
    return check_launch();
}

int launch_split_pixels_x6t(const float* x, void* rec, int N, int H, int W, int C, int stride, int rows_padded, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || C < 16 || C % 16 != 0 || stride < 1 || stride > 2 || rows_padded % 32 != 0) return FRCNN_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long R = (long long)N * Ho * Wo;
    if (R > rows_padded || R > 0x7fffffffLL) return FRCNN_EINVAL;
    const long long waves = (long long)(C / 16) * (rows_padded / 32);
    hipLaunchKernelGGL(split_pixels_x6t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, static_cast<unsigned char*>(rec), H, W,
                       Ho, Wo, C, stride, (int)R, rows_padded / 32, C / 16);
    return check_launch();
}

int launch_split_patches3x3_x6t(const float* x, void* rec, int N, int H, int W, int C, int stride, int rows_padded, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || C < 16 || C % 16 != 0 || stride < 1 || stride > 2 || rows_padded % 32 != 0) return FRCNN_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;      // (H + 2 - 3) / stride + 1
    const long long R = (long long)N * Ho * Wo;
    if (R > rows_padded || R > 0x7fffffffLL) return FRCNN_EINVAL;
    const long long waves = (long long)(9 * C / 16) * (rows_padded / 32);
    hipLaunchKernelGGL(split_patches3x3_x6t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, static_cast<unsigned char*>(rec), H,
                       W, Ho, Wo, C, stride, (int)R, rows_padded / 32, 9 * C / 16);
    return check_launch();
}

// Two tile shapes: cfg 0 = <5, 2, 2, 4> (320 x 256, 8 waves, one block per CU: the least operand traffic and LDS reads per MFMA);
// cfg 1 = <5, 1, 1, 4> (160 x 128, 4 waves, two blocks per CU: four times the blocks, for problems whose 320 x 256 grid leaves most of
// the chip idle -- the 589-row position GEMMs of a 37 x 62 map -- at twice the L2 operand traffic per MFMA).
struct GxPlan { int cfg, mtiles, ntiles, splits, chunks_per_split; };
static thread_local int g_gx_tiles = 0;
void gemm_x6t_set_tiles(int mode) { g_gx_tiles = mode; }
int gemm_x6t_get_tiles() { return g_gx_tiles; }

static GxPlan plan_gemm_x6t(int M, int N, int K, int batches, int tiles_mode = -1)
{
    GxPlan pl;
    const int chunks = K / 16;
    static const int env_force = []() { const char* e = frcnn_knob("FRCNN_GX_CFG"); return e ? atoi(e) : -1; }();   // experiments: 0 / 1
    const int mode = tiles_mode >= 0 ? tiles_mode : g_gx_tiles;
    const int force = (env_force >= 0 && tiles_mode < 0) ? env_force : (mode == 1 ? 0 : mode == 2 ? 1 : -1);
    // cost model in matrix-pipe cycles per 16-k stage of the busiest CU: cfg 0 runs one block per CU at 3840 cycles per stage; a cfg 1
    // block alone on a CU needs 960, two co-resident ones 1920 for both (0.85: its higher LDS / L2 traffic per MFMA)
    const long long u0 = (long long)cdiv(M, 320) * cdiv(N, 256) * batches;
    const long long u1 = (long long)cdiv(M, 160) * cdiv(N, 128) * batches;
    int splits0 = 1;
    while (u0 * splits0 * 2 <= 256 && chunks / (splits0 * 2) >= 8) splits0 *= 2;     // split K until the grid covers the chip once
    const double c0 = (double)((u0 * splits0 + 255) / 256) * 3840.0 / splits0 + (splits0 > 1 ? 600.0 : 0.0);   // + the reduction pass
    int splits1 = 1;
    while (u1 * splits1 * 2 <= 256 && chunks / (splits1 * 2) >= 8) splits1 *= 2;     // until the grid covers the chip once (256 blocks that
                                                                                     // already do are NOT split: measured, conv5_x 30 vs 50 us)
    const double c1 = (double)((u1 * splits1 + 255) / 256) * 960.0 / 0.85 / splits1 + (splits1 > 1 ? 600.0 : 0.0);
    pl.cfg = (force == 0 || force == 1) ? force : (c1 < c0 ? 1 : 0);
    if (pl.cfg == 1) {
        pl.mtiles = cdiv(M, 160); pl.ntiles = cdiv(N, 128);
        int splits = splits1 > chunks ? chunks : splits1;
        pl.chunks_per_split = cdiv(chunks, splits);
        pl.splits = cdiv(chunks, pl.chunks_per_split);
        return pl;
    }
    pl.mtiles = cdiv(M, 320);
    pl.ntiles = cdiv(N, 256);
    int splits = splits0;
    if (splits > chunks) splits = chunks;
    pl.chunks_per_split = cdiv(chunks, splits);
    pl.splits = cdiv(chunks, pl.chunks_per_split);
    return pl;
}

bool gemm_x6t_shape_ok(int M, int N, int K, int batches)
{
    return M >= 1 && N >= 4 && N % 4 == 0 && K >= 16 && K % 16 == 0 && batches >= 1 &&
           (long long)cdiv(M, 320) * cdiv(N, 256) * batches * (K / 16) < 0x7fffffffLL;
}

size_t gemm_x6t_workspace_bytes(int M, int N, int K, int batches)
{
    if (!gemm_x6t_shape_ok(M, N, K, batches)) return 0;
    size_t need = 0;                       // the largest over the tile modes: a scratch sized once serves every later launch
    for (int mode = 0; mode < 3; ++mode) {
        const GxPlan pl = plan_gemm_x6t(M, N, K, batches, mode);
        const size_t b = pl.splits > 1 ? (size_t)pl.splits * batches * M * N * sizeof(float) : 0;
        if (b > need) need = b;
    }
    return need;
}

// C_b = act(bias + residual_b + A_b B_b^T).  a_rec: records of A [batches][M][K] with rows padded to a_rows (multiple of 320, >= M); b_rec: records of B [batches][N][K] with rows
// padded to b_rows (multiple of 256, >= N); batch strides in BYTES (0 = the operand is shared by every batch).
int launch_gemm_x6t(const void* a_rec, int a_rows, size_t a_batch_bytes, const void* b_rec, int b_rows, size_t b_batch_bytes,
                    const float* bias, const float* residual, float* c, int ldc, size_t c_batch_floats, int M, int N, int K, int batches,
                    unsigned flags, void* ws, size_t ws_bytes, hipStream_t s, int tiles_mode)
{
    if (!gemm_x6t_shape_ok(M, N, K, batches)) return FRCNN_EUNSUPPORTED;
    if (!a_rec || !b_rec || !c || a_rows % 320 != 0 || a_rows < M || b_rows % 256 != 0 || b_rows < N || ldc < N || ldc % 4 != 0)
        return FRCNN_EINVAL;
    const GxPlan pl = plan_gemm_x6t(M, N, K, batches, tiles_mode);
    if (pl.splits > 1 && (ws == nullptr || ws_bytes < (size_t)pl.splits * batches * M * N * sizeof(float))) return FRCNN_EINVAL;
    GxParams p;
    p.a = static_cast<const unsigned char*>(a_rec);
    p.b = static_cast<const unsigned char*>(b_rec);
    p.c = c;
    p.ws = static_cast<float*>(ws);
    p.bias = bias;
    p.residual = residual;
    p.a_batch = a_batch_bytes; p.b_batch = b_batch_bytes; p.c_batch = c_batch_floats;
    p.a_rbt = a_rows / 32; p.b_rbt = b_rows / 32;
    p.M = M; p.N = N; p.ldc = ldc;
    p.nchunks = K / 16; p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits; p.batches = batches;
    p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
    p.relu = (flags & FRCNN_RELU) ? 1 : 0;
    const long long total = (long long)pl.mtiles * pl.ntiles * batches * pl.splits;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    p.total = (int)total;
    if (pl.cfg == 1) {
        using C = GxCfg<5, 1, 1, 4>;
        auto kern = gemm_x6t_kernel<5, 1, 1, 4>;
        FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(C::THREADS), C::LDS_BYTES, s, p);
    } else {
        using C = GxCfg<5, 2, 2, 4>;
        auto kern = gemm_x6t_kernel<5, 2, 2, 4>;
        FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(C::THREADS), C::LDS_BYTES, s, p);
    }
    int rc = check_launch();
    if (rc || pl.splits == 1) return rc;
    return launch_gemm_x6t_reduce(static_cast<const float*>(ws), bias, residual, c, ldc, c_batch_floats, M, N, batches, pl.splits, p.relu, s);
}

int launch_gemm_x6t_reduce(const float* ws, const float* bias, const float* residual, float* c, int ldc, size_t c_batch_floats, int M, int N,
                           int batches, int splits, int relu, hipStream_t s)
{
    const size_t n4 = (size_t)batches * M * (N / 4);
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_x6t_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, bias, residual, c, ldc, c_batch_floats, M, N, batches,
                       splits, relu);
    return check_launch();
}

}  // namespace frcnn
