// gemm_x3t.hip -- batched  C_b[m][n] = sum_k A_b[m][k] * B_b[n][k]  in the "f32x3" arithmetic on the fp16 matrix pipe: HALF the
// matrix instructions of the f32x6 form (csrc/gemm_x6t.hip) at the accuracy of a float32 GEMM.  Replaces the multiply-accumulate of
// the reference's fc1 / fc2 (pytorch/FasterRCNN/models/vgg16.py:129-133, F.linear in float32), of the 512-channel 3x3 convolutions
// (models/vgg16.py:89-96, models/rpn.py:88: the 16 Winograd position GEMMs, csrc/wino_x3.hip) and of ResNet's layer4 convolutions
// (models/resnet.py:109-118 over torchvision's Bottleneck) -- cuDNN / cuBLAS float32 there.
//
// Arithmetic.  Every operand ROW r carries a power-of-two scale 2^e(r) that puts its largest magnitude into [2^14, 2^15); the scaled
// float32 value is split into two fp16 terms
//     x 2^e = hi + lo + d,     hi = fp16(x 2^e),  lo = fp16(x 2^e - hi),     |d| <= 2^-22 |x 2^e|   (2^-25 absolute below 2^-3)
// and a product is the sum of the three largest partial products hi*hi + hi*lo + lo*hi on v_mfma_f32_32x32x16_f16 with float32
// accumulation (an 11-bit x 11-bit product is exact in float32); the dropped lo*lo term is <= 2^-22 |a b|.  The epilogue multiplies by
// 2^-e(m) and 2^-e(n) (exact).  So an operand is held to 22-23 bits instead of float32's 24: a relative perturbation of <= 2^-22 per
// factor with random sign, i.e. ~2^-23.7 rms -- below what the float32 ACCUMULATION of either matrix pipe adds for K >= 64 (measured
// against float64: tests/test_gemm_x3t_gpu.py holds the result to the exact-f32 kernel's error, as the x6 kernels are).  The row
// scales make that bound relative to each ROW's largest element, whatever the tensor's dynamic range (fp16 alone spans 2^-14 .. 2^16).
//
// x3t record layout of a row-major matrix X[R][K] (K % 16 == 0), rows padded to RBT row blocks of 32:
//     [K/16 chunks][RBT row blocks][2 terms hi, lo][1024 B],   1024 B = [k-half 2][row 32][8 fp16]
// = the x6t layout with two terms: a 1 KB piece is the register image of one MFMA operand fragment (lane l: row l & 31, k = 8 (l >> 5)
// .. + 7 at byte 16 l), staged by LDS-DMA and read back with ds_read_b128 at base + 16 * lane.  Scales: float32 2^-e per row.
//
// Kernel: the tiling, LDS-DMA staging, XCD-aware block order, swapped MFMA operands (16-byte epilogue accesses) and deterministic
// split-K of gemm_x6t_kernel.  Per 16-k stage a wave of the 320 x 256 tile issues 30 MFMAs against 14 ds_read_b128 and 5 LDS-DMA
// pieces; the lo*hi product of a stage is issued AFTER the stage's barrier, in the shadow of the next stage's first fragment reads
// (those go to the hi-A and lo-B registers, which the deferred product does not use).
#include "x3t.h"

namespace frcnn {

typedef _Float16 hx_f16x8 __attribute__((ext_vector_type(8)));

#ifndef HX_ABLATE
#define HX_ABLATE 0     // timing experiments only (tools/build_ablate.sh, results wrong): 1 no LDS-DMA after the prologue, 2 no MFMAs, 4 no epilogue stores, 16 no fragment reads after the first stage
#endif
// One wave per row: max |a[row][0 .. K)| -> inv_scale[row] = 2^-e (rows R .. rows_padded - 1: 1).  [batch][R][ld] float32.
__global__ __launch_bounds__(256)
void rows_scale_x3t_kernel(const float* __restrict__ a, int lda, size_t a_batch, float* __restrict__ inv_scale, int R, int rows_padded,
                           int K, int batches)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long long)batches * rows_padded) return;
    const int row = (int)(wave % rows_padded), batch = (int)(wave / rows_padded);
    float mx = 0.f;
    if (row < R) {
        const float* src = a + (size_t)batch * a_batch + (size_t)row * lda;
        for (int k = 4 * lane; k < K; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + k);
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float mult, inv;
    hx_row_scale(mx, mult, inv);
    if (lane == 0) inv_scale[(size_t)batch * rows_padded + row] = inv;
}

// [batch][R][ld] float32 -> x3t records [batch][K/16][rbt][2][1 KB] with the rows scaled by 1 / inv_scale[batch][row]; rows R .. are zero.
// One wave = one (batch, chunk, row block): lane l = row (l & 31), k-half (l >> 5); its two stores are whole 1 KB pieces.
__global__ __launch_bounds__(256)
void split_rows_x3t_kernel(const float* __restrict__ a, int lda, size_t a_batch, const float* __restrict__ inv_scale,
                           unsigned char* __restrict__ rec, int R, int rbt, int K16, int batches)
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
        const float inv = inv_scale[(size_t)batch * rbt * 32 + row];
        const float mult = hx_mult_of_inv(inv);
        const float* src = a + (size_t)batch * a_batch + (size_t)row * lda + k;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = v0[j] * mult; v[4 + j] = v1[j] * mult; }
    }
    uint4 ph, pl;
    hx_split8(v, ph, pl);
    unsigned char* dst = rec + (size_t)wave * HX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + HX_PIECE) = pl;
}

// NHWC [N][H][W][C] -> x3t records of the [N Ho Wo][C] matrix of its pixels taken with `stride` (a 1x1 convolution's A operand), each row
// scaled by the power of two that `cmax` (max_c |x| per INPUT pixel: launch_pixel_absmax) gives; inv[row] = 2^-e.  Waves as
// split_pixels_x6t_kernel (csrc/gemm_x6t.hip).
__global__ __launch_bounds__(256)
void split_pixels_x3t_kernel(const float* __restrict__ x, const float* __restrict__ cmax, unsigned char* __restrict__ rec,
                             float* __restrict__ inv_out, int H, int W, int Ho, int Wo, int C, int stride, int R, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long long)K16 * rbt) return;
    const int chunk = (int)(wave % K16), rb = (int)(wave / K16);
    const int row = rb * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mult = 1.f, inv = 1.f;
    if (row < R) {
        const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        const size_t pix = ((size_t)n * H + (size_t)oy * stride) * W + (size_t)ox * stride;
        hx_row_scale(cmax[pix], mult, inv);
        const float* src = x + pix * C + k;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = v0[j] * mult; v[4 + j] = v1[j] * mult; }
    }
    if (chunk == 0 && lane < 32) inv_out[row] = inv;
    uint4 ph, pl;
    hx_split8(v, ph, pl);
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + HX_PIECE) = pl;
}

// The same records WITHOUT the channel-maximum pass in front of it (round 6: the per-RoI head of the ResNets ran pixel_absmax_kernel +
// split_pixels_x3t_kernel per 1x1 convolution: 7 + 7 launches of an image, the tensor read twice from two launches).  One block = one
// row block of 32 output pixels: its four waves first reduce max_c |x| of eight rows each (coalesced 1 KB reads per row, the eight rows'
// loads independent), then write the block's K16 chunks as split_pixels_x3t_kernel does, the rows now coming from the caches.  max is
// exact, so the scales -- and with them every record byte and inv[] -- are the two-launch form's.
__global__ __launch_bounds__(256)
void split_pixels_x3t_max_kernel(const float* __restrict__ x, unsigned char* __restrict__ rec, float* __restrict__ inv_out, int H, int W, int Ho,
                                 int Wo, int C, int stride, int R, int rbt, int K16)
{
    __shared__ float smax[32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rb = blockIdx.x;
    auto pixel_of = [&](int row) -> size_t {
        const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        return ((size_t)n * H + (size_t)oy * stride) * W + (size_t)ox * stride;
    };
    {
        const float* src[8];
        float mx[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int row = rb * 32 + wave * 8 + r;
            src[r] = row < R ? x + pixel_of(row) * C : nullptr;
            mx[r] = 0.f;
        }
        for (int c = 4 * lane; c < C; c += 256) {
            f32x4 v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = src[r] ? *reinterpret_cast<const f32x4*>(src[r] + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 8; ++r)
                mx[r] = fmaxf(fmaxf(mx[r], fmaxf(fabsf(v[r][0]), fabsf(v[r][1]))), fmaxf(fabsf(v[r][2]), fabsf(v[r][3])));
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float m = mx[r];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
            if (lane == 0) smax[wave * 8 + r] = m;
        }
    }
    __syncthreads();
    const int row = rb * 32 + (lane & 31);
    float mult = 1.f, inv = 1.f;
    const float* src = nullptr;
    if (row < R) {
        hx_row_scale(smax[lane & 31], mult, inv);
        src = x + pixel_of(row) * C + 8 * (lane >> 5);
    }
    if (wave == 0 && lane < 32) inv_out[row] = inv;
    for (int chunk = wave; chunk < K16; chunk += 4) {
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (src) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + chunk * 16), v1 = *reinterpret_cast<const f32x4*>(src + chunk * 16 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = v0[j] * mult; v[4 + j] = v1[j] * mult; }
        }
        uint4 ph, pl;
        hx_split8(v, ph, pl);
        unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
        *reinterpret_cast<uint4*>(dst) = ph;
        *reinterpret_cast<uint4*>(dst + HX_PIECE) = pl;
    }
}

// im2col + split for a 3x3 convolution with padding 1 and stride 1 / 2 (split_patches3x3_x6t_kernel's rows and columns); a row's scale
// comes from the largest channel maximum among its (up to nine) patch pixels.
__global__ __launch_bounds__(256)
void split_patches3x3_x3t_kernel(const float* __restrict__ x, const float* __restrict__ cmax, unsigned char* __restrict__ rec,
                                 float* __restrict__ inv_out, int H, int W, int Ho, int Wo, int C, int stride, int R, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= (long long)K16 * rbt) return;
    const int chunk = (int)(wave % K16), rb = (int)(wave / K16);
    const int row = rb * 32 + (lane & 31), k = chunk * 16 + 8 * (lane >> 5);
    const int tap = k / C, c = k - tap * C;
    const int tr = tap / 3, ts = tap - 3 * tr;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float mult = 1.f, inv = 1.f;
    if (row < R) {
        const int n = row / (Ho * Wo), rem = row - n * (Ho * Wo);
        const int oy = rem / Wo, ox = rem - oy * Wo;
        float mx = 0.f;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int yy = oy * stride - 1 + a, xx = ox * stride - 1 + b;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) mx = fmaxf(mx, cmax[((size_t)n * H + yy) * W + xx]);
            }
        hx_row_scale(mx, mult, inv);
        const int iy = oy * stride - 1 + tr, ix = ox * stride - 1 + ts;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float* src = x + (((size_t)n * H + iy) * W + ix) * C + c;
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(src), v1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = v0[j] * mult; v[4 + j] = v1[j] * mult; }
        }
    }
    if (chunk == 0 && lane < 32) inv_out[row] = inv;
    uint4 ph, pl;
    hx_split8(v, ph, pl);
    unsigned char* dst = rec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
    *reinterpret_cast<uint4*>(dst) = ph;
    *reinterpret_cast<uint4*>(dst + HX_PIECE) = pl;
}

struct HxParams {
    const unsigned char* a;     // A records: [batch][chunk][a_rbt][2][1 KB]
    const unsigned char* b;     // B records: [batch][chunk][b_rbt][2][1 KB]
    const float* a_inv;         // 2^-e per A row: [batch][32 a_rbt]
    const float* b_inv;         // 2^-e per B row (output column): [batch][32 b_rbt]
    float* c;                   // C [batch][M][ldc]            (splits == 1)
    float* ws;                  // partials [split][batch][M][N] (splits > 1), already un-scaled
    const float* bias;          // per n, may be NULL (splits == 1 only)
    const float* residual;      // [batch][M][ldc] like C, added before the activation, may be NULL (splits == 1 only)
    size_t a_batch, b_batch, c_batch;       // bytes, bytes, floats
    size_t a_inv_batch, b_inv_batch;        // floats (0 = shared by every batch)
    int a_rbt, b_rbt;
    int M, N, ldc;
    int nchunks, chunks_per_split, splits, batches;
    int mtiles, ntiles;
    int relu;
    int total;
};

// NSUB = 16-k chunks per stage (one barrier per stage).  NSUB = 1: three stage buffers, the DMA two stages ahead (a 16-k stage is ~1 us of
// matrix work, about the L2 -> LDS latency of its 36 KB: one stage ahead left the pipe waiting).  NSUB = 2: 32-k stages, two buffers, the
// DMA one (twice as long) stage ahead and half the barriers; needs an even number of chunks in every split.
template <int WTM, int WTN, int WVM, int WVN, int NSUB>
struct HxCfg {
    static constexpr int NW = WVM * WVN, THREADS = 64 * NW;
    static constexpr int ARB = WVM * WTM, BRB = WVN * WTN;
    static constexpr int BM = 32 * ARB, BN = 32 * BRB;
    static constexpr int A_BYTES = ARB * HX_RB, B_BYTES = BRB * HX_RB;
    static constexpr int STAGE1 = A_BYTES + B_BYTES;                   // one chunk
    static constexpr int STAGE = NSUB * STAGE1;
    static constexpr int NBUF = NSUB == 1 ? 3 : 2;
    static constexpr int DIST = NBUF - 1;                              // stages the DMA runs ahead
    static constexpr size_t LDS_BYTES = NBUF * (size_t)STAGE;
    static constexpr int NPA = ARB * 2, NPB = BRB * 2, NP = NPA + NPB;   // 1 KB pieces per chunk
    static constexpr int PPW = (NSUB * NP + NW - 1) / NW;              // pieces per wave and stage
};

typedef __attribute__((address_space(3))) void* hx_lds_ptr;

template <int WTM, int WTN, int WVM, int WVN, int NSUB>
__global__ __launch_bounds__(64 * WVM * WVN, 2)
void gemm_x3t_kernel(const HxParams p)
{
    using C = HxCfg<WTM, WTN, WVM, WVN, NSUB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_hx[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WVN, wn = wave % WVN;

    int bid = blockIdx.x;                 // XCD-major logical order, n-tile fastest (as gemm_x6t_kernel)
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
    const int nst = (c_end - c_begin) / NSUB;                 // stages (the launcher picks NSUB = 2 only for even chunk counts)

    const size_t a_chunk = (size_t)p.a_rbt * HX_RB, b_chunk = (size_t)p.b_rbt * HX_RB;
    const unsigned char* ag = p.a + (size_t)batch * p.a_batch + (size_t)mt * C::A_BYTES + (size_t)c_begin * a_chunk + lane * 16;
    const unsigned char* bg = p.b + (size_t)batch * p.b_batch + (size_t)nt * C::B_BYTES + (size_t)c_begin * b_chunk + lane * 16;

    auto issue_stage = [&](int s, int buf) {
        const unsigned char* as = ag + (size_t)s * NSUB * a_chunk;
        const unsigned char* bs = bg + (size_t)s * NSUB * b_chunk;
        unsigned char* ldsb = smem_hx + buf * C::STAGE;
#pragma unroll
        for (int i = 0; i < C::PPW; ++i) {
            int q = wave + C::NW * i;
            q = q < NSUB * C::NP ? q : NSUB * C::NP - 1;             // a wave past the end repeats the last piece (same bytes, same place)
            const int sub = q / C::NP, r = q - sub * C::NP;
            const unsigned char* src = r < C::NPA ? as + sub * a_chunk + r * HX_PIECE : bs + sub * b_chunk + (r - C::NPA) * HX_PIECE;
            __builtin_amdgcn_global_load_lds(src, (hx_lds_ptr)(ldsb + sub * C::STAGE1 + r * HX_PIECE), 16, 0, 0);
        }
    };
    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    hx_f16x8 ah[WTM], al[WTM], bh[WTN], bl[WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) al[i][e] = (_Float16)0.f;
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) bh[j][e] = (_Float16)0.f;

#ifndef HX_NO_SCHED
#define HX_NO_SCHED 0
#endif
#define HX_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define HX_TERM(A, B)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < WTM; ++i)                                                     \
        _Pragma("unroll") for (int j = 0; j < WTN; ++j)                                                 \
            if (!(HX_ABLATE & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(B[j], A[i], acc[i][j], 0, 0, 0);

#ifdef HX_CLOCKS
    unsigned long long clk0 = 0, real0 = 0, clk1 = 0, real1 = 0;
    const unsigned long long real_entry = __builtin_amdgcn_s_memrealtime();
#endif
    if (nst > 0) {
        issue_stage(0, 0);
        if (C::DIST == 2) issue_stage(nst > 1 ? 1 : 0, 1);
        // wait for stage 0 only: LDS-DMA completes in issue order, so "at most (DIST - 1) PPW loads outstanding" = everything but the newer stages
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((C::DIST - 1) * C::PPW) : "memory");
        __builtin_amdgcn_s_barrier();
        int cur = 0;                          // buffer of stage s; stage s + DIST goes to the buffer stage s - 1 was read from
        const int a_off = wm * WTM * HX_RB + lane * 16;
        const int b_off = C::A_BYTES + wn * WTN * HX_RB + lane * 16;
#ifdef HX_CLOCKS
        clk0 = __builtin_readcyclecounter(); real0 = __builtin_amdgcn_s_memrealtime();
#endif
        for (int s = 0; s < nst; ++s) {
            const int nxt = C::NBUF == 3 ? (cur == 0 ? 2 : cur - 1) : (cur ^ 1);        // (s + DIST) % NBUF
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const unsigned char* at = smem_hx + cur * C::STAGE + sub * C::STAGE1 + a_off;
                const unsigned char* bt = smem_hx + cur * C::STAGE + sub * C::STAGE1 + b_off;
                // first reads of the chunk: hi of A, lo of B (registers the deferred product below does not touch)
                if (!(HX_ABLATE & 16) || s == 0) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) ah[i] = *reinterpret_cast<const hx_f16x8*>(at + i * HX_RB);
#pragma unroll
                for (int j = 0; j < WTN; ++j) bl[j] = *reinterpret_cast<const hx_f16x8*>(bt + j * HX_RB + HX_PIECE);
                }
                // unconditional (the last stages re-load the last one into an idle buffer): the stage is one basic block (see gemm_x6t_kernel)
                if (sub == 0 && !(HX_ABLATE & 1)) issue_stage(s + C::DIST < nst ? s + C::DIST : nst - 1, nxt);
                HX_TERM(al, bh)                                         // lo x hi of the PREVIOUS chunk (zeros before the first)
                HX_TERM(ah, bl)                                         // hi x lo
                if (!(HX_ABLATE & 16) || s == 0) {
#pragma unroll
                for (int j = 0; j < WTN; ++j) bh[j] = *reinterpret_cast<const hx_f16x8*>(bt + j * HX_RB);
#pragma unroll
                for (int i = 0; i < WTM; ++i) al[i] = *reinterpret_cast<const hx_f16x8*>(at + i * HX_RB + HX_PIECE);
                }
                HX_TERM(ah, bh)                                         // hi x hi
#if !HX_NO_SCHED && !HX_ABLATE
                constexpr int NT = WTM * WTN, NF = WTM + WTN, PW = C::PPW;
                HX_SGB(0x100, NF);                                                                          // ah, bl
                if (sub == 0) {
                    _Pragma("unroll") for (int q = 0; q < (PW < NT ? PW : NT); ++q) { HX_SGB(0x008, 1); HX_SGB(0x010, 1); }   // lh(prev) || the DMA pieces
                    if (PW > NT) HX_SGB(0x010, PW - NT);
                    if (NT > PW) HX_SGB(0x008, NT - PW);
                } else {
                    HX_SGB(0x008, NT);                                                                      // lh(prev)
                }
                _Pragma("unroll") for (int q = 0; q < (NF < NT ? NF : NT); ++q) { HX_SGB(0x008, 1); HX_SGB(0x100, 1); }   // hl || bh, al
                if (NF > NT) HX_SGB(0x100, NF - NT);
                if (NT > NF) HX_SGB(0x008, NT - NF);
                HX_SGB(0x008, NT);                                                                          // hh
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
            // stage s+1 has landed (with DIST = 2 the pieces of stage s+2 may still be in flight), this wave's fragment reads of stage s are
            // back (al is not consumed before the next chunk), and after the barrier nobody reads stage s from LDS any more
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((C::DIST - 1) * C::PPW) : "memory");
            __builtin_amdgcn_s_barrier();
            cur = cur == C::NBUF - 1 ? 0 : cur + 1;
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef HX_CLOCKS
        clk1 = __builtin_readcyclecounter(); real1 = __builtin_amdgcn_s_memrealtime();
#endif
        HX_TERM(al, bh)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the clamped re-loads of the last stages
    }
#undef HX_TERM
#undef HX_SGB

    // epilogue (operands swapped as in gemm_x6t_kernel: a lane holds ONE row m and groups of four consecutive columns)
    const bool direct = p.splits == 1;
    float* dst;
    int ldd;
    if (direct) { dst = p.c + (size_t)batch * p.c_batch; ldd = p.ldc; }
    else        { dst = p.ws + ((size_t)split * p.batches + batch) * (size_t)p.M * p.N; ldd = p.N; }
    const int m_base = mt * C::BM + 32 * WTM * wm + (lane & 31);
    const int n_base = nt * C::BN + 32 * WTN * wn + 4 * (lane >> 5);
    const float* ainv = p.a_inv + (size_t)batch * p.a_inv_batch;
    const float* binv = p.b_inv + (size_t)batch * p.b_inv_batch;
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
        const int m = m_base + 32 * i;
        if (m >= p.M) continue;
        const float sa = ainv[m];
        float* row = dst + (size_t)m * ldd;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n_base + 32 * j + 8 * g;
                if (n >= p.N) continue;
                const f32x4 sb = *reinterpret_cast<const f32x4*>(binv + n);
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (acc[i][j][4 * g + e] * sa) * sb[e];
                if (direct && p.bias != nullptr) v = v + *reinterpret_cast<const f32x4*>(p.bias + n);
                if (direct && p.residual != nullptr)
                    v = v + *reinterpret_cast<const f32x4*>(p.residual + (size_t)batch * p.c_batch + (size_t)m * ldd + n);
                if (direct && p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (HX_ABLATE & 4) asm volatile("" ::"v"(v));
                else *reinterpret_cast<f32x4*>(row + n) = v;
            }
        }
    }
#ifdef HX_CLOCKS
    // timing experiment (tools/hx_clocks.py; splits == 1 and a caller-provided ws of 32 B per wave)
    if (lane == 0 && p.ws != nullptr) {
        const unsigned long long real_exit = __builtin_amdgcn_s_memrealtime();
        float* o = p.ws + ((size_t)blockIdx.x * C::NW + wave) * 8;
        o[0] = (float)(clk1 - clk0); o[1] = (float)(real1 - real0); o[2] = (float)(real0 - real_entry); o[3] = (float)(real_exit - real1);
        o[4] = (float)nst; o[5] = (float)(real_entry & 0xFFFFFF); o[6] = (float)(real_exit & 0xFFFFFF); o[7] = 1.0f;
    }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------------
size_t x3t_record_bytes(int rows_padded, int K) { return (size_t)(K / 16) * (rows_padded / 32) * HX_RB; }

int launch_rows_scale_x3t(const float* a, int lda, size_t a_batch_floats, float* inv_scale, int R, int rows_padded, int K, int batches,
                          hipStream_t s)
{
    if (R < 1 || rows_padded < R || rows_padded % 32 != 0 || K < 16 || K % 16 != 0 || lda < K || lda % 4 != 0 || batches < 1)
        return FRCNN_EINVAL;
    const long long waves = (long long)batches * rows_padded;
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(rows_scale_x3t_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, lda, a_batch_floats, inv_scale, R, rows_padded, K,
                       batches);
    return check_launch();
}

int launch_split_rows_x3t(const float* a, int lda, size_t a_batch_floats, const float* inv_scale, void* rec, int R, int rows_padded, int K,
                          int batches, hipStream_t s)
{
    if (R < 1 || rows_padded < R || rows_padded % 32 != 0 || K < 16 || K % 16 != 0 || lda < K || lda % 4 != 0 || batches < 1 || !inv_scale)
        return FRCNN_EINVAL;
    const long long waves = (long long)batches * (K / 16) * (rows_padded / 32);
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(split_rows_x3t_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, lda, a_batch_floats, inv_scale,
                       static_cast<unsigned char*>(rec), R, rows_padded / 32, K / 16, batches);
    return check_launch();
}

// cmax: N * H * W floats (launch_pixel_absmax of x) or NULL (the kernel reduces the maxima of the rows it needs itself); inv: rows_padded floats out
int launch_split_pixels_x3t(const float* x, const float* cmax, void* rec, float* inv, int N, int H, int W, int C, int stride, int rows_padded,
                            hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || C < 16 || C % 16 != 0 || stride < 1 || stride > 2 || rows_padded % 32 != 0 || !inv) return FRCNN_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long R = (long long)N * Ho * Wo;
    if (R > rows_padded || R > 0x7fffffffLL) return FRCNN_EINVAL;
    if (!cmax) {                                                   // no maxima given: the one-launch form finds them itself (same bits)
        hipLaunchKernelGGL(split_pixels_x3t_max_kernel, dim3((unsigned)(rows_padded / 32)), dim3(256), 0, s, x, static_cast<unsigned char*>(rec), inv, H, W,
                           Ho, Wo, C, stride, (int)R, rows_padded / 32, C / 16);
        return check_launch();
    }
    const long long waves = (long long)(C / 16) * (rows_padded / 32);
    hipLaunchKernelGGL(split_pixels_x3t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, cmax, static_cast<unsigned char*>(rec),
                       inv, H, W, Ho, Wo, C, stride, (int)R, rows_padded / 32, C / 16);
    return check_launch();
}

int launch_split_patches3x3_x3t(const float* x, const float* cmax, void* rec, float* inv, int N, int H, int W, int C, int stride,
                                int rows_padded, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || C < 16 || C % 16 != 0 || stride < 1 || stride > 2 || rows_padded % 32 != 0 || !cmax || !inv) return FRCNN_EINVAL;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long long R = (long long)N * Ho * Wo;
    if (R > rows_padded || R > 0x7fffffffLL) return FRCNN_EINVAL;
    const long long waves = (long long)(9 * C / 16) * (rows_padded / 32);
    hipLaunchKernelGGL(split_patches3x3_x3t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, cmax,
                       static_cast<unsigned char*>(rec), inv, H, W, Ho, Wo, C, stride, (int)R, rows_padded / 32, 9 * C / 16);
    return check_launch();
}

// Tile shapes and the cost model of gemm_x6t.hip with half the matrix cycles per stage (1920 / 480).
struct HxPlan { int cfg, mtiles, ntiles, splits, chunks_per_split; };

static HxPlan plan_gemm_x3t(int M, int N, int K, int batches, int tiles_mode = -1)
{
    HxPlan pl;
    const int chunks = K / 16;
    static const int env_force = []() { const char* e = frcnn_knob("FRCNN_HX_CFG"); return e ? atoi(e) : -1; }();   // experiments: 0 / 1
    const int mode = tiles_mode >= 0 ? tiles_mode : gemm_x6t_get_tiles();
    const int force = (env_force >= 0 && tiles_mode < 0) ? env_force : (mode == 1 ? 0 : mode == 2 ? 1 : -1);
    const long long u0 = (long long)cdiv(M, 320) * cdiv(N, 256) * batches;
    const long long u1 = (long long)cdiv(M, 160) * cdiv(N, 128) * batches;
    int splits0 = 1;
    while (u0 * splits0 * 2 <= 256 && chunks / (splits0 * 2) >= 8) splits0 *= 2;
    const double c0 = (double)((u0 * splits0 + 255) / 256) * 1920.0 / splits0 + (splits0 > 1 ? 600.0 : 0.0);
    int splits1 = 1;
    while (u1 * splits1 * 2 <= 256 && chunks / (splits1 * 2) >= 8) splits1 *= 2;
    const double c1 = (double)((u1 * splits1 + 255) / 256) * 480.0 / 0.85 / splits1 + (splits1 > 1 ? 600.0 : 0.0);
    pl.cfg = (force == 0 || force == 1) ? force : (c1 < c0 ? 1 : 0);
    // experiment (make KNOBS=1, FRCNN_HX_BIG_UNSPLIT): where the model picks the 160 x 128 tiles WITHOUT a split reduction, the 320 x 256 tiles without one
    // -- fewer, longer blocks (less chip fill, which images in flight do not need) and the same bits (an output's chunks are summed in order either way).
    // MEASURED, round 6 (tools/exp_r50_big_unsplit.sh, ResNet-50, 4 images in flight, same box): 678-680 -> 682-685 images/sec steady, bursts of 20 unchanged: not taken
    static const bool big_unsplit = frcnn_knob("FRCNN_HX_BIG_UNSPLIT") != nullptr;
    if (big_unsplit && force < 0 && pl.cfg == 1 && splits1 == 1) { pl.cfg = 0; splits0 = 1; }
    int splits = pl.cfg == 1 ? splits1 : splits0;
    if (splits > chunks) splits = chunks;
    pl.mtiles = pl.cfg == 1 ? cdiv(M, 160) : cdiv(M, 320);
    pl.ntiles = pl.cfg == 1 ? cdiv(N, 128) : cdiv(N, 256);
    pl.chunks_per_split = cdiv(chunks, splits);
    pl.splits = cdiv(chunks, pl.chunks_per_split);
    return pl;
}

size_t gemm_x3t_workspace_bytes(int M, int N, int K, int batches)
{
    if (!gemm_x6t_shape_ok(M, N, K, batches)) return 0;
    size_t need = 0;
    for (int mode = 0; mode < 3; ++mode) {
        const HxPlan pl = plan_gemm_x3t(M, N, K, batches, mode);
        const size_t b = pl.splits > 1 ? (size_t)pl.splits * batches * M * N * sizeof(float) : 0;
        if (b > need) need = b;
    }
    return need;
}

// C_b = act(bias + residual_b + A_b B_b^T).  Records as launch_gemm_x6t's (rows padded to 320 / 256), plus the two 2^-e arrays:
// a_inv [batches][a_rows], b_inv [batches][b_rows] with batch strides in FLOATS (0 = shared by every batch).
int launch_gemm_x3t(const void* a_rec, const float* a_inv, int a_rows, size_t a_batch_bytes, size_t a_inv_batch, const void* b_rec,
                    const float* b_inv, int b_rows, size_t b_batch_bytes, size_t b_inv_batch, const float* bias, const float* residual,
                    float* c, int ldc, size_t c_batch_floats, int M, int N, int K, int batches, unsigned flags, void* ws, size_t ws_bytes,
                    hipStream_t s, int tiles_mode)
{
    if (!gemm_x6t_shape_ok(M, N, K, batches)) return FRCNN_EUNSUPPORTED;
    if (!a_rec || !b_rec || !a_inv || !b_inv || !c || a_rows % 320 != 0 || a_rows < M || b_rows % 256 != 0 || b_rows < N || ldc < N ||
        ldc % 4 != 0)
        return FRCNN_EINVAL;
    const HxPlan pl = plan_gemm_x3t(M, N, K, batches, tiles_mode);
    if (pl.splits > 1 && (ws == nullptr || ws_bytes < (size_t)pl.splits * batches * M * N * sizeof(float))) return FRCNN_EINVAL;
    HxParams p;
    p.a = static_cast<const unsigned char*>(a_rec);
    p.b = static_cast<const unsigned char*>(b_rec);
    p.a_inv = a_inv; p.b_inv = b_inv;
    p.c = c;
    p.ws = static_cast<float*>(ws);
    p.bias = bias;
    p.residual = residual;
    p.a_batch = a_batch_bytes; p.b_batch = b_batch_bytes; p.c_batch = c_batch_floats;
    p.a_inv_batch = a_inv_batch; p.b_inv_batch = b_inv_batch;
    p.a_rbt = a_rows / 32; p.b_rbt = b_rows / 32;
    p.M = M; p.N = N; p.ldc = ldc;
    p.nchunks = K / 16; p.chunks_per_split = pl.chunks_per_split; p.splits = pl.splits; p.batches = batches;
    p.mtiles = pl.mtiles; p.ntiles = pl.ntiles;
    p.relu = (flags & FRCNN_RELU) ? 1 : 0;
    const long long total = (long long)pl.mtiles * pl.ntiles * batches * pl.splits;
    if (total > 0x7fffffffLL) return FRCNN_EINVAL;
    p.total = (int)total;
    // 32-k stages for LONG reductions whose splits hold an even number of 16-k chunks (fc1: 98 chunks per split, 184 -> 176 us; the 16 / 32
    // chunk reductions of the Winograd layers and fc2 measure the same either way and keep the three-buffer form), 16-k stages otherwise
    static const int env_nsub = []() { const char* e = frcnn_knob("FRCNN_HX_NSUB"); return e ? atoi(e) : 0; }();     // experiments: 1 / 2
    const int last = p.nchunks - (pl.splits - 1) * pl.chunks_per_split;
    const bool even = pl.chunks_per_split % 2 == 0 && last % 2 == 0;
    const int nsub = !even ? 1 : (env_nsub == 1 || env_nsub == 2) ? env_nsub : (pl.chunks_per_split >= 64 ? 2 : 1);
#define HX_LAUNCH(TM, TN, VM, VN, NS)                                                                          \
    do {                                                                                                      \
        using C = HxCfg<TM, TN, VM, VN, NS>;                                                                    \
        auto kern = gemm_x3t_kernel<TM, TN, VM, VN, NS>;                                                        \
        FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);                                                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(C::THREADS), C::LDS_BYTES, s, p);                  \
    } while (0)
    if (pl.cfg == 1) { if (nsub == 2) HX_LAUNCH(5, 1, 1, 4, 2); else HX_LAUNCH(5, 1, 1, 4, 1); }
    else             { if (nsub == 2) HX_LAUNCH(5, 2, 2, 4, 2); else HX_LAUNCH(5, 2, 2, 4, 1); }
#undef HX_LAUNCH
    int rc = check_launch();
    if (rc || pl.splits == 1) return rc;
    return launch_gemm_x6t_reduce(static_cast<const float*>(ws), bias, residual, c, ldc, c_batch_floats, M, N, batches, pl.splits, p.relu, s);
}

}  // namespace frcnn
