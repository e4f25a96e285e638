// linear.hip -- dense layers on the exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32).
// Replaces nn.Linear(+ReLU) at models/vgg16.py:106-107,130-132, the detector heads at
// models/detector.py:29-30,76,78 and the two RPN 1x1 convolutions at models/rpn.py:40-41,89-90
// (a 1x1 conv on an NHWC map IS a row-major GEMM).
//
//   y[m][n] = act( bias[n] + sum_k a[m][k] * w[n][k] )        a: [M][lda]  w: [Npad][K]
//
// Both operands are K-contiguous, so they are staged exactly like the conv kernel's tiles:
// [rows][16 k] chunks in LDS, rows padded to 20 floats (conflict-free ds_read_b128), the lane
// half h owning k in [8g+4h, 8g+4h+4) so one b128 read feeds four MFMAs.  One barrier per
// 16-k stage, double-buffered; the K loop is software pipelined (see the comment in the kernel).
// The detector GEMMs have M = 300 rows only (one image's RoIs) against K = 25088 / 4096: the
// block tile spans ALL rows (320 x 128) so the 411 MB fc1 weight matrix is streamed exactly
// once, and the grid is filled by deterministic split-K (partials to scratch, fixed-order
// reduce fused with bias+ReLU) -- no atomics, results are run-to-run identical.
#include "common.h"
#include <cstdlib>

namespace frcnn {

static constexpr int LLDK = 20;

template <int TM, int TN, int WM, int WN>
struct GemmCfg {
    static constexpr int BM = 32 * TM * WM;
    static constexpr int BN = 32 * TN * WN;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int NA = (BM * 4 + THREADS - 1) / THREADS;      // 16-B pieces per thread per stage (surplus ones duplicate the last)
    static constexpr int NB = (BN * 4 + THREADS - 1) / THREADS;
    static constexpr int A_F = BM * LLDK;
    static constexpr int B_F = BN * LLDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_F + B_F) * sizeof(float);
    static_assert(THREADS == 256 || THREADS == 512, "4 or 8 waves per block");
};

// BATCHED: `batches` independent GEMMs of the same shape in one launch (the 16 Winograd positions of csrc/winograd.hip),
// operands of batch z at a + z*a_stride, w + z*w_stride, y + z*y_stride; no split-K, no bias (bias == nullptr).
// The grid is one-dimensional and the block order is XCD-aware: hardware block b runs on XCD b % 8, so the logical
// tile index is laid out XCD-major -- the nblocks column tiles that share an A tile and the neighbouring row tiles
// that share the batch's weights land in the same XCD's L2.
struct BatchGeom { long long a_stride, w_stride, y_stride; int nblocks, mblocks, batches; };

template <int TM, int TN, int WM, int WN, bool BATCHED, int NSETS = 2>
__global__ __launch_bounds__(64 * WM * WN)
void linear_mfma_kernel(const float* __restrict__ a, int lda, const float* __restrict__ w,
                        const float* __restrict__ bias, float* __restrict__ y, int ldy,
                        float* __restrict__ ws, int M, int N, int K, int stages_per_split, int relu, BatchGeom bg)
{
    using C = GemmCfg<TM, TN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const at0 = smem;
    float* const bt0 = smem + 2 * C::A_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    int bx = blockIdx.x, by = blockIdx.y;
    if constexpr (BATCHED) {
        const int total = bg.nblocks * bg.mblocks * bg.batches;
        const int per = total >> 3, rem = total & 7, xcd = bx & 7, j = bx >> 3;
        const int L = xcd < rem ? xcd * (per + 1) + j : rem * (per + 1) + (xcd - rem) * per + j;
        bx = L % bg.nblocks;
        const int t2 = L / bg.nblocks;
        by = t2 % bg.mblocks;
        const int bz = t2 / bg.mblocks;
        a += bz * bg.a_stride;
        w += bz * bg.w_stride;
        y += bz * bg.y_stride;
    }
    const int m0 = by * C::BM, n0 = bx * C::BN;

    const int total_stages = K >> 4;
    const int st_begin = blockIdx.z * stages_per_split;
    int st_end = st_begin + stages_per_split;
    if (st_end > total_stages) st_end = total_stages;
    const int nst = st_end - st_begin;

    // staging addresses: byte offsets from the uniform k-stage base (saddr-form loads).  Rows beyond M re-read row M-1:
    // they only feed output rows that the epilogue never stores, so no predication is needed.
    unsigned a_src[C::NA]; int a_dst[C::NA];
#pragma unroll
    for (int it = 0; it < C::NA; ++it) {
        int q = tid + C::THREADS * it;
        if (q >= C::BM * 4) q = C::BM * 4 - 1;
        const int row = q >> 2, p = q & 3;
        const int gr = (m0 + row) < M ? (m0 + row) : M - 1;
        a_src[it] = (unsigned)(((size_t)(gr - m0) * lda + 4 * p) * sizeof(float));
        a_dst[it] = row * LLDK + 4 * p;
    }
    unsigned b_src[C::NB]; int b_dst[C::NB];
#pragma unroll
    for (int it = 0; it < C::NB; ++it) {
        int q = tid + C::THREADS * it;
        if (q >= C::BN * 4) q = C::BN * 4 - 1;
        const int row = q >> 2, p = q & 3;
        b_src[it] = (unsigned)(((size_t)row * K + 4 * p) * sizeof(float));
        b_dst[it] = row * LLDK + 4 * p;
    }
    const float* const a_blk = a + (size_t)m0 * lda;
    const float* const w_blk = w + (size_t)n0 * K;

    // two register sets: the tile of stage s+1 and the tile of stage s+2 are in flight at the same time (the weights
    // of fc1 stream from HBM, one 2 us stage of prefetch distance is not enough with one wave per SIMD)
    // (NSETS = 1: one tile of prefetch distance, 16 registers fewer -- the batched Winograd GEMMs read L2 / Infinity-Cache
    // resident operands and gain more from a third resident block per CU than from the second tile in flight)
    f32x4 areg[NSETS][C::NA], breg[NSETS][C::NB];
    auto load_tiles = [&](int stage, f32x4 (&ar)[C::NA], f32x4 (&br)[C::NB]) {
        const char* ab = reinterpret_cast<const char*>(a_blk + (stage << 4));
        const char* wb = reinterpret_cast<const char*>(w_blk + (stage << 4));
#pragma unroll
        for (int it = 0; it < C::NA; ++it) ar[it] = *reinterpret_cast<const f32x4*>(ab + a_src[it]);
#pragma unroll
        for (int it = 0; it < C::NB; ++it) br[it] = *reinterpret_cast<const f32x4*>(wb + b_src[it]);
    };
    auto store_tiles = [&](int buf, const f32x4 (&ar)[C::NA], const f32x4 (&br)[C::NB]) {
#pragma unroll
        for (int it = 0; it < C::NA; ++it)
            *reinterpret_cast<f32x4*>(at0 + buf * C::A_F + a_dst[it]) = ar[it];
#pragma unroll
        for (int it = 0; it < C::NB; ++it)
            *reinterpret_cast<f32x4*>(bt0 + buf * C::B_F + b_dst[it]) = br[it];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_base = (32 * TM * wm + li) * LLDK + 4 * lh;
    const int b_base = (32 * TN * wn + li) * LLDK + 4 * lh;

    // Software-pipelined K loop (same schedule as conv3x3_mfma2_kernel, csrc/conv.hip): per 16-k stage
    //   F0(s) in registers | read F1(s) | LDS-write tile s+1 | global-load tile s+2 | MFMAs on F0, one staging
    //   instruction per MFMA | barrier | read F0(s+1) under the MFMAs on F1.
    // Prefetches past the last stage are clamped to it (harmless re-reads / re-writes of unused buffers).
    f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
    if (nst > 0) {
        const int last = st_end - 1;
        load_tiles(st_begin, areg[0], breg[0]);
        store_tiles(0, areg[0], breg[0]);
        load_tiles(st_begin + 1 < st_end ? st_begin + 1 : last, areg[0], breg[0]);     // tile 1 -> set 0
        if constexpr (NSETS == 2)
            load_tiles(st_begin + 2 < st_end ? st_begin + 2 : last, areg[1], breg[1]); // tile 2 -> set 1
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const f32x4*>(at0 + a_base + i * 32 * LLDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = *reinterpret_cast<const f32x4*>(bt0 + b_base + j * 32 * LLDK);
        constexpr int N_RD = TM + TN, N_ST = C::NA + C::NB;
        // stage s: writes the tile of stage s+1 from register set (s & 1), then refills that set with the tile of stage s+3
        auto stage = [&](int s, f32x4 (&ar)[C::NA], f32x4 (&br)[C::NB]) {
            const int cur = s & 1, nxt = cur ^ 1;
            const float* at = at0 + cur * C::A_F + a_base;
            const float* bt = bt0 + cur * C::B_F + b_base;
            // b. second-half fragments
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = *reinterpret_cast<const f32x4*>(at + i * 32 * LLDK + 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * LLDK + 8);
            // c. tile s+1 -> LDS; d. tile s+2 -> registers
            store_tiles(nxt, ar, br);
            {
                const int s3 = st_begin + s + 1 + NSETS;
                load_tiles(s3 < st_end ? s3 : last, ar, br);
            }
            // e. first half
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i][kk], b0[j][kk], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < N_RD; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
            for (int q = 0; q < N_ST; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
            for (int q = 0; q < N_ST; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            // f.
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            // g. first-half fragments of stage s+1
            {
                const float* nat = at0 + nxt * C::A_F + a_base;
                const float* nbt = bt0 + nxt * C::B_F + b_base;
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const f32x4*>(nat + i * 32 * LLDK);
#pragma unroll
                for (int j = 0; j < TN; ++j) b0[j] = *reinterpret_cast<const f32x4*>(nbt + j * 32 * LLDK);
            }
            // h. second half
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i][kk], b1[j][kk], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < N_RD; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr (NSETS == 2) {
            int s = 0;
            for (; s + 1 < nst; s += 2) { stage(s, areg[0], breg[0]); stage(s + 1, areg[1], breg[1]); }
            if (s < nst) stage(s, areg[0], breg[0]);
        } else {
            for (int s = 0; s < nst; ++s) stage(s, areg[0], breg[0]);
        }
    }

    // epilogue: acc[i][j][r] = out[m0 + 32(TM wm + i) + (r&3)+8(r>>2)+4lh][n0 + 32(TN wn + j) + li]
    const bool direct = (gridDim.z == 1);
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * N;
    const int ldd = direct ? ldy : N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + 32 * (TN * wn + j) + li;
        if (n >= N) continue;
        const float bv = (direct && bias != nullptr) ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + 32 * (TM * wm + i) + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < M) {
                    float v = acc[i][j][r] + bv;
                    if (direct && relu) v = fmaxf(v, 0.f);
                    dst[(size_t)m * ldd + n] = v;
                }
            }
        }
    }
}

// y[m][n] = act(bias[n] + sum_z ws[z][m][n]) in fixed z order (deterministic).
__global__ __launch_bounds__(256)
void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                          float* __restrict__ y, int ldy, int M, int N, int splits, int relu)
{
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i % N);
        const size_t m = i / N;
        float v = ws[i];
        for (int z = 1; z < splits; ++z) v += ws[(size_t)z * total + i];
        v += bias[n];
        if (relu) v = fmaxf(v, 0.f);
        y[m * ldy + n] = v;
    }
}

// Row softmax (detector.py:77 F.softmax(dim=1)): one wave per row, ncls <= 64.
__global__ __launch_bounds__(256)
void softmax_rows_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int M, int ncls)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    const float v0 = lane < ncls ? xr[lane] : -INFINITY, v1 = lane + 64 < ncls ? xr[lane + 64] : -INFINITY;     // ncls <= 128
    float mx = fmaxf(v0, v1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e0 = lane < ncls ? expf(v0 - mx) : 0.f, e1 = lane + 64 < ncls ? expf(v1 - mx) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < ncls) y[(size_t)row * ncls + lane] = e0 / sum;
    if (lane + 64 < ncls) y[(size_t)row * ncls + lane + 64] = e1 / sum;
}

// Detector head epilogue: logits [M][ldx] = [cls(ncls) | box deltas(ndelta) | pad] ->
// softmax(cls) (detector.py:77) and a dense copy of the deltas (detector.py:78).  One wave/row.
__global__ __launch_bounds__(256)
void head_finish_kernel(const float* __restrict__ x, int ldx, int M, int ncls, int ndelta,
                        float* __restrict__ classes, float* __restrict__ deltas)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * ldx;
    // lane l holds the classes l and l + 64 (ncls <= 128); max and sum in the order of a 64-wide butterfly over the pairwise combination
    const float v0 = lane < ncls ? xr[lane] : -INFINITY, v1 = lane + 64 < ncls ? xr[lane + 64] : -INFINITY;
    float mx = fmaxf(v0, v1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e0 = lane < ncls ? expf(v0 - mx) : 0.f, e1 = lane + 64 < ncls ? expf(v1 - mx) : 0.f;
    float sum = e0 + e1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < ncls) classes[(size_t)row * ncls + lane] = e0 / sum;
    if (lane + 64 < ncls) classes[(size_t)row * ncls + lane + 64] = e1 / sum;
    for (int j = lane; j < ndelta; j += 64) deltas[(size_t)row * ndelta + j] = xr[ncls + j];
}

__global__ void pack_fc_chw_to_hwc_kernel(const float* __restrict__ w, float* __restrict__ wp,
                                          int out_f, int c, int hw)
{
    const size_t K = (size_t)c * hw;
    const size_t total = (size_t)out_f * K;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t o = i / K, kk = i % K;      // kk = p*c + ch  (destination order)
        const int p = (int)(kk / c), ch = (int)(kk % c);
        wp[i] = w[o * K + (size_t)ch * hw + p];
    }
}

__global__ void pack_stack_rows_kernel(const float* __restrict__ w1, const float* __restrict__ b1, int n1,
                                       const float* __restrict__ w2, const float* __restrict__ b2, int n2,
                                       int k, int n_pad, float* __restrict__ wo, float* __restrict__ bo)
{
    const size_t total = (size_t)n_pad * k;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / k), col = (int)(i % k);
        float v = 0.f;
        if (row < n1) v = w1[(size_t)row * k + col];
        else if (row < n1 + n2) v = w2[(size_t)(row - n1) * k + col];
        wo[i] = v;
        if (col == 0) bo[row] = row < n1 ? b1[row] : (row < n1 + n2 ? b2[row - n1] : 0.f);
    }
}

// ---- host side ------------------------------------------------------------------------------
struct LinearPlan { bool big; int mblocks, nblocks, splits, stages_per_split; };

static LinearPlan plan_linear(int M, int N, int K)
{
    LinearPlan p;
    // "big" = every row of a small-M operand in one block tile (320 x 128): weights stream once.
    p.big = (M <= 320) || (M % 320 == 0);
    const int bm = p.big ? 320 : 128;
    p.mblocks = cdiv(M, bm);
    p.nblocks = cdiv(N, 128);
    const int stages = K / 16;
    static int target = -1;
    if (target < 0) { const char* e = frcnn_knob("FRCNN_LINEAR_BLOCKS_TARGET"); target = e ? atoi(e) : 256; if (target < 1) target = 1; }
    int want = target / (p.mblocks * p.nblocks);    // fill 256 CUs
    int cap = stages / 8;                           // >= 8 stages (128 k) per split
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    p.stages_per_split = cdiv(stages, want);
    p.splits = cdiv(stages, p.stages_per_split);
    return p;
}

size_t linear_workspace_bytes(int M, int N, int K)
{
    if (M < 1 || N < 1 || K < 16) return 0;
    const LinearPlan p = plan_linear(M, N, K);
    return p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
}

template <int TM, int TN, int WM, int WN>
static int launch_linear_cfg(const LinearPlan& p, const float* a, int lda, const float* w,
                             const float* bias, float* y, int ldy, float* ws, int M, int N, int K,
                             int relu, hipStream_t s)
{
    using C = GemmCfg<TM, TN, WM, WN>;
    auto kern = linear_mfma_kernel<TM, TN, WM, WN, false>;
    FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
    dim3 grid(p.nblocks, p.mblocks, p.splits);
    hipLaunchKernelGGL(kern, grid, dim3(C::THREADS), C::LDS_BYTES, s, a, lda, w, bias, y, ldy, ws, M, N, K,
                       p.stages_per_split, relu, BatchGeom{0, 0, 0, 0, 0, 0});
    return check_launch();
}

template <int TM, int TN, int WM, int WN, int NSETS>
static int launch_linear_batched_cfg(const float* a, int lda, const float* w, float* y, int ldy, int M, int N, int K,
                                     const BatchGeom& bg, hipStream_t s)
{
    using C = GemmCfg<TM, TN, WM, WN>;
    auto kern = linear_mfma_kernel<TM, TN, WM, WN, true, NSETS>;
    FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
    const long long total = (long long)bg.nblocks * bg.mblocks * bg.batches;
    if (total < 1 || total > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(C::THREADS), C::LDS_BYTES, s, a, lda, w,
                       (const float*)nullptr, y, ldy, (float*)nullptr, M, N, K, K >> 4, 0, bg);
    return check_launch();
}

// y_z[m][n] = sum_k a_z[m][k] * w_z[n][k] for z < batches; w_z holds N rounded up to the tile's columns (128 or 256) rows.
// Tile of the calling thread's next launches: 64 = 64 x 128 (84 registers, five blocks per CU: lowest latency for one
// image on the chip: never slower than 128 x 128 on the VGG-16 shapes, 20 % faster where 128-row tiles are few -- the
// 37 x 62 maps -- or short -- cin = 128), 128 = 128 x 128 (fewer LDS / L2 operand bytes per MFMA: with many images in flight
// the chip runs at its power limit and the cheaper tile wins, 372 -> 380 img/s).  0 = default (64).  FRCNN_WINO_TILE=64|128
// overrides for experiments; 256-row / 256-column tiles and a second register set of prefetch were measured slower or equal
// (tools/micro/README.md) and are not instantiated.
static thread_local int g_batched_tile = 0;
void linear_batched_set_tile(int rows) { g_batched_tile = rows; }

int launch_linear_batched(const float* a, int lda, size_t a_stride, const float* w, size_t w_stride, float* y, int ldy,
                          size_t y_stride, int M, int N, int K, int batches, hipStream_t s)
{
    if (M < 1 || N < 1 || K < 16 || K % 16 != 0 || lda % 4 != 0 || lda < K || ldy < N || batches < 1) return FRCNN_EINVAL;
    static int env_tile = -1;
    if (env_tile < 0) { const char* e = frcnn_knob("FRCNN_WINO_TILE"); env_tile = e ? atoi(e) : 0; }
    const int tile = env_tile ? env_tile : (g_batched_tile ? g_batched_tile : 64);
    BatchGeom bg{(long long)a_stride, (long long)w_stride, (long long)y_stride, cdiv(N, 128), cdiv(M, 128), batches};
    if (tile == 128) return launch_linear_batched_cfg<2, 2, 2, 2, 1>(a, lda, w, y, ldy, M, N, K, bg, s);
    bg.mblocks = cdiv(M, 64);
    return launch_linear_batched_cfg<1, 2, 2, 2, 1>(a, lda, w, y, ldy, M, N, K, bg, s);
}

int launch_linear(const float* a, int lda, const float* w, const float* bias, float* y, int ldy,
                  int M, int N, int K, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (M < 1 || N < 1 || K < 16 || K % 16 != 0 || lda % 4 != 0 || lda < K || ldy < N) return FRCNN_EINVAL;
    const LinearPlan p = plan_linear(M, N, K);
    const size_t need = p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && ws == nullptr)) return FRCNN_EINVAL;
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    // big: 8 waves (2 x 4) of 5 x 1 MFMA tiles -- two waves per SIMD instead of one wave of 5 x 2 tiles
    int rc = p.big ? launch_linear_cfg<5, 1, 2, 4>(p, a, lda, w, bias, y, ldy, (float*)ws, M, N, K, relu, s)
                   : launch_linear_cfg<2, 2, 2, 2>(p, a, lda, w, bias, y, ldy, (float*)ws, M, N, K, relu, s);
    if (rc != FRCNN_OK) return rc;
    if (p.splits > 1) {
        const size_t total = (size_t)M * N;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, bias,
                           y, ldy, M, N, p.splits, relu);
        rc = check_launch();
    }
    return rc;
}

int launch_softmax_rows(const float* x, int ldx, float* y, int M, int ncls, hipStream_t s)
{
    if (M < 1 || ncls < 1 || ncls > 128 || ldx < ncls) return FRCNN_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, y, M, ncls);
    return check_launch();
}

int launch_head_finish(const float* x, int ldx, int M, int ncls, int ndelta, float* classes,
                       float* deltas, hipStream_t s)
{
    if (M < 1 || ncls < 1 || ncls > 128 || ndelta < 0 || ldx < ncls + ndelta) return FRCNN_EINVAL;
    hipLaunchKernelGGL(head_finish_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, M, ncls, ndelta,
                       classes, deltas);
    return check_launch();
}

int launch_pack_fc_chw_to_hwc(const float* w, float* wp, int out_f, int c, int phw, hipStream_t s)
{
    if (out_f < 1 || c < 1 || phw < 1) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pack_fc_chw_to_hwc_kernel, dim3(8192), dim3(256), 0, s, w, wp, out_f, c, phw);
    return check_launch();
}

int launch_pack_stack_rows(const float* w1, const float* b1, int n1, const float* w2, const float* b2,
                           int n2, int k, int n_pad, float* wo, float* bo, hipStream_t s)
{
    if (n1 < 0 || n2 < 0 || n1 + n2 > n_pad || k < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)n_pad * k;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(pack_stack_rows_kernel, dim3(blocks), dim3(256), 0, s, w1, b1, n1, w2, b2, n2, k,
                       n_pad, wo, bo);
    return check_launch();
}

}  // namespace frcnn
