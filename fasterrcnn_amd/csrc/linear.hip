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
// 16-k stage, double-buffered.
// The detector GEMMs have M = 300 rows only (one image's RoIs) against K = 25088 / 4096: the
// block tile spans ALL rows (320 x 128) so the 411 MB fc1 weight matrix is streamed exactly
// once, and the grid is filled by deterministic split-K (partials to scratch, fixed-order
// reduce fused with bias+ReLU) -- no atomics, results are run-to-run identical.
#include "common.h"

namespace frcnn {

static constexpr int LLDK = 20;

template <int TM, int TN, int WM, int WN>
struct GemmCfg {
    static constexpr int BM = 32 * TM * WM;
    static constexpr int BN = 32 * TN * WN;
    static constexpr int NA = BM * 4 / 256;
    static constexpr int NB = BN * 4 / 256;
    static constexpr int A_F = BM * LLDK;
    static constexpr int B_F = BN * LLDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_F + B_F) * sizeof(float);
    static_assert((BM * 4) % 256 == 0 && (BN * 4) % 256 == 0, "tile rows must be multiples of 64");
};

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256)
void linear_mfma_kernel(const float* __restrict__ a, int lda, const float* __restrict__ w,
                        const float* __restrict__ bias, float* __restrict__ y, int ldy,
                        float* __restrict__ ws, int M, int N, int K, int stages_per_split, int relu)
{
    using C = GemmCfg<TM, TN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const at0 = smem;
    float* const bt0 = smem + 2 * C::A_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * C::BM, n0 = blockIdx.x * C::BN;

    const int total_stages = K >> 4;
    const int st_begin = blockIdx.z * stages_per_split;
    int st_end = st_begin + stages_per_split;
    if (st_end > total_stages) st_end = total_stages;
    const int nst = st_end - st_begin;

    size_t a_src[C::NA]; bool a_ok[C::NA]; int a_dst[C::NA];
#pragma unroll
    for (int it = 0; it < C::NA; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        a_ok[it] = (m0 + row) < M;
        a_src[it] = (size_t)(m0 + row) * lda + 4 * p;
        a_dst[it] = row * LLDK + 4 * p;
    }
    size_t b_src[C::NB]; int b_dst[C::NB];
#pragma unroll
    for (int it = 0; it < C::NB; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        b_src[it] = (size_t)(n0 + row) * K + 4 * p;
        b_dst[it] = row * LLDK + 4 * p;
    }

    f32x4 areg[C::NA], breg[C::NB];
    auto load_tiles = [&](int stage) {
        const int k0 = stage << 4;
#pragma unroll
        for (int it = 0; it < C::NA; ++it) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (a_ok[it]) v = *reinterpret_cast<const f32x4*>(a + a_src[it] + k0);
            areg[it] = v;
        }
#pragma unroll
        for (int it = 0; it < C::NB; ++it)
            breg[it] = *reinterpret_cast<const f32x4*>(w + b_src[it] + k0);
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int it = 0; it < C::NA; ++it)
            *reinterpret_cast<f32x4*>(at0 + buf * C::A_F + a_dst[it]) = areg[it];
#pragma unroll
        for (int it = 0; it < C::NB; ++it)
            *reinterpret_cast<f32x4*>(bt0 + buf * C::B_F + b_dst[it]) = breg[it];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nst > 0) {
        load_tiles(st_begin);
        store_tiles(0);
    }
    __syncthreads();

    const int a_base = (32 * TM * wm + li) * LLDK + 4 * lh;
    const int b_base = (32 * TN * wn + li) * LLDK + 4 * lh;

    for (int s = 0; s < nst; ++s) {
        const bool has_next = (s + 1) < nst;
        if (has_next) load_tiles(st_begin + s + 1);
        const float* at = at0 + (s & 1) * C::A_F + a_base;
        const float* bt = bt0 + (s & 1) * C::B_F + b_base;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(at + i * 32 * LLDK + 8 * g);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * LLDK + 8 * g);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
        }
        if (has_next) store_tiles((s + 1) & 1);
        __syncthreads();
    }

    // epilogue: acc[i][j][r] = out[m0 + 32(TM wm + i) + (r&3)+8(r>>2)+4lh][n0 + 32(TN wn + j) + li]
    const bool direct = (gridDim.z == 1);
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * N;
    const int ldd = direct ? ldy : N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + 32 * (TN * wn + j) + li;
        if (n >= N) continue;
        const float bv = direct ? bias[n] : 0.f;
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
    const float v = lane < ncls ? x[(size_t)row * ldx + lane] : -INFINITY;
    float mx = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = lane < ncls ? expf(v - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < ncls) y[(size_t)row * ncls + lane] = e / sum;
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
    const float v = lane < ncls ? xr[lane] : -INFINITY;
    float mx = v;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    const float e = lane < ncls ? expf(v - mx) : 0.f;
    float sum = e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (lane < ncls) classes[(size_t)row * ncls + lane] = e / sum;
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
    int want = 256 / (p.mblocks * p.nblocks);       // fill 256 CUs
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
    auto kern = linear_mfma_kernel<TM, TN, WM, WN>;
    static bool attr_set = false;
    if (!attr_set) {
        FRCNN_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        attr_set = true;
    }
    dim3 grid(p.nblocks, p.mblocks, p.splits);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, a, lda, w, bias, y, ldy, ws, M, N, K,
                       p.stages_per_split, relu);
    return check_launch();
}

int launch_linear(const float* a, int lda, const float* w, const float* bias, float* y, int ldy,
                  int M, int N, int K, unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    if (M < 1 || N < 1 || K < 16 || K % 16 != 0 || lda % 4 != 0 || lda < K || ldy < N) return FRCNN_EINVAL;
    const LinearPlan p = plan_linear(M, N, K);
    const size_t need = p.splits > 1 ? (size_t)p.splits * M * N * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && ws == nullptr)) return FRCNN_EINVAL;
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    int rc = p.big ? launch_linear_cfg<5, 2, 2, 2>(p, a, lda, w, bias, y, ldy, (float*)ws, M, N, K, relu, s)
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
    if (M < 1 || ncls < 1 || ncls > 64 || ldx < ncls) return FRCNN_EINVAL;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, ldx, y, M, ncls);
    return check_launch();
}

int launch_head_finish(const float* x, int ldx, int M, int ncls, int ndelta, float* classes,
                       float* deltas, hipStream_t s)
{
    if (M < 1 || ncls < 1 || ncls > 64 || ndelta < 0 || ldx < ncls + ndelta) return FRCNN_EINVAL;
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
