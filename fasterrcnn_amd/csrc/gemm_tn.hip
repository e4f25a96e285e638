// gemm_tn.hip -- the one matrix kernel of the backward pass (SURVEY.md section 8 row f3;
// reference: total_loss.backward() at models/faster_rcnn.py:355, i.e. autograd's conv2d / linear
// backward on cuDNN / cuBLAS).
//
//   C[m][n] = sum_r A[r][m] * B[r][n]          ("TN": both operands are stored reduction-major)
//
// Every gradient GEMM of the train step has this shape once the small operand is transposed:
//   linear weight gradient  dW[out][in]  = sum_sample dY[sample][out] * X[sample][in]
//   linear data gradient    dX[sample][in] = sum_out dY^T[out][sample] * W[out][in]
//   conv3x3 weight gradient dW[tap][co][ci] = sum_pixel dZ[pixel][co] * X[pixel + tap offset][ci]
// The conv form is the same kernel with the B row of reduction index r taken from the shifted
// pixel (zero outside the image) and blockIdx.z selecting the tap, so the result lands directly in
// the tap-major packed layout the forward kernel consumes (csrc/conv.hip) -- no im2col, no repack.
//
// Exact-f32 matrix pipe (v_mfma_f32_32x32x2_f32).  Block = 128 x 128 output tile, 4 waves of
// 64 x 64.  Both operands are staged as [16 reduction rows][128] floats in LDS (coalesced 512-B
// global rows, double buffered, one barrier per 32 MFMAs/wave, software-pipelined like the conv kernel).  The MFMA wants one A value per lane
// (m = lane & 31, k = lane >> 5); the wave's 64 m-values are assigned as m = 2*(lane&31) + mt so a
// single ds_read_b64 feeds both m-tiles (likewise n), and the half-waves read two different LDS rows
// -> conflict-free without padding.  The accumulator's column index is then n = 2*(lane&31) + nt:
// the two n-tiles of a lane are adjacent floats and leave as one 8-byte store.
// Split-R (deterministic): blockIdx.z also enumerates reduction ranges; partials go to a dense
// [split][tap][M][N] workspace and are summed in fixed order by gemm_tn_reduce_kernel.
#include "common.h"

namespace frcnn {

static constexpr int GT_T  = 128;   // tile edge (both m and n)
static constexpr int GT_RK = 16;    // reduction rows per stage

typedef float f32x2 __attribute__((ext_vector_type(2)));

// geometry of the convolution whose weight gradient is computed (CONV mode): reduction row r = output pixel (n, oy, ox)
struct WgradGeom { int H, W, Ho, Wo, ks, stride, pad; };

template <bool CONV>
__global__ __launch_bounds__(256)
void gemm_tn_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                    float* __restrict__ C, int ldc, int M, int N, int R,
                    int rows_per_split, int splits, float* __restrict__ ws, WgradGeom cg)
{
    __shared__ __attribute__((aligned(16))) float As[2][GT_RK][GT_T];
    __shared__ __attribute__((aligned(16))) float Bs[2][GT_RK][GT_T];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * GT_T, m0 = blockIdx.y * GT_T;
    const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
    const int r_begin = split * rows_per_split;
    int r_end = r_begin + rows_per_split;
    if (r_end > R) r_end = R;
    const int dy = CONV ? tap / cg.ks - cg.pad : 0, dx = CONV ? tap % cg.ks - cg.pad : 0;

    // this thread's two 16-B pieces per operand per stage: row = q >> 5, column = (q & 31) * 4
    const int prow0 = tid >> 5, pcol = (tid & 31) * 4;      // second piece: row + 8
    const bool a_col_ok = m0 + pcol + 4 <= lda;
    const bool b_col_ok = n0 + pcol + 4 <= ldb;
    unsigned ok_bits = 0;      // per piece: bit 0 = A piece valid, bit 1 = B piece valid (set by load_stage)

    f32x4 areg[2], breg[2];
    auto load_stage = [&](int r0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int r = r0 + prow0 + 8 * it;
            const bool r_ok = r < r_end;
            const bool a_ok = r_ok && a_col_ok;
            const size_t a_off = a_ok ? (size_t)r * lda + m0 + pcol : 0;
            areg[it] = *reinterpret_cast<const f32x4*>(A + a_off);
            bool b_ok = r_ok && b_col_ok;
            size_t b_row = (size_t)r;
            if (CONV) {
                const int hw = cg.Ho * cg.Wo;
                const int n = r / hw, rem = r - n * hw;
                const int oy = rem / cg.Wo, ox = rem - oy * cg.Wo;
                const int sy = oy * cg.stride + dy, sx = ox * cg.stride + dx;
                b_ok = b_ok && sy >= 0 && sy < cg.H && sx >= 0 && sx < cg.W;
                b_row = (size_t)n * cg.H * cg.W + (size_t)(sy * cg.W + sx);
            }
            const size_t b_off = b_ok ? b_row * ldb + n0 + pcol : 0;
            breg[it] = *reinterpret_cast<const f32x4*>(B + b_off);
            ok_bits = (ok_bits & ~(3u << (2 * it))) | ((a_ok ? 1u : 0u) << (2 * it)) | ((b_ok ? 2u : 0u) << (2 * it));
        }
    };
    // invalid pieces were loaded from offset 0 (always readable); they are zeroed here, on the way into LDS
    auto store_stage = [&](int buf, unsigned ok) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x4 av = areg[it], bv = breg[it];
            if (!((ok >> (2 * it)) & 1u)) av = f32x4{0.f, 0.f, 0.f, 0.f};
            if (!((ok >> (2 * it)) & 2u)) bv = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(&As[buf][prow0 + 8 * it][pcol]) = av;
            *reinterpret_cast<f32x4*>(&Bs[buf][prow0 + 8 * it][pcol]) = bv;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // Software-pipelined reduction loop (the schedule of conv3x3_mfma_kernel, csrc/conv.hip): per 16-row stage
    //   F0(s) = fragments of reduction rows 0..7 in registers | read F1(s) (rows 8..15) | LDS-write stage s+1 |
    //   global-load stage s+2 | 16 MFMAs on F0, one staging instruction per MFMA | barrier | read F0(s+1) under the
    //   16 MFMAs on F1.  Loads past the range are clamped by the validity bits (zero contribution).
    const int nstages = (r_end - r_begin + GT_RK - 1) / GT_RK;
    if (nstages > 0) {
        unsigned ok_w;                                // validity of the tile currently held in areg / breg
        load_stage(r_begin);
        store_stage(0, ok_bits);
        load_stage(r_begin + GT_RK);
        ok_w = ok_bits;
        __syncthreads();
        f32x2 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            a0[kk] = *reinterpret_cast<const f32x2*>(&As[0][2 * kk + lh][wm * 64 + 2 * li]);
            b0[kk] = *reinterpret_cast<const f32x2*>(&Bs[0][2 * kk + lh][wn * 64 + 2 * li]);
        }
        for (int s = 0; s < nstages; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a1[kk] = *reinterpret_cast<const f32x2*>(&As[cur][8 + 2 * kk + lh][wm * 64 + 2 * li]);
                b1[kk] = *reinterpret_cast<const f32x2*>(&Bs[cur][8 + 2 * kk + lh][wn * 64 + 2 * li]);
            }
            store_stage(nxt, ok_w);
            load_stage(r_begin + (s + 2) * GT_RK);
            ok_w = ok_bits;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[kk][mt], b0[kk][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a0[kk] = *reinterpret_cast<const f32x2*>(&As[nxt][2 * kk + lh][wm * 64 + 2 * li]);
                b0[kk] = *reinterpret_cast<const f32x2*>(&Bs[nxt][2 * kk + lh][wn * 64 + 2 * li]);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[kk][mt], b1[kk][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < 8; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // acc[mt][nt][reg] = C[m0 + 64 wm + 2 i + mt][n0 + 64 wn + 2 li + nt],  i = (reg&3) + 8 (reg>>2) + 4 lh
    float* out;
    int ld_out;
    if (ws != nullptr) {
        out = ws + ((size_t)split * gridDim.z / splits + tap) * (size_t)M * N;   // [split][tap][M][N]
        ld_out = N;
    } else {
        out = C + (size_t)tap * M * ldc;
        ld_out = ldc;
    }
    const int n = n0 + 64 * wn + 2 * li;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int i = (reg & 3) + 8 * (reg >> 2) + 4 * lh;
            const int m = m0 + 64 * wm + 2 * i + mt;
            if (m >= M) continue;
            float* p = out + (size_t)m * ld_out + n;
            if (n + 1 < N) *reinterpret_cast<f32x2*>(p) = f32x2{acc[mt][0][reg], acc[mt][1][reg]};
            else if (n < N) *p = acc[mt][0][reg];
        }
}

// ---- bf16 operands (BASELINE configs[4]: the reduced-precision train step) --------------------------
// The same GEMM with both operands rounded to bfloat16 (round to nearest even: v_cvt_pk_bf16_f32, what torch's .bfloat16() does)
// on their way into LDS and multiplied on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, 16x the f32 pipe's rate), float32
// accumulation.  Products of bf16 values are exact in f32, so the result differs from the f32 GEMM of the ROUNDED operands only
// by the accumulation order -- the oracle (oracle/train_oracle.py, grad_math="bf16") states exactly that.
//
// The MFMA wants 8 consecutive reduction indices of one m per lane while the operands are reduction-major in memory, so the
// staging transposes: a thread loads the same four columns of 8 consecutive reduction rows (8 x 16 B, each a coalesced row
// segment across the wave), converts, and writes four 16-byte LDS pieces [m][8 r].  LDS rows are 32 bf16 + 8 pad (80 B: the
// fragment reads of the 32 x 2 lane map are conflict-free).  Out-of-range rows / shifted pixels outside the image are buffer
// loads past the descriptor: the hardware returns zeros.
static constexpr int GB_RK = 32;     // reduction rows per stage (two MFMA k-steps)
static constexpr int GB_ROW = 40;    // bf16 per LDS row

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool CONV>
__global__ __launch_bounds__(256)
void gemm_tn_bf16_kernel(const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb,
                         float* __restrict__ C, int ldc, int M, int N, int R,
                         int rows_per_split, int splits, float* __restrict__ ws, WgradGeom cg, unsigned a_bytes, unsigned b_bytes)
{
    __shared__ __attribute__((aligned(16))) unsigned short As[2][GT_T][GB_ROW];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][GT_T][GB_ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * GT_T, m0 = blockIdx.y * GT_T;
    const int tap = blockIdx.z / splits, split = blockIdx.z - tap * splits;
    const int r_begin = split * rows_per_split;
    int r_end = r_begin + rows_per_split;
    if (r_end > R) r_end = R;
    const int dy = CONV ? tap / cg.ks - cg.pad : 0, dx = CONV ? tap % cg.ks - cg.pad : 0;

    // staging task of this thread: waves 0,1 stage A, waves 2,3 stage B; columns 4 mq .. + 3 of reduction rows 8 ro .. + 7
    const bool is_b = tid >= 128;
    const int q = tid & 127, mq = q & 31, ro = q >> 5;
    const int ld = is_b ? ldb : lda;
    const int col = (is_b ? n0 : m0) + 4 * mq;
    const bool col_ok = col + 4 <= ld;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(is_b ? B : A), 0, (int)(is_b ? b_bytes : a_bytes), 0x00020000);
    unsigned short (*const dst0)[GB_ROW] = is_b ? Bs[0] : As[0];

    f32x4 reg[8];
    auto load_stage = [&](int r0) {
        const int r = r0 + 8 * ro;
        int n_ = 0, oy = 0, ox = 0;
        if (CONV && is_b) {
            const int hw = cg.Ho * cg.Wo;
            n_ = r / hw;
            const int rem = r - n_ * hw;
            oy = rem / cg.Wo;
            ox = rem - oy * cg.Wo;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bool ok = (r + j) < r_end && col_ok;
            unsigned row = (unsigned)(r + j);
            if (CONV && is_b) {
                const int sy = oy * cg.stride + dy, sx = ox * cg.stride + dx;
                ok = ok && sy >= 0 && sy < cg.H && sx >= 0 && sx < cg.W;
                row = (unsigned)((n_ * cg.H + sy) * cg.W + sx);
                if (++ox == cg.Wo) { ox = 0; if (++oy == cg.Ho) { oy = 0; ++n_; } }
            }
            const unsigned off = ok ? (row * (unsigned)ld + (unsigned)col) * 4u : 0xFFFFFFF0u;
            reg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            bf16x8_t v;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const bf16x2_t pr = __builtin_convertvector(f32x2{reg[j][c], reg[j + 1][c]}, bf16x2_t);
                v[j] = pr[0];
                v[j + 1] = pr[1];
            }
            *reinterpret_cast<bf16x8_t*>(&dst0[buf * GT_T + 4 * mq + c][8 * ro]) = v;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nstages = (r_end - r_begin + GB_RK - 1) / GB_RK;
    if (nstages > 0) {
        load_stage(r_begin);
        store_stage(0);
        __syncthreads();
        for (int s = 0; s < nstages; ++s) {
            const int cur = s & 1;
            load_stage(r_begin + (s + 1) * GB_RK);           // past the range: zeros
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8_t af[2], bf[2];
#pragma unroll
                for (int tl = 0; tl < 2; ++tl) {
                    af[tl] = *reinterpret_cast<const bf16x8_t*>(&As[cur][wm * 64 + tl * 32 + li][ks * 16 + lh * 8]);
                    bf[tl] = *reinterpret_cast<const bf16x8_t*>(&Bs[cur][wn * 64 + tl * 32 + li][ks * 16 + lh * 8]);
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
            }
            store_stage(cur ^ 1);
            __syncthreads();
        }
    }

    // acc[mt][nt][reg] = C[m0 + 64 wm + 32 mt + i][n0 + 64 wn + 32 nt + li],  i = (reg&3) + 8 (reg>>2) + 4 lh
    float* out;
    int ld_out;
    if (ws != nullptr) {
        out = ws + ((size_t)split * gridDim.z / splits + tap) * (size_t)M * N;   // [split][tap][M][N]
        ld_out = N;
    } else {
        out = C + (size_t)tap * M * ldc;
        ld_out = ldc;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 64 * wn + 32 * nt + li;
            if (n >= N) continue;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int m = m0 + 64 * wm + 32 * mt + (reg & 3) + 8 * (reg >> 2) + 4 * lh;
                if (m < M) out[(size_t)m * ld_out + n] = acc[mt][nt][reg];
            }
        }
}

// C[tap][m][n] (row stride ldc) = sum over splits of ws[split][tap][m][n], ascending split order.
__global__ __launch_bounds__(256)
void gemm_tn_reduce_kernel(const float* __restrict__ ws, int splits, int taps, int M, int N,
                           float* __restrict__ C, int ldc)
{
    const size_t per = (size_t)taps * M * N;
    const int N2 = N >> 1;                     // N is even on this path
    const size_t total = (size_t)taps * M * N2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n2 = (int)(i % N2);
        const size_t row = i / N2;             // tap * M + m
        const size_t src = row * N + 2 * n2;
        f32x2 v = *reinterpret_cast<const f32x2*>(ws + src);
        for (int s = 1; s < splits; ++s) {
            const f32x2 t = *reinterpret_cast<const f32x2*>(ws + s * per + src);
            v[0] += t[0]; v[1] += t[1];
        }
        *reinterpret_cast<f32x2*>(C + row * ldc + 2 * n2) = v;
    }
}

namespace {
// Reduction ranges so that the launch has ~512+ blocks, each range >= 128 rows, partials fit in ws.  (Round 4: 1024 blocks / 64 rows gave the
// small weight matrices of the ResNets up to 512 ranges whose planes gemm_tn_reduce_kernel then streams: ResNet-101 bf16 step 12.14 ms; 512 /
// 64: 11.85, 512 / 128: 11.75, 256 / 128: 11.88, 2048 / 64: 12.29)
void choose_split(int M, int N, int R, int taps, size_t ws_bytes, bool have_ws, int* splits, int* rows_per_split, int rk)
{
    const long tiles = (long)cdiv(M, GT_T) * cdiv(N, GT_T) * taps;
    int s = 1;
    if (have_ws && (N % 2 == 0) && tiles < 768) {
        static const int target = []() { const char* e = frcnn_knob("FRCNN_WGRAD_BLOCKS"); return e ? atoi(e) : 512; }();        // experiments
        static const int min_rows = []() { const char* e = frcnn_knob("FRCNN_WGRAD_MIN_ROWS"); return e ? atoi(e) : 128; }();
        s = (int)((target + tiles - 1) / tiles);
        const int max_by_rows = R / min_rows > 0 ? R / min_rows : 1;
        if (s > max_by_rows) s = max_by_rows;
        const size_t per = (size_t)taps * M * N * sizeof(float);
        const size_t max_by_ws = per ? ws_bytes / per : 1;
        if ((size_t)s > max_by_ws) s = (int)max_by_ws;
        if (s < 1) s = 1;
    }
    int rps = cdiv(cdiv(R, s), rk) * rk;           // multiples of the bf16 kernel's 32-row stage (the f32 kernel's is 16)
    if (rps < rk) rps = rk;
    *splits = cdiv(R, rps) > 0 ? cdiv(R, rps) : 1;
    *rows_per_split = rps;
}

template <bool CONV>
int launch_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R, int taps,
              WgradGeom cg, void* ws, size_t ws_bytes, hipStream_t s, int math = FRCNN_GRAD_F32, size_t b_rows = 0)
{
    int splits, rps;
    choose_split(M, N, R, taps, ws_bytes, ws != nullptr, &splits, &rps, math == FRCNN_GRAD_BF16 ? GB_RK : GT_RK);
    if (cdiv(N, GT_T) > 65535 || cdiv(M, GT_T) > 65535 || taps * splits > 65535) return FRCNN_EINVAL;
    float* part = splits > 1 ? static_cast<float*>(ws) : nullptr;
    dim3 grid(cdiv(N, GT_T), cdiv(M, GT_T), taps * splits);
    if (math == FRCNN_GRAD_BF16) {
        const size_t a_bytes = (size_t)R * lda * sizeof(float), b_bytes = (CONV ? b_rows : (size_t)R) * ldb * sizeof(float);
        if (a_bytes >= 0xFFFFFFF0u || b_bytes >= 0xFFFFFFF0u) return FRCNN_EINVAL;      // 32-bit buffer offsets
        hipLaunchKernelGGL((gemm_tn_bf16_kernel<CONV>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, R, rps, splits,
                           part, cg, (unsigned)a_bytes, (unsigned)b_bytes);
    } else if (math == FRCNN_GRAD_F32) {
        hipLaunchKernelGGL((gemm_tn_kernel<CONV>), grid, dim3(256), 0, s, A, lda, B, ldb, C, ldc, M, N, R, rps, splits,
                           part, cg);
    } else {
        return FRCNN_EINVAL;
    }
    int rc = check_launch();
    if (rc || splits == 1) return rc;
    const size_t total = (size_t)taps * M * (N / 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, splits, taps, M, N, C, ldc);
    return check_launch();
}
}  // namespace

size_t gemm_tn_workspace_bytes(int M, int N, int R, int taps)
{
    int s16, s32, rps;                      // enough for either arithmetic (their stage lengths round the ranges differently)
    choose_split(M, N, R, taps, (size_t)1 << 40, true, &s16, &rps, GT_RK);
    choose_split(M, N, R, taps, (size_t)1 << 40, true, &s32, &rps, GB_RK);
    const int splits = s16 > s32 ? s16 : s32;
    return splits > 1 ? (size_t)splits * taps * M * N * sizeof(float) : 0;
}

int launch_gemm_tn(const float* A, int lda, const float* B, int ldb, float* C, int ldc, int M, int N, int R,
                   void* ws, size_t ws_bytes, hipStream_t s, int math)
{
    if (M < 1 || N < 1 || R < 1 || lda < M || ldb < N || ldc < N || (lda & 3) || (ldb & 3) || (ldc & 1)) return FRCNN_EINVAL;
    if ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) return FRCNN_EINVAL;
    if (reinterpret_cast<uintptr_t>(C) & 7) return FRCNN_EINVAL;
    return launch_tn<false>(A, lda, B, ldb, C, ldc, M, N, R, 1, WgradGeom{0, 0, 0, 0, 1, 1, 0}, ws, ws_bytes, s, math);
}

int launch_conv3x3_wgrad(const float* x, const float* dz, float* dwp, int H, int W, int cin, int cout,
                         void* ws, size_t ws_bytes, hipStream_t s, int math)
{
    if (H < 1 || W < 1 || cin < 4 || cout < 4 || (cin & 3) || (cout & 3)) return FRCNN_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) return FRCNN_EINVAL;
    if ((long)H * W > (1L << 30)) return FRCNN_EINVAL;
    // A = dz [pixel][cout], B = x [pixel (shifted)][cin], C = dwp [tap][cout][cin]
    return launch_tn<true>(dz, cout, x, cin, dwp, cin, cout, cin, H * W, 9, WgradGeom{H, W, H, W, 3, 1, 1}, ws, ws_bytes, s, math,
                           (size_t)H * W);
}

// general form (ResNet bottlenecks): x [N][H][W][cin], dz [N][Ho][Wo][cout] -> dwp [k*k][cout][cin]
int launch_conv_wgrad(const float* x, const float* dz, float* dwp, int N, int H, int W, int cin, int cout, int ks, int stride,
                      int pad, void* ws, size_t ws_bytes, hipStream_t s, int math)
{
    if (N < 1 || H < 1 || W < 1 || cin < 4 || cout < 4 || (cin & 3) || (cout & 3) || ks < 1 || ks > 7 || stride < 1 || pad < 0)
        return FRCNN_EINVAL;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dz)) & 15) return FRCNN_EINVAL;
    const int Ho = (H + 2 * pad - ks) / stride + 1, Wo = (W + 2 * pad - ks) / stride + 1;
    if (Ho < 1 || Wo < 1 || (long)N * H * W > (1L << 30)) return FRCNN_EINVAL;
    return launch_tn<true>(dz, cout, x, cin, dwp, cin, cout, cin, N * Ho * Wo, ks * ks, WgradGeom{H, W, Ho, Wo, ks, stride, pad},
                           ws, ws_bytes, s, math, (size_t)N * H * W);
}

}  // namespace frcnn
