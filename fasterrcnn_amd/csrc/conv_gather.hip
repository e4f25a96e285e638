// conv_gather.hip -- generic NHWC convolution as a gather implicit GEMM on the exact-f32 matrix
// pipe, for everything the halo-tile kernel of conv.hip does not cover in the ResNet backbones
// (reference: models/resnet.py:33-118, i.e. torchvision's Bottleneck v1.5 blocks):
//   * 1x1 convolutions (plain GEMM over pixels), with stride 2 for the downsample branches
//   * 3x3 stride-2 convolutions (layer2.0 / layer3.0 / layer4.0 conv2)
//   * 3x3 convolutions on the tiny per-RoI maps of the detector head (300 x 7x7 / 4x4)
// Output row m = (n, oy, ox) of a batch of N maps; for K-stage (16-channel chunk c, tap (r,s))
// the A row is the 64-byte run x[n][oy*stride-pad+r][ox*stride-pad+s][16c..16c+15] or zeros.
// Staging, LDS layout (rows padded to 20 floats), lane-half K ownership, double buffering and the
// deterministic split-K are those of linear.hip; the epilogue fuses the folded BatchNorm bias, the
// residual add and ReLU (Bottleneck: out = relu(bn3(conv3) + identity)).
#include "common.h"
#include "x3t.h"
#include <type_traits>

namespace frcnn {

static constexpr int GLDK = 20;

template <int TM, int TN, int WM, int WN>
struct GatherCfg {
    static constexpr int BM = 32 * TM * WM;
    static constexpr int BN = 32 * TN * WN;
    static constexpr int NA = BM * 4 / 256;
    static constexpr int NB = BN * 4 / 256;
    static constexpr int A_F = BM * GLDK;
    static constexpr int B_F = BN * GLDK;
    static constexpr size_t LDS_BYTES = (size_t)2 * (A_F + B_F) * sizeof(float);
};

struct GatherShape {
    int N, H, W, Cin, Cout, Ho, Wo, R, S, stride, pad;
    // transposed != 0: the data-gradient form.  Destination rows are the pixels (n, oy, ox) of the forward conv's INPUT
    // (Ho x Wo = that input's size), the source (H x W) is the gradient of the forward output, and tap (r, s) reads source
    // pixel ((oy + pad - r) / stride, (ox + pad - s) / stride) when both divisions are exact and in range.
    int transposed;
};

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256)
void conv_gather_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                             const float* __restrict__ bias, const float* __restrict__ residual,
                             float* __restrict__ y, float* __restrict__ ws, GatherShape g,
                             int stages_per_split, int relu)
{
    using C = GatherCfg<TM, TN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const at0 = smem;
    float* const bt0 = smem + 2 * C::A_F;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * C::BM, n0 = blockIdx.x * C::BN;
    const int M = g.N * g.Ho * g.Wo;
    const int taps = g.R * g.S;
    const int total_stages = (g.Cin >> 4) * taps;
    const int st_begin = blockIdx.z * stages_per_split;
    int st_end = st_begin + stages_per_split;
    if (st_end > total_stages) st_end = total_stages;
    const int nst = st_end - st_begin;

    // loop-invariant per-row gather coordinates
    int a_img[C::NA], a_iy[C::NA], a_ix[C::NA], a_dst[C::NA];
#pragma unroll
    for (int it = 0; it < C::NA; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        const int m = m0 + row;
        a_dst[it] = row * GLDK + 4 * p;
        if (m < M) {
            const int n = m / (g.Ho * g.Wo);
            const int rem = m - n * g.Ho * g.Wo;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_img[it] = n * g.H * g.W;              // pixel offset of the image
            a_iy[it] = g.transposed ? oy + g.pad : oy * g.stride - g.pad;
            a_ix[it] = g.transposed ? ox + g.pad : ox * g.stride - g.pad;
        } else {
            a_img[it] = -1; a_iy[it] = 0; a_ix[it] = 0;
        }
    }
    int b_row[C::NB], b_dst[C::NB];
#pragma unroll
    for (int it = 0; it < C::NB; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        b_row[it] = (n0 + row < g.Cout) ? (n0 + row) : -1;
        b_dst[it] = row * GLDK + 4 * p;
    }
    const int p4 = (tid & 3) * 4;

    // Loads are never predicated (a predicated load costs an immediate vmcnt(0)): an invalid piece is read from offset 0
    // and zeroed by a select on its way into LDS.  `ok` carries one validity bit per piece of the tile held in registers.
    f32x4 areg[C::NA], breg[C::NB];
    unsigned ok_bits = 0;
    auto load_tiles = [&](int stage) {
        const int chunk = stage / taps, tap = stage - chunk * taps;
        const int r = tap / g.S, s = tap - r * g.S;
        const int c0 = chunk * 16 + p4;
        unsigned ok = 0;
#pragma unroll
        for (int it = 0; it < C::NA; ++it) {
            int iy = a_iy[it] + r, ix = a_ix[it] + s;
            bool v_ok = a_img[it] >= 0;
            if (g.transposed) {
                const int ty = a_iy[it] - r, tx = a_ix[it] - s;
                iy = ty / g.stride; ix = tx / g.stride;
                v_ok = v_ok && ty >= 0 && tx >= 0 && iy * g.stride == ty && ix * g.stride == tx;
            }
            v_ok = v_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            const size_t off = v_ok ? ((size_t)a_img[it] + (size_t)iy * g.W + ix) * g.Cin + c0 : 0;
            areg[it] = *reinterpret_cast<const f32x4*>(x + off);
            ok |= (v_ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < C::NB; ++it) {
            const bool v_ok = b_row[it] >= 0;
            const size_t off = v_ok ? ((size_t)tap * g.Cout + b_row[it]) * g.Cin + c0 : 0;
            breg[it] = *reinterpret_cast<const f32x4*>(wp + off);
            ok |= (v_ok ? 1u : 0u) << (16 + it);
        }
        ok_bits = ok;
    };
    auto store_tiles = [&](int buf, unsigned ok) {
#pragma unroll
        for (int it = 0; it < C::NA; ++it) {
            f32x4 v = areg[it];
            if (!((ok >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(at0 + buf * C::A_F + a_dst[it]) = v;
        }
#pragma unroll
        for (int it = 0; it < C::NB; ++it) {
            f32x4 v = breg[it];
            if (!((ok >> (16 + it)) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(bt0 + buf * C::B_F + b_dst[it]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_base = (32 * TM * wm + li) * GLDK + 4 * lh;
    const int b_base = (32 * TN * wn + li) * GLDK + 4 * lh;

    // Software-pipelined K loop (the schedule of conv3x3_mfma_kernel, csrc/conv.hip): F0(s) in registers | read F1(s) |
    // LDS-write tile s+1 | gather-load tile s+2 | MFMAs on F0 with one staging instruction per MFMA | barrier |
    // read F0(s+1) under the MFMAs on F1.  Prefetches past the last stage are clamped to it.
    if (nst > 0) {
        const int last = st_end - 1;
        load_tiles(st_begin);
        store_tiles(0, ok_bits);
        load_tiles(st_begin + 1 < st_end ? st_begin + 1 : last);
        unsigned ok_w = ok_bits;
        __syncthreads();
        f32x4 a0[TM], b0[TN], a1[TM], b1[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const f32x4*>(at0 + a_base + i * 32 * GLDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) b0[j] = *reinterpret_cast<const f32x4*>(bt0 + b_base + j * 32 * GLDK);
        constexpr int N_RD = TM + TN, N_ST = C::NA + C::NB;
        for (int s = 0; s < nst; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const float* at = at0 + cur * C::A_F + a_base;
            const float* bt = bt0 + cur * C::B_F + b_base;
#pragma unroll
            for (int i = 0; i < TM; ++i) a1[i] = *reinterpret_cast<const f32x4*>(at + i * 32 * GLDK + 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b1[j] = *reinterpret_cast<const f32x4*>(bt + j * 32 * GLDK + 8);
            store_tiles(nxt, ok_w);
            {
                const int s2 = st_begin + s + 2;
                load_tiles(s2 < st_end ? s2 : last);
                ok_w = ok_bits;
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0[j][kk], a0[i][kk], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < N_RD; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
            for (int q = 0; q < N_ST; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
            for (int q = 0; q < N_ST; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
            {
                const float* nat = at0 + nxt * C::A_F + a_base;
                const float* nbt = bt0 + nxt * C::B_F + b_base;
#pragma unroll
                for (int i = 0; i < TM; ++i) a0[i] = *reinterpret_cast<const f32x4*>(nat + i * 32 * GLDK);
#pragma unroll
                for (int j = 0; j < TN; ++j) b0[j] = *reinterpret_cast<const f32x4*>(nbt + j * 32 * GLDK);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1[j][kk], a1[i][kk], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < N_RD; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // The weights are the MFMA's ROW operand and the pixels its column operand: a lane then holds, for its pixel m (= lane & 31), the
    // output channels 8 q + 4 (lane >> 5) + 0..3 in the registers 4 q .. 4 q + 3 -- four consecutive floats of one NHWC row, so the
    // bias / residual / output accesses are 16-byte ones (a product does not depend on which operand carries which factor: the
    // sums are those of the pixel-major form bit for bit).
    const bool direct = (gridDim.z == 1);
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * g.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + 32 * (TM * wm + i) + li;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            f32x4 res[4];
            if (direct && residual) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n = n0 + 32 * (TN * wn + j) + 8 * q + 4 * lh;
                    res[q] = n < g.Cout ? *reinterpret_cast<const f32x4*>(residual + (size_t)m * g.Cout + n) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 32 * (TN * wn + j) + 8 * q + 4 * lh;
                if (n >= g.Cout) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (direct) {
                    if (bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    }
                    if (residual) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += res[q][e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(dst + (size_t)m * g.Cout + n) = v;
            }
        }
    }
}

// ---- bf16 operands (BASELINE configs[4]: the reduced-precision train step's forward and data-gradient convolutions) ---------------------------
// The same gather implicit GEMM with both operands rounded to bfloat16 (round to nearest even: v_cvt_pk_bf16_f32, torch's .bfloat16()) on
// their way INTO LDS and multiplied on the bf16 matrix pipe: ONE v_mfma_f32_32x32x16_bf16 per K-stage and accumulator where the float32
// kernel issues eight v_mfma_f32_32x32x2f32 (16x the matrix rate), float32 accumulation, the float32 epilogue (bias, residual, ReLU)
// unchanged.  Products of bf16 values are exact in float32, so the result differs from the float32 convolution of the ROUNDED operands
// by the accumulation order only -- what oracle/train_oracle.py (grad_math = "bf16") states.  LDS rows are 16 bf16 = 32 bytes (a lane's
// fragment = the 16 bytes at [row][8 (lane >> 5)]: the natural k order of the instruction), half the float32 kernel's traffic.
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 gb_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef float gb_f32x2 __attribute__((ext_vector_type(2)));

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256)
void conv_gather_bf16_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                             const float* __restrict__ bias, const float* __restrict__ residual,
                             float* __restrict__ y, float* __restrict__ ws, GatherShape g,
                             int stages_per_split, int relu)
{
    using C = GatherCfg<TM, TN, WM, WN>;
    constexpr int ROW = 16;                                   // bf16 per LDS row
    __shared__ __attribute__((aligned(16))) __bf16 at_s[2][C::BM * ROW];
    __shared__ __attribute__((aligned(16))) __bf16 bt_s[2][C::BN * ROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.y * C::BM, n0 = blockIdx.x * C::BN;
    const int M = g.N * g.Ho * g.Wo;
    const int taps = g.R * g.S;
    const int total_stages = (g.Cin >> 4) * taps;
    const int st_begin = blockIdx.z * stages_per_split;
    int st_end = st_begin + stages_per_split;
    if (st_end > total_stages) st_end = total_stages;
    const int nst = st_end - st_begin;

    // loop-invariant per-row gather coordinates (those of conv_gather_mfma_kernel)
    int a_img[C::NA], a_iy[C::NA], a_ix[C::NA], a_dst[C::NA];
#pragma unroll
    for (int it = 0; it < C::NA; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        const int m = m0 + row;
        a_dst[it] = row * ROW + 4 * p;
        if (m < M) {
            const int n = m / (g.Ho * g.Wo);
            const int rem = m - n * g.Ho * g.Wo;
            const int oy = rem / g.Wo, ox = rem - oy * g.Wo;
            a_img[it] = n * g.H * g.W;
            a_iy[it] = g.transposed ? oy + g.pad : oy * g.stride - g.pad;
            a_ix[it] = g.transposed ? ox + g.pad : ox * g.stride - g.pad;
        } else {
            a_img[it] = -1; a_iy[it] = 0; a_ix[it] = 0;
        }
    }
    int b_row[C::NB], b_dst[C::NB];
#pragma unroll
    for (int it = 0; it < C::NB; ++it) {
        const int q = tid + 256 * it, row = q >> 2, p = q & 3;
        b_row[it] = (n0 + row < g.Cout) ? (n0 + row) : -1;
        b_dst[it] = row * ROW + 4 * p;
    }
    const int p4 = (tid & 3) * 4;

    f32x4 areg[C::NA], breg[C::NB];
    unsigned ok_bits = 0;
    auto load_tiles = [&](int stage) {
        const int chunk = stage / taps, tap = stage - chunk * taps;
        const int r = tap / g.S, s = tap - r * g.S;
        const int c0 = chunk * 16 + p4;
        unsigned ok = 0;
#pragma unroll
        for (int it = 0; it < C::NA; ++it) {
            int iy = a_iy[it] + r, ix = a_ix[it] + s;
            bool v_ok = a_img[it] >= 0;
            if (g.transposed) {
                const int ty = a_iy[it] - r, tx = a_ix[it] - s;
                iy = ty / g.stride; ix = tx / g.stride;
                v_ok = v_ok && ty >= 0 && tx >= 0 && iy * g.stride == ty && ix * g.stride == tx;
            }
            v_ok = v_ok && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
            const size_t off = v_ok ? ((size_t)a_img[it] + (size_t)iy * g.W + ix) * g.Cin + c0 : 0;
            areg[it] = *reinterpret_cast<const f32x4*>(x + off);
            ok |= (v_ok ? 1u : 0u) << it;
        }
#pragma unroll
        for (int it = 0; it < C::NB; ++it) {
            const bool v_ok = b_row[it] >= 0;
            const size_t off = v_ok ? ((size_t)tap * g.Cout + b_row[it]) * g.Cin + c0 : 0;
            breg[it] = *reinterpret_cast<const f32x4*>(wp + off);
            ok |= (v_ok ? 1u : 0u) << (16 + it);
        }
        ok_bits = ok;
    };
    auto to_bf16x4 = [](f32x4 v) {
        const gb_bf16x2 lo = __builtin_convertvector(gb_f32x2{v[0], v[1]}, gb_bf16x2), hi = __builtin_convertvector(gb_f32x2{v[2], v[3]}, gb_bf16x2);
        return gb_bf16x4{lo[0], lo[1], hi[0], hi[1]};
    };
    auto store_tiles = [&](int buf, unsigned ok) {
#pragma unroll
        for (int it = 0; it < C::NA; ++it) {
            f32x4 v = areg[it];
            if (!((ok >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<gb_bf16x4*>(&at_s[buf][a_dst[it]]) = to_bf16x4(v);
        }
#pragma unroll
        for (int it = 0; it < C::NB; ++it) {
            f32x4 v = breg[it];
            if (!((ok >> (16 + it)) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<gb_bf16x4*>(&bt_s[buf][b_dst[it]]) = to_bf16x4(v);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_base = (32 * TM * wm + li) * ROW + 8 * lh;
    const int b_base = (32 * TN * wn + li) * ROW + 8 * lh;
    if (nst > 0) {
        const int last = st_end - 1;
        load_tiles(st_begin);
        store_tiles(0, ok_bits);
        load_tiles(st_begin + 1 < st_end ? st_begin + 1 : last);
        unsigned ok_w = ok_bits;
        __syncthreads();
        for (int s = 0; s < nst; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            gb_bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const gb_bf16x8*>(&at_s[cur][a_base + i * 32 * ROW]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const gb_bf16x8*>(&bt_s[cur][b_base + j * 32 * ROW]);
            store_tiles(nxt, ok_w);                           // tile s + 1 (its buffer was read in iteration s - 1: behind the barrier)
            {
                const int s2 = st_begin + s + 2;
                load_tiles(s2 < st_end ? s2 : last);          // tile s + 2 leaves for registers
                ok_w = ok_bits;
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j], af[i], acc[i][j], 0, 0, 0);
            __syncthreads();
        }
    }

    // epilogue: conv_gather_mfma_kernel's (the weights were the instruction's row operand: a lane holds, for its pixel m = lane & 31, the
    // output channels 8 q + 4 (lane >> 5) + 0..3 in registers 4 q .. 4 q + 3)
    const bool direct = (gridDim.z == 1);
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * g.Cout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = m0 + 32 * (TM * wm + i) + li;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + 32 * (TN * wn + j) + 8 * q + 4 * lh;
                if (n >= g.Cout) continue;
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                if (direct) {
                    if (bias) {
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += bv[e];
                    }
                    if (residual) {
                        const f32x4 rv = *reinterpret_cast<const f32x4*>(residual + (size_t)m * g.Cout + n);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    if (relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                }
                *reinterpret_cast<f32x4*>(dst + (size_t)m * g.Cout + n) = v;
            }
        }
    }
}

// ---- f32x3 operands under ONE power-of-two scale per tensor (round 4: the ResNet-50 backbone at inference) ----------------------------------------
// Reference: the Bottleneck convolutions of models/resnet.py:38-46 (conv + frozen BatchNorm folded, residual add, ReLU), torchvision v1.5 layout.
// The gather implicit GEMM in the split-operand arithmetic of csrc/gemm_x3t.hip -- every operand value as two fp16 terms hi = fp16(v 2^e),
// lo = fp16(v 2^e - hi), three v_mfma_f32_32x32x16_f16 per product (lo x hi, hi x hi, hi x lo), float32 accumulation -- with the split done
// ON THE WAY INTO LDS (~2.5 vector instructions per value) instead of by a separate record-writing pass, which is what makes it a drop-in for the
// float32 kernel: no activation records, no per-row scale arrays, the float32 weight pack as is.  The scale is per TENSOR: 2^e with
// max|x| 2^e in [2^14, 2^15), max|x| read from a device float the PRODUCER of x left behind (this kernel's own epilogue, atomic maximum over
// its outputs; tensor_absmax_kernel for the max-pooled stem output) -- any upper bound works, so no convolution input is read twice.  A value far below the tensor's maximum keeps
// 22 bits of ITSELF down to 2^-14 of the maximum and an absolute error of 2^-38 of the maximum below that.  Held-out ResNet-50 (DESIGN.md
// section 4): 1.14 / 1.18 of the reference's own distance from the float64 truth (a CPU emulation through the oracle had said 0.92-0.94
// before the kernel existed: tools/exp_r50_global_scale.py), 2397 / 2400 of its rows.  Matrix time per 16 channels: 3 x 32 cycles against
// the float32 kernel's 8 x 64 -- and what bounds the kernel is neither: the L2 -> CU fetch of the operand tiles (tools/gx_clocks.py).
typedef _Float16 gx_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gx_u32x2 __attribute__((ext_vector_type(2)));
// A stage is 32 input channels of one filter tap.  LDS row of a stage: [hi 16 | lo 16] of channels 0..15, [hi 16 | lo 16] of 16..31,
// 8 fp16 of padding (144 B: the 16-byte fragment reads of 16 consecutive rows cover the 64 banks once)
// MODE 1 (the train step's bf16 forward / data-gradient convolutions, conv_gather_bf16_kernel's arithmetic in this kernel's pipeline): one
// bf16 term per value, LDS row = 32 bf16 + 8 of padding (80 B), one v_mfma_f32_32x32x16_bf16 per accumulator and 16 channels.
// MODE 2 (round 6): MODE 0 with the WEIGHT pack already split (launch_pack_x3g_weights: the rows of the pack in this kernel's LDS row format,
// byte for byte the size of the float32 pack) -- a weight piece is then a 16-byte copy into LDS instead of ten vector instructions in
// every block of every launch (half of the loop's operand-forming work; the weights are constants).  Same bits: the pack kernel splits with gx_split4.
// MODE 3: MODE 2 for a caller whose *xmax IS the tensor's maximum (the forwards' chains: every maximum comes from the producer's epilogue or from
// tensor_absmax_kernel): no clamp per activation value and no saturation bookkeeping (a fifth of the activation side's vector instructions);
// the clamp of MODE 0 / 2 never changes a value under a true maximum, so the bits are the same.
static constexpr int GX_F32X3 = 0, GX_BF16 = 1, GX_F32X3W = 2, GX_F32X3WT = 3;
#ifndef GX_ABLATE
#define GX_ABLATE 0      // timing experiments (tools/build_ablate.sh, results wrong): 1 = the weight side of a stage is neither split nor written to LDS, 2 = nor fetched
#endif
template <int TM, int TN, int WM, int WN, int D, int MODE = GX_F32X3>
struct GatherX3Cfg {
    static constexpr int ROW = MODE != GX_BF16 ? 72 : 40;                 // 16-bit elements per LDS row
    static constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    static constexpr int NA = BM / 32, NB = BN / 32;                      // 16-byte pieces of a stage per thread (8 per 32-channel row)
    static constexpr int OUT_LD = BN + 4;                                 // the epilogue's float32 tile in LDS: row stride (conflict-free 16-byte writes)
    static constexpr size_t STAGE_BYTES = (size_t)D * (BM + BN) * ROW * sizeof(_Float16);
    static constexpr size_t OUT_BYTES = (size_t)BM * OUT_LD * sizeof(float);
    static constexpr size_t LDS_BYTES = STAGE_BYTES > OUT_BYTES ? STAGE_BYTES : OUT_BYTES;
};

// four floats -> their hi and lo fp16 terms under the scale m: hi = fp16(v m), lo = fp16(v m - hi) (v m and the difference are exact in
// float32: m is a power of two, the difference has <= 13 significant bits).  Plain C so that the scheduler can place the conversions between
// the matrix instructions
// `bound` = 65504 / m: a value beyond it (the caller's maximum was not one) saturates instead of becoming hi = inf, lo = -inf -> NaN
// downstream (hx_split8's rule); one v_med3 per value, the products stay on v_fma_mix.
template <bool CLAMP = true>
__device__ __forceinline__ void gx_split4(const f32x4 v, float m, float bound, gx_u32x2& hi, gx_u32x2& lo)
{
    // Round 6 (csrc/wino_x3f.hip's recipe, tools/micro/split_fill.hip): t = clamp(v) m is ONE multiply (exact), the hi terms of a channel
    // PAIR are one v_cvt_pk_f16_f32 and lo = fp16(t - hi) is one v_fma_mix{lo,hi}_f16 with hi as its fp16 operand -- five issue slots per
    // value where `(_Float16)fma(c, m, -(float)h)` cost seven (an f16 <-> f32 converting instruction is half rate, and the compiler
    // converted hi BACK to float32 for the difference): the same bits (t and the difference are exact in float32).  One asm statement per
    // register pair: hipcc pads a v_fma_mixhi that follows inline-asm partial writes with an s_nop.
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (CLAMP ? __builtin_amdgcn_fmed3f(v[e], -bound, bound) : v[e]) * m;
    unsigned ha, hb, la, lb;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(ha) : "v"(t[0]), "v"(t[1]));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hb) : "v"(t[2]), "v"(t[3]));
    asm("v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(la), "=&v"(lb) : "v"(t[0]), "v"(t[1]), "v"(t[2]), "v"(t[3]), "v"(ha), "v"(hb));
    hi = gx_u32x2{ha, hb};
    lo = gx_u32x2{la, lb};
}

// 16 bytes written through to memory / read past the caches (system scope: sc0 sc1) -- the partial planes of an in-kernel split reduction
__device__ __forceinline__ void gx_store_through(float* p, f32x4 v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
// eight such loads in flight, then one wait (ONE asm statement: the compiler must not touch a destination before the wait)
__device__ __forceinline__ void gx_load8_past_caches(const float* const (&p)[8], f32x4 (&v)[8])
{
    asm volatile("global_load_dwordx4 %0, %8, off sc0 sc1\n\tglobal_load_dwordx4 %1, %9, off sc0 sc1\n\t"
                 "global_load_dwordx4 %2, %10, off sc0 sc1\n\tglobal_load_dwordx4 %3, %11, off sc0 sc1\n\t"
                 "global_load_dwordx4 %4, %12, off sc0 sc1\n\tglobal_load_dwordx4 %5, %13, off sc0 sc1\n\t"
                 "global_load_dwordx4 %6, %14, off sc0 sc1\n\tglobal_load_dwordx4 %7, %15, off sc0 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                 : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
                 : "memory");
}

// One block's maximum into *out (a float >= 0, bit order = float order): four wave maxima through LDS, then ONE atomic -- and only when the
// block's value exceeds what is already there (a stale read costs an atomic, never a wrong result): atomics on one address serialise at
// a few ns each, and thousands of waves hitting one float cost more than the convolution (measured: 134 us against 36)
__device__ __forceinline__ void gx_block_max(float vmax, float* out)
{
    __shared__ float red[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = vmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m > __builtin_nontemporal_load(out)) atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
    }
}

// Pipeline, depth D: D register sets and D LDS buffers in a ring.  While the matrix instructions of stage s run from LDS buffer s % D, the
// float32 pieces of stage s + 1 (register set (s + 1) % D, fetched D stages ago) are converted and written to buffer (s + 1) % D and
// the fetch of stage s + D + 1 is issued into the set just freed: a load has D whole stages to land (D = 2 at the 128-row tiles, 24-48
// MFMAs a stage; D = 3 at the 64 x 64 tile, whose stages are a quarter as long), where the float32 kernel's single stage of eight
// 64-cycle instructions per accumulator hid it by itself.  Measured (tools/gx_clocks.py): a stage takes 0.42 / 0.70 / 1.05 us at the
// 64 x 64 / 128 x 64 / 128 x 128 tile = its 16 / 24 / 32 KB of operands at ~40 GB/s per CU, whatever D and the instruction order.
// Block b -> tile: XCD b % 8 owns the row blocks m = 8 k + b % 8 and walks their column blocks, so an activation tile is fetched into
// one XCD's L2 once; the weights (<= 2.4 MB) sit in every L2.
// Epilogue: the accumulators go through LDS once so that a wave's 16-byte stores (and residual loads) cover whole 256 / 512-byte rows
// of y instead of 32 bytes of 32 different rows.
template <int TM, int TN, int WM, int WN, int D, int MODE>
__global__ __launch_bounds__(256, 2)
void conv_gather_x3_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                           const float* __restrict__ bias, const float* __restrict__ residual,
                           float* __restrict__ y, float* __restrict__ ws, GatherShape g,
                           int stages_per_split, int relu, const float* __restrict__ xmax, const float* __restrict__ wmax,
                           float* __restrict__ ymax, int mblocks, int nblocks, unsigned* __restrict__ sat_events,
                           unsigned* __restrict__ tile_counters)
{
    using C = GatherX3Cfg<TM, TN, WM, WN, D, MODE>;
    constexpr int GX_ROW = C::ROW;
#ifdef GX_CLOCKS
    const unsigned long long gx_t_in = __builtin_amdgcn_s_memrealtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char gx_smem[];
    _Float16* const at_s = reinterpret_cast<_Float16*>(gx_smem);            // [D][BM][ROW] (16-bit elements: fp16 hi | lo, or bf16)
    _Float16* const bt_s = at_s + D * C::BM * GX_ROW;                       // [D][BN][ROW]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int jb = blockIdx.x >> 3;
    const int mb = (jb / nblocks) * 8 + (blockIdx.x & 7), nb = jb % nblocks;
    if (mb >= mblocks) return;
    const int m0 = mb * C::BM, n0 = nb * C::BN;
    const int M = g.N * g.Ho * g.Wo;
    const int taps = g.R * g.S;
    const int total_stages = (g.Cin >> 5) * taps;
    const int st_begin = blockIdx.z * stages_per_split;
    int st_end = st_begin + stages_per_split;
    if (st_end > total_stages) st_end = total_stages;
    const int nst = st_end - st_begin;

    float xmult = 1.f, xinv = 1.f, wmult = 1.f, winv = 1.f;
    if constexpr (MODE != GX_BF16) {
        hx_row_scale(*xmax, xmult, xinv);
        hx_row_scale(*wmax, wmult, winv);
    }
    const float xbound = 65504.f * xinv, wbound = 65504.f * winv;        // the largest operand values whose hi term is finite
    float x_amax = 0.f;

    // thread -> piece: row (tid >> 3) + 32 it of the tile, channels 4 (tid & 7) .. + 3 of the stage.  Every fetch is a buffer load:
    // per-piece byte offset (constant over the stages) in the VGPR, the stage's (tap, channel chunk) offset in an SGPR, and bit 31 of the
    // VGPR offset = "outside" (row past the tile's data, tap outside the image): >= num_records, the load returns zeros -- no address
    // arithmetic, no validity select and no 64-bit multiply in the loop.  The activation base is moved back by the padding so that both
    // offsets stay non-negative.
    constexpr unsigned OUTSIDE = 0x80000000u;
    const int prow = tid >> 3, pc = tid & 7;
    const int p_dst = MODE != GX_BF16 ? prow * GX_ROW + (pc >> 2) * 32 + (pc & 3) * 4 : prow * GX_ROW + pc * 4;
    const int p_dst_w = prow * GX_ROW + pc * 8;                               // (MODE 2: piece pc of a pre-split weight row = 16 bytes of the LDS row)
    // the data-gradient form at stride 1 (g.transposed) is the forward form with the taps walked backwards and padding R - 1 - pad:
    // source pixel (oy + pad - r, ox + pad - s) = (oy - pad' + r', ox - pad' + s') with r' = R - 1 - r; only the weight tap index differs
    const int pad_e = g.transposed ? g.R - 1 - g.pad : g.pad;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x) - (pad_e * g.W + pad_e) * g.Cin, 0, (int)OUTSIDE, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, (int)OUTSIDE, 0x00020000);
    unsigned a_off[C::NA], a_bad[C::NA];          // a_bad: bit t = tap t of this output pixel reads outside the image
    const int hw = g.Ho * g.Wo;
    const float r_hw = __builtin_amdgcn_rcpf((float)hw), r_w = __builtin_amdgcn_rcpf((float)g.Wo);
#pragma unroll
    for (int it = 0; it < C::NA; ++it) {
        const int m = m0 + prow + 32 * it;
        a_off[it] = 0u; a_bad[it] = 0xFFFFFFFFu;
        if (m < M) {
            // m -> (image, row, column) by a float reciprocal and one correction step (exact for m < 2^24; an integer division is ~40
            // instructions, and 2 x NA of them were a third of a short block's life)
            int n = (int)((float)m * r_hw);
            int rem = m - n * hw;
            if (rem < 0) { --n; rem += hw; } else if (rem >= hw) { ++n; rem -= hw; }
            int oy = (int)((float)rem * r_w);
            int ox = rem - oy * g.Wo;
            if (ox < 0) { --oy; ox += g.Wo; } else if (ox >= g.Wo) { ++oy; ox -= g.Wo; }
            a_off[it] = (unsigned)((((n * g.H + oy * g.stride) * g.W + ox * g.stride) * g.Cin + pc * 4) * 4);
            const int iy0 = oy * g.stride - pad_e, ix0 = ox * g.stride - pad_e;
            unsigned br = 0u, bc = 0u;                              // rows / columns of the 3 x 3 taps outside the image
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                br |= (iy0 + r < 0 || iy0 + r >= g.H ? 1u : 0u) << r;
                bc |= (ix0 + r < 0 || ix0 + r >= g.W ? 1u : 0u) << r;
            }
            a_bad[it] = taps == 1 ? ((br | bc) & 1u)
                                  : ((br & 1u ? 7u : bc) | ((br & 2u ? 7u : bc) << 3) | ((br & 4u ? 7u : bc) << 6));
        }
    }
    unsigned b_off[C::NB];
#pragma unroll
    for (int it = 0; it < C::NB; ++it) {
        const int n = n0 + prow + 32 * it;
        b_off[it] = n < g.Cout ? (unsigned)((n * g.Cin + pc * 4) * 4) : OUTSIDE;
    }

    // the next stage to fetch (never past the split's last one: the tail re-fetches it and nobody reads the copy)
    int ld_stage = st_begin, ld_chunk = st_begin / taps, ld_tap = st_begin - (st_begin / taps) * taps;
    f32x4 areg[D][C::NA], breg[D][C::NB];
    auto load_tiles = [&](f32x4 (&ar)[C::NA], f32x4 (&br)[C::NB]) {
        const int r = g.S == 1 ? 0 : (ld_tap * 11) >> 5;                 // tap / 3 for tap < 9 (R = S in {1, 3})
        const int q = ld_tap - r * g.S;
        const int so_a = ((r * g.W + q) * g.Cin + ld_chunk * 32) * 4;
        const int so_b = ((g.transposed ? taps - 1 - ld_tap : ld_tap) * g.Cout * g.Cin + ld_chunk * 32) * 4;
        const unsigned sh = 31u - (unsigned)ld_tap;
#pragma unroll
        for (int it = 0; it < C::NA; ++it)
            ar[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)(a_off[it] | ((a_bad[it] << sh) & OUTSIDE)), so_a, 0));
        if (!(GX_ABLATE & 2)) {
#pragma unroll
        for (int it = 0; it < C::NB; ++it)
            br[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)b_off[it], so_b, 0));
        }
        // advance without a branch (a branch here would cut the stage into several scheduling regions)
        const int adv = ld_stage + 1 < st_end ? 1 : 0;
        ld_stage += adv;
        const int wrap = (ld_tap + adv == taps) ? 1 : 0;
        ld_tap = wrap ? 0 : ld_tap + adv;
        ld_chunk += wrap;
    };
    auto store_tiles = [&](int buf, const f32x4 (&ar)[C::NA], const f32x4 (&br)[C::NB]) {
        _Float16* const ad = at_s + buf * C::BM * GX_ROW + p_dst;
        _Float16* const bd = bt_s + buf * C::BN * GX_ROW + p_dst;
        if constexpr (MODE != GX_BF16) {
#pragma unroll
            for (int it = 0; it < C::NA; ++it) {
                gx_u32x2 hi, lo;
                // the largest |activation| this thread splits (two v_max3 per four values): beyond xbound the split SATURATES -- counted below
                if constexpr (MODE != GX_F32X3WT) {
                    x_amax = fmaxf(fmaxf(x_amax, fabsf(ar[it][0])), fabsf(ar[it][1]));
                    x_amax = fmaxf(fmaxf(x_amax, fabsf(ar[it][2])), fabsf(ar[it][3]));
                }
                gx_split4<MODE != GX_F32X3WT>(ar[it], xmult, xbound, hi, lo);
                *reinterpret_cast<gx_u32x2*>(ad + it * 32 * GX_ROW) = hi;
                *reinterpret_cast<gx_u32x2*>(ad + it * 32 * GX_ROW + 16) = lo;
            }
            if constexpr (MODE == GX_F32X3W || MODE == GX_F32X3WT) {
                _Float16* const bw = bt_s + buf * C::BN * GX_ROW + p_dst_w;
#pragma unroll
                for (int it = 0; it < C::NB; ++it) *reinterpret_cast<f32x4*>(bw + it * 32 * GX_ROW) = br[it];       // (already hi | lo halves: a copy)
            } else if (!(GX_ABLATE & 1)) {
#pragma unroll
            for (int it = 0; it < C::NB; ++it) {
                gx_u32x2 hi, lo;
                gx_split4(br[it], wmult, wbound, hi, lo);
                *reinterpret_cast<gx_u32x2*>(bd + it * 32 * GX_ROW) = hi;
                *reinterpret_cast<gx_u32x2*>(bd + it * 32 * GX_ROW + 16) = lo;
            }
            }
        } else {
            auto to_bf16x4 = [](f32x4 v) {                      // round to nearest even: conv_gather_bf16_kernel's conversion
                const gb_bf16x2 lo = __builtin_convertvector(gb_f32x2{v[0], v[1]}, gb_bf16x2), hi = __builtin_convertvector(gb_f32x2{v[2], v[3]}, gb_bf16x2);
                return gb_bf16x4{lo[0], lo[1], hi[0], hi[1]};
            };
#pragma unroll
            for (int it = 0; it < C::NA; ++it) *reinterpret_cast<gb_bf16x4*>(ad + it * 32 * GX_ROW) = to_bf16x4(ar[it]);
#pragma unroll
            for (int it = 0; it < C::NB; ++it) *reinterpret_cast<gb_bf16x4*>(bd + it * 32 * GX_ROW) = to_bf16x4(br[it]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int a_base = (32 * TM * wm + li) * GX_ROW + 8 * lh;
    const int b_base = (32 * TN * wn + li) * GX_ROW + 8 * lh;
    // stage s (P = s % D): fragments from LDS buffer P; register set Q = (s + 1) % D -> LDS buffer Q; stage s + D + 1 -> register set Q
    auto stage = [&](auto par) {
        constexpr int P = decltype(par)::value, Q = (P + 1) % D;
        const _Float16* const as = at_s + P * C::BM * GX_ROW + a_base;
        const _Float16* const bs = bt_s + P * C::BN * GX_ROW + b_base;
        if constexpr (MODE != GX_BF16) {
            gx_f16x8 ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[kk][i] = *reinterpret_cast<const gx_f16x8*>(as + i * 32 * GX_ROW + kk * 32);
                    al[kk][i] = *reinterpret_cast<const gx_f16x8*>(as + i * 32 * GX_ROW + kk * 32 + 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[kk][j] = *reinterpret_cast<const gx_f16x8*>(bs + j * 32 * GX_ROW + kk * 32);
                    bl[kk][j] = *reinterpret_cast<const gx_f16x8*>(bs + j * 32 * GX_ROW + kk * 32 + 16);
                }
            }
            store_tiles(Q, areg[Q], breg[Q]);
            load_tiles(areg[Q], breg[Q]);
            // per accumulator and 16 channels: weight lo x activation hi, hi x hi, hi x lo (gemm_x3t_kernel's order; the weights are the row operand)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kk][j], ah[kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk][j], ah[kk][i], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kk][j], al[kk][i], acc[i][j], 0, 0, 0);
            }
        } else {
            gb_bf16x8 af[2][TM], bf[2][TN];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = *reinterpret_cast<const gb_bf16x8*>(as + i * 32 * GX_ROW + kk * 16);
#pragma unroll
                for (int j = 0; j < TN; ++j) bf[kk][j] = *reinterpret_cast<const gb_bf16x8*>(bs + j * 32 * GX_ROW + kk * 16);
            }
            store_tiles(Q, areg[Q], breg[Q]);
            load_tiles(areg[Q], breg[Q]);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[kk][j], af[kk][i], acc[i][j], 0, 0, 0);
        }
        // order: the fragment reads, then per matrix instruction one piece's conversion, its LDS write and the refill of its registers --
        // a wave issues in order, so vector work placed BETWEEN two matrix instructions runs under the first one's 32 cycles; clustered
        // after them (the compiler's own choice) it adds to them (measured: 0.42 us a stage at the 64 x 64 tile for 6 MFMAs = 0.08 us)
        constexpr int NFR = MODE != GX_BF16 ? 4 * (TM + TN) : 2 * (TM + TN), NMF = MODE != GX_BF16 ? 6 * TM * TN : 2 * TM * TN;
        __builtin_amdgcn_sched_group_barrier(0x100, NFR, 0);
#pragma unroll
        for (int q = 0; q < (NMF > C::NA + C::NB ? NMF : C::NA + C::NB); ++q) {
            if (q < C::NA + C::NB) {
                if (!((MODE == GX_F32X3W || MODE == GX_F32X3WT) && q >= C::NA)) __builtin_amdgcn_sched_group_barrier(0x002, MODE == GX_F32X3WT ? 8 : MODE != GX_BF16 ? 10 : 4, 0);      // (a pre-split weight piece has no vector work)
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            if (q < NMF) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __syncthreads();
    };
#ifdef GX_CLOCKS
    unsigned long long gx_t_setup = 0, gx_t_first = 0;
#endif
    if (nst > 0) {
#ifdef GX_CLOCKS
        gx_t_setup = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
        for (int d = 0; d < D; ++d) load_tiles(areg[d], breg[d]);          // stages 0 .. D - 1
        store_tiles(0, areg[0], breg[0]);
        load_tiles(areg[0], breg[0]);                                      // stage D
        __syncthreads();
#ifdef GX_CLOCKS
        gx_t_first = __builtin_amdgcn_s_memrealtime();
#endif
        // groups of D stages WITHOUT a branch between them: a conditional stage inside the loop makes the compiler's wait-count
        // bookkeeping at the loop head assume the worst path (wait for the newest loads), which undoes the pipeline; the remainder
        // runs after the loop
        int s = 0;
        for (; s + D <= nst; s += D) {
            stage(std::integral_constant<int, 0>{});
            stage(std::integral_constant<int, 1>{});
            if constexpr (D == 3) stage(std::integral_constant<int, 2>{});
        }
        if (s < nst) stage(std::integral_constant<int, 0>{});
        if constexpr (D == 3) { if (s + 1 < nst) stage(std::integral_constant<int, 1>{}); }
    }

    // epilogue: accumulators -> LDS [BM][OUT_LD] (the last stage's barrier has retired every fragment read), then rows of the tile by
    // consecutive threads: scales taken out (exact powers of two), bias / residual / ReLU as conv_gather_mfma_kernel, the maximum kept
#ifdef GX_CLOCKS
    const unsigned long long gx_t_loop = __builtin_amdgcn_s_memrealtime();
#endif
    float* const out_s = reinterpret_cast<float*>(gx_smem);
    const bool direct = (gridDim.z == 1);
    constexpr int QN = C::BN / 4, RSTEP = 256 / QN, NPASS = C::BM / RSTEP;   // 16-byte pieces per tile row; rows per pass of the block; passes
    const int ec = (tid % QN) * 4, er = tid / QN;
    const int n = n0 + ec;
    // the residual rows and the bias are fetched BEFORE the accumulators go through LDS (the stage registers are free now)
    f32x4 rv[NPASS];
    f32x4 bv = f32x4{0.f, 0.f, 0.f, 0.f};
    if (direct && bias && n < g.Cout) bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
    for (int k = 0; k < NPASS; ++k) {
        const int m = m0 + er + k * RSTEP;
        rv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (direct && residual && n < g.Cout && m < M) rv[k] = *reinterpret_cast<const f32x4*>(residual + (size_t)m * g.Cout + n);
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * q + e];
                *reinterpret_cast<f32x4*>(out_s + (32 * (TM * wm + i) + li) * C::OUT_LD + 32 * (TN * wn + j) + 8 * q + 4 * lh) = v;
            }
    __syncthreads();
    float* const dst = direct ? y : ws + (size_t)blockIdx.z * M * g.Cout;
    const float unscale = xinv * winv;
    float vmax = 0.f;
    if (n < g.Cout) {
#pragma unroll
        for (int k = 0; k < NPASS; ++k) {
            const int r = er + k * RSTEP, m = m0 + r;
            if (m < M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(out_s + r * C::OUT_LD + ec);
                if (direct) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = (v[e] * unscale + bv[e]) + rv[k][e];
                        if (relu) v[e] = fmaxf(v[e], 0.f);
                        vmax = fmaxf(vmax, fabsf(v[e]));
                    }
                }
                if (!direct && tile_counters) gx_store_through(dst + (size_t)m * g.Cout + n, v);
                else *reinterpret_cast<f32x4*>(dst + (size_t)m * g.Cout + n) = v;
            }
        }
    }
    bool finished = direct;
    if (!direct && tile_counters) {
        // Split reduction finished by the LAST block of this output tile to arrive (round 6; gather_splitk_finish_kernel was 10 launches
        // and 5 % of a ResNet-50 image's kernel time): every block publishes its partial plane (release), takes a ticket, and the block
        // that draws the last one reads the planes of its tile back -- each thread the elements it wrote itself -- and sums them in
        // ASCENDING plane order, its own included: the bits of the separate pass.  No block waits for another one.
        // The planes travel WRITE-THROUGH and are read back PAST the caches (sc0 sc1 on both sides), and the ticket is an agent-scope atomic:
        // no cache-wide operation.  (With release / acquire fences at agent scope -- buffer_wbl2 / buffer_inv: every block writes back and
        // invalidates its XCD's whole L2 -- ResNet-50 fell from 628 to 515 images/sec with eight images in flight.)
        __shared__ int gx_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this thread's plane stores are acknowledged
        __syncthreads();
        if (tid == 0) {
            const unsigned old = __hip_atomic_fetch_add(tile_counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gx_last = old + 1u == gridDim.z ? 1 : 0;
            if (gx_last) __hip_atomic_store(tile_counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (every ticket of this launch is drawn: ready for the next one)
        }
        __syncthreads();
        if (gx_last) {
            finished = true;
            const size_t plane = (size_t)M * g.Cout;
            const int splits = (int)gridDim.z;
            f32x4 bq = f32x4{0.f, 0.f, 0.f, 0.f};
            if (bias && n < g.Cout) bq = *reinterpret_cast<const f32x4*>(bias + n);
            // eight 16-byte pieces per round trip: two row passes x four planes (a piece that does not exist re-reads element 0 of plane 0)
            static_assert(NPASS % 2 == 0, "row passes in pairs");
            if (n < g.Cout) {
#pragma unroll 1
                for (int k = 0; k < NPASS; k += 2) {
                    const int mA = m0 + er + k * RSTEP, mB = mA + RSTEP;
                    const bool okA = mA < M, okB = mB < M;
                    if (!okA) break;                                         // (rows ascend with k)
                    const size_t iA = (size_t)mA * g.Cout + n, iB = okB ? (size_t)mB * g.Cout + n : iA;
                    f32x4 vA = f32x4{0.f, 0.f, 0.f, 0.f}, vB = vA;
                    for (int q0 = 0; q0 < splits; q0 += 4) {
                        const float* ptr[8];
                        f32x4 t[8];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const size_t po = (size_t)(q0 + u < splits ? q0 + u : 0) * plane;
                            ptr[u] = ws + po + iA;
                            ptr[4 + u] = ws + po + iB;
                        }
                        gx_load8_past_caches(ptr, t);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (q0 + u < splits) {
                                if (q0 + u == 0) { vA = t[0]; vB = t[4]; }
                                else {
#pragma unroll
                                    for (int e = 0; e < 4; ++e) { vA[e] += t[u][e]; vB[e] += t[4 + u][e]; }
                                }
                            }
                        }
                    }
                    f32x4 rA = f32x4{0.f, 0.f, 0.f, 0.f}, rB = rA;
                    if (residual) { rA = *reinterpret_cast<const f32x4*>(residual + iA); rB = *reinterpret_cast<const f32x4*>(residual + iB); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tA = ((MODE != GX_BF16 ? vA[e] * unscale : vA[e]) + bq[e]) + rA[e];
                        const float tB = ((MODE != GX_BF16 ? vB[e] * unscale : vB[e]) + bq[e]) + rB[e];
                        vA[e] = relu ? fmaxf(tA, 0.f) : tA;
                        vB[e] = relu ? fmaxf(tB, 0.f) : tB;
                        vmax = fmaxf(vmax, fabsf(vA[e]));
                        if (okB) vmax = fmaxf(vmax, fabsf(vB[e]));
                    }
                    *reinterpret_cast<f32x4*>(y + iA) = vA;
                    if (okB) *reinterpret_cast<f32x4*>(y + iB) = vB;
                }
            }
        }
    }
    if constexpr (MODE != GX_BF16) { if (finished && ymax) gx_block_max(vmax, ymax); }
    if constexpr (MODE != GX_BF16) {
        // An activation beyond the fp16 range under the tensor's scale -- the maximum the caller passed was not one -- was CLAMPED by the split
        // (gx_split4) instead of becoming inf / NaN: not silent any more: one event per wave that saw one, in a host-mapped
        // counter (frcnn_x3_saturation_events).  With the maxima the producers' epilogues leave behind this never fires (tests/test_stress_gpu.py).
        if (sat_events && __builtin_amdgcn_ballot_w64(x_amax > xbound) != 0ull && lane == 0)
            __hip_atomic_fetch_add(sat_events, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#ifdef GX_CLOCKS
    // timing build (tools/gx_clocks.py): thread 0 of every block leaves its stamps (10 ns units) behind the output
    if (tid == 0 && direct) {
        const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();
        float* rec = y + (size_t)M * g.Cout + (size_t)blockIdx.x * 8;
        rec[0] = (float)(gx_t_setup - gx_t_in); rec[1] = (float)(gx_t_first - gx_t_setup); rec[2] = (float)(gx_t_loop - gx_t_first);
        rec[3] = (float)(t_out - gx_t_loop); rec[4] = (float)(gx_t_in & 0xFFFFFF); rec[5] = (float)(t_out & 0xFFFFFF); rec[6] = (float)nst; rec[7] = 1.0f;
    }
#endif
}

// xmax / wmax (f32x3 mode, else null): the partial planes hold accumulators still multiplied by the two tensor scales; ymax: see above
__global__ __launch_bounds__(256)
void gather_splitk_finish_kernel(const float* __restrict__ ws, int splits, const float* __restrict__ bias,
                                 const float* __restrict__ residual, float* __restrict__ y, int M, int Cout, int relu,
                                 const float* __restrict__ xmax = nullptr, const float* __restrict__ wmax = nullptr, float* __restrict__ ymax = nullptr)
{
    float unscale = 1.0f;
    if (xmax) {
        float xm, xi, wm_, wi;
        hx_row_scale(*xmax, xm, xi);
        hx_row_scale(*wmax, wm_, wi);
        unscale = xi * wi;
    }
    float vmax = 0.f;
    const int C4 = Cout >> 2;
    const size_t total = (size_t)M * C4;
    const size_t plane = (size_t)M * Cout;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        // every load of the item is issued before the first add (the planes, the bias, the residual: one round trip instead of one per
        // plane); the sum keeps the ascending plane order
        f32x4 b = {0.f, 0.f, 0.f, 0.f};
        if (bias) b = reinterpret_cast<const f32x4*>(bias)[c4];
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
        if (residual) r = reinterpret_cast<const f32x4*>(residual)[i];
        f32x4 v = reinterpret_cast<const f32x4*>(ws)[i];
        int k = 1;
        for (; k + 4 <= splits; k += 4) {
            f32x4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4*>(ws + (size_t)(k + u) * plane + i * 4);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += t[u][j];
        }
        for (; k < splits; ++k) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(ws + (size_t)k * plane + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += t[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = (xmax ? v[j] * unscale : v[j]) + b[j] + r[j];
            v[j] = relu ? fmaxf(t, 0.f) : t;
            vmax = fmaxf(vmax, fabsf(v[j]));
        }
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
    if (ymax) gx_block_max(vmax, ymax);
}

// ---- small ResNet-only kernels -----------------------------------------------------------------
// Stem: 7x7 stride-2 pad-3 convolution of the NCHW image (Cin = 3) with the folded bn1 and ReLU
// (models/resnet.py:39-41 = torchvision resnet.conv1/bn1/relu).  K = 147: VALU kernel, thread =
// output pixel x 16 channels, weights [147][cout] via scalar loads.
// Round 6: a thread computes PX horizontally adjacent output pixels (their 7-wide windows at stride 2 share input columns, and a
// tap's 16 weights -- one scalar load -- serve 16 PX fused multiply-adds instead of 16).  The fmaf order over k = (ci, r, s) per output is unchanged: same bits.
#ifndef FRCNN_STEM_PX
#define FRCNN_STEM_PX 2
#endif
template <int PX>
__global__ __launch_bounds__(256)
void conv7x7_s2_c3_kernel(const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias,
                          float* __restrict__ y, int H, int W, int Ho, int Wo, int Cout, int relu)
{
    const int WoP = (Wo + PX - 1) / PX;
    const int pp = blockIdx.x * 256 + threadIdx.x;
    const int og = blockIdx.y;
    if (pp >= Ho * WoP) return;
    const int oy = pp / WoP, ox = (pp - oy * WoP) * PX;
    float acc[PX][16];
#pragma unroll
    for (int px = 0; px < PX; ++px)
#pragma unroll
        for (int o = 0; o < 16; ++o) acc[px][o] = 0.f;
    const float* wg = wp + og * 16;
    for (int ci = 0; ci < 3; ++ci) {
        for (int r = 0; r < 7; ++r) {
            const int iy = oy * 2 - 3 + r;
            const bool yin = iy >= 0 && iy < H;
            float v[2 * PX + 5];                                            // input columns 2 ox - 3 .. 2 ox + 2 PX + 1
#pragma unroll
            for (int q = 0; q < 2 * PX + 5; ++q) {
                const int ix = ox * 2 - 3 + q;
                v[q] = (yin && ix >= 0 && ix < W) ? x[((size_t)ci * H + iy) * W + ix] : 0.f;
            }
#pragma unroll
            for (int s = 0; s < 7; ++s) {
                const float* wk = wg + (size_t)((ci * 7 + r) * 7 + s) * Cout;
#pragma unroll
                for (int o = 0; o < 16; ++o) {
                    const float wv = wk[o];
#pragma unroll
                    for (int px = 0; px < PX; ++px) acc[px][o] = fmaf(v[s + 2 * px], wv, acc[px][o]);
                }
            }
        }
    }
#pragma unroll
    for (int px = 0; px < PX; ++px) {
        if (ox + px >= Wo) break;
        float* out = y + (size_t)(oy * Wo + ox + px) * Cout + og * 16;
#pragma unroll
        for (int o4 = 0; o4 < 4; ++o4) {
            f32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float t = acc[px][o4 * 4 + j] + bias[og * 16 + o4 * 4 + j];
                v[j] = relu ? fmaxf(t, 0.f) : t;
            }
            *reinterpret_cast<f32x4*>(out + 4 * o4) = v;
        }
    }
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC (torchvision resnet.maxpool); padding never wins.
__global__ __launch_bounds__(256)
void maxpool3x3_s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int Ho, int Wo, int C)
{
    const int C4 = C >> 2;
    const size_t total = (size_t)Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        const size_t p = i / C4;
        const int ox = (int)(p % Wo), oy = (int)(p / Wo);
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (int r = 0; r < 3; ++r) {
            const int iy = oy * 2 - 1 + r;
            if (iy < 0 || iy >= H) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = ox * 2 - 1 + s;
                if (ix < 0 || ix >= W) continue;
                const f32x4 v = reinterpret_cast<const f32x4*>(x + ((size_t)iy * W + ix) * C)[c4];
#pragma unroll
                for (int j = 0; j < 4; ++j) m[j] = fmaxf(m[j], v[j]);
            }
        }
        reinterpret_cast<f32x4*>(y)[i] = m;
    }
}

// y[n][c] = mean over x then over y, as `y.mean(-1).mean(-1)` (models/resnet.py:117) computes it:
// first the mean of each row over x, then the mean of the row means.
__global__ __launch_bounds__(256)
void spatial_mean_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C)
{
    const size_t total = (size_t)N * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t n = i / C;
        float outer = 0.f;
        for (int yy = 0; yy < H; ++yy) {
            float inner = 0.f;
            for (int xx = 0; xx < W; ++xx) inner += x[((n * H + yy) * W + xx) * C + c];
            outer += inner / (float)W;
        }
        y[i] = outer / (float)H;
    }
}

// Frozen BatchNorm folded into the preceding convolution (models/resnet.py:58-77: BN always in eval
// mode): w'[o] = w[o] * gamma[o]/sqrt(var[o]+eps), b'[o] = beta[o] - mean[o]*gamma[o]/sqrt(var[o]+eps).
// Output layout is tap-major [R*S][cout][cin] (or [cin*R*S][cout] when cin == 3, for the stem).
__global__ void fold_bn_pack_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                    const float* __restrict__ beta, const float* __restrict__ mean,
                                    const float* __restrict__ var, float eps, int cout, int cin, int taps,
                                    float* __restrict__ wp, float* __restrict__ bp)
{
    const size_t total = (size_t)taps * cout * cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        int o, ci, tap;
        if (cin == 3) {            // [k = ci*taps + tap][cout]
            o = (int)(i % cout);
            const int k = (int)(i / cout);
            ci = k / taps; tap = k - ci * taps;
        } else {                   // [tap][cout][cin]
            ci = (int)(i % cin);
            const size_t t = i / cin;
            o = (int)(t % cout); tap = (int)(t / cout);
        }
        const float scale = gamma[o] / sqrtf(var[o] + eps);
        wp[i] = w[((size_t)o * cin + ci) * taps + tap] * scale;
        if (ci == 0 && tap == 0) bp[o] = beta[o] - mean[o] * scale;
    }
}

__global__ __launch_bounds__(256)
void tensor_absmax_kernel(const float* __restrict__ x, long long n4, long long n, float* __restrict__ out)
{
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (blockIdx.x == 0 && threadIdx.x < (int)(n - 4 * n4)) m = fmaxf(m, fabsf(x[4 * n4 + threadIdx.x]));
    gx_block_max(m, out);
}

// ---- host side ----------------------------------------------------------------------------------
int launch_tensor_absmax(const float* x, long long n, float* out, hipStream_t s)
{
    if (n < 1 || (reinterpret_cast<uintptr_t>(x) & 15) != 0) return FRCNN_EINVAL;
    const long long n4 = n / 4;
    long long blocks = (n4 + 1023) / 1024;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(tensor_absmax_kernel, dim3((int)blocks), dim3(256), 0, s, x, n4, n, out);
    return check_launch();
}

struct GatherPlan { int cfg; int mblocks, nblocks, splits, stages_per_split; };

// f32x3 kernel: stages of 32 channels; tiles 128 x 128 (cfg 0), 128 x 64 (cfg 2, cout <= 64), 64 x 64 (cfg 3).  What bounds the kernel is
// the L2 -> CU fetch (tools/gx_clocks.py: a block's stage takes the time its 16 / 24 / 32 KB take at ~40 GB/s per CU whatever the
// instruction schedule, and the chip moves ~6 TB/s this way), so the plan minimises an estimate built from exactly that: a block's chain
// of stages against the chip's time for all fetched bytes, a fixed entry / exit cost, and for a split reduction the finish launch and
// its planes.  Microseconds; the constants are measurements of the ResNet-50 backbone shapes at 600 x 1000.
static GatherPlan plan_gather_x3(int M, int Cout, int stages)
{
    static const double chip_bytes_per_us = []() { const char* e = frcnn_knob("FRCNN_GATHER_X3_CHIP"); return e ? atof(e) : 6.0e6; }();
    static const double split_us = []() { const char* e = frcnn_knob("FRCNN_GATHER_X3_SPLIT_US"); return e ? atof(e) : 3.5; }();
    struct Cand { int cfg, bm, bn; double stage_us; };
    const Cand big = Cout <= 64 ? Cand{2, 128, 64, 0.70} : Cand{0, 128, 128, 1.05};
    const Cand cands[2] = {big, Cand{3, 64, 64, 0.42}};
    GatherPlan best{};
    double best_t = 1e30;
    for (const Cand& c : cands) {
        const int mb = cdiv(M, c.bm), nb = cdiv(Cout, c.bn);
        const double tiles = (double)mb * nb;
        const double t_chip = tiles * stages * (c.bm + c.bn) * 128.0 / chip_bytes_per_us;      // a stage row: 32 channels x 4 bytes
        const int max_splits = stages / 4 < 1 ? 1 : (stages / 4 > 16 ? 16 : stages / 4);
        for (int want = 1; want <= max_splits; ++want) {
            const int sps = cdiv(stages, want), splits = cdiv(stages, sps);
            if (splits != want) continue;
            const double rounds = (double)cdiv((int)(tiles * splits), 512);
            double t = rounds * sps * c.stage_us;
            if (t < t_chip) t = t_chip;
            t += 3.5;
            if (splits > 1) {
                if ((size_t)splits * M * Cout * sizeof(float) > ((size_t)40 << 20)) continue;
                t += split_us + (double)(splits + 1) * M * Cout * 4.0 / 4.0e6;
            }
            if (t < best_t) { best_t = t; best = GatherPlan{c.cfg, mb, nb, splits, sps}; }
        }
    }
    return best;
}

static GatherPlan plan_gather(int M, int Cout, int stages, int math = FRCNN_GRAD_F32)
{
    GatherPlan p;
    p.cfg = (Cout <= 64) ? 1 : 0;                   // 1: 256 x 64 tile, 0: 128 x 128
    const int bm = p.cfg == 1 ? 256 : 128, bn = p.cfg == 1 ? 64 : 128;
    p.mblocks = cdiv(M, bm);
    p.nblocks = cdiv(Cout, bn);
    const int blocks = p.mblocks * p.nblocks;
    static const int target = []() { const char* e = frcnn_knob("FRCNN_GATHER_BLOCKS"); return e ? atoi(e) : 1280; }();   // experiments
    // bf16 operands: a stage is ONE matrix instruction per accumulator instead of eight, the kernel is bound by its loads and launch, and
    // the partial planes + the finish launch cost more than the tail they fill -- split only when the grid would leave half the chip idle
    int want = (math == FRCNN_GRAD_BF16 ? 256 : target) / (blocks > 0 ? blocks : 1);     // float32: ~5 blocks per CU (see conv.hip)
    int cap = stages / 8;
    if (cap < 1) cap = 1;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    if ((size_t)M * Cout * sizeof(float) > ((size_t)40 << 20)) want = 1;
    // a tall GEMM (a batch of images through a 1x1 convolution) with two blocks per CU already: the partial planes would cost more
    // than the tail they fill (measured at 8 x 75x125 pixels, 512 -> 128: 124 us unsplit, 127 + 17 us split in two)
    if (blocks >= 512 && M >= 32768) want = 1;
    p.stages_per_split = cdiv(stages, want);
    p.splits = cdiv(stages, p.stages_per_split);
    return p;
}

size_t conv_gather_workspace_bytes(int N, int H, int W, int cin, int cout, int R, int stride, int pad)
{
    if (cin % 16 != 0 || cout % 4 != 0) return 0;
    const int Ho = (H + 2 * pad - R) / stride + 1, Wo = (W + 2 * pad - R) / stride + 1;
    if (Ho < 1 || Wo < 1) return 0;
    const int M = N * Ho * Wo;
    const GatherPlan p = plan_gather(M, cout, (cin / 16) * R * R);
    const GatherPlan q = plan_gather_x3(M, cout, (cin / 32 > 0 ? cin / 32 : 1) * R * R);
    const int splits = p.splits > q.splits ? p.splits : q.splits;
    return splits > 1 ? (size_t)splits * M * cout * sizeof(float) : 0;
}

// the process-wide saturation counter of the per-tensor-scaled f32x3 convolutions: host-mapped (the kernels add to it with a system-scope
// atomic, the host reads it after synchronising the stream -- no copy, no device allocation); null if the allocation fails (nothing is counted).
// PORTABLE: one counter for the process, mapped for every device (allocated on first use with whichever device is current, it
// was only guaranteed to be mapped there) -- the count is process-wide across devices
unsigned* x3_saturation_counter()
{
    static unsigned* counter = [] {
        unsigned* q = nullptr;
        if (hipHostMalloc(reinterpret_cast<void**>(&q), 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return (unsigned*)nullptr; }
        *q = 0u;
        return q;
    }();
    return counter;
}

// the split reduction of a launch is finished inside the kernel (last block of a tile to arrive) when the caller brought tile counters
static bool gather_x3_in_kernel_finish(const GatherPlan& p, const GatherX3* x3)
{
    return p.splits > 1 && x3 && x3->tile_counters && 8LL * p.nblocks * cdiv(p.mblocks, 8) <= GX_TILE_COUNTERS;
}
template <int TM, int TN, int WM, int WN, int D, int MODE>
static int launch_gather_x3_cfg(const GatherPlan& p, const float* x, const float* wp, const float* bias, const float* residual, float* y, float* ws,
                                const GatherShape& g, int relu, hipStream_t s, const GatherX3* x3)
{
    using X = GatherX3Cfg<TM, TN, WM, WN, D, MODE>;
    auto kx = conv_gather_x3_kernel<TM, TN, WM, WN, D, MODE>;
    FRCNN_MAX_LDS_ONCE(kx, X::LDS_BYTES);
    hipLaunchKernelGGL(kx, dim3(8 * p.nblocks * cdiv(p.mblocks, 8), 1, p.splits), dim3(256), X::LDS_BYTES, s, x, wp, bias, residual, y, ws, g,
                       p.stages_per_split, relu, x3 ? x3->xmax : (const float*)nullptr, x3 ? x3->wmax : (const float*)nullptr,
                       x3 ? x3->ymax : (float*)nullptr, p.mblocks, p.nblocks, MODE != GX_BF16 ? x3_saturation_counter() : (unsigned*)nullptr,
                       gather_x3_in_kernel_finish(p, x3) ? x3->tile_counters : (unsigned*)nullptr);
    return check_launch();
}
template <int MODE>
static int launch_gather_x3_plan(const GatherPlan& p, const float* x, const float* wp, const float* bias, const float* residual, float* y, float* ws,
                                 const GatherShape& g, int relu, hipStream_t s, const GatherX3* x3)
{
    return p.cfg == 2 ? launch_gather_x3_cfg<2, 1, 2, 2, 2, MODE>(p, x, wp, bias, residual, y, ws, g, relu, s, x3)
         : p.cfg == 3 ? launch_gather_x3_cfg<1, 1, 2, 2, 3, MODE>(p, x, wp, bias, residual, y, ws, g, relu, s, x3)
                      : launch_gather_x3_cfg<2, 2, 2, 2, 2, MODE>(p, x, wp, bias, residual, y, ws, g, relu, s, x3);
}
// the shapes the pipelined kernel takes: 32-channel stages, 1x1 / 3x3 taps, 31-bit byte offsets into the activations (moved back by the
// padding) and the weights; the data-gradient form at stride 1 only
static bool gather_x3_takes(int N, int H, int W, int cin, int cout, int R, int pad, int transposed, int stride)
{
    return (R == 1 || R == 3) && cin % 32 == 0 && pad <= R - 1 && !(transposed && stride != 1) &&
           ((size_t)N * H * W + (size_t)R * W + R) * cin * sizeof(float) < ((size_t)1 << 31) &&
           (size_t)R * R * cout * cin * sizeof(float) < ((size_t)1 << 31);
}

template <int TM, int TN, int WM, int WN>
static int launch_gather_cfg(const GatherPlan& p, const float* x, const float* wp, const float* bias,
                             const float* residual, float* y, float* ws, const GatherShape& g, int relu, hipStream_t s, int math = FRCNN_GRAD_F32)
{
    using C = GatherCfg<TM, TN, WM, WN>;
    dim3 grid(p.nblocks, p.mblocks, p.splits);

    if (math == FRCNN_GRAD_BF16) {
        hipLaunchKernelGGL((conv_gather_bf16_kernel<TM, TN, WM, WN>), grid, dim3(256), 0, s, x, wp, bias, residual, y, ws, g, p.stages_per_split, relu);
        return check_launch();
    }
    auto kern = conv_gather_mfma_kernel<TM, TN, WM, WN>;
    FRCNN_MAX_LDS_ONCE(kern, C::LDS_BYTES);
    hipLaunchKernelGGL(kern, grid, dim3(256), C::LDS_BYTES, s, x, wp, bias, residual, y, ws, g, p.stages_per_split, relu);
    return check_launch();
}

// The float32 weight pack [rows = taps x cout][cin] -> the rows in conv_gather_x3_kernel's LDS row format under the tensor's scale: per 32-channel
// stage 128 bytes = [hi 16 | lo 16] of channels 0..15, [hi 16 | lo 16] of 16..31 (MODE 2 above).  Thread = one (row, 16-channel half stage).
__global__ __launch_bounds__(256)
void pack_x3g_weights_kernel(const float* __restrict__ wp, const float* __restrict__ wmax, unsigned char* __restrict__ out, long long rows, int cin)
{
    const int halves = cin >> 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * halves) return;
    float wmult, winv;
    hx_row_scale(*wmax, wmult, winv);
    const float wbound = 65504.f * winv;
    const long long row = i / halves;
    const int hf = (int)(i - row * halves);
    const float* src = wp + row * cin + hf * 16;
    gx_u32x2 hi[4], lo[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) gx_split4(*reinterpret_cast<const f32x4*>(src + 4 * q), wmult, wbound, hi[q], lo[q]);
    unsigned char* dst = out + (row * cin + hf * 16) * 4;                   // 64 bytes: 16 hi halves, then 16 lo halves
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        *reinterpret_cast<gx_u32x2*>(dst + 8 * q) = hi[q];
        *reinterpret_cast<gx_u32x2*>(dst + 32 + 8 * q) = lo[q];
    }
}

int launch_pack_x3g_weights(const float* wp, const float* wmax, void* out, long long rows, int cin, hipStream_t s)
{
    if (!wp || !wmax || !out || rows < 1 || cin < 32 || cin % 32 != 0) return FRCNN_EINVAL;
    const long long n = rows * (cin / 16);
    if ((n + 255) / 256 > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pack_x3g_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wp, wmax, static_cast<unsigned char*>(out), rows, cin);
    return check_launch();
}

int launch_conv_gather(const float* x, const float* wp, const float* bias, const float* residual, float* y,
                       int N, int H, int W, int cin, int cout, int R, int stride, int pad, unsigned flags,
                       void* ws, size_t ws_bytes, hipStream_t s, int math, const GatherX3* x3)
{
    if (N < 1 || H < 1 || W < 1 || cin % 16 != 0 || cout % 4 != 0 || R < 1 || stride < 1 || pad < 0) return FRCNN_EINVAL;
    if (math != FRCNN_GRAD_F32 && math != FRCNN_GRAD_BF16 && math != FRCNN_CONV_F32X3G) return FRCNN_EINVAL;
    if (math == FRCNN_CONV_F32X3G && (!x3 || !x3->xmax || !x3->wmax)) return FRCNN_EINVAL;
    GatherShape g;
    g.transposed = 0;
    g.N = N; g.H = H; g.W = W; g.Cin = cin; g.Cout = cout; g.R = R; g.S = R; g.stride = stride; g.pad = pad;
    g.Ho = (H + 2 * pad - R) / stride + 1;
    g.Wo = (W + 2 * pad - R) / stride + 1;
    if (g.Ho < 1 || g.Wo < 1) return FRCNN_EINVAL;
    const int M = N * g.Ho * g.Wo;
    const bool takes = gather_x3_takes(N, H, W, cin, cout, R, pad, 0, stride);
    if (math == FRCNN_CONV_F32X3G && !takes) return FRCNN_EUNSUPPORTED;
    // the pipelined kernel: the f32x3 arithmetic always, the bf16 arithmetic wherever its shape limits allow (else conv_gather_bf16_kernel)
    const bool gx3 = math == FRCNN_CONV_F32X3G || (math == FRCNN_GRAD_BF16 && takes);
    const int all_stages = gx3 ? (cin / 32) * R * R : (cin / 16) * R * R;
    GatherPlan p = gx3 ? plan_gather_x3(M, cout, all_stages) : plan_gather(M, cout, all_stages, math);
    const size_t need = p.splits > 1 ? (size_t)p.splits * M * cout * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && ws == nullptr)) {       // no scratch: run un-split
        p.splits = 1;
        p.stages_per_split = all_stages;
    }
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    int rc;
    if (gx3)
        rc = math == FRCNN_GRAD_BF16 ? launch_gather_x3_plan<GX_BF16>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, nullptr)
           : x3->wsplit && x3->trusted ? launch_gather_x3_plan<GX_F32X3WT>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, x3)
           : x3->wsplit            ? launch_gather_x3_plan<GX_F32X3W>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, x3)
                                   : launch_gather_x3_plan<GX_F32X3>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, x3);
    else
        rc = p.cfg == 1 ? launch_gather_cfg<2, 2, 4, 1>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, math)
                        : launch_gather_cfg<2, 2, 2, 2>(p, x, wp, bias, residual, y, (float*)ws, g, relu, s, math);
    if (rc) return rc;
    if (p.splits > 1 && !(gx3 && math == FRCNN_CONV_F32X3G && gather_x3_in_kernel_finish(p, x3))) {
        const size_t total = (size_t)M * (cout / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        const bool gx = math == FRCNN_CONV_F32X3G;
        hipLaunchKernelGGL(gather_splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, p.splits, bias,
                           residual, y, M, cout, relu, gx ? x3->xmax : (const float*)nullptr, gx ? x3->wmax : (const float*)nullptr,
                           gx ? x3->ymax : (float*)nullptr);
        rc = check_launch();
    }
    return rc;
}

// Data gradient of y = conv(x, w) (k x k, stride, pad; x [N][H][W][cin], y [N][Ho][Wo][cout]):
//   dx[n][iy][ix][ci] = residual + sum_{tap, co} dz[n][(iy+pad-r)/stride][(ix+pad-s)/stride][co] * wd[tap][ci][co]
// as the gather kernel in its transposed mode (wd = per-tap transpose of the forward pack, frcnn_pack_conv_dgrad).
int launch_conv_dgrad(const float* dz, const float* wd, const float* residual, float* dx, int N, int H, int W, int cin,
                      int cout, int R, int stride, int pad, void* ws, size_t ws_bytes, hipStream_t s, int math)
{
    if (N < 1 || H < 1 || W < 1 || cout % 16 != 0 || cin % 4 != 0 || R < 1 || stride < 1 || pad < 0) return FRCNN_EINVAL;
    if (math != FRCNN_GRAD_F32 && math != FRCNN_GRAD_BF16) return FRCNN_EINVAL;
    GatherShape g;
    g.transposed = 1;
    g.N = N; g.R = R; g.S = R; g.stride = stride; g.pad = pad;
    g.H = (H + 2 * pad - R) / stride + 1;           // source = dz
    g.W = (W + 2 * pad - R) / stride + 1;
    g.Ho = H; g.Wo = W;                             // destination = dx
    g.Cin = cout; g.Cout = cin;
    if (g.H < 1 || g.W < 1) return FRCNN_EINVAL;
    const int M = N * H * W;
    // bf16 at stride 1: the pipelined kernel (the forward form with the taps walked backwards)
    const bool gx3 = math == FRCNN_GRAD_BF16 && gather_x3_takes(N, g.H, g.W, cout, cin, R, pad, 1, stride);
    const int all_stages = gx3 ? (cout / 32) * R * R : (cout / 16) * R * R;
    GatherPlan p = gx3 ? plan_gather_x3(M, cin, all_stages) : plan_gather(M, cin, all_stages, math);
    const size_t need = p.splits > 1 ? (size_t)p.splits * M * cin * sizeof(float) : 0;
    if (need > ws_bytes || (need > 0 && ws == nullptr)) {
        p.splits = 1;
        p.stages_per_split = all_stages;
    }
    int rc = gx3 ? launch_gather_x3_plan<GX_BF16>(p, dz, wd, nullptr, residual, dx, (float*)ws, g, 0, s, nullptr)
           : p.cfg == 1 ? launch_gather_cfg<2, 2, 4, 1>(p, dz, wd, nullptr, residual, dx, (float*)ws, g, 0, s, math)
                        : launch_gather_cfg<2, 2, 2, 2>(p, dz, wd, nullptr, residual, dx, (float*)ws, g, 0, s, math);
    if (rc) return rc;
    if (p.splits > 1) {
        const size_t total = (size_t)M * (cin / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(gather_splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, p.splits,
                           (const float*)nullptr, residual, dx, M, cin, 0, (const float*)nullptr, (const float*)nullptr, (float*)nullptr);
        rc = check_launch();
    }
    return rc;
}

size_t conv_dgrad_workspace_bytes(int N, int H, int W, int cin, int cout, int R, int stride, int pad)
{
    if (N < 1 || H < 1 || W < 1 || cout % 16 != 0 || R < 1 || stride < 1) return 0;
    const int M = N * H * W;
    const GatherPlan p = plan_gather(M, cin, (cout / 16) * R * R);
    const GatherPlan q = plan_gather_x3(M, cin, (cout / 32 > 0 ? cout / 32 : 1) * R * R);
    const int splits = p.splits > q.splits ? p.splits : q.splits;
    return splits > 1 ? (size_t)splits * M * cin * sizeof(float) : 0;
}

int launch_conv7x7_s2_c3(const float* x, const float* wp, const float* b, float* y, int H, int W, int cout,
                         unsigned flags, hipStream_t s)
{
    if (cout % 16 != 0 || H < 1 || W < 1) return FRCNN_EINVAL;
    const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
    dim3 grid(cdiv(Ho * cdiv(Wo, FRCNN_STEM_PX), 256), cout / 16);          // (FRCNN_STEM_PX output pixels of a row per thread)
    hipLaunchKernelGGL(conv7x7_s2_c3_kernel<FRCNN_STEM_PX>, grid, dim3(256), 0, s, x, wp, b, y, H, W, Ho, Wo, cout,
                       (flags & FRCNN_RELU) ? 1 : 0);
    return check_launch();
}

int launch_maxpool3x3_s2(const float* x, float* y, int H, int W, int c, hipStream_t s)
{
    if (c % 4 != 0 || H < 1 || W < 1) return FRCNN_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const size_t total = (size_t)Ho * Wo * (c / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(maxpool3x3_s2_kernel, dim3(blocks), dim3(256), 0, s, x, y, H, W, Ho, Wo, c);
    return check_launch();
}

int launch_spatial_mean(const float* x, float* y, int N, int H, int W, int c, hipStream_t s)
{
    if (N < 1 || H < 1 || W < 1 || c < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)N * c;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(spatial_mean_kernel, dim3(blocks), dim3(256), 0, s, x, y, N, H, W, c);
    return check_launch();
}

int launch_fold_bn_pack(const float* w, const float* gamma, const float* beta, const float* mean, const float* var,
                        float eps, int cout, int cin, int ksize, float* wp, float* bp, hipStream_t s)
{
    if (cout < 1 || cin < 1 || ksize < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)ksize * ksize * cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(fold_bn_pack_kernel, dim3(blocks), dim3(256), 0, s, w, gamma, beta, mean, var, eps, cout, cin,
                       ksize * ksize, wp, bp);
    return check_launch();
}

}  // namespace frcnn
