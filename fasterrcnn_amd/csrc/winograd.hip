// winograd.hip -- 3x3 stride-1 "same" convolution as Winograd F(2x2,3x3) in float32, for the wide layers
// (cin >= 128, cout >= 256: conv3x3_uses_winograd, common.h) of models/vgg16.py:36-47 and the RPN trunk models/rpn.py:39,88.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A        d: 4x4 input patch, g: 3x3 filter, Y: 2x2 outputs
//
// 16 multiplies per 2x2 outputs and channel pair instead of 36: the matrix pipe does 2.25x less work.
// All arithmetic is float32 (the filter transform is evaluated in float64 and rounded once); on the
// reference's golden vectors the detections are reproduced at the same rate as with the direct kernel
// (tests/winograd_parity_probe.py is the CPU model of this file, tests/test_winograd_gpu.py the parity test).
//
// Three launches per layer, NHWC throughout (optionally a batch of N maps: the per-RoI 4 x 4 maps of ResNet's layer4),
// T = N * ceil(H/2) * ceil(W/2) tiles:
//   1. wino_input_kernel   x [H][W][cin]        -> V [16][T][cin]     (B^T d B, zero padding folded in)
//   2. linear_mfma_kernel  batched over the 16 positions: M_p [T][cout] = V_p [T][cin] . U_p [cout][cin]^T
//                          (csrc/linear.hip, exact-f32 MFMA, XCD-aware block order)
//   3. wino_output_kernel  M [16][T][cout]      -> y (A^T M A + bias, ReLU, optional fused 2x2 max-pool:
//                          a 2x2 output tile IS one pooling window)
// V and M are scratch (16 T (cin + cout) floats); they are HBM / Infinity-Cache traffic that the direct
// kernel does not have, which is why the narrow, large layers stay on csrc/conv.hip.
#include "common.h"

namespace frcnn {

// U[p = 4 i + j][k][c] = (G g G^T)[i][j],  G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]];  g: OIHW [cout][cin][3][3]
// `scale` (per cout, may be NULL): the frozen-BatchNorm fold of the ResNet layers -- the filter is first multiplied in float32
// exactly as fold_bn_pack_kernel does (csrc/conv_gather.hip), so both packs describe the same folded weights.
__global__ __launch_bounds__(256)
void wino_pack_kernel(const float* __restrict__ g, const float* __restrict__ scale, float* __restrict__ u, int cout, int cin)
{
    const size_t total = (size_t)cout * cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const float* gp = g + i * 9;
        const float sc = scale ? scale[i / cin] : 1.0f;
        double w[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) w[a][b] = (double)(scale ? gp[a * 3 + b] * sc : gp[a * 3 + b]);
        double r[4][3];                      // G g
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            r[0][b] = w[0][b];
            r[1][b] = 0.5 * (w[0][b] + w[1][b] + w[2][b]);
            r[2][b] = 0.5 * (w[0][b] - w[1][b] + w[2][b]);
            r[3][b] = w[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {        // (G g) G^T
            const double q0 = r[a][0];
            const double q1 = 0.5 * (r[a][0] + r[a][1] + r[a][2]);
            const double q2 = 0.5 * (r[a][0] - r[a][1] + r[a][2]);
            const double q3 = r[a][2];
            u[(size_t)(4 * a + 0) * total + i] = (float)q0;
            u[(size_t)(4 * a + 1) * total + i] = (float)q1;
            u[(size_t)(4 * a + 2) * total + i] = (float)q2;
            u[(size_t)(4 * a + 3) * total + i] = (float)q3;
        }
    }
}

// The same transform from the direct kernels' tap-major pack wp[tap][cout][cin] (the train step's master weights):
//   data_gradient == 0: U[p][k][c] of the layer's own filter (forward);
//   data_gradient == 1: the bank of the data-gradient convolution dz (cout channels) -> dx (cin channels), whose filter is
//                       the 180-degree rotated, channel-transposed one: g'[ci][co][tap] = wp[8 - tap][co][ci]; output [16][cin][cout].
__global__ __launch_bounds__(256)
void wino_pack_taps_kernel(const float* __restrict__ wp, float* __restrict__ u, int cout, int cin, int data_gradient)
{
    const size_t total = (size_t)cout * cin;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        double w[3][3];
        if (!data_gradient) {
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) w[tp / 3][tp % 3] = (double)wp[(size_t)tp * total + i];
        } else {
            const size_t ci = i / cout, co = i % cout;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) w[tp / 3][tp % 3] = (double)wp[(size_t)(8 - tp) * total + co * cin + ci];
        }
        double r[4][3];
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            r[0][b] = w[0][b];
            r[1][b] = 0.5 * (w[0][b] + w[1][b] + w[2][b]);
            r[2][b] = 0.5 * (w[0][b] - w[1][b] + w[2][b]);
            r[3][b] = w[2][b];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u[(size_t)(4 * a + 0) * total + i] = (float)r[a][0];
            u[(size_t)(4 * a + 1) * total + i] = (float)(0.5 * (r[a][0] + r[a][1] + r[a][2]));
            u[(size_t)(4 * a + 2) * total + i] = (float)(0.5 * (r[a][0] - r[a][1] + r[a][2]));
            u[(size_t)(4 * a + 3) * total + i] = (float)r[a][2];
        }
    }
}

// One thread = one tile x 4 channels.  B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].
// x: [N][H][W][cin]; tiles are numbered image-major (tpi = th * tw per image), T = N * tpi.
__global__ __launch_bounds__(256)
void wino_input_kernel(const float* __restrict__ x, float* __restrict__ v, int H, int W, int cin, int tw, int tpi, int T)
{
    const int c4n = cin >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)T * c4n) return;
    const int tile = (int)(idx / c4n), c = (int)(idx % c4n) * 4;
    const int img = tile / tpi, tin = tile - img * tpi;
    const int ty = tin / tw, tx = tin % tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    x += (size_t)img * H * W * cin;
    f32x4 d[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = y0 + a;
        const bool yok = yy >= 0 && yy < H;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = x0 + b;
            const bool ok = yok && xx >= 0 && xx < W;
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            d[a][b] = ok ? *reinterpret_cast<const f32x4*>(x + ((size_t)yy * W + xx) * cin + c) : zero;
        }
    }
    f32x4 r[4][4];                           // B^T d
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r[0][b] = d[0][b] - d[2][b];
        r[1][b] = d[1][b] + d[2][b];
        r[2][b] = d[2][b] - d[1][b];
        r[3][b] = d[1][b] - d[3][b];
    }
    const size_t plane = (size_t)T * cin;
    float* vp = v + (size_t)tile * cin + c;
#pragma unroll
    for (int a = 0; a < 4; ++a) {            // (B^T d) B
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * a + 0) * plane) = r[a][0] - r[a][2];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * a + 1) * plane) = r[a][1] + r[a][2];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * a + 2) * plane) = r[a][2] - r[a][1];
        *reinterpret_cast<f32x4*>(vp + (size_t)(4 * a + 3) * plane) = r[a][1] - r[a][3];
    }
}

// One thread = one tile x 4 output channels.  A^T = [[1,1,1,0],[0,1,-1,-1]].
// cmax_out (optional; the launcher guarantees cout % 256 == 0 and ReLU): a wave covers 256 consecutive channels of ONE tile, so the
// per-pixel channel maximum of the output is a wave reduction + one atomic maximum per pixel and wave (non-negative floats order like
// their bit patterns) -- what the next f32x3 layer would otherwise read the whole tensor again for (launch_pixel_absmax).
template <bool POOL>
__global__ __launch_bounds__(256)
void wino_output_kernel(const float* __restrict__ m, const float* __restrict__ bias, float* __restrict__ y,
                        int H, int W, int cout, int tw, int tpi, int T, int relu, float* __restrict__ cmax_out)
{
    const int k4n = cout >> 2;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const bool in_range = idx < (size_t)T * k4n;
    if (!in_range && cmax_out == nullptr) return;
    const int tile = in_range ? (int)(idx / k4n) : 0, k = in_range ? (int)(idx % k4n) * 4 : 0;
    const int img = tile / tpi, tin = tile - img * tpi;
    const int ty = tin / tw, tx = tin % tw;
    const int Ho = H >> 1, Wo = W >> 1;
    const bool live = in_range && !(POOL && (ty >= Ho || tx >= Wo));      // floor pooling drops the odd last row / column
    if (!live && cmax_out == nullptr) return;
    f32x4 o[2][2] = {};
    if (live) {
        const size_t plane = (size_t)T * cout;
        const float* mp = m + (size_t)tile * cout + k;
        f32x4 s[2][4];                           // A^T M
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 m0 = *reinterpret_cast<const f32x4*>(mp + (size_t)(0 + j) * plane);
            const f32x4 m1 = *reinterpret_cast<const f32x4*>(mp + (size_t)(4 + j) * plane);
            const f32x4 m2 = *reinterpret_cast<const f32x4*>(mp + (size_t)(8 + j) * plane);
            const f32x4 m3 = *reinterpret_cast<const f32x4*>(mp + (size_t)(12 + j) * plane);
            s[0][j] = (m0 + m1) + m2;
            s[1][j] = (m1 - m2) - m3;
        }
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + k);
#pragma unroll
        for (int a = 0; a < 2; ++a) {            // (A^T M) A
            o[a][0] = ((s[a][0] + s[a][1]) + s[a][2]) + bv;
            o[a][1] = ((s[a][1] - s[a][2]) - s[a][3]) + bv;
        }
        if (relu) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[a][b][e] = fmaxf(o[a][b][e], 0.f);
        }
    }
    float* yi = y + (size_t)img * (POOL ? (size_t)Ho * Wo : (size_t)H * W) * cout;
    float* ci = cmax_out ? cmax_out + (size_t)img * (POOL ? (size_t)Ho * Wo : (size_t)H * W) : nullptr;
    const int lane = threadIdx.x & 63;
    if (POOL) {
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(o[0][0][e], o[0][1][e]), fmaxf(o[1][0][e], o[1][1][e]));
        if (live) *reinterpret_cast<f32x4*>(yi + ((size_t)ty * Wo + tx) * cout + k) = r;
        if (cmax_out) {
            float mx = fmaxf(fmaxf(r[0], r[1]), fmaxf(r[2], r[3]));
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
            if (lane == 0 && live) atomicMax(reinterpret_cast<unsigned*>(ci + (size_t)ty * Wo + tx), __float_as_uint(mx));
        }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int yy = 2 * ty + a;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int xx = 2 * tx + b;
                const bool ok = live && yy < H && xx < W;
                if (ok) *reinterpret_cast<f32x4*>(yi + ((size_t)yy * W + xx) * cout + k) = o[a][b];
                if (cmax_out) {
                    float mx = fmaxf(fmaxf(o[a][b][0], o[a][b][1]), fmaxf(o[a][b][2], o[a][b][3]));
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
                    if (lane == 0 && ok) atomicMax(reinterpret_cast<unsigned*>(ci + (size_t)yy * W + xx), __float_as_uint(mx));
                }
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
static inline bool wino_shape_ok(int N, int H, int W, int cin, int cout)
{
    return N >= 1 && H >= 1 && W >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 128 && cout % 128 == 0 &&
           (size_t)N * cdiv(H, 2) * cdiv(W, 2) * 16 * (size_t)(cin > cout ? cin : cout) < ((size_t)1 << 31);
}

size_t conv3x3_winograd_workspace_bytes(int N, int H, int W, int cin, int cout)
{
    if (!wino_shape_ok(N, H, W, cin, cout)) return 0;
    const size_t T = (size_t)N * cdiv(H, 2) * cdiv(W, 2);
    return 16 * T * ((size_t)cin + cout) * sizeof(float);
}

int launch_pack_conv3x3_winograd(const float* w, const float* scale, float* u, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_pack_kernel, dim3(blocks), dim3(256), 0, s, w, scale, u, cout, cin);
    return check_launch();
}

int launch_pack_conv3x3_winograd_taps(const float* wp, float* u, int cout, int cin, int data_gradient, hipStream_t s)
{
    if (cout < 1 || cin < 1) return FRCNN_EINVAL;
    const size_t total = (size_t)cout * cin;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(wino_pack_taps_kernel, dim3(blocks), dim3(256), 0, s, wp, u, cout, cin, data_gradient ? 1 : 0);
    return check_launch();
}

// The three launches of one layer, separately callable so that the fused forward can time them per class.
int launch_winograd_input(const float* x, float* V, int N, int H, int W, int cin, hipStream_t s)
{
    const int tw = cdiv(W, 2), tpi = cdiv(H, 2) * tw, T = N * tpi;
    const size_t n = (size_t)T * (cin / 4);
    hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, V, H, W, cin, tw, tpi, T);
    return check_launch();
}

int launch_winograd_gemm(const float* V, const float* u, float* M, int N, int H, int W, int cin, int cout, hipStream_t s)
{
    const int T = N * cdiv(H, 2) * cdiv(W, 2);
    return launch_linear_batched(V, cin, (size_t)T * cin, u, (size_t)cout * cin, M, cout, (size_t)T * cout, T, cout, cin, 16, s);
}

bool winograd_output_emits_cmax(int cout, unsigned flags) { return cout % 256 == 0 && (flags & FRCNN_RELU) != 0; }

int launch_winograd_output(const float* M, const float* b, float* y, int N, int H, int W, int cout, unsigned flags, hipStream_t s, float* cmax_out)
{
    if (cmax_out && !winograd_output_emits_cmax(cout, flags)) return FRCNN_EINVAL;
    const int tw = cdiv(W, 2), tpi = cdiv(H, 2) * tw, T = N * tpi;
    const size_t n = (size_t)T * (cout / 4);
    const int relu = (flags & FRCNN_RELU) ? 1 : 0;
    if (flags & FRCNN_POOL2)
        hipLaunchKernelGGL(wino_output_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, M, b, y, H, W, cout, tw, tpi, T, relu, cmax_out);
    else
        hipLaunchKernelGGL(wino_output_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, M, b, y, H, W, cout, tw, tpi, T, relu, cmax_out);
    return check_launch();
}

// Validates shape and scratch; V = ws, M = ws + 16 T cin floats.
int winograd_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, float** V, float** M)
{
    if (!wino_shape_ok(N, H, W, cin, cout)) return FRCNN_EUNSUPPORTED;
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    if (ws == nullptr || ws_bytes < conv3x3_winograd_workspace_bytes(N, H, W, cin, cout)) return FRCNN_EINVAL;
    *V = static_cast<float*>(ws);
    *M = *V + (size_t)16 * N * cdiv(H, 2) * cdiv(W, 2) * cin;
    return FRCNN_OK;
}

int launch_conv3x3_winograd(const float* x, const float* u, const float* b, float* y, int N, int H, int W, int cin, int cout,
                            unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    float *V = nullptr, *M = nullptr;
    int rc = winograd_plan(N, H, W, cin, cout, flags, ws, ws_bytes, &V, &M);
    if (rc) return rc;
    if ((rc = launch_winograd_input(x, V, N, H, W, cin, s)) != FRCNN_OK) return rc;
    if ((rc = launch_winograd_gemm(V, u, M, N, H, W, cin, cout, s)) != FRCNN_OK) return rc;
    return launch_winograd_output(M, b, y, N, H, W, cout, flags, s);
}

}  // namespace frcnn
