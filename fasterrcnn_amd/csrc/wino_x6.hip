// wino_x6.hip -- 3x3 stride-1 "same" convolution as Winograd F(2x2,3x3) whose 16 position GEMMs run in the f32x6 arithmetic on the
// bf16 matrix pipe (round 3; VERDICT r2 "next" #1): the 512-channel layers conv4_1 ... conv5_3 of models/vgg16.py:89-96 and the RPN
// trunk models/rpn.py:88, which sat at 0.41-0.64 of the exact-f32 pipe with grids that do not fill the chip.
//
//   Y = A^T [ sum_c (G g G^T) .* (B^T d B) ] A
//
// Same transforms, same float32 operation order as csrc/winograd.hip / csrc/winofused.hip (so V and U are bit-identical to theirs);
// only the multiply-accumulate over the channels differs: each float32 V and U value is split exactly into three bfloat16 terms
// and the product is the sum of the six largest partial products with float32 accumulation (csrc/gemm_x6t.hip) -- 2.7x the
// exact-f32 pipe's rate at fp32-class accuracy (dropped terms <= 2^-24 relative).
//
// Three launches per layer, NHWC, optionally a batch of N maps (the per-RoI 4 x 4 maps of ResNet's layer4: models/resnet.py:110),
// T = N ceil(H/2) ceil(W/2) tiles, Tp = T rounded up to the GEMM's 320-row tile:
//   1. wino_input_x6t_kernel   x [H][W][cin] -> V as x6t records [16 positions][cin/16][Tp/32][3][1 KB]  (B^T d B, then the exact
//                              3-way split; zero padding folded in; each wave store is one whole 1 KB record piece)
//   2. gemm_x6t_kernel         16 batched GEMMs M_p [T][cout] = V_p [T][cin] . U_p [cout][cin]^T, U pre-split at pack time
//   3. wino_output_kernel      (csrc/winograd.hip) M [16][T][cout] -> y: A^T M A + bias, ReLU, optional fused 2x2 max-pool
// V (6 B per element) and M are scratch: 118 + 78 MB for a 75 x 125 x 512 layer, Infinity-Cache sized.  The transforms are HBM / L2
// bound and use no matrix pipe, so with several images in flight they overlap other images' GEMMs.
#include "common.h"

namespace frcnn {

static constexpr int WX_PIECE = 1024, WX_RB = 3072;

__device__ __forceinline__ unsigned short wx_bf16_rne(float f)
{
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}

// 8 float32 -> the three 16-byte record pieces (hi, mid, lo), x = hi + mid + lo exactly (csrc/gemm_x6t.hip: gx_split3)
__device__ __forceinline__ void wx_split8(const float (&v)[8], uint4& ph, uint4& pm, uint4& pl)
{
    unsigned hi[8], mid[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = wx_bf16_rne(v[j]);
        const float r1 = v[j] - __uint_as_float(hi[j] << 16);
        mid[j] = wx_bf16_rne(r1);
        const float r2 = r1 - __uint_as_float(mid[j] << 16);
        lo[j] = wx_bf16_rne(r2);
    }
    ph = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
    pm = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
    pl = make_uint4(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16), lo[4] | (lo[5] << 16), lo[6] | (lo[7] << 16));
}

// One wave = (16-channel chunk, block of 32 tiles): lane l = tile (l & 31), channels 8 (l >> 5) .. + 7 of the chunk.
// B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]] exactly as wino_input_kernel (csrc/winograd.hip).
__global__ __launch_bounds__(256)
void wino_input_x6t_kernel(const float* __restrict__ x, unsigned char* __restrict__ vrec, int H, int W, int cin, int tw, int tpi, int T,
                           int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= rbt * K16) return;
    // consecutive waves = consecutive chunks of the same tile block: the 16 pixels a lane reads are 64 contiguous bytes per chunk pair
    const int chunk = wave % K16, rb = wave / K16;
    const int tile = rb * 32 + (lane & 31);
    const int c = chunk * 16 + 8 * (lane >> 5);
    const bool live = tile < T;
    // x: [N][H][W][cin]; tiles are numbered map-major (tpi = tiles per map, T = N * tpi) as wino_input_kernel does (csrc/winograd.hip)
    const int img = live ? tile / tpi : 0, tin = live ? tile - img * tpi : 0;
    const int ty = tin / tw, tx = tin - ty * tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    x += (size_t)img * H * W * cin;
    float d[4][4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int yy = y0 + a;
        const bool yok = live && yy >= 0 && yy < H;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int xx = x0 + b;
            const bool ok = yok && xx >= 0 && xx < W;
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const float* src = x + ((size_t)yy * W + xx) * cin + c;
                v0 = *reinterpret_cast<const f32x4*>(src);
                v1 = *reinterpret_cast<const f32x4*>(src + 4);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[a][b][e] = v0[e]; d[a][b][4 + e] = v1[e]; }
        }
    }
    // r = B^T d (in place, column by column)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d0 = d[0][b][e], d1 = d[1][b][e], d2 = d[2][b][e], d3 = d[3][b][e];
            d[0][b][e] = d0 - d2;
            d[1][b][e] = d1 + d2;
            d[2][b][e] = d2 - d1;
            d[3][b][e] = d1 - d3;
        }
    }
    const size_t pos_stride = (size_t)K16 * rbt * WX_RB;
    unsigned char* dst = vrec + ((size_t)chunk * rbt + rb) * WX_RB + lane * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float r0 = d[a][0][e], r1 = d[a][1][e], r2 = d[a][2][e], r3 = d[a][3][e];
                v[e] = j == 0 ? r0 - r2 : j == 1 ? r1 + r2 : j == 2 ? r2 - r1 : r1 - r3;
            }
            uint4 ph, pm, pl;
            wx_split8(v, ph, pm, pl);
            unsigned char* o = dst + (size_t)(4 * a + j) * pos_stride;
            *reinterpret_cast<uint4*>(o) = ph;
            *reinterpret_cast<uint4*>(o + WX_PIECE) = pm;
            *reinterpret_cast<uint4*>(o + 2 * WX_PIECE) = pl;
        }
    }
}

// U[p = 4 i + j][k][c] = (G g G^T)[i][j] in float64 rounded once to float32 (identical values to wino_pack_kernel, csrc/winograd.hip),
// then split into x6t records [16][cin/16][rbt][3][1 KB] (rows = output channels, padded with zeros to 32 rbt).
// One wave = (chunk, block of 32 output channels); g: OIHW [cout][cin][3][3]; `scale` as in wino_pack_kernel (may be NULL).
__global__ __launch_bounds__(256)
void wino_pack_x6t_kernel(const float* __restrict__ g, const float* __restrict__ scale, unsigned char* __restrict__ urec, int cout,
                          int cin, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= rbt * K16) return;
    const int rb = wave % rbt, chunk = wave / rbt;
    const int k = rb * 32 + (lane & 31);
    const int c0 = chunk * 16 + 8 * (lane >> 5);
    float u[16][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        double w[3][3];
        if (k < cout) {
            const float* gp = g + ((size_t)k * cin + c0 + e) * 9;
            const float sc = scale ? scale[k] : 1.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) w[a][b] = (double)(scale ? gp[a * 3 + b] * sc : gp[a * 3 + b]);
        } else {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) w[a][b] = 0.0;
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
            u[4 * a + 0][e] = (float)r[a][0];
            u[4 * a + 1][e] = (float)(0.5 * (r[a][0] + r[a][1] + r[a][2]));
            u[4 * a + 2][e] = (float)(0.5 * (r[a][0] - r[a][1] + r[a][2]));
            u[4 * a + 3][e] = (float)r[a][2];
        }
    }
    const size_t pos_stride = (size_t)K16 * rbt * WX_RB;
    unsigned char* dst = urec + ((size_t)chunk * rbt + rb) * WX_RB + lane * 16;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        uint4 ph, pm, pl;
        wx_split8(u[p], ph, pm, pl);
        unsigned char* o = dst + (size_t)p * pos_stride;
        *reinterpret_cast<uint4*>(o) = ph;
        *reinterpret_cast<uint4*>(o + WX_PIECE) = pm;
        *reinterpret_cast<uint4*>(o + 2 * WX_PIECE) = pl;
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
bool conv3x3_uses_winograd_x6(int cin, int cout) { return cin >= 256 && cin % 16 == 0 && cout >= 256 && cout % 256 == 0; }

static inline bool wx_shape_ok(int N, int H, int W, int cin, int cout)
{
    return N >= 1 && H >= 1 && W >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 4 && cout % 4 == 0 &&
           (size_t)N * cdiv(H, 2) * cdiv(W, 2) * 16 * (size_t)(cin > cout ? cin : cout) < ((size_t)1 << 31);
}

static inline int wx_tiles_padded(int T) { return cdiv(T, gemm_x6t_row_tile(T)) * gemm_x6t_row_tile(T); }
static inline int wx_cout_padded(int cout) { return cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout); }

size_t conv3x3_winograd_x6_pack_bytes(int cout, int cin)
{
    if (cin < 16 || cin % 16 != 0 || cout < 1) return 0;
    return 16 * x6t_record_bytes(wx_cout_padded(cout), cin);
}

struct WxPlan { size_t v_bytes, m_bytes, g_bytes; int T, Tp; };

static WxPlan wx_plan(int N, int H, int W, int cin, int cout)
{
    WxPlan p;
    p.T = N * cdiv(H, 2) * cdiv(W, 2);
    p.Tp = wx_tiles_padded(p.T);
    p.v_bytes = 16 * x6t_record_bytes(p.Tp, cin);
    p.m_bytes = (size_t)16 * p.T * cout * sizeof(float);
    p.g_bytes = gemm_x6t_workspace_bytes(p.T, cout, cin, 16);
    return p;
}

size_t conv3x3_winograd_x6_workspace_bytes(int N, int H, int W, int cin, int cout)
{
    if (!wx_shape_ok(N, H, W, cin, cout)) return 0;
    const WxPlan p = wx_plan(N, H, W, cin, cout);
    return p.v_bytes + p.m_bytes + p.g_bytes;
}

int launch_pack_conv3x3_winograd_x6(const float* w, const float* scale, void* urec, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 16 || cin % 16 != 0) return FRCNN_EINVAL;
    const int rbt = wx_cout_padded(cout) / 32, K16 = cin / 16;
    const int waves = rbt * K16;
    hipLaunchKernelGGL(wino_pack_x6t_kernel, dim3(cdiv(waves, 4)), dim3(256), 0, s, w, scale, static_cast<unsigned char*>(urec), cout, cin,
                       rbt, K16);
    return check_launch();
}

// The three launches of one layer, separately callable so that the fused forward can time them per class.
int launch_winograd_x6_input(const float* x, void* vrec, int N, int H, int W, int cin, hipStream_t s)
{
    const int tw = cdiv(W, 2), tpi = cdiv(H, 2) * tw, T = N * tpi, rbt = wx_tiles_padded(T) / 32, K16 = cin / 16;
    const long long waves = (long long)rbt * K16;
    if (waves > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(wino_input_x6t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, static_cast<unsigned char*>(vrec),
                       H, W, cin, tw, tpi, T, rbt, K16);
    return check_launch();
}

int launch_winograd_x6_gemm(const void* vrec, const void* urec, float* M, int N, int H, int W, int cin, int cout, void* gws, size_t gws_bytes,
                            hipStream_t s)
{
    const int T = N * cdiv(H, 2) * cdiv(W, 2), Tp = wx_tiles_padded(T), Np = wx_cout_padded(cout);
    return launch_gemm_x6t(vrec, Tp, x6t_record_bytes(Tp, cin), urec, Np, x6t_record_bytes(Np, cin), nullptr, nullptr, M, cout,
                           (size_t)T * cout, T, cout, cin, 16, 0u, gws, gws_bytes, s);
}

int winograd_x6_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, void** V, float** M, void** G,
                     size_t* g_bytes)
{
    if (!wx_shape_ok(N, H, W, cin, cout)) return FRCNN_EUNSUPPORTED;
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    const WxPlan p = wx_plan(N, H, W, cin, cout);
    if (ws == nullptr || ws_bytes < p.v_bytes + p.m_bytes + p.g_bytes) return FRCNN_EINVAL;
    unsigned char* base = static_cast<unsigned char*>(ws);
    *V = base;
    *M = reinterpret_cast<float*>(base + p.v_bytes);
    *G = p.g_bytes ? base + p.v_bytes + p.m_bytes : nullptr;
    *g_bytes = p.g_bytes;
    return FRCNN_OK;
}

int launch_conv3x3_winograd_x6(const float* x, const void* urec, const float* b, float* y, int N, int H, int W, int cin, int cout,
                               unsigned flags, void* ws, size_t ws_bytes, hipStream_t s)
{
    void *V = nullptr, *G = nullptr;
    float* M = nullptr;
    size_t gb = 0;
    int rc = winograd_x6_plan(N, H, W, cin, cout, flags, ws, ws_bytes, &V, &M, &G, &gb);
    if (rc) return rc;
    if ((rc = launch_winograd_x6_input(x, V, N, H, W, cin, s)) != FRCNN_OK) return rc;
    if ((rc = launch_winograd_x6_gemm(V, urec, M, N, H, W, cin, cout, G, gb, s)) != FRCNN_OK) return rc;
    return launch_winograd_output(M, b, y, N, H, W, cout, flags, s);
}

}  // namespace frcnn
