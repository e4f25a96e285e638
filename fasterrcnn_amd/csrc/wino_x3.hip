// wino_x3.hip -- 3x3 stride-1 "same" convolution + bias + ReLU (+ MaxPool2d(2)) of the reference's 512-channel layers
// (pytorch/FasterRCNN/models/vgg16.py:89-96 conv4_1 ... conv5_3, models/rpn.py:88 the RPN trunk; ResNet: models/resnet.py:110 layer4's
// 3x3 over the RoI maps) as the Winograd F(2x2,3x3) layer of csrc/wino_x6.hip with its 16 position GEMMs in the f32x3 arithmetic
// (csrc/gemm_x3t.hip: two fp16 terms per row-scaled operand, three fp16 MFMAs per product -- half the matrix instructions of f32x6).
// Same transforms in the same float32 operation order (V and U are bit-identical to csrc/winograd.hip's before the split).
//
// Row scales.  A GEMM row of V is one Winograd tile at one position, its K = cin channels.  The scale is chosen per TILE (shared by
// the 16 positions and all channels, so it factors out of every dot product): |V| <= 4 max|d| over the tile's 4 x 4 input patch, and
// max|d| comes from a per-pixel channel maximum computed by one small pass over the layer input:
//   0. pixel_absmax_kernel     x [P][cin] -> cmax [P]                     (one wave per pixel; 19 MB read for a 75 x 125 x 512 map)
//   1. wino_input_x3t_kernel   x, cmax -> V as x3t records [16][cin/16][Tp/32][2][1 KB] + vinv [Tp] (2^-e per tile)
//   2. gemm_x3t_kernel         16 batched GEMMs M_p = V_p U_p^T; U's rows (position, output channel) carry their own scales
//   3. wino_output_kernel      (csrc/winograd.hip)
// Filter bank: the float32 bank [16][cout][cin] of launch_pack_conv3x3_winograd, then launch_pack_rows_x3t (records + scales in one blob).
#include "x3t.h"

namespace frcnn {

// cmax[p] = max_c |x[p][c]|, one wave per pixel (C % 4 == 0)
__global__ __launch_bounds__(256)
void pixel_absmax_kernel(const float* __restrict__ x, float* __restrict__ cmax, long long P, int C)
{
    const int lane = threadIdx.x & 63;
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= P) return;
    const float* src = x + (size_t)p * C;
    float mx = 0.f;
    for (int c = 4 * lane; c < C; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + c);
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    if (lane == 0) cmax[p] = mx;
}

// One wave = (16-channel chunk, block of 32 tiles): lane l = tile (l & 31), channels 8 (l >> 5) .. + 7 of the chunk
// (wino_input_x6t_kernel's work split; B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]).
__global__ __launch_bounds__(256)
void wino_input_x3t_kernel(const float* __restrict__ x, const float* __restrict__ cmax, unsigned char* __restrict__ vrec,
                           float* __restrict__ vinv, int H, int W, int cin, int tw, int tpi, int T, int rbt, int K16)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= rbt * K16) return;
    const int chunk = wave % K16, rb = wave / K16;
    const int tile = rb * 32 + (lane & 31);
    const int c = chunk * 16 + 8 * (lane >> 5);
    const bool live = tile < T;
    const int img = live ? tile / tpi : 0, tin = live ? tile - img * tpi : 0;
    const int ty = tin / tw, tx = tin - ty * tw;
    const int y0 = 2 * ty - 1, x0 = 2 * tx - 1;
    x += (size_t)img * H * W * cin;
    cmax += (size_t)img * H * W;
    float d[4][4][8];
    float dmax = 0.f;
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
                dmax = fmaxf(dmax, cmax[(size_t)yy * W + xx]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[a][b][e] = v0[e]; d[a][b][4 + e] = v1[e]; }
        }
    }
    // |B^T d B| <= 4 max|d|: the tile's scale, the same in every wave (chunk) that handles the tile
    float mult, inv;
    hx_row_scale(4.0f * dmax, mult, inv);
    if (chunk == 0 && lane < 32) vinv[tile] = inv;
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
    const size_t pos_stride = (size_t)K16 * rbt * HX_RB;
    unsigned char* dst = vrec + ((size_t)chunk * rbt + rb) * HX_RB + lane * 16;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float r0 = d[a][0][e], r1 = d[a][1][e], r2 = d[a][2][e], r3 = d[a][3][e];
                const float t = j == 0 ? r0 - r2 : j == 1 ? r1 + r2 : j == 2 ? r2 - r1 : r1 - r3;      // the float32 V of csrc/winograd.hip
                v[e] = t * mult;                                                                   // exact (power of two)
            }
            uint4 ph, pl;
            hx_split8(v, ph, pl);
            unsigned char* o = dst + (size_t)(4 * a + j) * pos_stride;
            *reinterpret_cast<uint4*>(o) = ph;
            *reinterpret_cast<uint4*>(o + HX_PIECE) = pl;
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
static inline bool wh_shape_ok(int N, int H, int W, int cin, int cout)
{
    return N >= 1 && H >= 1 && W >= 1 && cin >= 16 && cin % 16 == 0 && cout >= 4 && cout % 4 == 0 &&
           (size_t)N * cdiv(H, 2) * cdiv(W, 2) * 16 * (size_t)(cin > cout ? cin : cout) < ((size_t)1 << 31);
}
static inline size_t wh_align(size_t v) { return (v + 255) / 256 * 256; }
static inline int wh_tiles_padded(int T) { return cdiv(T, gemm_x6t_row_tile(T)) * gemm_x6t_row_tile(T); }
static inline int wh_cout_padded(int cout) { return cdiv(cout, gemm_x6t_col_tile(cout)) * gemm_x6t_col_tile(cout); }

// A packed x3t operand = its record arrays followed by its scale arrays: [batches][x3t_record_bytes(rows_padded, K)] then
// [batches][rows_padded] float32 (2^-e per row)
size_t x3t_blob_bytes(int rows_padded, int K, int batches)
{
    return (size_t)batches * (x3t_record_bytes(rows_padded, K) + (size_t)rows_padded * sizeof(float));
}

int launch_pack_rows_x3t(const float* a, int lda, size_t a_batch_floats, void* blob, int R, int rows_padded, int K, int batches, hipStream_t s)
{
    if (!a || !blob) return FRCNN_EINVAL;
    float* inv = reinterpret_cast<float*>(static_cast<unsigned char*>(blob) + (size_t)batches * x3t_record_bytes(rows_padded, K));
    int rc = launch_rows_scale_x3t(a, lda, a_batch_floats, inv, R, rows_padded, K, batches, s);
    if (rc) return rc;
    return launch_split_rows_x3t(a, lda, a_batch_floats, inv, blob, R, rows_padded, K, batches, s);
}

size_t conv3x3_winograd_x3_pack_bytes(int cout, int cin)
{
    if (cin < 16 || cin % 16 != 0 || cout < 1) return 0;
    return x3t_blob_bytes(wh_cout_padded(cout), cin, 16);
}

// u_f32: the float32 bank [16][cout][cin] of launch_pack_conv3x3_winograd (caller's scratch or cache)
int launch_pack_conv3x3_winograd_x3(const float* u_f32, void* ublob, int cout, int cin, hipStream_t s)
{
    if (cout < 1 || cin < 16 || cin % 16 != 0) return FRCNN_EINVAL;
    return launch_pack_rows_x3t(u_f32, cin, (size_t)cout * cin, ublob, cout, wh_cout_padded(cout), cin, 16, s);
}

struct WhPlan { size_t v_bytes, vinv_bytes, cmax_bytes, m_bytes, g_bytes; int T, Tp; };

static WhPlan wh_plan(int N, int H, int W, int cin, int cout)
{
    WhPlan p;
    p.T = N * cdiv(H, 2) * cdiv(W, 2);
    p.Tp = wh_tiles_padded(p.T);
    p.v_bytes = 16 * x3t_record_bytes(p.Tp, cin);
    p.vinv_bytes = wh_align((size_t)p.Tp * sizeof(float));
    p.cmax_bytes = wh_align((size_t)N * H * W * sizeof(float));
    p.m_bytes = (size_t)16 * p.T * cout * sizeof(float);
    p.g_bytes = gemm_x3t_workspace_bytes(p.T, cout, cin, 16);
    return p;
}

size_t conv3x3_winograd_x3_workspace_bytes(int N, int H, int W, int cin, int cout)
{
    if (!wh_shape_ok(N, H, W, cin, cout)) return 0;
    const WhPlan p = wh_plan(N, H, W, cin, cout);
    return p.v_bytes + p.vinv_bytes + p.cmax_bytes + p.m_bytes + p.g_bytes;
}

int winograd_x3_plan(int N, int H, int W, int cin, int cout, unsigned flags, void* ws, size_t ws_bytes, void** V, float** vinv, float** cmax,
                     float** M, void** G, size_t* g_bytes)
{
    if (!wh_shape_ok(N, H, W, cin, cout)) return FRCNN_EUNSUPPORTED;
    if ((flags & FRCNN_POOL2) && (H < 2 || W < 2)) return FRCNN_EINVAL;
    const WhPlan p = wh_plan(N, H, W, cin, cout);
    if (ws == nullptr || ws_bytes < p.v_bytes + p.vinv_bytes + p.cmax_bytes + p.m_bytes + p.g_bytes) return FRCNN_EINVAL;
    unsigned char* base = static_cast<unsigned char*>(ws);
    *V = base;
    *vinv = reinterpret_cast<float*>(base + p.v_bytes);
    *cmax = reinterpret_cast<float*>(base + p.v_bytes + p.vinv_bytes);
    *M = reinterpret_cast<float*>(base + p.v_bytes + p.vinv_bytes + p.cmax_bytes);
    *G = p.g_bytes ? base + p.v_bytes + p.vinv_bytes + p.cmax_bytes + p.m_bytes : nullptr;
    *g_bytes = p.g_bytes;
    return FRCNN_OK;
}

int launch_pixel_absmax(const float* x, float* cmax, long long pixels, int C, hipStream_t s)
{
    if (pixels < 1 || C < 4 || C % 4 != 0) return FRCNN_EINVAL;
    const long long blocks = (pixels + 3) / 4;
    if (blocks > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(pixel_absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, cmax, pixels, C);
    return check_launch();
}

int launch_winograd_x3_input(const float* x, float* cmax, void* vrec, float* vinv, int N, int H, int W, int cin, hipStream_t s, const float* cmax_ready)
{
    if (cmax_ready) {
        cmax = const_cast<float*>(cmax_ready);            // the producing layer's epilogue left the channel maxima behind
    } else {
        int rc = launch_pixel_absmax(x, cmax, (long long)N * H * W, cin, s);
        if (rc) return rc;
    }
    const int tw = cdiv(W, 2), tpi = cdiv(H, 2) * tw, T = N * tpi, rbt = wh_tiles_padded(T) / 32, K16 = cin / 16;
    const long long waves = (long long)rbt * K16;
    if (waves > 0x7fffffffLL) return FRCNN_EINVAL;
    hipLaunchKernelGGL(wino_input_x3t_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, x, cmax, static_cast<unsigned char*>(vrec),
                       vinv, H, W, cin, tw, tpi, T, rbt, K16);
    return check_launch();
}

int launch_winograd_x3_gemm(const void* vrec, const float* vinv, const void* ublob, float* M, int N, int H, int W, int cin, int cout, void* gws,
                            size_t gws_bytes, hipStream_t s)
{
    const int T = N * cdiv(H, 2) * cdiv(W, 2), Tp = wh_tiles_padded(T), Np = wh_cout_padded(cout);
    const size_t urec = x3t_record_bytes(Np, cin);
    const float* uinv = reinterpret_cast<const float*>(static_cast<const unsigned char*>(ublob) + 16 * urec);
    return launch_gemm_x3t(vrec, vinv, Tp, x3t_record_bytes(Tp, cin), 0, ublob, uinv, Np, urec, (size_t)Np, nullptr, nullptr, M, cout,
                           (size_t)T * cout, T, cout, cin, 16, 0u, gws, gws_bytes, s);
}

int launch_conv3x3_winograd_x3(const float* x, const void* ublob, const float* b, float* y, int N, int H, int W, int cin, int cout,
                               unsigned flags, void* ws, size_t ws_bytes, hipStream_t s, const float* cmax_ready, float* cmax_out)
{
    void *V = nullptr, *G = nullptr;
    float *M = nullptr, *vinv = nullptr, *cmax = nullptr;
    size_t gb = 0;
    int rc = winograd_x3_plan(N, H, W, cin, cout, flags, ws, ws_bytes, &V, &vinv, &cmax, &M, &G, &gb);
    if (rc) return rc;
    if ((rc = launch_winograd_x3_input(x, cmax, V, vinv, N, H, W, cin, s, cmax_ready)) != FRCNN_OK) return rc;
    if ((rc = launch_winograd_x3_gemm(V, vinv, ublob, M, N, H, W, cin, cout, G, gb, s)) != FRCNN_OK) return rc;
    return launch_winograd_output(M, b, y, N, H, W, cout, flags, s, cmax_out);
}

}  // namespace frcnn
