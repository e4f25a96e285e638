// x3t.h -- device helpers of the f32x3 arithmetic (csrc/gemm_x3t.hip): row scale, two-term fp16 split, record geometry.
#pragma once
#include "common.h"

namespace frcnn {

static constexpr int HX_PIECE = 1024;                 // bytes of one (chunk, row block, term) piece: [k-half 2][row 32][8 fp16]
static constexpr int HX_RB = 2 * HX_PIECE;            // bytes of one (chunk, row block): hi piece, lo piece

__device__ __forceinline__ unsigned hx_pack2(_Float16 a, _Float16 b)
{
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// 8 consecutive (already scaled) k of one row -> the two 16-byte pieces of its record slot
__device__ __forceinline__ void hx_split8(const float (&v)[8], uint4& ph, uint4& pl)
{
    _Float16 hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // every producer bounds its row (|v| < 2^15); the clamp keeps a value a producer failed to bound from becoming hi = inf,
        // lo = -inf -> NaN in the GEMM: it saturates instead
        const float vj = v[j] > 65504.f ? 65504.f : (v[j] < -65504.f ? -65504.f : v[j]);      // a NaN stays a NaN
        hi[j] = (_Float16)vj;                         // round to nearest even
        lo[j] = (_Float16)(vj - (float)hi[j]);
    }
    ph = make_uint4(hx_pack2(hi[0], hi[1]), hx_pack2(hi[2], hi[3]), hx_pack2(hi[4], hi[5]), hx_pack2(hi[6], hi[7]));
    pl = make_uint4(hx_pack2(lo[0], lo[1]), hx_pack2(lo[2], lo[3]), hx_pack2(lo[4], lo[5]), hx_pack2(lo[6], lo[7]));
}

// scale pair of a row whose magnitudes are bounded by mx: mult = 2^e with mx 2^e in [2^14, 2^15), inv = 2^-e (e clamped to +-100; a
// zero row gets 1).  Both exact powers of two.
__device__ __forceinline__ void hx_row_scale(float mx, float& mult, float& inv)
{
    int e = 0;
    if (mx > 0.f) {
        const int E = (int)((__float_as_uint(mx) >> 23) & 0xFFu) - 127;      // floor(log2 mx) for normal mx; -127 for subnormals
        e = 14 - E;
        e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    mult = __uint_as_float((unsigned)(127 + e) << 23);
    inv = __uint_as_float((unsigned)(127 - e) << 23);
}

// 2^e from 2^-e (exact powers of two within +-100)
__device__ __forceinline__ float hx_mult_of_inv(float inv) { return __uint_as_float((254u << 23) - __float_as_uint(inv)); }

}  // namespace frcnn
