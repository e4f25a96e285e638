// tools/micro/split_fill.hip -- development aid (round 6): what does ONE STEP of wino_x3d_kernel's loop (six v_mfma_f32_32x32x16_f16 + the
// operand formation of 8 values per lane: B^T d B adds, scale, two-term fp16 split) cost with the split written in different instructions?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/split_fill.hip -o /tmp/split_fill && /tmp/split_fill
// One wave per SIMD (256 threads, 512 registers), every instruction `asm volatile` (program order = source order), fillers spread over the six
// MFMA gaps of a step the way the kernel spreads them.  Part 1: NF independent instructions of ONE class per gap.  Part 2: whole steps --
//   A  the kernel today: 8 v_fma (r), 8 v_add/v_sub (t), 8 v_fma_mixlo/hi (hi), 8 v_fma_mixlo/hi (lo), 4 ds_read_b128        = 36 fillers
//   B  no v_fma_mix: r pre-scaled (4 v_mul), 8 add/fma (t s), 4 v_cvt_pk_f16_f32 (hi), 8 v_cvt_f32_f16 (4 SDWA), 8 v_sub, 4 v_cvt_pk (lo) = 48
//   D  hi by v_cvt_pk_f16_f32 from pre-scaled t, lo by v_fma_mixlo/hi                                                              = 36
//   M  MFMAs only;   V  variant A / B without the MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define A1(s, ...) asm volatile(s : __VA_ARGS__)

struct St {
    float r[4][8];        // the four r columns
    float rs[2][8];       // scaled columns 1, 2 (variants B, D)
    float t[8], hf[8], l[8];
    unsigned hi[4], lo[4];
    f32x4 d0, d1, d2, d3;
    float s, sgn;
    unsigned lds;
};

__device__ __forceinline__ void rd(St& x, int k)
{
    if (k == 0) { A1("ds_read_b128 %0, %1", "=v"(x.d0) : "v"(x.lds)); A1("ds_read_b128 %0, %1 offset:2048", "=v"(x.d1) : "v"(x.lds)); }
    else        { A1("ds_read_b128 %0, %1 offset:16", "=v"(x.d2) : "v"(x.lds)); A1("ds_read_b128 %0, %1 offset:2064", "=v"(x.d3) : "v"(x.lds)); }
}
// r column c, channels [4 half, 4 half + 4) from the data read one step earlier (the kernel waits with lgkmcnt before; here too)
__device__ __forceinline__ void make_r(St& x, int c, int half)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int e = 0; e < 4; ++e) {
        if (half == 0) A1("v_fma_f32 %0, %1, %2, %3", "=v"(x.r[c][e]) : "v"(x.d1[e]), "v"(x.sgn), "v"(x.d0[e]));
        else           A1("v_fma_f32 %0, %1, %2, %3", "=v"(x.r[c][4 + e]) : "v"(x.d3[e]), "v"(x.sgn), "v"(x.d2[e]));
    }
}
__device__ __forceinline__ void adds(St& x, int q0)                       // t[q0 .. q0 + 3] = r[1] - r[3]
{
    for (int q = q0; q < q0 + 4; ++q) A1("v_sub_f32 %0, %1, %2", "=v"(x.t[q]) : "v"(x.r[1][q]), "v"(x.r[3][q]));
}
__device__ __forceinline__ void adds_scaled(St& x, int q0)                // t s = fma(r3, -s, rs1): one rounding, the scaled t
{
    for (int q = q0; q < q0 + 4; ++q) A1("v_fma_f32 %0, %1, %2, %3", "=v"(x.t[q]) : "v"(x.r[3][q]), "v"(x.sgn), "v"(x.rs[0][q]));
}
__device__ __forceinline__ void mix_hi(St& x, int p0)                     // pairs p0, p0 + 1 from t[2 p0 .. 2 p0 + 3], ONE asm statement as in the kernel
{
    asm volatile("v_fma_mixlo_f16 %0, %2, %6, 0\n\tv_fma_mixlo_f16 %1, %4, %6, 0\n\tv_fma_mixhi_f16 %0, %3, %6, 0\n\tv_fma_mixhi_f16 %1, %5, %6, 0"
                 : "=&v"(x.hi[p0]), "=&v"(x.hi[p0 + 1]) : "v"(x.t[2 * p0]), "v"(x.t[2 * p0 + 1]), "v"(x.t[2 * p0 + 2]), "v"(x.t[2 * p0 + 3]), "v"(x.s));
}
__device__ __forceinline__ void mix_lo(St& x, int p0, float mul)
{
    asm volatile("v_fma_mixlo_f16 %0, %2, %6, -%7 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\tv_fma_mixlo_f16 %1, %4, %6, -%8 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mixhi_f16 %0, %3, %6, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\tv_fma_mixhi_f16 %1, %5, %6, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                 : "=&v"(x.lo[p0]), "=&v"(x.lo[p0 + 1])
                 : "v"(x.t[2 * p0]), "v"(x.t[2 * p0 + 1]), "v"(x.t[2 * p0 + 2]), "v"(x.t[2 * p0 + 3]), "v"(mul), "v"(x.hi[p0]), "v"(x.hi[p0 + 1]));
}
// lo through float32: d = t - float(hi) exactly (v_fma_mix_f32 reads the fp16 half in place, full rate), then ONE v_cvt_pk_f16_f32 per pair
__device__ __forceinline__ void lo_f32(St& x, int p)
{
    A1("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,0] op_sel_hi:[0,0,1]", "=v"(x.l[2 * p]) : "v"(x.t[2 * p]), "v"(x.hi[p]));
    A1("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]", "=v"(x.l[2 * p + 1]) : "v"(x.t[2 * p + 1]), "v"(x.hi[p]));
}
__device__ __forceinline__ void pk(unsigned& dst, float a, float b) { A1("v_cvt_pk_f16_f32 %0, %1, %2", "=v"(dst) : "v"(a), "v"(b)); }
__device__ __forceinline__ void unpack(St& x, int p)
{
    A1("v_cvt_f32_f16_e32 %0, %1", "=v"(x.hf[2 * p]) : "v"(x.hi[p]));
    A1("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "=v"(x.hf[2 * p + 1]) : "v"(x.hi[p]));
}
__device__ __forceinline__ void subs(St& x, int q0)
{
    for (int q = q0; q < q0 + 4; ++q) A1("v_sub_f32 %0, %1, %2", "=v"(x.l[q]) : "v"(x.t[q]), "v"(x.hf[q]));
}
__device__ __forceinline__ void muls(St& x, int c, int q0, int n)
{
    for (int q = q0; q < q0 + n; ++q) A1("v_mul_f32 %0, %1, %2", "=v"(x.rs[c][q]) : "v"(x.r[1 + c][q]), "v"(x.s));
}

template <int KIND> __device__ __forceinline__ void gap(St& x, int g)
{
    if (KIND == 'A') {
        switch (g) {
        case 0: rd(x, 0); adds(x, 0); break;
        case 1: mix_hi(x, 0); make_r(x, 2, 1); break;
        case 2: mix_lo(x, 0, x.s); rd(x, 1); break;
        case 3: adds(x, 4); mix_hi(x, 2); break;
        case 4: mix_lo(x, 2, x.s); break;
        default: make_r(x, 0, 0); break;
        }
    } else if (KIND == 'B') {
        switch (g) {
        case 0: rd(x, 0); adds_scaled(x, 0); pk(x.hi[0], x.t[0], x.t[1]); pk(x.hi[1], x.t[2], x.t[3]); break;
        case 1: unpack(x, 0); unpack(x, 1); subs(x, 0); break;
        case 2: pk(x.lo[0], x.l[0], x.l[1]); pk(x.lo[1], x.l[2], x.l[3]); rd(x, 1); adds_scaled(x, 4); break;
        case 3: pk(x.hi[2], x.t[4], x.t[5]); pk(x.hi[3], x.t[6], x.t[7]); unpack(x, 2); unpack(x, 3); muls(x, 0, 0, 2); break;
        case 4: subs(x, 4); pk(x.lo[2], x.l[4], x.l[5]); pk(x.lo[3], x.l[6], x.l[7]); muls(x, 0, 2, 2); break;
        default: make_r(x, 2, 1); make_r(x, 0, 0); break;
        }
    } else if (KIND == 'F') {           // D with lo = v_cvt_pk(t - hi in float32): 4 more plain instructions, 8 fewer v_fma_mix, 4 more v_cvt_pk
        switch (g) {
        case 0: rd(x, 0); adds_scaled(x, 0); pk(x.hi[0], x.t[0], x.t[1]); pk(x.hi[1], x.t[2], x.t[3]); break;
        case 1: lo_f32(x, 0); lo_f32(x, 1); muls(x, 0, 0, 2); break;
        case 2: rd(x, 1); pk(x.lo[0], x.l[0], x.l[1]); pk(x.lo[1], x.l[2], x.l[3]); adds_scaled(x, 4); break;
        case 3: pk(x.hi[2], x.t[4], x.t[5]); pk(x.hi[3], x.t[6], x.t[7]); lo_f32(x, 2); break;
        case 4: lo_f32(x, 3); make_r(x, 2, 1); muls(x, 0, 2, 2); break;
        default: pk(x.lo[2], x.l[4], x.l[5]); pk(x.lo[3], x.l[6], x.l[7]); make_r(x, 0, 0); break;
        }
    } else if (KIND == 'D') {
        switch (g) {
        case 0: rd(x, 0); adds_scaled(x, 0); pk(x.hi[0], x.t[0], x.t[1]); pk(x.hi[1], x.t[2], x.t[3]); break;
        case 1: mix_lo(x, 0, 1.0f); muls(x, 0, 0, 2); break;
        case 2: rd(x, 1); adds_scaled(x, 4); break;
        case 3: pk(x.hi[2], x.t[4], x.t[5]); pk(x.hi[3], x.t[6], x.t[7]); mix_lo(x, 2, 1.0f); break;
        case 4: make_r(x, 2, 1); muls(x, 0, 2, 2); break;
        default: make_r(x, 0, 0); break;
        }
    }
}

// part 1: NF instructions of one class per gap, independent of each other (16 destinations in turn)
template <int CLS, int NF> __device__ __forceinline__ void cls_gap(float (&v)[16], unsigned (&h)[16], float c1, float c2, int g)
{
#pragma unroll
    for (int i = 0; i < NF; ++i) {
        const int k = (g * NF + i) & 15, k2 = (k + 5) & 15;
        if (CLS == 0) A1("v_fma_f32 %0, %1, %2, %3", "=v"(v[k]) : "v"(v[k2]), "v"(c1), "v"(c2));
        else if (CLS == 1) A1("v_cvt_pk_f16_f32 %0, %1, %2", "=v"(h[k]) : "v"(v[k2]), "v"(v[k]));
        else if (CLS == 2) A1("v_cvt_f32_f16_e32 %0, %1", "=v"(v[k]) : "v"(h[k2]));
        else if (CLS == 3) A1("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "=v"(v[k]) : "v"(h[k2]));
        else if (CLS == 4) A1("v_fma_mixlo_f16 %0, %1, %2, 0", "+v"(h[k]) : "v"(v[k2]), "v"(c1));
        else if (CLS == 5) A1("v_fma_mixhi_f16 %0, %1, %2, 0", "+v"(h[k]) : "v"(v[k2]), "v"(c1));
        else if (CLS == 6) A1("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]", "+v"(h[k]) : "v"(v[k2]), "v"(c1), "v"(h[k2]));
        else if (CLS == 7) A1("v_sub_f32 %0, %1, %2", "=v"(v[k]) : "v"(v[k2]), "v"(c2));
        else if (CLS == 8) A1("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]", "=v"(v[k]) : "v"(v[k2]), "v"(c1), "v"(h[k2]));
        else if (CLS == 9) A1("v_and_b32 %0, %1, %2", "=v"(h[k]) : "v"(h[k2]), "v"(h[(k + 9) & 15]));
    }
}

template <int KIND, int CLS, int NF, bool MFMA>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 8192; i += 256) lds[i] = 1.0f + 0.001f * (float)i;
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    St x;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 8; ++e) x.r[c][e] = 1.0f + 0.001f * (float)(tid + 8 * c + e);
    for (int c = 0; c < 2; ++c) for (int e = 0; e < 8; ++e) x.rs[c][e] = 0.5f + 0.002f * (float)(tid + e);
    for (int e = 0; e < 8; ++e) { x.t[e] = 0.25f * e; x.hf[e] = 0.f; x.l[e] = 0.f; }
    for (int p = 0; p < 4; ++p) { x.hi[p] = 0; x.lo[p] = 0; }
    x.d0 = x.d1 = x.d2 = x.d3 = f32x4{0.f, 0.f, 0.f, 0.f};
    x.s = 4.0f; x.sgn = -1.0f;
    x.lds = (unsigned)(size_t)(lds) + (unsigned)((tid & 63) * 80 + (tid >> 6) * 32);
    float v[16]; unsigned h[16];
    for (int i = 0; i < 16; ++i) { v[i] = 1.0f + 0.001f * (float)(tid + i); h[i] = 0x3c003c00u + (unsigned)i; }
    const float c1 = 1.0001f, c2 = 1e-6f * (float)tid;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)(tid & 7)); b[i] = (_Float16)0.5f; }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            if (MFMA) {
                if (KIND != 0) {
                    // the operand is the pair formed in the previous step, as in the kernel
                    f16x8 bb = __builtin_bit_cast(f16x8, uint4{x.hi[0], x.hi[1], x.hi[2], x.hi[3]});
                    f16x8 bl = __builtin_bit_cast(f16x8, uint4{x.lo[0], x.lo[1], x.lo[2], x.lo[3]});
                    if (g < 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(bb));
                    else       asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(bl));
                } else {
                    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
                }
            }
            if (KIND != 0) gap<KIND>(x, g);
            else cls_gap<CLS, NF>(v, h, c1, c2, g);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i] + (float)h[i];
    for (int e = 0; e < 8; ++e) s += x.t[e] + x.hf[e] + x.l[e] + x.r[0][e] + x.r[2][e] + x.rs[0][e];
    for (int p = 0; p < 4; ++p) s += (float)(x.hi[p] + x.lo[p]);
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

// part 3: the FILTER STREAM.  48 MFMAs per iteration (one 16-channel chunk of wino_x3d_kernel: 64 tiles x 64 output channels) with NL
// one-kilobyte fragment loads per wave (buffer_load_dwordx4, L2-resident 2 MB region shared by all CUs, each fragment consumed as an MFMA
// operand one iteration later): NL = 16 is today's block, NL = 32 what a 32-tile x 128-channel block would stream for the same 48 MFMAs.
template <int NL>
__global__ __launch_bounds__(256, 1) void kstream(const unsigned char* __restrict__ blob, float* out, unsigned long long* cyc, int iters)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    constexpr int NU = NL > 0 ? NL : 1;
    f16x8 U[2][NU], b;
    for (int i = 0; i < 8; ++i) b[i] = (_Float16)0.5f;
    for (int n = 0; n < NU; ++n) for (int i = 0; i < 8; ++i) { U[0][n][i] = (_Float16)0.25f; U[1][n][i] = (_Float16)0.125f; }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(blob), 0, 2 << 20, 0x00020000);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int so = (((it + par) * NU * 4) & 2047) * 1024 + wave * 1024;
#pragma unroll
            for (int m = 0; m < 48; ++m) {
                acc[m & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(U[par][m % NU], b, acc[m & 15], 0, 0, 0);
                if (NL > 0 && m % (48 / NU) == 0) {
                    const int n = m / (48 / NU);
                    if (n < NL)
                        U[par ^ 1][n] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, so + n * 4096, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int NL> static void run_stream(int blocks, const unsigned char* blob, float* out, unsigned long long* cyc)
{
    const int iters = 2000;
    std::vector<unsigned long long> h((size_t)blocks * 4);
    double avg = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((kstream<NL>), dim3(blocks), dim3(256), 0, 0, blob, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
        avg = 0;
        for (size_t i = 0; i < h.size(); ++i) avg += (double)h[i];
        avg /= (double)h.size();
    }
    printf("filter stream: 48 MFMAs + %2d one-kilobyte loads per wave and iteration (%3d KB per CU): %.0f cycles per iteration (1536 = the MFMAs) -> %.1f B / clock / CU\n",
           NL, NL * 4, avg / iters, NL * 4096.0 / (avg / iters));
}

template <int KIND, int CLS, int NF, bool MFMA>
static double run(int blocks, float* out, unsigned long long* cyc)
{
    const int iters = 2000;
    std::vector<unsigned long long> h((size_t)blocks * 4);
    double avg = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, (size_t)blocks * 4 * 8);
        hipLaunchKernelGGL((k<KIND, CLS, NF, MFMA>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost);
        avg = 0;
        for (size_t i = 0; i < h.size(); ++i) avg += (double)h[i];
        avg /= (double)h.size();
    }
    return avg / iters;                 // cycles per step of six gaps
}

template <int CLS> static void cls_rows(const char* name, int blocks, float* out, unsigned long long* cyc)
{
    const double n0 = run<0, CLS, 4, false>(blocks, out, cyc), n1 = run<0, CLS, 8, false>(blocks, out, cyc);
    const double m2 = run<0, CLS, 2, true>(blocks, out, cyc), m4 = run<0, CLS, 4, true>(blocks, out, cyc), m6 = run<0, CLS, 6, true>(blocks, out, cyc);
    const double m8 = run<0, CLS, 8, true>(blocks, out, cyc), m12 = run<0, CLS, 12, true>(blocks, out, cyc);
    printf("%-28s alone: %.2f cyc/instr | beside MFMAs, cycles per gap at 2 / 4 / 6 / 8 / 12 per gap: %.1f %.1f %.1f %.1f %.1f  -> slope 8..12: %.2f cyc/instr\n",
           name, (n1 - n0) / 24.0, m2 / 6, m4 / 6, m6 / 6, m8 / 6, m12 / 6, (m12 - m8) / 24.0);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
    printf("MFMAs only: %.1f cycles per step of six\n", run<0, 0, 0, true>(blocks, out, cyc));
    cls_rows<0>("v_fma_f32", blocks, out, cyc);
    cls_rows<7>("v_sub_f32", blocks, out, cyc);
    cls_rows<9>("v_and_b32", blocks, out, cyc);
    cls_rows<1>("v_cvt_pk_f16_f32", blocks, out, cyc);
    cls_rows<2>("v_cvt_f32_f16", blocks, out, cyc);
    cls_rows<3>("v_cvt_f32_f16 sdwa WORD_1", blocks, out, cyc);
    cls_rows<4>("v_fma_mixlo_f16 (2 src)", blocks, out, cyc);
    cls_rows<5>("v_fma_mixhi_f16 (2 src)", blocks, out, cyc);
    cls_rows<6>("v_fma_mixlo_f16 (3 src)", blocks, out, cyc);
    cls_rows<8>("v_fma_mix_f32 (f16 c)", blocks, out, cyc);
    printf("step A (kernel today, 36 fillers, 16 v_fma_mix): %.1f cycles per step with MFMAs, %.1f without\n",
           run<'A', 0, 0, true>(blocks, out, cyc), run<'A', 0, 0, false>(blocks, out, cyc));
    printf("step B (no v_fma_mix, 48 fillers):               %.1f cycles per step with MFMAs, %.1f without\n",
           run<'B', 0, 0, true>(blocks, out, cyc), run<'B', 0, 0, false>(blocks, out, cyc));
    printf("step D (hi by v_cvt_pk, lo by v_fma_mix, 36):    %.1f cycles per step with MFMAs, %.1f without\n",
           run<'D', 0, 0, true>(blocks, out, cyc), run<'D', 0, 0, false>(blocks, out, cyc));
    printf("step F (hi by v_cvt_pk, lo by v_fma_mix_f32 + v_cvt_pk, 40): %.1f cycles per step with MFMAs, %.1f without\n",
           run<'F', 0, 0, true>(blocks, out, cyc), run<'F', 0, 0, false>(blocks, out, cyc));
    unsigned char* blob;
    hipMalloc(&blob, 2 << 20); hipMemset(blob, 0x3c, 2 << 20);
    run_stream<0>(blocks, blob, out, cyc);
    run_stream<8>(blocks, blob, out, cyc);
    run_stream<16>(blocks, blob, out, cyc);
    run_stream<24>(blocks, blob, out, cyc);
    run_stream<48>(blocks, blob, out, cyc);
    return 0;
}
