// tools/micro/mfma_fill2.hip -- development aid (round 5): what does a SIMD pay for NF "filler" instructions per v_mfma_f32_32x32x16_f16 gap
// with ONE wave on it (256-thread block, 512 registers) and with TWO (512-thread block, 256 registers each)?  The question behind the
// 8-wave form of the one-launch f32x3 Winograd kernel (csrc/wino_x3e.hip): the 4-wave kernel pays MFMA + fillers as a SUM.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_fill2.hip -o /tmp/mfma_fill2 && /tmp/mfma_fill2
// Every instruction is `asm volatile` (program order = source order).  Filler kinds:
//   0 v_fma_f32 on 16 independent chains          1 v_fma_mixlo_f16 / v_fma_mixhi_f16 pairs (the operand split of wino_x3d_kernel)
//   2 the kernel's mix per gap: 1 ds_read_b128, 2 v_add_f32, 2 mixlo, 2 mixhi, v_fma_f32 (8 per gap = "NF 8"; NF scales the VALU part)
//   3 v_pk_add_f32
// Modes: 0 = every wave runs MFMA + fillers;  1 = (two waves per SIMD) waves 0-3 MFMA only, waves 4-7 fillers only, same counts;
//        2 = mode 0 with s_setprio 1 on waves 4-7
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NF, int KIND>
__device__ __forceinline__ void fillers(float (&x)[16], unsigned (&h)[4], f32x4& d, const float* lds_p, float c1, float c2, int g)
{
    if (KIND == 0) {
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(g * NF + i) & 15]) : "v"(c1), "v"(c2));
    } else if (KIND == 1) {
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            if ((i & 1) == 0) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[(i >> 1) & 3]) : "v"(x[(g + i) & 15]), "v"(c1));
            else asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[(i >> 1) & 3]) : "v"(x[(g + i) & 15]), "v"(c1));
        }
    } else if (KIND == 2) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"((unsigned)(size_t)lds_p));
        constexpr int NV = NF - 1;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int r = i % 7;
            if (r == 0 || r == 3) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[(g + i) & 15]) : "v"(x[(g + i + 5) & 15]), "v"(c2));
            else if (r == 1) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[0]) : "v"(x[(g + 9) & 15]), "v"(c1));
            else if (r == 2) asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h[1]) : "v"(x[(g + 10) & 15]), "v"(c1));
            else if (r == 4) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[0]) : "v"(x[(g + 11) & 15]), "v"(c1));
            else if (r == 5) asm volatile("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h[1]) : "v"(x[(g + 12) & 15]), "v"(c1));
            else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[(g + 13) & 15]) : "v"(c1), "v"(c2), "v"(x[(g + 14) & 15]));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(x[(g + 15) & 15]) : "v"(d[0]), "v"(c2));
    } else {
        f32x2* xp = reinterpret_cast<f32x2*>(x);
#pragma unroll
        for (int i = 0; i < NF; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp[(g * NF + i) & 7]) : "v"(f32x2{c1, c2}));
    }
}

template <int NF, int KIND, int WPS>
__global__ __launch_bounds__(256 * WPS) void k(float* out, unsigned long long* cyc, int iters, int mode)
{
    __shared__ __attribute__((aligned(16))) float lds[8192];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    lds[tid] = (float)tid; lds[tid + 512] = 1.0f;
    __syncthreads();
    f32x16 acc[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = 1.0f + 0.001f * (float)(tid + i);
    unsigned h[4] = {0, 0, 0, 0};
    f32x4 d = {0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (float)(tid & 7)); b[i] = (_Float16)0.5f; }
    const float c1 = 1.0001f, c2 = 1e-6f * (float)tid;
    const float* lds_p = lds + (tid & 255) * 4;
    const bool do_mfma = !(mode == 1 && wave >= 4), do_fill = !(mode == 1 && wave < 4);
    if (mode == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma && do_fill) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 6; ++g) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
                fillers<NF, KIND>(x, h, d, lds_p, c1, c2, g);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 6; ++g) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 6; ++g) fillers<NF, KIND>(x, h, d, lds_p, c1, c2, g);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    s += (float)(h[0] + h[1] + h[2] + h[3]) + d[0] + d[1];
    out[blockIdx.x * 256 * WPS + tid] = s;
    if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NF, int KIND, int WPS>
static void run(const char* kname, int mode, int blocks, float* out, unsigned long long* cyc)
{
    const int iters = 2000;
    std::vector<unsigned long long> h((size_t)blocks * 8);
    double lo = 0, hi = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(cyc, 0, (size_t)blocks * 8 * 8);
        hipLaunchKernelGGL((k<NF, KIND, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, out, cyc, iters, mode);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, (size_t)blocks * 8 * 8, hipMemcpyDeviceToHost);
        lo = hi = 0;
        for (int bI = 0; bI < blocks; ++bI) {
            double a = 0, c = 0;
            for (int w = 0; w < 4; ++w) a += (double)h[bI * 8 + w];
            for (int w = 4; w < 4 * WPS; ++w) c += (double)h[bI * 8 + w];
            lo += a / 4; hi += WPS == 2 ? c / 4 : 0;
        }
        lo /= blocks; hi /= blocks;
    }
    const double per = 6.0 * iters;
    // SIMD cost per MFMA-with-its-fillers: one wave: cycles / count; two symmetric waves: the slower wave's cycles / (2 x count)
    const double simd = WPS == 1 ? lo / per : (mode == 1 ? (lo > hi ? lo : hi) / per : (lo > hi ? lo : hi) / (2 * per));
    printf("%-22s NF %2d  waves/SIMD %d  mode %d : waves 0-3 %.1f cyc per gap, waves 4-7 %.1f  ->  SIMD pays %.1f cycles per (MFMA + %d fillers)\n",
           kname, NF, WPS, mode, lo / per, hi / per, simd, NF);
}

template <int NF, int KIND>
static void both(const char* kname, int blocks, float* out, unsigned long long* cyc)
{
    run<NF, KIND, 1>(kname, 0, blocks, out, cyc);
    run<NF, KIND, 2>(kname, 0, blocks, out, cyc);
    run<NF, KIND, 2>(kname, 2, blocks, out, cyc);
    run<NF, KIND, 2>(kname, 1, blocks, out, cyc);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&cyc, (size_t)blocks * 8 * 8);
    both<0, 0>("none", blocks, out, cyc);
    both<2, 0>("v_fma_f32", blocks, out, cyc);
    both<4, 0>("v_fma_f32", blocks, out, cyc);
    both<6, 0>("v_fma_f32", blocks, out, cyc);
    both<8, 0>("v_fma_f32", blocks, out, cyc);
    both<12, 0>("v_fma_f32", blocks, out, cyc);
    both<4, 1>("v_fma_mix lo/hi", blocks, out, cyc);
    both<8, 1>("v_fma_mix lo/hi", blocks, out, cyc);
    both<12, 1>("v_fma_mix lo/hi", blocks, out, cyc);
    both<5, 2>("kernel mix (1 ds_read)", blocks, out, cyc);
    both<8, 2>("kernel mix (1 ds_read)", blocks, out, cyc);
    both<10, 2>("kernel mix (1 ds_read)", blocks, out, cyc);
    both<4, 3>("v_pk_add_f32", blocks, out, cyc);
    return 0;
}
