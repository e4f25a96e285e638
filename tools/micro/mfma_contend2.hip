// tools/micro/mfma_contend2.hip -- development aid: one 512-thread block per CU; waves 0-3 run a register-only MFMA loop,
// waves 4-7 (which share the four SIMDs with them) run a "noise" loop.  Reports cycles per MFMA of the MFMA waves.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, unsigned* hwid, int iters, int kind, const float* gsrc)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    __shared__ volatile int done;
    const int tid = threadIdx.x, wave = tid >> 6;
    if (tid == 0) done = 0;
    __syncthreads();
    if ((tid & 63) == 0) hwid[blockIdx.x * 8 + wave] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float a = 1.0f + tid * 1e-3f, b = 0.5f;
        const unsigned long long t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        const unsigned long long t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + tid] = s;
        if ((tid & 63) == 0) { cyc[blockIdx.x * 4 + wave] = t1 - t0; done = 1; }
    } else {
        f32x4 v = {1.f, 2.f, 3.f, 4.f}, w = {0.f, 0.f, 0.f, 0.f};
        long n = 0;
        if (kind != 0)
            while (!done) {
                ++n;
                if (kind == 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { f32x4 r = *reinterpret_cast<volatile f32x4*>(&lds[(tid * 4 + j * 2048) & 16380]); w += r; }
                } else if (kind == 2) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) *reinterpret_cast<volatile f32x4*>(&lds[(tid * 4 + j * 2048) & 16380]) = v;
                } else if (kind == 3) {
#pragma unroll
                    for (int j = 0; j < 32; ++j) { v[0] = v[0] * 1.0001f + v[1]; v[1] = v[1] * 0.9999f + v[2]; }
                } else if (kind == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { f32x4 r = *reinterpret_cast<const volatile f32x4*>(gsrc + ((tid * 4 + (n & 63) * 2048 + j * 131072) & 0xFFFFC)); w += r; }
                }
            }
        out[blockIdx.x * 512 + tid] = w[0] + w[1] + w[2] + w[3] + v[0] + v[1] + (float)n;
        if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (wave - 4) + 4096] = (unsigned long long)n;
    }
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount;
    float *out, *gsrc; unsigned long long* cyc; unsigned* hwid;
    hipMalloc(&out, (size_t)blocks * 512 * 4); hipMalloc(&cyc, 8192 * 8); hipMalloc(&gsrc, 4 << 20); hipMalloc(&hwid, blocks * 8 * 4);
    hipMemset(gsrc, 0, 4 << 20);
    const int iters = 3000;
    const char* names[] = {"none", "LDS reads b128", "LDS writes b128", "VALU fma", "global loads b128"};
    for (int kind = 0; kind < 5; ++kind)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(cyc, 0, 8192 * 8);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, hwid, iters, kind, gsrc);
            hipDeviceSynchronize();
            static unsigned long long h[8192];
            hipMemcpy(h, cyc, 8192 * 8, hipMemcpyDeviceToHost);
            double avg = 0, nn = 0;
            for (int i = 0; i < blocks * 4; ++i) { avg += (double)h[i]; nn += (double)h[4096 + i]; }
            avg /= blocks * 4; nn /= blocks * 4;
            if (rep == 1) printf("noise=%-18s cycles/MFMA = %.2f   (noise iterations per wave while the MFMA waves ran: %.0f)\n", names[kind],
                                 avg / (iters * 32.0), nn);
        }
    static unsigned hh[2048];
    hipMemcpy(hh, hwid, blocks * 8 * 4, hipMemcpyDeviceToHost);
    printf("HW_ID of block 0 waves 0..7:"); for (int w = 0; w < 8; ++w) printf(" %08x", hh[w]); printf("\n");
    printf("HW_ID of block 9 waves 0..7:"); for (int w = 0; w < 8; ++w) printf(" %08x", hh[72 + w]); printf("\n");
    return 0;
}
