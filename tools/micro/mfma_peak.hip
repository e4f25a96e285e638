// tools/micro/mfma_peak.hip -- development aid: issue-rate ceiling of v_mfma_f32_32x32x2_f32 on this device.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak && tools/micro/mfma_peak
// Each wave runs a register-only loop of independent MFMAs (NACC accumulators); grids of 1, 2, 3 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int i = 0; i < NACC; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
void run(int blocks_per_cu, int cus)
{
    float* out;
    hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * blocks_per_cu * 4 * iters * 8 * NACC * 4096.0;
        if (rep == 2) printf("NACC=%d waves/SIMD=%d  %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, ms, flops / ms / 1e9);
    }
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    // warm the clocks
    run<4>(3, p.multiProcessorCount); run<4>(3, p.multiProcessorCount);
    for (int w = 1; w <= 3; ++w) { run<1>(w, p.multiProcessorCount); run<2>(w, p.multiProcessorCount); run<4>(w, p.multiProcessorCount); }
    return 0;
}
