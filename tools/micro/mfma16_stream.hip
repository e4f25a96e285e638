// tools/micro/mfma16_stream.hip -- development aid: issue-rate ceiling of the v_mfma_f32_16x16x4_f32 stream of csrc/winofused.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma16_stream.hip -o build/mfma16_stream && build/mfma16_stream
// 32 accumulators per wave (16 positions x 2 cout halves), groups of 8 MFMAs that alternate two accumulators (as the kernel's
// phases do), two blocks of four waves per CU (80 KB of dynamic LDS each).  Variants:
//   0  MFMAs only, operands constant                       1  + 4 VALU per group, results NOT consumed by the MFMAs
//   2  + 4 VALU per group feeding the next group's B operand (the kernel's dependency)
//   3  as 2 with 4 independent accumulators per group (ua/uc x 2 positions interleaved)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256, 2) void stream(float* out, int iters, float a0, float b0)
{
    extern __shared__ float lds[];
    f32x4 acc[16][2];
    for (int p = 0; p < 16; ++p) for (int c = 0; c < 2; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ua = {a0, a0 + 1, a0 + 2, a0 + 3}, uc = {a0, a0 - 1, a0 - 2, a0 - 3};
    f32x4 r0 = {b0, b0 * 2, b0 * 3, b0 * 4}, r1 = r0 * 0.5f + threadIdx.x * 1e-3f;
    f32x4 v = r0 - r1, side = r0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int p = 0; p < 16; p += (VAR == 3 ? 2 : 1)) {
            f32x4 vn = v;
            if (VAR == 1) side = side - r1;
            if (VAR >= 2) vn = (p & 1) ? v - r1 : v + r1;
            if (VAR == 3) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s], v[s], acc[p][0], 0, 0, 0);
                    acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[s], v[s], acc[p][1], 0, 0, 0);
                    acc[p + 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s], r1[s], acc[p + 1][0], 0, 0, 0);
                    acc[p + 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[s], r1[s], acc[p + 1][1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    acc[p][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ua[s], v[s], acc[p][0], 0, 0, 0);
                    acc[p][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(uc[s], v[s], acc[p][1], 0, 0, 0);
                }
            }
            v = vn;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f32x4 s4 = side + v;
    for (int p = 0; p < 16; ++p) for (int c = 0; c < 2; ++c) s4 += acc[p][c];
    out[blockIdx.x * 256 + threadIdx.x] = s4[0] + s4[1] + s4[2] + s4[3] + lds[threadIdx.x];
}

template <int VAR>
void run(int cus)
{
    float* out;
    hipMalloc(&out, (size_t)cus * 2 * 256 * 4);
    const int iters = 4000;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stream<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(stream<VAR>, dim3(cus * 2), dim3(256), 80 * 1024, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)cus * 2 * 4 * iters * 128 * 2048.0;
        if (rep == 3) printf("variant %d: %.3f ms  %.1f TFLOP/s (%.3f of 157.3)\n", VAR, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
    }
    hipFree(out);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s, %d CUs\n", p.gcnArchName, p.multiProcessorCount);
    run<0>(p.multiProcessorCount); run<0>(p.multiProcessorCount); run<0>(p.multiProcessorCount);
    run<0>(p.multiProcessorCount); run<1>(p.multiProcessorCount); run<2>(p.multiProcessorCount); run<3>(p.multiProcessorCount);
    return 0;
}
