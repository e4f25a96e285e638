// tools/micro/stage_skeleton.hip -- development aid: skeleton of the conv stage loop (32 MFMAs + LDS staging + barrier per
// stage, no global memory), to find what keeps co-resident waves from filling each other's staging gaps.
//   variant bits: 1 = ds_reads feeding the MFMAs, 2 = ds_writes, 4 = s_barrier, 8 = global loads (L2 resident) feeding the writes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int V, int P>
__global__ __launch_bounds__(256) void skel(float* out, unsigned long long* cyc, int stages, const float* gsrc, int stagger)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < 8192; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 af[4], bf[4];
    for (int j = 0; j < 4; ++j) { af[j] = f32x4{1.f, 1.f, 1.f, 1.f}; bf[j] = f32x4{.5f, .5f, .5f, .5f}; }
    f32x4 g[6];
    for (int j = 0; j < 6; ++j) g[j] = f32x4{1.f, 2.f, 3.f, 4.f};
    if (stagger) {
        const int k = ((blockIdx.x >> 3) >> 5) % 3;      // which of the CU's three resident blocks this is (see mfma_contend.hip)
        for (int i = 0; i < k * stagger; ++i) __builtin_amdgcn_s_sleep(8);     // k * stagger * 512 cycles
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int s = 0; s < stages; ++s) {
        const int buf = (s & 1) * 4096;
        if (P) __builtin_amdgcn_s_setprio(3);
        if (V & 8) {
#pragma unroll
            for (int j = 0; j < 6; ++j) g[j] = *reinterpret_cast<const f32x4*>(gsrc + ((tid * 4 + j * 1024 + (s & 31) * 8192) & 0xFFFFC));
        }
        if (V & 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                af[j] = *reinterpret_cast<const f32x4*>(&lds[buf + ((lane * 20 + j * 1280) & 4092)]);
                bf[j] = *reinterpret_cast<const f32x4*>(&lds[buf + ((lane * 20 + j * 1280 + 640) & 4092)]);
            }
        }
        if (P) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2 * h + (i >> 1)][kk], bf[2 * h + (i & 1)][kk], acc[i], 0, 0, 0);
        if (P) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(3); __builtin_amdgcn_sched_barrier(0); }
        if (V & 2) {
#pragma unroll
            for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(&lds[(buf ^ 4096) + ((tid * 4 + j * 1024) & 4092)]) = g[j];
        }
        if (V & 4) __syncthreads();
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + tid] = sum;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int P = 0>
void run(int per_cu, int cus, const char* what, int stagger = 0)
{
    const int blocks = per_cu * cus, stages = 2000;
    float *out, *gsrc; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, blocks * 8); hipMalloc(&gsrc, 4 << 20); hipMemset(gsrc, 0, 4 << 20);
    const int lds_bytes = 53 * 1024;     // 3 blocks per CU at most, like the conv kernel
    hipFuncSetAttribute(reinterpret_cast<const void*>(skel<V, P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (skel<V, P>), 256, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((skel<V, P>), dim3(blocks), dim3(256), lds_bytes, 0, out, cyc, stages, gsrc, stagger);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = (double)blocks * 4 * stages * 32 * 4096.0 / ms / 1e9;
    unsigned long long* h = (unsigned long long*)malloc(blocks * 8);
    hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
    printf("%d blocks/CU (occupancy %d)  %-44s own cycles/stage %.0f   wall %.3f ms  %.1f TFLOP/s  = %.1f%% of 157.3\n", per_cu, occ, what,
           avg / stages, ms, tf, 100.0 * tf / 157.3);
    free(h); hipFree(out); hipFree(cyc); hipFree(gsrc);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run<0>(3, cus, "warm-up");
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        run<0>(per_cu, cus, "MFMA only");
        run<1>(per_cu, cus, "+ ds_reads");
        run<3>(per_cu, cus, "+ ds_reads + ds_writes");
        run<4>(per_cu, cus, "+ barrier only");
        run<7>(per_cu, cus, "+ ds_reads + ds_writes + barrier");
        run<15>(per_cu, cus, "+ ds_reads + ds_writes + barrier + global");
    }
    run<15, 1>(1, cus, "all, setprio 3 around staging");
    run<15, 1>(2, cus, "all, setprio 3 around staging");
    run<15, 1>(3, cus, "all, setprio 3 around staging");
    run<7, 1>(3, cus, "no global, setprio 3 around staging");
    run<15>(3, cus, "all, stagger 1 (k*512 cycles)", 1);
    run<15>(3, cus, "all, stagger 2 (k*1024 cycles)", 2);
    run<15>(3, cus, "all, stagger 4 (k*2048 cycles)", 4);
    run<15>(3, cus, "all, stagger 7 (k*3584 cycles)", 7);
    run<7>(3, cus, "no global, stagger 4", 4);
    return 0;
}
