// tools/micro/stage_pipelined.hip -- development aid: the same stage work as stage_skeleton.hip (32 MFMAs, 8 ds_read_b128,
// 6 ds_write_b128, 6 global loads, one barrier per stage) in a software-pipelined order where no LDS or memory latency is
// exposed inside a wave:
//   F0(s) already in registers | read F1(s) | write stage s+1 tile | load stage s+2 | 16 MFMA on F0 | barrier |
//   read F0(s+1) | 16 MFMA on F1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)

template <int V, int IL>
__global__ __launch_bounds__(256) void skel(float* out, int stages, const float* gsrc)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < 8192; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    f32x4 a0[2], b0[2], a1[2], b1[2];     // F0 = (a0, b0): fragments of the first half of the stage; F1 = (a1, b1)
    f32x4 g[6];
    const int ra = (lane * 20) & 4092, rb = (lane * 20 + 640) & 4092;
#pragma unroll
    for (int j = 0; j < 2; ++j) { a0[j] = *reinterpret_cast<const f32x4*>(&lds[ra + j * 1280]); b0[j] = *reinterpret_cast<const f32x4*>(&lds[rb + j * 1280]); }
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = *reinterpret_cast<const f32x4*>(gsrc + ((tid * 4 + j * 1024) & 0xFFFFC));
    for (int s = 0; s < stages; ++s) {
        const int cur = (s & 1) * 4096, nxt = cur ^ 4096;
        // b. F1(s)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            a1[j] = *reinterpret_cast<const f32x4*>(&lds[cur + ((ra + (j + 2) * 1280) & 4092)]);
            b1[j] = *reinterpret_cast<const f32x4*>(&lds[cur + ((rb + (j + 2) * 1280) & 4092)]);
        }
        if (!IL) SB();
        // c. stage s+1 tile -> LDS
#pragma unroll
        for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(&lds[nxt + ((tid * 4 + j * 1024) & 4092)]) = g[j];
        if (!IL) SB();
        // d. global loads for stage s+2
        if (V & 8) {
#pragma unroll
            for (int j = 0; j < 6; ++j) g[j] = *reinterpret_cast<const f32x4*>(gsrc + ((tid * 4 + j * 1024 + (s & 31) * 8192) & 0xFFFFC));
        }
        if (!IL) SB();
        // e. 16 MFMAs on F0
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i >> 1][kk], b0[i & 1][kk], acc[i], 0, 0, 0);
        if (IL) {
            // one staging instruction in the shadow of each MFMA: 4 ds_read, 6 ds_write, 6 global loads under 16 MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
#pragma unroll
            for (int q = 0; q < 6; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
            for (int q = 0; q < 6; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        SB();
        // f. barrier
        __syncthreads();
        SB();
        // g. F0(s+1)
#pragma unroll
        for (int j = 0; j < 2; ++j) { a0[j] = *reinterpret_cast<const f32x4*>(&lds[nxt + ra + j * 1280 - (ra + j * 1280 > 4092 ? 4096 : 0)]);
                                      b0[j] = *reinterpret_cast<const f32x4*>(&lds[nxt + ((rb + j * 1280) & 4092)]); }
        if (!IL) SB();
        // h. 16 MFMAs on F1
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i >> 1][kk], b1[i & 1][kk], acc[i], 0, 0, 0);
        if (IL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        }
        SB();
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + tid] = sum + g[0][0];
}

template <int V, int IL>
void run(int per_cu, int cus, const char* what)
{
    const int blocks = per_cu * cus, stages = 2000;
    float *out, *gsrc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&gsrc, 4 << 20); hipMemset(gsrc, 0, 4 << 20);
    const int lds_bytes = 53 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(skel<V, IL>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (skel<V, IL>), 256, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((skel<V, IL>), dim3(blocks), dim3(256), lds_bytes, 0, out, stages, gsrc);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = (double)blocks * 4 * stages * 32 * 4096.0 / ms / 1e9;
    printf("%d blocks/CU (occupancy %d)  %-30s wall %.3f ms  %.1f TFLOP/s  = %.1f%% of 157.3\n", per_cu, occ, what, ms, tf, 100.0 * tf / 157.3);
    hipFree(out); hipFree(gsrc);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run<8, 0>(3, cus, "warm-up");
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        run<0, 0>(per_cu, cus, "grouped, no global"); run<8, 0>(per_cu, cus, "grouped, with global");
        run<0, 1>(per_cu, cus, "interleaved, no global"); run<8, 1>(per_cu, cus, "interleaved, with global");
    }
    return 0;
}
