// tools/micro/gemm_stage_glds.hip -- development aid for round 2: the stage loop of the batched Winograd GEMM
// (csrc/linear.hip, 128 x 128 tile: per 16-k stage and wave 32 MFMAs, 8 ds_read_b128 of fragments, one barrier) with the
// operand tile of the next stage brought in two ways:
//   R: registers  -- 4 global_load_dwordx4 + 4 ds_write_b128 per thread and stage (what the library does today)
//   D: direct     -- 4 global_load_lds_dwordx4 per wave and stage into a lane-linear, unpadded LDS tile; fragment reads use the
//                    XOR swizzle chunk' = chunk ^ ((row >> 2) & 3), applied to the per-lane SOURCE address of the load
// Both in the software-pipelined order of the library kernel.  No result check: the point is the MFMA rate.
//   hipcc --offload-arch=gfx950 -O3 gemm_stage_glds.hip -o gemm_stage_glds && ./gemm_stage_glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SB() __builtin_amdgcn_sched_barrier(0)
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// LDS per buffer: A tile 128 rows + B tile 128 rows of 16 floats.  R: rows padded to 20 floats; D: 16 floats, swizzled.
template <int DIRECT>
__global__ __launch_bounds__(256) void stage_loop(float* out, int stages, const float* gsrc, int ld, const float* big, size_t big_floats)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int ROWF = DIRECT ? 16 : 20;
    constexpr int BUF = 256 * ROWF;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < 2 * BUF; i += 256) lds[i] = 1.0f + i * 1e-6f;
    __syncthreads();
    // fragment offsets (floats) of half h for the A rows 64 wm + 32 i + li and the B rows 128 + 64 wn + 32 j + li
    int fa[2], fb[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int chunk = lh + 2 * h;
        const int ra = 64 * wm + li, rb = 128 + 64 * wn + li;
        fa[h] = ra * ROWF + 4 * (DIRECT ? (chunk ^ ((ra >> 2) & 3)) : chunk);
        fb[h] = rb * ROWF + 4 * (DIRECT ? (chunk ^ ((rb >> 2) & 3)) : chunk);
    }
    // staging: 1024 16-byte pieces per stage (256 rows x 4 chunks)
    //   R: thread t, piece q = t + 256 it: row q >> 2, chunk q & 3
    //   D: wave w, instruction it: rows 64 w + 16 it + (lane >> 2), physical chunk lane & 3 <- logical chunk (lane & 3) ^ ((row >> 2) & 3)
    unsigned src[4]; int dst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        if (DIRECT) {
            const int row = 64 * wave + 16 * it + (lane >> 2), pc = lane & 3, lc = pc ^ ((row >> 2) & 3);
            src[it] = (unsigned)(((size_t)row * ld + 4 * lc) * sizeof(float));
            dst[it] = (64 * wave + 16 * it) * 16;                     // wave-uniform base of the 1 KB the instruction fills
        } else {
            const int q = tid + 256 * it, row = q >> 2, pc = q & 3;
            src[it] = (unsigned)(((size_t)row * ld + 4 * pc) * sizeof(float));
            dst[it] = row * 20 + 4 * pc;
        }
    }
    f32x4 g[4];
    f32x4 a0[2], b0[2], a1[2], b1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { a0[j] = *reinterpret_cast<const f32x4*>(&lds[fa[0] + j * 32 * ROWF]); b0[j] = *reinterpret_cast<const f32x4*>(&lds[fb[0] + j * 32 * ROWF]); }
    if (!DIRECT) {
#pragma unroll
        for (int it = 0; it < 4; ++it) g[it] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(gsrc) + src[it]);
    }
    for (int s = 0; s < stages; ++s) {
        const int cur = (s & 1) * BUF, nxt = cur ^ BUF;
        const char* gs = reinterpret_cast<const char*>(gsrc + ((s + 2) & 31) * 16);
        // streaming variant (register path only): the A half of the tile (threads whose pieces are rows 0..127, it < 2) comes from a
        // window of a buffer far larger than L2 + Infinity Cache that moves with block and stage, like V does in the real GEMM
        const char* ga = big ? reinterpret_cast<const char*>(big + (((size_t)blockIdx.x * 128 * ld + (size_t)(s + 2) * 16) % (big_floats - (size_t)129 * ld)))
                             : gs;
        // second-half fragments of this stage
#pragma unroll
        for (int j = 0; j < 2; ++j) { a1[j] = *reinterpret_cast<const f32x4*>(&lds[cur + fa[1] + j * 32 * ROWF]); b1[j] = *reinterpret_cast<const f32x4*>(&lds[cur + fb[1] + j * 32 * ROWF]); }
        if (!DIRECT) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(&lds[nxt + dst[it]]) = g[it];          // tile s+1 -> LDS
#pragma unroll
            for (int it = 0; it < 4; ++it) g[it] = *reinterpret_cast<const f32x4*>((it < 2 ? ga : gs) + src[it]);   // tile s+2 -> registers
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[i >> 1][kk], b0[i & 1][kk], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        if (!DIRECT) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x200, 1, 0); }
#pragma unroll
            for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x020, 1, 0); }
        }
        SB();
        if (DIRECT) {
            // tile s+1 (issued one stage ago) must have landed, and this wave's reads of `cur` must be complete, before the
            // barrier publishes nxt and frees cur
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            __syncthreads();
        }
        SB();
        if (DIRECT) {
            // tile s+2 straight into the buffer that was just freed
#pragma unroll
            for (int it = 0; it < 4; ++it)
                __builtin_amdgcn_global_load_lds(GLB_PTR(gs + src[it]), LDS_PTR(&lds[cur + dst[it]]), 16, 0, 0);
        }
        // first-half fragments of stage s+1
#pragma unroll
        for (int j = 0; j < 2; ++j) { a0[j] = *reinterpret_cast<const f32x4*>(&lds[nxt + fa[0] + j * 32 * ROWF]); b0[j] = *reinterpret_cast<const f32x4*>(&lds[nxt + fb[0] + j * 32 * ROWF]); }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[i >> 1][kk], b1[i & 1][kk], acc[i], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { __builtin_amdgcn_sched_group_barrier(0x008, 2, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); }
        SB();
    }
    float sum = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[blockIdx.x * 256 + tid] = sum + (DIRECT ? 0.f : g[0][0]);
}

// random_data: the operand tiles stream from a buffer of uniform random floats in (-1, 1) instead of zeros -- the MFMA operands
// then toggle like real activations / weights do (data-dependent power), everything else is identical.
template <int DIRECT>
// mode 0: one long block per resident slot (20000 stages); 1: the real GEMM's shape -- 32 x as many blocks of 32 stages (K = 512 per
// tile row, as conv4_x), hot operands; 2: the same with the A half of every tile streaming from a 640 MB buffer
void run(int per_cu, int cus, const char* what, bool random_data = false, int mode = 0)
{
    const bool streaming = mode == 2;
    const int mult = mode ? 32 : 1;
    const int blocks = per_cu * cus, stages = mode ? 32 : 20000, ld = 512;
    float* big = nullptr;
    const size_t big_floats = (size_t)160 << 20;                                        // 640 MB
    if (streaming) { hipMalloc(&big, big_floats * 4); hipMemset(big, 0, big_floats * 4); }
    float *out, *gsrc;
    hipMalloc(&out, (size_t)blocks * mult * 256 * 4); hipMalloc(&gsrc, 4 << 20); hipMemset(gsrc, 0, 4 << 20);
    if (random_data) {
        float* h = (float*)malloc(4 << 20);
        unsigned x = 12345u;
        for (int i = 0; i < (1 << 20); ++i) { x = x * 1664525u + 1013904223u; h[i] = ((int)(x >> 8) - (1 << 23)) / (float)(1 << 23); }
        hipMemcpy(gsrc, h, 4 << 20, hipMemcpyHostToDevice);
        free(h);
    }
    // 3 blocks per CU at most, like the library kernel (136 registers)
    const int lds_bytes = 53 * 1024;
    hipFuncSetAttribute(reinterpret_cast<const void*>(stage_loop<DIRECT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (stage_loop<DIRECT>), 256, lds_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((stage_loop<DIRECT>), dim3(blocks * mult), dim3(256), lds_bytes, 0, out, stages, gsrc, ld,
                           (const float*)big, big_floats);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    const double tf = (double)blocks * mult * 4 * stages * 32 * 4096.0 / ms / 1e9;
    if (big) hipFree(big);
    printf("%d blocks/CU (occupancy %d)  %-34s wall %.3f ms  %.1f TFLOP/s  = %.1f%% of 157.3\n", per_cu, occ, what, ms, tf, 100.0 * tf / 157.3);
    hipFree(out); hipFree(gsrc);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    run<0>(3, cus, "warm-up");
    for (int per_cu = 1; per_cu <= 3; ++per_cu) {
        run<0>(per_cu, cus, "register staging (load + ds_write)");
        run<1>(per_cu, cus, "direct global_load_lds");
    }
    run<0>(3, cus, "register staging, RANDOM operands", true);
    run<1>(3, cus, "direct global_load_lds, RANDOM operands", true);
    run<0>(3, cus, "register staging (zeros again)");
    run<0>(3, cus, "short blocks (32 stages), hot operands", false, 1);
    run<0>(3, cus, "short blocks, A streams from HBM", false, 2);
    return 0;
}
