// Calibrates the two in-kernel clocks used by tools/wf_clocks.py: s_memtime (__builtin_readcyclecounter / clock64) and
// s_memrealtime (wall_clock64) against hipEvent time.   hipcc --offload-arch=gfx950 -O3 tools/micro/clock_cal.hip -o build/clock_cal
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, unsigned long long ticks)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long r1 = r0;
    float a = 1.f;
    while (r1 - r0 < ticks) { for (int i = 0; i < 64; ++i) a = a * 1.0001f + 0.5f; r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
int main()
{
    int wc = 0, cr = 0;
    hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&cr, hipDeviceAttributeClockRate, 0);
    printf("hipDeviceAttributeWallClockRate %d kHz, ClockRate %d kHz\n", wc, cr);
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        spin<<<1, 64>>>(d, 1000000ull);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("event %.3f ms: s_memtime %llu (%.1f MHz)  s_memrealtime %llu (%.1f MHz)\n", ms, h[0], h[0] / ms / 1e3, h[1], h[1] / ms / 1e3);
    }
    return 0;
}
