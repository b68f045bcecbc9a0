// Workgroup dispatch rate on gfx950: how long does a grid of W workgroups of 256 threads take when each wave does nothing, or
// busy-waits a fixed time?  (Round 4: is the headline launch — 4,096 workgroups x 4 waves, 64 VGPRs, 2 KB of LDS — limited by how
// fast workgroups can be placed?)   hipcc --offload-arch=gfx950 -O3 scripts/ubench_dispatch.hip -o scripts/ubench_dispatch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int TPB>
__global__ __launch_bounds__(TPB, 8) void k_wait(unsigned long long ticks, float* sink)
{
    __shared__ float lds[512];                                 // 2 KB, as the planar kernel's records
    lds[threadIdx.x & 511] = (float)threadIdx.x;
    __syncthreads();
    float v = lds[(threadIdx.x * 7) & 511];
    // hold ~60 VGPRs alive so that the allocation matches the product kernel's
    float r[48];
#pragma unroll
    for (int i = 0; i < 48; ++i) r[i] = v * (float)(i + 1);
    const unsigned long long t0 = wall_clock64();              // 100 MHz
    while (wall_clock64() - t0 < ticks) { __builtin_amdgcn_s_sleep(8); }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 48; ++i) s += r[i];
    if (s == 12345.678f) sink[0] = s;
}
template <int TPB> static float run(int wgs, unsigned long long ticks, float* sink)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_wait<TPB>, dim3(wgs), dim3(TPB), 0, 0, ticks, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_wait<TPB>, dim3(wgs), dim3(TPB), 0, 0, ticks, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 20 * 1e3f;
}
int main()
{
    float* sink; hipMalloc(&sink, 4);
    printf("workgroups of 256 threads (4 waves), 2 KB LDS, ~60 VGPRs; us per launch, back to back\n");
    printf("%8s %12s %12s %12s %12s\n", "WGs", "no wait", "wait 10 us", "wait 30 us", "wait 45 us");
    for (int wgs : {256, 1024, 2048, 4096, 8192, 16384})
        printf("%8d %12.1f %12.1f %12.1f %12.1f\n", wgs, run<256>(wgs, 0, sink), run<256>(wgs, 1000, sink), run<256>(wgs, 3000, sink), run<256>(wgs, 4500, sink));
    printf("workgroups of 1024 threads (16 waves): the same number of WAVES per row as above\n");
    for (int wgs : {64, 256, 512, 1024, 2048, 4096})
        printf("%8d %12.1f %12.1f %12.1f %12.1f\n", wgs, run<1024>(wgs, 0, sink), run<1024>(wgs, 1000, sink), run<1024>(wgs, 3000, sink), run<1024>(wgs, 4500, sink));
    return 0;
}
