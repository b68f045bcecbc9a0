// Third issue-cost table (gfx950): what makes an fp32 VALU op lose its double rate (SGPR / inline-constant / literal operands)
// and when v_cndmask_b32 reading VCC turns slow — at 8 waves per SIMD, 4 independent chains per wave.
// Cycles = ns per wave-instruction per SIMD x the in-kernel clock (s_memtime / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue3.hip -o scripts/ubench_issue3 && ./scripts/ubench_issue3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

#define KERNEL(NAME, ...)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(double* out, unsigned long long* clk, float seed, int iters)       \
    {                                                                                                              \
        __shared__ float lds[1024];                                                                                \
        lds[threadIdx.x] = seed + threadIdx.x; lds[threadIdx.x + 256] = seed; lds[threadIdx.x + 512] = seed; lds[threadIdx.x + 768] = seed; \
        __syncthreads();                                                                                           \
        float a = seed + threadIdx.x * 1e-3f, b = a + 1, c = a + 2, d = a + 3;                                     \
        const float m = 1.0000001f, p = 0.999f;                                                                    \
        double x = a, y = b, z = c, w = d;                                                                         \
        const double md = 1.0000001, pd = 0.999;                                                                   \
        int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;                                               \
        unsigned long long s01 = 0x5555555555555555ull;                                                            \
        unsigned ldsaddr = (threadIdx.x >> 6) * 64;                                                                \
        (void)x; (void)y; (void)z; (void)w; (void)md; (void)pd; (void)ia; (void)ib; (void)ic; (void)id; (void)s01; (void)ldsaddr; (void)m; (void)p; \
        unsigned long long t0, t1, r0, r1;                                                                         \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory"); \
        for (int i = 0; i < iters; ++i) {                                                                          \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) { __VA_ARGS__; }                                                \
        }                                                                                                          \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory"); \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w + ia + ib + ic + id;            \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }                \
    }

#define F4(op) asm volatile(op " %0, %0, %4\n\t" op " %1, %1, %4\n\t" op " %2, %2, %4\n\t" op " %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m))
#define I4(op) asm volatile(op " %0, %0, %4\n\t" op " %1, %1, %4\n\t" op " %2, %2, %4\n\t" op " %3, %3, %4" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia))


KERNEL(k_fma32, asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_fma32_inl, asm volatile("v_fma_f32 %0, %0, %4, 1.0\n\tv_fma_f32 %1, %1, %4, 1.0\n\tv_fma_f32 %2, %2, %4, 1.0\n\tv_fma_f32 %3, %3, %4, 1.0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m)))
KERNEL(k_mul32_inl, asm volatile("v_mul_f32 %0, 0.5, %0\n\tv_mul_f32 %1, 2.0, %1\n\tv_mul_f32 %2, 0.5, %2\n\tv_mul_f32 %3, 2.0, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)))
KERNEL(k_mul32_sgpr, asm volatile("v_mul_f32 %0, %4, %0\n\tv_mul_f32 %1, %4, %1\n\tv_mul_f32 %2, %4, %2\n\tv_mul_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed)))
KERNEL(k_fmaak, asm volatile("v_fmaak_f32 %0, %0, %4, 0x3f7fbe77\n\tv_fmaak_f32 %1, %1, %4, 0x3f7fbe77\n\tv_fmaak_f32 %2, %2, %4, 0x3f7fbe77\n\tv_fmaak_f32 %3, %3, %4, 0x3f7fbe77" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m)))
KERNEL(k_add32_neg, asm volatile("v_sub_f32 %0, %4, %0\n\tv_sub_f32 %1, %4, %1\n\tv_sub_f32 %2, %4, %2\n\tv_sub_f32 %3, %4, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m)))
KERNEL(k_mov_sgpr, asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "s"(seed)))
KERNEL(k_mov_inl, asm volatile("v_mov_b32 %0, 1.0\n\tv_mov_b32 %1, 2.0\n\tv_mov_b32 %2, 0.5\n\tv_mov_b32 %3, 4.0" : "=v"(a), "=v"(b), "=v"(c), "=v"(d)))
KERNEL(k_add64_sgpr, asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "s"(pd)))
KERNEL(k_cvt_abs, asm volatile("v_cvt_f32_f64 %0, |%4|\n\tv_cvt_f32_f64 %1, |%5|\n\tv_cvt_f32_f64 %2, |%6|\n\tv_cvt_f32_f64 %3, |%7|" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(x), "v"(y), "v"(z), "v"(w)))
// v_cndmask_b32 reading VCC: four in a row on a VCC nobody writes / VOP3 encoding / VCC written by SALU once per group /
// VCC written by one v_cmp per group / one v_cmp per cndmask / SGPR-pair mask written by v_cmp_e64
KERNEL(k_cnd_vcc4, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc"))
KERNEL(k_cnd_vcc4_e64, asm volatile("v_cndmask_b32_e64 %0, %0, %4, vcc\n\tv_cndmask_b32_e64 %1, %1, %4, vcc\n\tv_cndmask_b32_e64 %2, %2, %4, vcc\n\tv_cndmask_b32_e64 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc"))
KERNEL(k_cnd_salu_vcc4, asm volatile("s_mov_b64 vcc, %5\n\ts_nop 3\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "s"(s01) : "vcc"))
KERNEL(k_cnd_cmp_vcc4, asm volatile("v_cmp_gt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc"))
KERNEL(k_cnd_cmp_sgpr4, { unsigned long long mk; asm volatile("v_cmp_gt_f32_e64 %4, %0, %5\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %5, %4\n\tv_cndmask_b32_e64 %1, %1, %5, %4\n\tv_cndmask_b32_e64 %2, %2, %5, %4\n\tv_cndmask_b32_e64 %3, %3, %5, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(mk) : "v"(m)); })
KERNEL(k_cnd_vcc1_fma3, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p) : "vcc"))
KERNEL(k_cnd_sgpr1_fma3, asm volatile("v_cndmask_b32_e64 %0, %0, %4, %6\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p), "s"(s01)))
// compiler-generated selects on one per-lane condition (what the layer set-up of the planar kernel is made of)
KERNEL(k_select_cc, { const bool sw = a > b; x = sw ? y : x; z = sw ? w : z; y = sw ? x : y; w = sw ? z : w; a += 1e-9f; })

typedef void (*kern_t)(double*, unsigned long long*, float, int);


static void run(const char* name, kern_t k, int waves_per_simd, int per_group = 4)
{
    const int blocks = 256 * waves_per_simd;
    double* out; unsigned long long* clk;
    (void)hipMalloc(&out, sizeof(double) * blocks * 256);
    (void)hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    const int iters = 3000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int warm = 0; warm < 3; ++warm) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 1.5f, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, 1.5f, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks);
    for (int i = 0; i < blocks; ++i) ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
    std::sort(ghz.begin(), ghz.end());
    const double winst = (double)waves_per_simd * iters * 8 * per_group;
    const double ns_per = ms * 1e6 / winst;
    printf("%-22s waves/SIMD=%d  %8.3f ms  %6.2f ns/inst/SIMD  clock %.2f GHz  => %6.2f cycles per instruction (group of %d)\n",
           name, waves_per_simd, ms, ns_per, ghz[blocks / 2], ns_per * ghz[blocks / 2], per_group);
    (void)hipFree(out); (void)hipFree(clk);
}


int main()
{
    const int w = 8;
    run("v_fma_f32 vgpr", k_fma32, w); run("v_fma_f32 inline 1.0", k_fma32_inl, w); run("v_mul_f32 inline 0.5", k_mul32_inl, w);
    run("v_mul_f32 sgpr", k_mul32_sgpr, w); run("v_fmaak_f32 literal", k_fmaak, w); run("v_sub_f32 vgpr", k_add32_neg, w);
    run("v_mov_b32 sgpr", k_mov_sgpr, w); run("v_mov_b32 inline", k_mov_inl, w); run("v_add_f64 sgpr", k_add64_sgpr, w);
    run("v_cvt_f32_f64 |x|", k_cvt_abs, w);
    run("cndmask vcc x4", k_cnd_vcc4, w); run("cndmask_e64 vcc x4", k_cnd_vcc4_e64, w);
    run("s_mov vcc + cndmask x4", k_cnd_salu_vcc4, w); run("v_cmp vcc + cndmask x4", k_cnd_cmp_vcc4, w, 5);
    run("v_cmp_e64 s + cndmask x4", k_cnd_cmp_sgpr4, w, 5);
    run("cndmask vcc + 3 fma", k_cnd_vcc1_fma3, w); run("cndmask sgpr + 3 fma", k_cnd_sgpr1_fma3, w);
    run("4 f64 selects (compiler)", k_select_cc, w, 4);
    return 0;
}
