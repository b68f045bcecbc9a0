// Second issue-cost table (gfx950): the non-FMA instructions of the planar Fermat kernel — moves, selects, compares,
// min/max, bit ops, readlane vs LDS broadcast reads — at 8 waves per SIMD, 4 independent chains per wave.
// Cycles = ns per wave-instruction per SIMD x the in-kernel clock (s_memtime / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue2.hip -o scripts/ubench_issue2 && ./scripts/ubench_issue2
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
KERNEL(k_fmac32, asm volatile("v_fmac_f32 %0, %4, %5\n\tv_fmac_f32 %1, %4, %5\n\tv_fmac_f32 %2, %4, %5\n\tv_fmac_f32 %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_mul32, F4("v_mul_f32"))
KERNEL(k_add32, F4("v_add_f32"))
KERNEL(k_sub32, F4("v_sub_f32"))
KERNEL(k_max32, F4("v_max_f32"))
KERNEL(k_min32, F4("v_min_f32"))
KERNEL(k_max3, asm volatile("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %4, %5\n\tv_max3_f32 %2, %2, %4, %5\n\tv_max3_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_fma32_abs, asm volatile("v_fma_f32 %0, |%0|, %4, -%5\n\tv_fma_f32 %1, |%1|, %4, -%5\n\tv_fma_f32 %2, |%2|, %4, -%5\n\tv_fma_f32 %3, |%3|, %4, -%5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_fma32_sgpr, asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(seed), "v"(p)))
KERNEL(k_mov, asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m)))
KERNEL(k_and, I4("v_and_b32"))
KERNEL(k_addu32, I4("v_add_u32"))
KERNEL(k_lshl, asm volatile("v_lshlrev_b32 %0, 1, %0\n\tv_lshlrev_b32 %1, 1, %1\n\tv_lshlrev_b32 %2, 1, %2\n\tv_lshlrev_b32 %3, 1, %3" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id)))
KERNEL(k_and_or, asm volatile("v_and_or_b32 %0, %0, %4, %5\n\tv_and_or_b32 %1, %1, %4, %5\n\tv_and_or_b32 %2, %2, %4, %5\n\tv_and_or_b32 %3, %3, %4, %5" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia), "v"(ib)))
KERNEL(k_bfi, asm volatile("v_bfi_b32 %0, %4, %0, %5\n\tv_bfi_b32 %1, %4, %1, %5\n\tv_bfi_b32 %2, %4, %2, %5\n\tv_bfi_b32 %3, %4, %3, %5" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "v"(ia), "v"(ib)))
KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc"))
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32 %0, %0, %4, %5\n\tv_cndmask_b32 %1, %1, %4, %5\n\tv_cndmask_b32 %2, %2, %4, %5\n\tv_cndmask_b32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "s"(s01)))
KERNEL(k_cmp32_vcc, asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cmp_gt_f32 vcc, %1, %2\n\tv_cmp_gt_f32 vcc, %2, %3\n\tv_cmp_gt_f32 vcc, %3, %0" :: "v"(a), "v"(b), "v"(c), "v"(d) : "vcc"))
KERNEL(k_cmp32_cnd, asm volatile("v_cmp_gt_f32 vcc, %0, %4\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cmp_gt_f32 vcc, %2, %4\n\tv_cndmask_b32 %3, %3, %4, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc"))
KERNEL(k_cmp64_vcc, asm volatile("v_cmp_gt_f64 vcc, %0, %1\n\tv_cmp_gt_f64 vcc, %1, %2\n\tv_cmp_gt_f64 vcc, %2, %3\n\tv_cmp_gt_f64 vcc, %3, %0" :: "v"(x), "v"(y), "v"(z), "v"(w) : "vcc"))
KERNEL(k_readlane, { int s0, s1, s2, s3; asm volatile("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 5\n\tv_readlane_b32 %2, %6, 7\n\tv_readlane_b32 %3, %7, 9" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(ia), "v"(ib), "v"(ic), "v"(id)); asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3)); })
KERNEL(k_readfirstlane, { int s0, s1, s2, s3; asm volatile("v_readfirstlane_b32 %0, %4\n\tv_readfirstlane_b32 %1, %5\n\tv_readfirstlane_b32 %2, %6\n\tv_readfirstlane_b32 %3, %7" : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(ia), "v"(ib), "v"(ic), "v"(id)); asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3)); })
// LDS broadcast reads (all lanes the same address): LDS pipe, no VALU slot?  4 reads + one wait per group
KERNEL(k_ds_read_b32, asm volatile("ds_read_b32 %0, %4\n\tds_read_b32 %1, %4 offset:4\n\tds_read_b32 %2, %4 offset:8\n\tds_read_b32 %3, %4 offset:12\n\ts_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(ldsaddr) : "memory"))
KERNEL(k_ds_read_b128, { f4 q4; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q4) : "v"(ldsaddr) : "memory"); a = q4.x; b = q4.y; c = q4.z; d = q4.w; })
// the same LDS reads next to fp32 FMAs: does the LDS read steal VALU issue slots?
KERNEL(k_fma_plus_ds, { f4 q4; asm volatile("ds_read_b128 %0, %1" : "=v"(q4) : "v"(ldsaddr) : "memory");
        asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q4) :: "memory"); x += q4.x; })
KERNEL(k_fma64, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "v"(pd)))
// alternating fp32 FMA and fp64 FMA (does an fp32 op hide behind an fp64 one?)
KERNEL(k_mix_32_64, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f32 %2, %2, %6, %7\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f32 %3, %3, %6, %7" : "+v"(x), "+v"(y), "+v"(a), "+v"(b) : "v"(md), "v"(pd), "v"(m), "v"(p)))
// alternating transcendental and FMA (trans unit beside the main ALU?)
KERNEL(k_mix_rsq_fma, asm volatile("v_rsq_f32 %0, %0\n\tv_fma_f32 %2, %2, %4, %5\n\tv_rsq_f32 %1, %1\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_mix_rsq_3fma, asm volatile("v_rsq_f32 %0, %0\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p)))
KERNEL(k_mix_rsq_f64, asm volatile("v_rsq_f32 %0, %0\n\tv_fma_f64 %2, %2, %4, %5\n\tv_rsq_f32 %1, %1\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(b), "+v"(x), "+v"(y) : "v"(md), "v"(pd)))
KERNEL(k_salu, { unsigned s0 = 1, s1 = 2, s2 = 3, s3 = 4; asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)); asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3)); })
// SALU next to VALU: do scalar instructions take VALU issue slots of the same wave / SIMD?
KERNEL(k_fma_plus_salu, { unsigned s0 = 1, s1 = 2; asm volatile("v_fma_f32 %0, %0, %6, %7\n\ts_add_u32 %4, %4, 1\n\tv_fma_f32 %1, %1, %6, %7\n\ts_add_u32 %5, %5, 1\n\tv_fma_f32 %2, %2, %6, %7\n\ts_add_u32 %4, %4, 1\n\tv_fma_f32 %3, %3, %6, %7\n\ts_add_u32 %5, %5, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1) : "v"(m), "v"(p)); asm volatile("" :: "s"(s0), "s"(s1)); })

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
    run("v_fma_f32", k_fma32, w); run("v_fmac_f32", k_fmac32, w); run("v_mul_f32", k_mul32, w); run("v_add_f32", k_add32, w);
    run("v_sub_f32", k_sub32, w); run("v_fma_f32 |a|,-c", k_fma32_abs, w); run("v_fma_f32 sgpr", k_fma32_sgpr, w);
    run("v_max_f32", k_max32, w); run("v_min_f32", k_min32, w); run("v_max3_f32", k_max3, w);
    run("v_mov_b32", k_mov, w); run("v_and_b32", k_and, w); run("v_add_u32", k_addu32, w); run("v_lshlrev_b32", k_lshl, w);
    run("v_and_or_b32", k_and_or, w); run("v_bfi_b32", k_bfi, w);
    run("v_cndmask vcc", k_cndmask_vcc, w); run("v_cndmask sgpr", k_cndmask_sgpr, w);
    run("v_cmp_gt_f32 vcc", k_cmp32_vcc, w); run("v_cmp+v_cndmask", k_cmp32_cnd, w); run("v_cmp_gt_f64 vcc", k_cmp64_vcc, w);
    run("v_readlane_b32", k_readlane, w); run("v_readfirstlane_b32", k_readfirstlane, w);
    run("ds_read_b32 bcast", k_ds_read_b32, w); run("ds_read_b128 bcast", k_ds_read_b128, w, 1);
    run("4 fma + ds_read_b128", k_fma_plus_ds, w, 4);
    run("v_fma_f64", k_fma64, w); run("fma64,fma32 alternating", k_mix_32_64, w);
    run("rsq,fma alternating", k_mix_rsq_fma, w); run("rsq + 3 fma", k_mix_rsq_3fma, w); run("rsq,fma64 alternating", k_mix_rsq_f64, w);
    run("s_add_u32", k_salu, w); run("4 fma + 4 s_add", k_fma_plus_salu, w, 4);
    return 0;
}
