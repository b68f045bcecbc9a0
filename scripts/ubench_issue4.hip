// Fourth issue-cost table (gfx950): the forms of an fp64 Horner step with a 64-bit constant addend, per step (group of 4).
// Cycles = ns per wave-instruction per SIMD x the in-kernel clock (s_memtime / s_memrealtime).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue4.hip -o scripts/ubench_issue4 && ./scripts/ubench_issue4
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



// Horner step forms for a 64-bit constant addend C: (a) all-VGPR v_fma_f64, (b) C in an SGPR pair loaded once, (c) C moved into a
// fresh SGPR pair by s_mov right in front of each v_fma_f64 (what RTUS_FMA_C compiles to), (d) the compiler's default: two
// v_mov_b32 of the literal halves into the destination + v_fmac_f64.  Each group = 4 independent Horner steps.
KERNEL(k_fma64_v, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "v"(pd)))
KERNEL(k_fma64_s, asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "s"(pd)))
KERNEL(k_fma64_smov, asm volatile(
    "s_mov_b32 s40, 0x55555555\n\ts_mov_b32 s41, 0x3fc55555\n\tv_fma_f64 %0, %0, %4, s[40:41]\n\t"
    "s_mov_b32 s42, 0x11111111\n\ts_mov_b32 s43, 0x3f811111\n\tv_fma_f64 %1, %1, %4, s[42:43]\n\t"
    "s_mov_b32 s40, 0x1a01a01a\n\ts_mov_b32 s41, 0x3f2a01a0\n\tv_fma_f64 %2, %2, %4, s[40:41]\n\t"
    "s_mov_b32 s42, 0x5555aaaa\n\ts_mov_b32 s43, 0x3ec71de3\n\tv_fma_f64 %3, %3, %4, s[42:43]"
    : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md) : "s40", "s41", "s42", "s43"))
// the accumulators live in v[40:47] across iterations here (the fmac destination is the freshly moved constant, so the chain
// alternates between two register pairs per stream, as the compiler's code does)
KERNEL(k_fmac64_vmov, asm volatile(
    "v_mov_b32 v40, 0x55555555\n\tv_mov_b32 v41, 0x3fc55555\n\tv_fmac_f64 v[40:41], %0, %4\n\t"
    "v_mov_b32 v42, 0x11111111\n\tv_mov_b32 v43, 0x3f811111\n\tv_fmac_f64 v[42:43], %1, %4\n\t"
    "v_mov_b32 v44, 0x1a01a01a\n\tv_mov_b32 v45, 0x3f2a01a0\n\tv_fmac_f64 v[44:45], %2, %4\n\t"
    "v_mov_b32 v46, 0x5555aaaa\n\tv_mov_b32 v47, 0x3ec71de3\n\tv_fmac_f64 v[46:47], %3, %4\n\t"
    "v_mov_b32 v48, 0x55555555\n\tv_mov_b32 v49, 0x3fc55555\n\tv_fmac_f64 v[48:49], v[40:41], %4\n\t"
    "v_mov_b32 v50, 0x11111111\n\tv_mov_b32 v51, 0x3f811111\n\tv_fmac_f64 v[50:51], v[42:43], %4\n\t"
    "v_mov_b32 v52, 0x1a01a01a\n\tv_mov_b32 v53, 0x3f2a01a0\n\tv_fmac_f64 v[52:53], v[44:45], %4\n\t"
    "v_mov_b32 v54, 0x5555aaaa\n\tv_mov_b32 v55, 0x3ec71de3\n\tv_fmac_f64 v[54:55], v[46:47], %4"
    : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md) : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55"))

// Is the scalar ALU free beside a VALU-bound stream?  (e) SALU alone, (f) 4 v_fma_f64 + 4 independent s_add_u32, (g) 4 v_fma_f64 + 8,
// (h) 4 v_fma_f32 + 4 s_add_u32, (i) constants s_mov'ed four instructions AHEAD of the v_fma_f64 that reads them.
KERNEL(k_salu, asm volatile("s_add_u32 s40, s40, 1\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s43, s43, 1" ::: "s40", "s41", "s42", "s43", "scc"))
KERNEL(k_fma64_salu4, asm volatile("v_fma_f64 %0, %0, %4, %5\n\ts_add_u32 s40, s40, 1\n\tv_fma_f64 %1, %1, %4, %5\n\ts_add_u32 s41, s41, 1\n\tv_fma_f64 %2, %2, %4, %5\n\ts_add_u32 s42, s42, 1\n\tv_fma_f64 %3, %3, %4, %5\n\ts_add_u32 s43, s43, 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "v"(pd) : "s40", "s41", "s42", "s43", "scc"))
KERNEL(k_fma64_salu8, asm volatile("v_fma_f64 %0, %0, %4, %5\n\ts_add_u32 s40, s40, 1\n\ts_add_u32 s44, s44, 1\n\tv_fma_f64 %1, %1, %4, %5\n\ts_add_u32 s41, s41, 1\n\ts_add_u32 s45, s45, 1\n\tv_fma_f64 %2, %2, %4, %5\n\ts_add_u32 s42, s42, 1\n\ts_add_u32 s46, s46, 1\n\tv_fma_f64 %3, %3, %4, %5\n\ts_add_u32 s43, s43, 1\n\ts_add_u32 s47, s47, 1" : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "v"(pd) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc"))
KERNEL(k_fma32_salu4, asm volatile("v_fma_f32 %0, %0, %4, %5\n\ts_add_u32 s40, s40, 1\n\tv_fma_f32 %1, %1, %4, %5\n\ts_add_u32 s41, s41, 1\n\tv_fma_f32 %2, %2, %4, %5\n\ts_add_u32 s42, s42, 1\n\tv_fma_f32 %3, %3, %4, %5\n\ts_add_u32 s43, s43, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p) : "s40", "s41", "s42", "s43", "scc"))
KERNEL(k_fma64_smov_ahead, asm volatile(
    "s_mov_b32 s40, 0x55555555\n\ts_mov_b32 s41, 0x3fc55555\n\ts_mov_b32 s42, 0x11111111\n\ts_mov_b32 s43, 0x3f811111\n\t"
    "s_mov_b32 s44, 0x1a01a01a\n\ts_mov_b32 s45, 0x3f2a01a0\n\ts_mov_b32 s46, 0x5555aaaa\n\ts_mov_b32 s47, 0x3ec71de3\n\t"
    "v_fma_f64 %0, %0, %4, s[40:41]\n\tv_fma_f64 %1, %1, %4, s[42:43]\n\tv_fma_f64 %2, %2, %4, s[44:45]\n\tv_fma_f64 %3, %3, %4, s[46:47]"
    : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md) : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47"))

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
    for (int w = 6; w <= 8; w += 2) {
        run("v_fma_f64 vgpr C", k_fma64_v, w); run("v_fma_f64 sgpr C", k_fma64_s, w);
        run("s_add_u32 alone", k_salu, w); run("4 v_fma_f64 + 4 s_add", k_fma64_salu4, w); run("4 v_fma_f64 + 8 s_add", k_fma64_salu8, w);
        run("4 v_fma_f32 + 4 s_add", k_fma32_salu4, w); run("8 s_mov ahead + 4 v_fma_f64 s", k_fma64_smov_ahead, w);
        run("2 s_mov + v_fma_f64 s", k_fma64_smov, w); run("2 v_mov + v_fmac_f64", k_fmac64_vmov, w, 8);
    }
    return 0;
}
