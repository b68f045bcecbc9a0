// Issue cost (shader cycles per wave-instruction per SIMD) of the instructions the planar Fermat kernel is made of,
// with the in-kernel clock read beside it (s_memtime / s_memrealtime), at 8 and 4 waves per SIMD.
// Decides whether packed fp32 (v_pk_fma_f32, two focal points per lane) can cut issue slots on gfx950.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_issue.hip -o scripts/ubench_issue && ./scripts/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef float f2 __attribute__((ext_vector_type(2)));

enum { FMA32, PKFMA32, PKMUL32, PKADD32, RSQ32, RCP32, CVT_F64_F32, CVT_F32_F64, FMA64, FMA64_SGPR, MUL64, ADD64, READLANE,
       CNDMASK32, MAX32, FMA64_CHAIN1, LDEXP64, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, unsigned long long* clk, float seed, int iters, double sarg)
{
    float a = seed + threadIdx.x * 1e-3f, b = a + 1, c = a + 2, d = a + 3;
    const float m = 1.0000001f, p = 0.999f;
    f2 A = {a, b}, B = {c, d}, C = {a + 4, b + 4}, D = {c + 4, d + 4};
    const f2 M = {m, m}, P = {p, p};
    double x = a, y = b, z = c, w = d;
    const double md = 1.0000001, pd = 0.999;
    int ia = threadIdx.x, ib = ia + 1, ic = ia + 2, id = ia + 3;
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0) :: "memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m), "v"(p));
            if (OP == PKFMA32) asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                                            : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(M), "v"(P));
            if (OP == PKMUL32) asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
                                            : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(M));
            if (OP == PKADD32) asm volatile("v_pk_add_f32 %0, %0, %4\n\tv_pk_add_f32 %1, %1, %4\n\tv_pk_add_f32 %2, %2, %4\n\tv_pk_add_f32 %3, %3, %4"
                                            : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(P));
            if (OP == RSQ32) asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %1, %1\n\tv_rsq_f32 %2, %2\n\tv_rsq_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == RCP32) asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (OP == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7"
                                                : "=v"(x), "=v"(y), "=v"(z), "=v"(w) : "v"(a), "v"(b), "v"(c), "v"(d));
            if (OP == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %4\n\tv_cvt_f32_f64 %1, %5\n\tv_cvt_f32_f64 %2, %6\n\tv_cvt_f32_f64 %3, %7"
                                                : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(x), "v"(y), "v"(z), "v"(w));
            if (OP == FMA64) asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                                          : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md), "v"(pd));
            if (OP == FMA64_SGPR) asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                                               : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "s"(sarg), "v"(pd));
            if (OP == MUL64) asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4"
                                          : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(md));
            if (OP == ADD64) asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4"
                                          : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(pd));
            if (OP == READLANE) { int s0, s1, s2, s3;
                asm volatile("v_readlane_b32 %0, %4, 3\n\tv_readlane_b32 %1, %5, 5\n\tv_readlane_b32 %2, %6, 7\n\tv_readlane_b32 %3, %7, 9"
                             : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(ia), "v"(ib), "v"(ic), "v"(id));
                asm volatile("" :: "s"(s0), "s"(s1), "s"(s2), "s"(s3)); }
            if (OP == CNDMASK32) asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n\tv_cndmask_b32 %1, %1, %4, vcc\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cndmask_b32 %3, %3, %4, vcc"
                                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m) : "vcc");
            if (OP == MAX32) asm volatile("v_max_f32 %0, %0, %4\n\tv_max_f32 %1, %1, %4\n\tv_max_f32 %2, %2, %4\n\tv_max_f32 %3, %3, %4"
                                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(m));
            if (OP == FMA64_CHAIN1) asm volatile("v_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2\n\tv_fma_f64 %0, %0, %1, %2"
                                                 : "+v"(x) : "v"(md), "v"(pd));
            if (OP == LDEXP64) asm volatile("v_ldexp_f64 %0, %0, 1\n\tv_ldexp_f64 %1, %1, 1\n\tv_ldexp_f64 %2, %2, -1\n\tv_ldexp_f64 %3, %3, -1"
                                            : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1) :: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + A.x + A.y + B.x + B.y + C.x + C.y + D.x + D.y + x + y + z + w + ia + ib + ic + id;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int OP>
void run(const char* name, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd blocks of 4 waves)
    double* out; unsigned long long* clk;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int warm = 0; warm < 3; ++warm) k<OP><<<blocks, 256>>>(out, clk, 1.5f, iters, 1.0000001);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(out, clk, 1.5f, iters, 1.0000001);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> ghz(blocks), cyc(blocks);
    for (int i = 0; i < blocks; ++i) { ghz[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0); cyc[i] = (double)h[2 * i]; }
    std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
    const double winst = (double)waves_per_simd * iters * 8 * 4;     // wave-instructions per SIMD
    const double ns_per = ms * 1e6 / winst;
    printf("%-14s waves/SIMD=%d  %8.3f ms  %6.2f ns/wave-op/SIMD  clock %.2f GHz  => %5.2f cycles/wave-op (block-median in-kernel: %5.2f)\n",
           name, waves_per_simd, ms, ns_per, ghz[blocks / 2], ns_per * ghz[blocks / 2], cyc[blocks / 2] / winst);
    hipFree(out); hipFree(clk);
}

int main()
{
    for (int w : {8, 4, 2}) {
        run<NOPS>("empty", w);
        run<FMA32>("v_fma_f32", w); run<PKFMA32>("v_pk_fma_f32", w); run<PKMUL32>("v_pk_mul_f32", w); run<PKADD32>("v_pk_add_f32", w);
        run<MAX32>("v_max_f32", w); run<CNDMASK32>("v_cndmask_b32", w);
        run<RSQ32>("v_rsq_f32", w); run<RCP32>("v_rcp_f32", w);
        run<CVT_F64_F32>("v_cvt_f64_f32", w); run<CVT_F32_F64>("v_cvt_f32_f64", w);
        run<FMA64>("v_fma_f64", w); run<FMA64_SGPR>("v_fma_f64 sgpr", w); run<MUL64>("v_mul_f64", w); run<ADD64>("v_add_f64", w);
        run<LDEXP64>("v_ldexp_f64", w);
        run<FMA64_CHAIN1>("v_fma_f64 dep", w); run<READLANE>("v_readlane_b32", w);
    }
    return 0;
}
