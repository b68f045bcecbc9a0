// Micro-benchmarks that guide the fp64 kernels: accuracy of v_rsq_f64 / v_rcp_f64 seeds and the
// issue cost (cycles per wave-instruction per SIMD) of the fp64 ops the solvers are made of.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_fp64.hip -o scripts/ubench_fp64 && ./scripts/ubench_fp64
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void acc_kernel(const double* x, double* rsq, double* rcp, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rsq[i] = __builtin_amdgcn_rsq(x[i]); rcp[i] = __builtin_amdgcn_rcp(x[i]); }
}

template <int OP>
__global__ __launch_bounds__(256) void tp_kernel(double* out, double seed, int iters)
{
    double a = seed + threadIdx.x * 1e-3, b = 1.0000001, c = 0.999, d = a + 1, e = a + 2, f = a + 3;
    unsigned long long acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a = fma(a, b, c); d = fma(d, b, c); e = fma(e, b, c); f = fma(f, b, c); }
            if (OP == 1) { a = a * b; d = d * b; e = e * b; f = f * b; }
            if (OP == 2) { a = a + b; d = d + b; e = e + b; f = f + b; }
            if (OP == 3) { a = __builtin_amdgcn_rsq(a); d = __builtin_amdgcn_rsq(d); e = __builtin_amdgcn_rsq(e); f = __builtin_amdgcn_rsq(f); }
            if (OP == 4) { a = __builtin_amdgcn_rcp(a); d = __builtin_amdgcn_rcp(d); e = __builtin_amdgcn_rcp(e); f = __builtin_amdgcn_rcp(f); }
            if (OP == 5) { a = sqrt(a); d = sqrt(d); e = sqrt(e); f = sqrt(f); }
            if (OP == 6) { a = b / a; d = b / d; e = b / e; f = b / f; }
            if (OP == 7) { a = (a < d) ? e : f; d = (d > e) ? f : a; e = (e < f) ? a : d; f = (f > a) ? d : e; }   // cmp + 2 cndmask
            if (OP == 8) { a = fmax(a, b); d = fmax(d, b); e = fmin(e, b); f = fmin(f, b); }
            if (OP == 9) { a = sin(a); d = sin(d); e = sin(e); f = sin(f); }
            if (OP == 10) { a = atan2(a, b); d = atan2(d, b); e = atan2(e, b); f = atan2(f, b); }
            if (OP == 11) { a = asin(a * 1e-3); d = asin(d * 1e-3); e = asin(e * 1e-3); f = asin(f * 1e-3); }
            if (OP == 12) { a = tan(a); d = tan(d); e = tan(e); f = tan(f); }
            if (OP == 14) { float x=(float)a,y=(float)d,z=(float)e,w=(float)f; for(int q=0;q<4;++q){x=fmaf(x,1.0000001f,0.999f);y=fmaf(y,1.0000001f,0.999f);z=fmaf(z,1.0000001f,0.999f);w=fmaf(w,1.0000001f,0.999f);} a=x;d=y;e=z;f=w; }  // 16 fmaf + 8 cvt
            if (OP == 15) { float x=(float)a,y=(float)d,z=(float)e,w=(float)f; for(int q=0;q<4;++q){x=__builtin_amdgcn_rsqf(x);y=__builtin_amdgcn_rsqf(y);z=__builtin_amdgcn_rsqf(z);w=__builtin_amdgcn_rsqf(w);} a=x;d=y;e=z;f=w; }  // 16 rsqf + 8 cvt
            if (OP == 16) { float x=(float)a,y=(float)d,z=(float)e,w=(float)f; a=x;d=y;e=z;f=w; }  // 8 cvt only
            if (OP == 17) { a = (b > 1.0) ? d : e; d = (b > 1.0) ? e : f; e = (b > 1.0) ? f : a; f = (b > 1.0) ? a : d; b += 1e-9; }  // selects
            if (OP == 18) { unsigned long long m0 = __ballot(a < d), m1 = __ballot(d < e), m2 = __ballot(e < f), m3 = __ballot(f < a);
                            acc ^= m0 ^ (m1 << 1) ^ (m2 << 2) ^ (m3 << 3); a += 1e-9; d += 1e-9; e += 1e-9; f += 1e-9; }   // 4 v_cmp_f64 + 4 add
            if (OP == 19) { unsigned long long m0 = __ballot(__double2hiint(a - d) < 0), m1 = __ballot(__double2hiint(d - e) < 0), m2 = __ballot(__double2hiint(e - f) < 0), m3 = __ballot(__double2hiint(f - a) < 0);
                            acc ^= m0 ^ (m1 << 1) ^ (m2 << 2) ^ (m3 << 3); a += 1e-9; d += 1e-9; e += 1e-9; f += 1e-9; }   // 4 (sub + v_cmp_i32) + 4 add
            if (OP == 13) { float x = (float)a; x = __builtin_amdgcn_rsqf(x); a = x; float y = (float)d; y = __builtin_amdgcn_rsqf(y); d = y;
                            float z = (float)e; z = __builtin_amdgcn_rsqf(z); e = z; float w = (float)f; w = __builtin_amdgcn_rsqf(w); f = w; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + d + e + f + (double)(acc & 0xff);
}

template <int OP>
double run(const char* name, int waves_per_simd)
{
    double* out;
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (waves_per_simd blocks of 4 waves)
    hipMalloc(&out, sizeof(double) * blocks * 256);
    const int iters = (OP >= 9 && OP <= 12) ? 200 : 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    tp_kernel<OP><<<blocks, 256>>>(out, 1.5, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    tp_kernel<OP><<<blocks, 256>>>(out, 1.5, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: waves_per_simd * iters * 8 * 4
    double winst = (double)waves_per_simd * iters * 8 * 4;
    double ns_per = ms * 1e6 / winst;
    printf("%-10s waves/SIMD=%d  %.3f ms  %.2f ns per wave-op per SIMD (= %.1f cyc @2.4GHz)\n", name, waves_per_simd,
           ms, ns_per, ns_per * 2.4);
    hipFree(out);
    return ns_per;
}

int main()
{
    const int n = 1 << 20;
    std::vector<double> x(n), r1(n), r2(n);
    for (int i = 0; i < n; ++i) x[i] = exp(((double)rand() / RAND_MAX - 0.5) * 40.0);
    double *dx, *d1, *d2;
    hipMalloc(&dx, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    acc_kernel<<<n / 256, 256>>>(dx, d1, d2, n);
    hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
    double e1 = 0, e2 = 0;
    for (int i = 0; i < n; ++i) {
        e1 = fmax(e1, fabs(r1[i] * sqrt(x[i]) - 1.0));
        e2 = fmax(e2, fabs(r2[i] * x[i] - 1.0));
    }
    printf("v_rsq_f64 max rel err %.3e   v_rcp_f64 max rel err %.3e\n", e1, e2);
    for (int w : {8}) {
        run<0>("fma", w); run<1>("mul", w); run<2>("add", w); run<3>("rsq", w); run<4>("rcp", w);
        run<5>("sqrt", w); run<6>("div", w); run<7>("cmp+sel", w); run<13>("cvt+rsqf", w); run<14>("16fmaf+8cvt", w); run<15>("16rsqf+8cvt", w); run<16>("8cvt", w); run<17>("select", w); run<18>("cmpf64+add", w); run<19>("sub+cmpi32+add", w);
    }
    run<9>("sin", 8); run<10>("atan2", 8); run<11>("asin", 8); run<12>("tan", 8);
    return 0;
}
