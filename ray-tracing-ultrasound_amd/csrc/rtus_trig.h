// rtus_trig.h — sin / tan for the forward trace's angle arithmetic, sized for what the trace actually feeds them.
//
// The reference (main_rt.py:267-292, 341-349, 375, 396-401) works in ANGLES: every refraction / reflection is
// atan2 -> +-pi/2 -> asin -> tan.  The generic library routines spend most of their instructions on generality the
// trace never uses (Payne-Hanek reduction for |x| up to 1e308, both polynomials + compare/select chains; on gfx950:
// sin 148, tan 178 VALU instructions).  Every angle here is a short sum of atan2 / atan / asin results and +-pi/2:
// |x| < 8.  So: three-constant Cody-Waite reduction (exact to 118 bits for |n| < 2^20: tan(fl(pi/2)) = 1.633e16 comes
// out to the last bit, which matters because exactly vertical rays ARE such angles), the fdlibm / FreeBSD msun kernel
// polynomials on [-pi/4, pi/4] (< 1 ulp), quadrant fix-up by bit operations instead of select chains.
//
// Plain C++ on purpose (fma / rint only): the same text compiles for the host, where tests/test_trig_kernels.py checks it
// against 50-digit values without a GPU.
//
// The polynomial coefficients, the pi/2 splitting constants and the shape of the kernels (k_sin, k_cos, the medium branch of
// rem_pio2, atan's aT[], asin's pS / qS) restate fdlibm as shipped in FreeBSD msun (k_sin.c, k_cos.c, e_rem_pio2.c, s_atan.c,
// e_asin.c), whose files carry this notice:
//
//   ====================================================
//   Copyright (C) 1993 by Sun Microsystems, Inc. All rights reserved.
//
//   Developed at SunSoft, a Sun Microsystems, Inc. business.
//   Permission to use, copy, modify, and distribute this
//   software is freely granted, provided that this notice
//   is preserved.
//   ====================================================
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define RTUS_HD __host__ __device__ __forceinline__
#else
#define RTUS_HD static inline
#endif

// Polynomial coefficients as SCALAR operands.  hipcc materialises the 64-bit literal addend of a Horner step into the
// destination VGPR pair (two v_mov_b32) and issues v_fmac_f64: 7.8 cycles of VALU issue per step against 4.2 for the
// v_fma_f64 alone; moving the literal into an SGPR pair right in front of the v_fma_f64 is worse (two s_mov_b32: 9.0 —
// the scalar ALU is its own pipe, one instruction per ~4.4 cycles per SIMD; scripts/ubench_issue4.hip).  So the device
// build reads the coefficients from a table in constant memory: one s_load_dwordx8/x16 brings 4 / 8 of them into SGPRs,
// and each step is the bare v_fma_f64 with one scalar operand.  (__constant__ and not const: the compiler must not fold
// the values back into literals.)  The host build uses the literals.
#define RTUS_TRIG_TABLE(X)                                                                                              \
    X(S1, -1.66666666666666324348e-01) X(S2, 8.33333333332248946124e-03) X(S3, -1.98412698298579493134e-04)             \
    X(S4, 2.75573137070700676789e-06) X(S5, -2.50507602534068634195e-08) X(S6, 1.58969099521155010221e-10)              \
    X(C1, 4.16666666666666019037e-02) X(C2, -1.38888888888741095749e-03) X(C3, 2.48015872894767294178e-05)              \
    X(C4, -2.75573143513906633035e-07) X(C5, 2.08757232129817482790e-09) X(C6, -1.13596475577881948265e-11)             \
    X(invpio2, 6.36619772367581382433e-01) X(pio2_1, 1.57079632673412561417e+00)                                        \
    X(pio2_2, 6.07710050630396597660e-11) X(pio2_2t, 2.02226624879595063154e-21)                                        \
    X(aT0, 3.33333333333329318027e-01) X(aT1, -1.99999999998764832476e-01) X(aT2, 1.42857142725034663711e-01)           \
    X(aT3, -1.11111104054623557880e-01) X(aT4, 9.09088713343650656196e-02) X(aT5, -7.69187620504482999495e-02)          \
    X(aT6, 6.66107313738753120669e-02) X(aT7, -5.83357013379057348645e-02) X(aT8, 4.97687799461593236017e-02)           \
    X(aT9, -3.65315727442169155270e-02) X(aT10, 1.62858201153657823623e-02)                                             \
    X(pio4_hi, 7.85398163397448278999e-01) X(pio4_lo, 3.06161699786838301793e-17) X(tan_pio8, 0.41421356237309503)      \
    X(pS0, 1.66666666666666657415e-01) X(pS1, -3.25565818622400915405e-01) X(pS2, 2.01212532134862925881e-01)           \
    X(pS3, -4.00555345006794114027e-02) X(pS4, 7.91534994289814532176e-04) X(pS5, 3.47933107596021167570e-05)           \
    X(qS1, -2.40339491173441421878e+00) X(qS2, 2.02094576023350569471e+00) X(qS3, -6.88283971605453293030e-01)          \
    X(qS4, 7.70381505559019352791e-02) X(pio2_lo, 6.12323399573676603587e-17)
struct RtusTrigTable {
#define X(name, value) double name;
    RTUS_TRIG_TABLE(X)
#undef X
};
#define X(name, value) value,
#ifdef __HIPCC__
static __constant__ RtusTrigTable rtus_k_dev = {RTUS_TRIG_TABLE(X)};
#endif
static const RtusTrigTable rtus_k_host = {RTUS_TRIG_TABLE(X)};
#undef X
// RTUS_KTAB; at the top of a function, then RTUS_K(name).  On the device the table's address passes through an empty asm
// per use site: without it the compiler merges the loads of all call sites of a kernel, keeps ~90 SGPRs of coefficients live
// across the whole trace and spills them to VGPR lanes (v_readlane per use: worse than the literals).
#ifdef __HIP_DEVICE_COMPILE__
// (the pointer carries the constant address space: a uniform load through a plain pointer is only a scalar load while the
// compiler can prove nothing in the kernel writes there, and it cannot once the pointer went through the asm)
typedef const RtusTrigTable __attribute__((address_space(4)))* rtus_ktab_ptr;
// (a __device__-only accessor: a static device variable named inside a __host__ __device__ function is made an external
// symbol, and every use then loads its address from the GOT)
__device__ __forceinline__ rtus_ktab_ptr rtus_ktab_addr() { return (rtus_ktab_ptr)(uintptr_t)&rtus_k_dev; }
#define RTUS_KTAB rtus_ktab_ptr ktab_ = rtus_ktab_addr(); asm("" : "+s"(ktab_))
#define RTUS_K(name) (ktab_->name)
#else
#define RTUS_KTAB
#define RTUS_K(name) rtus_k_host.name
#endif
// RTUS_FMA_K(x, acc, name) = x * acc + table.name as ONE v_fma_f64 whose addend is the scalar register pair the table load
// filled.  Written as asm because the compiler's own choice for fma(x, acc, sgpr) is the two-address v_fmac_f64 behind two
// v_mov_b32 that copy the scalar into the destination (7.8 cycles of issue instead of 4.2).
#ifdef __HIP_DEVICE_COMPILE__
__device__ __forceinline__ double rtus_fma_s(double x, double acc, double c)
{
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(acc), "s"(c));
    return r;
}
#define RTUS_FMA_K(x, acc, name) rtus_fma_s(x, acc, RTUS_K(name))
#else
#define RTUS_FMA_K(x, acc, name) fma(x, acc, RTUS_K(name))
#endif

RTUS_HD uint64_t rtus_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
RTUS_HD double rtus_from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

// a / b, correctly rounded: the compiler's own fp64 division sequence (v_rcp_f64, two Newton steps on the reciprocal, the
// quotient and one correction, v_div_fixup for zero / infinite / NaN operands) WITHOUT its two v_div_scale_f64 and the
// scaling half of v_div_fmas_f64.  Those only act when an exponent is extreme (a denormal or > 2^1000 denominator, or a
// quotient about to leave the normal range); everywhere else they pass their operands through, so the result is the same
// bits.  The trace divides lengths, slopes (<= 1.7e16) and sines.  Two VALU slots fewer per division, and no VCC traffic.
RTUS_HD double rtus_div(double a, double b)
{
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    double q = a * r;
    q = __builtin_fma(__builtin_fma(-b, q, a), r, q);
    return __builtin_amdgcn_div_fixup(q, b, a);
#else
    return a / b;
#endif
}

// a1 / b and a2 / b: rtus_div twice with the reciprocal refined once (13 VALU slots instead of 18; the same bits, since
// each quotient goes through rtus_div's own last steps with the same reciprocal).
RTUS_HD void rtus_div2(double a1, double a2, double b, double& q1, double& q2)
{
#ifdef __HIP_DEVICE_COMPILE__
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    double p1 = a1 * r, p2 = a2 * r;
    p1 = __builtin_fma(__builtin_fma(-b, p1, a1), r, p1);
    p2 = __builtin_fma(__builtin_fma(-b, p2, a2), r, p2);
    q1 = __builtin_amdgcn_div_fixup(p1, b, a1);
    q2 = __builtin_amdgcn_div_fixup(p2, b, a2);
#else
    q1 = a1 / b; q2 = a2 / b;
#endif
}

// a / c for a divisor every lane shares (a sound speed, a lens constant), rc = 1 / c rounded ONCE on the host (an IEEE
// division: the correctly rounded reciprocal).  The quotient estimate a rc is within an ulp; its exact residual
// (one fma) times rc, added back (one fma), is the correctly rounded quotient — the last two steps of rtus_div, which
// spends its first five instructions getting a reciprocal no better than this one.  v_div_fixup for infinite / NaN
// dividends.  4 VALU slots instead of 9; checked bit for bit against a / c by rtus_selftest.
RTUS_HD double rtus_div_by(double a, double c, double rc)
{
#ifdef __HIP_DEVICE_COMPILE__
    double q = a * rc;
    q = __builtin_fma(__builtin_fma(-c, q, a), rc, q);
    return __builtin_amdgcn_div_fixup(q, c, a);
#else
    (void)rc;
    return a / c;
#endif
}

// sqrt(x), correctly rounded: likewise the compiler's own sequence (v_rsq_f64, one coupled Newton step on (g, h) = (sqrt x,
// 1 / 2 sqrt x), two residual corrections, x itself for +-0 and +inf) without the 2^256 pre-scaling it applies to x < 2^-767
// (two v_ldexp_f64, a compare and two selects per call).  Same bits for every x >= 2^-767; the trace's radicands are
// squared lengths and discriminants of them (an exact 0 — a tangent hit — takes the x = 0 branch).
RTUS_HD double rtus_sqrt(double x)
{
#ifdef __HIP_DEVICE_COMPILE__
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;        // +-0, +inf
#else
    return sqrt(x);
#endif
}

// x = n pi/2 + (r + t), |r| <= pi/4 (+ rounding), t the tail of r.  fdlibm __ieee754_rem_pio2, medium branch,
// always with the second pair of constants: pi/2 = p1 + p2 + p2t to 118 bits; n p1 and n p2 are exact for |n| < 2^20.
RTUS_HD void rtus_rem_pio2(double x, double& r, double& t, int& n)
{
    RTUS_KTAB;
    const double invpio2 = RTUS_K(invpio2);
    const double p1 = RTUS_K(pio2_1);     // first 33 bits of pi/2
    const double p2 = RTUS_K(pio2_2);     // next 33 bits
    const double p2t = RTUS_K(pio2_2t);   // pi/2 - (p1 + p2)
    const double fn = rint(x * invpio2);
    n = (int)fn;
    const double r0 = fma(-fn, p1, x);
    double w = fn * p2;
    const double r1 = r0 - w;
    w = fma(fn, p2t, -((r0 - r1) - w));
    r = r1 - w;
    t = (r1 - r) - w;
}

// sin(x + y) for |x| <= pi/4, y the tail of x (FreeBSD msun k_sin.c; error < 0.56 ulp)
RTUS_HD double rtus_ksin(double x, double y)
{
    RTUS_KTAB;
    const double z = x * x;
    double r = RTUS_FMA_K(z, RTUS_K(S6), S5);             // Horner: one v_fma_f64 per step, the coefficient a scalar operand
    r = RTUS_FMA_K(z, r, S4);
    r = RTUS_FMA_K(z, r, S3);
    r = RTUS_FMA_K(z, r, S2);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * RTUS_K(S1));
}

// cos(x + y) for |x| <= pi/4 (FreeBSD msun k_cos.c; error < 0.51 ulp)
RTUS_HD double rtus_kcos(double x, double y)
{
    RTUS_KTAB;
    const double z = x * x;
    double r = RTUS_FMA_K(z, RTUS_K(C6), C5);
    r = RTUS_FMA_K(z, r, C4);
    r = RTUS_FMA_K(z, r, C3);
    r = RTUS_FMA_K(z, r, C2);
    r = z * RTUS_FMA_K(z, r, C1);
    const double hz = 0.5 * z;
    double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + (z * r - x * y));
}

// a for an all-ones mask, b for a zero mask, per 32-bit half (v_bfi_b32): a compare + select chain on VCC costs several
// times as much on gfx950 as two bit-field inserts.
RTUS_HD double rtus_bitsel(uint32_t mask, double a, double b)
{
#ifdef __HIP_DEVICE_COMPILE__
    asm("" : "+v"(mask));               // opaque value: keeps the compiler from turning the bit operations back into selects
                                        // (NOT volatile: a volatile asm counts as a memory clobber and turns every later
                                        // wave-uniform scalar load of the kernel into a vector load)
#endif
    const uint64_t m = ((uint64_t)mask << 32) | mask;
    return rtus_from_bits((rtus_bits(a) & m) | (rtus_bits(b) & ~m));
}

RTUS_HD double rtus_sin(double x)
{
    double r, t; int n;
    rtus_rem_pio2(x, r, t, n);
    const double s = rtus_ksin(r, t), c = rtus_kcos(r, t);
    const double v = rtus_bitsel((uint32_t)-(n & 1), c, s);                  // odd quadrant: cos
    return rtus_from_bits(rtus_bits(v) ^ ((uint64_t)(n & 2) << 62));        // quadrants 2, 3: negate
}

RTUS_HD double rtus_cos(double x)
{
    double r, t; int n;
    rtus_rem_pio2(x, r, t, n);
    const double s = rtus_ksin(r, t), c = rtus_kcos(r, t);
    const double v = rtus_bitsel((uint32_t)-(n & 1), s, c);
    return rtus_from_bits(rtus_bits(v) ^ ((uint64_t)((n + 1) & 2) << 62)); // quadrants 1, 2: negate
}

RTUS_HD void rtus_sincos(double x, double& sn, double& cs)
{
    double r, t; int n;
    rtus_rem_pio2(x, r, t, n);
    const double s = rtus_ksin(r, t), c = rtus_kcos(r, t);
    const uint32_t odd = (uint32_t)-(n & 1);
    sn = rtus_from_bits(rtus_bits(rtus_bitsel(odd, c, s)) ^ ((uint64_t)(n & 2) << 62));
    cs = rtus_from_bits(rtus_bits(rtus_bitsel(odd, s, c)) ^ ((uint64_t)((n + 1) & 2) << 62));
}

// tan as sin / cos of the reduced argument (odd quadrant: -cos / sin); one IEEE division; < 2 ulp.
// x = fl(pi/2) gives 1.633123935319537e16 exactly as the libraries do (the reduction keeps 118 bits).
// steep: |tan x| > ~300 (the line is within 0.2 degrees of vertical).  The reference forms lines in slope-intercept
// form and intersects the first one with the pipe through the quadratic formula (main_rt.py:349-364): for such a line that
// arithmetic amplifies a last-bit difference of the angle by ~|tan|^3, so trace_ray (rtus_shoot.hip) keeps the library
// routines for it.
RTUS_HD double rtus_tan(double x, bool& steep)
{
    double r, t; int n;
    rtus_rem_pio2(x, r, t, n);
    const double s = rtus_ksin(r, t), c = rtus_kcos(r, t);
    const uint32_t odd = (uint32_t)-(n & 1);
    steep = (n & 1) && fabs(r) < (1.0 / 300.0);
    const double num = rtus_bitsel(odd, -c, s), den = rtus_bitsel(odd, s, c);
    return rtus_div(num, den);
}
RTUS_HD double rtus_tan(double x) { bool steep; return rtus_tan(x, steep); }

// ---- inverse functions ---------------------------------------------------------------------------------------------
// The library atan2 / atan / asin cost 105 / 83 / 100 VALU instructions each on gfx950, two in five of them v_mov's that
// materialise polynomial coefficients (see RTUS_TRIG_TABLE).  The trace calls five of them per ray.  Below: the fdlibm
// polynomials (e_atan.c aT[], e_asin.c pS / qS; < 1 ulp on their intervals) behind argument reductions that need no
// select chains: the reduction's choices are carried as small numbers (0 / 1, +-1, a count of quarter turns) that enter
// through fma's.

// Hardware seeds, ~5e-8 relative (v_rsq_f64 / v_rcp_f64); the host twins are off by 4e-8 so the CPU tests exercise the same
// tolerance.
RTUS_HD double rtus_rsq_seed(double x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_rsq(x);
#else
    return (1.0 / sqrt(x)) * (1.0 + 4e-8);
#endif
}
RTUS_HD double rtus_rcp_seed(double x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_amdgcn_rcp(x);
#else
    return (1.0 / x) * (1.0 - 4e-8);
#endif
}

// atan(r) = r - r S(r^2) for |r| <= 0.4375 (fdlibm s_atan.c, the unreduced branch): returns S, z = r^2
RTUS_HD double rtus_katan(double z)
{
    RTUS_KTAB;
    const double w = z * z;
    double s1 = RTUS_FMA_K(w, RTUS_K(aT10), aT8);
    s1 = RTUS_FMA_K(w, s1, aT6);
    s1 = RTUS_FMA_K(w, s1, aT4);
    s1 = RTUS_FMA_K(w, s1, aT2);
    s1 = z * RTUS_FMA_K(w, s1, aT0);
    double s2 = RTUS_FMA_K(w, RTUS_K(aT9), aT7);
    s2 = RTUS_FMA_K(w, s2, aT5);
    s2 = RTUS_FMA_K(w, s2, aT3);
    s2 = w * RTUS_FMA_K(w, s2, aT1);
    return s1 + s2;
}

// atan2(y, x), NumPy / C semantics: NaN in -> NaN out, signed zeros, x = y = 0 gives 0 or +-pi by the sign of x, one infinite
// argument (both infinite is not supported: NaN).  Magnitudes in {0} U [1e-290, 1e290].
// t = min(|x|,|y|) / max(|x|,|y|) in [0,1]; above tan(pi/8) one more reduction (t-1)/(t+1) — folded into the SAME division:
// r = (mn - s mx) / (mx + s mn), s = 0 / 1.  Then atan2 = sigma atan(r) + K pi/4 with sigma = +-1 and K = 0..4 counting the
// reflections (swap: pi/2 - a; x < 0: pi - a).  One reciprocal, no fp64 selects.  The rounding errors of the numerator and
// the denominator are carried (Fast2Sum) and enter as a first-order correction of the result: < 1.3 ulp measured.
RTUS_HD double rtus_atan2(double y, double x)
{
    RTUS_KTAB;
    const double pio4_hi = RTUS_K(pio4_hi), pio4_lo = RTUS_K(pio4_lo);
    const double ax = fabs(x), ay = fabs(y);
    const double mn = fmin(ax, ay), mx = fmin(fmax(ax, ay), 1e300);          // an infinite argument: ratio 0 all the same
    const bool s = mn > RTUS_K(tan_pio8) * mx;
    const bool sw = ay > ax;
    const bool nx = (int64_t)rtus_bits(x) < 0;
    const float sf = s ? 1.0f : 0.0f;
    const double sd = (double)sf;
    const double num = fma(-sd, mx, mn), en = fma(-sd, num + mx, mn) - (1.0 - sd) * num;   // mn - s mx = num + en
    const double den0 = fma(sd, mn, mx), ed = (mx - den0) + sd * mn;                       // mx + s mn = den0 + ed
    const double den = fmax(den0, 2.2250738585072014e-308);                                // x = y = 0: 0 / tiny = 0
    double q = rtus_rcp_seed(den);
    q = fma(fma(-den, q, 1.0), q, q);
    q = fma(fma(-den, q, 1.0), q, q);                                                      // 1 / den to ~1 ulp
    double r = num * q;
    r = fma(fma(-r, den, num), q, r);                                                      // num / den to ~0.5 ulp
    const double z = r * r;
    const double tail = fma(-r, ed, en) * q;                                               // what the two roundings took
    const double m1 = fma(-r, rtus_katan(z), fma(-z, tail, tail));                         // atan(r) - r; d atan = dr / (1 + r^2)
    float K = sw ? 2.0f - sf : sf;
    K = nx ? 4.0f - K : K;
    K = (x != x || y != y) ? NAN : K;
    const float sg = (sw != nx) ? -1.0f : 1.0f;
    const double Kd = (double)K;
    const double sgd = (double)sg;
    const double v = fma(Kd, pio4_hi, sgd * r) + fma(sgd, m1, Kd * pio4_lo);               // two roundings at the size of v
    return rtus_from_bits((rtus_bits(v) & 0x7fffffffffffffffull) | (rtus_bits(y) & 0x8000000000000000ull));
}

RTUS_HD double rtus_atan(double x) { return rtus_atan2(x, 1.0); }

// asin(x) (fdlibm e_asin.c): |x| < 0.5: x + x R(x^2), R = p/q rational; otherwise with t = (1 - |x|)/2, s = sqrt(t):
// pi/2 - 2 (s + s R(t)), evaluated fdlibm's way around pi/4 with s split into a short head sh (exact products) and the rest
// c = sqrt(t) - sh = (t - sh^2) / (s + sh) — so s itself only has to be good to ~1e-14: hardware seed + one coupled
// Newton step instead of a correctly rounded square root.  |x| > 1 -> NaN (the trace's total internal reflection);
// |x| = 1 -> +-pi/2.
RTUS_HD double rtus_asin_R(double t)
{
    RTUS_KTAB;
    double p = RTUS_FMA_K(t, RTUS_K(pS5), pS4);
    p = RTUS_FMA_K(t, p, pS3);
    p = RTUS_FMA_K(t, p, pS2);
    p = RTUS_FMA_K(t, p, pS1);
    p = t * RTUS_FMA_K(t, p, pS0);
    double q = RTUS_FMA_K(t, RTUS_K(qS4), qS3);
    q = RTUS_FMA_K(t, q, qS2);
    q = RTUS_FMA_K(t, q, qS1);
    q = fma(t, q, 1.0);
    return rtus_div(p, q);
}
RTUS_HD double rtus_asin_small(double x) { return fma(x, rtus_asin_R(x * x), x); }       // |x| < 0.5
RTUS_HD double rtus_asin_big(double x)                                                   // |x| >= 0.5 (> 1: NaN)
{
    RTUS_KTAB;
    const double pio2_lo = RTUS_K(pio2_lo), pio4_hi = RTUS_K(pio4_hi);
    const double ax = fabs(x);
    const double t0 = fma(ax, -0.5, 0.5);
    const double t = t0 == 0.0 ? 1e-60 : t0;              // |x| = 1: s = 1e-30, the result rounds to pi/2
    const double w = rtus_asin_R(t);
    const double y0 = rtus_rsq_seed(t);                   // t < 0: NaN from here on
    const double s0 = t * y0, h0 = 0.5 * y0;
    const double e = fma(-h0, s0, 0.5);
    const double s = fma(s0, e, s0), h = fma(h0, e, h0);  // s = sqrt(t) (1 + ~1e-14), h = 1 / (2 s)
    const double sh = rtus_from_bits(rtus_bits(s) & 0xffffffff00000000ull);
    const double c = (s - sh) + fma(-s, s, t) * h;        // sqrt(t) - sh
    const double sa = sh + c;                             // sqrt(t), rounded
    const double p = fma(sa + sa, w, -fma(-2.0, c, pio2_lo));
    const double q = fma(-2.0, sh, pio4_hi);              // exact
    const double r = pio4_hi - (p - q);
    return rtus_from_bits(rtus_bits(r) | (rtus_bits(x) & 0x8000000000000000ull));
}
RTUS_HD double rtus_asin(double x)
{
    return fabs(x) >= 0.5 ? rtus_asin_big(x) : rtus_asin_small(x);      // NaN (a dead ray) takes the short form
}
