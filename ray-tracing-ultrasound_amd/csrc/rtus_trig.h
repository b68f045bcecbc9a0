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
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __HIPCC__
#define RTUS_HD __host__ __device__ __forceinline__
#else
#define RTUS_HD static inline
#endif

RTUS_HD uint64_t rtus_bits(double v) { uint64_t u; memcpy(&u, &v, 8); return u; }
RTUS_HD double rtus_from_bits(uint64_t u) { double v; memcpy(&v, &u, 8); return v; }

// x = n pi/2 + (r + t), |r| <= pi/4 (+ rounding), t the tail of r.  fdlibm __ieee754_rem_pio2, medium branch,
// always with the second pair of constants: pi/2 = p1 + p2 + p2t to 118 bits; n p1 and n p2 are exact for |n| < 2^20.
RTUS_HD void rtus_rem_pio2(double x, double& r, double& t, int& n)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double p1 = 1.57079632673412561417e+00;    // first 33 bits of pi/2
    const double p2 = 6.07710050630396597660e-11;    // next 33 bits
    const double p2t = 2.02226624879595063154e-21;   // pi/2 - (p1 + p2)
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
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = x * x, w = z * z;
    const double r = fma(z, fma(z, S4, S3), S2) + z * w * fma(z, S6, S5);
    const double v = z * x;
    return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

// cos(x + y) for |x| <= pi/4 (FreeBSD msun k_cos.c; error < 0.51 ulp)
RTUS_HD double rtus_kcos(double x, double y)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = x * x;
    double w = z * z;
    const double r = z * fma(z, fma(z, C3, C2), C1) + (w * w) * fma(z, fma(z, C6, C5), C4);
    const double hz = 0.5 * z;
    w = 1.0 - hz;
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
    return num / den;
}
RTUS_HD double rtus_tan(double x) { bool steep; return rtus_tan(x, steep); }
