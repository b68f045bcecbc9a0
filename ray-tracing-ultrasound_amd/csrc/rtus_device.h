// rtus_device.h — device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rtus.h"
#include "rtus_trig.h"

#define RTUS_WAVE 64
#define RTUS_BLOCK 256          // 4 waves per workgroup, one per SIMD
#define RTUS_CURVE_BLK 64       // polyline points per bounding block (= one wave of the curve kernel)

#define RTUS_PI_2 1.57079632679489661923

// Lens constants hoisted out of the per-ray math (all wave-uniform -> SGPRs).
// Restates the scalar prologue of h_from_alpha / dh_from_alpha (main_rt.py:181-201).
struct LensK {
    double c1, c2, d;
    double inv_c1, inv_c2;   // 1/c1, 1/c2, host-rounded: the vector-form mode multiplies by them, the compat mode divides through
                             // them (rtus_div_by: correctly rounded)
    double inv_twoA;         // 1/(2A), likewise
    double eta12, eta21;     // c1/c2, c2/c1: the refraction ratios of main_rt.py:398 / :345, divided once on the host
    double A;        // c1^2/c2^2 - 1                    main_rt.py:183
    double C4A;      // 4*A*C, C = c1^2 T^2 - d^2        main_rt.py:185, 172
    double twoA;     // 2*A                              main_rt.py:173
    double phi_1;    // -1/(2A)                          main_rt.py:199
    double phi_2;    // -2 T c1^2 / c2                   main_rt.py:200
    double phi_3;    // 2 d                              main_rt.py:201
    double twoTc;    // 2 T c1^2 / c2                    main_rt.py:184
};

static inline LensK make_lens_k(const rtus_lens& L)
{
    LensK k;
    k.c1 = L.c1; k.c2 = L.c2; k.d = L.d;
    k.inv_c1 = 1.0 / L.c1; k.inv_c2 = 1.0 / L.c2;
    k.eta12 = L.c1 / L.c2; k.eta21 = L.c2 / L.c1;
    double T = L.l0 / L.c1 + L.h0 / L.c2;
    double c1sq = L.c1 * L.c1;
    k.A = c1sq / (L.c2 * L.c2) - 1.0;
    double C = c1sq * (T * T) - L.d * L.d;
    k.C4A = 4.0 * k.A * C;
    k.twoA = 2.0 * k.A;
    k.inv_twoA = 1.0 / k.twoA;
    k.phi_1 = -1.0 / (2.0 * k.A);
    k.phi_2 = (-2.0 * T * c1sq) / L.c2;
    k.phi_3 = 2.0 * L.d;
    k.twoTc = 2.0 * T * c1sq / L.c2;
    return k;
}

// Lens point + tangent at polar angle alpha.
// h:  main_rt.py:180-189 (root index [1] of roots_bhaskara :171-177)
// dh: main_rt.py:192-214;  x,z: :217-223;  dz,dx: :226-234
// lens_eval_sc takes sin(alpha), cos(alpha): the trace needs the lens at alpha = atan2(x_i, z_i) (main_rt.py:396-397),
// whose sine and cosine are x_i / rho and z_i / rho — no angle has to be formed at all.
__device__ __forceinline__ void lens_eval_sc(const LensK& k, double s, double c, double& x, double& z,
                                             double& dz, double& dx)
{
    double B = k.phi_3 * c - k.twoTc;              // 2 d cos(a) - 2 T c1^2/c2
    double sq = rtus_sqrt(B * B - k.C4A);
    double h = rtus_div_by(-B - sq, k.twoA, k.inv_twoA);
    double dB = -k.phi_3 * s;
    double dS = rtus_div(1.0, 2.0 * sq) * (2.0 * B * dB);
    double dh = k.phi_1 * (dB + dS);
    x = h * s;
    z = h * c;
    dz = dh * c - h * s;
    dx = dh * s + h * c;
}
__device__ __forceinline__ void lens_eval(const LensK& k, double alpha, double& x, double& z,
                                          double& dz, double& dx)
{
    double s, c;
    sincos(alpha, &s, &c);
    lens_eval_sc(k, s, c, x, z, dz, dx);
}

// refraction(), angle form (main_rt.py:267-280) with the surface-slope angle already known.
// theta_2 = asin((v2/v1) sin(theta_1)); |arg| > 1 -> NaN = total internal reflection (Q6).
__device__ __forceinline__ double refract_angle(double phi_in, double phi_slope, double v2_over_v1)
{
    double theta_1 = phi_in - (phi_slope + RTUS_PI_2);
    double theta_2 = rtus_asin(v2_over_v1 * rtus_sin(theta_1));     // |theta_1| < 8: bounded-range kernels (rtus_trig.h)
    return phi_slope - RTUS_PI_2 + theta_2;
}

__device__ __forceinline__ double dist2d(double x1, double z1, double x2, double z2)
{
    double dx = x1 - x2, dz = z1 - z2;             // main_rt.py:444-445
    return rtus_sqrt(dx * dx + dz * dz);
}

// np.isclose(a, b, rtol, atol) for scalars, equal_nan=False.
__device__ __forceinline__ bool np_isclose(double a, double b, double rtol, double atol)
{
    if (isnan(a) || isnan(b)) return false;
    if (isinf(a) || isinf(b)) return a == b;
    return fabs(a - b) <= atol + rtol * fabs(b);
}
