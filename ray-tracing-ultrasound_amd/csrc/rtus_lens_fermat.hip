// rtus_lens_fermat.hip — element x focal-point Fermat travel times through the reference's CURVED
// lens surface (BASELINE config 4: "curved parametric interface", fp32 or fp64).
//
// Geometry is the reference's: element A = (x_a, z_a) inside the lens (speed c1), target F in the
// water below it (speed c2), interface P(alpha) = h(alpha) (sin alpha, cos alpha) with the aplanatic
// h(alpha) of main_rt.py:180-189 and its derivative main_rt.py:192-214.  The reference itself only
// traces rays FORWARD from a launch-angle grid; it has no two-point solver, so this kernel is the
// build's own ("parity unpinned" as a solver) — but Fermat <=> Snell, so it is pinned indirectly:
// for every forward-traced reference ray, solving (A, F = its pipe hit point) must return that ray's
// alpha and tof_1 + tof_2 (tests/test_gpu_lens_fermat.py uses the reference goldens for exactly that).
//
// Per lane: one (element, target) pair; safeguarded Newton on g(alpha) = dT/dalpha with the analytic
// g' (needs h''), bracket kept by sign of g, bisection step whenever Newton leaves the bracket.
// Lanes = consecutive targets (coalesced loads/stores); the workgroup loops over `eb` elements and
// starts each solve from the previous element's alpha (continuation).
#include "rtus_device.h"

template <typename R> struct LensFermatArgs {
    R c1inv, c2inv;          // 1/c1, 1/c2
    R phi_3, twoTc, C4A, inv2A;   // lens constants (see LensK)
    R a_lo, a_hi;            // search interval for alpha
    const R* __restrict__ xe;
    const R* __restrict__ ze;
    const R* __restrict__ xf;
    const R* __restrict__ zf;
    R* __restrict__ tt;
    R* __restrict__ alpha_out;   // nullable
    int n_e, n_f, eb;
};

template <typename R> __device__ __forceinline__ R rsqrt_r(R v);
template <> __device__ __forceinline__ float rsqrt_r<float>(float v) { return __builtin_amdgcn_rsqf(v); }
template <> __device__ __forceinline__ double rsqrt_r<double>(double v)
{
    double y = (double)__builtin_amdgcn_rsqf((float)v);
    const double e = fma(-v * y, y, 1.0);
    y = fma(y * e, fma(e, 0.375, 0.5), y);                 // cubic step: 1e-7 -> ~1e-21 (rounding-limited)
    return y;
}
template <typename R> __device__ __forceinline__ void sincos_r(R a, R* s, R* c);
template <> __device__ __forceinline__ void sincos_r<float>(float a, float* s, float* c) { sincosf(a, s, c); }
template <> __device__ __forceinline__ void sincos_r<double>(double a, double* s, double* c) { sincos(a, s, c); }

// T(alpha), g = dT/dalpha, g' for one (A, F).
template <typename R>
__device__ __forceinline__ void lens_time(const LensFermatArgs<R>& k, R alpha, R xa, R za, R xf, R zf, R& T, R& g,
                                          R& gp)
{
    R s, c;
    sincos_r<R>(alpha, &s, &c);
    const R B = k.phi_3 * c - k.twoTc;                      // main_rt.py:184
    const R B1 = -k.phi_3 * s, B2 = -k.phi_3 * c;           // B', B''
    const R disc = B * B - k.C4A;
    const R rS = rsqrt_r<R>(disc);                          // 1/S
    const R S = disc * rS;
    const R h = -(B + S) * k.inv2A;                         // :171-177 root [1]
    const R BrS = B * rS;
    const R h1 = -B1 * (R(1) + BrS) * k.inv2A;              // :199-212
    const R h2 = -(B2 * (R(1) + BrS) + B1 * B1 * rS * (R(1) - BrS * BrS)) * k.inv2A;
    const R px = h * s, pz = h * c;                         // :220-221
    const R p1x = h1 * s + pz, p1z = h1 * c - px;           // :231-232
    const R p2x = h2 * s + R(2) * h1 * c - px, p2z = h2 * c - R(2) * h1 * s - pz;
    const R ax = px - xa, az = pz - za, fx = px - xf, fz = pz - zf;
    const R ra = rsqrt_r<R>(ax * ax + az * az), rf = rsqrt_r<R>(fx * fx + fz * fz);
    const R la = (ax * ax + az * az) * ra, lf = (fx * fx + fz * fz) * rf;
    const R ua = (ax * p1x + az * p1z) * ra, uf = (fx * p1x + fz * p1z) * rf;     // u . P'
    const R pp = p1x * p1x + p1z * p1z;
    T = la * k.c1inv + lf * k.c2inv;
    g = ua * k.c1inv + uf * k.c2inv;
    gp = ((pp - ua * ua) * ra + (ax * p2x + az * p2z) * ra) * k.c1inv
       + ((pp - uf * uf) * rf + (fx * p2x + fz * p2z) * rf) * k.c2inv;
}

template <typename R>
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_tt_lens_kernel(LensFermatArgs<R> k)
{
    const int f_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = f_raw < k.n_f;
    const int f = live ? f_raw : k.n_f - 1;
    const R xf = k.xf[f], zf = k.zf[f];
    const int e0 = blockIdx.y * k.eb, e1 = min(e0 + k.eb, k.n_e);
    const int lane = threadIdx.x & 63;
    const int el = min(e0 + lane, k.n_e - 1);
    const R xe_v = k.xe[el], ze_v = k.ze[el];
    const R tol = sizeof(R) == 4 ? R(1e-5) : R(1e-13);      // |d alpha| at which Newton has converged (rad)
    // ... or when the step can no longer lower T noticeably (T is FLAT in alpha near the lens focus — the lens is
    // aplanatic — so alpha is ill-conditioned there while T is not): predicted gain g*step/2 below the type's resolution
    const R tolT = sizeof(R) == 4 ? R(2e-12) : R(1e-21);

    R alpha = R(0.5) * (k.a_lo + k.a_hi);
    bool have = false;
    for (int e = e0; e < e1; ++e) {
        const R xa = __shfl(xe_v, e - e0), za = __shfl(ze_v, e - e0);
        if (!have) {
            // first guess: polar angle of the point where the straight chord A-F meets the lens apex height
            const R hz = -(k.phi_3 - k.twoTc + sqrt((k.phi_3 - k.twoTc) * (k.phi_3 - k.twoTc) - k.C4A)) * k.inv2A;
            const R t = (za - hz) / (za - zf);
            alpha = atan2(xa + t * (xf - xa), hz);
            have = true;
        }
        alpha = fmin(fmax(alpha, k.a_lo), k.a_hi);
        R lo = k.a_lo, hi = k.a_hi, T, g, gp;
        bool done = false;
        for (int trip = 0; trip < 80; ++trip) {             // wave-uniform trip count, ballot exit
            lens_time<R>(k, alpha, xa, za, xf, zf, T, g, gp);
            if (g > R(0)) hi = alpha; else lo = alpha;      // T decreases left of the minimum
            R step = -g / gp;
            R next = alpha + step;
            const bool bad = !(gp > R(0)) || !(next > lo) || !(next < hi);
            if (bad) { next = R(0.5) * (lo + hi); step = next - alpha; }
            const bool small = !(fabs(step) > tol) || !(fabs(g * step) > tolT) || done;
            if (__all(small)) break;
            if (!small) alpha = next; else done = true;     // a finished lane keeps its alpha (T, g belong to it)
        }
        // second-order polish without another evaluation: T(a*) = T(a) - g^2 / (2 g')
        // (only where Newton converged in the interior; a minimum pinned at an interval end keeps T(alpha))
        const bool interior = gp > R(0) && (fabs(g) <= gp * (R(16) * tol) || fabs(g * g) <= gp * (R(16) * tolT));
        if (interior) T -= R(0.5) * g * g / gp;
        if (live) {
            const size_t o = (size_t)e * k.n_f + f;
            k.tt[o] = T;
            if (k.alpha_out) k.alpha_out[o] = interior ? alpha - g / gp : alpha;
        }
    }
}

template <typename R>
static hipError_t launch_lens(const rtus_lens& L, double a_lo, double a_hi, const R* xe, const R* ze, int n_e,
                              const R* xf, const R* zf, int n_f, R* tt, R* alpha_out, hipStream_t s)
{
    const LensK kk = make_lens_k(L);
    LensFermatArgs<R> k;
    k.c1inv = (R)(1.0 / L.c1); k.c2inv = (R)(1.0 / L.c2);
    k.phi_3 = (R)kk.phi_3; k.twoTc = (R)kk.twoTc; k.C4A = (R)kk.C4A; k.inv2A = (R)(1.0 / kk.twoA);
    k.a_lo = (R)a_lo; k.a_hi = (R)a_hi;
    k.xe = xe; k.ze = ze; k.xf = xf; k.zf = zf; k.tt = tt; k.alpha_out = alpha_out;
    k.n_e = n_e; k.n_f = n_f;
    const long long wave_solves = (long long)((n_f + 63) / 64) * n_e;
    int eb = (int)(wave_solves / (1024LL * 4));
    k.eb = eb < 1 ? 1 : (eb > 32 ? 32 : eb);
    while ((n_e + k.eb - 1) / k.eb > 65535 && k.eb < 64) ++k.eb;   // grid.y limit (eb <= 64: one lane per element)
    const dim3 grid((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK, (n_e + k.eb - 1) / k.eb);
    hipLaunchKernelGGL(rtus_tt_lens_kernel<R>, grid, dim3(RTUS_BLOCK), 0, s, k);
    return hipGetLastError();
}

hipError_t rtus_launch_tt_lens_f64(const rtus_lens& L, double a_lo, double a_hi, const double* xe, const double* ze,
                                   int n_e, const double* xf, const double* zf, int n_f, double* tt,
                                   double* alpha_out, hipStream_t s)
{
    return launch_lens<double>(L, a_lo, a_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, s);
}

hipError_t rtus_launch_tt_lens_f32(const rtus_lens& L, double a_lo, double a_hi, const float* xe, const float* ze,
                                   int n_e, const float* xf, const float* zf, int n_f, float* tt, float* alpha_out,
                                   hipStream_t s)
{
    return launch_lens<float>(L, a_lo, a_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, s);
}
