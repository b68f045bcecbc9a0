// rtus_fermat.hip — element x focal-point least-time (Fermat) travel times through horizontal
// layers, one (element, focal point) solve per lane.  NOT IN THE REFERENCE (parity unpinned):
// BASELINE configs 2/3/5 ask for planar interfaces the reference does not model.
//
// Formulation.  Snell's invariant p = sin(theta_i)/c_i.  With cm = fastest traversed speed,
// r_i = c_i/cm, k_i = 1 - r_i^2 and q = tan(theta) in the fastest layer, the horizontal reach is
//     X(q) = sum_i h_i r_i q / sqrt(1 + k_i q^2)          (increasing, concave, X(0) = 0)
//     X'(q) = sum_i h_i r_i (1 + k_i q^2)^(-3/2)
// and the travel time  T(q) = sqrt(1 + q^2) * sum_i (h_i / c_i) / sqrt(1 + k_i q^2).
// Newton on X(q) = |xf - xe| started from a LOWER bound of the root climbs monotonically
// (concavity: the tangent overestimates X), so no bracketing / bisection fallback is needed;
// q stays finite even at grazing incidence (unlike p -> 1/cm).  Convergence is tested per wave
// with a ballot so a wave leaves the loop as soon as all 64 lanes are done.
//
// Layout: tt[e][f], f fastest — each wave stores 512 contiguous bytes; xf/zf loads are coalesced,
// the element coordinates and all layer constants are wave-uniform (SGPRs).
#include "rtus_device.h"

struct LayerArgs {
    double z_if[RTUS_MAX_LAYERS];      // interface depths (n_if used)
    double c[RTUS_MAX_LAYERS + 1];     // speeds
    int n_if;
    const double* __restrict__ xe;
    const double* __restrict__ ze;
    const double* __restrict__ xf;
    const double* __restrict__ zf;
    double* __restrict__ tt;
    uint8_t* __restrict__ iters;
    int n_e, n_f;
};

// 1/sqrt(a) to ~1 ulp: hardware v_rsq_f64 seed + two Newton steps.
__device__ __forceinline__ double rsqrt_nr(double a)
{
    double y = __builtin_amdgcn_rsq(a);
    double h = 0.5 * a;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

template <int NL>   // NL = number of layers the medium has (n_if + 1)
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_tt_layers_kernel(LayerArgs a)
{
    const int f_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const int e = blockIdx.y;
    const bool live = f_raw < a.n_f;
    const int f = live ? f_raw : a.n_f - 1;
    const double xe = a.xe[e], ze = a.ze[e];
    const double xf = a.xf[f], zf = a.zf[f];
    const double X = fabs(xf - xe);

    // thickness of each layer along the path (0 for layers below the focal point)
    double h[NL], cm = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const double top = (i == 0) ? ze : fmax(a.z_if[i - 1], ze);
        const double bot = (i < NL - 1) ? fmin(a.z_if[i], zf) : zf;
        h[i] = fmax(bot - top, 0.0);
        cm = (h[i] > 0.0) ? fmax(cm, a.c[i]) : cm;
    }
    double hr[NL], kk[NL], hc[NL], s0 = 0.0, asym = 0.0, hm = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const double r = a.c[i] / cm;
        hr[i] = h[i] * r;
        kk[i] = fmax(1.0 - r * r, 0.0);
        hc[i] = h[i] / a.c[i];
        s0 += hr[i];
        // fastest layer(s): linear term h q; slower layers saturate at h r / sqrt(k)
        if (kk[i] == 0.0) hm += h[i]; else asym += hr[i] * rsqrt_nr(kk[i]);
    }
    // two lower bounds of the root: X <= X'(0) q, and X <= hm q + asym
    double q = fmax(X / s0, (X - asym) / hm);
    int it = 0;
    bool done = !(zf > ze);
    while (true) {
        double Xq = 0.0, dX = 0.0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const double w = rsqrt_nr(fma(kk[i], q * q, 1.0));
            const double hw = hr[i] * w;
            Xq = fma(hw, q, Xq);
            dX = fma(hw, w * w, dX);
        }
        const double dq = (X - Xq) / dX;
        if (!done) { q += dq; ++it; }
        done = done || !(fabs(dq) > 1e-11 * q) || it >= 60;
        if (__all(done)) break;
    }
    // T at the converged q with correctly-rounded sqrt / divide
    const double q2 = q * q, s1 = 1.0 + q2;
    double T = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) T += hc[i] * sqrt(s1 / fma(kk[i], q2, 1.0));
    if (!(zf > ze)) T = NAN;
    if (!live) return;
    const size_t o = (size_t)e * a.n_f + f;
    a.tt[o] = T;
    if (a.iters) a.iters[o] = (uint8_t)it;
}

hipError_t rtus_launch_tt_layers(const double* z_if, const double* c, int n_if, const double* xe,
                                 const double* ze, int n_e, const double* xf, const double* zf, int n_f,
                                 double* tt, uint8_t* iters, hipStream_t s)
{
    LayerArgs a;
    for (int i = 0; i < RTUS_MAX_LAYERS; ++i) a.z_if[i] = i < n_if ? z_if[i] : INFINITY;
    for (int i = 0; i <= RTUS_MAX_LAYERS; ++i) a.c[i] = i <= n_if ? c[i] : 1.0;
    a.n_if = n_if; a.xe = xe; a.ze = ze; a.xf = xf; a.zf = zf; a.tt = tt; a.iters = iters;
    a.n_e = n_e; a.n_f = n_f;
    const dim3 grid((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK, n_e), block(RTUS_BLOCK);
    switch (n_if + 1) {
#define RTUS_CASE(NL) case NL: hipLaunchKernelGGL(rtus_tt_layers_kernel<NL>, grid, block, 0, s, a); break;
        RTUS_CASE(1) RTUS_CASE(2) RTUS_CASE(3) RTUS_CASE(4) RTUS_CASE(5) RTUS_CASE(6) RTUS_CASE(7) RTUS_CASE(8)
        RTUS_CASE(9)
#undef RTUS_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
