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
// Cost model (measured, scripts/ubench_fp64.hip): an fp64 fma/mul/add wave-op costs ~2 ns of SIMD
// time, v_rsq_f64 / v_rcp_f64 ~6.8 ns (seed accuracy 5e-8), IEEE sqrt ~36 ns, IEEE divide ~25 ns.
// So the Newton iteration runs entirely on the fp32 pipe (v_fma_f32 ~1 ns, v_rsq_f32 / v_rcp_f32
// ~3.4 ns): it only has to bring q within 3e-4 of the root — Newton needs neither an exact Jacobian nor
// an exact residual for that — and the fp64 result is produced afterwards; no IEEE sqrt/divide anywhere.
//   * a lane stops when the step it would take is |dq| <= 3e-4 q; it does NOT take that step, so the
//     seeds of its last evaluation belong to its q and are refined (one cubic step, ~1 ulp) instead
//     of being recomputed;
//   * T at the exact root follows from the Fermat expansion in the residual dXr = X - X(q):
//     T = T(q) + p dXr + (1/2)(dp/dX) dXr^2,  p = dT/dX = sin(theta)/c  — error O(dXr^3): measured max 3.3e-17 s
//     (median 3e-21 s) against the long-double oracle on BASELINE config 2.
//
// Layout: tt[e][f], f fastest — each wave stores 512 contiguous bytes; xf/zf loads are coalesced,
// the element coordinates and all layer constants are wave-uniform (SGPRs).
#include "rtus_device.h"

struct LayerArgs {
    double z_if[RTUS_MAX_LAYERS];      // interface depths (n_if used, +inf beyond)
    double c[RTUS_MAX_LAYERS + 1];     // speeds
    double inv_c[RTUS_MAX_LAYERS + 1]; // 1/c (host-computed, correctly rounded)
    int n_if;
    const double* __restrict__ xe;
    const double* __restrict__ ze;
    const double* __restrict__ xf;
    const double* __restrict__ zf;
    double* __restrict__ tt;
    uint8_t* __restrict__ iters;
    int n_e, n_f;
    int eb;                            // elements per workgroup
};

// Refine an fp32-pipe seed y ~ a^(-1/2) (relative error d <= ~1.5e-7: v_rsq_f32 plus the rounding of its argument)
// with one Newton step in fp64: e = 1 - a y^2,  y <- y (1 + e/2)  -> error 1.5 d^2 <= 4e-14.  On the travel time
// that is < 2e-18 s — an order below the O(dX^3) remainder of the expansion it feeds — and fp64 instructions are what
// this kernel is short of (half the fp32 issue rate): the cubic step used before cost one more per square root.
__device__ __forceinline__ double rsqrt_refine(double a, double y)
{
    const double e = fma(-a * y, y, 1.0);
    return fma(y * e, 0.5, y);
}

// Seed for a^(-1/2), a >= 1: the fp32 pipe (cvt + v_rsq_f32 + cvt, ~3.6 ns) is cheaper than
// v_rsq_f64 (~6.8 ns) at about the same accuracy (1e-7).  (a > 3e38 would need q > 1e19: not a ray.)
__device__ __forceinline__ double rsqrt_seed(double a)
{
    return (double)__builtin_amdgcn_rsqf((float)a);
}
// Seed for 1/a through the fp32 pipe (1e-7); a must be within float range (sums of h r w^3: it is).
__device__ __forceinline__ double rcp_seed(double a)
{
    return (double)__builtin_amdgcn_rcpf((float)a);
}

// v_readlane of a double held one-per-lane (the lane index is wave-uniform: an SGPR)
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l),
                            __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float readlane_f32(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// A workgroup = 256 focal points x `eb` consecutive elements (loop).  Besides re-using the layer
// set-up while ze repeats, the loop gives each lane a CONTINUATION PREDICTOR: the signed solution
// qs = sign(xf - xe) q is a smooth function of the element position, so the three previous
// solutions extrapolate the next one to ~1e-3..1e-4 and Newton needs ~1.5 evaluations instead of ~4.5.
// The extrapolation weights depend on the element positions only: lane l works them out once for element
// e0 + l (Lagrange form) and the loop fetches them with v_readlane — three fp32 FMAs per solve.
// The predictor is only a guess: every iterate is clamped to the rigorous lower bound of the root,
// from which Newton is monotone, so convergence never depends on the elements being evenly spaced.
template <int NL, bool ITERS>   // NL = number of layers the medium has (n_if + 1)
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_tt_layers_kernel(LayerArgs a)
{
    const int f_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = f_raw < a.n_f;
    const int f = live ? f_raw : a.n_f - 1;
    const double xf = a.xf[f], zf = a.zf[f];
    const int e0 = blockIdx.y * a.eb;
    const int e1 = min(e0 + a.eb, a.n_e);
    // the workgroup's element coordinates: ONE vector load per wave (lane l holds element e0+l, eb <= 64),
    // then v_readlane per loop trip — no memory latency inside the element loop.
    const int lane = threadIdx.x & 63;
    const int el = min(e0 + lane, a.n_e - 1);
    const double xe_v = a.xe[el], ze_v = a.ze[el];
    // predictor set-up for element e0 + lane: how many predecessors inside this block share its depth
    // (the history restarts when ze changes), and the extrapolation weights on their solutions.
    float w1_v = 0.0f, w2_v = 0.0f, w3_v = 0.0f;
    int info_v;                                            // bits 0-1: 0 no guess, 1 proportional, 2 weights; bit 2: new ze
    {
        const int b1 = max(el - 1, e0), b2 = max(el - 2, e0), b3 = max(el - 3, e0);
        const double x1 = a.xe[b1], x2 = a.xe[b2], x3 = a.xe[b3];
        const double z1 = a.ze[b1], z2 = a.ze[b2], z3 = a.ze[b3];     // all loads issued together: one round trip
        // '&', not '&&': keeps the compiler from sinking a load behind the previous comparison
        const bool s1 = (lane >= 1) & (z1 == ze_v), s2 = s1 & (lane >= 2) & (z2 == ze_v), s3 = s2 & (lane >= 3) & (z3 == ze_v);
        const int hist = s3 ? 3 : (s2 ? 2 : (s1 ? 1 : 0));
        // branch-free on purpose (selects): a branch here lets the compiler sink the x loads behind it,
        // which costs the prologue a second memory round trip
        // three differences in fp64, the other three from them on the fp32 pipe (t2 = xe - x2 = t1 + d12, ...)
        const float d12 = (float)(x1 - x2), d13 = (float)(x1 - x3), t1 = (float)(xe_v - x1);
        const float d23 = d13 - d12, t2 = t1 + d12, t3 = t1 + d13;
        const bool lin = (hist >= 2) & (x1 != x2);                       // else: duplicate positions, no slope
        const bool quad = lin & (hist >= 3) & (x3 != x1) & (x3 != x2);   // quadratic through the last three solutions
        const float r12 = __builtin_amdgcn_rcpf(d12);
        const float q1 = t2 * t3 * __builtin_amdgcn_rcpf(d12 * d13);
        const float q2 = -t1 * t3 * __builtin_amdgcn_rcpf(d12 * d23);
        const float q3 = t1 * t2 * __builtin_amdgcn_rcpf(d13 * d23);
        w1_v = quad ? q1 : (lin ? t2 * r12 : 0.0f);                      // linear through the last two otherwise
        w2_v = quad ? q2 : (lin ? -t1 * r12 : 0.0f);
        w3_v = quad ? q3 : 0.0f;
        const int mode = lin ? 2 : (hist >= 1 ? 1 : 0);
        info_v = mode | (hist == 0 ? 4 : 0);
    }

    // Per-lane layer table, PERMUTED so that slot 0 is the lane's fastest traversed layer: there k = 0 and
    // w = (1 + k q^2)^(-1/2) = 1 exactly, so slot 0 needs no rsqrt anywhere (hr0, hc0 enter the sums directly).
    double hr0 = 0.0, hc0 = 0.0, hr[NL], kk[NL], hc[NL];             // slots 1 .. NL-1 of the arrays are used
    double inv_cm = 0.0;
    float hr0f = 0.0f, hrf[NL], kkf[NL], rs0f = 0.0f, rhmf = 0.0f, asymf = 0.0f;   // fp32 copies for the Newton loop
    float qs1 = 0.0f, qs2 = 0.0f, qs3 = 0.0f;             // signed solutions of the three previous elements
    bool valid = false;
    float rS3 = 0.0f;                                      // 1 / X'(q) of the latest Newton evaluation (kept across elements)
    float tau = INFINITY;                                  // relative step below which a lane stops iterating
    for (int e = e0; e < e1; ++e) {                         // wave-uniform loop
        const int li = e - e0;
        const int info = __builtin_amdgcn_readlane(info_v, li);
        const double xe = readlane_f64(xe_v, li);
        if (info & 4) {                                     // wave-uniform: redo the layer set-up only when ze changes
            const double ze = readlane_f64(ze_v, li);
            valid = zf > ze;
            tau = valid ? 3e-4f : INFINITY;
            qs1 = qs2 = qs3 = 0.0f;                         // a lane may carry NaN history from a depth at which its target was
                                                            // not below the element; the linear predictor multiplies qs3 by 0
            // thickness of each layer along the path (0 for layers the path does not enter); fastest speed
            double cm = 0.0, h[NL], hr_l[NL], hc_l[NL], kk_l[NL];   // _l: in layer order
            inv_cm = 0.0;
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const double top = (i == 0) ? ze : fmax(a.z_if[i - 1], ze);
                const double bot = (i < NL - 1) ? fmin(a.z_if[i], zf) : zf;
                h[i] = fmax(bot - top, 0.0);
                const bool faster = h[i] > 0.0 && a.c[i] > cm;
                cm = faster ? a.c[i] : cm;
                inv_cm = faster ? a.inv_c[i] : inv_cm;
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                const bool fastest = a.c[i] == cm;          // exact: cm IS one of the c[i]
                const double r = fastest ? 1.0 : a.c[i] * inv_cm;
                hr_l[i] = h[i] * r;
                hc_l[i] = h[i] * a.inv_c[i];
                kk_l[i] = fastest ? 0.0 : fmax(fma(-r, r, 1.0), 0.0);
            }
            // slot 0 <-> the first layer whose speed is cm (per lane: which layers a path crosses depends on zf)
            int jf = 0;
#pragma unroll
            for (int i = NL - 1; i >= 1; --i) jf = (a.c[i] == cm) ? i : jf;
            jf = (a.c[0] == cm) ? 0 : jf;
            hr0 = hr_l[0]; hc0 = hc_l[0];
#pragma unroll
            for (int i = 1; i < NL; ++i) {
                const bool sw = i == jf;
                hr0 = sw ? hr_l[i] : hr0;        hc0 = sw ? hc_l[i] : hc0;
                hr[i] = sw ? hr_l[0] : hr_l[i];  hc[i] = sw ? hc_l[0] : hc_l[i];  kk[i] = sw ? kk_l[0] : kk_l[i];
            }
#pragma unroll
            for (int i = 1; i < NL; ++i) { hrf[i] = (float)hr[i]; kkf[i] = (float)kk[i]; }
            hr0f = (float)hr0;
            // the two lower bounds of a cold start, formed on the fp32 pipe (they are only a starting guess):
            // X <= s0 q with s0 = X'(0), and X <= hm q + asym — the fastest layer(s) (k = 0) contribute the linear term
            // h q, every slower layer saturates at h r / sqrt(k).  Shaved so that fp32 rounding keeps them lower bounds.
            float s0f = hr0f, hmf = hr0f, asf = 0.0f;
#pragma unroll
            for (int i = 1; i < NL; ++i) {
                const bool lin = kkf[i] == 0.0f;
                s0f += hrf[i];
                hmf += lin ? hrf[i] : 0.0f;
                asf += lin ? 0.0f : hrf[i] * __builtin_amdgcn_rsqf(kkf[i]);
            }
            rs0f = __builtin_amdgcn_rcpf(s0f) * (1.0f - 4e-6f);
            rhmf = __builtin_amdgcn_rcpf(hmf) * (1.0f - 4e-6f);
            asymf = asf * (1.0f + 4e-6f);
            hc0 = valid ? hc0 : NAN;                        // target not below the element: T = NaN falls out of the sums
        }
        const double dxs = xf - xe;
        const double X = fabs(dxs);
        // ---- Newton iteration in fp32 -----------------------------------------------------------
        // The loop only has to bring q within ~1e-4 of the root (the fp64 expansion below removes the
        // rest to third order), so it runs on the fp32 pipe: half the issue cost of fp64 and native
        // v_rsq_f32 / v_rcp_f32.  X(q) = q S1 is evaluated to ~1e-7 relative: noise three orders below
        // the stopping threshold.  Lanes whose target is not below the element (!valid) carry garbage
        // through the arithmetic (never a step: `big` is masked) and get NaN at the store.
        const float Xf = (float)X;
        // two lower bounds of the root: X <= X'(0) q, and X <= hm q + asym (shaved so rounding keeps them lower)
        const float lb = fmaxf(fmaxf(Xf * rs0f, (Xf - asymf) * rhmf), 0.0f);
        float q = lb;
        if ((info & 3) == 2) {                              // wave-uniform
            const float w1 = readlane_f32(w1_v, li), w2 = readlane_f32(w2_v, li), w3 = readlane_f32(w3_v, li);
            q = fmaxf(fabsf(fmaf(w1, qs1, fmaf(w2, qs2, w3 * qs3))), lb);
        } else if ((info & 3) == 1) {                       // one solution so far: first-order Taylor step from it —
            const double xe1 = readlane_f64(xe_v, li - 1);  // d(qs)/d(xe) = -1 / X'(q), which the last solve left in rS3
            q = fmaxf(fabsf(fmaf(-rS3, (float)(xe - xe1), qs1)), lb);
        }
        float y[NL], dq = 0.0f;                             // dq: the (small, untaken) Newton step of the last trip
        int it = 0;
        for (int trip = 0; trip < 64; ++trip) {             // wave-uniform trip count, ballot exit
            const float q2 = q * q;
            float S1 = hr0f, S3 = hr0f;                     // slot 0: k = 0, y = 1
#pragma unroll
            for (int i = 1; i < NL; ++i) {
                y[i] = __builtin_amdgcn_rsqf(fmaf(kkf[i], q2, 1.0f));
                const float hw = hrf[i] * y[i];
                S1 += hw;
                S3 = fmaf(hw, y[i] * y[i], S3);
            }
            rS3 = __builtin_amdgcn_rcpf(S3);
            dq = fmaf(-S1, q, Xf) * rS3;
            // A lane is done when the step it WOULD take is small; it does not take it, so y[] stays the
            // y of its q (and a done lane re-derives the same small dq on later trips: no state needed).
            const bool big = fabsf(dq) > tau * q;           // tau = +inf on !valid lanes: never a step
            if (!__builtin_amdgcn_ballot_w64(big)) break;
            q = big ? fmaxf(q + dq, lb) : q;
            if (ITERS) it += big ? 1 : 0;
        }
        // ---- fp64: accurate T at q + the Fermat expansion in the residual dXr = X - X(q) ----------
        // T(root) = T(q) + p dXr + (1/2) (dp/dX) dXr^2 + O(dXr^3),
        // p = sin(theta)/cm = q u / cm,  dp/dX = u^3 / (cm X'(q)),  u = 1/sqrt(1+q^2).
        const double qd = (double)q;
        const double q2 = qd * qd;
        const double a1 = 1.0 + q2;
        const float us = __builtin_amdgcn_rsqf(fmaf(q, q, 1.0f));   // u to 1e-7 on the fp32 pipe: seed + 2nd-order term
        const double u = rsqrt_refine(a1, (double)us);
        double A1 = hr0, ST = hc0;                          // slot 0: w = 1
#pragma unroll
        for (int i = 1; i < NL; ++i) {
            const double w = rsqrt_refine(fma(kk[i], q2, 1.0), (double)y[i]);
            A1 = fma(hr[i], w, A1);
            ST = fma(hc[i], w, ST);
        }
        const double dXr = fma(-A1, qd, X);
        // T = T(q) + dXr (u/cm) (q + (u^2 / (2 X'(q))) dXr) = u (a1 ST + (dXr / cm) (q + s2 dXr)): the coefficient s2 only
        // scales the 2nd-order term (relative size (dXr/X)^2 ~ 1e-7), so it is formed on the fp32 pipe from the seeds.
        const float s2 = (0.5f * us) * (us * rS3);
        const double T = u * fma(inv_cm, dXr * fma((double)s2, dXr, qd), a1 * ST);
        if (live) {
            const size_t o = (size_t)e * a.n_f + f;
            a.tt[o] = T;
            if (ITERS) a.iters[o] = (uint8_t)it;
        }
        // history for the predictor (fp32): the root itself, q + (untaken step), with the sign of xf - xe
        const float qroot = q + dq;
        qs3 = qs2; qs2 = qs1;
        qs1 = __int_as_float((__float_as_int(qroot) & 0x7fffffff) | (__double2hiint(dxs) & 0x80000000));
    }
}

hipError_t rtus_launch_tt_layers(const double* z_if, const double* c, int n_if, const double* xe,
                                 const double* ze, int n_e, const double* xf, const double* zf, int n_f,
                                 double* tt, uint8_t* iters, hipStream_t s)
{
    LayerArgs a;
    for (int i = 0; i < RTUS_MAX_LAYERS; ++i) a.z_if[i] = i < n_if ? z_if[i] : INFINITY;
    for (int i = 0; i <= RTUS_MAX_LAYERS; ++i) { a.c[i] = i <= n_if ? c[i] : 1.0; a.inv_c[i] = 1.0 / a.c[i]; }
    a.n_if = n_if; a.xe = xe; a.ze = ze; a.xf = xf; a.zf = zf; a.tt = tt; a.iters = iters;
    a.n_e = n_e; a.n_f = n_f;
    // elements per workgroup: as many as possible (predictor + set-up reuse) while keeping >= ~4 waves per SIMD
    const long long wave_solves = (long long)((n_f + 63) / 64) * n_e;
    int eb = (int)(wave_solves / (1024LL * 4));
    eb = eb < 1 ? 1 : (eb > 32 ? 32 : eb);
    while ((n_e + eb - 1) / eb > 65535 && eb < 64) ++eb;   // grid.y limit (eb <= 64: one lane per element)
    a.eb = eb;
    const dim3 grid((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK, (n_e + eb - 1) / eb), block(RTUS_BLOCK);
    switch (n_if + 1) {
#define RTUS_CASE(NL) case NL: if (iters) hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, true>), grid, block, 0, s, a); \
                               else hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, false>), grid, block, 0, s, a); break;
        RTUS_CASE(1) RTUS_CASE(2) RTUS_CASE(3) RTUS_CASE(4) RTUS_CASE(5) RTUS_CASE(6) RTUS_CASE(7) RTUS_CASE(8)
        RTUS_CASE(9)
#undef RTUS_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
