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
// Cost model (measured on gfx950, scripts/ubench_issue*.hip -> profiles/r02_ubench_issue.txt; cycles per
// wave-instruction per SIMD at 8 waves/SIMD): fp32 fma / mul / add / mov with VGPR or inline-constant operands 2.3;
// the same with an SGPR operand 4.2; every fp64 op, v_cvt, v_max/v_min, v_cmp, v_readlane, v_bfi 4.2; v_rsq_f32 /
// v_rcp_f32 8.2; v_pk_fma_f32 4.2 (packed fp32 buys no issue slots on this part); IEEE fp64 sqrt ~85, divide ~60.
// So the Newton iteration runs entirely on the fp32 pipe with per-lane (VGPR) tables: it only has to bring q within
// 3e-4 of the root — Newton needs neither an exact Jacobian nor an exact residual for that — and the fp64 result is
// produced afterwards; no IEEE sqrt/divide anywhere.
//   * a lane stops when the step it would take is |dq| <= 3e-4 q; it does NOT take that step, so the
//     seeds of its last evaluation belong to its q and are refined (one Newton step in fp64, 1.5 d^2) instead
//     of being recomputed;
//   * T at the exact root follows from the Fermat expansion in the residual dXr = X - X(q):
//     T = T(q) + p dXr + (1/2)(dp/dX) dXr^2,  p = dT/dX = sin(theta)/c  — error O(dXr^3): measured max 3.4e-18 s
//     (median 3e-20 s) against the long-double oracle on BASELINE config 3.
//
// Layout: tt[e][f], f fastest — each wave stores 512 contiguous bytes; xf/zf loads are coalesced; the
// per-element data of a workgroup (coordinates, extrapolation weights) sits in LDS and is broadcast to the lanes.
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
    int eb;                            // elements per workgroup (<= 64): a function of the WHOLE table's size, see rtus_table_rows_per_block
    const int* __restrict__ row_of;    // PERM kernels: output row of element i (the elements arrive sorted by (ze, xe), see rtus_rank_kernel)
    int row0;                          // index of xe[0] in the whole table when this launch solves a block of its rows (else 0):
                                       // workgroups cover rows [k eb, (k + 1) eb) of the WHOLE table, so a row is solved with the
                                       // same predecessors — and comes out with the same bits — whether the table is solved in one
                                       // launch or in row shards
    // batched entry (rtus_tt_layers_batch): problem b = blockIdx.z uses elements / targets / output shifted by these
    long long e_stride, f_stride, t_stride;   // in elements of the respective arrays (0: shared by all problems)
    int gx, gy, n_items;               // work items: gx columns of 256 targets x gy blocks of rows (x n_batch problems) = n_items
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Refine an fp32-pipe seed y ~ a^(-1/2) (relative error d <= ~1.5e-7: v_rsq_f32 plus the rounding of its argument)
// with one Newton step in fp64: e = 1 - a y^2,  y <- y (1 + e/2)  -> error 1.5 d^2 <= 4e-14.  On the travel time
// that is < 2e-18 s — an order below the O(dX^3) remainder of the expansion it feeds.
__device__ __forceinline__ double rsqrt_refine(double a, double y)
{
    const double e = fma(-a * y, y, 1.0);
    return fma(y * e, 0.5, y);
}

#ifdef RTUS_EXP_COUNT
static __device__ unsigned long long planar_dbg[8];
extern "C" int rtus_dbg_read_planar(unsigned long long* out)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(planar_dbg), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(planar_dbg), z, sizeof(z));
    return 0;
}
#endif

// One element's record in LDS: every lane of the workgroup reads the same address (broadcast, no VALU slot —
// a v_readlane costs 4.2 cycles and leaves its value in an SGPR, which halves the rate of every fp32 op reading it).
struct __attribute__((aligned(16))) ElemRec {
    float w1, w2, w3, w4;   // extrapolation weights on the signed solutions of the four previous elements
    double xe;
    int info;               // bits 0-2: previous elements usable as history (0..4); bit 3: depth differs from the previous
                            // element's (layer set-up needed); bit 4: uniform pitch from four elements behind to three ahead (HOLD);
                            // bits 8..: length of the run of 4-history elements starting here
    int row;                // PERM: the element's output row
};

typedef const __attribute__((address_space(3))) ElemRec* RecPtr;   // a record where it lives: LDS (ds_read at a register + immediate offset)

// Relative size of the Newton step at which a lane stops (and does not take it).  A HELD solve of the tau-p tier (HOLD = 2) forms its
// second-order term from quantities good to ~0.1 % (3.6 % at worst): its error is that fraction of (tau^2 / 8) T, so it stops at a third.
#define RTUS_PLANAR_TAU 3e-4f

// Per-lane solver state: the layer table of this lane's target, PERMUTED so that slot 0 is the lane's fastest
// traversed layer (there k = 0 and w = (1 + k q^2)^(-1/2) = 1 exactly: slot 0 needs no rsqrt anywhere).
template <int NL>
struct Lane {
    double hr0, hc0, hr[NL], kk[NL], hc[NL], inv_cm;       // slots 1 .. NL-1 of the arrays are used
    float hr0f, hrf[NL], kkf[NL], rs0f, rhmf, asymf;       // fp32 copies for the Newton loop, cold-start bounds
    float hic;                                             // 1 / (2 cm) (tau-p tail: scales its second-order term)
    float G, dG;                                           // tau-p tail: u^3 / (2 cm X'(q)) of the latest solve + its change per element (see HOLD)
    float tau;                                             // relative step below which a lane stops (+inf: target not below the element)
    float rS3;                                             // 1 / X'(q) of the latest evaluation
};

template <int NL>
__device__ __forceinline__ void layer_setup(const LayerArgs& a, double ze, const double* __restrict__ zf_p, unsigned f8, Lane<NL>& L)
{
    // The target's depth is needed here and nowhere else, and the set-up runs once per depth of the aperture: it is LOADED here (an
    // L2 hit), through an index the compiler cannot see through.  Left to itself it keeps zf — and the parts of the set-up that
    // depend on the target alone, hoisted out of the element loop: four more doubles — in SCRATCH across the loop (no register is
    // free for them): 6 % more HBM writes per launch than the table itself (rocprofv3 WRITE_SIZE), now none.
    asm volatile("" : "+v"(f8));
    const double zf = *(const double*)((const char*)zf_p + f8);   // (the byte offset the stores use anyway: no second index register)
    const bool valid = zf > ze;
    L.tau = valid ? RTUS_PLANAR_TAU : INFINITY;
    // thickness of each layer along the path (0 for layers the path does not enter); fastest speed
    double cm = 0.0, h[NL], hr_l[NL], hc_l[NL], kk_l[NL];   // _l: in layer order
    L.inv_cm = 0.0;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        // (selects, not fmax / fmin: those quiet their operands first — a v_max_f64 of each interface depth with itself, loop-invariant,
        // hoisted and then parked in scratch like zf above; the interface depths are finite or +inf, never NaN)
        const double top = (i == 0) ? ze : (a.z_if[i - 1] > ze ? a.z_if[i - 1] : ze);
        const double bot = (i < NL - 1) ? (a.z_if[i] < zf ? a.z_if[i] : zf) : zf;
        h[i] = fmax(bot - top, 0.0);
        const bool faster = h[i] > 0.0 && a.c[i] > cm;
        cm = faster ? a.c[i] : cm;
        L.inv_cm = faster ? a.inv_c[i] : L.inv_cm;
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
        const bool fastest = a.c[i] == cm;          // exact: cm IS one of the c[i]
        const double r = fastest ? 1.0 : a.c[i] * L.inv_cm;
        hr_l[i] = h[i] * r;
        hc_l[i] = h[i] * a.inv_c[i];
        kk_l[i] = fastest ? 0.0 : fmax(fma(-r, r, 1.0), 0.0);
    }
    // slot 0 <-> the first layer whose speed is cm (per lane: which layers a path crosses depends on zf)
    int jf = 0;
#pragma unroll
    for (int i = NL - 1; i >= 1; --i) jf = (a.c[i] == cm) ? i : jf;
    jf = (a.c[0] == cm) ? 0 : jf;
    L.hr0 = hr_l[0]; L.hc0 = hc_l[0];
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        const bool sw = i == jf;
        L.hr0 = sw ? hr_l[i] : L.hr0;        L.hc0 = sw ? hc_l[i] : L.hc0;
        L.hr[i] = sw ? hr_l[0] : hr_l[i];    L.hc[i] = sw ? hc_l[0] : hc_l[i];    L.kk[i] = sw ? kk_l[0] : kk_l[i];
    }
#pragma unroll
    for (int i = 1; i < NL; ++i) { L.hrf[i] = (float)L.hr[i]; L.kkf[i] = (float)L.kk[i]; }
    L.hr0f = (float)L.hr0;
    // the two lower bounds of a cold start, formed on the fp32 pipe (they are only a starting guess):
    // X <= s0 q with s0 = X'(0), and X <= hm q + asym — the fastest layer(s) (k = 0) contribute the linear term
    // h q, every slower layer saturates at h r / sqrt(k).  Shaved so that fp32 rounding keeps them lower bounds.
    float s0f = L.hr0f, hmf = L.hr0f, asf = 0.0f;
#pragma unroll
    for (int i = 1; i < NL; ++i) {
        const bool lin = L.kkf[i] == 0.0f;
        s0f += L.hrf[i];
        hmf += lin ? L.hrf[i] : 0.0f;
        asf += lin ? 0.0f : L.hrf[i] * __builtin_amdgcn_rsqf(L.kkf[i]);
    }
    L.rs0f = __builtin_amdgcn_rcpf(s0f) * (1.0f - 4e-6f);
    L.rhmf = __builtin_amdgcn_rcpf(hmf) * (1.0f - 4e-6f);
    L.asymf = asf * (1.0f + 4e-6f);
    L.hic = 0.5f * (float)L.inv_cm;
    L.G = L.dG = 0.0f;
    L.hc0 = valid ? L.hc0 : NAN;                    // target not below the element: T = NaN falls out of the sums
    L.rS3 = 0.0f;
}

// One (element, target) solve.  h1 .. h4: signed solutions qs = sign(xf - xe) q of the four previous elements
// (h1 the latest); the new one is returned.  FAST: the element has four usable predecessors — the cubic extrapolation
// is within 3e-4 of the root for all but ~1e-4 of the solves, so the lower-bound clamp is only formed when a lane
// of the wave asks for a second Newton evaluation.
// TAUP: the travel time from the tau-p form T = p X + sum (h_i / c_i) cos(theta_i) instead of T(q) + its Fermat expansion — see
// the tail below.
template <int NL, bool ITERS, bool FAST, bool TAUP, int HOLD = 0>
__device__ __forceinline__ float solve_elem(uint8_t* __restrict__ iters, Lane<NL>& L, RecPtr R, int hist, double xf,
                                            float h1, float h2, float h3, float h4, bool live, size_t row, unsigned f8,
                                            __amdgpu_buffer_rsrc_t rs, unsigned soff, float hold_age_rcp = 0.0f, bool* hold_ok = nullptr)
{
    static_assert(HOLD == 0 || (FAST && TAUP), "HOLD is a mode of the tau-p tier's four-history runs");
    const double dxs = xf - R->xe;
    const double X = fabs(dxs);
    // ---- Newton iteration in fp32 -----------------------------------------------------------
    // The loop only has to bring q within tau of the root (the fp64 expansion below removes the
    // rest to third order), so it runs on the fp32 pipe: half the issue cost of fp64 and native
    // v_rsq_f32 / v_rcp_f32.  X(q) = q S1 is evaluated to ~1e-7 relative: noise two orders below
    // the stopping threshold.  Lanes whose target is not below the element carry garbage
    // through the arithmetic (never a step: tau = +inf) and get NaN at the store.
    // FAST (a four-history run) works on the SIGNED problem X(qs) = xf - xe — X(q) = q S1(q^2) is odd, the predictor's history is
    // signed anyway — so the common path has no |.| to materialise and no sign to put back (a v_and and a v_bfi per solve); the
    // rare second-evaluation branch stays signed too and clamps to the signed lower bound (one v_med3_f32).
    float Xf = FAST ? (float)dxs : (float)X;
    double Xt = FAST ? dxs : X;                             // the reach that pairs with q in the tail (same sign as q)
    float y[NL], dq = 0.0f, dXf = 0.0f;                     // dq: the (small, untaken) Newton step of the last evaluation, dXf its residual
    int it = 0;
    // one evaluation of X(q), X'(q) on the fp32 pipe -> Newton step dq; y[] = the rsqrt seeds of this q
    auto eval = [&](float qq, bool second = false) {
        const float q2 = qq * qq;
        float S1 = L.hr0f, S3 = L.hr0f;                     // slot 0: k = 0, y = 1
        const bool held = HOLD == 2 && !second;             // (compile-time where it matters: `second` is a literal at every call)
#pragma unroll
        for (int i = 1; i < NL; ++i) {
            y[i] = __builtin_amdgcn_rsqf(fmaf(L.kkf[i], q2, 1.0f));
            if (held) S1 = fmaf(L.hrf[i], y[i], S1);
            else {
                const float hw = L.hrf[i] * y[i];
                S1 += hw;
                S3 = fmaf(hw, y[i] * y[i], S3);
            }
        }
        // 1 / X'(q): v_rcp_f32 — or, HOLD = 2, the reciprocal of the group's first element as it is (see the kernel).  (Round 3 took
        // ONE Newton step from the previous element's reciprocal instead, error = the change of X' squared: fine on a fine regular
        // pitch, but on a coarse or random one — X' moving by tens of per cent per element, the predictor missing by ~tau — it put
        // 2.8e-10 relative on a table: scripts/fuzz_layers.py --taup with tables large enough for four-history runs found it.)
        if (!held) L.rS3 = __builtin_amdgcn_rcpf(S3);
        dXf = fmaf(-S1, qq, Xf);
        dq = dXf * L.rS3;
    };
    // two lower bounds of the root: X <= X'(0) q, and X <= hm q + asym (shaved so rounding keeps them lower)
    auto lower_bound = [&]() { return fmaxf(fmaxf(fabsf(Xf) * L.rs0f, (fabsf(Xf) - L.asymf) * L.rhmf), 0.0f); };   // (FAST: Xf is signed)
    // Newton from q, every iterate clamped to lb.  A lane is done when the step it WOULD take is small; it does not
    // take it, so y[] stays the y of its q (and a done lane re-derives the same small dq on later trips: no state needed).
    auto newton = [&](float& q, float lb) {
        for (int trip = 0; trip < 64; ++trip) {             // wave-uniform trip count, ballot exit
            eval(q, true);
            const bool big = fabsf(dq) > L.tau * q;         // tau = +inf on lanes without a path: never a step
            if (!__builtin_amdgcn_ballot_w64(big)) break;
            q = big ? fmaxf(q + dq, lb) : q;
            if (ITERS) it += big ? 1 : 0;
        }
    };
    float q;
    if (FAST) {
        // the cubic extrapolation is nearly always within tau of the root: one evaluation, no clamp, no select
        q = fmaf(R->w1, h1, fmaf(R->w2, h2, fmaf(R->w3, h3, R->w4 * h4)));
        eval(q);
#ifdef RTUS_EXP_COUNT   // experiment builds only (scripts/exp_planar_miss.py): how far the cubic predictor lands from the root
        if (live && L.tau < 1.0f) {
            const float rel = fabsf(dq) / fabsf(q);
            atomicAdd(&planar_dbg[0], 1ull);
            if (rel > 3e-5f) atomicAdd(&planar_dbg[1], 1ull);
            atomicMax((unsigned long long*)&planar_dbg[7], (unsigned long long)__float_as_uint(rel));
        }
        {   // waves with any lane beyond 3e-5 / 5e-5 / 1e-4
            const float relw = (live && L.tau < 1.0f) ? fabsf(dq) / fabsf(q) : 0.0f;
            const bool l0 = (threadIdx.x & 63) == 0;
            if (__ballot(relw > 3e-5f) && l0) atomicAdd(&planar_dbg[2], 1ull);
            if (__ballot(relw > 5e-5f) && l0) atomicAdd(&planar_dbg[3], 1ull);
            if (__ballot(relw > 1e-4f) && l0) atomicAdd(&planar_dbg[4], 1ull);
        }
#endif
        // (a prediction of the wrong sign is never "small": the step it asks for is larger than itself)
        const float tau_f = HOLD == 2 ? L.tau * (1.0f / 3.0f) : L.tau;   // (loop-invariant: one register, no instruction per solve)
        const bool big = fabsf(dq) > tau_f * fabsf(q);
        if (__builtin_amdgcn_ballot_w64(big)) {             // wave-uniform: some lane wants a second evaluation
            asm volatile("" : "+v"(q));                     // keeps this block a branch (nothing of it is speculated)
            // Newton on the signed problem, clamped to the signed lower bound: max(., lb) for targets to the right of the element,
            // min(., -lb) to the left = the median with that side's infinity (one v_med3_f32 either way)
            const float sgb = __int_as_float(__double2hiint(dxs));
            const float lbs = __builtin_copysignf(lower_bound(), sgb), sinf = __builtin_copysignf(INFINITY, sgb);
            q = big ? __builtin_amdgcn_fmed3f(q + dq, lbs, sinf) : q;
            if (ITERS) it += big ? 1 : 0;
            for (int trip = 0; trip < 64; ++trip) {         // wave-uniform trip count, ballot exit
                eval(q, true);
                const bool bigger = fabsf(dq) > L.tau * fabsf(q);
                if (!__builtin_amdgcn_ballot_w64(bigger)) break;
                q = bigger ? __builtin_amdgcn_fmed3f(q + dq, lbs, sinf) : q;
                if (ITERS) it += bigger ? 1 : 0;
            }
        }
    } else {
        const float lb = lower_bound();
        q = lb;
        if (hist > 0)                                       // wave-uniform; unused weights are 0
            q = fmaxf(fabsf(fmaf(R->w1, h1, fmaf(R->w2, h2, fmaf(R->w3, h3, R->w4 * h4)))), lb);
        newton(q, lb);
    }
    // ---- fp64: accurate T at q + the Fermat expansion in the residual dXr = X - X(q) ----------
    // T(root) = T(q) + p dXr + (1/2) (dp/dX) dXr^2 + O(dXr^3),
    // p = sin(theta)/cm = q u / cm,  dp/dX = u^3 / (cm X'(q)),  u = 1/sqrt(1+q^2).
    const double qd = (double)q;
    const double q2 = qd * qd;
    const double a1 = 1.0 + q2;
    const float us = __builtin_amdgcn_rsqf(fmaf(q, q, 1.0f));   // u to 1e-7 on the fp32 pipe: seed + 2nd-order term
    const double u = rsqrt_refine(a1, (double)us);
    double T;
    if (!TAUP) {
        double A1 = L.hr0, ST = L.hc0;                      // slot 0: w = 1
#pragma unroll
        for (int i = 1; i < NL; ++i) {
            const double w = rsqrt_refine(fma(L.kk[i], q2, 1.0), (double)y[i]);
            A1 = fma(L.hr[i], w, A1);
            ST = fma(L.hc[i], w, ST);
        }
        const double dXr = fma(-A1, qd, Xt);
        // T = T(q) + dXr (u/cm) (q + (u^2 / (2 X'(q))) dXr) = u (a1 ST + (dXr / cm) (q + s2 dXr)): the coefficient s2 only
        // scales the 2nd-order term (relative size (dXr/X)^2 ~ 1e-7), so it is formed on the fp32 pipe from the seeds.
        const float s2 = (0.5f * us) * (us * L.rS3);
        T = u * fma(L.inv_cm, dXr * fma((double)s2, dXr, qd), a1 * ST);
    } else {
        // tau-p form (the faster accuracy tier, RTUS_TT_TAUP_TAIL): with p = sin(theta)/c = q u / cm and cos(theta_i) =
        // u sqrt(1 + k_i q^2), F(p) = p X + sum (h_i/c_i) cos(theta_i) is STATIONARY in p at the ray (dF/dp = X - X(p)): evaluated
        // at the Newton iterate its first-order error vanishes by itself — no second weighted sum (X(q) in fp64), no fp64
        // residual — and T = F + (1/2)(dp/dX) dXr^2 with the residual the fp32 evaluation already has:
        // (1/2)(dp/dX) dXr^2 = (1/2) u^3 / (cm X') dXr^2 = (1/2) u^3 (dq dXr) / cm.  The fp32 residual carries ~2e-7 X of
        // rounding, so that term — and T — is off by up to (dq/q) 2e-7 T: 6e-11 T at the stopping threshold, ~3e-14 T typically
        // (the accurate tier: 1e-13 T at most).  Four fp64 instructions fewer per solve (24 against 28 with three layers).
        double ST = L.hc0;                                  // slot 0: cos(theta_0) / u = 1
#pragma unroll
        for (int i = 1; i < NL; ++i) {
            const double d = fma(L.kk[i], q2, 1.0), yi = (double)y[i];
            const double g = d * yi;                        // sqrt(d) from the seed ...
            const double t = fma(-g, g, d) * yi;            // ... and one Newton step: sqrt(d) = g + t / 2 (error ~1e-14)
            ST = fma(L.hc[i], fma(t, 0.5, g), ST);
        }
        // the second-order term = G dXf^2 with G = u^3 / (2 cm X'(q)); HOLD = 2: the G of the group's first element (see the kernel)
        float G;
        if (HOLD == 2) G = L.G = L.G + L.dG;                // one element further along the line through the last two G formed
        else {
            G = ((L.hic * us) * (us * us)) * L.rS3;
            if (HOLD == 1) {
                L.dG = (G - L.G) * hold_age_rcp;                // hold_age_rcp here: 1 / (elements since L.G was formed)
                // G is a smooth function of q (|d ln G / dq| <= 3, |G'' / G| <= ~20 along the aperture): a step of the root of <= 0.02
                // per element keeps the line through the last two G within 10 (3 x 0.02)^2 = 3.6 % of G over the three elements ahead
                // (typically ~0.1 %), and rules out a group that straddles the extremum of G under a target (first difference ~0,
                // second difference not).  (A separate test on the first difference of G was redundant with this one: dropped.)
                *hold_ok = !__builtin_amdgcn_ballot_w64(fabsf(h1 - h2) > 0.02f) &&   // (lanes without a path: NaN, never true)
                           (__builtin_amdgcn_readfirstlane(R->info) & 16);            // the pitch is uniform around this group
            }
            L.G = G;
        }
        const float Sf = (G * dXf) * dXf;
        T = fma(u, fma(qd * L.inv_cm, Xt, ST), (double)Sf);
    }
    // store through the workgroup's buffer descriptor (base: its first output row, extent: its block of rows): the
    // element's row enters as the scalar offset, each lane supplies a 32-bit byte offset (no 64-bit per-lane address
    // arithmetic, no descriptor rebuilt per row); lanes beyond the last target have redone the last target and store
    // to its address (gfx950 range-checks scalar + lane offset together, so a one-row extent cannot drop them)
    {
        const u32x2 bits = {(unsigned)__double2loint(T), (unsigned)__double2hiint(T)};
        __builtin_amdgcn_raw_buffer_store_b64(bits, rs, f8, soff, 0);        // f8: the target's byte offset in a row (8 f)
        if (ITERS && live) (iters + row)[f8 >> 3] = (uint8_t)it;
    }
    // history for the predictor (fp32): the root itself, q + (untaken step), with the sign of xf - xe (FAST: it has it)
    const float qroot = q + dq;
    return FAST ? qroot : __builtin_copysignf(qroot, __int_as_float(__double2hiint(dxs)));    // one v_bfi_b32
}

// A workgroup = 256 focal points x `eb` consecutive elements (loop).  Besides re-using the layer
// set-up while ze repeats, the loop gives each lane a CONTINUATION PREDICTOR: the signed solution
// qs = sign(xf - xe) q is a smooth function of the element position, so the four previous
// solutions extrapolate the next one (cubic Lagrange form on the actual element positions) to ~1e-5 .. 1e-7
// and Newton needs one evaluation instead of ~4.5.  The weights depend on the element positions only: thread t of
// the workgroup works them out once for element e0 + t and parks them in LDS.
// The predictor is only a guess: iterates are clamped to the rigorous lower bound of the root,
// from which Newton is monotone, so convergence never depends on the elements being evenly spaced.
// NL = number of layers the medium has (n_if + 1); TAUP: the tau-p tail (accuracy tier); PERM: the elements arrive sorted by
// (depth, position) and element i's row of the table is a.row_of[i] — an aperture handed over in any order gets the predictor.
template <int NL, bool ITERS, bool TAUP = false, bool PERM = false>
__global__ __launch_bounds__(RTUS_BLOCK, (NL <= 3 && !ITERS) ? 8 : 1) void rtus_tt_layers_kernel(LayerArgs a)
{
    __shared__ ElemRec rec[64];
    // Work items = (column of 256 targets, block of rows, problem of a batch): one per workgroup.  -DRTUS_EXP_PERSIST (experiment
    // builds only, scripts/gpu_planar_r04.sh) runs them as a PERSISTENT grid instead — 8 workgroups per CU, workgroup w takes items
    // w, w + gridDim.x, ... — to find out whether the half-empty wave slots between the two rounds of a configs[2] launch cost time:
    // they do not (round 4: slots full, launch 1-2 % SLOWER) — the kernel is bound by VALU issue, not by residency.
#ifdef RTUS_EXP_PERSIST
    for (int item = blockIdx.x; item < a.n_items; item += gridDim.x) {
    const int bx = item % a.gx, byz = item / a.gx, by = byz % a.gy, bz = byz / a.gy;
    if (item != (int)blockIdx.x) __syncthreads();           // the previous item's records are still being read by the slower waves
#else
    {
    const int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#endif
    // batched launch: problem bz
    const double* __restrict__ xe_p = a.xe + (size_t)bz * a.e_stride;
    const double* __restrict__ ze_p = a.ze + (size_t)bz * a.e_stride;
    const double* __restrict__ xf_p = a.xf + (size_t)bz * a.f_stride;
    const double* __restrict__ zf_p = a.zf + (size_t)bz * a.f_stride;
    double* __restrict__ tt_p = a.tt + (size_t)bz * a.t_stride;
    uint8_t* __restrict__ it_p = ITERS ? a.iters + (size_t)bz * a.t_stride : nullptr;

    const int f_raw = bx * RTUS_BLOCK + threadIdx.x;
    const bool live = f_raw < a.n_f;
    const int f = live ? f_raw : a.n_f - 1;
    const double xf = xf_p[f];
    const unsigned f8 = (unsigned)f * 8u;                   // the target's byte offset in a row of the table (n_f x 8 < 2^32: the launcher checks)
    const int gb = a.row0 / a.eb + by;                      // the workgroup's block of the whole table
    const int e0 = max(gb * a.eb - a.row0, 0);              // ... in this launch's rows (a shard that starts inside a block keeps its tail)
    const int ne = min((gb + 1) * a.eb - a.row0, a.n_e) - e0;   // elements of this workgroup (<= 64)

    // ---- per-element records: the first wave works them out, one element per lane ----------------------
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int el = min(e0 + lane, a.n_e - 1);
        const int b1 = max(el - 1, e0), b2 = max(el - 2, e0), b3 = max(el - 3, e0), b4 = max(el - 4, e0);
        const double x0 = xe_p[el], z0 = ze_p[el];
        const double x1 = xe_p[b1], x2 = xe_p[b2], x3 = xe_p[b3], x4 = xe_p[b4];
        const double z1 = ze_p[b1], z2 = ze_p[b2], z3 = ze_p[b3], z4 = ze_p[b4];     // all loads issued together
        // how many predecessors inside this workgroup share the element's depth (the history restarts when ze changes)
        // (a predecessor at a non-finite position is no history either — its solution is NaN, and 0 x NaN in the predictor's unused
        // terms would carry it into up to four rows behind it: scripts/exp_nan_rows.py, tests/test_gpu_edge_sizes.py)
        const bool s1 = (lane >= 1) & (z1 == z0) & isfinite(x1), s2 = s1 & (lane >= 2) & (z2 == z0) & isfinite(x2),
                   s3 = s2 & (lane >= 3) & (z3 == z0) & isfinite(x3), s4 = s3 & (lane >= 4) & (z4 == z0) & isfinite(x4);
        int hist = s4 ? 4 : (s3 ? 3 : (s2 ? 2 : (s1 ? 1 : 0)));
        // Lagrange weights of the extrapolation to x0 from the nodes x1 .. x_m (differences in fp64, products in fp32)
        const float t1 = (float)(x0 - x1), t2 = (float)(x0 - x2), t3 = (float)(x0 - x3), t4 = (float)(x0 - x4);
        const float d12 = (float)(x1 - x2), d13 = (float)(x1 - x3), d14 = (float)(x1 - x4), d23 = (float)(x2 - x3),
                    d24 = (float)(x2 - x4), d34 = (float)(x3 - x4);
        // duplicate positions leave no slope: use as many nodes as are pairwise distinct, at least the latest one
        const bool ok2 = d12 != 0.0f, ok3 = ok2 & (d13 != 0.0f) & (d23 != 0.0f),
                   ok4 = ok3 & (d14 != 0.0f) & (d24 != 0.0f) & (d34 != 0.0f);
        const int m = (hist >= 4 && ok4) ? 4 : ((hist >= 3 && ok3) ? 3 : ((hist >= 2 && ok2) ? 2 : (hist >= 1 ? 1 : 0)));
        float w1 = 0.0f, w2 = 0.0f, w3 = 0.0f, w4 = 0.0f;
        if (m == 4) {
            w1 = t2 * t3 * t4 / (d12 * d13 * d14);
            w2 = -t1 * t3 * t4 / (d12 * d23 * d24);
            w3 = t1 * t2 * t4 / (d13 * d23 * d34);
            w4 = -t1 * t2 * t3 / (d14 * d24 * d34);
        } else if (m == 3) {
            w1 = t2 * t3 / (d12 * d13);
            w2 = -t1 * t3 / (d12 * d23);
            w3 = t1 * t2 / (d13 * d23);
        } else if (m == 2) {
            w1 = t2 / d12;
            w2 = -t1 / d12;
        } else if (m == 1) {
            w1 = 1.0f;
        }
        // run of consecutive elements (from this one on) that all have four distinct same-depth predecessors
        const unsigned long long m4 = __ballot(m == 4 && lane < ne);
        const unsigned long long rest = ~(m4 >> lane);
        const int run = (m == 4 && lane < ne) ? (rest ? __ffsll((long long)rest) - 1 : 64 - lane) : 0;
        // HOLD (tau-p tier) extrapolates along the ELEMENT INDEX: only where the pitch is uniform (2 %) over the four elements behind
        // and the three ahead (an aperture of random spacing put 4.5e-9 relative on a table before this test existed —
        // tests/test_gpu_irregular_apertures.py::test_planar_taup_tier_irregular_and_coarse_apertures).  Behind: this lane's own
        // differences; ahead: the same statement of the lane three further on (they share the step x_i - x_(i-1)).  No loads, no LDS
        // word of its own: the workgroup's start-up is not overlapped with anything (0.1 us there is 1 % of a configs[1] launch).
        bool uni = false;
        if (TAUP && a.eb >= 8) {                            // (a block of fewer rows has no group of four behind four cold rows)
            const float tolu = 0.02f * fabsf(t1);
            const bool ub = m == 4 && lane < ne && fabsf(d12 - t1) <= tolu && fabsf(d23 - t1) <= tolu && fabsf(d34 - t1) <= tolu;
            const unsigned long long ubm = __ballot(ub);
            uni = __builtin_amdgcn_inverse_ballot_w64(ubm & (ubm >> 3));
        }
        ElemRec r;
#ifdef RTUS_EXP_BAD_PREDICTOR                               // experiment builds only (scripts/selftest_predictor.sh): the continuation tests must notice
        w1 *= 1.02f; w3 *= 0.97f;
#endif
        r.w1 = w1; r.w2 = w2; r.w3 = w3; r.w4 = w4; r.xe = x0;
        r.info = m | (hist == 0 ? 8 : 0) | (uni ? 16 : 0) | (run << 8);
        r.row = PERM ? a.row_of[el] : 0;
        rec[lane] = r;
    }
    __syncthreads();

    const RecPtr rec3 = (RecPtr)rec;
    Lane<NL> L;
    L.tau = INFINITY; L.rS3 = 0.0f; L.G = L.dG = 0.0f; L.hic = 0.0f; L.inv_cm = 0.0; L.hr0 = L.hc0 = 0.0; L.hr0f = L.rs0f = L.rhmf = L.asymf = 0.0f;
#pragma unroll
    for (int i = 0; i < NL; ++i) { L.hr[i] = L.kk[i] = L.hc[i] = 0.0; L.hrf[i] = L.kkf[i] = 0.0f; }
    float qa = 0.0f, qb = 0.0f, qc = 0.0f, qd = 0.0f;       // signed solutions of the four previous elements, qa the latest
    const size_t nf = (size_t)a.n_f;
    size_t o = (size_t)e0 * nf;                              // output row of element e0 + li (wave-uniform; ITERS only)
    const unsigned row_bytes = (unsigned)a.n_f * 8u;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(tt_p + o, 0, (unsigned)ne * row_bytes, 0x00020000);
    unsigned so = 0;                                         // li * row_bytes (< 2^32: the launcher sizes eb for it)
    int li = 0;
    // where element `idx` of the block is stored: its row inside the workgroup's block of rows (one descriptor, the row as the
    // instruction's scalar offset) or, PERM, a descriptor of the row it belongs to (a few scalar instructions per element)
    auto dest_o = [&](int idx, size_t off) { return PERM ? (size_t)(unsigned)__builtin_amdgcn_readfirstlane(rec[idx].row) * nf : off; };
    auto dest_rs = [&](int idx, size_t off) {
        if (!PERM) return rs;
        return __builtin_amdgcn_make_buffer_rsrc(tt_p + dest_o(idx, off), 0, row_bytes, 0x00020000);
    };
    auto dest_so = [&](unsigned off) { return PERM ? 0u : off; };
    while (li < ne) {                                        // wave-uniform loop
        const int info = __builtin_amdgcn_readfirstlane(rec[li].info);
        if (info & 8) {                                      // depth changed: redo the layer set-up, forget the history
            layer_setup<NL>(a, ze_p[e0 + li], zf_p, f8, L);
            qa = qb = qc = qd = 0.0f;                        // a lane may carry NaN history from a depth at which its target
                                                             // was not below the element
        }
        const int run4 = (info >> 8) & ~3;
        if (run4 > 0) {
            // four-history run, unrolled by four so that the history rotates through its registers without moves.  The records are
            // read at ONE per-lane address register advanced once per trip, with immediate offsets (left to itself the compiler
            // forms each record's address on the scalar unit and pays a v_mov per ds_read: two VALU slots per solve)
            unsigned rp_off = (unsigned)li * (unsigned)sizeof(ElemRec);
            asm volatile("" : "+v"(rp_off));
            RecPtr rp = (RecPtr)((const __attribute__((address_space(3))) char*)rec3 + rp_off);
            // HOLD (tau-p tier).  The untaken Newton step dq = dXf / X' and the tail's second-order term G dXf^2 need X'(q) and u^3
            // only to a few per cent — dq moves the history by <= tau q, the term is <= (tau^2 / 8) T — and both drift by ~1-2 % from
            // one element to the next.  So the FIRST solve of a group of four forms them exactly (HOLD = 1: the sum for X', v_rcp_f32,
            // u^3) and, where the root moves by <= 0.02 per element (wave-uniform test: |d ln G / dq| <= 3 and its curvature bound the
            // line through the last two G to within ~3.6 % of G over three elements) and the pitch is uniform around the group (bit 4
            // of the record: the line runs along the element index), the other three reuse them (HOLD = 2: eleven fp32 instructions
            // fewer per solve, stopping at tau / 3); otherwise they form their own (HOLD = 0).  Error of the held term: <= 3.6 % of
            // ((tau / 3)^2 / 8) T = 4.5e-11 T at the stopping threshold, 1e-15 T typically.
            // A row's bits stay a function of the table (the groups start where the four-history run starts: a function of the
            // aperture and of the rows per block).
            bool held = false;                              // L.G is one element old (else: four)
            for (int r = 0; r < run4; r += 4) {
#ifdef RTUS_EXP_NO_HOLD
                if (false) {
#else
                if (TAUP) {
#endif
                    bool ok = false;
                    qd = solve_elem<NL, ITERS, true, TAUP, TAUP ? 1 : 0>(it_p, L, rp + 0, 4, xf, qa, qb, qc, qd, live, dest_o(li + r, o), f8, dest_rs(li + r, o), dest_so(so), held ? 0.25f : 1.0f, &ok);
                    if (ok) {
                        qc = solve_elem<NL, ITERS, true, TAUP, TAUP ? 2 : 0>(it_p, L, rp + 1, 4, xf, qd, qa, qb, qc, live, dest_o(li + r + 1, o + nf), f8, dest_rs(li + r + 1, o + nf), dest_so(so + row_bytes));
                        qb = solve_elem<NL, ITERS, true, TAUP, TAUP ? 2 : 0>(it_p, L, rp + 2, 4, xf, qc, qd, qa, qb, live, dest_o(li + r + 2, o + 2 * nf), f8, dest_rs(li + r + 2, o + 2 * nf), dest_so(so + 2 * row_bytes));
                        qa = solve_elem<NL, ITERS, true, TAUP, TAUP ? 2 : 0>(it_p, L, rp + 3, 4, xf, qb, qc, qd, qa, live, dest_o(li + r + 3, o + 3 * nf), f8, dest_rs(li + r + 3, o + 3 * nf), dest_so(so + 3 * row_bytes));
                    } else {
                        qc = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 1, 4, xf, qd, qa, qb, qc, live, dest_o(li + r + 1, o + nf), f8, dest_rs(li + r + 1, o + nf), dest_so(so + row_bytes));
                        qb = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 2, 4, xf, qc, qd, qa, qb, live, dest_o(li + r + 2, o + 2 * nf), f8, dest_rs(li + r + 2, o + 2 * nf), dest_so(so + 2 * row_bytes));
                        qa = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 3, 4, xf, qb, qc, qd, qa, live, dest_o(li + r + 3, o + 3 * nf), f8, dest_rs(li + r + 3, o + 3 * nf), dest_so(so + 3 * row_bytes));
                    }
                    held = ok;
#ifdef RTUS_EXP_COUNT   // groups of four, and how many of them reused the first solve's X' and u^3
                    if ((threadIdx.x & 63) == 0) { atomicAdd(&planar_dbg[5], 1ull); if (ok) atomicAdd(&planar_dbg[6], 1ull); }
#endif
                } else {
                    qd = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 0, 4, xf, qa, qb, qc, qd, live, dest_o(li + r, o), f8, dest_rs(li + r, o), dest_so(so));
                    qc = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 1, 4, xf, qd, qa, qb, qc, live, dest_o(li + r + 1, o + nf), f8, dest_rs(li + r + 1, o + nf), dest_so(so + row_bytes));
                    qb = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 2, 4, xf, qc, qd, qa, qb, live, dest_o(li + r + 2, o + 2 * nf), f8, dest_rs(li + r + 2, o + 2 * nf), dest_so(so + 2 * row_bytes));
                    qa = solve_elem<NL, ITERS, true, TAUP>(it_p, L, rp + 3, 4, xf, qb, qc, qd, qa, live, dest_o(li + r + 3, o + 3 * nf), f8, dest_rs(li + r + 3, o + 3 * nf), dest_so(so + 3 * row_bytes));
                }
                o += 4 * nf; so += 4 * row_bytes;
                rp += 4;
            }
            li += run4;
        } else {
            const float qn = solve_elem<NL, ITERS, false, TAUP>(it_p, L, rec3 + li, info & 7, xf, qa, qb, qc, qd, live, dest_o(li, o), f8, dest_rs(li, o), dest_so(so));
            qd = qc; qc = qb; qb = qa; qa = qn;
            o += nf; so += row_bytes;
            ++li;
        }
    }
    }   // items
}

// Order of the aperture.  The predictor extrapolates from the four previous elements of a workgroup's block, which only works when
// consecutive elements are neighbours in space at one depth; an aperture handed over in another order fell back to cold-started
// Newton (~4.5 evaluations instead of 1).  rtus_rank_kernel sorts it by (depth, position, index) — every element counts the
// elements before it in that order (O(n^2) compares, a wave per element: 65 k for 256 elements) and drops its coordinates and its
// index at that rank — and the PERM kernels store each row where it belongs.  Keys are the IEEE bits made monotone, so the order
// is total whatever the values.
__device__ __forceinline__ unsigned long long order_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ __launch_bounds__(256) void rtus_rank_kernel(const double* __restrict__ xe, const double* __restrict__ ze, int n,
                                                        double* __restrict__ xs, double* __restrict__ zs, int* __restrict__ row_of)
{
    // one WAVE per element: 64 compares per step, a ballot count each (a thread per element looping over all the others is a
    // chain of n dependent steps on one wave: 33 us for 256 elements)
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;                                       // wave-uniform
    const double x = xe[i], z = ze[i];
    const unsigned long long mx = order_key(x), mz = order_key(z);
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        const unsigned long long ox = order_key(xe[min(j, n - 1)]), oz = order_key(ze[min(j, n - 1)]);
        const bool before = j < n && ((oz < mz) || (oz == mz && (ox < mx || (ox == mx && j < i))));
        rank += __popcll(__ballot(before));
    }
    if (lane == 0) { xs[rank] = x; zs[rank] = z; row_of[rank] = i; }
}

// Elements (table rows) per workgroup: as many as possible (predictor + set-up reuse: the first four elements of a workgroup
// start Newton cold) while keeping >= ~4 waves per SIMD; 64 once that still leaves two full rounds of 8 waves per SIMD (measured
// on BASELINE config 3: 32 -> 64 elements per workgroup = -4.5 % time; on config 2: 8 is the optimum).  A function of the
// WHOLE table (n_rows_total x n_f, x n_batch problems), never of the launch: row shards of a table use the table's value.
int rtus_rows_per_block(long long n_rows_total, int n_f, int n_batch, int elem_bytes)
{
    const long long wave_solves = (long long)((n_f + 63) / 64) * n_rows_total * n_batch;
    int eb = (int)(wave_solves / (1024LL * 4));
    eb = eb < 1 ? 1 : (eb > 32 ? 32 : eb);
    if (wave_solves >= 1024LL * 16 * 64) eb = 64;
    // fp32 tables are the curved-lens kernel's (the planar kernel is fp64 only): its block starts with three cold solves (~490 of
    // a 64-row block's 3,600 VALU instructions once most rows need no Newton step), so twice the rows per block where that still
    // leaves two full rounds of waves
    const int eb_max = elem_bytes == 4 ? 128 : 64;
    if (elem_bytes == 4 && wave_solves >= 1024LL * 16 * 128) eb = 128;
#ifdef RTUS_EXP_EB                                          // experiment builds only (scripts/ab_planar.py)
    eb = RTUS_EXP_EB < 1 ? 1 : (RTUS_EXP_EB > eb_max ? eb_max : RTUS_EXP_EB);
#endif
    while ((n_rows_total + eb - 1) / eb > 65535 && eb < eb_max) ++eb;   // grid.y limit (eb <= 64 / 128: one LDS record per element)
    if ((unsigned long long)eb * (unsigned long long)n_f * (unsigned)elem_bytes >= 0xffffffffull)   // row offsets inside a block are 32-bit
        eb = (int)(0xffffffffull / ((unsigned long long)n_f * (unsigned)elem_bytes));
    return eb;
}

#ifndef RTUS_PLANAR_SHAPE_LDS
#define RTUS_PLANAR_SHAPE_LDS 0u   // experiment builds: unused LDS per workgroup = fewer workgroups per CU (see launch_layers)
#endif
static hipError_t launch_layers(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                                const double* xf, const double* zf, int n_f, double* tt, uint8_t* iters, int n_batch,
                                long long e_stride, long long f_stride, long long t_stride, int row0, long long n_rows_total,
                                unsigned flags, const int* row_of, hipStream_t s)
{
    LayerArgs a;
    a.row_of = row_of;
    for (int i = 0; i < RTUS_MAX_LAYERS; ++i) a.z_if[i] = i < n_if ? z_if[i] : INFINITY;
    for (int i = 0; i <= RTUS_MAX_LAYERS; ++i) { a.c[i] = i <= n_if ? c[i] : 1.0; a.inv_c[i] = 1.0 / a.c[i]; }
    a.n_if = n_if; a.xe = xe; a.ze = ze; a.xf = xf; a.zf = zf; a.tt = tt; a.iters = iters;
    a.n_e = n_e; a.n_f = n_f;
    a.e_stride = e_stride; a.f_stride = f_stride; a.t_stride = t_stride;
    const int eb = rtus_rows_per_block(n_rows_total, n_f, n_batch, 8);
    if (eb < 1 || (n_rows_total + eb - 1) / eb > 65535) return hipErrorInvalidValue;
    a.row0 = row0;
    a.eb = eb;
    a.gx = (n_f + RTUS_BLOCK - 1) / RTUS_BLOCK; a.gy = (row0 + n_e - 1) / eb - row0 / eb + 1;
    const long long items = (long long)a.gx * a.gy * n_batch;
    if (items > 0x7fffffffLL) return hipErrorInvalidValue;
    a.n_items = (int)items;
#ifdef RTUS_EXP_PERSIST                                      // experiment builds only: see the kernel
    long long cap = items;
    if (n_if + 1 <= 3 && !iters) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cap = 8LL * n;
    }
    const dim3 grid((unsigned)(items < cap ? items : cap)), block(RTUS_BLOCK);
#else
    const dim3 grid(a.gx, a.gy, n_batch), block(RTUS_BLOCK);
#endif
    const bool taup = (flags & RTUS_TT_TAUP_TAIL) != 0;
    // Workgroups per CU: what the registers allow (8 for the 64-VGPR kernels).  Capping a launch of two rounds or more BELOW that with
    // LDS the kernel never touches (-DRTUS_PLANAR_SHAPE_LDS=bytes, experiment builds) was worth -3.5 % at six per CU while the kernel
    // still parked set-up values in scratch and loaded extra header words (start-up that nothing overlapped when every wave of a SIMD
    // began at once); on the kernel as it is now six per CU is 0.5 - 1.6 % SLOWER than eight and seven is even: no shaping.
    const unsigned lds = (n_if + 1 <= 3 && !iters && items >= 4096) ? RTUS_PLANAR_SHAPE_LDS : 0u;
    switch (n_if + 1) {
#define RTUS_CASE(NL) case NL: if (iters) hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, true, false, false>), grid, block, lds, s, a); \
                               else if (row_of && taup) hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, false, true, true>), grid, block, lds, s, a); \
                               else if (row_of) hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, false, false, true>), grid, block, lds, s, a); \
                               else if (taup) hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, false, true, false>), grid, block, lds, s, a); \
                               else hipLaunchKernelGGL((rtus_tt_layers_kernel<NL, false, false, false>), grid, block, lds, s, a); break;
        RTUS_CASE(1) RTUS_CASE(2) RTUS_CASE(3) RTUS_CASE(4) RTUS_CASE(5) RTUS_CASE(6) RTUS_CASE(7) RTUS_CASE(8)
        RTUS_CASE(9)
#undef RTUS_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t rtus_launch_tt_layers(const double* z_if, const double* c, int n_if, const double* xe,
                                 const double* ze, int n_e, const double* xf, const double* zf, int n_f,
                                 double* tt, uint8_t* iters, unsigned flags, hipStream_t s)
{
    return launch_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, iters, 1, 0, 0, 0, 0, n_e, flags, nullptr, s);
}

// rows [row0, row0 + n_e) of a table of n_rows_total rows: the same bits as the whole table's launch gives those rows when
// row0 is a multiple of rtus_rows_per_block(n_rows_total, n_f, 1, 8)
hipError_t rtus_launch_tt_layers_rows(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                                      int row0, long long n_rows_total, const double* xf, const double* zf, int n_f, double* tt,
                                      unsigned flags, hipStream_t s)
{
    return launch_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, nullptr, 1, 0, 0, 0, row0, n_rows_total, flags, nullptr, s);
}

// n_batch independent problems of one shape and one medium in ONE launch (several apertures and / or target sets):
// problem b reads xe/ze + b e_stride, xf/zf + b f_stride and writes tt + b t_stride.
hipError_t rtus_launch_tt_layers_batch(const double* z_if, const double* c, int n_if, const double* xe, const double* ze,
                                       int n_e, long long e_stride, const double* xf, const double* zf, int n_f,
                                       long long f_stride, double* tt, long long t_stride, int n_batch, unsigned flags, hipStream_t s)
{
    return launch_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, nullptr, n_batch, e_stride, f_stride, t_stride, 0, n_e, flags, nullptr, s);
}

// The whole table with the aperture in ANY order: sorted copies of the coordinates + each element's row in `ws`
// (rtus_layers_sort_ws_bytes), then the PERM kernel.  `presorted_row_of`: the caller (the host-buffer twin) has sorted on the
// host and uploaded sorted xe / ze and the row indices itself — no rank launch.
size_t rtus_layers_sort_ws_bytes(int n_e) { return ((size_t)n_e * 8 + 255) / 256 * 256 * 2 + ((size_t)n_e * 4 + 255) / 256 * 256; }
hipError_t rtus_launch_tt_layers_sorted(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                                        const double* xf, const double* zf, int n_f, double* tt, void* ws, const int* presorted_row_of,
                                        unsigned flags, hipStream_t s)
{
    if (presorted_row_of)
        return launch_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, nullptr, 1, 0, 0, 0, 0, n_e, flags, presorted_row_of, s);
    const size_t seg = ((size_t)n_e * 8 + 255) / 256 * 256;
    double* xs = (double*)ws;
    double* zs = (double*)((char*)ws + seg);
    int* row_of = (int*)((char*)ws + 2 * seg);
    hipLaunchKernelGGL(rtus_rank_kernel, dim3((n_e + 3) / 4), dim3(256), 0, s, xe, ze, n_e, xs, zs, row_of);
    return launch_layers(z_if, c, n_if, xs, zs, n_e, xf, zf, n_f, tt, nullptr, 1, 0, 0, 0, 0, n_e, flags, row_of, s);
}
