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

#ifdef RTUS_EXP_COUNT   // experiment builds only (scripts/exp_lens_tonly.py): how often the T-only rows run
static __device__ unsigned long long lens_dbg[8];
#define LDBG(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&lens_dbg[i], 1ull); } while (0)
extern "C" int rtus_dbg_read_lens(unsigned long long* out)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(lens_dbg), sizeof(unsigned long long) * 8);
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(lens_dbg), z, sizeof(z));
    return 0;
}
#else
#define LDBG(i) do {} while (0)
#endif

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
    int row0;                // index of xe[0] in the whole table (row shards: workgroups stay aligned to the table's blocks, see rtus_fermat.hip)
    int poly_trig;           // 1: [a_lo, a_hi] lies inside [-1, 1] rad -> sin/cos by polynomial, no range reduction
    R gp_min, gp_dx;         // a lane whose g' at its minimum is below gp_min (+ gp_dx x the block's reach) may have a second minimum (kernel comment)
    unsigned long long* __restrict__ stats;   // nullable (rtus_tt_lens*_stats_dev): per wave-element, how it was solved —
                                              // [0] T alone at the extrapolated start, [1] one evaluation (lite step), [2] the safeguarded
                                              // iteration, [3] of those: with a scan of the whole interval, [4] evaluations inside [2]
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
// 1/v to ~1e-7 through the fp32 pipe: Newton steps, the T polish and the alpha output only multiply small
// corrections by it (IEEE fp64 divide costs 25 ns per wave-op, this ~7 ns)
template <typename R> __device__ __forceinline__ R rcp_r(R v) { return (R)__builtin_amdgcn_rcpf((float)v); }

// sin and cos for |a| <= 1 rad (the lens is only defined inside +-50.6 deg, main_rt.py:479): Taylor polynomials
// in Horner form, truncation 4e-23 (fp64, degree 21/22) / 2.5e-8 (fp32, degree 9/10) — the generic sincos spends
// most of its time on a range reduction that is never needed here.
template <typename R> __device__ __forceinline__ void sincos_poly(R a, R& s, R& c);
template <> __device__ __forceinline__ void sincos_poly<float>(float a, float& s, float& c)
{
    // degree 9 / 10: the first dropped terms, a^11/11! and a^12/12!, are 2.5e-8 and 2.1e-9 at |a| = 1 — below half an fp32 ulp of
    // sin 1 and cos 1 (and 6e-9 / 5e-10 at the lens's own +-0.884 rad)
    const float x2 = a * a;
    float ps = fmaf(x2, 2.7557319e-6f, -1.9841270e-4f);     //  1/9!, -1/7!
    ps = fmaf(x2, ps, 8.3333333e-3f);                        //  1/5!
    ps = fmaf(x2, ps, -1.6666667e-1f);                       // -1/3!
    s = fmaf(a * x2, ps, a);
    float pc = fmaf(x2, -2.7557319e-7f, 2.4801587e-5f);     // -1/10!, 1/8!
    pc = fmaf(x2, pc, -1.3888889e-3f);                       // -1/6!
    pc = fmaf(x2, pc, 4.1666667e-2f);                        //  1/4!
    pc = fmaf(x2, pc, -0.5f);
    c = fmaf(x2, pc, 1.0f);
}
template <> __device__ __forceinline__ void sincos_poly<double>(double a, double& s, double& c)
{
    const double x2 = a * a;
    double ps = fma(x2, -1.9572941063391263e-20, 8.2206352466243295e-18);   // -1/21!, 1/19!
    ps = fma(x2, ps, -2.8114572543455206e-15);               // -1/17!
    ps = fma(x2, ps, 7.6471637318198164e-13);                //  1/15!
    ps = fma(x2, ps, -1.6059043836821613e-10);               // -1/13!
    ps = fma(x2, ps, 2.5052108385441720e-8);                 //  1/11!
    ps = fma(x2, ps, -2.7557319223985893e-6);                // -1/9!
    ps = fma(x2, ps, 1.9841269841269841e-4);                 //  1/7!
    ps = fma(x2, ps, -8.3333333333333333e-3);                // -1/5!
    ps = fma(x2, ps, 1.6666666666666666e-1);                 //  1/3!
    s = fma(-(a * x2), ps, a);
    double pc = fma(x2, 8.8967913924505741e-22, -4.1103176233121648e-19);   // 1/22!, -1/20!
    pc = fma(x2, pc, 1.5619206968586225e-16);                //  1/18!
    pc = fma(x2, pc, -4.7794773323873853e-14);               // -1/16!
    pc = fma(x2, pc, 1.1470745597729725e-11);                //  1/14!
    pc = fma(x2, pc, -2.0876756987868100e-9);                // -1/12!
    pc = fma(x2, pc, 2.7557319223985888e-7);                 //  1/10!
    pc = fma(x2, pc, -2.4801587301587302e-5);                // -1/8!
    pc = fma(x2, pc, 1.3888888888888889e-3);                 //  1/6!
    pc = fma(x2, pc, -4.1666666666666664e-2);                // -1/4!
    pc = fma(x2, pc, 0.5);
    c = fma(-x2, pc, 1.0);
}
template <typename R> __device__ __forceinline__ void sincos_r(R a, R* s, R* c);
template <> __device__ __forceinline__ void sincos_r<float>(float a, float* s, float* c) { sincosf(a, s, c); }
template <> __device__ __forceinline__ void sincos_r<double>(double a, double* s, double* c) { sincos(a, s, c); }

// The lens constants as the solver's arithmetic reads them.  On gfx950 an fp32 VALU instruction with an SGPR operand
// issues at half the rate of one with VGPR / inline-constant operands (4.2 vs 2.3 cycles, scripts/ubench_issue3.hip), and
// kernel arguments live in SGPRs: the fp32 instantiation keeps per-lane (VGPR) copies, made opaque to the compiler so
// that it does not fold them back into scalar operands.  (fp64 instructions cost 4.2 cycles either way: no copies.)
template <typename R> struct LensConst {
    R c1inv, c2inv, phi_3, twoTc, C4A, inv2A;
    int poly_trig;
};
template <typename R> __device__ __forceinline__ LensConst<R> lens_const(const LensFermatArgs<R>& a)
{
    LensConst<R> k;
    k.c1inv = a.c1inv; k.c2inv = a.c2inv; k.phi_3 = a.phi_3; k.twoTc = a.twoTc; k.C4A = a.C4A; k.inv2A = a.inv2A;
    k.poly_trig = a.poly_trig;
    if (sizeof(R) == 4) asm("" : "+v"(k.c1inv), "+v"(k.c2inv), "+v"(k.phi_3), "+v"(k.twoTc), "+v"(k.C4A), "+v"(k.inv2A));
    return k;
}

// T(alpha), g = dT/dalpha and (WITH_GP) g' for one (A, F).  Without g' the second derivatives h'', P'' are skipped:
// about a quarter of the arithmetic.
template <typename R, bool WITH_GP, bool POLY>
__device__ __forceinline__ void lens_time(const LensConst<R>& k, R alpha, R xa, R za, R xf, R zf, R& T, R& g,
                                          R& gp)
{
    R s, c;
    if (POLY) sincos_poly<R>(alpha, s, c);                  // [a_lo, a_hi] inside +-1 rad: chosen at launch
    else sincos_r<R>(alpha, &s, &c);
    const R B = k.phi_3 * c - k.twoTc;                      // main_rt.py:184
    const R B1 = -k.phi_3 * s, B2 = -k.phi_3 * c;           // B', B''
    const R disc = B * B - k.C4A;
    const R rS = rsqrt_r<R>(disc);                          // 1/S
    const R S = disc * rS;
    const R h = -(B + S) * k.inv2A;                         // :171-177 root [1]
    const R BrS = B * rS;
    const R h1 = -B1 * (R(1) + BrS) * k.inv2A;              // :199-212
    const R px = h * s, pz = h * c;                         // :220-221
    const R p1x = h1 * s + pz, p1z = h1 * c - px;           // :231-232
    const R ax = px - xa, az = pz - za, fx = px - xf, fz = pz - zf;
    const R ra = rsqrt_r<R>(ax * ax + az * az), rf = rsqrt_r<R>(fx * fx + fz * fz);
    const R la = (ax * ax + az * az) * ra, lf = (fx * fx + fz * fz) * rf;
    const R ua = (ax * p1x + az * p1z) * ra, uf = (fx * p1x + fz * p1z) * rf;     // u . P'
    T = la * k.c1inv + lf * k.c2inv;
    g = ua * k.c1inv + uf * k.c2inv;
    if (WITH_GP) {
        const R h2 = -(B2 * (R(1) + BrS) + B1 * B1 * rS * (R(1) - BrS * BrS)) * k.inv2A;
        const R p2x = h2 * s + R(2) * h1 * c - px, p2z = h2 * c - R(2) * h1 * s - pz;
        const R pp = p1x * p1x + p1z * p1z;
        gp = ((pp - ua * ua) * ra + (ax * p2x + az * p2z) * ra) * k.c1inv
           + ((pp - uf * uf) * rf + (fx * p2x + fz * p2z) * rf) * k.c2inv;
    }
}

// T(alpha) alone: no P', no g.  T is stationary in alpha at the ray (Fermat), so evaluated delta away from the minimiser it is off by
// g' delta^2 / 2 — with g' ~ 1e-4 s/rad^2: 5e-15 s at delta = 1e-5 rad (fp32's tolerance), 5e-21 s at 1e-8 rad (fp64's).
template <typename R, bool POLY>
__device__ __forceinline__ R lens_time_only(const LensConst<R>& k, R alpha, R xa, R za, R xf, R zf)
{
    R s, c;
    if (POLY) sincos_poly<R>(alpha, s, c);
    else sincos_r<R>(alpha, &s, &c);
    const R B = k.phi_3 * c - k.twoTc;                      // main_rt.py:184
    const R disc = B * B - k.C4A;
    const R S = disc * rsqrt_r<R>(disc);
    const R h = -(B + S) * k.inv2A;                         // :171-177 root [1]
    const R px = h * s, pz = h * c;                         // :220-221
    const R ax = px - xa, az = pz - za, fx = px - xf, fz = pz - zf;
    const R da = ax * ax + az * az, df = fx * fx + fz * fz;
    return da * rsqrt_r<R>(da) * k.c1inv + df * rsqrt_r<R>(df) * k.c2inv;
}

typedef unsigned int lens_u32x2 __attribute__((ext_vector_type(2)));
// Stores through ONE descriptor per workgroup: its base is the workgroup's first output row, its extent the workgroup's
// block of rows (gfx950 range-checks the scalar offset together with the per-lane one: a one-row extent drops every
// row but the first); the row of the current element enters as the instruction's scalar offset — one s_add per element
// instead of rebuilding a descriptor from a 64-bit row pointer.  Lanes past the last target redo the last target
// (same inputs, same result) and store to ITS address: no exec masking, no reliance on the range check.
template <typename R> __device__ __forceinline__ void store_at(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, R v);
template <> __device__ __forceinline__ void store_at<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, float v)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, soff, 0);
}
template <> __device__ __forceinline__ void store_at<double>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, double v)
{
    const lens_u32x2 bits = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(bits, rs, voff, soff, 0);
}
template <typename R> __device__ __forceinline__ void store_row(R* row, unsigned n_f, unsigned f, R v);
template <> __device__ __forceinline__ void store_row<float>(float* row, unsigned n_f, unsigned f, float v)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, n_f * 4u, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, f * 4u, 0, 0);
}
template <> __device__ __forceinline__ void store_row<double>(double* row, unsigned n_f, unsigned f, double v)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(row, 0, n_f * 8u, 0x00020000);
    const lens_u32x2 bits = {(unsigned)__double2loint(v), (unsigned)__double2hiint(v)};
    __builtin_amdgcn_raw_buffer_store_b64(bits, rs, f * 8u, 0, 0);
}

// One element's record in LDS (broadcast reads: no v_readlane, no SGPR operands — see rtus_fermat.hip).
template <typename R> struct __attribute__((aligned(16))) LensRec {
    R xa, za;
    float w1, w3;            // Lagrange weights of the newest / oldest of the three previous solutions (w2 = 1 - w1 - w3)
    int mode;                // 0: no history, 1: previous alpha, 2: extrapolate
    int run;                 // number of consecutive mode-2 elements from this one on (inside the workgroup's block)
    float reach;             // the farthest any element of the block lies from this one (m): how far g' can drift before the block ends
};

// A workgroup = 256 targets x `eb` consecutive elements.  The minimiser alpha*(element) is smooth in the element
// position, so the three previous solutions extrapolate the next start (Lagrange weights from the element
// positions, worked out once per workgroup by thread t for element e0 + t and parked in LDS, as in
// rtus_fermat.hip): from the third element of a block on, Newton starts ~1e-7 rad from the root.
// POLY: sin / cos by polynomial (search interval inside +-1 rad); WA: the minimising alpha is written too.  Both are
// launch-time facts: as template parameters they leave no wave-uniform branches in the element loop.
template <typename R, bool POLY, bool WA>
__global__ __launch_bounds__(RTUS_BLOCK, sizeof(R) == 4 ? 8 : 3) void rtus_tt_lens_kernel(LensFermatArgs<R> a)
{
    __shared__ LensRec<R> rec[128];                          // eb <= 128 (fp32 tables; 64 for fp64: rtus_rows_per_block)
    __shared__ unsigned long long m2s[2];                    // per header wave: which of its elements extrapolate (mode 2)
    __shared__ float xext[4];                                // per header wave: least and greatest element position (block extent)
    const int f_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = f_raw < a.n_f;
    const int f = live ? f_raw : a.n_f - 1;
    const R xf = a.xf[f], zf = a.zf[f];
    const int gb = a.row0 / a.eb + blockIdx.y;              // the workgroup's block of the whole table
    const int e0 = max(gb * a.eb - a.row0, 0), ne = min((gb + 1) * a.eb - a.row0, a.n_e) - e0;
    LensRec<R> r;
    const int idx = threadIdx.x, lane = threadIdx.x & 63;    // header: thread idx works out the record of the block's element idx
    if (idx < 128) {
        const int el = min(e0 + idx, a.n_e - 1);
        const R xe_v = a.xe[el], ze_v = a.ze[el];
        const int b1 = max(el - 1, e0), b2 = max(el - 2, e0), b3 = max(el - 3, e0);
        const R x1 = a.xe[b1], x2 = a.xe[b2], x3 = a.xe[b3];
        const R z1 = a.ze[b1], z2 = a.ze[b2], z3 = a.ze[b3];            // all loads issued together
        const bool s1 = (idx >= 1) & (z1 == ze_v), s2 = s1 & (idx >= 2) & (z2 == ze_v), s3 = s2 & (idx >= 3) & (z3 == ze_v);
        const int hist = s3 ? 3 : (s2 ? 2 : (s1 ? 1 : 0));
        const float d12 = (float)(x1 - x2), d13 = (float)(x1 - x3), d23 = (float)(x2 - x3);
        const float t1 = (float)(xe_v - x1), t2 = (float)(xe_v - x2), t3 = (float)(xe_v - x3);
        const bool lin = (hist >= 2) & (x1 != x2);
        const bool quad = lin & (hist >= 3) & (x3 != x1) & (x3 != x2);
        const float r12 = __builtin_amdgcn_rcpf(d12);
        const float q1 = t2 * t3 * __builtin_amdgcn_rcpf(d12 * d13);
        const float q3 = t1 * t2 * __builtin_amdgcn_rcpf(d13 * d23);
        r.xa = xe_v; r.za = ze_v;
        r.w1 = quad ? q1 : (lin ? t2 * r12 : 0.0f);
        r.w3 = quad ? q3 : 0.0f;
#ifdef RTUS_EXP_BAD_PREDICTOR                               // experiment builds only (scripts/selftest_predictor.sh): the continuation tests must notice
        r.w1 *= 1.05f; r.w3 *= 0.9f;
#endif
        r.mode = lin ? 2 : (hist >= 1 ? 1 : 0);
        const unsigned long long m2 = __builtin_amdgcn_ballot_w64(r.mode == 2 && idx < ne);
        float xmn = (float)xe_v, xmx = (float)xe_v;          // (idx >= ne repeats the last element: no effect on the extent)
        for (int sh = 32; sh > 0; sh >>= 1) { xmn = fminf(xmn, __shfl_xor(xmn, sh)); xmx = fmaxf(xmx, __shfl_xor(xmx, sh)); }
        if (lane == 0) { m2s[idx >> 6] = m2; xext[2 * (idx >> 6)] = xmn; xext[2 * (idx >> 6) + 1] = xmx; }
    }
    __syncthreads();
    if (idx < 128) {
        // the run of mode-2 elements from this one on: the rest of this wave's 64 elements, continued into the second wave's
        const unsigned long long mine = m2s[idx >> 6], next = idx < 64 ? m2s[1] : 0ull;
        const unsigned long long rest = ~(mine >> lane);
        const int here = rest ? __ffsll((long long)rest) - 1 : 64 - lane;
        const int more = (here == 64 - lane && idx < 64) ? (~next ? __ffsll((long long)~next) - 1 : 64) : 0;   // (the shift fills `rest` with ones from bit 64 - lane on)
        r.run = (r.mode == 2 && idx < ne) ? here + more : 0;
        const float xlo = fminf(xext[0], xext[2]), xhi = fmaxf(xext[1], xext[3]);
        r.reach = fmaxf(xhi - (float)r.xa, (float)r.xa - xlo) * (1.0f + 1e-6f);
        rec[idx] = r;
    }
    __syncthreads();
    LensConst<R> k = lens_const<R>(a);                       // (re-made after a scan, which works from the kernel arguments: see there)
    R a_lo = a.a_lo, a_hi = a.a_hi;
    if (sizeof(R) == 4) asm("" : "+v"(a_lo), "+v"(a_hi));
    // |d alpha| below which a lane stops iterating (rad).  The step it would have taken is still applied — to alpha
    // directly and to T through the second-order term below — so what is left is third order: with
    // |d alpha| <= 1e-8 rad that is ~1e-15 rad and < 1e-25 s in fp64 (measured against the reference rays:
    // tests/test_gpu_lens_fermat.py).  fp32: 1e-5 rad is the type's resolution of alpha.
    // (A looser fp32 tolerance for tables without the alpha output was measured — 1e-4 / 1e-3 rad: 2-7 % faster — and dropped:
    // the mean error against the fp64 table grew from 3 to 29 / 233 fp32 ulps on coarse apertures close to the lens.)
    constexpr bool T_ONLY = !WA;
    const R tol = sizeof(R) == 4 ? R(1e-5) : R(1e-8);
    // ... or when the step can no longer lower T noticeably (T is FLAT in alpha near the lens focus — the lens is
    // aplanatic — so alpha is ill-conditioned there while T is not): predicted gain g*step/2 below the type's resolution
    const R tolT = sizeof(R) == 4 ? R(2e-12) : R(1e-21);

    R al1 = R(0), al2 = R(0), al3 = R(0);                   // solutions of the three previous elements (al1 the newest)
    R rgp = R(0);                                           // 1 / g' of this lane's latest full evaluation (usable if that solve
                                                            // ended at an interior minimum: rgp_bad below)
    const unsigned row_bytes = (unsigned)a.n_f * (unsigned)sizeof(R);
    const unsigned blk_bytes = (unsigned)ne * row_bytes;     // < 2^32: the launcher sizes eb for it
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(a.tt + (size_t)e0 * a.n_f, 0, blk_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a =
        __builtin_amdgcn_make_buffer_rsrc((WA ? a.alpha_out : a.tt) + (size_t)e0 * a.n_f, 0, blk_bytes, 0x00020000);
    const unsigned voff = (unsigned)f * (unsigned)sizeof(R); // f: clamped to the last target
    unsigned soff = 0;                                       // li * row_bytes

    unsigned long long rgp_bad = ~0ull;                      // lanes whose rgp is not usable (wave-uniform mask: the test is one scalar compare)
    int n_tonly = 0, n_lite = 0, n_full = 0, n_scan = 0, n_trip = 0;   // wave-uniform tallies for a.stats (scalar adds)

    // WHEN THE MINIMUM FOLLOWED FROM ELEMENT TO ELEMENT MAY NOT BE THE LEAST TIME.  The lens is aplanatic: for a target at its
    // focus T(alpha) is constant, around the focus it is nearly flat, and an off-axis element can then have TWO local minima (an
    // interior ray and an end of the interval, or both ends — beyond the focus T has its maximum in the middle).  A continuation
    // follows one of them; the table entry is the lesser (Fermat).  Measured on the CPU over the water below the reference lens
    // (elements within +-20 mm, 4,001 samples of alpha per pair): wherever a pair has more than one local
    // minimum, g' at every INTERIOR one is <= 3.7e-6 s/rad^2, against >= 3e-5 on tables away from the focus (BASELINE configs[3]:
    // 2.9e-5 ... 2.3e-4; targets 1 cm from the focus: ~5e-6 ... 1.2e-5).  So a lane is SUSPECT when its minimum is pinned at an end
    // of the interval or its g' there is below gp_min (= 7.5e-6 for the reference lens: twice the bound): the generic step then looks
    // at the whole interval (`scan`).  The fast rows below do not evaluate g'; along the followed minimum it moves by 1.5 ... 2.6e-4
    // s/rad^2 per metre of element position (faster only where it is below gp_min anyway), so they run only while the latest full
    // evaluation says g' >= gp_min + gp_dx * (the farthest any element of the block is from that one) in every lane, gp_dx = 3.9e-4:
    // no element of the block can be suspect then.  Not suspect => one minimum => the continuation is right.
    const R gp_min = a.gp_min, gp_dx = a.gp_dx;              // (only the generic step reads them: scalar operands will do)
    unsigned long long flat_ahead = ~0ull;                   // lanes whose g' could fall below gp_min before the block ends (wave-uniform mask)

    // ---- the fast rows: elements of a run (mode 2), every lane with a usable, unsuspicious g' ----------------------------------
    // ONE evaluation of T and g at the start extrapolated from the three previous solutions (n1 the newest; differences first:
    // alpha* varies slowly, so the weights — 3, -3, 1 on an even pitch — act on small numbers), the Newton step with the g' of the
    // lane's latest full evaluation (g' varies by ~1e-3 from one element to the next; it only scales a step that is already below
    // the stopping tolerance) and its second-order term — no bracket, no g'', no loop.  Returns false — nothing stored, the
    // history untouched — when some lane's step is not small or some lane has become suspect: the generic step takes over.
    R miss = R(0);                                           // minimiser minus extrapolated start of the latest lite step (T_ONLY)
    auto lite = [&](int idx, R n1, R n2, R& n3) -> bool {
        const R xa = rec[idx].xa, za = rec[idx].za;
        const float w1 = rec[idx].w1, w3 = rec[idx].w3;     // w2 = 1 - w1 - w3
        const R alpha = fmin(fmax(n1 + ((R)(w1 - 1.0f) * (n1 - n2) + (R)w3 * (n3 - n2)), a_lo), a_hi);
        R T, g, gp = R(0);
        lens_time<R, false, POLY>(k, alpha, xa, za, xf, zf, T, g, gp);
        const R step = -g * rgp;
        const unsigned long long big = __builtin_amdgcn_ballot_w64(fabs(step) > tol) & __builtin_amdgcn_ballot_w64(fabs(g * step) > tolT);
        LDBG(2);
        if (big) { miss = step; return false; }              // (the generic step starts its iteration one Newton step on: `miss`)
        LDBG(3);
        n_lite += 1;
        miss = step;
        store_at<R>(rs_t, voff, soff, T + R(0.5) * g * step);    // T(a*) = T(a) - g^2 / (2 g')
        if (WA) store_at<R>(rs_a, voff, soff, alpha + step);
        soff += row_bytes;
        n3 = alpha + step;
        return true;
    };
    // Tables without the alpha output (T_ONLY): inside a run, after an element whose lite step succeeded (every lane's
    // extrapolated start within tol of its minimiser), the next elements take T at their extrapolated start and nothing else —
    // no P', no g, no Newton step: stationarity makes T(start) exact to g' tol^2 / 2 (fp32: 5e-15 s, a thousandth of an ulp of T;
    // fp64, tol = 1e-8 rad: 5e-21 s, below the resolution of T), and a later element verifies again (its Newton step re-anchors
    // alpha and measures how far the extrapolation had drifted).  Which elements verify is a function of the table (the element's
    // position in its run and where lite steps failed), never of the launch: shards reproduce it.
    // The unverified starts stay in the history the next starts are extrapolated from.  Left alone their errors feed
    // back (weights 3, -3, 1): with a truncation error tau per extrapolation the starts are off by tau, 4 tau, 10 tau, then
    // -10, -25, -44, ... tau — after a few triples the verified element's step exceeds tol and the whole wave iterates (measured:
    // 27 % SLOWER than verifying everything).  The verified element measures its own 10 tau (`miss`), so the two entries are
    // corrected by 0.1 and 0.4 of it, and the pattern tau, 4 tau, 10 tau repeats instead of growing.
    auto tonly = [&](int idx, R n1, R n2, R& n3) {
        const R xa = rec[idx].xa, za = rec[idx].za;
        const float w1 = rec[idx].w1, w3 = rec[idx].w3;
        const R alpha = fmin(fmax(n1 + ((R)(w1 - 1.0f) * (n1 - n2) + (R)w3 * (n3 - n2)), a_lo), a_hi);
        store_at<R>(rs_t, voff, soff, lens_time_only<R, POLY>(k, alpha, xa, za, xf, zf));
        soff += row_bytes;
        n_tonly += 1;
        n3 = alpha;
    };

    // ---- the generic step: any element, one at a time ----------------------------------------------------------------------------
    R xa_cur = R(0), za_cur = R(0);                          // the element the generic step is working on
    int nun = 0;                                             // unverified starts (T-only rows) in a row since the latest verified element (wave-uniform)
    // safeguarded Newton on g inside the bracket [lo, hi] (kept by the sign of g: T decreases left of a minimum), bisection
    // whenever Newton leaves it; wave-uniform trip count, ballot exit.  On return T, g, gp belong to alpha.
    auto iterate = [&](R& alpha, R lo, R hi, R& T, R& g, R& gp) {
        bool done = false;
        for (int trip = 0; trip < 80; ++trip) {
            LDBG(6);
            n_trip += 1;
            lens_time<R, true, POLY>(k, alpha, xa_cur, za_cur, xf, zf, T, g, gp);
            if (g > R(0)) hi = alpha; else lo = alpha;
            R step = -g * rcp_r<R>(gp);
            R next = alpha + step;
            const bool bad = !(gp > R(0)) || !(next > lo) || !(next < hi);
            if (bad) { next = R(0.5) * (lo + hi); step = next - alpha; }
            const bool go = fabs(step) > tol && fabs(g * step) > tolT && !done;
            if (!__builtin_amdgcn_ballot_w64(go)) break;
            if (go) alpha = next; else done = true;         // a finished lane keeps its alpha (T, g belong to it)
        }
    };
    // second-order polish without another evaluation: T(a*) = T(a) - g^2 / (2 g') — only where Newton converged in the interior;
    // a minimum pinned at an end of the bracket keeps T(alpha)
    auto polish = [&](R& alpha, R& T, R g, R gp) -> bool {
        const bool interior = gp > R(0) && (fabs(g) <= gp * (R(16) * tol) || fabs(g * g) <= gp * (R(16) * tolT));
        const R dal = interior ? -g * rcp_r<R>(gp) : R(0);
        T = interior ? T + R(0.5) * g * dal : T;
        alpha += dal;
        return interior;
    };
    // The whole interval for a wave that holds suspect lanes: T and g at NS + 1 even samples (any sample is an upper bound of the
    // least time: the ends — a pinned minimum — come in here), and in every cell whose ends say "a minimum inside" (g < 0 left,
    // g >= 0 right: 0.055 rad per cell against features of T that are ~1 rad wide) the zero of g by the Illinois variant of regula
    // falsi, a fixed number of evaluations of T and g (no g', no second derivatives: few registers — this is the rare path, and the
    // table's common rows must not pay for it in occupancy).  T at the zero needs no polish (stationary).  The least T of all of
    // it and of the local solve wins; only suspect lanes take it.  Returns with `moved` = the scan's own minimum was taken.
    auto scan = [&](R& bA, R& bT, bool& moved) {
        constexpr int NS = 32, NI = sizeof(R) == 4 ? 12 : 28;
        // the lens constants and the interval straight from the kernel arguments (scalar operands: half rate in fp32, which does not
        // matter here): the per-lane copies the fast rows use are dead across the scan and made again after it — eight registers
        LensConst<R> ks;
        ks.c1inv = a.c1inv; ks.c2inv = a.c2inv; ks.phi_3 = a.phi_3; ks.twoTc = a.twoTc; ks.C4A = a.C4A; ks.inv2A = a.inv2A; ks.poly_trig = a.poly_trig;
        const R s_lo = a.a_lo, s_hi = a.a_hi;
        const R dA = (s_hi - s_lo) * R(1.0 / NS);
        R pa = s_lo, pg, dummy = R(0);
        asm("" : "+v"(pa) : "v"(xa_cur));                    // (or the first sample's target-side half is hoisted to the top of the kernel and spilled)
        {
            R pT;
            lens_time<R, false, POLY>(ks, pa, xa_cur, za_cur, xf, zf, pT, pg, dummy);
            if (pT < bT) { bT = pT; bA = pa; moved = true; }
        }
        for (int j = 1; j <= NS; ++j) {
            const R ca = j == NS ? s_hi : fma((R)j, dA, s_lo);
            R cT, cg;
            lens_time<R, false, POLY>(ks, ca, xa_cur, za_cur, xf, zf, cT, cg, dummy);
            if (cT < bT) { bT = cT; bA = ca; moved = true; }
            const bool cell = pg < R(0) && cg >= R(0);
            if (__builtin_amdgcn_ballot_w64(cell)) {         // wave-uniform
                // (lo, glo < 0), (hi, ghi >= 0); a lane without a minimum in this cell iterates on a made-up bracket and keeps nothing
                R lo = pa, glo = cell ? pg : R(-1), hi = ca, ghi = cell ? cg : R(1);
                int side = 0;                                // which end the latest iterate replaced (Illinois: the other one's g is halved
                for (int it = 0; it < NI; ++it) {            //  when the same end is replaced twice in a row)
                    R x = lo - glo * (hi - lo) * rcp_r<R>(ghi - glo);
                    x = (x > lo && x < hi) ? x : R(0.5) * (lo + hi);
                    R T, g;
                    lens_time<R, false, POLY>(ks, x, xa_cur, za_cur, xf, zf, T, g, dummy);
                    if (T < bT && cell) { bT = T; bA = x; moved = true; }
                    if (g < R(0)) { lo = x; glo = g; ghi = side < 0 ? R(0.5) * ghi : ghi; side = -1; }
                    else { hi = x; ghi = g; glo = side > 0 ? R(0.5) * glo : glo; side = 1; }
                }
            }
            pa = ca; pg = cg;
        }
    };
    // the fast rows were left at a verified element: its measured miss corrects the unverified starts still in the history, as a
    // lite step's does (before any scan: a suspect lane's minimiser may jump, the local solve's miss is what the history needs)
    auto hist_fix = [&](R ms) {
        if (T_ONLY && nun > 0) {
            al1 += (nun == 2 ? R(0.4) : (nun == 3 ? R(0.5) : R(35.0 / 56.0))) * ms;
            al2 += (nun == 2 ? R(0.1) : (nun == 3 ? R(0.2) : R(20.0 / 56.0))) * ms;
            nun = 0;
        }
    };
    int good = 0;                                            // elements in a row, up to the latest, whose ONE evaluation was enough (wave-uniform)
    auto generic = [&](int idx, int mode, bool try_lite) {
        bool took_lite = !try_lite;                          // a lite step at this element's start has been evaluated (here or in the fast rows)
        xa_cur = rec[idx].xa; za_cur = rec[idx].za;
        R alpha;
        if (mode == 2) {
            const float w1 = rec[idx].w1, w3 = rec[idx].w3;
            alpha = al1 + ((R)(w1 - 1.0f) * (al1 - al2) + (R)w3 * (al3 - al2));
        } else if (mode == 1) {
            alpha = al1;
        } else {
            // first guess: polar angle of the point where the straight chord A-F meets the lens apex height
            const R hz = -(k.phi_3 - k.twoTc + sqrt((k.phi_3 - k.twoTc) * (k.phi_3 - k.twoTc) - k.C4A)) * k.inv2A;
            const R t = (za_cur - hz) / (za_cur - zf);
            alpha = atan2(xa_cur + t * (xf - xa_cur), hz);
        }
        alpha = fmin(fmax(alpha, a_lo), a_hi);
        const R start = alpha;
        R T, g, gp = R(0);
        bool one = false;                                    // wave-uniform: the lite step was enough
        // a run's first elements extrapolate from fewer than three solutions (w3 = 0): their lite step nearly always fails
        if (try_lite && mode == 2 && (rgp_bad | flat_ahead) == 0 && __builtin_amdgcn_readfirstlane(__float_as_int(rec[idx].w3)) != 0) {   // as in the fast rows
            lens_time<R, false, POLY>(k, alpha, xa_cur, za_cur, xf, zf, T, g, gp);
            const R step = -g * rgp;
            const unsigned long long big = __builtin_amdgcn_ballot_w64(fabs(step) > tol) & __builtin_amdgcn_ballot_w64(fabs(g * step) > tolT);
            LDBG(2);
            miss = step;
            took_lite = true;
            if (!big) {
                LDBG(3);
                n_lite += 1;
                T += R(0.5) * g * step;
                alpha += step;
                hist_fix(step);
                one = true;
            }
        }
        if (!one) {
            LDBG(4);
            n_full += 1;
            // the fast rows' failed lite step (or the one just above) has evaluated `start` already: the iteration begins one
            // Newton step on (with the previous element's g': iterate() takes it from there, inside the interval)
            alpha = fmin(fmax(start + (took_lite ? miss : R(0)), a_lo), a_hi);
            iterate(alpha, a_lo, a_hi, T, g, gp);
            const bool interior = polish(alpha, T, g, gp);
            hist_fix(alpha - start);
            R gpl = interior ? gp : R(0);                    // this lane's g' at its minimum (0: none)
            const bool sus = !(gpl >= gp_min);               // pinned, NaN, or flat enough for a second minimum to exist
            if (__builtin_amdgcn_ballot_w64(sus)) {          // wave-uniform
                LDBG(5);
                n_scan += 1;
                R bA = alpha, bT = T;
                bool moved = false;
                scan(bA, bT, moved);
                if (sus) { alpha = bA; T = bT; gpl = moved ? R(0) : gpl; }   // (no g' at a minimum the scan found: the next element is a generic step)
                k = lens_const<R>(a);
                a_lo = a.a_lo; a_hi = a.a_hi;
                if (sizeof(R) == 4) asm("" : "+v"(a_lo), "+v"(a_hi));
            }
            rgp = rcp_r<R>(gpl);
            rgp_bad = __builtin_amdgcn_ballot_w64(!(gpl > R(0)));
            flat_ahead = __builtin_amdgcn_ballot_w64(!(gpl >= fma(gp_dx, (R)rec[idx].reach, gp_min)));
        }
        good = one ? good + 1 : 0;
        store_at<R>(rs_t, voff, soff, T);
        if (WA) store_at<R>(rs_a, voff, soff, alpha);
        soff += row_bytes;
        al3 = al2; al2 = al1; al1 = alpha;
    };

    int li = 0;
    while (li < ne) {                                        // wave-uniform
        const int mode = __builtin_amdgcn_readfirstlane(rec[li].mode);
        const int run = __builtin_amdgcn_readfirstlane(rec[li].run);
        bool left = false;                                   // the fast rows were left at a lite step that failed: no second try
        if (mode == 2 && run >= 3 && (rgp_bad | flat_ahead) == 0 && (!T_ONLY || good >= 2)) {
            // the run of mode-2 elements from here on, unrolled by three so that the history rotates through its registers
            // without moves; left at the first lite step that fails (rot = how far into its triple).  T_ONLY tables come here after
            // two elements in a row that needed one evaluation, and start with a triple of T-only rows.
            int rot = 0;
            nun = 0;
            bool skip_next = true;                           // the next triple starts unverified as well
            int t = run / 3;
            for (; t > 0; --t) {
                if (T_ONLY && skip_next) {                   // every other triple does not verify at all
                    tonly(li, al1, al2, al3);
                    skip_next = false;
                    nun += 1;
                } else {
                    if (!lite(li, al1, al2, al3)) { rot = 0; break; }   // newest .. oldest = al3, al1, al2
                    // n unverified starts in a row are off by C(k+2, 3) tau (k = 1..n) and the verified one after them by C(n+3, 3) tau
                    // = miss: the two still in the history (k = n, n - 1) are corrected by their share of it (n = 2, 3 or 5)
                    if (T_ONLY && nun > 0) {
                        al1 += (nun == 2 ? R(0.4) : (nun == 3 ? R(0.5) : R(35.0 / 56.0))) * miss;
                        al2 += (nun == 2 ? R(0.1) : (nun == 3 ? R(0.2) : R(20.0 / 56.0))) * miss;
                    }
                    skip_next = true;
                    nun = 0;
                }
                LDBG(0);
                if (T_ONLY) {
                    LDBG(1);
                    tonly(li + 1, al3, al1, al2);
                    tonly(li + 2, al2, al3, al1);
                    nun += 2;
                } else {
                    if (!lite(li + 1, al3, al1, al2)) { rot = 1; break; }   //                    al2, al3, al1
                    if (!lite(li + 2, al2, al3, al1)) { rot = 2; break; }   //                    al1, al2, al3
                }
                li += 3;
            }
            if (t == 0) nun = 0;
            if (t > 0) {                                     // left early: the history back in canonical order
                if (rot == 1) { const R n = al3; al3 = al2; al2 = al1; al1 = n; }
                if (rot == 2) { const R n = al2, m = al3; al3 = al1; al1 = n; al2 = m; }
                li += rot;
                left = true;
            } else if (li >= ne || __builtin_amdgcn_readfirstlane(rec[li].mode) != 2) {
                continue;
            }
        }
        generic(li, mode, !left);
        ++li;
    }
    if (a.stats && lane == 0) {                              // (null in every product call)
        atomicAdd(a.stats + 0, (unsigned long long)n_tonly); atomicAdd(a.stats + 1, (unsigned long long)n_lite);
        atomicAdd(a.stats + 2, (unsigned long long)n_full); atomicAdd(a.stats + 3, (unsigned long long)n_scan);
        atomicAdd(a.stats + 4, (unsigned long long)n_trip);
    }
}

int rtus_rows_per_block(long long n_rows_total, int n_f, int n_batch, int elem_bytes);   // rtus_fermat.hip

template <typename R>
static hipError_t launch_lens(const rtus_lens& L, double a_lo, double a_hi, const R* xe, const R* ze, int n_e,
                              const R* xf, const R* zf, int n_f, R* tt, R* alpha_out, int row0, long long n_rows_total, unsigned long long* stats, hipStream_t s)
{
    const LensK kk = make_lens_k(L);
    LensFermatArgs<R> k;
    k.c1inv = (R)(1.0 / L.c1); k.c2inv = (R)(1.0 / L.c2);
    k.phi_3 = (R)kk.phi_3; k.twoTc = (R)kk.twoTc; k.C4A = (R)kk.C4A; k.inv2A = (R)(1.0 / kk.twoA);
    k.a_lo = (R)a_lo; k.a_hi = (R)a_hi;
    k.xe = xe; k.ze = ze; k.xf = xf; k.zf = zf; k.tt = tt; k.alpha_out = alpha_out;
    k.n_e = n_e; k.n_f = n_f; k.stats = stats;
    k.poly_trig = (a_lo >= -1.0 && a_hi <= 1.0) ? 1 : 0;
    // the lens's own time and speed scale the two constants measured on the reference lens: 7.5e-6 s/rad^2 (twice the largest g' seen
    // at an interior minimum of a pair with two minima; h0 / c2 = 5.96e-5 s) and 3.9e-4 s/rad^2 per metre (g' at the followed minimum
    // moves by 1.5 ... 2.6e-4 per metre of element position wherever it is above that — 1 / c1 = 1.56e-4 —, faster only next to the focus)
    k.gp_min = (R)(0.125 * L.h0 / L.c2); k.gp_dx = (R)(2.5 / L.c1);
    k.eb = rtus_rows_per_block(n_rows_total, n_f, 1, (int)sizeof(R));   // of the WHOLE table: row shards reproduce its bits
    if (k.eb < 1 || (n_rows_total + k.eb - 1) / k.eb > 65535) return hipErrorInvalidValue;
    k.row0 = row0;
    const dim3 grid((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK, (row0 + n_e - 1) / k.eb - row0 / k.eb + 1);
    const bool poly = k.poly_trig != 0, wa = alpha_out != nullptr;
    const unsigned lds = 0;   // (fewer workgroups per CU through unused LDS, as the planar kernel's launch does: no gain here — 7: +0.2 %, 6: +0.7 %, 5: +2 %)
    if (poly && wa) hipLaunchKernelGGL((rtus_tt_lens_kernel<R, true, true>), grid, dim3(RTUS_BLOCK), lds, s, k);
    else if (poly) hipLaunchKernelGGL((rtus_tt_lens_kernel<R, true, false>), grid, dim3(RTUS_BLOCK), lds, s, k);
    else if (wa) hipLaunchKernelGGL((rtus_tt_lens_kernel<R, false, true>), grid, dim3(RTUS_BLOCK), lds, s, k);
    else hipLaunchKernelGGL((rtus_tt_lens_kernel<R, false, false>), grid, dim3(RTUS_BLOCK), lds, s, k);
    return hipGetLastError();
}

hipError_t rtus_launch_tt_lens_f64(const rtus_lens& L, double a_lo, double a_hi, const double* xe, const double* ze,
                                   int n_e, const double* xf, const double* zf, int n_f, double* tt,
                                   double* alpha_out, int row0, long long n_rows_total, hipStream_t s, unsigned long long* stats)
{
    return launch_lens<double>(L, a_lo, a_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, row0, n_rows_total, stats, s);
}

hipError_t rtus_launch_tt_lens_f32(const rtus_lens& L, double a_lo, double a_hi, const float* xe, const float* ze,
                                   int n_e, const float* xf, const float* zf, int n_f, float* tt, float* alpha_out,
                                   int row0, long long n_rows_total, hipStream_t s, unsigned long long* stats)
{
    return launch_lens<float>(L, a_lo, a_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, row0, n_rows_total, stats, s);
}
