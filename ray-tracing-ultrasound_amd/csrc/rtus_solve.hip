// rtus_solve.hip — pulse-echo travel times by root-finding x_land(alpha) = x_rx (the north-star's replacement for the
// reference's grid scan + tolerance matcher, main_rt.py:479-501).  Its own translation unit because it is compiled with
// -mllvm -disable-machine-licm (Makefile): the iteration is a LOOP around trace_ray, and the machine-level loop-invariant
// code motion hoists the ~60 64-bit literals and scalar table loads of the trigonometric kernels out of it — 242 VGPRs
// and 110 spilled SGPRs; without the hoisting 123 - 131 VGPRs and (almost) nothing in scratch.
#include "rtus_trace.h"

// ---- pulse-echo root-finding solve ------------------------------------------------------------
// The reference matches elements to whichever GRID ray happens to land within a tolerance
// (main_rt.py:487-501).  Here x_land(alpha) = x_rx is solved: the grid trace (rtus_shoot_kernel)
// only brackets the roots — consecutive finite grid rays whose landing points straddle the element —
// and each bracket is refined by a bracketed interpolation iteration, every evaluation being a
// full trace_ray at that lane's own alpha.  x_land(alpha) is U-shaped, so an element usually has two
// ray paths; all (up to RTUS_MAX_ROOTS) are returned in ascending alpha, plus the least-time one.
//
// One kernel, three phases per workgroup (RTUS_SOLVE_WAVES wave-tasks = that many (row, 64-element chunk) pairs):
//   A  brackets: lanes = rx elements of the task's row.  MASKS (apertures up to RTUS_SOLVE_MASK_MAX_RX elements): the grid
//      trace has already asked the question pair by pair (rtus_shoot_kernel<., true>) — the lane reads one 64-bit mask per
//      64-ray block and takes its set bits in ascending order, plus the pair that straddles two blocks.  Otherwise: a
//      lock-step scan of the row's landing points (wave-uniform index -> scalar loads; 64-ray blocks that cannot bracket
//      any of the wave's elements are skipped through the (min, max) intervals the grid trace left behind).  Either way up
//      to four brackets per element, in ascending alpha;
//   B  the workgroup's brackets are compacted into ONE list in LDS (wave prefix by ballots, the wave bases after a
//      barrier): an element of the reference's aperture has 0.6 brackets on average, so lanes = elements would leave
//      most lanes idle through every trace_ray — and would refine an element's brackets one after another;
//   C  waves take 64 list entries at a time: lanes = brackets, all refined together (different rows in one wave:
//      the per-row inputs are per-lane values here);
//   D  after a barrier the element lanes of phase A collect their roots from LDS (ascending alpha, compacted).
// The iteration (scripts/proto_solve_iter.c, CPU model on the reference sweep's 40,459 brackets): start = inverse
// quadratic interpolation through the bracket's two grid rays and the neighbouring grid ray on the side of the smaller
// residual; then inverse quadratic interpolation through the three latest points, secant / bisection as the safeguard
// (candidate outside the bracket, or |f| not halved); a lane stops when |f| <= 1e-13 m, or — from its second evaluation
// on — when |f| <= 1e-9 m and the interpolation step it would take next is below 1e-8 rad: that step is NOT taken but
// applied to alpha and (linearly, through the two latest evaluations) to T; what is left is third order.  2.01
// evaluations per bracket (Illinois, round 2: 3.92) and 99.3 % of the brackets done after two — which matters more than
// the mean: a wave iterates until its slowest lane is done; |dT| <= 4e-15 s against bisection, the same root / no-root
// decisions on all of them.
#ifdef RTUS_EXP_TIMING   // experiment builds only (scripts/exp_solve_timing.py): wall-clock stamps (100 MHz) per phase and wave
__device__ unsigned long long rtus_solve_stamps[4096][16];
#define STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6) < 4096) \
    rtus_solve_stamps[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)][i] = wall_clock64(); } while (0)
extern "C" int rtus_solve_stamps_read(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(rtus_solve_stamps), sizeof(unsigned long long) * 4096 * 16);
}
#else
#define STAMP(i) do {} while (0)
#endif
#ifndef RTUS_SOLVE_WAVES
#define RTUS_SOLVE_WAVES 4
#endif
#define RTUS_SOLVE_TPB (RTUS_SOLVE_WAVES * 64)
#define RTUS_RESCUE_MAX 21          // brackets of one wave that can borrow two finished lanes each
// Waves per SIMD the register allocation aims at.  5 (96 VGPRs, what VERDICT r02 asked for) costs ~70 spilled registers since
// the straggler hand-over came in; 4 (128 VGPRs, 2 spilled) measured faster at both sizes: 73.6 -> 71.5 us per sweep-sized pass,
// 841 -> 822 us at scale (vector form 707 -> 663 us); 3 is slower again at scale.
#ifndef RTUS_SOLVE_MIN_WAVES
#define RTUS_SOLVE_MIN_WAVES 4
#endif
struct SolveArgs {
    ShootArgs s;                        // lens, geometry, tx, polyline + boxes, flags
    const double* __restrict__ alpha;   // [n]  grid
    const double* __restrict__ land_x;  // [rows][n] landing x of the grid rays on z = z_land
    const double2* __restrict__ land_box;  // [rows][nb] (min, max) of the finite landing x of rays 64B .. 64B+63 (scan mode)
    const unsigned long long* __restrict__ pair_mask;   // [rows][nb][rx_pad] (mask mode), see ShootArgs
    int nb, rx_pad;
    const double* __restrict__ x_rx;    // [n_rx]
    double z_land;
    int n_rx, chunks;                   // chunks = ceil(n_rx / 64)
    long long n_tasks;                  // mask mode: rows * n_rx (flat element lanes); scan mode: rows * chunks (wave-tasks)
    double* __restrict__ tt;            // [rows][n_rx] least-time root (NaN: none)
    double* __restrict__ alpha_root;    // nullable, its launch angle
    double* __restrict__ tt_all;        // nullable [rows][n_rx][RTUS_MAX_ROOTS]
    double* __restrict__ alpha_all;     // nullable [rows][n_rx][RTUS_MAX_ROOTS]
    uint8_t* __restrict__ n_roots;      // nullable [rows][n_rx]
};

// 1 / v to ~1e-15 (the iteration only forms candidates with it)
__device__ __forceinline__ double solve_rcp(double v)
{
    double y = __builtin_amdgcn_rcp(v);
    y = fma(y, fma(-v, y, 1.0), y);
    return fma(y, fma(-v, y, 1.0), y);
}
// inverse quadratic interpolation to f = 0 through (xa, fa), (xb, fb), (xc, fc), relative to xa
__device__ __forceinline__ double solve_iqi(double xa, double fa, double xb, double fb, double xc, double fc)
{
    const double db = xb - xa, dc = xc - xa;
    const double q = solve_rcp((fb - fa) * (fb - fc) * (fc - fa));
    // db fa fc / ((fb-fa)(fb-fc)) + dc fa fb / ((fc-fa)(fc-fb)) over the common denominator
    return xa + fa * q * (db * fc * (fc - fa) - dc * fb * (fb - fa));
}

// EPB (mask mode): element lanes per workgroup.  256 = all four waves; 128 = two waves of elements, four waves of brackets: with
// about one bracket per element the list then fits ONE trip of the four waves even where elements have two (a list of 260
// brackets on 256 lanes costs its workgroup a second trip for four of them — in a launch of a few hundred waves that trip is
// the kernel's critical path).  The launcher takes 128 for launches that leave SIMDs idle anyway.
// The three-lanes-per-bracket refinement of ONE trip of a wave (see rtus_solve_kernel's TRIO comment): lanes 3 jb, 3 jb + 1, 3 jb + 2
// hold bracket `br` of the row whose grid landing points are `lrow` (global memory, or LDS in the one-launch kernel) and trace
// cand - d, cand, cand + d; `tri` is this wave's [64][2] exchange buffer in LDS.  All lanes of the wave call it together.
template <bool FAST>
__device__ __forceinline__ void trio_refine(const SolveArgs& q, const ShootArgs& a, const LensK& k, const double* lrow, RayIn in, double xr,
                                            int br, int n, int lane, int jb, int role, double (*tri)[2], bool& root, double& T_out,
                                            double& x_out)
{
    double xlo = q.alpha[br], xhi = q.alpha[br + 1];
    double flo = lrow[br] - xr, fhi = lrow[br + 1] - xr;
    double xt, ft;                                                   // third grid ray: as in the one-lane scheme below
    {
        const bool left = fabs(flo) < fabs(fhi);
        const int t1 = left ? br - 1 : br + 2, t2 = left ? br + 2 : br - 1;
        const int c1 = min(max(t1, 0), n - 1), c2 = min(max(t2, 0), n - 1);
        const double l1 = lrow[c1], l2 = lrow[c2];
        const bool ok1 = c1 == t1 && isfinite(l1), ok2 = c2 == t2 && isfinite(l2);
        xt = q.alpha[ok1 ? c1 : c2];
        ft = (ok1 ? l1 : (ok2 ? l2 : NAN)) - xr;
    }
    // one evaluation at a grid ray: it IS the root (fhi == 0), both ends are within 1e-13 m (x_land flat to rounding), or ONE end is
    // within 1e-15 m — a root at a grid ray up to rounding, where a triple around the candidate would have no width (the centre
    // element over a centred or mirror-symmetric pipe: x_land(0) = x_e to ~1e-17 m)
    const bool end_hi = fabs(fhi) <= 1e-15, end_lo = fabs(flo) <= 1e-15;
    const bool single = fhi == 0.0 || (fabs(flo) <= 1e-13 && fabs(fhi) <= 1e-13) || end_hi || end_lo;
    const double sec = xlo - flo * (xhi - xlo) * solve_rcp(fhi - flo);
    double cand = solve_iqi(xlo, flo, xhi, fhi, xt, ft);             // NaN without a third ray
    if (!(cand > xlo && cand < xhi)) cand = sec;
    if (!(cand > xlo && cand < xhi)) cand = 0.5 * (xlo + xhi);
    // how far the candidate may be off: a fraction of what the quadratic term moved it from the secant's zero
    double delta = fmin(fmax(0.08 * fabs(cand - sec), 3e-6), 1e-4);
    if (single) { cand = (fhi == 0.0 || fabs(fhi) < fabs(flo)) ? xhi : xlo; delta = 0.0; }
    double fprev_abs = INFINITY;
    double x_fin = NAN, T_fin = NAN, f_fin = NAN;
    bool done = false, dead = false;
    for (int it = 0; it < 48; ++it) {
        if (!__any(!done)) break;
        STAMP(3 + min(it, 9));
        if (!single) delta = fmin(delta, 0.999 * fmin(cand - xlo, xhi - cand));
        const double ac = cand + (double)(role - 1) * delta;
        double sn, cs, px, pz, dz, dx;
        rtus_sincos(ac, sn, cs);
        lens_eval_sc(k, sn, cs, px, pz, dz, dx);                     // main_rt.py:338, 344 at this lane's alpha
        in.P = make_double2(px, pz);
        { const double rt = rsqrt_fast(dx * dx + dz * dz); in.tu = make_double2(dx * rt, dz * rt); }
        if (!FAST) in.phis = rtus_atan2(dz, dx);
        RayOut o;
        trace_ray<FAST>(a, in, o);                                   // all 64 lanes together
        const double t1 = seg_time<FAST>(in.xa, in.za, px, pz, k.c1, k.inv_c1);
        const double t2 = seg_time<FAST>(px, pz, o.xq, o.zq, k.c2, k.inv_c2);
        const double t3 = seg_time<FAST>(o.xq, o.zq, o.xi, o.zi, k.c2, k.inv_c2);
        const double t4 = seg_time<FAST>(o.xi, o.zi, o.x_in, q.z_land, k.c1, k.inv_c1);
        tri[lane][0] = o.x_in - xr;
        tri[lane][1] = ((t1 + t2) + t3) + t4;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const double fm = tri[3 * jb][0], f0 = tri[3 * jb + 1][0], fq = tri[3 * jb + 2][0];
        const double Tm = tri[3 * jb][1], T0 = tri[3 * jb + 1][1], Tq = tri[3 * jb + 2][1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (!done) {                                                 // (the three lanes of a bracket decide alike: same inputs)
            if (!isfinite(f0)) { dead = true; done = true; }         // the branch ends inside the bracket
            else if (single || f0 == 0.0) { done = true; x_fin = cand; T_fin = T0; f_fin = f0; }
            else {
                const bool mono = isfinite(fm) && isfinite(fq) && (fq - f0) * (f0 - fm) > 0.0 && delta > 0.0;
                double x3 = NAN;
                if (mono) {
                    x3 = solve_iqi(cand, f0, cand - delta, fm, cand + delta, fq);
                    const double u = (x3 - cand) * solve_rcp(delta);
                    // (within 2 d of the centre: a kink in between costs <= 2 d x the slope's jump — 1e-4 relative on the reference's 905-point
                    // polyline, ~1e-3 on a 120-point one: <= 6e-12 rad at d = 3e-9.)  And the branch must be CONTINUOUS across the
                    // triple — x_land also changes sign at jumps (the first crossing moves to another chord: every element between the two
                    // landing points is "bracketed"), and a jump is no ray path: a slope beyond 1e3 m/rad over 2e-8 rad is one
                    // (found by scripts/fuzz_solve.py: whole rows reported an extra root; the reference's own sweep has no jump)
                    if (delta <= 1e-8 && fabs(u) <= 2.0 && fabs(fq - fm) <= 2e3 * delta) {
                        done = true;
                        f_fin = 0.0;                                 // a root next to three fresh points
                        x_fin = x3;
                        T_fin = T0 + 0.5 * u * ((Tq - Tm) + u * ((Tq - T0) - (T0 - Tm)));
                    }
                }
                // a jump: the sign changes across a triple 2e-8 rad wide by more than a continuous branch's could (> 5e3 m/rad): no root
                if (!done && delta > 0.0 && delta <= 1e-8 && isfinite(fm) && isfinite(fq) && (fm < 0.0) != (fq < 0.0) && fabs(fq - fm) > 1e-4) {
                    dead = true; done = true;
                }
                if (!done) {
                    // the three points tighten the bracket; the next centre comes from them
                    const double xs0 = cand - delta, xs2 = cand + delta;
                    if (isfinite(fm)) { if ((fm < 0.0) == (flo < 0.0)) { if (xs0 > xlo) { xlo = xs0; flo = fm; } } else if (xs0 < xhi) { xhi = xs0; fhi = fm; } }
                    { if ((f0 < 0.0) == (flo < 0.0)) { if (cand > xlo) { xlo = cand; flo = f0; } } else if (cand < xhi) { xhi = cand; fhi = f0; } }
                    if (isfinite(fq)) { if ((fq < 0.0) == (flo < 0.0)) { if (xs2 > xlo) { xlo = xs2; flo = fq; } } else if (xs2 < xhi) { xhi = xs2; fhi = fq; } }
                    if (xhi - xlo <= 1e-13) { done = true; x_fin = cand; T_fin = T0; f_fin = f0; }      // a jump — or the root itself
                    else {
                        double nc = x3;
                        if (!(nc > xlo && nc < xhi)) nc = xlo - flo * (xhi - xlo) * solve_rcp(fhi - flo);
                        if (!(nc > xlo && nc < xhi) || fabs(f0) > 0.5 * fprev_abs) { nc = 0.5 * (xlo + xhi); delta = 0.25 * (xhi - xlo); }
                        else delta = fmax(fmin(3e-9, 0.25 * fabs(nc - cand)), 1e-11);
                        fprev_abs = fabs(f0);
                        x_fin = cand; T_fin = T0; f_fin = f0;        // (what an exhausted iteration reports)
                        cand = nc;
                    }
                }
            }
        }
    }
    root = done && !dead && fabs(f_fin) < 1e-9;              // |f| large at convergence: a jump, not a root
    T_out = T_fin; x_out = x_fin;
}

// TRIO (round 4; small launches only — the launcher's choice is a function of the call's size, so which scheme refines a bracket is
// too): THREE LANES PER BRACKET.  In a launch that leaves most SIMDs idle the kernel lasts as long as its slowest wave's chain of
// dependent evaluations (~11 us each), and with one lane per bracket 0.7 % of the brackets — a third of the waves — need a third one.
// Here every round evaluates cand - d, cand, cand + d on three lanes of one wave and the bracket's root is the zero of the inverse
// quadratic through the three fresh points, T the parabola through their times.  x_land(alpha) has KINKS — where the returning ray
// moves to the next chord of the lens polyline its slope jumps by ~1e-4 relative — so a wide triple that straddles one is off by
// d x that jump: the zero is ACCEPTED only from a triple with d <= 1e-8 rad (error <= 1e-12 rad, third order without a kink); the
// first round (d = 3e-6 ... 1e-4, from how far the quadratic term moved the candidate off the secant's zero) only supplies the
// second round's centre, good to ~3e-10.  scripts/proto_solve_3lane.c on the sweep's 40,459 brackets: two rounds for 99.95 % (one
// lane per bracket: 99.3 %, and the rest cost their waves a third evaluation), the same root / no-root decisions as bisection,
// |dT| <= 7e-17 s, |dalpha| <= 1.1e-11 rad.  21 brackets per wave.
#define RTUS_TRIO_PER_WAVE 21
template <bool FAST, bool MASKS, int EPB = RTUS_SOLVE_TPB, bool TRIO = false>
__global__ __launch_bounds__(RTUS_SOLVE_TPB) __attribute__((amdgpu_waves_per_eu(RTUS_SOLVE_MIN_WAVES, 8))) void rtus_solve_kernel(SolveArgs q)
{
    __shared__ double tri[TRIO ? RTUS_SOLVE_WAVES : 1][TRIO ? 64 : 1][2];   // TRIO: per lane, landing residual and time of its latest evaluation
    __shared__ unsigned items[RTUS_SOLVE_TPB * RTUS_MAX_ROOTS];        // (slot << 2 | k) is implied by position: see below
    __shared__ int item_r[RTUS_SOLVE_TPB * RTUS_MAX_ROOTS];
    __shared__ double res_t[RTUS_SOLVE_TPB * RTUS_MAX_ROOTS], res_a[RTUS_SOLVE_TPB * RTUS_MAX_ROOTS];
    __shared__ int wave_tot[RTUS_SOLVE_WAVES];
    __shared__ double resc_in[RTUS_SOLVE_WAVES][RTUS_RESCUE_MAX][6];      // xa, za, r_outer, pipe_offset, candidate, delta
    __shared__ double resc_out[RTUS_SOLVE_WAVES][RTUS_RESCUE_MAX][2][2];  // [below / above the candidate] = landing x, travel time
    const ShootArgs& a = q.s;
    const LensK& k = a.k;
    const int n = a.n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;

    STAMP(0);
    // ---- A: brackets -----------------------------------------------------------------------------------------------
    // MASKS: lanes = (row, element) pairs, flat — a 65-element aperture fills its waves (64 lanes of one row + 1 lane in a wave
    // of its own would run every phase for that one lane); scan mode: wave-task = (row, 64-element chunk), the row wave-uniform.
    long long t_row;
    int e;
    bool live, task_live;
    if (MASKS) {
        const long long flat = (long long)blockIdx.x * EPB + threadIdx.x;
        live = flat < q.n_tasks && (int)threadIdx.x < EPB;             // n_tasks = rows * n_rx here
        const long long fl = live ? flat : q.n_tasks - 1;
        t_row = fl / q.n_rx;
        e = (int)(fl - t_row * q.n_rx);
        task_live = __any(live);
    } else {
        const long long task = (long long)blockIdx.x * RTUS_SOLVE_WAVES + wv;
        task_live = task < q.n_tasks;                                  // wave-uniform; n_tasks = rows * chunks here
        t_row = task_live ? task / q.chunks : 0;
        const int chunk = task_live ? (int)(task - t_row * q.chunks) : 0;
        const int e_raw = chunk * 64 + lane;
        live = task_live && e_raw < q.n_rx;
        e = min(e_raw, q.n_rx - 1);
    }
    const double xe = q.x_rx[e];
    int b0 = -1, b1 = -1, b2 = -1, b3 = -1, cnt = 0;
    auto add_bracket = [&](bool yes, int r) {
        if (yes) {
            b0 = cnt == 0 ? r : b0; b1 = cnt == 1 ? r : b1;
            b2 = cnt == 2 ? r : b2; b3 = cnt == 3 ? r : b3;
            ++cnt;
        }
    };
    if (MASKS && task_live) {
        const double* __restrict__ lrow = q.land_x + (size_t)t_row * n;                                  // per lane: a wave may span rows
        const unsigned long long* __restrict__ mrow = q.pair_mask + (size_t)t_row * q.nb * q.rx_pad + e;
        for (int B0 = 0; B0 < q.nb; B0 += 8) {                          // eight blocks per trip: their loads are in flight together
            unsigned long long m[8];
            double la[8], lb[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int B = min(B0 + i, q.nb - 1);
                m[i] = B0 + i < q.nb ? mrow[(size_t)B * q.rx_pad] : 0ull;
                la[i] = lrow[min(B * 64 + 63, n - 1)]; lb[i] = lrow[min(B * 64 + 64, n - 1)];   // the pair that straddles blocks B, B + 1
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rb = (B0 + i) * 64;
                // set bits in ascending order, until every lane that still has some holds its four brackets (a centred pipe
                // under the centre element lands every ray within rounding of the element: sign noise in half of all pairs)
                while (__any(m[i] != 0ull && cnt < RTUS_MAX_ROOTS)) {
                    const bool has = m[i] != 0ull;
                    add_bracket(has, rb + (int)__builtin_ctzll(has ? m[i] : 1ull));
                    m[i] &= m[i] - 1ull;
                }
                const bool edge = rb + 64 < n && isfinite(la[i]) && isfinite(lb[i]) &&
                                  ((la[i] < xe && xe <= lb[i]) || (la[i] > xe && xe >= lb[i]));
                add_bracket(edge, rb + 63);
            }
        }
    }
    if (!MASKS && task_live) {
        const double* __restrict__ lrow = q.land_x + (size_t)t_row * n;
        // the wave's elements span [xe_lo, xe_hi]; a 64-pair block of grid rays whose landing points all lie
        // outside that span cannot bracket any of them and is skipped (wave-uniform decision, scalar loads)
        double xe_lo = xe, xe_hi = xe;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { xe_lo = fmin(xe_lo, __shfl_xor(xe_lo, o)); xe_hi = fmax(xe_hi, __shfl_xor(xe_hi, o)); }
        const double2* __restrict__ brow = q.land_box + (size_t)t_row * q.nb;
        for (int B = 0; B < q.nb; ++B) {
            // pairs (r - 1, r) for r = max(64 B, 1) .. 64 B + 63: the block's own rays and the last ray of the block before
            const double2 bx = brow[B];
            const int rb = B * 64;
            const double lprev = lrow[max(rb - 1, 0)];
            const double lo = fmin(bx.x, isfinite(lprev) ? lprev : INFINITY), hi = fmax(bx.y, isfinite(lprev) ? lprev : -INFINITY);
            if (lo > xe_hi || hi < xe_lo || !(lo <= hi)) continue;
            double fprev = lprev - xe;
            for (int r0 = max(rb, 1); r0 <= min(rb + 63, n - 1); r0 += 8) {      // 8 grid rays per trip: one batch of scalar loads
                double lx[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) lx[i] = lrow[min(r0 + i, n - 1)];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = r0 + i;
                    const double fcur = lx[i] - xe;
                    const bool in = r < n && r <= rb + 63;
                    const bool st = in && isfinite(fprev) && isfinite(fcur) &&
                                    ((fprev < 0.0 && fcur >= 0.0) || (fprev > 0.0 && fcur <= 0.0));
                    add_bracket(st, r - 1);
                    fprev = in ? fcur : fprev;
                }
            }
        }
    }
    cnt = live ? min(cnt, RTUS_MAX_ROOTS) : 0;
    STAMP(1);

    // ---- B: one list of brackets per workgroup, in (wave, element, bracket) order ---------------------------------------
    // position of a lane's first bracket inside its wave: sum over j of the lanes below it that own a (j+1)-th bracket
    int pre = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        const lanemask mj = __ballot(cnt > j);
        pre += __builtin_amdgcn_mbcnt_hi((unsigned)(mj >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mj, 0));
        tot += __popcll(mj);
    }
    if (lane == 0) wave_tot[wv] = tot;
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) { res_t[threadIdx.x * RTUS_MAX_ROOTS + j] = NAN; res_a[threadIdx.x * RTUS_MAX_ROOTS + j] = NAN; }
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < RTUS_SOLVE_WAVES; ++w) { const int tw = wave_tot[w]; base += w < wv ? tw : 0; total += tw; }
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        if (j < cnt) {
            items[base + pre + j] = (unsigned)threadIdx.x * RTUS_MAX_ROOTS + j;      // the result slot: (wave, lane, bracket)
            item_r[base + pre + j] = j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3));
        }
    }
    __syncthreads();

    STAMP(2);
    // ---- C: lanes = brackets ---------------------------------------------------------------------------------------------
    if constexpr (TRIO) {
    for (int i0 = wv * RTUS_TRIO_PER_WAVE; i0 < total; i0 += RTUS_SOLVE_WAVES * RTUS_TRIO_PER_WAVE) {   // wave-uniform
        const int jb = min(lane / 3, RTUS_TRIO_PER_WAVE - 1), role = lane - 3 * (lane / 3);   // role 0 / 1 / 2 = cand - d / cand / cand + d (lane 63 idles along)
        const bool mine = i0 + jb < total && lane < 3 * RTUS_TRIO_PER_WAVE && role == 1;
        const int ii = min(i0 + jb, total - 1);
        const unsigned slot = items[ii];
        const int br = item_r[ii];
        const int owner = (int)(slot >> 2);
        long long row;
        int ie;
        if (MASKS) {
            const long long fl = (long long)blockIdx.x * EPB + owner;
            row = fl / q.n_rx;
            ie = (int)(fl - row * q.n_rx);
        } else {
            const long long itask = (long long)blockIdx.x * RTUS_SOLVE_WAVES + (owner >> 6);
            row = itask / q.chunks;
            ie = (int)(itask - row * q.chunks) * 64 + (owner & 63);
        }
        const int g = (int)(row / a.n_tx), tx = (int)(row - (long long)g * a.n_tx);
        const double xr = q.x_rx[ie];
        const double* __restrict__ lrow = q.land_x + (size_t)row * n;
        RayIn in;
        in.r_outer = a.geoms[2 * g]; in.off = a.geoms[2 * g + 1];
        in.xa = a.x_a[tx]; in.za = a.z_a[tx]; in.zf = q.z_land;
        bool root;
        double T_fin, x_fin;
        trio_refine<FAST>(q, a, k, lrow, in, xr, br, n, lane, jb, role, tri[wv], root, T_fin, x_fin);
        if (mine && root) { res_t[slot] = T_fin; res_a[slot] = x_fin; }
    }
    } else {
    for (int i0 = wv * 64; i0 < total; i0 += RTUS_SOLVE_TPB) {          // wave-uniform
        const bool mine = i0 + lane < total;
        const int ii = min(i0 + lane, total - 1);                       // idle lanes redo the last bracket (no store)
        const unsigned slot = items[ii];
        const int br = item_r[ii];
        const int owner = (int)(slot >> 2);                             // the thread that found the bracket (RTUS_MAX_ROOTS = 4)
        long long row;
        int ie;
        if (MASKS) {
            const long long fl = (long long)blockIdx.x * EPB + owner;
            row = fl / q.n_rx;
            ie = (int)(fl - row * q.n_rx);
        } else {
            const long long itask = (long long)blockIdx.x * RTUS_SOLVE_WAVES + (owner >> 6);
            row = itask / q.chunks;
            ie = (int)(itask - row * q.chunks) * 64 + (owner & 63);
        }
        const int g = (int)(row / a.n_tx), tx = (int)(row - (long long)g * a.n_tx);
        const double xr = q.x_rx[ie];
        const double* __restrict__ lrow = q.land_x + (size_t)row * n;
        RayIn in;
        in.r_outer = a.geoms[2 * g]; in.off = a.geoms[2 * g + 1];
        in.xa = a.x_a[tx]; in.za = a.z_a[tx]; in.zf = q.z_land;
        double xlo = q.alpha[br], xhi = q.alpha[br + 1];
        double flo = lrow[br] - xr, fhi = lrow[br + 1] - xr;
        // third grid ray for the first candidate: the neighbour on the side of the smaller residual, else the other one
        double xt, ft;
        {
            const bool left = fabs(flo) < fabs(fhi);
            const int t1 = left ? br - 1 : br + 2, t2 = left ? br + 2 : br - 1;
            const int c1 = min(max(t1, 0), n - 1), c2 = min(max(t2, 0), n - 1);
            const double l1 = lrow[c1], l2 = lrow[c2];
            const bool ok1 = c1 == t1 && isfinite(l1), ok2 = c2 == t2 && isfinite(l2);
            xt = q.alpha[ok1 ? c1 : c2];
            ft = (ok1 ? l1 : (ok2 ? l2 : NAN)) - xr;
        }
        // fhi == 0: the grid ray itself is the root; both ends within 1e-13 m already (x_land is flat to rounding there:
        // the centre element over a centred pipe, tangent hits): the end nearer to zero — one evaluation either way
        const bool single = fhi == 0.0 || (fabs(flo) <= 1e-13 && fabs(fhi) <= 1e-13);
        double cand = solve_iqi(xlo, flo, xhi, fhi, xt, ft);             // NaN without a third ray
        if (!(cand > xlo && cand < xhi)) cand = xlo - flo * (xhi - xlo) * solve_rcp(fhi - flo);
        if (!(cand > xlo && cand < xhi)) cand = 0.5 * (xlo + xhi);
        if (single) cand = (fhi == 0.0 || fabs(fhi) < fabs(flo)) ? xhi : xlo;
        double x1 = xhi, f1 = fhi, x2 = xlo, f2 = flo;                   // the two latest points besides the current one
        double xp = NAN, fp = NAN, Tp = NAN;                             // the previous EVALUATED point
        double fprev_abs = INFINITY;
        double x_fin = NAN, T_fin = NAN, f_fin = NAN;
        bool done = false, dead = false;
        int nev = 0;
        for (int it = 0; it < 64; ++it) {
            if (!__any(!done)) break;
            STAMP(3 + min(it, 9));
            // Stragglers.  99.3 % of the brackets are done after two evaluations, but a wave runs until its slowest lane is, and
            // in a small launch (the reference's sweep: 214 waves on 1024 SIMDs) the kernel lasts as long as its slowest wave:
            // four evaluations of ~12 us for a handful of brackets.  From the third evaluation on, an unfinished bracket borrows
            // two FINISHED lanes of its wave: they trace its candidate c moved by -delta / +delta (delta = half the step that led
            // to c), and the bracket finishes from three fresh points around its root — inverse quadratic interpolation for
            // alpha, the parabola through the three travel times for T, error third order in delta — instead of iterating on.
            bool helper = false, rescued = false;
            int hj = 0, hs = 0, rj = 0;
            double ac = cand;                                            // a finished lane re-traces its last point: same bits, ignored
            double delta = 0.0;
            if (it >= 2) {                                               // wave-uniform
                const lanemask U = __ballot(!done), Dm = __ballot(done);
                const int nu = __popcll(U);
                if (nu <= RTUS_RESCUE_MAX && __popcll(Dm) >= 2 * nu) {
                    const int j = __builtin_amdgcn_mbcnt_hi((unsigned)(U >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)U, 0));
                    const int h = __builtin_amdgcn_mbcnt_hi((unsigned)(Dm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Dm, 0));
                    // (round 4: at most 1e-8 rad — x_land has kinks where the returning ray changes chords, and a wide triple that
                    // straddles one is off by delta x the slope's jump: scripts/fuzz_solve.py measured 1.6e-9 rad / 1.8e-12 s with the
                    // round-3 rule, half the step that led here, on 120-point polylines; see trio_refine)
                    delta = fmin(fmin(0.5 * fabs(cand - xp), 1e-8), 0.999 * fmin(cand - xlo, xhi - cand));
                    rescued = !done && delta > 0.0;
                    rj = j;
                    if (!done) {
                        double* r = resc_in[wv][j];
                        r[0] = in.xa; r[1] = in.za; r[2] = in.r_outer; r[3] = in.off; r[4] = cand; r[5] = rescued ? delta : 0.0;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (done && h < 2 * nu) {
                        const double* r = resc_in[wv][h >> 1];
                        if (r[5] > 0.0) {
                            helper = true; hj = h >> 1; hs = h & 1;
                            in.xa = r[0]; in.za = r[1]; in.r_outer = r[2]; in.off = r[3];
                            ac = hs ? r[4] + r[5] : r[4] - r[5];
                        }
                    }
                }
            }
            double sn, cs, px, pz, dz, dx;
            rtus_sincos(ac, sn, cs);                                     // |alpha| < 8: the bounded-range kernels
            lens_eval_sc(k, sn, cs, px, pz, dz, dx);                     // main_rt.py:338, 344 at this lane's alpha
            in.P = make_double2(px, pz);
            { const double rt = rsqrt_fast(dx * dx + dz * dz); in.tu = make_double2(dx * rt, dz * rt); }
            if (!FAST) in.phis = rtus_atan2(dz, dx);
            RayOut o;
            trace_ray<FAST>(a, in, o);                                   // all 64 lanes together
            const double fc = o.x_in - xr;
            const double t1 = seg_time<FAST>(in.xa, in.za, px, pz, k.c1, k.inv_c1);
            const double t2 = seg_time<FAST>(px, pz, o.xq, o.zq, k.c2, k.inv_c2);
            const double t3 = seg_time<FAST>(o.xq, o.zq, o.xi, o.zi, k.c2, k.inv_c2);
            const double t4 = seg_time<FAST>(o.xi, o.zi, o.x_in, q.z_land, k.c1, k.inv_c1);
            const double T = ((t1 + t2) + t3) + t4;
            if (it >= 2) {                                               // wave-uniform
                if (helper) { resc_out[wv][hj][hs][0] = o.x_in; resc_out[wv][hj][hs][1] = T; }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (rescued && isfinite(fc)) {
                    const double fm = resc_out[wv][rj][0][0] - xr, fq = resc_out[wv][rj][1][0] - xr;
                    const double Tm = resc_out[wv][rj][0][1], Tq = resc_out[wv][rj][1][1];
                    // x_land monotone over the three points and (round 4) CONTINUOUS over them: a sign change across a jump of
                    // x_land is no ray path — see trio_refine
                    if ((fq - fc) * (fc - fm) > 0.0 && fc != 0.0 && fabs(fq - fm) <= 2e3 * delta) {
                        const double xr3 = solve_iqi(ac, fc, ac - delta, fm, ac + delta, fq);
                        const double u = (xr3 - ac) * solve_rcp(delta);  // within two delta of the centre, as in trio_refine
                        if (fabs(u) <= 2.0) {
                            ++nev;
                            done = true;
                            f_fin = 0.0;                                 // a root next to three fresh points
                            x_fin = xr3;
                            T_fin = T + 0.5 * u * ((Tq - Tm) + u * ((Tq - T) - (T - Tm)));
                        }
                    } else if (fc == 0.0) { ++nev; done = true; f_fin = 0.0; x_fin = ac; T_fin = T; }
                }
            }
            if (!done) {
                ++nev;
                if (!isfinite(fc)) { dead = true; done = true; }         // the branch ends inside the bracket
                else {
                    // the step the iteration would take next: inverse quadratic interpolation through the three latest
                    // points (secant when that misbehaves) — taken as the next candidate, or, once the residual is small,
                    // NOT taken but applied to alpha and T: what is left is third order in the distances of the three points
                    double c1 = solve_iqi(ac, fc, x1, f1, x2, f2);
                    const double c2 = ac - fc * (ac - x1) * solve_rcp(fc - f1);
                    if (!(fabs(c1 - ac) <= 2.0 * fabs(c2 - ac))) c1 = c2;
                    double step = c1 - ac;
                    step = (fabs(step) <= (nev >= 2 ? 1e-8 : 1e-10) && fc != 0.0) ? step : 0.0;   // NaN -> 0
                    const bool stop = single || fabs(fc) <= 1e-13 || (xhi - xlo) <= 1e-13 ||
                                      (step != 0.0 && fabs(fc) <= (nev >= 2 ? 1e-9 : 1e-11));
                    if (stop) {
                        done = true;
                        f_fin = fc;
                        x_fin = single ? ac : ac + step;
                        // T is smooth in alpha: linear through the two latest evaluations (first evaluation: the step is
                        // below 1e-10 rad, |dT/dalpha| ~ 1e-5 s/rad)
                        T_fin = (nev >= 2 && !single && step != 0.0) ? T + (T - Tp) * solve_rcp(ac - xp) * step : T;
                    } else {
                        if ((fc < 0.0) == (flo < 0.0)) { xlo = ac; flo = fc; } else { xhi = ac; fhi = fc; }
                        if (!(c1 > xlo && c1 < xhi)) c1 = c2;
                        if (!(c1 > xlo && c1 < xhi) || fabs(fc) > 0.5 * fprev_abs) c1 = 0.5 * (xlo + xhi);
                        cand = c1;
                        fprev_abs = fabs(fc);
                        xp = ac; fp = fc; Tp = T;
                        x2 = x1; f2 = f1; x1 = ac; f1 = fc;
                    }
                }
            }
        }
        if (!done) { x_fin = xp; f_fin = fp; T_fin = Tp; }              // iteration cap: the last point evaluated
        const bool root = !dead && fabs(f_fin) < 1e-9;                   // |f| large at convergence: a jump, not a root
        if (mine && root) { res_t[slot] = T_fin; res_a[slot] = x_fin; }
    }
    }
    STAMP(13);
    __syncthreads();
    STAMP(14);

    // ---- D: the element lanes collect their roots (ascending alpha, compacted) ---------------------------------------------
    if (!live) return;
    double tmin = NAN, amin = NAN, tk[RTUS_MAX_ROOTS], ak[RTUS_MAX_ROOTS];
    int nr = 0;
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) { tk[j] = NAN; ak[j] = NAN; }
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        const double T = res_t[threadIdx.x * RTUS_MAX_ROOTS + j], A = res_a[threadIdx.x * RTUS_MAX_ROOTS + j];
        if (T == T) {
            tk[0] = nr == 0 ? T : tk[0]; ak[0] = nr == 0 ? A : ak[0];
            tk[1] = nr == 1 ? T : tk[1]; ak[1] = nr == 1 ? A : ak[1];
            tk[2] = nr == 2 ? T : tk[2]; ak[2] = nr == 2 ? A : ak[2];
            tk[3] = nr == 3 ? T : tk[3]; ak[3] = nr == 3 ? A : ak[3];
            if (!(tmin <= T)) { tmin = T; amin = A; }
            ++nr;
        }
    }
    const size_t o1 = (size_t)t_row * q.n_rx + e;
    q.tt[o1] = tmin;
    if (q.alpha_root) q.alpha_root[o1] = amin;
    if (q.n_roots) q.n_roots[o1] = (uint8_t)nr;
#pragma unroll
    for (int kk = 0; kk < RTUS_MAX_ROOTS; ++kk) {
        if (q.tt_all) q.tt_all[o1 * RTUS_MAX_ROOTS + kk] = tk[kk];
        if (q.alpha_all) q.alpha_all[o1 * RTUS_MAX_ROOTS + kk] = ak[kk];
    }
    STAMP(15);
}

// ---- ONE LAUNCH per pass at the reference's size (round 4) ----------------------------------------------------------------------
// A workgroup of 1,024 threads = ONE ROW (geometry, transmit point) from the grid trace to the stored roots: thread r traces grid
// ray r (n <= 1,024: the reference's sweep has 905), the landing points and the pair masks stay in LDS (no round trip through
// global memory, no kernel boundary between the trace and the refinement: a row starts refining as soon as ITS rays are down,
// not when the slowest wave of the whole grid trace is), the first n_rx <= 128 threads are the element lanes, and all sixteen
// waves refine the row's brackets three lanes per bracket (trio_refine: the same arithmetic — the same bits — as
// rtus_solve_kernel<., ., ., true>).  The polyline and its box records come from rtus_geom1_kernel as before (a separate launch
// the caller skips with RTUS_POLYLINE_READY).
#define RTUS_ROW_TPB 1024
#define RTUS_ROW_WAVES (RTUS_ROW_TPB / 64)
#define RTUS_ROW_MAX_RX 128
template <bool FAST>
__global__ __launch_bounds__(RTUS_ROW_TPB) void rtus_solve_row_kernel(SolveArgs q)
{
    __shared__ double land_s[RTUS_ROW_TPB + 1];
    __shared__ unsigned long long mask_s[RTUS_ROW_TPB / 64][RTUS_ROW_MAX_RX];
    __shared__ unsigned items[RTUS_ROW_MAX_RX * RTUS_MAX_ROOTS];
    __shared__ int item_r[RTUS_ROW_MAX_RX * RTUS_MAX_ROOTS];
    __shared__ double res_t[RTUS_ROW_MAX_RX * RTUS_MAX_ROOTS], res_a[RTUS_ROW_MAX_RX * RTUS_MAX_ROOTS];
    __shared__ int wave_tot[RTUS_ROW_WAVES];
    __shared__ double tri[RTUS_ROW_WAVES][64][2];
    const ShootArgs& a = q.s;
    const LensK& k = a.k;
    const int n = a.n, nb = (n + 63) >> 6;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long row = blockIdx.x;
    const int g = (int)(row / a.n_tx), tx = (int)(row - (long long)g * a.n_tx);
    RayIn in0;
    in0.r_outer = a.geoms[2 * g]; in0.off = a.geoms[2 * g + 1];
    in0.xa = a.x_a[tx]; in0.za = a.z_a[tx]; in0.zf = q.z_land;
    STAMP(0);

    // ---- the grid trace: thread r = ray r (threads past the grid redo its last ray, so that waves stay whole) ----------------------
    {
        const int r_raw = threadIdx.x;
        const bool live = r_raw < n;
        const int r = live ? r_raw : n - 1;
        RayIn in = in0;
        in.P = a.curve[r];
        in.tu = a.tan_u[r];
        if (!FAST) in.phis = a.phi_s[r];
        RayOut o;
        trace_ray<FAST>(a, in, o);
        const double x_in = o.x_in;
        land_s[threadIdx.x] = live ? x_in : NAN;
        if (threadIdx.x == 0) land_s[RTUS_ROW_TPB] = NAN;
        // which elements each pair of consecutive rays brackets: rtus_shoot_kernel<., true>'s emission, into LDS
        const double l0 = live ? x_in : NAN;
        double l1 = __shfl_down(l0, 1);
        l1 = lane == 63 ? NAN : l1;
        const double plo = (l0 == l0 && l1 == l1) ? fmin(l0, l1) : NAN, phi = (l0 == l0 && l1 == l1) ? fmax(l0, l1) : NAN;
        const bool pv = isfinite(plo) && isfinite(phi);
        double wlo = pv ? plo : INFINITY, whi = pv ? phi : -INFINITY;
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) { wlo = fmin(wlo, __shfl_xor(wlo, o2)); whi = fmax(whi, __shfl_xor(whi, o2)); }
        for (int c0 = 0; c0 < q.rx_pad; c0 += 64) {
            const bool ev = c0 + lane < q.n_rx;
            const double xv = q.x_rx[min(c0 + lane, q.n_rx - 1)];
            const double xn = __shfl_down(xv, 1);
            const bool asc = !__ballot(ev && lane < 63 && c0 + lane + 1 < q.n_rx && !(xv <= xn));
            const int nlo = __popcll(__ballot(ev && xv < wlo)), nhi = __popcll(__ballot(ev && xv <= whi));
            const int e0 = asc ? nlo : 0, e1 = asc ? nhi : min(64, q.n_rx - c0);
            unsigned long long mine = 0;
            for (int e = e0; e < e1; ++e) {
                const double x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xv), e), __builtin_amdgcn_readlane(__double2loint(xv), e));
                const lanemask m = __ballot(plo <= x && x <= phi && x != l0);
                mine = lane == e ? m : mine;
            }
            mask_s[wv][c0 + lane] = mine;
        }
    }
    STAMP(1);
    __syncthreads();

    // ---- A: the element lanes (threads 0 .. n_rx - 1) take their brackets from the masks, in ascending alpha -------------------
    const bool live = (int)threadIdx.x < q.n_rx;
    const int e = live ? (int)threadIdx.x : q.n_rx - 1;
    const double xe = q.x_rx[e];
    int b0 = -1, b1 = -1, b2 = -1, b3 = -1, cnt = 0;
    auto add_bracket = [&](bool yes, int r) {
        if (yes) {
            b0 = cnt == 0 ? r : b0; b1 = cnt == 1 ? r : b1;
            b2 = cnt == 2 ? r : b2; b3 = cnt == 3 ? r : b3;
            ++cnt;
        }
    };
    if (wv < (q.n_rx + 63) / 64) {                                       // wave-uniform
        for (int B = 0; B < nb; ++B) {
            unsigned long long m = mask_s[B][e];
            const int rb = B * 64;
            while (__any(m != 0ull && cnt < RTUS_MAX_ROOTS)) {
                const bool has = m != 0ull;
                add_bracket(has, rb + (int)__builtin_ctzll(has ? m : 1ull));
                m &= m - 1ull;
            }
            const double la = land_s[min(rb + 63, n - 1)], lb = land_s[min(rb + 64, n - 1)];    // the pair that straddles blocks B, B + 1
            const bool edge = rb + 64 < n && isfinite(la) && isfinite(lb) && ((la < xe && xe <= lb) || (la > xe && xe >= lb));
            add_bracket(edge, rb + 63);
        }
    }
    cnt = live ? min(cnt, RTUS_MAX_ROOTS) : 0;

    // ---- B: one list of the row's brackets, in (element, bracket) order -------------------------------------------------------------
    int pre = 0, tot = 0;
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        const lanemask mj = __ballot(cnt > j);
        pre += __builtin_amdgcn_mbcnt_hi((unsigned)(mj >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mj, 0));
        tot += __popcll(mj);
    }
    if (lane == 0) wave_tot[wv] = tot;
    if ((int)threadIdx.x < RTUS_ROW_MAX_RX) {
#pragma unroll
        for (int j = 0; j < RTUS_MAX_ROOTS; ++j) { res_t[threadIdx.x * RTUS_MAX_ROOTS + j] = NAN; res_a[threadIdx.x * RTUS_MAX_ROOTS + j] = NAN; }
    }
    __syncthreads();
    int base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < RTUS_ROW_WAVES; ++w) { const int tw = wave_tot[w]; base += w < wv ? tw : 0; total += tw; }
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        if (j < cnt) {
            items[base + pre + j] = (unsigned)threadIdx.x * RTUS_MAX_ROOTS + j;
            item_r[base + pre + j] = j == 0 ? b0 : (j == 1 ? b1 : (j == 2 ? b2 : b3));
        }
    }
    __syncthreads();
    STAMP(2);

    // ---- C: three lanes per bracket, 21 brackets per wave and trip ------------------------------------------------------------------
    for (int i0 = wv * RTUS_TRIO_PER_WAVE; i0 < total; i0 += RTUS_ROW_WAVES * RTUS_TRIO_PER_WAVE) {   // wave-uniform
        const int jb = min(lane / 3, RTUS_TRIO_PER_WAVE - 1), role = lane - 3 * (lane / 3);
        const bool mine = i0 + jb < total && lane < 3 * RTUS_TRIO_PER_WAVE && role == 1;
        const int ii = min(i0 + jb, total - 1);
        const unsigned slot = items[ii];
        const int br = item_r[ii];
        const double xr = q.x_rx[(int)(slot >> 2)];
        bool root;
        double T_fin, x_fin;
        trio_refine<FAST>(q, a, k, land_s, in0, xr, br, n, lane, jb, role, tri[wv], root, T_fin, x_fin);
        if (mine && root) { res_t[slot] = T_fin; res_a[slot] = x_fin; }
    }
    STAMP(13);
    __syncthreads();
    STAMP(14);

    // ---- D: the element lanes collect their roots (ascending alpha, compacted) ------------------------------------------------------
    if (!live) return;
    double tmin = NAN, amin = NAN, tk[RTUS_MAX_ROOTS], ak[RTUS_MAX_ROOTS];
    int nr = 0;
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) { tk[j] = NAN; ak[j] = NAN; }
#pragma unroll
    for (int j = 0; j < RTUS_MAX_ROOTS; ++j) {
        const double T = res_t[threadIdx.x * RTUS_MAX_ROOTS + j], A = res_a[threadIdx.x * RTUS_MAX_ROOTS + j];
        if (T == T) {
            tk[0] = nr == 0 ? T : tk[0]; ak[0] = nr == 0 ? A : ak[0];
            tk[1] = nr == 1 ? T : tk[1]; ak[1] = nr == 1 ? A : ak[1];
            tk[2] = nr == 2 ? T : tk[2]; ak[2] = nr == 2 ? A : ak[2];
            tk[3] = nr == 3 ? T : tk[3]; ak[3] = nr == 3 ? A : ak[3];
            if (!(tmin <= T)) { tmin = T; amin = A; }
            ++nr;
        }
    }
    const size_t o1 = (size_t)row * q.n_rx + e;
    q.tt[o1] = tmin;
    if (q.alpha_root) q.alpha_root[o1] = amin;
    if (q.n_roots) q.n_roots[o1] = (uint8_t)nr;
#pragma unroll
    for (int kk = 0; kk < RTUS_MAX_ROOTS; ++kk) {
        if (q.tt_all) q.tt_all[o1 * RTUS_MAX_ROOTS + kk] = tk[kk];
        if (q.alpha_all) q.alpha_all[o1 * RTUS_MAX_ROOTS + kk] = ak[kk];
    }
}

// Workspace of the solve = shoot workspace + land_x[rows][n] + land intervals[rows][nb].
static size_t sws_land_off(int n) { return align32(shoot_ws_bytes(n)); }
static size_t sws_box_off(int n, int n_geom, int n_tx) { return align32(sws_land_off(n) + (size_t)n_geom * n_tx * (size_t)n * sizeof(double)); }
// after the landing points: the pair masks [rows][nb][rx_pad] (apertures up to RTUS_SOLVE_MASK_MAX_RX) or the landing intervals [rows][nb]
size_t rtus_solve_ws_bytes(int n, int n_geom, int n_tx, int n_rx)
{
    const size_t per_block = n_rx <= RTUS_SOLVE_MASK_MAX_RX ? (size_t)((n_rx + 63) & ~63) * sizeof(unsigned long long) : sizeof(double2);
    return sws_box_off(n, n_geom, n_tx) + (size_t)n_geom * n_tx * (size_t)((n + 63) / 64) * per_block;
}

// Four launches: polyline, box records, grid trace (landing points + which elements each ray pair brackets), refine.
hipError_t rtus_launch_solve(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, int n, const double* x_rx, int n_rx,
                             double z_land, double* tt, double* alpha_root, double* tt_all,
                             double* alpha_all, uint8_t* n_roots, void* ws, unsigned flags, hipStream_t s)
{
    char* w = (char*)ws;
    double* land = (double*)(w + sws_land_off(n));
    const bool masks = n_rx <= RTUS_SOLVE_MASK_MAX_RX;
    // small calls with a grid of <= 1,024 rays and <= 128 elements (the reference's own sweep): ONE launch, a workgroup per row
    const long long rows = (long long)n_geom * n_tx;
    if (rows * n_rx <= 32768 && n <= RTUS_ROW_TPB && n_rx <= RTUS_ROW_MAX_RX && rows <= 0x7fffffffLL && !(flags & (RTUS_SOLVE_ONE_LANE | RTUS_SOLVE_THREE_LAUNCHES))) {
        SolveArgs q;
        ShootArgs& a = q.s;
        a.k = make_lens_k(lens);
        a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = nullptr; a.zf_const = z_land;
        shoot_args_workspace(a, w, n);
        a.out8 = nullptr; a.tof4 = nullptr; a.tof = nullptr; a.land_x = nullptr; a.status = nullptr; a.land_box = nullptr; a.pair_mask = nullptr; a.x_rx = nullptr; a.n_rx = 0; a.rx_pad = 0; a.tof_lazy = 0;
        a.n_tx = n_tx; a.n_geom = n_geom; a.flags = flags;
        if (!(flags & RTUS_POLYLINE_READY)) rtus_launch_geometry_only(a, alpha, s);
        q.alpha = alpha; q.land_x = nullptr; q.land_box = nullptr; q.pair_mask = nullptr; q.rx_pad = (n_rx + 63) & ~63; q.nb = (n + 63) / 64; q.x_rx = x_rx;
        q.z_land = z_land; q.n_rx = n_rx; q.chunks = (n_rx + 63) / 64; q.n_tasks = rows * n_rx;
        q.tt = tt; q.alpha_root = alpha_root; q.tt_all = tt_all; q.alpha_all = alpha_all; q.n_roots = n_roots;
        if (flags & RTUS_SHOOT_FAST_MATH) hipLaunchKernelGGL((rtus_solve_row_kernel<true>), dim3((unsigned)rows), dim3(RTUS_ROW_TPB), 0, s, q);
        else hipLaunchKernelGGL((rtus_solve_row_kernel<false>), dim3((unsigned)rows), dim3(RTUS_ROW_TPB), 0, s, q);
        return hipGetLastError();
    }
    double2* boxes = masks ? nullptr : (double2*)(w + sws_box_off(n, n_geom, n_tx));
    unsigned long long* pmask = masks ? (unsigned long long*)(w + sws_box_off(n, n_geom, n_tx)) : nullptr;
    // 1. grid trace with z_f = z_land for every ray (+ which elements each pair of consecutive rays brackets)
    hipError_t e = rtus_launch_shoot_ex(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, nullptr, z_land, n, nullptr, nullptr,
                                        nullptr, land, nullptr, boxes, pmask, masks ? x_rx : nullptr, masks ? n_rx : 0, ws, flags, s);
    if (e != hipSuccess) return e;
    // 2. bracket + refine
    SolveArgs q;
    ShootArgs& a = q.s;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = nullptr; a.zf_const = z_land;
    shoot_args_workspace(a, w, n);
    a.out8 = nullptr; a.tof4 = nullptr; a.tof = nullptr; a.land_x = nullptr; a.status = nullptr; a.land_box = nullptr; a.pair_mask = nullptr; a.x_rx = nullptr; a.n_rx = 0; a.rx_pad = 0; a.tof_lazy = 0;
    a.n_tx = n_tx; a.n_geom = n_geom;
    a.flags = flags;
    q.alpha = alpha; q.land_x = land; q.land_box = boxes; q.pair_mask = pmask; q.rx_pad = (n_rx + 63) & ~63; q.nb = (n + 63) / 64; q.x_rx = x_rx; q.z_land = z_land; q.n_rx = n_rx;
    q.chunks = (n_rx + 63) / 64;
    q.n_tasks = masks ? (long long)n_geom * n_tx * n_rx : (long long)n_geom * n_tx * q.chunks;
    q.tt = tt; q.alpha_root = alpha_root; q.tt_all = tt_all; q.alpha_all = alpha_all; q.n_roots = n_roots;
    // mask mode, launches that leave SIMDs idle anyway (fewer than two waves of elements per SIMD): 128 element lanes per workgroup
    const bool half = masks && q.n_tasks <= 2LL * 1024 * 64;
    // ... and three lanes per bracket where even that leaves the chip mostly idle (<= 32,768 elements: ~1.5 waves per SIMD with
    // three lanes per bracket): 32 element lanes per workgroup, so that its brackets x 3 lanes fit ONE trip of its four waves (84
    // brackets) even where most elements have two or three — a workgroup that needs a second trip doubles the kernel's critical
    // path (measured with 64 element lanes: the first evaluation of the slowest waves started 31 us into a 43 us kernel)
    const bool trio = rows * n_rx <= 32768 && !(flags & RTUS_SOLVE_ONE_LANE);   // (either bracket-finding path)
    const long long epb = (trio && masks) ? RTUS_SOLVE_TPB / 8 : (half ? RTUS_SOLVE_TPB / 2 : RTUS_SOLVE_TPB);
    const long long blocks = masks ? (q.n_tasks + epb - 1) / epb : (q.n_tasks + RTUS_SOLVE_WAVES - 1) / RTUS_SOLVE_WAVES;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    const dim3 grid((unsigned)blocks);
    const bool fast = (flags & RTUS_SHOOT_FAST_MATH) != 0;
#define RTUS_SOLVE_LAUNCH(F, M, E) hipLaunchKernelGGL((rtus_solve_kernel<F, M, E>), grid, dim3(RTUS_SOLVE_TPB), 0, s, q)
    if (trio && masks) {
        if (fast) hipLaunchKernelGGL((rtus_solve_kernel<true, true, RTUS_SOLVE_TPB / 8, true>), grid, dim3(RTUS_SOLVE_TPB), 0, s, q);
        else hipLaunchKernelGGL((rtus_solve_kernel<false, true, RTUS_SOLVE_TPB / 8, true>), grid, dim3(RTUS_SOLVE_TPB), 0, s, q);
    } else if (trio) {
        if (fast) hipLaunchKernelGGL((rtus_solve_kernel<true, false, RTUS_SOLVE_TPB, true>), grid, dim3(RTUS_SOLVE_TPB), 0, s, q);
        else hipLaunchKernelGGL((rtus_solve_kernel<false, false, RTUS_SOLVE_TPB, true>), grid, dim3(RTUS_SOLVE_TPB), 0, s, q);
    } else if (half) { if (fast) RTUS_SOLVE_LAUNCH(true, true, RTUS_SOLVE_TPB / 2); else RTUS_SOLVE_LAUNCH(false, true, RTUS_SOLVE_TPB / 2); }
    else if (masks) { if (fast) RTUS_SOLVE_LAUNCH(true, true, RTUS_SOLVE_TPB); else RTUS_SOLVE_LAUNCH(false, true, RTUS_SOLVE_TPB); }
    else { if (fast) RTUS_SOLVE_LAUNCH(true, false, RTUS_SOLVE_TPB); else RTUS_SOLVE_LAUNCH(false, false, RTUS_SOLVE_TPB); }
#undef RTUS_SOLVE_LAUNCH
    return hipGetLastError();
}
