/*
 * rtus.h — C ABI of librtus.so, the MI355X (gfx950) travel-time ray tracer.
 *
 * This is the drop-in boundary for ONE hot path of edudscrc/ray-tracing-ultrasound: the
 * ray-tracing loop of main_rt.py (shoot_rays + element matcher) and the element x focal-point
 * travel-time solves built on it.  The reference has no FFI of its own — its boundary is the
 * Python function shoot_rays(x_a, z_a, z_f, alpha, plot) -> dict of 8 float64[N]
 * (main_rt.py:337, 432-441) plus module globals (main_rt.py:449-467).  Every entry point below
 * names the reference lines it replaces; INTEGRATION.md shows the ctypes stub a maintainer adds.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; all functions return 0 (RTUS_OK) or a
 *     negative rtus_status; nothing throws.  The "*_dev" entry points keep no state and are re-entrant per
 *     (device, stream).  The host-buffer twins keep ONE thing: a per-device staging arena (a grow-only device
 *     allocation + a non-blocking stream, created on first use, freed by rtus_release) — the reference's calling
 *     pattern is hundreds of small sequential calls (main_rt.py:464-482), and a hipMalloc per buffer and call cost
 *     more than the kernels.  Host-buffer calls on one device are serialised by that arena's lock.
 *   - all real data is float64 unless the name ends in _f32.
 *   - "*_dev" entry points take DEVICE pointers and a hipStream_t (passed as void*), launch
 *     asynchronously and never allocate or synchronise.  The un-suffixed twins take HOST
 *     buffers, stage them through HBM on `device`, run the same kernels and synchronise.
 *   - invalid rays are NaN (never sentinels), exactly as the reference's consumers expect
 *     (main_compare.py:531-534, main_rt.py:495).
 */
#ifndef RTUS_H
#define RTUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTUS_VERSION 103 /* 0.1.2: rtus_solve_workspace_bytes takes n_rx */

typedef enum rtus_status {
    RTUS_OK = 0,
    RTUS_ERR_INVALID_ARG = -1,   /* null pointer, non-positive size, bad flag (reference: ValueError main_rt.py:24-29) */
    RTUS_ERR_NO_DEVICE = -2,     /* no HIP device / device index out of range */
    RTUS_ERR_HIP = -3,           /* a HIP runtime call failed; rtus_last_hip_error() has the code */
    RTUS_ERR_WORKSPACE = -4,     /* workspace pointer null, not 64-byte aligned, or too small */
    RTUS_ERR_UNSUPPORTED = -5    /* e.g. more layers than RTUS_MAX_LAYERS */
} rtus_status;

/*
 * Acoustic-lens constants.  Replaces the module globals c1, c2, l0, h0, d that the reference's
 * helpers read implicitly (main_rt.py:181-201; set at main_rt.py:449-455).
 */
typedef struct rtus_lens {
    double c1; /* speed in the lens   [m/s]  main_rt.py:449 */
    double c2; /* speed in the water  [m/s]  main_rt.py:450 */
    double l0; /* lens design length  [m]    main_rt.py:453 */
    double h0; /* lens design height  [m]    main_rt.py:454 */
    double d;  /* array plane z = l0+h0 [m]  main_rt.py:455 */
} rtus_lens;

/* Slots of the 8-array result of shoot_rays (main_rt.py:432-441), in this order. */
enum {
    RTUS_LENS_1_X = 0, RTUS_LENS_1_Z = 1, RTUS_PIPE_X = 2, RTUS_PIPE_Z = 3,
    RTUS_LENS_2_X = 4, RTUS_LENS_2_Z = 5, RTUS_TARGET_X = 6, RTUS_TARGET_Z = 7
};

/* Per-ray status bits (optional output). */
#define RTUS_RAY_REF_RAISES 0x1u /* pipe point is NaN: the reference raises LinAlgError here (np.polyfit, main_rt.py:73); we return NaN */

const char *rtus_strerror(int status);
int rtus_version(void);
int rtus_last_hip_error(void);
int rtus_device_count(int *count);
/* Frees the host-buffer twins' staging arena of `device` (all devices: -1).  Optional: a process that exits without
 * calling it leaks nothing the driver does not reclaim. */
int rtus_release(int device);
/* Diagnostic (not in the reference): checks, on `device`, two facts the forward trace relies on — (a) its correctly
 * rounded division / square root sequences return the bits of a / b and sqrt(a) on n_math pseudo-random operand pairs
 * (exponents within +-500, every special value), (b) the depth-first bounding-box records of an n_rays-point lens
 * polyline form a tree (forward skip links that land where each box ends, leaves covering the polyline in order).
 * counts[0] = division mismatches, [1] = square-root mismatches, [2] = structure violations, [3] = n_math.  All zero
 * on a healthy build. */
/* Runs on `device` (validated; the caller's current device is restored) on the staging arena's own stream. */
int rtus_selftest(const rtus_lens *lens, int n_rays, long long n_math, unsigned long long *counts, int device);

/* ------------------------------------------------------------------------------------------
 * Forward trace — replaces shoot_rays (main_rt.py:337-405, 432-441) and its helpers
 * h_from_alpha/dh_from_alpha/x_z_from_alpha/dz_dx_from_alpha (:180-234), refraction/reflection
 * (:267-292), dzdx_pipe (:237-238), roots_bhaskara (:171-177) and the per-ray
 * find_line_curve_intersection loop (:6-168, :384-393), batched over n_geom pipe geometries
 * and n_tx transmit points.  Segment times replace dist + the TOF sums (main_rt.py:444-445,
 * 497-500; main_compare.py:514-517).
 *
 *   geoms   [n_geom][2]  (r_outer, pipe_offset)           main_rt.py:466-467
 *   x_a,z_a [n_tx]       transmit points                  main_rt.py:482
 *   alpha   [n_rays]     launch-angle grid; ALSO defines the lens polyline the reflected line is
 *                        intersected with (main_rt.py:338, 390)
 *   z_f     [n_rays]     landing plane per ray            main_rt.py:404
 *   out8    [n_geom][n_tx][8][n_rays]   nullable          main_rt.py:432-441
 *   tof4    [n_geom][n_tx][4][n_rays]   nullable          main_compare.py:514-517
 *   tof     [n_geom][n_tx][n_rays]      nullable  ((t1+t2)+t3)+t4, order of main_rt.py:497-500
 *   land_x  [n_geom][n_tx][n_rays]      nullable  (= out8 slot RTUS_TARGET_X alone)
 *   status  [n_geom][n_tx][n_rays]      nullable  RTUS_RAY_* bits
 * ---------------------------------------------------------------------------------------- */
/* flags for rtus_shoot*:
 *   0                     REFERENCE-COMPATIBLE.  What that means since round 3 (it is no longer the reference's angle chain
 *                         operation for operation):
 *                         - the reference's laws, quirks and decisions: chord polyline of the alpha grid (main_rt.py:385-390), pipe
 *                           tangent without the offset (:237-238, 367), lens root [1] / upper circle root (:187, 362-364), first sign
 *                           change in index order with np.sign(0) = 0 (:82-99), the +-1e-9 bounds check (:153-168), NaN for
 *                           total reflection and for |x_q| > r, no FMA contraction;
 *                         - formed AS THE REFERENCE FORMS THEM: the lens point and tangent (:180-234), the refracted angle
 *                           phi_pq = phi_s - pi/2 + asin(eta sin theta_1) and a = tan(phi_pq) (:267-280, 348: with the library's own
 *                           atan2 / sin / asin / tan where |a| > 300 — there the quadratic of :349-364 amplifies a last bit by |a|^3),
 *                           the circle intersection in slope-intercept form (:349-364), the chord intersection (:127-150), the landing
 *                           point (:402-404), the four segment times (:497-500), all with correctly rounded division / square root;
 *                         - formed ALGEBRAICALLY (the same real-number value, a few ulp from the reference's chain): sin theta_1 of the
 *                           entry refraction as a dot product of unit vectors; the reflected slope tan(2 atan(s) - phi_pq) as a rational
 *                           form in s's numerator and denominator (:283-292, 375); the exit refraction tan(phi_s - pi/2 + asin(eta
 *                           sin(...))) in vector form (:396-401) — total reflection is decided on eta sin theta_1 > 1 as in the reference,
 *                           but on a value ~1 ulp from its own, so a ray within a few ulp of the critical angle may be NaN on one side
 *                           and finite on the other (parity unpinned AT the critical angle; no golden ray sits there).
 *                         Guarantee, tested: NaN masks identical and every output within 1e-12 m / 1e-15 s on every golden captured from
 *                         the reference (tests/golden, the .npz files, 30 calls), database_2.csv's 13,650 flags and compare.csv's 1,810 rows
 *                         reproduced, and on random inputs |gpu - oracle| <= 1e-12 m + 16 x the oracle's OWN spread under 1-2 ulp
 *                         noise of its trigonometry (scripts/fuzz_shoot.py).  Measured on ill-conditioned rays (near-vertical lines,
 *                         grazing chords — the oracle moves as much under that noise): up to 8e-6 m from the oracle; and
 *                         tests/test_gpu_sweep_random_rows.py: the matcher's hit / first-ray decisions on 2,100 random rows outside
 *                         database_2.csv agree wherever the oracle keeps its own decision under that noise.
 *   RTUS_SHOOT_FAST_MATH  the same laws in vector form throughout (no trigonometry, reciprocal / rsqrt seeds + Newton): ~1e-15
 *                         relative from the above on regular rays, may differ on degenerate ones (exactly vertical / tangent rays). */
#define RTUS_SHOOT_FAST_MATH 0x1u
/* Physically-correct variants — these DEPART from the reference on purpose:
 *   RTUS_TRUE_PIPE_TANGENT  reflect on the tangent of the circle where it actually is; the reference evaluates
 *                           the tangent as if the pipe were centred at x = 0 (main_rt.py:237-238, 367)
 *   RTUS_ANALYTIC_LENS      intersect the reflected line with the analytic lens curve instead of the chords of
 *                           the alpha-grid polyline (main_rt.py:385-390) */
#define RTUS_TRUE_PIPE_TANGENT 0x2u
#define RTUS_ANALYTIC_LENS 0x4u
/* Device entry points only (the host-buffer twins ignore it): the workspace already holds the lens polyline of THIS alpha
 * grid, lens and n_rays — left there by a previous rtus_shoot_dev / rtus_solve_dev call on the same workspace — so the
 * call skips rebuilding it.  The polyline depends on nothing else (main_rt.py:338: x_p, z_p = x_z_from_alpha(alpha)), and
 * the reference's driver traces 210 geometries over one grid (main_rt.py:464-482). */
#define RTUS_POLYLINE_READY 0x8u

/* rtus_shoot (host buffers) keeps the lens polyline of its previous call on the device in a per-device arena and rebuilds it
 * only when alpha's VALUES or the lens constants differ from that call's (compared byte for byte) — the reference's script
 * calls shoot_rays 210 times over one grid (main_rt.py:464-482).  Invisible apart from the time it saves; rtus_release()
 * drops it.
 *
 * Device scratch of one call (polyline, tangents, bounding boxes and their depth-first records): 64-byte aligned
 * (hipMalloc gives 256), rebuilt by every call, not shared between calls that may run concurrently. */
size_t rtus_shoot_workspace_bytes(int n_rays);

int rtus_shoot_dev(const rtus_lens *lens, const double *d_geoms, int n_geom,
                   const double *d_x_a, const double *d_z_a, int n_tx,
                   const double *d_alpha, const double *d_z_f, int n_rays,
                   double *d_out8, double *d_tof4, double *d_tof, double *d_land_x,
                   uint8_t *d_status, void *d_workspace, size_t workspace_bytes, unsigned flags,
                   void *stream);

int rtus_shoot(const rtus_lens *lens, const double *geoms, int n_geom,
               const double *x_a, const double *z_a, int n_tx,
               const double *alpha, const double *z_f, int n_rays,
               double *out8, double *tof4, double *tof, double *land_x, uint8_t *status,
               unsigned flags, int device);

/* ------------------------------------------------------------------------------------------
 * Pulse-echo travel times by ROOT-FINDING: tx -> lens -> pipe -> lens -> rx with x_land(alpha) = x_rx solved
 * per (geometry, tx, rx element) — the north-star's replacement for the reference's grid scan
 * (main_rt.py:479-501: shoot a launch-angle grid, accept the first ray landing within atol of the element).
 * The ray chain is exactly rtus_shoot's (same flags); the alpha grid only brackets the roots and defines the
 * lens polyline.  x_land(alpha) is U-shaped: an element usually has two ray paths.
 *
 *   alpha [n_rays]  bracketing grid, STRICTLY ASCENDING = the lens polyline grid  main_rt.py:479, 338
 *                   (rtus_solve checks it: RTUS_ERR_INVALID_ARG; rtus_solve_dev cannot — device memory)
 *   x_rx  [n_rx]    receive elements on the plane z = z_land                     main_rt.py:469-477
 *   tt        [n_geom][n_tx][n_rx]      least travel time over the element's roots; NaN if none
 *   alpha_root[n_geom][n_tx][n_rx]      nullable: launch angle of that path
 *   tt_all, alpha_all [..][n_rx][RTUS_MAX_ROOTS]  nullable: every root in ascending alpha, NaN padded
 *   n_roots   [n_geom][n_tx][n_rx] uint8 nullable
 * Where the reference's scan reports a hit, one of the roots reproduces its tof to the scan's own resolution
 * (the hit ray lands up to atol + rtol|x| away from the element: <= ~3e-9 s, more at turning points of x_land).
 * ---------------------------------------------------------------------------------------- */
#define RTUS_MAX_ROOTS 4
/* How a bracket is refined depends on the SIZE of the call (a function of n_geom * n_tx * n_rx, nothing else): up to 32,768 (row,
 * element) pairs — the reference's own sweep has 13,650 — three lanes per bracket (the candidate and its two neighbours together:
 * two rounds of evaluations instead of two or three; such a call leaves most of the chip idle and lasts as long as its slowest
 * bracket), beyond that one lane per bracket.  Against a bisection to the last bit (scripts/fuzz_solve.py, 420 random trials, 82,472
 * roots, identical root counts): three lanes |dt| <= 8e-15 s, |dalpha| <= 1e-11 rad + the traces' own landing noise / slope; one
 * lane — it stops once the landing point is within 1e-9 m and its next step below 1e-8 rad, and applies that step — |dt| <= 6e-15 s,
 * |dalpha| <= 5e-11 rad + the same noise term (measured worst 1.9e-11 rad).  Within one scheme a bracket's bits do not
 * depend on the aperture's order or on the other brackets of the call.  RTUS_SOLVE_ONE_LANE asks for the one-lane scheme whatever
 * the size (bits independent of the call's size too). */
#define RTUS_SOLVE_ONE_LANE 0x10u
/* Such a small call with a grid of <= 1,024 rays and <= 128 elements (the reference's sweep: 905, 65) runs as ONE kernel after the
 * polyline's — a workgroup per (geometry, transmit point) row, landing points and pair masks in LDS.  RTUS_SOLVE_THREE_LAUNCHES
 * keeps the grid trace and the refinement as separate launches (the same bits: the refinement's arithmetic is one function). */
#define RTUS_SOLVE_THREE_LAUNCHES 0x20u

size_t rtus_solve_workspace_bytes(int n_rays, int n_geom, int n_tx, int n_rx);

int rtus_solve_dev(const rtus_lens *lens, const double *d_geoms, int n_geom,
                   const double *d_x_a, const double *d_z_a, int n_tx,
                   const double *d_alpha, int n_rays, const double *d_x_rx, int n_rx, double z_land,
                   double *d_tt, double *d_alpha_root, double *d_tt_all, double *d_alpha_all,
                   uint8_t *d_n_roots, void *d_workspace, size_t workspace_bytes, unsigned flags,
                   void *stream);

int rtus_solve(const rtus_lens *lens, const double *geoms, int n_geom,
               const double *x_a, const double *z_a, int n_tx,
               const double *alpha, int n_rays, const double *x_rx, int n_rx, double z_land,
               double *tt, double *alpha_root, double *tt_all, double *alpha_all, uint8_t *n_roots,
               unsigned flags, int device);

/* ------------------------------------------------------------------------------------------
 * Element matcher — replaces the scan of main_rt.py:487-501: for each receive element the FIRST
 * ray (ascending index) with np.isclose(land_x[ray], x_rx[e], rtol, atol), i.e.
 * |land_x - x_rx| <= atol + rtol*|x_rx| for finite operands, land_x == x_rx when either is infinite,
 * never with a NaN (np.isclose's rules).  n_batch = n_geom*n_tx rows.
 *
 *   land_x, tof [n_batch][n_rays]    (outputs of rtus_shoot*)
 *   x_rx        [n_rx]
 *   first_ray   [n_batch][n_rx] int32, -1 when no ray hits   (also the kernel's scratch)
 *   hit         [n_batch][n_rx] uint8  nullable               main_rt.py:496
 *   tof_hit     [n_batch][n_rx]        nullable, 0.0 if none  main_rt.py:493, 497-500
 * rtus_ray_hits: per ray, does ANY element match (main_compare.py:518-521).
 * ---------------------------------------------------------------------------------------- */
int rtus_match_dev(const double *d_land_x, const double *d_tof, int n_batch, int n_rays,
                   const double *d_x_rx, int n_rx, double atol, double rtol,
                   int32_t *d_first_ray, uint8_t *d_hit, double *d_tof_hit, void *stream);

int rtus_match(const double *land_x, const double *tof, int n_batch, int n_rays,
               const double *x_rx, int n_rx, double atol, double rtol,
               int32_t *first_ray, uint8_t *hit, double *tof_hit, int device);

int rtus_ray_hits_dev(const double *d_land_x, int n_batch, int n_rays, const double *d_x_rx,
                      int n_rx, double atol, double rtol, uint8_t *d_ray_hit, void *stream);

int rtus_ray_hits(const double *land_x, int n_batch, int n_rays, const double *x_rx, int n_rx,
                  double atol, double rtol, uint8_t *ray_hit, int device);

/* ------------------------------------------------------------------------------------------
 * Fused sweep — one body of the reference's parameter loop, main_rt.py:464-501: shoot_rays for every
 * (geometry, transmit element) row with the element matcher on its landing points inside the trace
 * kernel (the matcher runs on the landing points while they are still in registers; a small second
 * kernel turns the winners into first_ray / hit / tof_hit).  Results are bit-identical to rtus_shoot*
 * followed by rtus_match* on its land_x / tof outputs.
 *
 *   geoms, x_a, z_a, alpha, z_f, flags   as rtus_shoot*
 *   x_rx [n_rx], atol, rtol               as rtus_match*   (any n_rx; rows * (n_rx rounded up to 64) < 2^31)
 *   first_ray / hit / tof_hit [n_geom*n_tx][n_rx]   as rtus_match*   (hit, tof_hit nullable)
 *   tof, land_x [n_geom*n_tx][n_rays]     nullable: the per-ray arrays, stored only when asked for.  Without `tof` (what the
 *                                         reference's loop needs: main_rt.py:497-500 sums the segments of the HIT ray only) the
 *                                         four segment times are worked out only in waves that matched an element: 8 % less
 *                                         time at 8 M rays, the same tof_hit bits
 *   workspace  rtus_sweep_workspace_bytes(n_rays, n_geom, n_tx, n_rx) bytes, 64-byte aligned.
 *              RTUS_POLYLINE_READY here means: the previous call on this workspace was rtus_sweep_dev
 *              with the same alpha, n_rays, n_geom, n_tx and n_rx (the lens polyline is kept AND the
 *              matcher's scratch was left idle by that call).
 * ---------------------------------------------------------------------------------------- */
size_t rtus_sweep_workspace_bytes(int n_rays, int n_geom, int n_tx, int n_rx);

int rtus_sweep_dev(const rtus_lens *lens, const double *d_geoms, int n_geom,
                   const double *d_x_a, const double *d_z_a, int n_tx,
                   const double *d_alpha, const double *d_z_f, int n_rays,
                   const double *d_x_rx, int n_rx, double atol, double rtol,
                   int32_t *d_first_ray, uint8_t *d_hit, double *d_tof_hit,
                   double *d_tof, double *d_land_x,
                   void *d_workspace, size_t workspace_bytes, unsigned flags, void *stream);

int rtus_sweep(const rtus_lens *lens, const double *geoms, int n_geom,
               const double *x_a, const double *z_a, int n_tx,
               const double *alpha, const double *z_f, int n_rays,
               const double *x_rx, int n_rx, double atol, double rtol,
               int32_t *first_ray, uint8_t *hit, double *tof_hit,
               double *tof, double *land_x, unsigned flags, int device);

/* ------------------------------------------------------------------------------------------
 * Element x focal-point Fermat travel times through horizontal layers (BASELINE configs 2, 3, 5).
 * NOT IN THE REFERENCE (it has no planar interfaces) — the build's own solver, "parity
 * unpinned"; it replaces nothing in main_rt.py and is checked against oracle/ + closed forms.
 *
 *   z_if [n_if]    interface depths, strictly ascending (host memory, copied into kernel args)
 *   c    [n_if+1]  speeds; c[i] applies between interface i-1 and interface i
 *   xe,ze [n_e]    sources (elements), ze < z_if[0]
 *   xf,zf [n_f]    targets (focal points), zf > ze; the path stops in whichever layer holds zf
 *   tt   [n_e][n_f] travel times [s]; NaN where zf <= ze
 *   iters [n_e][n_f] uint8 nullable: Newton iterations used (diagnostic)
 *
 * rtus_tt_layers_batch_dev: n_batch independent problems of one shape and one medium in ONE launch — several
 * apertures and / or several target sets, the way the reference's driver sweeps 210 geometries (main_rt.py:464-467).
 * Problem b reads d_xe/d_ze + b*e_stride, d_xf/d_zf + b*f_stride and writes d_tt + b*t_stride (strides in doubles;
 * a stride of 0 shares that input between all problems; t_stride >= n_e*n_f).  A small problem (BASELINE config 2 is a
 * single round of 4 waves per SIMD) no longer pays a launch ramp and drain of its own.
 * ---------------------------------------------------------------------------------------- */
#define RTUS_MAX_LAYERS 8
/* Accuracy tier of the planar solver (flags of the *_ex entries, rtus_tt_layers_rows_dev, rtus_tt_layers_sorted_dev).  0: the travel time from an fp64
 * evaluation at the Newton iterate + its second-order Fermat expansion in an fp64 residual: <= 1e-13 relative (measured 3e-18 s
 * on BASELINE config 3).  RTUS_TT_TAUP_TAIL: from the tau-p form T = p X + sum (h_i/c_i) cos(theta_i), stationary in p, with the
 * second-order term from the fp32 residual (and, on a uniform pitch where the solution moves slowly from element to element, with
 * the term's coefficient carried over groups of four elements): <= 6e-11 relative (~2e-15 s) at worst, ~3e-14 typically — six orders
 * inside the 1e-9 s bar of the north-star.  (Held to that bar on random, coarse and irregular apertures by scripts/fuzz_layers.py
 * --taup; round 3's version of this tier exceeded it — 2.8e-10 relative — on coarse random pitches.) */
#define RTUS_TT_TAUP_TAIL 0x1u

int rtus_tt_layers_dev(const double *z_if, const double *c, int n_if,
                       const double *d_xe, const double *d_ze, int n_e,
                       const double *d_xf, const double *d_zf, int n_f,
                       double *d_tt, uint8_t *d_iters, void *stream);

int rtus_tt_layers_batch_dev(const double *z_if, const double *c, int n_if,
                             const double *d_xe, const double *d_ze, int n_e, long long e_stride,
                             const double *d_xf, const double *d_zf, int n_f, long long f_stride,
                             double *d_tt, long long t_stride, int n_batch, void *stream);

int rtus_tt_layers(const double *z_if, const double *c, int n_if,
                   const double *xe, const double *ze, int n_e,
                   const double *xf, const double *zf, int n_f,
                   double *tt, uint8_t *iters, int device);

/* The same entries with the accuracy tier as an argument (flags: 0 or RTUS_TT_TAUP_TAIL; the un-suffixed entries above are
 * flags = 0).  iters is a diagnostic of the default tier's kernel: RTUS_ERR_INVALID_ARG with RTUS_TT_TAUP_TAIL.  bench.py's
 * headline times RTUS_TT_TAUP_TAIL; every caller — device pointers, host buffers, batches, several GPUs (below) — can ask for it. */
int rtus_tt_layers_ex_dev(const double *z_if, const double *c, int n_if,
                          const double *d_xe, const double *d_ze, int n_e,
                          const double *d_xf, const double *d_zf, int n_f,
                          double *d_tt, uint8_t *d_iters, unsigned flags, void *stream);
int rtus_tt_layers_batch_ex_dev(const double *z_if, const double *c, int n_if,
                                const double *d_xe, const double *d_ze, int n_e, long long e_stride,
                                const double *d_xf, const double *d_zf, int n_f, long long f_stride,
                                double *d_tt, long long t_stride, int n_batch, unsigned flags, void *stream);
int rtus_tt_layers_ex(const double *z_if, const double *c, int n_if,
                      const double *xe, const double *ze, int n_e,
                      const double *xf, const double *zf, int n_f,
                      double *tt, uint8_t *iters, unsigned flags, int device);

/* The aperture in ANY order.  The kernel starts each solve from the four previous elements of its workgroup's block, which pays
 * (1 evaluation instead of ~4.5) when consecutive elements are neighbours in space at one depth.  rtus_tt_layers_dev takes the
 * elements as they come; this entry sorts them on the device by (depth, position) first and stores each row where it belongs
 * (d_workspace: rtus_tt_layers_sort_workspace_bytes(n_e) bytes, 256-byte aligned; n_e <= 32768).  The result of an element does
 * not depend on the order the aperture was handed over in.  The host-buffer twin rtus_tt_layers does the same by itself (it sorts
 * on the host); flags: RTUS_TT_TAUP_TAIL or 0. */
size_t rtus_tt_layers_sort_workspace_bytes(int n_e);
int rtus_tt_layers_sorted_dev(const double *z_if, const double *c, int n_if,
                              const double *d_xe, const double *d_ze, int n_e,
                              const double *d_xf, const double *d_zf, int n_f,
                              double *d_tt, void *d_workspace, size_t workspace_bytes, unsigned flags, void *stream);

/* ------------------------------------------------------------------------------------------
 * Element x focal-point Fermat travel times through the reference's CURVED lens surface
 * (BASELINE config 4).  Interface = P(alpha) = h(alpha)(sin alpha, cos alpha) with h, dh/dalpha of
 * main_rt.py:180-214 (x_z_from_alpha :217-223, dz_dx_from_alpha :226-234); element in the lens
 * (c1), target in the water (c2).  The reference has no two-point solver (it only shoots rays from a
 * launch-angle grid, main_rt.py:479-482), so as a SOLVER this is the build's own; it is pinned to the
 * reference through Fermat <=> Snell: for a forward-traced ray, (A, F = pipe point) must give back
 * that ray's alpha and tof_1 + tof_2 (main_compare.py:514-515).
 *
 *   alpha_lo/hi    search interval for the refraction point's polar angle (e.g. -/+ alpha_max, main_rt.py:457)
 *   xe,ze [n_e]    elements;  xf,zf [n_f]  targets
 *   tt [n_e][n_f]  travel times;  alpha_out [n_e][n_f] nullable: polar angle of the refraction point
 * The entry is the LEAST time over [alpha_lo, alpha_hi] (Fermat).  The lens is aplanatic: around its focus T(alpha) is nearly flat
 * and has two local minima (an interior ray and an end of the interval, or both ends beyond the focus); the kernel follows one
 * minimum from element to element and looks at the whole interval wherever a second one can exist — a target pinned at an end, or
 * d2T/dalpha2 at the minimum below 0.125 h0 / c2 (7.5e-6 s/rad^2 for the reference lens: twice the largest value measured at an
 * interior minimum of a pair with two minima, scripts/study_lens_minima.py).  That threshold is CALIBRATED ON THE REFERENCE LENS
 * (main_rt.py:449-457) and scaled with the lens's own time h0 / c2; for another lens design check it with the same study.
 * ---------------------------------------------------------------------------------------- */
int rtus_tt_lens_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                     const double *d_xe, const double *d_ze, int n_e,
                     const double *d_xf, const double *d_zf, int n_f,
                     double *d_tt, double *d_alpha_out, void *stream);

int rtus_tt_lens(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                 const double *xe, const double *ze, int n_e,
                 const double *xf, const double *zf, int n_f,
                 double *tt, double *alpha_out, int device);

int rtus_tt_lens_f32_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                         const float *d_xe, const float *d_ze, int n_e,
                         const float *d_xf, const float *d_zf, int n_f,
                         float *d_tt, float *d_alpha_out, void *stream);

int rtus_tt_lens_f32(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                     const float *xe, const float *ze, int n_e,
                     const float *xf, const float *zf, int n_f,
                     float *tt, float *alpha_out, int device);

/* ------------------------------------------------------------------------------------------
 * Row shards of a travel-time table, and one call spread over several GPUs (SURVEY 8(b) "rtus_allgather(...)/multi-GPU
 * variant", 8(e)).  NOT IN THE REFERENCE (it has no device or process boundary at all).
 *
 * The table kernels solve `rows_per_block` consecutive elements per workgroup and start each solve from its predecessors
 * in the block.  rows_per_block is a function of the WHOLE table (rtus_table_rows_per_block), and the *_rows_dev entry
 * points keep workgroups aligned to the whole table's blocks: a shard [row0, row0 + n_rows) whose row0 is a multiple of
 * rows_per_block gets, bit for bit, the rows the one-launch table has (an unaligned shard is still correct to the solver's
 * accuracy; its first partial block has fewer predecessors).  rtus_shard_rows = rows per shard for n_shards equal shards,
 * rounded up to that multiple.
 *
 * rtus_*_multi: host buffers in, host table out, the rows spread over `devices[0 .. n_dev)` (a device may be listed more
 * than once: one arena and stream per entry); every device copies its block straight into the caller's rows — a host
 * result needs no exchange between the GPUs.
 * rtus_*_multi_dev: device i holds its own copies of the inputs (d_xe[i], d_ze[i]: ALL n_e elements; d_xf[i], d_zf[i]) and
 * of the padded table d_tt[i] [n_dev * rtus_shard_rows(...)][n_f]; it solves its row block in place on streams[i].
 * gather = 0 leaves the table sharded; gather = 1 reassembles it on every device by an in-place ncclAllGather over
 * single-process communicators (ncclCommInitAll; RCCL over xGMI, bound with dlopen at the first such call —
 * RTUS_ERR_UNSUPPORTED when librccl cannot be loaded or the communicators cannot be made).  Asynchronous.
 * ---------------------------------------------------------------------------------------- */
int rtus_table_rows_per_block(long long n_rows_total, int n_f, int elem_bytes);              /* elem_bytes: 8 (fp64) or 4 (fp32) */
long long rtus_shard_rows(long long n_rows_total, int n_f, int elem_bytes, int n_shards);

int rtus_tt_layers_rows_dev(const double *z_if, const double *c, int n_if,
                            const double *d_xe, const double *d_ze, int n_rows, long long row0, long long n_rows_total,
                            const double *d_xf, const double *d_zf, int n_f, double *d_tt, unsigned flags, void *stream);
int rtus_tt_lens_rows_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                          const double *d_xe, const double *d_ze, int n_rows, long long row0, long long n_rows_total,
                          const double *d_xf, const double *d_zf, int n_f, double *d_tt, double *d_alpha_out, void *stream);
int rtus_tt_lens_f32_rows_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                              const float *d_xe, const float *d_ze, int n_rows, long long row0, long long n_rows_total,
                              const float *d_xf, const float *d_zf, int n_f, float *d_tt, float *d_alpha_out, void *stream);

/* Diagnostic: the rows of rtus_tt_lens[_f32]_rows_dev (no alpha output) plus HOW they were solved, added to d_stats (device memory,
 * 5 x uint64, zeroed by the caller), counted in wave-elements (64 targets of one row): [0] T alone at the extrapolated start,
 * [1] one evaluation of T and dT/dalpha, [2] the safeguarded iteration, [3] of those: with a look at the whole search interval
 * (a target pinned at an end of it, or T nearly flat in alpha — around the lens focus two minima compete), [4] evaluations in [2]. */
int rtus_tt_lens_stats_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                           const double *d_xe, const double *d_ze, int n_rows, long long row0, long long n_rows_total,
                           const double *d_xf, const double *d_zf, int n_f, double *d_tt, unsigned long long *d_stats, void *stream);
int rtus_tt_lens_f32_stats_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                               const float *d_xe, const float *d_ze, int n_rows, long long row0, long long n_rows_total,
                               const float *d_xf, const float *d_zf, int n_f, float *d_tt, unsigned long long *d_stats, void *stream);

int rtus_tt_layers_multi(const double *z_if, const double *c, int n_if,
                         const double *xe, const double *ze, int n_e, const double *xf, const double *zf, int n_f,
                         double *tt, const int *devices, int n_dev);
int rtus_tt_lens_f32_multi(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                           const float *xe, const float *ze, int n_e, const float *xf, const float *zf, int n_f,
                           float *tt, const int *devices, int n_dev);

/* ... with the accuracy tier (flags: 0 or RTUS_TT_TAUP_TAIL).  rtus_tt_layers_multi[_ex] sorts the aperture by (depth, position) on
 * the host first, as rtus_tt_layers does: the table is the one-device table bit for bit whatever order the elements come in. */
int rtus_tt_layers_multi_ex(const double *z_if, const double *c, int n_if,
                            const double *xe, const double *ze, int n_e, const double *xf, const double *zf, int n_f,
                            double *tt, const int *devices, int n_dev, unsigned flags);
int rtus_tt_layers_multi_ex_dev(const double *z_if, const double *c, int n_if,
                                const double *const *d_xe, const double *const *d_ze, int n_e,
                                const double *const *d_xf, const double *const *d_zf, int n_f, double *const *d_tt,
                                const int *devices, int n_dev, void *const *streams, int gather, unsigned flags);

int rtus_tt_layers_multi_dev(const double *z_if, const double *c, int n_if,
                             const double *const *d_xe, const double *const *d_ze, int n_e,
                             const double *const *d_xf, const double *const *d_zf, int n_f, double *const *d_tt,
                             const int *devices, int n_dev, void *const *streams, int gather);
int rtus_tt_lens_f32_multi_dev(const rtus_lens *lens, double alpha_lo, double alpha_hi,
                               const float *const *d_xe, const float *const *d_ze, int n_e,
                               const float *const *d_xf, const float *const *d_zf, int n_f, float *const *d_tt,
                               const int *devices, int n_dev, void *const *streams, int gather);

/* ------------------------------------------------------------------------------------------
 * Consumers of a travel-time table.  NOT IN THE REFERENCE, which stops at the travel times
 * (main_rt.py:497-504); checked against a NumPy restatement on synthetic point-scatterer data.
 *
 * rtus_focal_delays: transmit focal law, delays[e][f] = max over e' of tt[e'][f] - tt[e][f] — what element e must wait so
 *   that all wavefronts reach focal point f together.  NaN (no ray path) is ignored by the maximum and stays NaN.
 *   d_delays may be d_tt (in place).
 * rtus_tfm: total-focusing-method delay-and-sum over full-matrix-capture data,
 *   image[f] = sum over (tx, rx) of fmc[tx][rx][.] linearly interpolated at the sample position
 *   s = (tt_tx[tx][f] + tt_rx[rx][f] - t0) * fs, i.e. (1 - w) sample[i] + w sample[i + 1] with i = floor(s), w = s - i.  A
 *   position before the record (s < 0) or at / past its end (s >= n_t) contributes nothing; in [n_t - 1, n_t) the missing
 *   sample n_t counts as zero.  A pair without a ray path (NaN — or any non-finite or absurd travel time, |s| >= 1e8)
 *   contributes nothing.  (oracle/tfm_numpy.py is the same definition.)
 *     fmc     [n_tx][n_rx][n_t] float32 A-scans, fs samples per second, first sample at time t0
 *     tt_tx   [n_tx][n_f], tt_rx [n_rx][n_f]  travel times (rtus_tt_layers* / rtus_tt_lens outputs; may be one table)
 *     image   [n_f] float32
 * ---------------------------------------------------------------------------------------- */
int rtus_focal_delays_dev(const double *d_tt, int n_e, int n_f, double *d_delays, void *stream);
int rtus_focal_delays(const double *tt, int n_e, int n_f, double *delays, int device);

int rtus_tfm_dev(const float *d_fmc, int n_tx, int n_rx, int n_t, double fs, double t0,
                 const double *d_tt_tx, const double *d_tt_rx, int n_f, float *d_image, void *stream);
int rtus_tfm(const float *fmc, int n_tx, int n_rx, int n_t, double fs, double t0,
             const double *tt_tx, const double *tt_rx, int n_f, float *image, int device);

#ifdef __cplusplus
}
#endif
#endif /* RTUS_H */
