// rtus_shoot.hip — forward 4-segment ray trace (reference shoot_rays, main_rt.py:337-405) for
// gfx950: one ray per lane, 64-lane waves walking the lens polyline in lock-step.
//
// Layout in HBM
//   curve   double2[n]      (x_p, z_p) of the alpha grid     — shared by every tx / geometry
//   phi_s   double[n]       atan2(dz, dx) of the lens tangent at alpha[j]
//   blk     double4[n/64]   (xmin, xmax, zmin, zmax) of each 64-point polyline block
//   out8    [n_geom][n_tx][8][n]   SoA per (geometry, tx): every store is a coalesced 512-B row
//
// Crossing search (reference find_line_curve_intersection, main_rt.py:78-99: FIRST index j with
// sign(d_j) != sign(d_{j+1}), d_j = z_p[j] - (m x_p[j] + b)).  The reference scans all n points
// per ray.  Here a wave scans in lock-step: the polyline index is wave-uniform, so polyline
// points arrive by scalar loads (SGPRs, broadcast for free) and each lane only evaluates its
// own line.  A 64-point block is skipped when EVERY lane's line is provably on one side of the
// block's bounding box by more than a rounding margin — the skip cannot change which index is
// found, it only avoids evaluating points whose sign is already certain.
#include "rtus_device.h"

struct ShootArgs {
    LensK k;
    const double* __restrict__ geoms;   // [n_geom][2]
    const double* __restrict__ x_a;     // [n_tx]
    const double* __restrict__ z_a;     // [n_tx]
    const double* __restrict__ z_f;     // [n]
    const double2* __restrict__ curve;  // [n]
    const double* __restrict__ phi_s;   // [n]
    const double4* __restrict__ blk;    // [nblk]
    double* __restrict__ out8;          // nullable
    double* __restrict__ tof4;          // nullable
    double* __restrict__ tof;           // nullable
    double* __restrict__ land_x;        // nullable
    uint8_t* __restrict__ status;       // nullable
    int n, n_tx, n_geom, nblk;
};

// ---- polyline + per-block bounding boxes -----------------------------------------------------
// One wave per 64-point block: lanes compute (x_p, z_p, phi_s) then min/max-reduce across the wave.
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_curve_kernel(LensK k, const double* __restrict__ alpha,
                                                                 int n, double2* __restrict__ curve,
                                                                 double* __restrict__ phi_s,
                                                                 double4* __restrict__ blk)
{
    const int j = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    double x = 0, z = 0, dz, dx;
    const bool live = j < n;
    if (live) {
        lens_eval(k, alpha[j], x, z, dz, dx);       // main_rt.py:338, 344
        curve[j] = make_double2(x, z);
        phi_s[j] = atan2(dz, dx);                   // main_rt.py:272 (tuple branch of refraction)
    }
    double xmin = live ? x : INFINITY, xmax = live ? x : -INFINITY;
    double zmin = live ? z : INFINITY, zmax = live ? z : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, o));
        xmax = fmax(xmax, __shfl_xor(xmax, o));
        zmin = fmin(zmin, __shfl_xor(zmin, o));
        zmax = fmax(zmax, __shfl_xor(zmax, o));
    }
    if ((threadIdx.x & 63) == 0 && live) blk[j >> 6] = make_double4(xmin, xmax, zmin, zmax);
}

// ---- the forward trace -----------------------------------------------------------------------
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_shoot_kernel(ShootArgs a)
{
    const int n = a.n;
    const int r_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = r_raw < n;
    const int r = live ? r_raw : n - 1;            // tail lanes redo the last ray (no stores): keeps waves full
    const int tx = blockIdx.y, g = blockIdx.z;
    const LensK& k = a.k;

    const double r_outer = a.geoms[2 * g], off = a.geoms[2 * g + 1];   // main_rt.py:466-467
    const double xa = a.x_a[tx], za = a.z_a[tx];

    // --- element -> lens, refraction lens -> water (main_rt.py:338-349) -----------------------
    const double2 P = a.curve[r];
    const double phis = a.phi_s[r];
    const double phi_ap = atan2(za - P.y, xa - P.x);                   // :341
    const double phi_pq = refract_angle(phi_ap, phis, k.c2 / k.c1);   // :345
    const double a_pq = tan(phi_pq);                                   // :348
    const double b_pq = P.y - a_pq * P.x;                              // :349

    // --- line ∩ circle, keep the upper root (main_rt.py:351-364) ------------------------------
    const double qA = a_pq * a_pq + 1.0;
    const double qB = 2.0 * (a_pq * b_pq - off);
    const double qC = off * off + b_pq * b_pq - r_outer * r_outer;
    const double sq = sqrt(qB * qB - 4.0 * qA * qC);
    const double den = 2.0 * qA;
    const double xq1 = (-qB + sq) / den, xq2 = (-qB - sq) / den;
    const double zq1 = a_pq * xq1 + b_pq, zq2 = a_pq * xq2 + b_pq;
    const bool upper = zq1 > zq2;
    const double xq = upper ? xq1 : xq2, zq = upper ? zq1 : zq2;

    // --- reflection on the pipe (main_rt.py:367-376); tangent ignores pipe_offset --
    const double slope = -xq / sqrt(r_outer * r_outer - xq * xq);      // :237-238
    const double phi_sl = atan(slope);                                 // :287
    const double phi_l = phi_sl - RTUS_PI_2 - (phi_pq - (phi_sl + RTUS_PI_2));   // :289-291
    const double m = tan(phi_l);                                       // :375
    const double b = zq - m * xq;                                      // :376

    // --- first sign change of d_j along the polyline (main_rt.py:78-99) -----------------------
    const bool fin = isfinite(m) && isfinite(b);   // non-finite line -> the reference ends in (None, None)
    bool found = !fin;
    int idx = -1;
    bool p_lt = false, p_gt = false;
    for (int B = 0; B < a.nblk; ++B) {
        if (__all(found)) break;
        const double4 bb = a.blk[B];               // wave-uniform -> scalar load
        const double t0 = fma(m, bb.x, b), t1 = fma(m, bb.y, b);
        const double tmax = fmax(t0, t1), tmin = fmin(t0, t1);
        const double margin = 2e-8 + 1e-13 * (fabs(t0) + fabs(t1) + fabs(bb.z) + fabs(bb.w));
        const bool cpos = (bb.z - tmax) > margin;  // every d_j in the block > 0
        const bool cneg = (bb.w - tmin) < -margin; // every d_j in the block < 0
        if (__any(!found && !(cpos || cneg))) {
            const int j0 = B * RTUS_CURVE_BLK;
            const int j1 = min(j0 + RTUS_CURVE_BLK, n);
            for (int j = j0; j < j1; ++j) {
                const double2 c = a.curve[j];      // wave-uniform -> scalar load
                const double t = fma(m, c.x, b);
                const bool lt = c.y < t, gt = c.y > t;
                const bool chg = (lt != p_lt) | (gt != p_gt);
                if (!found && j > 0 && chg) { found = true; idx = j - 1; }
                p_lt = lt; p_gt = gt;
            }
        } else if (!found) {
            if (B > 0 && ((cneg != p_lt) | (cpos != p_gt))) { found = true; idx = B * RTUS_CURVE_BLK - 1; }
            p_lt = cneg; p_gt = cpos;
        }
    }

    double xi = NAN, zi = NAN;
    // No sign change anywhere: first polyline point within isclose(d, 0) of the line, else None
    // (main_rt.py:84-96).  Rare; brute-force pass only for waves that need it.
    if (__any(fin && idx < 0)) {
        int on = -1;
        for (int j = 0; j < n; ++j) {
            const double2 c = a.curve[j];
            const double dj = c.y - (m * c.x + b);
            if (on < 0 && fabs(dj) <= 1e-8) on = j;
        }
        if (fin && idx < 0 && on >= 0) { const double2 c = a.curve[on]; xi = c.x; zi = c.y; }
    }
    if (idx >= 0) {                                                    // main_rt.py:106-168
        const double2 c1p = a.curve[idx], c2p = a.curve[idx + 1];
        const double x1 = c1p.x, y1 = c1p.y, x2 = c2p.x, y2 = c2p.y;
        if (np_isclose(x1, x2, 1e-5, 1e-8)) {                          // :110-123 vertical segment
            const double y = m * x1 + b;
            if (y >= fmin(y1, y2) - 1e-9 && y <= fmax(y1, y2) + 1e-9) { xi = x1; zi = y; }
        } else {
            const double m_seg = (y2 - y1) / (x2 - x1);                // :127-128
            const double b_seg = y1 - m_seg * x1;
            if (np_isclose(m, m_seg, 1e-5, 1e-8)) {                    // :131-144
                if (np_isclose(b, b_seg, 1e-5, 1e-8)) { xi = (x1 + x2) / 2.0; zi = m * xi + b; }
            } else {
                const double x = (b_seg - b) / (m - m_seg);            // :147
                const double y = m * x + b;                            // :150
                const double xlo = x1 < x2 ? x1 : x2, xhi = x1 < x2 ? x2 : x1;
                const double ylo = y1 < y2 ? y1 : y2, yhi = y1 < y2 ? y2 : y1;
                if (x >= xlo - 1e-9 && x <= xhi + 1e-9 && y >= ylo - 1e-9 && y <= yhi + 1e-9) {   // :157-158
                    xi = x; zi = y;
                }
            }
        }
    }

    // --- refraction water -> lens and landing on z = z_f (main_rt.py:396-405) ------------------
    const double alpha_i = atan2(xi, zi);                              // :396
    double lx, lz, ldz, ldx;
    lens_eval(k, alpha_i, lx, lz, ldz, ldx);                           // :397 (analytic tangent at the chord point's polar angle)
    const double phi_last = refract_angle(phi_l, atan2(ldz, ldx), k.c1 / k.c2);   // :398
    const double a3 = tan(phi_last);                                   // :401
    const double b3 = zi - a3 * xi;                                    // :402
    const double zf = a.z_f[r];
    const double x_in = (zf - b3) / a3;                                // :404

    if (!live) return;
    const size_t row = (size_t)g * a.n_tx + tx;
    if (a.out8) {
        double* o = a.out8 + row * 8 * (size_t)n + r;
        o[0] = P.x; o[(size_t)n] = P.y; o[2 * (size_t)n] = xq; o[3 * (size_t)n] = zq;
        o[4 * (size_t)n] = xi; o[5 * (size_t)n] = zi; o[6 * (size_t)n] = x_in; o[7 * (size_t)n] = zf;
    }
    if (a.land_x) a.land_x[row * n + r] = x_in;
    if (a.status) a.status[row * n + r] = isnan(xq) ? RTUS_RAY_REF_RAISES : 0;
    if (a.tof4 || a.tof) {
        const double t1 = dist2d(xa, za, P.x, P.y) / k.c1;             // main_compare.py:514
        const double t2 = dist2d(P.x, P.y, xq, zq) / k.c2;             // :515
        const double t3 = dist2d(xq, zq, xi, zi) / k.c2;               // :516
        const double t4 = dist2d(xi, zi, x_in, zf) / k.c1;             // :517
        if (a.tof4) {
            double* t = a.tof4 + row * 4 * (size_t)n + r;
            t[0] = t1; t[(size_t)n] = t2; t[2 * (size_t)n] = t3; t[3 * (size_t)n] = t4;
        }
        if (a.tof) a.tof[row * n + r] = ((t1 + t2) + t3) + t4;         // main_rt.py:497-500
    }
}

// ---- host-side launchers (called from rtus_capi.hip) -----------------------------------------
size_t rtus_ws_curve_off(int) { return 0; }
size_t rtus_ws_phis_off(int n) { return (size_t)n * sizeof(double2); }
size_t rtus_ws_blk_off(int n) { return (((size_t)n * 24) + 31) & ~(size_t)31; }
size_t rtus_ws_bytes(int n) { return rtus_ws_blk_off(n) + (size_t)((n + 63) / 64) * sizeof(double4); }

hipError_t rtus_launch_shoot(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, const double* z_f, int n,
                             double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                             void* ws, hipStream_t s)
{
    char* w = (char*)ws;
    ShootArgs a;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = z_f;
    a.curve = (const double2*)(w + rtus_ws_curve_off(n));
    a.phi_s = (const double*)(w + rtus_ws_phis_off(n));
    a.blk = (const double4*)(w + rtus_ws_blk_off(n));
    a.out8 = out8; a.tof4 = tof4; a.tof = tof; a.land_x = land_x; a.status = status;
    a.n = n; a.n_tx = n_tx; a.n_geom = n_geom; a.nblk = (n + 63) / 64;
    const int gx = (n + RTUS_BLOCK - 1) / RTUS_BLOCK;
    hipLaunchKernelGGL(rtus_curve_kernel, dim3(gx), dim3(RTUS_BLOCK), 0, s, a.k, alpha, n,
                       (double2*)a.curve, (double*)a.phi_s, (double4*)a.blk);
    hipLaunchKernelGGL(rtus_shoot_kernel, dim3(gx, n_tx, n_geom), dim3(RTUS_BLOCK), 0, s, a);
    return hipGetLastError();
}
