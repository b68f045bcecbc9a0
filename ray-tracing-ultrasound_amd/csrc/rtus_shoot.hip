// rtus_shoot.hip — forward 4-segment ray trace (reference shoot_rays, main_rt.py:337-405) for
// gfx950: one ray per lane, 64-lane waves walking the lens polyline in lock-step.
//
// Layout in HBM (workspace, built once per call by rtus_curve_kernel + rtus_tree_kernel; shared by every tx / geometry)
//   curve   double2[n]        (x_p, z_p) of the alpha grid
//   phi_s   double[n]         atan2(dz, dx) of the lens tangent at alpha[j]
//   tan_u   double2[n]        the same tangent as a unit vector (fast-math mode)
//   node0   double4[n/8]      bounding box (xc, xh, zc, zh = centre / half-extent) of 8 polyline points
//   node1   double4[n/64]     ... of 64 points
//   node2   double4[n/512]    ... of 512 points
//   tree    TreeNode[n0+n1+n2+n3+1]  the same boxes (+ 4096-point boxes on long polylines) in depth-first order with
//                             skip links: what the crossing search walks
//   out8    [n_geom][n_tx][8][n]   SoA per (geometry, tx): every store is a coalesced 512-B row
//
// Crossing search (reference find_line_curve_intersection, main_rt.py:78-99: FIRST index j with
// sign(d_j) != sign(d_{j+1}), d_j = z_p[j] - (m x_p[j] + b)).  The reference scans all n points
// per ray (90 % of its run time).  Here a wave scans in lock-step: the polyline index is
// wave-uniform, so polyline points and boxes arrive by scalar loads (SGPRs, broadcast for free)
// and each lane only evaluates its own line.  A box (4096 -> 512 -> 64 -> 8 points) is skipped when EVERY
// lane's line is provably on one side of it by more than a rounding margin; otherwise the wave
// descends, and only 8-point leaves are evaluated point by point.  Skipping cannot change which
// index is found — it only avoids evaluating points whose sign is already certain — so the result
// is the reference's "first sign change in index order", np.sign(0) = 0 semantics included.
#include "rtus_trace.h"
#include <vector>

#define RTUS_CURVE_TPB 512      // curve kernel: one workgroup = one 512-point node2

__device__ __forceinline__ double4 make_box(double xmin, double xmax, double zmin, double zmax)
{
    // centre / half-extent, half-extents nudged up so the box still contains its points after rounding
    const double xc = 0.5 * (xmin + xmax), zc = 0.5 * (zmin + zmax);
    const double xh = fmax(xmax - xc, xc - xmin) * (1.0 + 1e-15), zh = fmax(zmax - zc, zc - zmin) * (1.0 + 1e-15);
    return make_double4(xc, xh, zc, zh);
}

// ---- polyline + bounding-box hierarchy -------------------------------------------------------
// One polyline point per thread; `part` = which 512-point box (node2) the thread's group of 512 threads builds, `t` = the thread's
// index inside it.  Called by every thread of the group (it contains a barrier).
__device__ __forceinline__ void curve_part(const LensK& k, const double* __restrict__ alpha, int n, double2* __restrict__ curve,
                                           double* __restrict__ phi_s, double2* __restrict__ tan_u, double4* node0,
                                           double4* node1, double4* node2, int part, int t,
                                           double (*red)[RTUS_CURVE_TPB / 64])
{
    const int j = part * RTUS_CURVE_TPB + t;
    double x = 0, z = 0, dz, dx;
    const bool live = j < n;
    if (live) {
        lens_eval(k, alpha[j], x, z, dz, dx);       // main_rt.py:338, 344
        curve[j] = make_double2(x, z);
        if (j == n - 1) for (int q = n; q < ((n + 7) & ~7); ++q) curve[q] = make_double2(x, z);   // leaf padding
        phi_s[j] = atan2(dz, dx);                   // main_rt.py:272 (tuple branch of refraction)
        const double rt = 1.0 / sqrt(dx * dx + dz * dz);
        tan_u[j] = make_double2(dx * rt, dz * rt);
    }
    double xmin = live ? x : INFINITY, xmax = live ? x : -INFINITY;
    double zmin = live ? z : INFINITY, zmax = live ? z : -INFINITY;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        xmin = fmin(xmin, __shfl_xor(xmin, o));
        xmax = fmax(xmax, __shfl_xor(xmax, o));
        zmin = fmin(zmin, __shfl_xor(zmin, o));
        zmax = fmax(zmax, __shfl_xor(zmax, o));
        if (o == 4 && (t & 7) == 0 && live) node0[j >> 3] = make_box(xmin, xmax, zmin, zmax);
    }
    const int lane = t & 63, wv = t >> 6;
    if (lane == 0) {
        if (live) node1[j >> 6] = make_box(xmin, xmax, zmin, zmax);
        red[0][wv] = xmin; red[1][wv] = xmax; red[2][wv] = zmin; red[3][wv] = zmax;
    }
    __syncthreads();
    if (t == 0 && part * RTUS_CURVE_TPB < n) {
        for (int w = 1; w < RTUS_CURVE_TPB / 64; ++w) {
            xmin = fmin(xmin, red[0][w]); xmax = fmax(xmax, red[1][w]);
            zmin = fmin(zmin, red[2][w]); zmax = fmax(zmax, red[3][w]);
        }
        node2[part] = make_box(xmin, xmax, zmin, zmax);
    }
}

__global__ __launch_bounds__(RTUS_CURVE_TPB) void rtus_curve_kernel(LensK k, const double* __restrict__ alpha,
                                                                     int n, double2* __restrict__ curve,
                                                                     double* __restrict__ phi_s,
                                                                     double2* __restrict__ tan_u,
                                                                     double4* __restrict__ node0,
                                                                     double4* __restrict__ node1,
                                                                     double4* __restrict__ node2, int32_t* __restrict__ m_init, int m_init_n)
{
    __shared__ double red[4][RTUS_CURVE_TPB / 64];
    // (the fused sweep's matcher scratch rides along, as in rtus_geom1_kernel)
    for (int i = blockIdx.x * RTUS_CURVE_TPB + threadIdx.x; i < m_init_n; i += gridDim.x * RTUS_CURVE_TPB) m_init[i] = RTUS_NO_RAY;
    curve_part(k, alpha, n, curve, phi_s, tan_u, node0, node1, node2, blockIdx.x, threadIdx.x, red);
}

__device__ __forceinline__ void tree_record(const double4* node0, const double4* node1, const double4* node2, int n0, int n1, int n2,
                                            int n3, TreeNode* tree, int id)
{
    const int total = n0 + n1 + n2 + n3;
    if (id >= total) return;
    int L, k;
    const double4* src;
    if (id < n0) { L = 0; k = id; src = node0; }
    else if (id < n0 + n1) { L = 1; k = id - n0; src = node1; }
    else if (id < n0 + n1 + n2) { L = 2; k = id - n0 - n1; src = node2; }
    else { L = 3; k = id - n0 - n1 - n2; src = nullptr; }
    const int top = n3 > 0 ? 3 : 2;
    const int sz[4] = {1, 9, 73, 585};
    int pos = (k >> (3 * (top - L))) * sz[top];
    for (int l = top - 1; l >= L; --l) pos += 1 + ((k >> (3 * (l - L))) & 7) * sz[l];
    double4 b;
    if (L < 3) b = src[k];
    else {                              // a 4096-point box (long polylines only): the union of eight 512-point boxes
        double xmin = INFINITY, xmax = -INFINITY, zmin = INFINITY, zmax = -INFINITY;
        for (int j = k * 8; j < min(k * 8 + 8, n2); ++j) {
            const double4 c = node2[j];
            xmin = fmin(xmin, c.x - c.y); xmax = fmax(xmax, c.x + c.y);
            zmin = fmin(zmin, c.z - c.w); zmax = fmax(zmax, c.z + c.w);
        }
        b = make_box(xmin, xmax, zmin, zmax);
    }
    TreeNode t;
    t.xc = b.x; t.xh = b.y; t.zc = b.z; t.zh = b.w;
    t.j0 = k << (3 * (L + 1)); t.j1 = t.j0 + (8 << (3 * L));
    t.skip_off = (unsigned)min(pos + sz[L], total) * (unsigned)sizeof(TreeNode);
    t.leafm = L == 0 ? ~0ull : 0ull;
    t.pad[0] = t.pad[1] = t.pad[2] = 0;
    tree[pos] = t;
    if (pos == 0) {
        TreeNode e = t;
        e.xc = 0.0; e.xh = 0.0;
        for (int q = 0; q < n2; ++q) {
            const double4 bx = node2[q];
            e.xc = fmax(e.xc, fabs(bx.x) + bx.y);
            e.xh = fmax(e.xh, fabs(bx.z) + bx.w);
        }
        tree[total] = e;
    }
}

__global__ void rtus_tree_kernel(const double4* __restrict__ node0, const double4* __restrict__ node1,
                                 const double4* __restrict__ node2, int n0, int n1, int n2, int n3,
                                 TreeNode* __restrict__ tree)
{
    tree_record(node0, node1, node2, n0, n1, n2, n3, tree, blockIdx.x * blockDim.x + threadIdx.x);
}

// Polylines of up to 1024 points (the reference's 905): polyline, boxes and records by ONE workgroup in ONE launch — a launch
// of a few hundred threads costs ~5 us whatever it does, and the reference's calling pattern is many small calls.
#define RTUS_GEOM1_TPB (2 * RTUS_CURVE_TPB)
__global__ __launch_bounds__(RTUS_GEOM1_TPB) void rtus_geom1_kernel(LensK k, const double* __restrict__ alpha, int n,
                                                                     double2* __restrict__ curve, double* __restrict__ phi_s,
                                                                     double2* __restrict__ tan_u, double4* node0, double4* node1,
                                                                     double4* node2, int n0, int n1, int n2, TreeNode* tree,
                                                                     int32_t* m_init, int m_init_n)
{
    __shared__ double red[2][4][RTUS_CURVE_TPB / 64];
    // the fused sweep's matcher scratch ("no ray yet"), when it is small enough to ride along here
    for (int i = threadIdx.x; i < m_init_n; i += RTUS_GEOM1_TPB) m_init[i] = RTUS_NO_RAY;
    const int part = threadIdx.x / RTUS_CURVE_TPB;
    curve_part(k, alpha, n, curve, phi_s, tan_u, node0, node1, node2, part, threadIdx.x % RTUS_CURVE_TPB, red[part]);
    // the boxes are read back from global memory by other waves of this workgroup (no __restrict__ on them here): the barrier's
    // workgroup-scope ordering is all that needs — a device-scope fence here (rounds 1-3) also wrote the whole L2 back: 3 us of
    // the reference sweep's 28.  (Keeping the boxes in LDS for the records as well was measured: no further gain.)
    __syncthreads();
    tree_record(node0, node1, node2, n0, n1, n2, 0, tree, threadIdx.x);
}

// ---- the forward trace: one ray of the alpha grid per lane --------------------------------------
// 6 waves per SIMD (<= 80 VGPRs; the unconstrained allocation of 81 fell one register short): measured +6 %.
#ifndef RTUS_SHOOT_MIN_WAVES
#define RTUS_SHOOT_MIN_WAVES 6
#endif
// EMIT (the solve's grid trace): besides its landing point every wave also says which receive elements each of its 63
// pairs of consecutive rays brackets.  The reference-side question "which ray pairs bracket element e" is asked per
// element; scanning a row's landing points per element costs ~13,500 VALU instructions per row — ten traced rays' worth —
// and ~100 dependent scalar loads of latency.  Asked per PAIR it is three compares against the few elements the wave's
// landing interval covers, right where the landing points are still in registers: a ballot per covered element is the
// 64-bit mask of the pairs that bracket it, and the element's lane of the solve kernel later reads one mask per 64-ray block.
// Counters, 16,384 rows x 905 rays: 1,340 VALU + 553 scalar instructions per wave against 1,200 + 440 without the emission.
// MATCH (the fused sweep, main_rt.py:464-501): the element matcher on the landing points while they are in registers — per
// covered element a ballot of "within atol + rtol*|x_e|" whose lowest set bit is the wave's first matching ray, one integer
// atomicMin per element chunk.  No matcher launch (staging the aperture per workgroup, a second read of the landing points).
template <bool FAST, bool EMIT, bool MATCH>
__global__ __launch_bounds__(RTUS_BLOCK) __attribute__((amdgpu_waves_per_eu(RTUS_SHOOT_MIN_WAVES, 8))) void rtus_shoot_kernel(ShootArgs a)
{
    const int n = a.n;
    const int r_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = r_raw < n;
    const int r = live ? r_raw : n - 1;            // tail lanes redo the last ray (no stores): keeps waves full
    const int tx = blockIdx.y, g = blockIdx.z;
    const LensK& k = a.k;

    RayIn in;
    in.r_outer = a.geoms[2 * g]; in.off = a.geoms[2 * g + 1];          // main_rt.py:466-467
    in.xa = a.x_a[tx]; in.za = a.z_a[tx];
    in.P = a.curve[r];
    in.tu = a.tan_u[r];
    if (!FAST) in.phis = a.phi_s[r];
    in.zf = a.z_f ? a.z_f[r] : a.zf_const;
    RayOut o_;
    trace_ray<FAST>(a, in, o_);
    const double2 P = in.P;
    const double xa = in.xa, za = in.za, zf = in.zf;
    const double xq = o_.xq, zq = o_.zq, xi = o_.xi, zi = o_.zi, x_in = o_.x_in;

    const size_t row = (size_t)g * a.n_tx + tx;
    if (EMIT) {
        const int lane = threadIdx.x & 63;
        // pair (r, r + 1): this lane's landing point and the next lane's; lane 63's partner belongs to the next wave (the
        // solve kernel looks at those block-boundary pairs itself), the last ray has none
        const double l0 = live ? x_in : NAN;
        double l1 = __shfl_down(l0, 1);
        l1 = lane == 63 ? NAN : l1;
        // an element x is bracketed by the pair when both landing points are finite and
        // (l0 < x <= l1 or l0 > x >= l1)  <=>  min <= x <= max and x != l0   (NaN fails every compare)
        const double plo = (l0 == l0 && l1 == l1) ? fmin(l0, l1) : NAN, phi = (l0 == l0 && l1 == l1) ? fmax(l0, l1) : NAN;
        const bool pv = isfinite(plo) && isfinite(phi);
        double wlo = pv ? plo : INFINITY, whi = pv ? phi : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { wlo = fmin(wlo, __shfl_xor(wlo, o)); whi = fmax(whi, __shfl_xor(whi, o)); }
        const int B = r_raw >> 6, nb = (n + 63) >> 6;
        if (B < nb) {                                                   // (waves past the end of the row trace clones of the last ray)
            unsigned long long* __restrict__ mrow = a.pair_mask + (row * nb + B) * (size_t)a.rx_pad;
            // 64 elements at a time, one per lane (a coalesced load per wave: no LDS copy, no workgroup barrier in front of the
            // trace).  A chunk in ascending order: its elements inside the wave's landing interval are [#{x < lo}, #{x <= hi}) —
            // two ballot counts; any other order: all of them.  An element's mask is a ballot over the pairs (its x by
            // v_readlane: no memory access inside the loop) and lands in the element's lane.
            for (int c0 = 0; c0 < a.rx_pad; c0 += 64) {
                const bool ev = c0 + lane < a.n_rx;
                const double xv = a.x_rx[min(c0 + lane, a.n_rx - 1)];
                const double xn = __shfl_down(xv, 1);
                const bool asc = !__ballot(ev && lane < 63 && c0 + lane + 1 < a.n_rx && !(xv <= xn));
                const int nlo = __popcll(__ballot(ev && xv < wlo)), nhi = __popcll(__ballot(ev && xv <= whi));
                const int e0 = asc ? nlo : 0, e1 = asc ? nhi : min(64, a.n_rx - c0);
                unsigned long long mine = 0;
                for (int e = e0; e < e1; ++e) {
                    const double x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xv), e), __builtin_amdgcn_readlane(__double2loint(xv), e));
                    const lanemask m = __ballot(plo <= x && x <= phi && x != l0);
                    mine = lane == e ? m : mine;
                }
                mrow[c0 + lane] = mine;
            }
        }
    }
    if (a.land_box) {
        // the solve's bracket scan skips 64-ray blocks whose landing points all lie outside a wave's element span: a wave
        // IS such a block (rays 64B .. 64B+63 of one row), so the interval costs a wave reduction here instead of a launch
        const bool use = live && isfinite(x_in);
        double lo = use ? x_in : INFINITY, hi = use ? x_in : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
        if ((threadIdx.x & 63) == 0 && live) a.land_box[row * ((n + 63) >> 6) + (r_raw >> 6)] = make_double2(lo, hi);
    }
    if (!MATCH && !live) return;
    if (live) {
    if (a.out8) {
        double* o = a.out8 + row * 8 * (size_t)n + r;
        o[0] = P.x; o[(size_t)n] = P.y; o[2 * (size_t)n] = xq; o[3 * (size_t)n] = zq;
        o[4 * (size_t)n] = xi; o[5 * (size_t)n] = zi; o[6 * (size_t)n] = x_in; o[7 * (size_t)n] = zf;
    }
    if (a.land_x) a.land_x[row * n + r] = x_in;
    if (a.status) a.status[row * n + r] = isnan(xq) ? RTUS_RAY_REF_RAISES : 0;
    if (a.tof4 || (a.tof && !(MATCH && a.tof_lazy))) {
        const double t1 = seg_time<FAST>(xa, za, P.x, P.y, k.c1, k.inv_c1);   // main_compare.py:514
        const double t2 = seg_time<FAST>(P.x, P.y, xq, zq, k.c2, k.inv_c2);   // :515
        const double t3 = seg_time<FAST>(xq, zq, xi, zi, k.c2, k.inv_c2);     // :516
        const double t4 = seg_time<FAST>(xi, zi, x_in, zf, k.c1, k.inv_c1);   // :517
        if (a.tof4) {
            double* t = a.tof4 + row * 4 * (size_t)n + r;
            t[0] = t1; t[(size_t)n] = t2; t[2 * (size_t)n] = t3; t[3 * (size_t)n] = t4;
        }
        if (a.tof) a.tof[row * n + r] = ((t1 + t2) + t3) + t4;         // main_rt.py:497-500
    }
    }
    if (MATCH) {
        const int lane = threadIdx.x & 63;
        const int B = r_raw >> 6, nb = (n + 63) >> 6;
        if (B >= nb) return;                                            // (wave-uniform: clones of the last ray past the end of the row)
        // np.isclose(x_land, x_e): |x_land - x_e| <= atol + rtol*|x_e| for finite operands, x_land == x_e when either is infinite,
        // never with a NaN
        const bool use = live && x_in == x_in;
        double wlo = use ? x_in : INFINITY, whi = use ? x_in : -INFINITY;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { wlo = fmin(wlo, __shfl_xor(wlo, o)); whi = fmax(whi, __shfl_xor(whi, o)); }
        int32_t* __restrict__ mrow = a.m_first + row * (size_t)a.rx_pad;
        lanemask any_cand = 0;                                          // (wave-uniform) some element's first matching ray so far is in this wave
        for (int c0 = 0; c0 < a.rx_pad; c0 += 64) {
            const bool ev = c0 + lane < a.n_rx;
            const double xv = a.x_rx[min(c0 + lane, a.n_rx - 1)];
            const double tolv = isfinite(xv) ? a.atol + a.rtol * fabs(xv) : -1.0;   // (unfused in this file: two roundings, as NumPy; negative: an
                                                                                    // infinite element stays out of the first test)
            const double xn = __shfl_down(xv, 1);
            const bool asc = !__ballot(ev && lane < 63 && c0 + lane + 1 < a.n_rx && !(xv <= xn));
            // ascending chunk: x + tol and x - tol ascend with x (rtol < 1), so the elements a landing point of this wave can
            // match are [#{x + tol < lo}, #{x - tol <= hi}) — with the tolerance widened by a few ulp of the operands, so that the
            // rounding of x + tol here can never drop an element whose rounded |x_land - x| is exactly tol; the test
            // itself, below, is np.isclose's
            const double tolw = tolv + 8.0 * 2.220446049250313e-16 * (fabs(xv) + fabs(tolv));
            const int nlo = __popcll(__ballot(ev && xv + tolw < wlo)), nhi = __popcll(__ballot(ev && xv - tolw <= whi));
            const bool mono = asc && a.rtol < 1.0;
            const int e0 = mono ? nlo : 0, e1 = mono ? nhi : min(64, a.n_rx - c0);
            int cand = RTUS_NO_RAY;
            for (int e = e0; e < e1; ++e) {
                const double xe = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xv), e), __builtin_amdgcn_readlane(__double2loint(xv), e));
                const double te = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(tolv), e), __builtin_amdgcn_readlane(__double2loint(tolv), e));
                const lanemask m = __ballot(use && (fabs(x_in - xe) <= te || (isinf(xe) && x_in == xe)));
                const int first = (B << 6) + (int)__builtin_ctzll(m | (1ull << 63));
                cand = (lane == e && m) ? first : cand;
            }
            if (cand != RTUS_NO_RAY) atomicMin(&mrow[c0 + lane], cand);
            any_cand |= __ballot(cand != RTUS_NO_RAY);
        }
        // The caller wants the hits only (no per-ray travel times): the finalize kernel reads the time of an element's WINNING ray,
        // and a winner lies in a wave that matched something — every other wave (nearly all: a ray within 1e-6 m of an element is
        // rare) skips the four segment times, ~135 of its ~950 VALU instructions.  The same arithmetic in the same wave: the same bits.
        if (a.tof_lazy && any_cand && live) {
            const double t1 = seg_time<FAST>(xa, za, P.x, P.y, k.c1, k.inv_c1);
            const double t2 = seg_time<FAST>(P.x, P.y, xq, zq, k.c2, k.inv_c2);
            const double t3 = seg_time<FAST>(xq, zq, xi, zi, k.c2, k.inv_c2);
            const double t4 = seg_time<FAST>(xi, zi, x_in, zf, k.c1, k.inv_c1);
            a.tof[row * n + r] = ((t1 + t2) + t3) + t4;                 // main_rt.py:497-500
        }
        // (rtus_sweep_finalize_kernel turns the winners into first_ray / hit / tof_hit.  Finalizing inside this kernel — the last
        // wave of a row to arrive, counted by an atomic — was built and measured: no faster for the reference's sweep (29.7 us
        // against 28.7) and, with a release fence per wave, 85 us: buffer_wbl2 writes the whole L2 back)
    }
}

// ---- host-side launchers (called from rtus_capi.hip) -----------------------------------------
size_t rtus_ws_bytes(int n) { return shoot_ws_bytes(n); }

// polyline + boxes + depth-first records of the alpha grid into the workspace `a` points to
static void rtus_launch_geometry(const ShootArgs& a, const double* alpha, hipStream_t s, int32_t* m_init = nullptr, int m_init_n = 0)
{
    const int n = a.n;
    if (n <= RTUS_GEOM1_TPB) {
        hipLaunchKernelGGL(rtus_geom1_kernel, dim3(1), dim3(RTUS_GEOM1_TPB), 0, s, a.k, alpha, n, (double2*)a.curve, (double*)a.phi_s,
                           (double2*)a.tan_u, (double4*)a.node0, (double4*)a.node1, (double4*)a.node2, a.n0, a.n1, a.n2, (TreeNode*)a.tree,
                           m_init, m_init_n);
        return;
    }
    hipLaunchKernelGGL(rtus_curve_kernel, dim3(a.n2), dim3(RTUS_CURVE_TPB), 0, s, a.k, alpha, n,
                       (double2*)a.curve, (double*)a.phi_s, (double2*)a.tan_u, (double4*)a.node0, (double4*)a.node1,
                       (double4*)a.node2, m_init, m_init_n);
    hipLaunchKernelGGL(rtus_tree_kernel, dim3((a.n_tree + 255) / 256), dim3(256), 0, s, a.node0, a.node1, a.node2, a.n0, a.n1, a.n2,
                       a.n3, (TreeNode*)a.tree);
}

// z_f == nullptr: every ray lands on z = zf_const; land_box: optional per-wave (min, max) of the landing points (the solve)
// the polyline, tangents, boxes and box records of `alpha` into the workspace `a` points to (rtus_solve's one-launch path)
void rtus_launch_geometry_only(const ShootArgs& a, const double* alpha, hipStream_t s) { rtus_launch_geometry(a, alpha, s); }

hipError_t rtus_launch_shoot_ex(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                                       const double* z_a, int n_tx, const double* alpha, const double* z_f, double zf_const, int n,
                                       double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                                       double2* land_box, unsigned long long* pair_mask, const double* x_rx, int n_rx,
                                       void* ws, unsigned flags, hipStream_t s)
{
    char* w = (char*)ws;
    ShootArgs a;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = z_f; a.zf_const = zf_const; a.land_box = land_box;
    a.pair_mask = pair_mask; a.x_rx = x_rx; a.n_rx = n_rx; a.rx_pad = (n_rx + 63) & ~63;
    shoot_args_workspace(a, w, n);
    if ((unsigned long long)a.n_tree * sizeof(TreeNode) >= 0xffffffffull) return hipErrorInvalidValue;   // the walk's record offsets are 32-bit
    a.out8 = out8; a.tof4 = tof4; a.tof = tof; a.land_x = land_x; a.status = status;
    a.n_tx = n_tx; a.n_geom = n_geom;
    a.tof_lazy = 0;
    a.flags = flags;
    if (!(flags & RTUS_POLYLINE_READY)) rtus_launch_geometry(a, alpha, s);
    const dim3 grid((n + RTUS_BLOCK - 1) / RTUS_BLOCK, n_tx, n_geom);
    const bool fast = (flags & RTUS_SHOOT_FAST_MATH) != 0;
    if (pair_mask) {
        if (n_rx <= 0 || n_rx > RTUS_SOLVE_MASK_MAX_RX || !x_rx) return hipErrorInvalidValue;
        if (fast) hipLaunchKernelGGL((rtus_shoot_kernel<true, true, false>), grid, dim3(RTUS_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((rtus_shoot_kernel<false, true, false>), grid, dim3(RTUS_BLOCK), 0, s, a);
    } else {
        if (fast) hipLaunchKernelGGL((rtus_shoot_kernel<true, false, false>), grid, dim3(RTUS_BLOCK), 0, s, a);
        else hipLaunchKernelGGL((rtus_shoot_kernel<false, false, false>), grid, dim3(RTUS_BLOCK), 0, s, a);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void rtus_sweep_fill_kernel(int4* __restrict__ p, long long n4)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_int4(RTUS_NO_RAY, RTUS_NO_RAY, RTUS_NO_RAY, RTUS_NO_RAY);
}

// the matcher's winners -> first_ray (-1 when none) / hit / tof of the first hitting ray (0.0 when none: main_rt.py:493); the scratch is idle after
__global__ __launch_bounds__(256) void rtus_sweep_finalize_kernel(int32_t* __restrict__ m_first, int rx_pad, const double* __restrict__ tof, int n,
                                                                   int n_rx, long long rows, int32_t* __restrict__ first_ray,
                                                                   uint8_t* __restrict__ hit, double* __restrict__ tof_hit)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * rx_pad) return;
    const long long row = i / rx_pad;
    const int e = (int)(i - row * rx_pad);
    const int f = m_first[i];
    if (f == RTUS_NO_RAY && e >= n_rx) return;
    m_first[i] = RTUS_NO_RAY;
    if (e >= n_rx) return;
    const bool h = f != RTUS_NO_RAY;
    const size_t o = (size_t)row * n_rx + e;
    first_ray[o] = h ? f : -1;
    if (hit) hit[o] = h ? 1 : 0;
    if (tof_hit) tof_hit[o] = h ? tof[(size_t)row * n + f] : 0.0;
}

// ---- the fused sweep: forward trace + element matcher (main_rt.py:464-501) in one kernel ---------------------------------
// workspace: [forward trace's geometry][m_first: rows x rx_pad int32][tof: rows x n doubles, used when the caller does not ask
// for the per-ray times]
static inline size_t sweep_al(size_t b) { return (b + 255) & ~(size_t)255; }
size_t rtus_sweep_ws_bytes(int n, int n_geom, int n_tx, int n_rx)
{
    const size_t rows = (size_t)n_geom * n_tx, rx_pad = ((size_t)n_rx + 63) & ~(size_t)63;
    return sweep_al(shoot_ws_bytes(n)) + sweep_al(4 * rows * rx_pad) + sweep_al(8 * rows * (size_t)n);
}

hipError_t rtus_launch_sweep(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
                             const double* alpha, const double* z_f, int n, const double* x_rx, int n_rx, double atol, double rtol,
                             int32_t* first_ray, uint8_t* hit, double* tof_hit, double* tof, double* land_x, void* ws, unsigned flags,
                             hipStream_t s)
{
    char* w = (char*)ws;
    const size_t rows = (size_t)n_geom * n_tx, rx_pad = ((size_t)n_rx + 63) & ~(size_t)63;
    if (rows * rx_pad > 0x7fffffffull) return hipErrorInvalidValue;
    ShootArgs a;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = z_f; a.zf_const = 0.0; a.land_box = nullptr; a.pair_mask = nullptr;
    a.x_rx = x_rx; a.n_rx = n_rx; a.rx_pad = (int)rx_pad;
    shoot_args_workspace(a, w, n);
    if ((unsigned long long)a.n_tree * sizeof(TreeNode) >= 0xffffffffull) return hipErrorInvalidValue;
    int32_t* scratch = (int32_t*)(w + sweep_al(shoot_ws_bytes(n)));
    a.m_first = scratch;
    a.atol = atol; a.rtol = rtol;
    a.out8 = nullptr; a.tof4 = nullptr; a.land_x = land_x; a.status = nullptr;
    a.tof = tof ? tof : (double*)(w + sweep_al(shoot_ws_bytes(n)) + sweep_al(4 * rows * rx_pad));
    a.tof_lazy = tof ? 0 : 1;                                           // hits only: times of the waves that matched something
    a.n_tx = n_tx; a.n_geom = n_geom; a.flags = flags;
    if (!(flags & RTUS_POLYLINE_READY)) {
        // the matcher's scratch starts idle: inside the polyline kernel when that costs it at most 64 stores per thread, by a fill
        // kernel otherwise (RTUS_POLYLINE_READY: the finalize kernel of the previous launch on this workspace left it idle)
        const size_t geom_threads = n <= RTUS_GEOM1_TPB ? RTUS_GEOM1_TPB : (size_t)a.n2 * RTUS_CURVE_TPB;
        const bool ride = rows * rx_pad <= 64 * geom_threads;
        if (!ride) hipLaunchKernelGGL(rtus_sweep_fill_kernel, dim3((unsigned)((rows * rx_pad + 1023) / 1024)), dim3(256), 0, s, (int4*)scratch, (long long)(rows * rx_pad / 4));
        rtus_launch_geometry(a, alpha, s, ride ? scratch : nullptr, ride ? (int)(rows * rx_pad) : 0);
    }
    const dim3 grid((n + RTUS_BLOCK - 1) / RTUS_BLOCK, n_tx, n_geom);
    if (flags & RTUS_SHOOT_FAST_MATH) hipLaunchKernelGGL((rtus_shoot_kernel<true, false, true>), grid, dim3(RTUS_BLOCK), 0, s, a);
    else hipLaunchKernelGGL((rtus_shoot_kernel<false, false, true>), grid, dim3(RTUS_BLOCK), 0, s, a);
    hipLaunchKernelGGL(rtus_sweep_finalize_kernel, dim3((unsigned)((rows * rx_pad + 255) / 256)), dim3(256), 0, s, a.m_first, (int)rx_pad,
                           a.tof, n, n_rx, (long long)rows, first_ray, hit, tof_hit);
    return hipGetLastError();
}

hipError_t rtus_launch_shoot(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, const double* z_f, int n,
                             double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                             void* ws, unsigned flags, hipStream_t s)
{
    return rtus_launch_shoot_ex(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, z_f, 0.0, n, out8, tof4, tof, land_x, status,
                                nullptr, nullptr, nullptr, 0, ws, flags, s);
}

// ---- self-test (rtus_selftest): the claims the forward trace leans on, checked on the device it runs on -------------
// (1) rtus_div / rtus_div_by / rtus_sqrt return the bits of the correctly rounded a / b and sqrt(a) — over pseudo-random operands with
//     exponents in +-500 and every special value (+-0, +-inf, NaN, +-1, the smallest normal; quotients that come out
//     denormal are outside rtus_div's contract and not counted); (2) the depth-first records rtus_tree_kernel
//     builds are a tree: skip links move forward and land on the record that starts where the box ends, leaves cover the
//     polyline in order.
__device__ __forceinline__ unsigned long long selftest_mix(unsigned long long z)
{
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
__device__ __forceinline__ double selftest_operand(unsigned long long r)
{
    const double specials[8] = {0.0, -0.0, INFINITY, -INFINITY, NAN, 1.0, -1.0, 2.2250738585072014e-308};
    if ((r & 63) == 0) return specials[(r >> 6) & 7];
    const unsigned long long mant = r >> 12, sign = (r >> 11) & 1;
    const int e = (int)((r >> 1) % 1001) - 500;                      // 2^-500 .. 2^500
    return rtus_from_bits((sign << 63) | ((unsigned long long)(e + 1023) << 52) | mant);
}
struct SelftestDivisors { double c[8], rc[8]; };
__global__ void rtus_selftest_math_kernel(unsigned long long seed, long long n, SelftestDivisors d, unsigned long long* __restrict__ bad)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = selftest_operand(selftest_mix(seed + 2 * (unsigned long long)i));
    const double b = selftest_operand(selftest_mix(seed + 2 * (unsigned long long)i + 1));
    const double q1 = rtus_div(a, b), q2 = a / b;
    const double s1 = rtus_sqrt(a), s2 = sqrt(a);
    const bool q_denormal = q2 != 0.0 && fabs(q2) < 2.2250738585072014e-308;     // outside rtus_div's contract (no pre-scaling)
    if (rtus_bits(q1) != rtus_bits(q2) && !(q1 != q1 && q2 != q2) && !q_denormal) atomicAdd(bad, 1ull);
    if (rtus_bits(s1) != rtus_bits(s2) && !(s1 != s1 && s2 != s2)) atomicAdd(bad + 1, 1ull);
    // division by a wave-uniform divisor through its host-rounded reciprocal (sound speeds, the lens constant 2A, ...)
    const int j = blockIdx.x & 7;
    const double u1 = rtus_div_by(a, d.c[j], d.rc[j]), u2 = a / d.c[j];
    const bool u_denormal = u2 != 0.0 && fabs(u2) < 2.2250738585072014e-308;
    if (rtus_bits(u1) != rtus_bits(u2) && !(u1 != u1 && u2 != u2) && !u_denormal) atomicAdd(bad, 1ull);
}

hipError_t rtus_selftest_run(const rtus_lens& lens, int n, long long n_math, unsigned long long counts[4], hipStream_t s)
{
    counts[0] = counts[1] = counts[2] = 0; counts[3] = (unsigned long long)n_math;
    unsigned long long* d_bad = nullptr;
    hipError_t e = hipMalloc(&d_bad, 2 * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
    (void)hipMemsetAsync(d_bad, 0, 2 * sizeof(unsigned long long), s);
    SelftestDivisors dv;
    const LensK kk = make_lens_k(lens);
    const double divisors[8] = {lens.c1, lens.c2, kk.twoA, 1483.0, 5900.0, 2330.0, 0.7853981633974483, 6399.999999999999};
    for (int i = 0; i < 8; ++i) { dv.c[i] = divisors[i]; dv.rc[i] = 1.0 / divisors[i]; }
    hipLaunchKernelGGL(rtus_selftest_math_kernel, dim3((unsigned)((n_math + 255) / 256)), dim3(256), 0, s, 0x1234567ull, n_math, dv, d_bad);
    // the tree of an n-point polyline on the reference's launch-angle interval
    ShootArgs a;
    a.k = make_lens_k(lens);
    char* w = nullptr;
    double* d_alpha = nullptr;
    if ((e = hipMalloc(&w, rtus_ws_bytes(n))) != hipSuccess || (e = hipMalloc(&d_alpha, sizeof(double) * n)) != hipSuccess) {
        (void)hipFree(d_bad); (void)hipFree(w); return e;
    }
    std::vector<double> alpha(n);
    const double amax = 50.62033040986099 * (3.14159265358979323846 / 180.0);
    for (int i = 0; i < n; ++i) alpha[i] = n > 1 ? -amax + 2.0 * amax * i / (n - 1) : 0.0;
    (void)hipMemcpyAsync(d_alpha, alpha.data(), sizeof(double) * n, hipMemcpyHostToDevice, s);
    shoot_args_workspace(a, w, n);
    rtus_launch_geometry(a, d_alpha, s);
    std::vector<TreeNode> t(a.n_tree + 1);
    unsigned long long bad[2] = {0, 0};
    (void)hipMemcpyAsync(t.data(), a.tree, sizeof(TreeNode) * t.size(), hipMemcpyDeviceToHost, s);
    (void)hipMemcpyAsync(bad, d_bad, sizeof(bad), hipMemcpyDeviceToHost, s);
    e = hipStreamSynchronize(s);
    (void)hipFree(d_bad); (void)hipFree(w); (void)hipFree(d_alpha);
    if (e != hipSuccess) return e;
    counts[0] = bad[0]; counts[1] = bad[1];
    unsigned long long viol = 0;
    const unsigned end = (unsigned)a.n_tree * (unsigned)sizeof(TreeNode);
    int next_leaf = 0;
    for (int i = 0; i < a.n_tree; ++i) {
        const TreeNode& r = t[i];
        const unsigned off = (unsigned)i * (unsigned)sizeof(TreeNode);
        const bool leaf = r.leafm != 0;
        if (!(r.skip_off > off && r.skip_off <= end && r.skip_off % sizeof(TreeNode) == 0)) { ++viol; continue; }
        if (leaf != (r.j1 - r.j0 == 8) || (r.leafm != 0 && r.leafm != ~0ull) || !(r.xh >= 0.0 && r.zh >= 0.0)) ++viol;
        if (leaf) { if (r.j0 != next_leaf || r.skip_off != off + sizeof(TreeNode)) ++viol; next_leaf += 8; }
        else if (t[i + 1].j0 != r.j0) ++viol;                       // an inner box is followed by its first child
        const unsigned nx = r.skip_off / (unsigned)sizeof(TreeNode);
        if (nx < (unsigned)a.n_tree && t[nx].j0 != r.j1) ++viol;   // past the subtree: the box that starts where this one ends
        if (nx == (unsigned)a.n_tree && r.j1 < n) ++viol;          // ... or the end of the polyline
    }
    if (next_leaf != ((n + 7) & ~7)) ++viol;
    counts[2] = viol;
    return hipSuccess;
}

#ifdef RTUS_EXP_COUNT   // experiment builds only (scripts/exp_count.py)
extern "C" int rtus_dbg_read(unsigned long long* out, int reset)
{
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(rtus_dbg), sizeof(unsigned long long) * 8);
    if (reset) { unsigned long long z[8] = {0}; hipMemcpyToSymbol(HIP_SYMBOL(rtus_dbg), z, sizeof(z)); }
    return 0;
}
#endif
