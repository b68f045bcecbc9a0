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
#include "rtus_device.h"
#include <vector>

#define RTUS_CURVE_TPB 512      // curve kernel: one workgroup = one 512-point node2

struct ShootArgs {
    LensK k;
    const double* __restrict__ geoms;   // [n_geom][2]
    const double* __restrict__ x_a;     // [n_tx]
    const double* __restrict__ z_a;     // [n_tx]
    const double* __restrict__ z_f;     // [n]
    const double2* __restrict__ curve;  // [n]
    const double* __restrict__ phi_s;   // [n]
    const double2* __restrict__ tan_u;  // [n] unit tangent (cos phi_s, sin phi_s)
    const double4* __restrict__ node0;  // [n0]
    const double4* __restrict__ node1;  // [n1]
    const double4* __restrict__ node2;  // [n2]
    const struct TreeNode* __restrict__ tree;   // [n0 + n1 + n2 + n3] the boxes in depth-first order (see TreeNode)
    double* __restrict__ out8;          // nullable
    double* __restrict__ tof4;          // nullable
    double* __restrict__ tof;           // nullable
    double* __restrict__ land_x;        // nullable
    uint8_t* __restrict__ status;       // nullable
    int n, n_tx, n_geom, n0, n1, n2, n3, n_tree;    // n3: 4096-point boxes, only when n2 > 8 (else 0); they live in the tree only
    unsigned flags;
};

__device__ __forceinline__ double4 make_box(double xmin, double xmax, double zmin, double zmax)
{
    // centre / half-extent, half-extents nudged up so the box still contains its points after rounding
    const double xc = 0.5 * (xmin + xmax), zc = 0.5 * (zmin + zmax);
    const double xh = fmax(xmax - xc, xc - xmin) * (1.0 + 1e-15), zh = fmax(zmax - zc, zc - zmin) * (1.0 + 1e-15);
    return make_double4(xc, xh, zc, zh);
}

// ---- polyline + bounding-box hierarchy -------------------------------------------------------
__global__ __launch_bounds__(RTUS_CURVE_TPB) void rtus_curve_kernel(LensK k, const double* __restrict__ alpha,
                                                                     int n, double2* __restrict__ curve,
                                                                     double* __restrict__ phi_s,
                                                                     double2* __restrict__ tan_u,
                                                                     double4* __restrict__ node0,
                                                                     double4* __restrict__ node1,
                                                                     double4* __restrict__ node2)
{
    __shared__ double red[4][RTUS_CURVE_TPB / 64];
    const int j = blockIdx.x * RTUS_CURVE_TPB + threadIdx.x;
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
        if (o == 4 && (threadIdx.x & 7) == 0 && live) node0[j >> 3] = make_box(xmin, xmax, zmin, zmax);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        if (live) node1[j >> 6] = make_box(xmin, xmax, zmin, zmax);
        red[0][wv] = xmin; red[1][wv] = xmax; red[2][wv] = zmin; red[3][wv] = zmax;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RTUS_CURVE_TPB / 64; ++w) {
            xmin = fmin(xmin, red[0][w]); xmax = fmax(xmax, red[1][w]);
            zmin = fmin(zmin, red[2][w]); zmax = fmax(zmax, red[3][w]);
        }
        node2[blockIdx.x] = make_box(xmin, xmax, zmin, zmax);
    }
}

// The box hierarchy in depth-first (pre-order) order, one 64-byte record per box: what the crossing search walks.  A visit
// is ONE scalar load at a wave-uniform index i; the next index is i + 1 (first child / next leaf) while some ray is
// still undecided about this box and `skip` (the record after this box's whole subtree) once every ray is decided — a
// single loop without per-level counters, bounds or address arithmetic: on gfx950 the scalar ALU issues one instruction per
// ~4.4 cycles per SIMD (scripts/ubench_issue4.hip) and the nested-loop form of this walk spent ~25 scalar instructions per
// box, three times the cost of the box test itself.
// Only the LAST subtree of a level can be incomplete (boxes group consecutive points), so with the full-subtree sizes
// 1, 9, 73, 585 the record of box k of level L sits at  t sz[top] + sum over the levels l below the top of
// (1 + digit_l(k) sz[l])  — every box computes its own place, no scan.
struct __attribute__((aligned(64))) TreeNode {
    double xc, xh, zc, zh;        // the box: centre / half-extent
    int j0;                       // its first polyline point
    unsigned skip_off;            // byte offset of the record after this box's subtree
    unsigned long long leafm;     // all ones: an 8-point unit; 0: an inner box (a lane mask, so the walk needs no branch on it)
    int j1;                       // one past its last polyline point
    int pad[3];
};
// The record after the last box carries the polyline's extent instead of a box: xc = max |x|, xh = max |z| (the rounding
// part of the certification margin).

__global__ void rtus_tree_kernel(const double4* __restrict__ node0, const double4* __restrict__ node1,
                                 const double4* __restrict__ node2, int n0, int n1, int n2, int n3,
                                 TreeNode* __restrict__ tree)
{
    const int total = n0 + n1 + n2 + n3;
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
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

// Crossing search state.  The reference wants the first j with sign(d_j) != sign(d_{j+1})
// (main_rt.py:82, 99).  Every point before that change has the class c0 = np.sign(d_0) of polyline
// point 0, so equivalently: p = the first polyline point whose class differs from c0, idx = p - 1.
// Everything that is one bit per ray lives in 64-bit lane masks (SGPR pairs): v_cmp writes them
// directly and the bookkeeping is scalar-ALU work.
typedef unsigned long long lanemask;
#ifdef RTUS_EXP_COUNT
__device__ unsigned long long rtus_dbg[8];
#define DBG(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&rtus_dbg[i], 1ull); } while (0)
#else
#define DBG(i) do {} while (0)
#endif

struct Walk {
    // per lane: the ray's line z = m x + b multiplied by sg = +1 / -1 so that "same class as point 0" always
    // reads d > 0 (multiplying by +-1 is exact: every certification decision is the one the unsigned form makes)
    double ms, bs, sg, am;
    double marg;             // certification margin; +inf for rays with d_0 == 0 exactly (class 0 is never certified)
    int slot;                // per lane: first point of the box where the lane left the walk (its answer lies in slot .. slot + 7)
    int start;               // per lane: polyline points before this index are known to have class c0
    lanemask active;         // rays still walking
};

// this lane's bit of a wave-uniform mask: the mask itself becomes the v_cndmask / exec operand (no per-lane shift + compare)
__device__ __forceinline__ bool lane_bit(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }

// Which rays have the whole box on one side of their line?  pos: every point keeps the class of point 0,
// neg: every point has the opposite class.  Each VALU op reads ONE scalar operand (the box lives in SGPRs and
// gfx9 VOP3 has a single constant-bus slot — otherwise the compiler adds v_mov's per visit).
__device__ __forceinline__ void certify(const Walk& W, const double4 bx, double marg, lanemask& pos, lanemask& neg)
{
    const double e = fma(W.sg, bx.z, -fma(W.ms, bx.x, W.bs));   // sg * d at the box centre
    const double s = fma(W.am, bx.y, bx.w + marg);               // how far d can move inside the box + margin
    pos = __ballot(e > s);
    neg = __ballot(e < -s);
}
// retry pass: points before `start` are known to be class c0
__device__ __forceinline__ void known_prefix(const Walk& W, int j0, int j1, lanemask& pos, lanemask& neg)
{
    const lanemask before = __ballot(j1 <= W.start), partly = __ballot(j0 < W.start) & ~before;
    pos = (pos | before) & ~partly;
    neg &= ~(before | partly);
}

// One lock-step pass over the box hierarchy, boxes in index order (depth-first records, see TreeNode).  At an inner box
// the rays certified "opposite" are found (p = its first point) and leave the walk; the wave descends only if some ray
// could not be decided.  At a leaf (8 points) every ray that is not certified "same" leaves the walk — found, or parked
// on the leaf to look at its points itself.  RETRY = a pass after the first (rays whose parked leaf held no change resume
// behind it): compiled separately so that the first pass, which is nearly always the only one, carries none of the
// prefix bookkeeping.
// One record into SGPRs: ONE scalar-memory instruction at base + byte offset (the compiler's own form of tree[i] is a 64-bit
// shift / add / add-with-carry in front of two or three narrower loads).  Not volatile (a volatile asm is a memory clobber
// and would turn every later wave-uniform load of the kernel into a vector load); the tree is read-only here.
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ TreeNode load_record(const TreeNode* __restrict__ base, unsigned off)
{
    v16i r;
    asm("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(r) : "s"(base), "s"(off));
    return __builtin_bit_cast(TreeNode, r);
}

// The first pass of the walk (nearly always the only one), written out: the loop below is walk_pass<false> instruction for
// instruction, with the scalar bookkeeping the way the hardware offers it — the mask instructions set SCC themselves, so
// "does the wave descend" and "is any ray left" cost no compare, and the two exits are two branches.  12 scalar
// instructions per box; the compiler's version of the same C++ has 20 (compares re-materialised as 64-bit selects,
// register copies for the loop-carried masks), and the scalar ALU issues one instruction per ~4.4 cycles per SIMD, so
// those eight are ~10 % of the kernel.  Fixed scalar registers (s75-s79, s84-s99) because an asm operand cannot name the
// halves of a 16-register load.  No hardware hazard in here needs a manual wait state (gfx9 list: VALU-written SGPR / VCC
// read by the scalar ALU and VCC written by the scalar ALU read by v_cndmask are interlocked); the order of the dependent
// pairs is the one the compiler emits for the C++ form.  Record layout: see TreeNode.
__device__ __forceinline__ void walk_first_pass(Walk& W, const TreeNode* __restrict__ tree, int n_tree)
{
    const unsigned end = (unsigned)n_tree * (unsigned)sizeof(TreeNode);
    lanemask active = W.active;
    unsigned off = 0;
    int slot = W.slot, tmp;
    double e, sm;
    asm("s_cmp_eq_u64 %[act], 0\n\t"
        "s_cbranch_scc1 2f\n"
        "1:\n\t"
        "s_load_dwordx16 s[84:99], %[base], %[off]\n\t"
        "s_add_u32 s75, %[off], 64\n\t"
        "s_waitcnt lgkmcnt(0)\n\t"
        "v_fma_f64 %[e], s[84:85], %[ms], %[bs]\n\t"           // ms xc + bs
        "v_add_f64 %[sm], %[marg], s[90:91]\n\t"               // zh + margin
        "v_fma_f64 %[e], %[sg], s[88:89], -%[e]\n\t"           // e = sg zc - (ms xc + bs): sg d at the box centre
        "v_fma_f64 %[sm], %[am], s[86:87], %[sm]\n\t"          // s = |m| xh + zh + margin
        "v_cmp_gt_f64_e32 vcc, %[e], %[sm]\n\t"                // pos: the whole box keeps the class of point 0
        "v_cmp_lt_f64_e64 s[78:79], %[e], -%[sm]\n\t"          // neg: the whole box has the opposite class
        "s_andn2_b64 s[76:77], %[act], vcc\n\t"                // np = active & ~pos
        "s_or_b64 s[78:79], s[78:79], s[94:95]\n\t"            // neg | leafm
        "s_and_b64 vcc, s[76:77], s[78:79]\n\t"                // gone = np & (neg | leafm)
        "s_max_u32 s93, s93, s75\n\t"                          // the walk advances whatever the record holds
        "s_andn2_b64 s[76:77], s[76:77], s[78:79]\n\t"         // undecided rays of an inner box; SCC = any
        "s_cselect_b32 %[off], s75, s93\n\t"                   // descend: next record; else: past this subtree
        "v_mov_b32_e32 %[tmp], s92\n\t"
        "s_andn2_b64 %[act], %[act], vcc\n\t"                  // active &= ~gone; SCC = any ray left
        "v_cndmask_b32_e32 %[slot], %[slot], %[tmp], vcc\n\t"  // rays that leave here remember the box's first point
        "s_cbranch_scc0 2f\n\t"
        "s_cmp_lt_u32 %[off], %[end]\n\t"
        "s_cbranch_scc1 1b\n"
        "2:"
        : [act] "+s"(active), [off] "+s"(off), [slot] "+v"(slot), [tmp] "=&v"(tmp), [e] "=&v"(e), [sm] "=&v"(sm)
        : [base] "s"(tree), [end] "s"(end), [ms] "v"(W.ms), [bs] "v"(W.bs), [sg] "v"(W.sg), [am] "v"(W.am), [marg] "v"(W.marg)
        : "vcc", "scc", "s75", "s76", "s77", "s78", "s79", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93",
          "s94", "s95", "s96", "s97", "s98", "s99");
    W.active = active;
    W.slot = slot;
}

template <bool RETRY>
__device__ __forceinline__ void walk_pass(Walk& W, const TreeNode* __restrict__ tree, int n_tree)
{
    const unsigned end = (unsigned)n_tree * (unsigned)sizeof(TreeNode);
    unsigned off = 0;
    while (off < end && W.active) {
        const TreeNode nd = load_record(tree, off);
        DBG(0);
        lanemask pos, neg;
        certify(W, make_double4(nd.xc, nd.xh, nd.zc, nd.zh), W.marg, pos, neg);
        if (RETRY) known_prefix(W, nd.j0, nd.j1, pos, neg);
        // inner box: rays certified "opposite" leave here (their answer is its first point), the wave descends if some
        // ray is undecided; leaf: every ray not certified "same" leaves.  One formula, leafm = all / none.  A ray that
        // leaves looks at the 8 points from j0 on by itself afterwards — for a certified ray the first of them differs,
        // so "found here" and "parked on this leaf" need no separate bookkeeping.
        const lanemask np = W.active & ~pos, und = np & ~neg;
        const lanemask gone = np & (neg | nd.leafm);
        W.active &= ~gone;
        W.slot = lane_bit(gone) ? nd.j0 : W.slot;
        const unsigned next = off + (unsigned)sizeof(TreeNode);
        off = (und & ~nd.leafm) ? next : max(nd.skip_off, next);   // (max: the walk advances whatever the record holds)
    }
}

// ---- cheap reciprocal / square roots for the vector-form (FAST) mode ----------------------------------
// IEEE fp64 divide / sqrt cost 25 / 36 ns per wave-op (profiles/r01_ubench_fp64.txt); the hardware seeds
// v_rcp_f64 / v_rsq_f64 (6.8 ns, 5e-8) + Newton steps reach ~1 ulp in ~17 / ~19 ns, and most sqrt+divide
// pairs of the trace are really one reciprocal square root.  The reference-compatible mode (FAST = false)
// keeps the correctly rounded operations NumPy uses.
__device__ __forceinline__ double rcp_fast(double b)
{
    double y = __builtin_amdgcn_rcp(b);
    y = fma(y, fma(-b, y, 1.0), y);
    y = fma(y, fma(-b, y, 1.0), y);
    return y;                                   // b = 0 / inf / NaN -> NaN (degenerate rays only)
}
__device__ __forceinline__ double rsqrt_fast(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);   // cubic step: 5e-8 -> rounding level; x <= 0 -> NaN
}
template <bool FAST> __device__ __forceinline__ double m_div(double a, double b) { return FAST ? a * rcp_fast(b) : rtus_div(a, b); }
template <bool FAST> __device__ __forceinline__ double m_sqrt(double x)
{
    if (!FAST) return rtus_sqrt(x);
    const double r = x * rsqrt_fast(x);
    return x == 0.0 ? 0.0 : r;                  // keeps sqrt(0) = 0 (tangent hits, grazing refraction)
}

// One segment's travel time, dist / c (main_rt.py:444-445, 497-500).
template <bool FAST> __device__ __forceinline__ double seg_time(double x1, double z1, double x2, double z2, double c,
                                                                 double inv_c)
{
    if (!FAST) return rtus_div_by(dist2d(x1, z1, x2, z2), c, inv_c);
    const double dx = x1 - x2, dz = z1 - z2;
    return m_sqrt<true>(dx * dx + dz * dz) * inv_c;
}

// Fast-math mode keeps lines in slope form like the reference does; an exactly vertical direction would
// make the slope infinite where the reference gets tan(pi/2 rounded) = 1.633e16.  Same cap here.
__device__ __forceinline__ double cap_vertical(double ux, double uz)
{
    const double lim = 6.123233995736766e-17 * fabs(uz);       // 1 / tan(fl(pi/2))
    return fabs(ux) < lim ? copysign(lim, ux) : ux;           // NaN stays NaN
}

// ---- one ray: lens point P (+ its tangent) -> pipe -> lens -> landing ---------------------------
// Called by all 64 lanes of a wave together (the crossing search is a lock-step walk).
struct RayIn { double2 P; double phis; double2 tu; double xa, za, r_outer, off, zf; };
struct RayOut { double xq, zq, xi, zi, x_in; };

// Reference-compatible mode (FAST = false): the reference's angle arithmetic, operation for operation, with sin / tan /
// atan2 / atan / asin through the kernels of rtus_trig.h, correctly rounded division and square root (rtus_div,
// rtus_sqrt) and the lens at alpha_i = atan2(x_i, z_i) through x_i / rho, z_i / rho (no angle formed): 1,200 instead
// of 2,066 executed VALU instructions per wave.  ONE place keeps the library's routines: the first refraction of a wave
// that holds a near-vertical refracted line (|a_pq| > 300: ~1 % of the waves).  The
// reference intersects that line with the pipe through the quadratic formula in slope-intercept form
// (main_rt.py:349-364), which amplifies a last-bit difference of the angle by ~|a_pq|^3 — bit-level agreement with its
// libm decides the pipe point there, nowhere else (measured on 2.5 M random rays, the oracle with either trigonometry:
// without this branch up to 5e-9 m apart on the pipe point above |slope| 1e4; with it <= 4e-15 m at every slope, and
// <= 2.2e-13 m on the landing point, the level of the rays next to the critical angle).  tests/golden/edge_cfg.npz
// `offtx`, ray 470 (a_pq = 14,217) is the fixture that finds it.
template <bool FAST>
__device__ __forceinline__ void trace_ray(const ShootArgs& a, const RayIn& in, RayOut& out)
{
    const LensK& k = a.k;
    const int n = a.n;
    const double2 P = in.P;
    const double xa = in.xa, za = in.za, r_outer = in.r_outer, off = in.off, zf = in.zf;

    // --- element -> lens, refraction lens -> water (main_rt.py:338-349) -----------------------
    double a_pq, ux = 0.0, uz = 0.0;               // slope of the refracted line; FAST: its unit direction
    double phi_pq = 0.0;
    if (!FAST) {
        const double phis = in.phis;
        const double eta = k.eta21;                                     // c2 / c1, host-rounded (the same IEEE division)
        const double phi_ap = rtus_atan2(za - P.y, xa - P.x);           // :341
        const double theta_1 = phi_ap - (phis + RTUS_PI_2);             // :267-280 refraction, tuple branch
        const double sn = eta * rtus_sin(theta_1);
        // entering the slower medium |eta sin| stays below 1/2: the wave-uniform test drops asin's second form
        phi_pq = phis - RTUS_PI_2 + (eta < 0.5 ? rtus_asin_small(sn) : rtus_asin(sn));   // :345
        bool steep;
        a_pq = rtus_tan(phi_pq, steep);                                // :348
        if (__any(steep)) {                                            // wave-uniform, rare: see the note above
            const double th = atan2(za - P.y, xa - P.x) - (phis + RTUS_PI_2);
            phi_pq = phis - RTUS_PI_2 + asin(eta * sin(th));
            a_pq = tan(phi_pq);
        }
    } else {
#pragma clang fp contract(fast)   // vector-form mode is not bound to NumPy's multiply-then-add rounding
        // Same law without angles.  With t = unit tangent, n = (-tz, tx), v = unit(A - P):
        // sin(theta_1) = sin(phi_ap - phi_n) = -(t.v); theta_2 = asin(eta sin theta_1) (|.|>1 -> NaN = TIR);
        // direction at phi_pq = phi_s - pi/2 + theta_2 is  u = -n cos(theta_2) + t sin(theta_2).
        const double2 t = in.tu;
        const double vx = xa - P.x, vz = za - P.y;
        const double s2 = -k.eta21 * (t.x * vx + t.y * vz) * rsqrt_fast(vx * vx + vz * vz);
        const double c2 = m_sqrt<true>(1.0 - s2 * s2);
        uz = -t.x * c2 + t.y * s2;
        ux = cap_vertical(t.y * c2 + t.x * s2, uz);
        a_pq = uz * rcp_fast(ux);
    }
    const double b_pq = P.y - a_pq * P.x;                              // :349

    // --- line ∩ circle, keep the upper root (main_rt.py:351-364) ------------------------------
    const double qA = a_pq * a_pq + 1.0;
    const double qB = 2.0 * (a_pq * b_pq - off);
    const double qC = off * off + b_pq * b_pq - r_outer * r_outer;
    const double sq = m_sqrt<FAST>(qB * qB - 4.0 * qA * qC);
    const double den = 2.0 * qA;
    double xq1, xq2;
    if (FAST) { const double rd = rcp_fast(den); xq1 = (-qB + sq) * rd; xq2 = (-qB - sq) * rd; }
    else rtus_div2(-qB + sq, -qB - sq, den, xq1, xq2);
    const double zq1 = a_pq * xq1 + b_pq, zq2 = a_pq * xq2 + b_pq;
    const bool upper = zq1 > zq2;
    const double xq = upper ? xq1 : xq2, zq = upper ? zq1 : zq2;

    // --- reflection on the pipe (main_rt.py:367-376); tangent ignores pipe_offset --
    // RTUS_TRUE_PIPE_TANGENT (not the reference): tangent of the circle where it actually is
    const double xt = (a.flags & RTUS_TRUE_PIPE_TANGENT) ? xq - off : xq;
    const double slope = FAST ? -xt * rsqrt_fast(r_outer * r_outer - xt * xt)
                              : rtus_div(-xt, rtus_sqrt(r_outer * r_outer - xt * xt));   // :237-238
    double m, phi_l = 0.0, lx_u = 0.0, lz_u = 0.0;
    if (!FAST) {
        const double phi_sl = rtus_atan(slope);                        // :287
        phi_l = phi_sl - RTUS_PI_2 - (phi_pq - (phi_sl + RTUS_PI_2));  // :289-291
        m = rtus_tan(phi_l);                                           // :375
    } else {
#pragma clang fp contract(fast)
        // phi_l = 2 phi_sl - phi_pq with tan(phi_sl) = slope: rotate u by 2 phi_sl and mirror.
        const double s_2 = slope * slope, inv = rcp_fast(1.0 + s_2);
        lz_u = (2.0 * slope * ux - (1.0 - s_2) * uz) * inv;
        lx_u = cap_vertical(((1.0 - s_2) * ux + 2.0 * slope * uz) * inv, lz_u);
        m = lz_u * rcp_fast(lx_u);
    }
    const double b = zq - m * xq;                                      // :376

    // --- first sign change of d_j along the polyline (main_rt.py:78-99) -----------------------
    const bool fin = isfinite(m) && isfinite(b);   // non-finite line -> the reference ends in (None, None)
    // polyline extent (for the rounding part of the margin): kept in the first record of the tree
    const double xabs = a.tree[a.n_tree].xc, zabs = a.tree[a.n_tree].xh;
    Walk W;
    // 2e-8 >= the reference's isclose(d, 0) atol, so a skipped box can hold no "point on the line"
    // (main_rt.py:86); the relative part is ~450x the worst fp64 rounding of d_j.
    const double marg0 = 2e-8 + 1e-13 * (fma(fabs(m), xabs, fabs(b)) + zabs);
    lanemask c0pos, c0neg;                          // class of polyline point 0 per ray (neither bit: d_0 == 0)
    {
        const double2 c0 = a.curve[0];
        const double t0 = fma(m, c0.x, b);
        c0pos = __ballot(c0.y > t0); c0neg = __ballot(c0.y < t0);       // np.sign(d_0)
    }
    const lanemask c0zero = ~(c0pos | c0neg);
    W.sg = lane_bit(c0neg) ? -1.0 : 1.0;
    W.ms = W.sg * m; W.bs = W.sg * b; W.am = fabs(m);
    W.marg = lane_bit(c0zero) ? INFINITY : marg0;
    W.slot = 0; W.start = 0;
    W.active = __ballot(fin);                       // non-finite lines: nothing to find
    int idx = -1;                                   // p - 1 (segment idx .. idx+1 holds the first sign change), -1 if none
    // Lock-step part: boxes are visited in index order with wave-uniform indices (scalar loads); a ray
    // leaves the walk when a box certifies its answer or when it reaches an 8-point leaf it cannot
    // decide — that leaf it then reads itself (per-lane gather), so the wave never evaluates the
    // union of all 64 rays' leaves point by point.
    for (int pass = 0; pass < 4096; ++pass) {       // > 1 pass only if a parked leaf turned out to hold no change
        const lanemask active0 = W.active;
#ifdef RTUS_WALK_CXX   // the same pass from the C++ template (for comparison builds)
        if (pass == 0) walk_pass<false>(W, a.tree, a.n_tree); else walk_pass<true>(W, a.tree, a.n_tree);
#else
        if (pass == 0) walk_first_pass(W, a.tree, a.n_tree); else walk_pass<true>(W, a.tree, a.n_tree);
#endif
        // Rays still active walked off the end: no class change anywhere, idx stays -1.  Every ray that left the walk
        // did so at a box whose first 8 points hold its answer.
        const lanemask left = active0 & ~W.active;
        W.active = 0;
        if (!left) break;
        DBG(3);
        // Per-lane leaf: 8 polyline points (the array is padded to a multiple of 8 with copies of the last
        // point: a copy never changes class, so the padding cannot produce a hit).
        const bool mine = lane_bit(left);
        const double2* __restrict__ cp = a.curve + W.slot;
        double2 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) c[i] = cp[i];                        // 8 gathers in flight together
        int hit = -1;
        if (!c0zero) {
            // np.sign(d_j) != c0 with the line in its signed form (sg d > 0 <=> "same class as point 0"; the products by
            // +-1 are exact): one multiply, one fma, one compare and the select per point, nothing on the scalar unit
#pragma unroll
            for (int i = 7; i >= 0; --i) hit = (W.sg * c[i].y > fma(W.ms, c[i].x, W.bs)) ? hit : i;
        } else {
            // some ray's point 0 lies exactly on its line (class 0 must stay ==): the three-class form, as lane masks
#pragma unroll
            for (int i = 7; i >= 0; --i) {
                const double t = fma(m, c[i].x, b);
                const lanemask differs = (c0pos & ~__ballot(c[i].y > t)) | (c0neg & ~__ballot(c[i].y < t)) |
                                         (c0zero & __ballot(c[i].y != t));
                hit = lane_bit(differs) ? i : hit;
            }
        }
        const bool got = mine && hit >= 0;
        idx = got ? W.slot + hit - 1 : idx;
        W.start = (mine && !got) ? W.slot + 8 : W.start;             // nothing here: resume after this leaf
        W.active = __ballot(mine && !got);
        if (!W.active) break;
    }

    double xi = NAN, zi = NAN;
    // No sign change anywhere: first polyline point within isclose(d, 0) of the line, else None
    // (main_rt.py:84-96).  Such a point can only sit in a box the line could not be certified
    // against, so the same walk finds it; lines that miss the lens by a clear margin cost only
    // the top-level box tests.
    const lanemask need_on = __ballot(fin && idx < 0);
    if (need_on) {
        DBG(4);
        lanemask on_found = 0;
        int on = -1;
        for (int S = 0; S < a.n2; ++S) {
            lanemask pos, neg;
            certify(W, a.node2[S], marg0, pos, neg);
            DBG(5);
            if (!(need_on & ~(pos | neg))) continue;
            const int B1 = min(S * 8 + 8, a.n1);
            for (int B = S * 8; B < B1; ++B) {
                certify(W, a.node1[B], marg0, pos, neg);
                if (!(need_on & ~(pos | neg))) continue;
                const int U1 = min(B * 8 + 8, a.n0);
                for (int U = B * 8; U < U1; ++U) {
                    certify(W, a.node0[U], marg0, pos, neg);
                    DBG(6);
                    if (!(need_on & ~(pos | neg))) continue;
                    DBG(7);
                    const int j1 = min(U * 8 + 8, n);
                    for (int j = U * 8; j < j1; ++j) {
                        const double2 c = a.curve[j];
                        const double dj = c.y - (m * c.x + b);        // :78-79, NumPy rounding
                        const lanemask hit = __ballot(fabs(dj) <= 1e-8) & ~on_found;   // :86 isclose(diffs, 0)
                        on_found |= hit;
                        on = lane_bit(hit) ? j : on;
                    }
                }
            }
        }
        if (fin && idx < 0 && on >= 0) { const double2 c = a.curve[on]; xi = c.x; zi = c.y; }   // :88-90
    }
    if (idx >= 0) {                                                    // main_rt.py:106-168
        const double2 c1p = a.curve[idx], c2p = a.curve[idx + 1];
        const double x1 = c1p.x, y1 = c1p.y, x2 = c2p.x, y2 = c2p.y;
        if (np_isclose(x1, x2, 1e-5, 1e-8)) {                          // :110-123 vertical segment
            const double y = m * x1 + b;
            if (y >= fmin(y1, y2) - 1e-9 && y <= fmax(y1, y2) + 1e-9) { xi = x1; zi = y; }
        } else {
            const double m_seg = m_div<FAST>(y2 - y1, x2 - x1);         // :127-128
            const double b_seg = y1 - m_seg * x1;
            if (np_isclose(m, m_seg, 1e-5, 1e-8)) {                    // :131-144
                if (np_isclose(b, b_seg, 1e-5, 1e-8)) { xi = (x1 + x2) / 2.0; zi = m * xi + b; }
            } else {
                const double x = m_div<FAST>(b_seg - b, m - m_seg);     // :147
                const double y = m * x + b;                            // :150
                const double xlo = x1 < x2 ? x1 : x2, xhi = x1 < x2 ? x2 : x1;
                const double ylo = y1 < y2 ? y1 : y2, yhi = y1 < y2 ? y2 : y1;
                if (x >= xlo - 1e-9 && x <= xhi + 1e-9 && y >= ylo - 1e-9 && y <= yhi + 1e-9) {   // :157-158
                    xi = x; zi = y;
                }
            }
        }
    }

    // RTUS_ANALYTIC_LENS (not the reference): slide the chord intersection onto the analytic curve,
    // Newton on alpha for z(alpha) = m x(alpha) + b from the chord point's polar angle.
    if (a.flags & RTUS_ANALYTIC_LENS) {                                // wave-uniform flag
        double al = atan2(xi, zi), lx = xi, lz = zi;
        for (int it = 0; it < 20; ++it) {
            double ldz, ldx;
            lens_eval(k, al, lx, lz, ldz, ldx);
            const double step = (lz - (m * lx + b)) / (ldz - m * ldx);
            if (!__any(fabs(step) > 1e-15)) break;
            al -= (fabs(step) > 1e-15) ? step : 0.0;
        }
        xi = isnan(xi) ? xi : lx; zi = isnan(zi) ? zi : lz;
    }

    // --- refraction water -> lens and landing on z = z_f (main_rt.py:396-405) ------------------
    double a3;
    if (!FAST) {
        // :396-397 alpha_i = atan2(x_i, z_i) is only ever used through its sine and cosine: x_i / rho, z_i / rho
        const double rho = rtus_sqrt(xi * xi + zi * zi);
        double lx, lz, ldz, ldx;
        double si, ci;
        rtus_div2(xi, zi, rho, si, ci);
        lens_eval_sc(k, si, ci, lx, lz, ldz, ldx);         // analytic tangent at the chord point's polar angle
        const double phi_last = refract_angle(phi_l, rtus_atan2(ldz, ldx), k.eta12);   // :398
        a3 = rtus_tan(phi_last);                                       // :401
    } else {
#pragma clang fp contract(fast)
        // sin / cos of alpha_i = atan2(xi, zi) are xi/rho, zi/rho; then the refraction law as above.
        const double rr = rsqrt_fast(xi * xi + zi * zi);
        const double si = xi * rr, ci = zi * rr;
        const double B = k.phi_3 * ci - k.twoTc;
        const double dsc = B * B - k.C4A, rsqd = rsqrt_fast(dsc), sqd = dsc * rsqd;
        const double h = -(-B - sqd) * k.phi_1;                         // phi_1 = -1/(2A)
        const double dB = -k.phi_3 * si;
        const double dh = k.phi_1 * (dB + B * dB * rsqd);
        double tz = dh * ci - h * si, tx_ = dh * si + h * ci;          // (dz, dx)
        const double rt = rsqrt_fast(tx_ * tx_ + tz * tz);
        tx_ *= rt; tz *= rt;
        const double s2 = -k.eta12 * (tx_ * lx_u + tz * lz_u);   // u_l is a unit vector
        const double c2 = m_sqrt<true>(1.0 - s2 * s2);                 // NaN = total internal reflection
        const double w3z = -tx_ * c2 + tz * s2;
        a3 = w3z * rcp_fast(cap_vertical(tz * c2 + tx_ * s2, w3z));
    }
    const double b3 = zi - a3 * xi;                                    // :402
    const double x_in = m_div<FAST>(zf - b3, a3);                       // :404

    out.xq = xq; out.zq = zq; out.xi = xi; out.zi = zi; out.x_in = x_in;
}

// ---- the forward trace: one ray of the alpha grid per lane --------------------------------------
// 6 waves per SIMD (<= 80 VGPRs; the unconstrained allocation of 81 fell one register short): measured +6 %.
#ifndef RTUS_SHOOT_MIN_WAVES
#define RTUS_SHOOT_MIN_WAVES 6
#endif
template <bool FAST>
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
    if (FAST) in.tu = a.tan_u[r]; else in.phis = a.phi_s[r];
    in.zf = a.z_f[r];
    RayOut o_;
    trace_ray<FAST>(a, in, o_);
    const double2 P = in.P;
    const double xa = in.xa, za = in.za, zf = in.zf;
    const double xq = o_.xq, zq = o_.zq, xi = o_.xi, zi = o_.zi, x_in = o_.x_in;

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

// ---- pulse-echo root-finding solve ------------------------------------------------------------
// The reference matches elements to whichever GRID ray happens to land within a tolerance
// (main_rt.py:487-501).  Here x_land(alpha) = x_rx is solved: the grid trace (rtus_shoot_kernel)
// only brackets the roots — consecutive finite grid rays whose landing points straddle the element —
// and each bracket is refined with the Illinois variant of regula falsi, every evaluation being a
// full trace_ray at that lane's own alpha.  x_land(alpha) is U-shaped, so an element usually has two
// ray paths; all (up to RTUS_MAX_ROOTS) are returned in ascending alpha, plus the least-time one.
// Lanes = consecutive rx elements of one (geometry, tx) row; brackets are found by a lock-step scan of
// the row's land_x (wave-uniform index -> scalar loads) against each lane's own element.
struct SolveArgs {
    ShootArgs s;                        // lens, geometry, tx, polyline + boxes, flags
    const double* __restrict__ alpha;   // [n]  grid
    const double* __restrict__ land_x;  // [rows][n] landing x of the grid rays on z = z_land
    const double2* __restrict__ land_box;  // [rows][nb] (min, max) of the finite landing x of rays 64B .. 64B+64
    int nb;
    const double* __restrict__ x_rx;    // [n_rx]
    double z_land;
    int n_rx;
    double* __restrict__ tt;            // [rows][n_rx] least-time root (NaN: none)
    double* __restrict__ alpha_root;    // nullable, its launch angle
    double* __restrict__ tt_all;        // nullable [rows][n_rx][RTUS_MAX_ROOTS]
    double* __restrict__ alpha_all;     // nullable [rows][n_rx][RTUS_MAX_ROOTS]
    uint8_t* __restrict__ n_roots;      // nullable [rows][n_rx]
    int row0;                           // first (geometry, tx) row of this launch (grid.y <= 65535)
};

template <bool FAST>
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_solve_kernel(SolveArgs q)
{
    const ShootArgs& a = q.s;
    const LensK& k = a.k;
    const int n = a.n;
    const int e_raw = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    const bool live = e_raw < q.n_rx;
    const int e = live ? e_raw : q.n_rx - 1;
    const int row = q.row0 + blockIdx.y, g = row / a.n_tx, tx = row - g * a.n_tx;
    const double xe = q.x_rx[e];
    const double* __restrict__ lrow = q.land_x + (size_t)row * n;

    // brackets: grid pairs (r, r+1), both finite, f(r) != 0 and f(r+1) on the other side or zero
    int b0 = -1, b1 = -1, b2 = -1, b3 = -1, cnt = 0;
    // the wave's elements span [xe_lo, xe_hi]; a 64-pair block of grid rays whose landing points all lie
    // outside that span cannot bracket any of them and is skipped (wave-uniform decision, scalar loads)
    double xe_lo = xe, xe_hi = xe;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { xe_lo = fmin(xe_lo, __shfl_xor(xe_lo, o)); xe_hi = fmax(xe_hi, __shfl_xor(xe_hi, o)); }
    const double2* __restrict__ brow = q.land_box + (size_t)row * q.nb;
    for (int B = 0; B < q.nb; ++B) {
        const double2 bx = brow[B];
        if (bx.x > xe_hi || bx.y < xe_lo || !(bx.x <= bx.y)) continue;
        const int rb = B * 64;
        double fprev = lrow[rb] - xe;
        for (int r0 = rb + 1; r0 <= min(rb + 64, n - 1); r0 += 8) {      // 8 grid rays per trip: one batch of scalar loads
            double lx[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) lx[i] = lrow[min(r0 + i, n - 1)];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = r0 + i;
                const double fcur = lx[i] - xe;
                const bool in = r < n && r <= rb + 64;
                const bool st = in && isfinite(fprev) && isfinite(fcur) &&
                                ((fprev < 0.0 && fcur >= 0.0) || (fprev > 0.0 && fcur <= 0.0));
                if (st) {
                    b0 = cnt == 0 ? r - 1 : b0; b1 = cnt == 1 ? r - 1 : b1;
                    b2 = cnt == 2 ? r - 1 : b2; b3 = cnt == 3 ? r - 1 : b3;
                    ++cnt;
                }
                fprev = in ? fcur : fprev;
            }
        }
    }
    cnt = min(cnt, RTUS_MAX_ROOTS);

    RayIn in;
    in.r_outer = a.geoms[2 * g]; in.off = a.geoms[2 * g + 1];
    in.xa = a.x_a[tx]; in.za = a.z_a[tx]; in.zf = q.z_land;
    double tmin = NAN, amin = NAN, tk[RTUS_MAX_ROOTS], ak[RTUS_MAX_ROOTS];
    int nr = 0;
#pragma unroll
    for (int kk = 0; kk < RTUS_MAX_ROOTS; ++kk) {
        tk[kk] = NAN; ak[kk] = NAN;
        const int br = kk == 0 ? b0 : (kk == 1 ? b1 : (kk == 2 ? b2 : b3));
        const bool mine = kk < cnt;
        if (!__any(mine)) continue;                                    // wave-uniform
        const int bs = mine ? br : 0;
        double al = q.alpha[bs], ah = q.alpha[bs + 1];
        double fl = lrow[bs] - xe, fh = lrow[bs + 1] - xe;
        double ac = ah, fc = fh;
        RayOut o;
        bool ok = mine && fh != 0.0;                                   // fh == 0: the grid ray itself is the root
        bool dead = false;
        for (int it = 0; it < 64; ++it) {
            const bool work = ok && !dead && fabs(fc) > 1e-13 && fabs(ah - al) > 1e-13;   // |f| < 0.1 pm or bracket < 1e-13 rad (dT/dalpha ~ 1e-5 s/rad)
            if (!__any(work) && it > 0) break;
            if (work || it == 0) {
                double cand = m_div<FAST>(al * fh - ah * fl, fh - fl);
                if (!(cand > fmin(al, ah) && cand < fmax(al, ah))) cand = 0.5 * (al + ah);
                ac = (work) ? cand : ac;
            }
            double px, pz, dz, dx;
            lens_eval(k, ac, px, pz, dz, dx);                          // main_rt.py:338, 344 at this lane's alpha
            in.P = make_double2(px, pz);
            if (FAST) { const double rt = rsqrt_fast(dx * dx + dz * dz); in.tu = make_double2(dx * rt, dz * rt); }
            else in.phis = atan2(dz, dx);
            trace_ray<FAST>(a, in, o);                                 // all 64 lanes together
            fc = o.x_in - xe;
            if (work) {
                if (!isfinite(fc)) dead = true;                        // the branch ends inside the bracket
                else if ((fc < 0.0) == (fh < 0.0)) { ah = ac; fh = fc; fl *= 0.5; }   // Illinois: halve the stale end
                else { al = ah; fl = fh; ah = ac; fh = fc; }
            }
        }
        const bool root = mine && !dead && fabs(fc) < 1e-9;             // |f| large at convergence: a jump, not a root
        const double t1 = seg_time<FAST>(in.xa, in.za, in.P.x, in.P.y, k.c1, k.inv_c1);
        const double t2 = seg_time<FAST>(in.P.x, in.P.y, o.xq, o.zq, k.c2, k.inv_c2);
        const double t3 = seg_time<FAST>(o.xq, o.zq, o.xi, o.zi, k.c2, k.inv_c2);
        const double t4 = seg_time<FAST>(o.xi, o.zi, o.x_in, q.z_land, k.c1, k.inv_c1);
        const double T = ((t1 + t2) + t3) + t4;
        if (root) {
            // roots are stored compacted (ascending alpha)
            tk[0] = nr == 0 ? T : tk[0]; ak[0] = nr == 0 ? ac : ak[0];
            tk[1] = nr == 1 ? T : tk[1]; ak[1] = nr == 1 ? ac : ak[1];
            tk[2] = nr == 2 ? T : tk[2]; ak[2] = nr == 2 ? ac : ak[2];
            tk[3] = nr == 3 ? T : tk[3]; ak[3] = nr == 3 ? ac : ak[3];
            if (!(tmin <= T)) { tmin = T; amin = ac; }
            ++nr;
        }
    }
    if (!live) return;
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

// ---- host-side launchers (called from rtus_capi.hip) -----------------------------------------
static size_t align32(size_t v) { return (v + 31) & ~(size_t)31; }
static int pad8(int n) { return (n + 7) & ~7; }
static size_t ws_phis_off(int n) { return align32((size_t)pad8(n) * sizeof(double2)); }
static size_t ws_tanu_off(int n) { return align32(ws_phis_off(n) + (size_t)n * sizeof(double)); }
static size_t ws_node0_off(int n) { return align32(ws_tanu_off(n) + (size_t)n * sizeof(double2)); }
static size_t ws_node1_off(int n) { return ws_node0_off(n) + (size_t)((n + 7) / 8) * sizeof(double4); }
static size_t ws_node2_off(int n) { return ws_node1_off(n) + (size_t)((n + 63) / 64) * sizeof(double4); }
static int n_tree_nodes(int n) { return (n + 7) / 8 + (n + 63) / 64 + (n + 511) / 512 + (n + 4095) / 4096 + 1; }   // upper bound (node3 may be unused) + the extent record
static size_t ws_tree_off(int n) { return (ws_node2_off(n) + (size_t)((n + 511) / 512) * sizeof(double4) + 63) & ~(size_t)63; }
size_t rtus_ws_bytes(int n) { return ws_tree_off(n) + (size_t)n_tree_nodes(n) * sizeof(TreeNode); }

// Workspace pointers and sizes of a ShootArgs (the workspace base must be 64-byte aligned: hipMalloc gives 256).
static void shoot_args_workspace(ShootArgs& a, char* w, int n)
{
    a.curve = (const double2*)w;
    a.phi_s = (const double*)(w + ws_phis_off(n));
    a.tan_u = (const double2*)(w + ws_tanu_off(n));
    a.node0 = (const double4*)(w + ws_node0_off(n));
    a.node1 = (const double4*)(w + ws_node1_off(n));
    a.node2 = (const double4*)(w + ws_node2_off(n));
    a.tree = (const TreeNode*)(w + ws_tree_off(n));
    a.n = n;
    a.n0 = (n + 7) / 8; a.n1 = (n + 63) / 64; a.n2 = (n + 511) / 512;
    a.n3 = a.n2 > 8 ? (n + 4095) / 4096 : 0;
    a.n_tree = a.n0 + a.n1 + a.n2 + a.n3;
}

hipError_t rtus_launch_shoot(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, const double* z_f, int n,
                             double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                             void* ws, unsigned flags, hipStream_t s)
{
    char* w = (char*)ws;
    ShootArgs a;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = z_f;
    shoot_args_workspace(a, w, n);
    if ((unsigned long long)a.n_tree * sizeof(TreeNode) >= 0xffffffffull) return hipErrorInvalidValue;   // the walk's record offsets are 32-bit
    a.out8 = out8; a.tof4 = tof4; a.tof = tof; a.land_x = land_x; a.status = status;
    a.n_tx = n_tx; a.n_geom = n_geom;
    a.flags = flags;
    hipLaunchKernelGGL(rtus_curve_kernel, dim3(a.n2), dim3(RTUS_CURVE_TPB), 0, s, a.k, alpha, n,
                       (double2*)a.curve, (double*)a.phi_s, (double2*)a.tan_u, (double4*)a.node0, (double4*)a.node1,
                       (double4*)a.node2);
    hipLaunchKernelGGL(rtus_tree_kernel, dim3((a.n_tree + 255) / 256), dim3(256), 0, s, a.node0, a.node1, a.node2, a.n0, a.n1, a.n2,
                       a.n3, (TreeNode*)a.tree);
    const dim3 grid((n + RTUS_BLOCK - 1) / RTUS_BLOCK, n_tx, n_geom);
    if (flags & RTUS_SHOOT_FAST_MATH) hipLaunchKernelGGL(rtus_shoot_kernel<true>, grid, dim3(RTUS_BLOCK), 0, s, a);
    else hipLaunchKernelGGL(rtus_shoot_kernel<false>, grid, dim3(RTUS_BLOCK), 0, s, a);
    return hipGetLastError();
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
    hipLaunchKernelGGL(rtus_curve_kernel, dim3(a.n2), dim3(RTUS_CURVE_TPB), 0, s, a.k, d_alpha, n, (double2*)a.curve, (double*)a.phi_s,
                       (double2*)a.tan_u, (double4*)a.node0, (double4*)a.node1, (double4*)a.node2);
    hipLaunchKernelGGL(rtus_tree_kernel, dim3((a.n_tree + 255) / 256), dim3(256), 0, s, a.node0, a.node1, a.node2, a.n0, a.n1, a.n2,
                       a.n3, (TreeNode*)a.tree);
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

// (min, max) of the finite landing x over the 65 grid rays 64B .. 64B+64 of one row: one wave per block.
__global__ __launch_bounds__(64) void rtus_land_box_kernel(const double* __restrict__ land, int n, int nb,
                                                          long long n_boxes, double2* __restrict__ box)
{
    const long long id = blockIdx.x;
    if (id >= n_boxes) return;
    const long long row = id / nb;
    const int B = (int)(id - row * nb), l = threadIdx.x;
    const double* __restrict__ lr = land + row * (long long)n;
    const int r0 = B * 64 + l;
    const double v0 = r0 < n ? lr[r0] : NAN, v1 = (l == 63 && r0 + 1 < n) ? lr[r0 + 1] : NAN;
    double lo = fmin(isfinite(v0) ? v0 : INFINITY, isfinite(v1) ? v1 : INFINITY);
    double hi = fmax(isfinite(v0) ? v0 : -INFINITY, isfinite(v1) ? v1 : -INFINITY);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo = fmin(lo, __shfl_xor(lo, o)); hi = fmax(hi, __shfl_xor(hi, o)); }
    if (l == 0) box[id] = make_double2(lo, hi);
}

// Workspace of the solve = shoot workspace + z_f[n] (all z_land) + land_x[rows][n] + land boxes[rows][nb].
static size_t sws_zf_off(int n) { return align32(rtus_ws_bytes(n)); }
static size_t sws_land_off(int n) { return align32(sws_zf_off(n) + (size_t)n * sizeof(double)); }
static size_t sws_box_off(int n, int n_geom, int n_tx) { return align32(sws_land_off(n) + (size_t)n_geom * n_tx * (size_t)n * sizeof(double)); }
size_t rtus_solve_ws_bytes(int n, int n_geom, int n_tx)
{
    return sws_box_off(n, n_geom, n_tx) + (size_t)n_geom * n_tx * (size_t)((n + 63) / 64) * sizeof(double2);
}

__global__ void rtus_fill_kernel(double* __restrict__ p, double v, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

hipError_t rtus_launch_solve(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, int n, const double* x_rx, int n_rx,
                             double z_land, double* tt, double* alpha_root, double* tt_all,
                             double* alpha_all, uint8_t* n_roots, void* ws, unsigned flags, hipStream_t s)
{
    char* w = (char*)ws;
    double* z_f_scratch = (double*)(w + sws_zf_off(n));
    double* land = (double*)(w + sws_land_off(n));
    // 1. grid trace with z_f = z_land for every ray
    hipLaunchKernelGGL(rtus_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, s, z_f_scratch, z_land, n);
    hipError_t e = rtus_launch_shoot(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, z_f_scratch, n, nullptr, nullptr,
                                     nullptr, land, nullptr, ws, flags, s);
    if (e != hipSuccess) return e;
    // 2. per-row bounding intervals of the landing points, 64 ray pairs each
    const long long rows_ll = (long long)n_geom * n_tx;
    const int nb = (n + 63) / 64;
    double2* boxes = (double2*)(w + sws_box_off(n, n_geom, n_tx));
    hipLaunchKernelGGL(rtus_land_box_kernel, dim3((unsigned)(rows_ll * nb)), dim3(64), 0, s, land, n, nb, rows_ll * nb, boxes);
    // 3. bracket + refine
    SolveArgs q;
    ShootArgs& a = q.s;
    a.k = make_lens_k(lens);
    a.geoms = geoms; a.x_a = x_a; a.z_a = z_a; a.z_f = z_f_scratch;
    shoot_args_workspace(a, w, n);
    a.out8 = nullptr; a.tof4 = nullptr; a.tof = nullptr; a.land_x = nullptr; a.status = nullptr;
    a.n_tx = n_tx; a.n_geom = n_geom;
    a.flags = flags;
    q.alpha = alpha; q.land_x = land; q.land_box = boxes; q.nb = nb; q.x_rx = x_rx; q.z_land = z_land; q.n_rx = n_rx;
    q.tt = tt; q.alpha_root = alpha_root; q.tt_all = tt_all; q.alpha_all = alpha_all; q.n_roots = n_roots;
    const long long rows = (long long)n_geom * n_tx;
    for (long long row0 = 0; row0 < rows; row0 += 65535) {
        q.row0 = (int)row0;
        const dim3 grid((n_rx + RTUS_BLOCK - 1) / RTUS_BLOCK, (unsigned)((rows - row0) < 65535 ? (rows - row0) : 65535));
        if (flags & RTUS_SHOOT_FAST_MATH) hipLaunchKernelGGL(rtus_solve_kernel<true>, grid, dim3(RTUS_BLOCK), 0, s, q);
        else hipLaunchKernelGGL(rtus_solve_kernel<false>, grid, dim3(RTUS_BLOCK), 0, s, q);
    }
    return hipGetLastError();
}
