// rtus_trace.h — what the forward-trace kernels (rtus_shoot.hip) and the root-finding solve (rtus_solve.hip) share: the
// launch arguments, the depth-first box records, the lock-step crossing search and trace_ray (one ray: lens point ->
// pipe -> lens -> landing; reference shoot_rays, main_rt.py:337-405), and the layout of the device workspace.
#pragma once
#include "rtus_device.h"

// Apertures up to this many receive elements take the solve's fast path (pair masks written by the grid trace); larger
// ones the lock-step scan of the landing points inside rtus_solve_kernel.
#define RTUS_SOLVE_MASK_MAX_RX 512

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

struct ShootArgs {
    LensK k;
    const double* __restrict__ geoms;   // [n_geom][2]
    const double* __restrict__ x_a;     // [n_tx]
    const double* __restrict__ z_a;     // [n_tx]
    const double* __restrict__ z_f;     // [n]; nullptr: every ray lands on z = zf_const (the solve's grid trace)
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
    double2* __restrict__ land_box;     // nullable [n_geom][n_tx][ceil(n/64)]: (min, max) of the finite landing x of rays 64B .. 64B+63
    // the solve's grid trace (rtus_shoot_kernel<., true>): which receive elements does each pair of consecutive rays bracket
    unsigned long long* __restrict__ pair_mask;   // [n_geom][n_tx][ceil(n/64)][rx_pad]: bit i of [row][B][e] = pair (64B+i, 64B+i+1) brackets element e (i < 63)
    const double* __restrict__ x_rx;    // [n_rx] receive elements (n_rx <= RTUS_SOLVE_MASK_MAX_RX)
    int n_rx, rx_pad;                   // rx_pad = n_rx rounded up to a multiple of 64
    // the fused sweep (rtus_shoot_kernel<., ., true>): the element matcher of main_rt.py:487-501 asked while the landing points are in registers
    int32_t* __restrict__ m_first;      // scratch [rows][rx_pad]: smallest matching ray index so far, RTUS_NO_RAY when idle
    double atol, rtol;
    int tof_lazy;                       // fused sweep without a per-ray `tof` output: only waves that matched an element work out their rays' times
    double zf_const;
    int n, n_tx, n_geom, n0, n1, n2, n3, n_tree;    // n3: 4096-point boxes, only when n2 > 8 (else 0); they live in the tree only
    unsigned flags;
};

// Crossing search state.  The reference wants the first j with sign(d_j) != sign(d_{j+1})
// (main_rt.py:82, 99).  Every point before that change has the class c0 = np.sign(d_0) of polyline
// point 0, so equivalently: p = the first polyline point whose class differs from c0, idx = p - 1.
// Everything that is one bit per ray lives in 64-bit lane masks (SGPR pairs): v_cmp writes them
// directly and the bookkeeping is scalar-ALU work.
typedef unsigned long long lanemask;
#define RTUS_NO_RAY 0x7f7f7f7f   // hipMemsetAsync(0x7f) sentinel of the matcher's "first ray"; larger than any ray index
#ifdef RTUS_EXP_COUNT
static __device__ unsigned long long rtus_dbg[8];   // (one copy per translation unit; rtus_dbg_read reads rtus_shoot.hip's)
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
struct RayIn { double2 P; double phis; double2 tu; double xa, za, r_outer, off, zf; };   // phis: compat; tu: both modes
struct RayOut { double xq, zq, xi, zi, x_in; };

// Reference-compatible mode (FAST = false).  Round 1 - 2: the reference's angle arithmetic, operation for operation (sin / tan /
// atan2 / atan / asin through the kernels of rtus_trig.h).  Since round 3 only phi_pq = phi_s - pi/2 + asin(eta sin theta_1) and
// a = tan(phi_pq) are still formed as angles; sin theta_1, the reflection and the exit refraction are algebraic restatements (a few
// ulp from the reference's chain; include/rtus.h "flags for rtus_shoot*" says what that guarantees and what it does not — e.g. a ray
// within a few ulp of the critical angle at the exit may be NaN here and finite there).  Division and square root are correctly
// rounded (rtus_div, rtus_sqrt); the lens at alpha_i = atan2(x_i, z_i) goes through x_i / rho, z_i / rho (no angle formed).
// ONE place keeps the library's routines: the first refraction of a wave
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
    const TreeNode* __restrict__ tree = a.tree;
    const double2* __restrict__ curve = a.curve;
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
        double sn;
        if (eta < 1.0) {
            // Into the slower medium (wave-uniform; the reference's lens: eta = 0.23) |eta sin(theta_1)| < 1 whatever the ray
            // does: no total-reflection decision hangs on the last bits of sin(theta_1), and the reference's
            // sin(atan2(A - P) - (phi_s + pi/2))  (:341, :267-280)  is -(t . v) / |v| with t = (cos phi_s, sin phi_s) the unit
            // tangent the polyline kernel keeps beside phi_s: a dot product and a reciprocal square root (~12 instructions,
            // ~1 ulp) for atan2 + sin (~84).  The angle phi_pq is still formed below (the reflection needs it).
            const double vx = xa - P.x, vz = za - P.y;
            sn = -eta * (fma(in.tu.x, vx, in.tu.y * vz) * rsqrt_fast(fma(vx, vx, vz * vz)));
        } else {
            const double phi_ap = rtus_atan2(za - P.y, xa - P.x);       // :341
            const double theta_1 = phi_ap - (phis + RTUS_PI_2);         // :267-280 refraction, tuple branch
            sn = eta * rtus_sin(theta_1);
        }
        // entering the slower medium |eta sin| stays below 1/2: the wave-uniform test drops asin's second form
        phi_pq = phis - RTUS_PI_2 + (eta < 0.5 ? rtus_asin_small(sn) : rtus_asin(sn));   // :345
        bool steep;
        a_pq = rtus_tan(phi_pq, steep);                                // :348
        if (__any(steep)) {                                            // wave-uniform branch, rare: see the note above
            // adopted PER LANE: what a ray gets never depends on which rays share its wave (the root-finding solve packs
            // rays of different rows into one wave, and sharded launches must reproduce the unsharded bits)
            const double th = atan2(za - P.y, xa - P.x) - (phis + RTUS_PI_2);
            const double phi_lib = phis - RTUS_PI_2 + asin(eta * sin(th));
            const double a_lib = tan(phi_lib);
            phi_pq = steep ? phi_lib : phi_pq;
            a_pq = steep ? a_lib : a_pq;
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
    double m, lx_u = 0.0, lz_u = 0.0;
    unsigned cl_neg = 0;                            // compat: bit 31 = cos(phi_l) < 0 (with m, that is phi_l's unit vector: the exit refraction's input)
    const double slope = FAST ? -xt * rsqrt_fast(r_outer * r_outer - xt * xt) : 0.0;
    if (!FAST) {
#ifdef RTUS_ANGLE_FORM_REFLECTION                            // comparison builds: the reference's angle chain, operation for operation
        const double slope_c = rtus_div(-xt, rtus_sqrt(r_outer * r_outer - xt * xt));   // :237-238
        const double phi_sl = rtus_atan(slope_c);                      // :287
        const double phi_l = phi_sl - RTUS_PI_2 - (phi_pq - (phi_sl + RTUS_PI_2));      // :289-291
        m = rtus_tan(phi_l);                                           // :375
        cl_neg = (unsigned)(int)rint(phi_l * 0.31830988618379067154) << 31;   // phi_l - k pi in [-pi/2, pi/2]: cos(phi_l) has the sign (-1)^k
#else
        // :237-238, 287-291, 375  tan(phi_l) with phi_l = 2 atan(s) - phi_pq, s = -x_t / sqrt(r^2 - x_t^2), without the angles: with
        // s = p / q (p = -x_t, q = sqrt(r^2 - x_t^2) >= 0) and a = tan(phi_pq),
        //     cos(phi_l) ~ sigma ((q^2 - p^2) + 2 p q a) = sigma D,      sin(phi_l) ~ sigma (2 p q - (q^2 - p^2) a) = sigma N
        // up to a positive factor, sigma = the sign of cos(phi_pq) — and phi_pq IS formed as an angle above.  m = N / D needs no
        // division for the slope and no atan / tan (~28 instructions for ~107); a tangent hit (q = 0: the reference's slope is
        // infinite, atan gives +-pi/2) comes out as m = -a by itself; |x_t| > r is NaN through the square root, as in
        // the reference.  The crossing search sees m a few ulps from tan(fl(phi_l)) — the rounding of another math library.
        const double q2 = r_outer * r_outer - xt * xt;
        const double pq2 = -2.0 * (xt * rtus_sqrt(q2));                // 2 p q
        const double c2q = q2 - xt * xt;                               // q^2 - p^2
        const double Np = fma(-c2q, a_pq, pq2);
        const double Dp = cap_vertical(fma(pq2, a_pq, c2q), Np);
        m = rtus_div(Np, Dp);
        // cos(phi_pq) has the sign (-1)^k, k = rint(phi_pq / pi); cos(phi_l) that sign times the sign of D
        cl_neg = ((unsigned)(int)rint(phi_pq * 0.31830988618379067154) << 31) ^ ((unsigned)__double2hiint(Dp) & 0x80000000u);
#endif
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
    const double xabs = tree[a.n_tree].xc, zabs = tree[a.n_tree].xh;
    Walk W;
    // 2e-8 >= the reference's isclose(d, 0) atol, so a skipped box can hold no "point on the line"
    // (main_rt.py:86); the relative part is ~450x the worst fp64 rounding of d_j.
    const double marg0 = 2e-8 + 1e-13 * (fma(fabs(m), xabs, fabs(b)) + zabs);
    lanemask c0pos, c0neg;                          // class of polyline point 0 per ray (neither bit: d_0 == 0)
    {
        const double2 c0 = curve[0];
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
        if (pass == 0) walk_pass<false>(W, tree, a.n_tree); else walk_pass<true>(W, tree, a.n_tree);
#else
        if (pass == 0) walk_first_pass(W, tree, a.n_tree); else walk_pass<true>(W, tree, a.n_tree);
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
        const double2* __restrict__ cp = curve + W.slot;
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
                        const double2 c = curve[j];
                        const double dj = c.y - (m * c.x + b);        // :78-79, NumPy rounding
                        const lanemask hit = __ballot(fabs(dj) <= 1e-8) & ~on_found;   // :86 isclose(diffs, 0)
                        on_found |= hit;
                        on = lane_bit(hit) ? j : on;
                    }
                }
            }
        }
        if (fin && idx < 0 && on >= 0) { const double2 c = curve[on]; xi = c.x; zi = c.y; }   // :88-90
    }
    if (idx >= 0) {                                                    // main_rt.py:106-168
        const double2 c1p = curve[idx], c2p = curve[idx + 1];
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
#ifdef RTUS_ANGLE_FORM_EXIT                                 // comparison builds: the reference's angle chain for the exit refraction too
        const double cl_ = rtus_from_bits(0x3ff0000000000000ull ^ ((uint64_t)cl_neg << 32));
        const double phi_last = refract_angle(atan2(m * cl_, cl_), rtus_atan2(ldz, ldx), k.eta12);   // :398 (phi_l back from its slope and the sign of its cosine)
        a3 = rtus_tan(phi_last);                                       // :401
#else
        // :398-401  tan(phi_s - pi/2 + asin(eta sin(phi_l - (phi_s + pi/2)))), phi_s = atan2(dz, dx), without forming an angle:
        // with the unit tangent (cs, ss) = (dx, dz) / |.| and the unit vector (cl, sl) of phi_l from the reflection above —
        //   sin(theta_1) = -cos(phi_l - phi_s) = -(cl cs + sl ss),   s2 = eta sin(theta_1),   c2 = sqrt(1 - s2^2)  (NaN beyond
        //   the critical angle: the reference's asin of |x| > 1),    tan(phi_last) = (-cs c2 + ss s2) / (ss c2 + cs s2).
        // Nothing downstream decides anything by the last bits of a3 (b3 and the landing point follow from it by one subtraction
        // and one division); what IS a decision — total reflection — is taken on the same quantity, |eta sin(theta_1)| > 1.
        // ~48 instructions for atan2 + sin + asin + tan (~163).
        const double rl = rsqrt_fast(fma(ldx, ldx, ldz * ldz));
        const double cs = ldx * rl, ss = ldz * rl;
        const double cl = rtus_from_bits(rtus_bits(rsqrt_fast(fma(m, m, 1.0))) ^ ((uint64_t)cl_neg << 32));   // (cl, sl) = +-(1, m) / sqrt(1 + m^2)
        const double sl = m * cl;
        const double s2 = -k.eta12 * fma(cl, cs, sl * ss);
        const double c2 = rtus_sqrt(fma(-s2, s2, 1.0));
        const double num = fma(ss, s2, -(cs * c2));
        a3 = rtus_div(num, cap_vertical(fma(ss, c2, cs * s2), num));
#endif
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

// ---- device workspace of a forward trace (host side) ------------------------------------------------------------------
static inline size_t align32(size_t v) { return (v + 31) & ~(size_t)31; }
static inline int pad8(int n) { return (n + 7) & ~7; }
static inline size_t ws_phis_off(int n) { return align32((size_t)pad8(n) * sizeof(double2)); }
static inline size_t ws_tanu_off(int n) { return align32(ws_phis_off(n) + (size_t)n * sizeof(double)); }
static inline size_t ws_node0_off(int n) { return align32(ws_tanu_off(n) + (size_t)n * sizeof(double2)); }
static inline size_t ws_node1_off(int n) { return ws_node0_off(n) + (size_t)((n + 7) / 8) * sizeof(double4); }
static inline size_t ws_node2_off(int n) { return ws_node1_off(n) + (size_t)((n + 63) / 64) * sizeof(double4); }
static inline int n_tree_nodes(int n) { return (n + 7) / 8 + (n + 63) / 64 + (n + 511) / 512 + (n + 4095) / 4096 + 1; }   // upper bound (node3 may be unused) + the extent record
static inline size_t ws_tree_off(int n) { return (ws_node2_off(n) + (size_t)((n + 511) / 512) * sizeof(double4) + 63) & ~(size_t)63; }
static inline size_t shoot_ws_bytes(int n) { return ws_tree_off(n) + (size_t)n_tree_nodes(n) * sizeof(TreeNode); }

// Workspace pointers and sizes of a ShootArgs (the workspace base must be 64-byte aligned: hipMalloc gives 256).
static inline void shoot_args_workspace(ShootArgs& a, char* w, int n)
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


// rtus_shoot.hip.  z_f == nullptr: every ray lands on z = zf_const; land_box: optional per-wave (min, max) of the landing points
void rtus_launch_geometry_only(const ShootArgs& a, const double* alpha, hipStream_t s);
hipError_t rtus_launch_shoot_ex(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                                const double* z_a, int n_tx, const double* alpha, const double* z_f, double zf_const, int n,
                                double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                                double2* land_box, unsigned long long* pair_mask, const double* x_rx, int n_rx,
                                void* ws, unsigned flags, hipStream_t s);
// the fused sweep: forward trace + element matcher in one kernel (main_rt.py:464-501)
size_t rtus_sweep_ws_bytes(int n, int n_geom, int n_tx, int n_rx);
hipError_t rtus_launch_sweep(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
                             const double* alpha, const double* z_f, int n, const double* x_rx, int n_rx, double atol, double rtol,
                             int32_t* first_ray, uint8_t* hit, double* tof_hit, double* tof, double* land_x, void* ws, unsigned flags,
                             hipStream_t s);
