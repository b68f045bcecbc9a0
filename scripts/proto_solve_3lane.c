/* Scratch prototype (CPU, not product): the THREE-LANES-PER-BRACKET refinement of rtus_solve (round 4) on the reference sweep's
 * brackets — candidate c by inverse quadratic interpolation through three grid rays, evaluations at c - d, c, c + d together,
 * the root by inverse quadratic interpolation through the three fresh points and T by the parabola through their times.
 * How many rounds does a bracket need for a given rule for d, and how far are alpha and T from a bisection to 1e-15 rad?
 *   gcc -O2 -ffp-contract=off -fopenmp scripts/proto_solve_3lane.c -o /tmp/proto3 -lm && /tmp/proto3
 * The numbers this printed are quoted in DESIGN.md (root-find section). */
#include "../oracle/rt_oracle.c"
#include <stdio.h>
#include <stdlib.h>

typedef struct { const orc_lens *L; double r, off, xa, za, zl; const double *xc, *zc; int n; double xe; long evals; } Ctx;
static double F(Ctx *c, double a, double *T)
{
    double o[8];
    c->evals++;
    orc_trace_alpha(c->L, c->r, c->off, c->xa, c->za, c->zl, a, c->xc, c->zc, c->n, 0, o);
    if (T) *T = hypot(o[0] - c->xa, o[1] - c->za) / c->L->c1 + hypot(o[2] - o[0], o[3] - o[1]) / c->L->c2 + hypot(o[4] - o[2], o[5] - o[3]) / c->L->c2
              + hypot(o[6] - o[4], o[7] - o[5]) / c->L->c1;
    return o[6] - c->xe;
}
static double iqi(double xa, double fa, double xb, double fb, double xc, double fc)
{
    const double db = xb - xa, dc = xc - xa;
    const double q = 1.0 / ((fb - fa) * (fb - fc) * (fc - fa));
    return xa + fa * q * (db * fc * (fc - fa) - dc * fb * (fb - fa));
}
/* reference: bisection to the last bit */
static int bisect(Ctx *c, double a, double fa, double b, double fb, double *root, double *T)
{
    if (fb == 0.0) { *root = b; F(c, b, T); return 1; }
    double fm = fa;
    for (int it = 0; it < 200 && b - a > 1e-16; ++it) {
        double m = 0.5 * (a + b);
        if (m <= a || m >= b) break;
        fm = F(c, m, NULL);
        if (!isfinite(fm)) return 0;
        if ((fm < 0) == (fa < 0)) { a = m; fa = fm; } else { b = m; fb = fm; }
    }
    *root = fabs(fa) < fabs(fb) ? a : b;
    fm = F(c, *root, T);
    return fabs(fm) < 1e-9;
}
static double DSCALE = 4.0, DMIN = 3e-6, DMAX = 1e-4, DACC = 1e-8, D2 = 3e-9, UMAX = 2.0;   /* the kernel's values */
/* the rule the kernel uses (rtus_solve_kernel<., ., ., true>): every round evaluates cand - d, cand, cand + d; the root of the inverse
 * quadratic through the three is ACCEPTED when d <= DACC (x_land has kinks where the return ray moves to the next polyline chord:
 * a wide triple that straddles one is off by d x the slope's jump), otherwise it is the next candidate with d = D2 */
static int three(Ctx *c, const double *alpha, const double *land, int br, int n, double *root, double *Tr, int *rounds)
{
    double xlo = alpha[br], xhi = alpha[br + 1], flo = land[br] - c->xe, fhi = land[br + 1] - c->xe;
    const int left = fabs(flo) < fabs(fhi);
    const int t1 = left ? br - 1 : br + 2, t2 = left ? br + 2 : br - 1;
    double xt = NAN, ft = NAN;
    if (t1 >= 0 && t1 < n && isfinite(land[t1])) { xt = alpha[t1]; ft = land[t1] - c->xe; }
    else if (t2 >= 0 && t2 < n && isfinite(land[t2])) { xt = alpha[t2]; ft = land[t2] - c->xe; }
    const int single = fhi == 0.0 || (fabs(flo) <= 1e-13 && fabs(fhi) <= 1e-13) || fabs(fhi) <= 1e-15 || fabs(flo) <= 1e-15;
    double cand = iqi(xlo, flo, xhi, fhi, xt, ft);
    const double sec = xlo - flo * (xhi - xlo) / (fhi - flo);
    if (!(cand > xlo && cand < xhi)) cand = sec;
    if (!(cand > xlo && cand < xhi)) cand = 0.5 * (xlo + xhi);
    double d = fmin(fmax(DSCALE * 0.02 * fabs(cand - sec), DMIN), DMAX);
    if (single) { cand = (fhi == 0.0 || fabs(fhi) < fabs(flo)) ? xhi : xlo; double T; double f = F(c, cand, &T); *rounds = 1; *root = cand; *Tr = T; return isfinite(f) && fabs(f) < 1e-9; }
    double fprev = INFINITY;
    for (int rd = 1; rd <= 40; ++rd) {
        d = fmin(d, 0.999 * fmin(cand - xlo, xhi - cand));
        double T0, Tm, Tq;
        const double f0 = F(c, cand, &T0), fm = F(c, cand - d, &Tm), fq = F(c, cand + d, &Tq);
        *rounds = rd;
        if (getenv("P3_TRACE") && c->evals > 0 && rd >= 1 && c->r == atof(getenv("P3_R")) && fabs(c->off - atof(getenv("P3_OFF"))) < 1e-9 && br == atoi(getenv("P3_Q"))) printf("  rd %d cand %.12f d %.3g fm %.3e f0 %.3e fq %.3e bracket [%.12f, %.12f]\n", rd, cand, d, fm, f0, fq, xlo, xhi);
        if (!isfinite(f0)) return 0;                                                /* the branch ends inside the bracket */
        if (f0 == 0.0) { *root = cand; *Tr = T0; return 1; }
        const int fin3 = isfinite(fm) && isfinite(fq);
        const int mono = fin3 && (fq - f0) * (f0 - fm) > 0.0;
        double x3 = NAN;
        if (mono && d > 0) {
            x3 = iqi(cand, f0, cand - d, fm, cand + d, fq);
            const double u = (x3 - cand) / d;
            if (d <= DACC && fabs(u) <= UMAX && fabs(fq - fm) <= 2e3 * d) {          /* ... and continuous across the triple: a jump is no root */ *root = x3; *Tr = T0 + 0.5 * u * ((Tq - Tm) + u * ((Tq - T0) - (T0 - Tm))); return 1; }
        }
        if (d > 0 && d <= DACC && fin3 && (fm < 0) != (fq < 0) && fabs(fq - fm) > 1e-4) return 0;     /* a jump across 2e-8 rad: no root */
        const double xs[3] = {cand - d, cand, cand + d}, fs[3] = {fm, f0, fq};
        for (int i = 0; i < 3; ++i) { if (!isfinite(fs[i])) continue; if ((fs[i] < 0) == (flo < 0)) { if (xs[i] > xlo) { xlo = xs[i]; flo = fs[i]; } } else if (xs[i] < xhi) { xhi = xs[i]; fhi = fs[i]; } }
        if (xhi - xlo <= 1e-13) { *root = cand; *Tr = T0; return fabs(f0) < 1e-9; }  /* a jump, or the root itself */
        double nc = x3;
        if (!(nc > xlo && nc < xhi)) nc = xlo - flo * (xhi - xlo) / (fhi - flo);
        if (!(nc > xlo && nc < xhi) || fabs(f0) > 0.5 * fprev) { nc = 0.5 * (xlo + xhi); d = 0.25 * (xhi - xlo); }
        else d = fmax(fmin(D2, 0.25 * fabs(nc - cand)), 1e-11);
        fprev = fabs(f0);
        cand = nc;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1) DSCALE = atof(argv[1]);
    if (argc > 2) DMIN = atof(argv[2]);
    if (argc > 3) DMAX = atof(argv[3]);
    if (argc > 4) UMAX = atof(argv[4]);
    if (argc > 5) D2 = atof(argv[5]);
    orc_lens L = {6400.0, 1483.0, 0.12156646438729327, 0.08843353561270673, 0};
    L.d = L.l0 + L.h0;
    const int n = 905, ne = 65;
    const double amax = 50.62033040986099 * (M_PI / 180.0);
    double *alpha = malloc(sizeof(double) * n), *xc = malloc(sizeof(double) * n), *zc = malloc(sizeof(double) * n), *land = malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) alpha[i] = -amax + (2.0 * amax) * i / (n - 1);
    for (int i = 0; i < n; ++i) orc_lens_point(&L, alpha[i], &xc[i], &zc[i]);
    double xel[65];
    { double mean = 0; for (int i = 0; i < 64; ++i) mean += i * 0.0006; mean /= 64;
      for (int i = 0, j = 0; i < 65; ++i) xel[i] = (i == 32) ? 0.0 : (j++) * 0.0006 - mean; }
    long nbr = 0, hist[32] = {0}, agree = 0, disagree = 0, nroot = 0;
    double worst_a = 0, worst_t = 0; long nbig = 0;
    const double txs[3] = {0.0, -0.0123, 0.0081};
    for (int t = 0; t < 3; ++t)
    for (int ri = 1; ri <= 10; ++ri) for (int oi = -10; oi <= 10; ++oi) {
        const double r = ri * 1e-2, off = oi * 1e-3;
        for (int i = 0; i < n; ++i) { double o[8]; orc_trace_alpha(&L, r, off, txs[t], L.d, L.d, alpha[i], xc, zc, n, 0, o); land[i] = o[6]; }
        for (int e = 0; e < ne; ++e) {
            Ctx c = {&L, r, off, txs[t], L.d, L.d, xc, zc, n, xel[e], 0};
            int cnt = 0;
            for (int q = 0; q + 1 < n && cnt < 4; ++q) {
                double f0 = land[q] - xel[e], f1 = land[q + 1] - xel[e];
                if (!(isfinite(f0) && isfinite(f1) && ((f0 < 0 && f1 >= 0) || (f0 > 0 && f1 <= 0)))) continue;
                ++cnt; ++nbr;
                double rb, Tb, r3, T3; int rounds = 0;
                const int okb = bisect(&c, alpha[q], f0, alpha[q + 1], f1, &rb, &Tb);
                const int ok3 = three(&c, alpha, land, q, n, &r3, &T3, &rounds);
                hist[rounds < 31 ? rounds : 31]++;
                if (rounds >= 4 && getenv("P3_LIST")) printf("SLOW r=%.2f off=%.3f tx=%g e=%d q=%d rounds %d f0=%g f1=%g\n", r, off, txs[t], e, q, rounds, f0, f1);
                if (okb == ok3) { ++agree; if (okb) { ++nroot; const int degenerate = oi == 0 && txs[t] == 0.0;   /* every ray retraces itself: a continuum of roots */
                    if (!degenerate) { if (fabs(rb - r3) > worst_a) worst_a = fabs(rb - r3); if (fabs(Tb - T3) > worst_t) worst_t = fabs(Tb - T3); if (fabs(rb - r3) > 1e-11 || fabs(Tb - T3) > 1e-13) { ++nbig; if (nbig < 8) printf("BIG r=%.2f off=%.3f tx=%g e=%d q=%d rounds %d dalpha %.3g dT %.3g\n", r, off, txs[t], e, q, rounds, rb - r3, Tb - T3); } } } }
                else { ++disagree; if (disagree < 12) printf("DISAGREE r=%.2f off=%.3f tx=%g e=%d q=%d bisect=%d three=%d (%d rounds) f0=%g f1=%g\n", r, off, txs[t], e, q, okb, ok3, rounds, f0, f1); }
            }
        }
    }
    printf("d1 = clamp(%g x 0.02 |iqi - secant|, %g, %g), accept at d <= 2e-8: brackets %ld, roots %ld, agree %ld disagree %ld; worst |dalpha| %.3g rad, worst |dT| %.3g s\n", DSCALE, DMIN, DMAX, nbr, nroot, agree, disagree, worst_a, worst_t);
    printf("beyond 1e-11 rad or 1e-13 s (degenerate continuum excluded): %ld\n", nbig);
    printf("rounds:"); for (int i = 0; i < 32; ++i) if (hist[i]) printf("  %d: %ld", i, hist[i]); printf("\n");
    return 0;
}
