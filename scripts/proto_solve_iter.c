/* Scratch prototype (CPU, not product): how many trace evaluations do candidate bracketed iterations need on the
 * reference sweep's brackets, and do they reach the same decision as the oracle's bisection?
 *   gcc -O2 -ffp-contract=off -fopenmp scripts/proto_solve_iter.c -o /tmp/proto_solve_iter -lm && /tmp/proto_solve_iter
 * The numbers this printed are quoted in DESIGN.md (root-find section). */
#include "../oracle/rt_oracle.c"
#include <stdio.h>
#include <stdlib.h>

typedef struct { const orc_lens *L; double r, off, xa, za, zl; const double *xc, *zc; int n; double xe; long evals; } Ctx;
static double F(Ctx *c, double a, double *o8)
{
    c->evals++;
    orc_trace_alpha(c->L, c->r, c->off, c->xa, c->za, c->zl, a, c->xc, c->zc, c->n, 0, o8);
    return o8[6] - c->xe;
}

/* current GPU iteration: Illinois */
static int illinois(Ctx *c, double al, double fl, double ah, double fh, double *root, int *its)
{
    double ac = ah, fc = fh, o[8];
    int dead = 0, it;
    if (fh == 0.0) { *root = ah; *its = 0; return 1; }
    for (it = 0; it < 64; ++it) {
        if (!(fabs(fc) > 1e-13 && fabs(ah - al) > 1e-13) && it > 0) break;
        double cand = (al * fh - ah * fl) / (fh - fl);
        if (!(cand > fmin(al, ah) && cand < fmax(al, ah))) cand = 0.5 * (al + ah);
        ac = cand;
        fc = F(c, ac, o);
        if (!isfinite(fc)) { dead = 1; break; }
        if ((fc < 0.0) == (fh < 0.0)) { ah = ac; fh = fc; fl *= 0.5; }
        else { al = ah; fl = fh; ah = ac; fh = fc; }
    }
    *its = it; *root = ac;
    return !dead && fabs(fc) < 1e-9;
}

/* candidate: secant through the two latest points, kept inside the bracket; bisection when the bracket has not halved in
 * two steps; early "no root" when both ends are further from zero than any continuous branch could bridge */
static int secant(Ctx *c, double a, double fa, double b, double fb, double *root, int *its, int *why)
{
    double x0 = a, f0 = fa, x1 = b, f1 = fb, o[8], xc = b, fc = fb;
    double w2 = b - a, w1 = b - a;
    int it;
    *why = 0;
    if (fb == 0.0) { *root = b; *its = 0; return 1; }
    for (it = 0; it < 64; ++it) {
        if (!(fabs(fc) > 1e-13 && (b - a) > 1e-13) && it > 0) break;
        double cand = x1 - f1 * (x1 - x0) / (f1 - f0);
        const int force = (b - a) > 0.5 * w2;                /* not halved over the last two steps */
        if (it > 1 && force) cand = 0.5 * (a + b);
        if (!(cand > a && cand < b)) cand = 0.5 * (a + b);
        w2 = w1; w1 = b - a;
        xc = cand;
        fc = F(c, xc, o);
        if (!isfinite(fc)) { *why = 1; *its = it + 1; *root = xc; return 0; }
        if ((fc < 0.0) == (fa < 0.0)) { a = xc; fa = fc; } else { b = xc; fb = fc; }
        x0 = x1; f0 = f1; x1 = xc; f1 = fc;
        /* a continuous branch steeper than 5e6 m/rad is not resolvable by the oracle's bisection either */
        if (fmin(fabs(fa), fabs(fb)) > 1e-9 + 5e6 * (b - a)) { *why = 2; ++it; break; }
    }
    *its = it; *root = xc;
    return fabs(fc) < 1e-9;
}

int main(void)
{
    orc_lens L = {6400.0, 1483.0, 0.12156646438729327, 0.08843353561270673, 0};
    L.d = L.l0 + L.h0;
    const int n = 905, ne = 65;
    const double amax = 50.62033040986099 * (M_PI / 180.0);
    double *alpha = malloc(sizeof(double) * n), *xc = malloc(sizeof(double) * n), *zc = malloc(sizeof(double) * n);
    double *land = malloc(sizeof(double) * n);
    for (int i = 0; i < n; ++i) alpha[i] = -amax + (2.0 * amax) * i / (n - 1);
    for (int i = 0; i < n; ++i) orc_lens_point(&L, alpha[i], &xc[i], &zc[i]);
    double xel[65];
    { double mean = 0; for (int i = 0; i < 64; ++i) mean += i * 0.0006; mean /= 64;
      for (int i = 0, j = 0; i < 65; ++i) xel[i] = (i == 32) ? 0.0 : (j++) * 0.0006 - mean; }
    long nbr = 0, hist_i[70] = {0}, hist_s[70] = {0}, agree = 0, disagree = 0, ev_i = 0, ev_s = 0, nroot = 0, njump = 0, ndead = 0;
    long why2 = 0, maxper_wave_i = 0;
    double worst = 0;
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
                double ri_, rs_; int ii, is, why;
                c.evals = 0; int oki = illinois(&c, alpha[q], f0, alpha[q + 1], f1, &ri_, &ii); ev_i += c.evals;
                c.evals = 0; int oks = secant(&c, alpha[q], f0, alpha[q + 1], f1, &rs_, &is, &why); ev_s += c.evals;
                hist_i[ii]++; hist_s[is]++;
                if (why == 2) ++why2;
                if (oki == oks) { ++agree; if (oki) { ++nroot; if (fabs(ri_ - rs_) > worst) worst = fabs(ri_ - rs_); } else { if (why == 1) ++ndead; else ++njump; } }
                else { ++disagree; printf("DISAGREE r=%.2f off=%.3f e=%d q=%d illinois=%d(%d its) secant=%d(%d its, why %d) f0=%g f1=%g\n", r, off, e, q, oki, ii, oks, is, why, f0, f1); }
            }
        }
    }
    printf("brackets %ld: roots %ld, jumps %ld, dead %ld; agree %ld disagree %ld; worst |dalpha| %.3g\n", nbr, nroot, njump, ndead, agree, disagree, worst);
    printf("evals: illinois %ld (%.2f/bracket), secant %ld (%.2f/bracket); early no-root exits %ld\n", ev_i, (double)ev_i / nbr, ev_s, (double)ev_s / nbr, why2);
    printf("its  illinois  secant\n");
    for (int i = 0; i < 70; ++i) if (hist_i[i] || hist_s[i]) printf("%3d %8ld %8ld\n", i, hist_i[i], hist_s[i]);
    (void)maxper_wave_i;
    return 0;
}
