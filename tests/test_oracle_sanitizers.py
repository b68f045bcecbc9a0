"""CPU: the C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (sanitizers run on the
CPU build; the GPU pool refuses device ASan).  A small C driver exercises the forward trace on regular, missing
and NaN-producing geometries, the matcher and the root-finding solve; any report fails the test."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

DRIVER = r"""
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>
typedef struct { double c1, c2, l0, h0, d; } orc_lens;
void orc_shoot(const orc_lens*, double, double, double, double, const double*, const double*, int, double*, uint8_t*);
void orc_tof4(const orc_lens*, double, double, const double*, int, double*);
void orc_match(const double*, const double*, int, const double*, int, double, double, uint8_t*, double*, int32_t*);
void orc_solve(const orc_lens*, double, double, double, double, double, const double*, int, const double*, int, unsigned,
               double*, double*, double*);
void orc_tt_layers(const double*, const double*, int, const double*, const double*, int, const double*, const double*, int, double*);
int main(void) {
    orc_lens L = {6400, 1483, 0.12156646438729327, 0.08843353561270673, 0.21};
    const int sizes[] = {2, 3, 9, 905};
    double x[65], tt[65], ta[65 * 4], aa[65 * 4], th[65];
    uint8_t hit[65]; int32_t fr[65];
    for (int i = 0; i < 65; i++) x[i] = (i - 32) * 0.0006;
    for (int s = 0; s < 4; s++) {
        int n = sizes[s];
        double *a = malloc(n * 8), *zf = malloc(n * 8), *o = malloc(8 * n * 8), *t4 = malloc(4 * n * 8);
        uint8_t *st = malloc(n);
        double am = 50.62033040986099 * M_PI / 180;
        for (int i = 0; i < n; i++) { a[i] = -am + 2 * am * i / (n - 1); zf[i] = 0.21; }
        orc_shoot(&L, 0.037, 0.0038, 0.0, 0.21, zf, a, n, o, st);      /* regular */
        orc_shoot(&L, 0.005, 0.05, 0.0, 0.21, zf, a, n, o, st);        /* rays miss the pipe */
        orc_shoot(&L, 0.01, -0.01, 0.0, 0.21, zf, a, n, o, st);        /* NaN tangents */
        orc_tof4(&L, 0.0, 0.21, o, n, t4);
        orc_match(o + 6 * n, t4, n, x, 65, 1e-6, 1e-5, hit, th, fr);
        orc_solve(&L, 0.037, 0.0038, 0.001, 0.21, 0.21, a, n, x, 65, 6u, tt, ta, aa);
        free(a); free(zf); free(o); free(t4); free(st);
    }
    double zi[2] = {0.01, 0.02}, c[3] = {2330, 1483, 5900}, xe[2] = {0, 0.001}, ze[2] = {0, 0.015}, xf[3] = {0.0, 0.01, 0.3},
           zz[3] = {0.005, 0.03, 0.012}, t[6];
    orc_tt_layers(zi, c, 2, xe, ze, 2, xf, zz, 3, t);
    printf("ok %g\n", t[1]);
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="needs gcc")
def test_oracle_clean_under_asan_ubsan(tmp_path):
    drv = tmp_path / "drv.c"
    drv.write_text(DRIVER)
    exe = tmp_path / "drv"
    cmd = ["gcc", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-ffp-contract=off", os.path.join(ROOT, "oracle", "rt_oracle.c"), str(drv), "-o", str(exe), "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok") and "ERROR" not in r.stderr and "runtime error" not in r.stderr
