// Host build of ray-tracing-ultrasound_amd/csrc/rtus_trig.h for tests/test_trig_kernels.py (the header is plain C++:
// the same text the gfx950 kernels compile).  g++ -O2 -mfma -ffp-contract=off -shared -fPIC
#include "../../ray-tracing-ultrasound_amd/csrc/rtus_trig.h"
extern "C" {
void t_sin(const double* x, int n, double* o) { for (int i = 0; i < n; ++i) o[i] = rtus_sin(x[i]); }
void t_cos(const double* x, int n, double* o) { for (int i = 0; i < n; ++i) o[i] = rtus_cos(x[i]); }
void t_tan(const double* x, int n, double* o) { for (int i = 0; i < n; ++i) o[i] = rtus_tan(x[i]); }
}
