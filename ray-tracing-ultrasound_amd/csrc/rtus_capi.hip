// rtus_capi.hip — the extern "C" boundary declared in include/rtus.h.
// *_dev: device pointers + stream, asynchronous, no allocation.  Host twins: stage through HBM.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <vector>
#include "../../include/rtus.h"

// launchers (rtus_shoot.hip / rtus_match.hip / rtus_fermat.hip)
size_t rtus_ws_bytes(int n);
hipError_t rtus_launch_shoot(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, const double* z_f, int n,
                             double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                             void* ws, unsigned flags, hipStream_t s);
size_t rtus_solve_ws_bytes(int n, int n_geom, int n_tx);
hipError_t rtus_launch_solve(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, int n, const double* x_rx, int n_rx,
                             double z_land, double* tt, double* alpha_root, double* tt_all,
                             double* alpha_all, uint8_t* n_roots, void* ws, unsigned flags, hipStream_t s);
hipError_t rtus_launch_match(const double* land_x, const double* tof, int n_batch, int n, const double* x_rx,
                             int n_rx, double atol, double rtol, int32_t* first_ray,
                             uint8_t* hit, double* tof_hit, uint8_t* ray_hit, hipStream_t s);
hipError_t rtus_launch_tt_layers(const double* z_if, const double* c, int n_if, const double* xe,
                                 const double* ze, int n_e, const double* xf, const double* zf, int n_f,
                                 double* tt, uint8_t* iters, hipStream_t s);
hipError_t rtus_launch_tt_layers_batch(const double* z_if, const double* c, int n_if, const double* xe, const double* ze,
                                       int n_e, long long e_stride, const double* xf, const double* zf, int n_f,
                                       long long f_stride, double* tt, long long t_stride, int n_batch, hipStream_t s);

hipError_t rtus_launch_tt_lens_f64(const rtus_lens& L, double a_lo, double a_hi, const double* xe, const double* ze,
                                   int n_e, const double* xf, const double* zf, int n_f, double* tt,
                                   double* alpha_out, hipStream_t s);
hipError_t rtus_launch_tt_lens_f32(const rtus_lens& L, double a_lo, double a_hi, const float* xe, const float* ze,
                                   int n_e, const float* xf, const float* zf, int n_f, float* tt, float* alpha_out,
                                   hipStream_t s);

static thread_local int g_last_hip = 0;
static int hip_fail(hipError_t e) { g_last_hip = (int)e; return RTUS_ERR_HIP; }
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_); } while (0)

// RAII device buffer for the host-staging twins.
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
    hipError_t upload(const void* src, size_t bytes)
    {
        hipError_t e = alloc(bytes);
        return e != hipSuccess ? e : hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
    }
    template <class T> T* as() { return (T*)p; }
};

// The host twins run on `device` and put the caller's current device back when they return.
struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
static int select_device_impl(int device, DeviceGuard& g)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RTUS_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return RTUS_ERR_NO_DEVICE;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != device) g.prev = cur;
    HIP_TRY(hipSetDevice(device));
    return RTUS_OK;
}
#define select_device(dev) select_device_impl((dev), device_guard_)

// curved-lens helpers (C++ linkage: templates)
static int check_lens(const rtus_lens* lens, double a_lo, double a_hi, const void* xe, const void* ze, int n_e,
                      const void* xf, const void* zf, int n_f, const void* tt)
{
    if (!lens || !xe || !ze || !xf || !zf || !tt || n_e <= 0 || n_f <= 0) return RTUS_ERR_INVALID_ARG;
    if (n_e > 65535 * 64) return RTUS_ERR_UNSUPPORTED;
    if (!(a_hi > a_lo) || !isfinite(a_lo) || !isfinite(a_hi)) return RTUS_ERR_INVALID_ARG;
    if (!(lens->c1 > 0) || !(lens->c2 > 0) || lens->c1 == lens->c2) return RTUS_ERR_INVALID_ARG;
    return RTUS_OK;
}

template <typename R, typename F>
static int lens_host(const rtus_lens* lens, double a_lo, double a_hi, const R* xe, const R* ze, int n_e, const R* xf,
                     const R* zf, int n_f, R* tt, R* alpha_out, int device, F launch)
{
    int st = check_lens(lens, a_lo, a_hi, xe, ze, n_e, xf, zf, n_f, tt);
    if (st) return st;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t tot = (size_t)n_e * n_f;
    DevBuf dxe, dze, dxf, dzf, dtt, dal;
    HIP_TRY(dxe.upload(xe, sizeof(R) * n_e));
    HIP_TRY(dze.upload(ze, sizeof(R) * n_e));
    HIP_TRY(dxf.upload(xf, sizeof(R) * n_f));
    HIP_TRY(dzf.upload(zf, sizeof(R) * n_f));
    HIP_TRY(dtt.alloc(sizeof(R) * tot));
    if (alpha_out) HIP_TRY(dal.alloc(sizeof(R) * tot));
    HIP_TRY(launch(*lens, a_lo, a_hi, dxe.as<R>(), dze.as<R>(), n_e, dxf.as<R>(), dzf.as<R>(), n_f, dtt.as<R>(),
                   dal.as<R>(), (hipStream_t)0));
    HIP_TRY(hipStreamSynchronize(0));
    HIP_TRY(hipMemcpy(tt, dtt.p, sizeof(R) * tot, hipMemcpyDeviceToHost));
    if (alpha_out) HIP_TRY(hipMemcpy(alpha_out, dal.p, sizeof(R) * tot, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

extern "C" {

const char* rtus_strerror(int status)
{
    switch (status) {
        case RTUS_OK: return "ok";
        case RTUS_ERR_INVALID_ARG: return "invalid argument";
        case RTUS_ERR_NO_DEVICE: return "no usable HIP device";
        case RTUS_ERR_HIP: return "HIP runtime error (see rtus_last_hip_error)";
        case RTUS_ERR_WORKSPACE: return "workspace null or too small";
        case RTUS_ERR_UNSUPPORTED: return "unsupported size";
        default: return "unknown status";
    }
}
int rtus_version(void) { return RTUS_VERSION; }
int rtus_last_hip_error(void) { return g_last_hip; }
int rtus_device_count(int* count)
{
    if (!count) return RTUS_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return RTUS_ERR_NO_DEVICE; }
    *count = n;
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- forward trace
size_t rtus_shoot_workspace_bytes(int n_rays) { return n_rays > 0 ? rtus_ws_bytes(n_rays) : 0; }

static int check_shoot(const rtus_lens* lens, const void* geoms, int n_geom, const void* x_a, const void* z_a,
                       int n_tx, const void* alpha, const void* z_f, int n_rays)
{
    if (!lens || !geoms || !x_a || !z_a || !alpha || !z_f) return RTUS_ERR_INVALID_ARG;
    if (n_geom <= 0 || n_tx <= 0 || n_rays < 2) return RTUS_ERR_INVALID_ARG;   // curve needs >= 2 points (main_rt.py:26-27)
    if (n_geom > 65535 || n_tx > 65535) return RTUS_ERR_UNSUPPORTED;
    if (!(lens->c1 > 0) || !(lens->c2 > 0) || lens->c1 == lens->c2) return RTUS_ERR_INVALID_ARG;
    return RTUS_OK;
}
#define RTUS_SHOOT_KNOWN_FLAGS (RTUS_SHOOT_FAST_MATH | RTUS_TRUE_PIPE_TANGENT | RTUS_ANALYTIC_LENS)

int rtus_shoot_dev(const rtus_lens* lens, const double* d_geoms, int n_geom, const double* d_x_a,
                   const double* d_z_a, int n_tx, const double* d_alpha, const double* d_z_f, int n_rays,
                   double* d_out8, double* d_tof4, double* d_tof, double* d_land_x, uint8_t* d_status,
                   void* d_workspace, size_t workspace_bytes, unsigned flags, void* stream)
{
    int st = check_shoot(lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays);
    if (st) return st;
    if (flags & ~RTUS_SHOOT_KNOWN_FLAGS) return RTUS_ERR_INVALID_ARG;
    if (!d_workspace || workspace_bytes < rtus_ws_bytes(n_rays)) return RTUS_ERR_WORKSPACE;
    HIP_TRY(rtus_launch_shoot(*lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays, d_out8,
                              d_tof4, d_tof, d_land_x, d_status, d_workspace, flags, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_shoot(const rtus_lens* lens, const double* geoms, int n_geom, const double* x_a, const double* z_a,
               int n_tx, const double* alpha, const double* z_f, int n_rays, double* out8, double* tof4,
               double* tof, double* land_x, uint8_t* status, unsigned flags, int device)
{
    int st = check_shoot(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, z_f, n_rays);
    if (st) return st;
    if (flags & ~RTUS_SHOOT_KNOWN_FLAGS) return RTUS_ERR_INVALID_ARG;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t rows = (size_t)n_geom * n_tx, n = (size_t)n_rays;
    DevBuf g, xa, za, al, zf, ws, o8, t4, tt, lx, sb;
    HIP_TRY(g.upload(geoms, sizeof(double) * 2 * n_geom));
    HIP_TRY(xa.upload(x_a, sizeof(double) * n_tx));
    HIP_TRY(za.upload(z_a, sizeof(double) * n_tx));
    HIP_TRY(al.upload(alpha, sizeof(double) * n));
    HIP_TRY(zf.upload(z_f, sizeof(double) * n));
    HIP_TRY(ws.alloc(rtus_ws_bytes(n_rays)));
    if (out8) HIP_TRY(o8.alloc(sizeof(double) * rows * 8 * n));
    if (tof4) HIP_TRY(t4.alloc(sizeof(double) * rows * 4 * n));
    if (tof) HIP_TRY(tt.alloc(sizeof(double) * rows * n));
    if (land_x) HIP_TRY(lx.alloc(sizeof(double) * rows * n));
    if (status) HIP_TRY(sb.alloc(rows * n));
    HIP_TRY(rtus_launch_shoot(*lens, g.as<double>(), n_geom, xa.as<double>(), za.as<double>(), n_tx,
                              al.as<double>(), zf.as<double>(), n_rays, o8.as<double>(), t4.as<double>(),
                              tt.as<double>(), lx.as<double>(), sb.as<uint8_t>(), ws.p, flags, 0));
    HIP_TRY(hipStreamSynchronize(0));
    if (out8) HIP_TRY(hipMemcpy(out8, o8.p, sizeof(double) * rows * 8 * n, hipMemcpyDeviceToHost));
    if (tof4) HIP_TRY(hipMemcpy(tof4, t4.p, sizeof(double) * rows * 4 * n, hipMemcpyDeviceToHost));
    if (tof) HIP_TRY(hipMemcpy(tof, tt.p, sizeof(double) * rows * n, hipMemcpyDeviceToHost));
    if (land_x) HIP_TRY(hipMemcpy(land_x, lx.p, sizeof(double) * rows * n, hipMemcpyDeviceToHost));
    if (status) HIP_TRY(hipMemcpy(status, sb.p, rows * n, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- root-finding solve
size_t rtus_solve_workspace_bytes(int n_rays, int n_geom, int n_tx)
{
    return (n_rays > 0 && n_geom > 0 && n_tx > 0) ? rtus_solve_ws_bytes(n_rays, n_geom, n_tx) : 0;
}

static int check_solve(const rtus_lens* lens, const void* geoms, int n_geom, const void* x_a, const void* z_a, int n_tx,
                       const void* alpha, int n_rays, const void* x_rx, int n_rx, double z_land, const void* tt,
                       unsigned flags)
{
    int st = check_shoot(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, alpha, n_rays);
    if (st) return st;
    if (!x_rx || !tt || n_rx <= 0 || !isfinite(z_land)) return RTUS_ERR_INVALID_ARG;
    if (flags & ~RTUS_SHOOT_KNOWN_FLAGS) return RTUS_ERR_INVALID_ARG;
    if ((long long)n_geom * n_tx > 0x7fffffffLL / (n_rx > 0 ? n_rx : 1)) return RTUS_ERR_UNSUPPORTED;
    if ((long long)n_geom * n_tx * ((n_rays + 63) / 64) > 0x7fffffffLL) return RTUS_ERR_UNSUPPORTED;
    return RTUS_OK;
}

int rtus_solve_dev(const rtus_lens* lens, const double* d_geoms, int n_geom, const double* d_x_a, const double* d_z_a,
                   int n_tx, const double* d_alpha, int n_rays, const double* d_x_rx, int n_rx, double z_land,
                   double* d_tt, double* d_alpha_root, double* d_tt_all, double* d_alpha_all, uint8_t* d_n_roots,
                   void* d_workspace, size_t workspace_bytes, unsigned flags, void* stream)
{
    int st = check_solve(lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, n_rays, d_x_rx, n_rx, z_land, d_tt, flags);
    if (st) return st;
    if (!d_workspace || workspace_bytes < rtus_solve_ws_bytes(n_rays, n_geom, n_tx)) return RTUS_ERR_WORKSPACE;
    HIP_TRY(rtus_launch_solve(*lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, n_rays, d_x_rx, n_rx, z_land, d_tt,
                              d_alpha_root, d_tt_all, d_alpha_all, d_n_roots, d_workspace, flags, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_solve(const rtus_lens* lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
               const double* alpha, int n_rays, const double* x_rx, int n_rx, double z_land, double* tt,
               double* alpha_root, double* tt_all, double* alpha_all, uint8_t* n_roots, unsigned flags, int device)
{
    int st = check_solve(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, n_rays, x_rx, n_rx, z_land, tt, flags);
    if (st) return st;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t tot = (size_t)n_geom * n_tx * n_rx;
    DevBuf g, xa, za, al, rx, ws, dt, da, dta, daa, dn;
    HIP_TRY(g.upload(geoms, sizeof(double) * 2 * n_geom));
    HIP_TRY(xa.upload(x_a, sizeof(double) * n_tx));
    HIP_TRY(za.upload(z_a, sizeof(double) * n_tx));
    HIP_TRY(al.upload(alpha, sizeof(double) * n_rays));
    HIP_TRY(rx.upload(x_rx, sizeof(double) * n_rx));
    HIP_TRY(ws.alloc(rtus_solve_ws_bytes(n_rays, n_geom, n_tx)));
    HIP_TRY(dt.alloc(sizeof(double) * tot));
    if (alpha_root) HIP_TRY(da.alloc(sizeof(double) * tot));
    if (tt_all) HIP_TRY(dta.alloc(sizeof(double) * tot * RTUS_MAX_ROOTS));
    if (alpha_all) HIP_TRY(daa.alloc(sizeof(double) * tot * RTUS_MAX_ROOTS));
    if (n_roots) HIP_TRY(dn.alloc(tot));
    HIP_TRY(rtus_launch_solve(*lens, g.as<double>(), n_geom, xa.as<double>(), za.as<double>(), n_tx, al.as<double>(),
                              n_rays, rx.as<double>(), n_rx, z_land, dt.as<double>(), da.as<double>(), dta.as<double>(),
                              daa.as<double>(), dn.as<uint8_t>(), ws.p, flags, 0));
    HIP_TRY(hipStreamSynchronize(0));
    HIP_TRY(hipMemcpy(tt, dt.p, sizeof(double) * tot, hipMemcpyDeviceToHost));
    if (alpha_root) HIP_TRY(hipMemcpy(alpha_root, da.p, sizeof(double) * tot, hipMemcpyDeviceToHost));
    if (tt_all) HIP_TRY(hipMemcpy(tt_all, dta.p, sizeof(double) * tot * RTUS_MAX_ROOTS, hipMemcpyDeviceToHost));
    if (alpha_all) HIP_TRY(hipMemcpy(alpha_all, daa.p, sizeof(double) * tot * RTUS_MAX_ROOTS, hipMemcpyDeviceToHost));
    if (n_roots) HIP_TRY(hipMemcpy(n_roots, dn.p, tot, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- element matcher
static int check_match(const void* land_x, int n_batch, int n_rays, const void* x_rx, int n_rx, double atol,
                       double rtol)
{
    if (!land_x || !x_rx || n_batch <= 0 || n_rays <= 0 || n_rx <= 0) return RTUS_ERR_INVALID_ARG;
    if (!(atol >= 0) || !(rtol >= 0)) return RTUS_ERR_INVALID_ARG;
    if (n_rx > 4000) return RTUS_ERR_UNSUPPORTED;   // x_rx + tolerances live in < 64 KiB of LDS
    return RTUS_OK;
}

int rtus_match_dev(const double* d_land_x, const double* d_tof, int n_batch, int n_rays, const double* d_x_rx,
                   int n_rx, double atol, double rtol, int32_t* d_first_ray, uint8_t* d_hit, double* d_tof_hit,
                   void* stream)
{
    int st = check_match(d_land_x, n_batch, n_rays, d_x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (!d_first_ray) return RTUS_ERR_INVALID_ARG;
    HIP_TRY(rtus_launch_match(d_land_x, d_tof, n_batch, n_rays, d_x_rx, n_rx, atol, rtol, d_first_ray,
                              d_hit, d_tof_hit, nullptr, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_ray_hits_dev(const double* d_land_x, int n_batch, int n_rays, const double* d_x_rx, int n_rx,
                      double atol, double rtol, uint8_t* d_ray_hit, void* stream)
{
    int st = check_match(d_land_x, n_batch, n_rays, d_x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (!d_ray_hit) return RTUS_ERR_INVALID_ARG;
    HIP_TRY(rtus_launch_match(d_land_x, nullptr, n_batch, n_rays, d_x_rx, n_rx, atol, rtol, nullptr,
                              nullptr, nullptr, d_ray_hit, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_match(const double* land_x, const double* tof, int n_batch, int n_rays, const double* x_rx, int n_rx,
               double atol, double rtol, int32_t* first_ray, uint8_t* hit, double* tof_hit, int device)
{
    int st = check_match(land_x, n_batch, n_rays, x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (tof_hit && !tof) return RTUS_ERR_INVALID_ARG;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t rn = (size_t)n_batch * n_rays, re = (size_t)n_batch * n_rx;
    DevBuf lx, tf, rx, fr, hb, th;
    HIP_TRY(lx.upload(land_x, sizeof(double) * rn));
    if (tof) HIP_TRY(tf.upload(tof, sizeof(double) * rn));
    HIP_TRY(rx.upload(x_rx, sizeof(double) * n_rx));
    HIP_TRY(fr.alloc(sizeof(int32_t) * re));
    if (hit) HIP_TRY(hb.alloc(re));
    if (tof_hit) HIP_TRY(th.alloc(sizeof(double) * re));
    HIP_TRY(rtus_launch_match(lx.as<double>(), tf.as<double>(), n_batch, n_rays, rx.as<double>(), n_rx, atol,
                              rtol, fr.as<int32_t>(), hb.as<uint8_t>(), th.as<double>(), nullptr, 0));
    HIP_TRY(hipStreamSynchronize(0));
    if (first_ray) HIP_TRY(hipMemcpy(first_ray, fr.p, sizeof(int32_t) * re, hipMemcpyDeviceToHost));
    if (hit) HIP_TRY(hipMemcpy(hit, hb.p, re, hipMemcpyDeviceToHost));
    if (tof_hit) HIP_TRY(hipMemcpy(tof_hit, th.p, sizeof(double) * re, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

int rtus_ray_hits(const double* land_x, int n_batch, int n_rays, const double* x_rx, int n_rx, double atol,
                  double rtol, uint8_t* ray_hit, int device)
{
    int st = check_match(land_x, n_batch, n_rays, x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (!ray_hit) return RTUS_ERR_INVALID_ARG;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t rn = (size_t)n_batch * n_rays;
    DevBuf lx, rx, rh;
    HIP_TRY(lx.upload(land_x, sizeof(double) * rn));
    HIP_TRY(rx.upload(x_rx, sizeof(double) * n_rx));
    HIP_TRY(rh.alloc(rn));
    HIP_TRY(rtus_launch_match(lx.as<double>(), nullptr, n_batch, n_rays, rx.as<double>(), n_rx, atol, rtol,
                              nullptr, nullptr, nullptr, rh.as<uint8_t>(), 0));
    HIP_TRY(hipStreamSynchronize(0));
    HIP_TRY(hipMemcpy(ray_hit, rh.p, rn, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- planar layers
static int check_layers(const double* z_if, const double* c, int n_if, const void* xe, const void* ze, int n_e,
                        const void* xf, const void* zf, int n_f, const void* tt)
{
    if (!c || (n_if > 0 && !z_if) || !xe || !ze || !xf || !zf || !tt) return RTUS_ERR_INVALID_ARG;
    if (n_if < 0 || n_e <= 0 || n_f <= 0) return RTUS_ERR_INVALID_ARG;
    if (n_if > RTUS_MAX_LAYERS) return RTUS_ERR_UNSUPPORTED;
    if (n_e > 65535 * 64) return RTUS_ERR_UNSUPPORTED;
    for (int i = 0; i <= n_if; ++i) if (!(c[i] > 0) || !isfinite(c[i])) return RTUS_ERR_INVALID_ARG;
    for (int i = 0; i < n_if; ++i) {
        if (!isfinite(z_if[i])) return RTUS_ERR_INVALID_ARG;
        if (i && !(z_if[i] > z_if[i - 1])) return RTUS_ERR_INVALID_ARG;
    }
    return RTUS_OK;
}

int rtus_tt_layers_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                       int n_e, const double* d_xf, const double* d_zf, int n_f, double* d_tt, uint8_t* d_iters,
                       void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    HIP_TRY(rtus_launch_tt_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_iters,
                                  (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_layers_batch_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                             int n_e, long long e_stride, const double* d_xf, const double* d_zf, int n_f,
                             long long f_stride, double* d_tt, long long t_stride, int n_batch, void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    if (n_batch <= 0 || e_stride < 0 || f_stride < 0) return RTUS_ERR_INVALID_ARG;
    if (n_batch > 1 && t_stride < (long long)n_e * n_f) return RTUS_ERR_INVALID_ARG;   // outputs of two problems would overlap
    if (n_batch > 65535) return RTUS_ERR_UNSUPPORTED;                                   // grid.z
    HIP_TRY(rtus_launch_tt_layers_batch(z_if, c, n_if, d_xe, d_ze, n_e, e_stride, d_xf, d_zf, n_f, f_stride, d_tt,
                                        t_stride, n_batch, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_layers(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                   const double* xf, const double* zf, int n_f, double* tt, uint8_t* iters, int device)
{
    int st = check_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt);
    if (st) return st;
    DeviceGuard device_guard_;
    if ((st = select_device(device))) return st;
    const size_t tot = (size_t)n_e * n_f;
    DevBuf dxe, dze, dxf, dzf, dtt, dit;
    HIP_TRY(dxe.upload(xe, sizeof(double) * n_e));
    HIP_TRY(dze.upload(ze, sizeof(double) * n_e));
    HIP_TRY(dxf.upload(xf, sizeof(double) * n_f));
    HIP_TRY(dzf.upload(zf, sizeof(double) * n_f));
    HIP_TRY(dtt.alloc(sizeof(double) * tot));
    if (iters) HIP_TRY(dit.alloc(tot));
    HIP_TRY(rtus_launch_tt_layers(z_if, c, n_if, dxe.as<double>(), dze.as<double>(), n_e, dxf.as<double>(),
                                  dzf.as<double>(), n_f, dtt.as<double>(), dit.as<uint8_t>(), 0));
    HIP_TRY(hipStreamSynchronize(0));
    HIP_TRY(hipMemcpy(tt, dtt.p, sizeof(double) * tot, hipMemcpyDeviceToHost));
    if (iters) HIP_TRY(hipMemcpy(iters, dit.p, tot, hipMemcpyDeviceToHost));
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- curved lens
int rtus_tt_lens_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const double* d_xe, const double* d_ze,
                     int n_e, const double* d_xf, const double* d_zf, int n_f, double* d_tt, double* d_alpha_out,
                     void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    HIP_TRY(rtus_launch_tt_lens_f64(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_alpha_out,
                                    (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_lens_f32_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* d_xe, const float* d_ze,
                         int n_e, const float* d_xf, const float* d_zf, int n_f, float* d_tt, float* d_alpha_out,
                         void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    HIP_TRY(rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_alpha_out,
                                    (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_lens(const rtus_lens* lens, double alpha_lo, double alpha_hi, const double* xe, const double* ze, int n_e,
                 const double* xf, const double* zf, int n_f, double* tt, double* alpha_out, int device)
{
    return lens_host<double>(lens, alpha_lo, alpha_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, device,
                             rtus_launch_tt_lens_f64);
}

int rtus_tt_lens_f32(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* xe, const float* ze,
                     int n_e, const float* xf, const float* zf, int n_f, float* tt, float* alpha_out, int device)
{
    return lens_host<float>(lens, alpha_lo, alpha_hi, xe, ze, n_e, xf, zf, n_f, tt, alpha_out, device,
                            rtus_launch_tt_lens_f32);
}

}   // extern "C"
