// rtus_capi.hip — the extern "C" boundary declared in include/rtus.h.
// *_dev: device pointers + stream, asynchronous, no allocation.  Host twins: stage through HBM.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <algorithm>
#include <dlfcn.h>
#include <mutex>
#include <numeric>
#include <string.h>
#include <thread>
#include <vector>
#include "../../include/rtus.h"

// launchers (rtus_shoot.hip / rtus_match.hip / rtus_fermat.hip)
size_t rtus_ws_bytes(int n);
hipError_t rtus_selftest_run(const rtus_lens& lens, int n, long long n_math, unsigned long long counts[4], hipStream_t s);
hipError_t rtus_launch_shoot(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, const double* z_f, int n,
                             double* out8, double* tof4, double* tof, double* land_x, uint8_t* status,
                             void* ws, unsigned flags, hipStream_t s);
size_t rtus_solve_ws_bytes(int n, int n_geom, int n_tx, int n_rx);
hipError_t rtus_launch_solve(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a,
                             const double* z_a, int n_tx, const double* alpha, int n, const double* x_rx, int n_rx,
                             double z_land, double* tt, double* alpha_root, double* tt_all,
                             double* alpha_all, uint8_t* n_roots, void* ws, unsigned flags, hipStream_t s);
size_t rtus_sweep_ws_bytes(int n, int n_geom, int n_tx, int n_rx);
hipError_t rtus_launch_sweep(const rtus_lens& lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
                             const double* alpha, const double* z_f, int n, const double* x_rx, int n_rx, double atol, double rtol,
                             int32_t* first_ray, uint8_t* hit, double* tof_hit, double* tof, double* land_x, void* ws, unsigned flags,
                             hipStream_t s);
hipError_t rtus_launch_match(const double* land_x, const double* tof, int n_batch, int n, const double* x_rx,
                             int n_rx, double atol, double rtol, int32_t* first_ray,
                             uint8_t* hit, double* tof_hit, uint8_t* ray_hit, hipStream_t s);
hipError_t rtus_launch_tt_layers(const double* z_if, const double* c, int n_if, const double* xe,
                                 const double* ze, int n_e, const double* xf, const double* zf, int n_f,
                                 double* tt, uint8_t* iters, unsigned flags, hipStream_t s);
hipError_t rtus_launch_tt_layers_batch(const double* z_if, const double* c, int n_if, const double* xe, const double* ze,
                                       int n_e, long long e_stride, const double* xf, const double* zf, int n_f,
                                       long long f_stride, double* tt, long long t_stride, int n_batch, unsigned flags, hipStream_t s);
hipError_t rtus_launch_tt_layers_rows(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                                      int row0, long long n_rows_total, const double* xf, const double* zf, int n_f, double* tt,
                                      unsigned flags, hipStream_t s);
int rtus_rows_per_block(long long n_rows_total, int n_f, int n_batch, int elem_bytes);
size_t rtus_layers_sort_ws_bytes(int n_e);
hipError_t rtus_launch_tt_layers_sorted(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                                        const double* xf, const double* zf, int n_f, double* tt, void* ws, const int* presorted_row_of,
                                        unsigned flags, hipStream_t s);

hipError_t rtus_launch_tt_lens_f64(const rtus_lens& L, double a_lo, double a_hi, const double* xe, const double* ze,
                                   int n_e, const double* xf, const double* zf, int n_f, double* tt,
                                   double* alpha_out, int row0, long long n_rows_total, hipStream_t s, unsigned long long* stats = nullptr);
hipError_t rtus_launch_tt_lens_f32(const rtus_lens& L, double a_lo, double a_hi, const float* xe, const float* ze,
                                   int n_e, const float* xf, const float* zf, int n_f, float* tt, float* alpha_out,
                                   int row0, long long n_rows_total, hipStream_t s, unsigned long long* stats = nullptr);

hipError_t rtus_launch_focal_delays(const double* tt, int n_e, int n_f, double* delays, hipStream_t s);
hipError_t rtus_launch_tfm(const float* fmc, int n_tx, int n_rx, int n_t, double fs, double t0, const double* tt_tx,
                           const double* tt_rx, int n_f, float* image, hipStream_t s);

static thread_local int g_last_hip = 0;
static int hip_fail(hipError_t e) { g_last_hip = (int)e; return RTUS_ERR_HIP; }
#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hip_fail(e_); } while (0)
// a launcher reports hipGetLastError(): clear it first, so that a stale error of an unrelated earlier HIP call (the
// caller's, torch's) is not taken for this launch's
#define LAUNCH_TRY(x) do { (void)hipGetLastError(); HIP_TRY(x); } while (0)

// ---------------------------------------------------------------------------------------------------------
// Host-buffer twins: staging through a per-device arena.  The reference's calling pattern is hundreds of small
// sequential calls (main_rt.py:464-482: 210 x shoot_rays(N = 905)); a hipMalloc / hipFree pair per buffer and call
// (up to 11 of them) plus the null stream's implicit synchronisation cost more than the kernels.  So: one grow-only
// device allocation and one non-blocking stream per device, created on first use and kept until rtus_release(); a call
// locks its device's arena for its duration (host-buffer calls on ONE device are serialised; the *_dev entry points
// are untouched: caller's pointers, caller's stream, no state).
// ---------------------------------------------------------------------------------------------------------
namespace {
constexpr int kMaxDevices = 64;
constexpr int kMaxSlots = 4;                        // arenas per device: a multi-device call may list a device more than once
constexpr size_t kPinBytes = (size_t)1 << 20;       // per direction: calls that move less than this go through ONE copy
struct Arena {
    std::mutex mu;
    void* dev = nullptr;
    size_t cap = 0;
    char* pin = nullptr;                            // 2 x kPinBytes of page-locked host memory: [0, k) up, [k, 2k) down
    hipStream_t stream = nullptr;
    // rtus_shoot's lens polyline, kept between calls: the reference's script calls shoot_rays 210 times over ONE launch-angle
    // grid (main_rt.py:464-482), and the polyline launch is 9 of such a call's 43 us.  Valid for exactly the alpha values and
    // lens constants it was built from (compared byte for byte on every call).
    void* poly_ws = nullptr;
    size_t poly_cap = 0;
    std::vector<double> poly_alpha;
    rtus_lens poly_lens = {};
    bool poly_valid = false;
};
Arena g_arena[kMaxDevices][kMaxSlots];
inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// The twins run on `device` and put the caller's current device back when they return.
struct DeviceGuard {
    int prev = -1;
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};

struct Session {                       // one host-buffer call on one device
    struct Xfer { void* host; size_t off, bytes; };
    DeviceGuard guard;                 // (declared first: restored last, after the lock is gone)
    std::unique_lock<std::mutex> lock;
    Arena* a = nullptr;
    size_t off = 0;
    Xfer up[8], down[8];
    int n_up = 0, n_down = 0;

    int dev_index = -1;
    // the session ends with its stream drained whatever happened in between (an early return after flush() must not leave
    // copies in flight while the next call re-uses the page-locked buffer)
    bool drained = false;              // finish() came back clean: nothing in flight
    ~Session() { if (a && a->stream && !drained) (void)hipStreamSynchronize(a->stream); }
    int open(int device, size_t dev_bytes, int slot = 0)
    {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RTUS_ERR_NO_DEVICE;
        if (device < 0 || device >= n || device >= kMaxDevices || slot < 0 || slot >= kMaxSlots) return RTUS_ERR_NO_DEVICE;
        dev_index = device;
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != device) guard.prev = cur;
        HIP_TRY(hipSetDevice(device));
        Arena* ar = &g_arena[device][slot];
        lock = std::unique_lock<std::mutex>(ar->mu);
        a = ar;
        if (!a->stream) HIP_TRY(hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking));
        if (!a->pin) HIP_TRY(hipHostMalloc((void**)&a->pin, 2 * kPinBytes, hipHostMallocDefault));
        if (a->cap < dev_bytes) {                                    // grow-only; the previous call has synchronised
            if (a->dev) { (void)hipFree(a->dev); a->dev = nullptr; a->cap = 0; }
            const size_t want = (dev_bytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
            HIP_TRY(hipMalloc(&a->dev, want));
            a->cap = want;
        }
        return RTUS_OK;
    }
    template <class T> T* take(size_t count)                          // nullptr for count == 0
    {
        if (!count) return nullptr;
        T* p = (T*)((char*)a->dev + off);
        off += al256(count * sizeof(T));
        return p;
    }
    // inputs: carved first, so they sit next to each other at the start of the arena; copied by flush()
    template <class T> void upload(T*& d, const T* h, size_t count)
    {
        up[n_up++] = {(void*)h, off, count * sizeof(T)};
        d = take<T>(count);
    }
    // all inputs in one host-to-device copy through the page-locked buffer when they are small (the reference's calls
    // are: 15 KB in, 58 KB out), one copy each otherwise
    hipError_t flush()
    {
        if (!n_up) return hipSuccess;
        const size_t lo = up[0].off, span = up[n_up - 1].off + up[n_up - 1].bytes - lo;
        if (span <= kPinBytes) {
            for (int i = 0; i < n_up; ++i) memcpy(a->pin + (up[i].off - lo), up[i].host, up[i].bytes);
            return hipMemcpyAsync((char*)a->dev + lo, a->pin, span, hipMemcpyHostToDevice, a->stream);
        }
        for (int i = 0; i < n_up; ++i) {
            hipError_t e = hipMemcpyAsync((char*)a->dev + up[i].off, up[i].host, up[i].bytes, hipMemcpyHostToDevice, a->stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    template <class T> void download(T* h, const T* d, size_t count)
    {
        if (h && !((const char*)d >= a->pin && (const char*)d < a->pin + 2 * kPinBytes))      // (direct outputs are already registered)
            down[n_down++] = {(void*)h, (size_t)((const char*)d - (const char*)a->dev), count * sizeof(T)};
    }
    // Small results written by the kernel STRAIGHT into the page-locked buffer (it is device-visible): no device-to-host copy
    // command behind the kernel (9 of a 32-us call) — the stores cross PCIe while the kernel runs.  (The same for the inputs —
    // the kernel reading them from the page-locked buffer instead of one small host-to-device copy — was measured: no gain.)
    // Use: direct_ok(bytes of all results the call will ask for) once, then take_out() per result instead of take() + download().
    bool direct = false;
    size_t direct_off = 0;
    Xfer direct_down[8];
    int n_direct = 0;
    void direct_ok(size_t total_result_bytes) { direct = total_result_bytes + 8 * 256 <= kPinBytes / 4; }
    template <class T> T* take_out(T* host, size_t count)            // nullptr when the caller does not want this result
    {
        if (!host || !count) return nullptr;
        if (!direct) return take<T>(count);
        T* p = (T*)(a->pin + kPinBytes + direct_off);
        direct_down[n_direct++] = {(void*)host, direct_off, count * sizeof(T)};
        direct_off += al256(count * sizeof(T));
        return p;
    }
    // results back + synchronise: one device-to-host copy through the page-locked buffer when they are small
    hipError_t finish()
    {
        if (n_direct) {                                              // results are in the page-locked buffer once the stream has drained
            const hipError_t e = hipStreamSynchronize(a->stream);
            if (e != hipSuccess) return e;
            drained = true;
            for (int i = 0; i < n_direct; ++i) memcpy(direct_down[i].host, a->pin + kPinBytes + direct_down[i].off, direct_down[i].bytes);
            if (!n_down) return hipSuccess;
        }
        if (n_down) {
            size_t lo = down[0].off, hi = 0;
            for (int i = 0; i < n_down; ++i) { lo = down[i].off < lo ? down[i].off : lo; hi = down[i].off + down[i].bytes > hi ? down[i].off + down[i].bytes : hi; }
            if (hi - lo <= kPinBytes) {
                char* stage = a->pin + kPinBytes;
                hipError_t e = hipMemcpyAsync(stage, (char*)a->dev + lo, hi - lo, hipMemcpyDeviceToHost, a->stream);
                if (e != hipSuccess) return e;
                e = hipStreamSynchronize(a->stream);
                if (e != hipSuccess) return e;
                drained = true;
                for (int i = 0; i < n_down; ++i) memcpy(down[i].host, stage + (down[i].off - lo), down[i].bytes);
                return hipSuccess;
            }
            for (int i = 0; i < n_down; ++i) {
                hipError_t e = hipMemcpyAsync(down[i].host, (char*)a->dev + down[i].off, down[i].bytes, hipMemcpyDeviceToHost, a->stream);
                if (e != hipSuccess) return e;
            }
        }
        const hipError_t e = hipStreamSynchronize(a->stream);
        drained = e == hipSuccess;
        return e;
    }
};
}   // namespace

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
    const size_t tot = (size_t)n_e * n_f;
    Session S;
    if ((st = S.open(device, 2 * al256(sizeof(R) * n_e) + 2 * al256(sizeof(R) * n_f) + (alpha_out ? 2 : 1) * al256(sizeof(R) * tot))))
        return st;
    R *dxe, *dze, *dxf, *dzf;
    S.upload(dxe, xe, n_e);
    S.upload(dze, ze, n_e);
    S.upload(dxf, xf, n_f);
    S.upload(dzf, zf, n_f);
    R* dtt = S.take<R>(tot);
    R* dal = alpha_out ? S.take<R>(tot) : nullptr;
    HIP_TRY(S.flush());
    LAUNCH_TRY(launch(*lens, a_lo, a_hi, dxe, dze, n_e, dxf, dzf, n_f, dtt, dal, 0, n_e, S.a->stream, nullptr));
    S.download(tt, dtt, tot);
    S.download(alpha_out, dal, tot);
    HIP_TRY(S.finish());
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
int rtus_release(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return RTUS_ERR_NO_DEVICE;
    if (device >= n || device >= kMaxDevices) return RTUS_ERR_NO_DEVICE;
    DeviceGuard guard;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) guard.prev = cur;
    for (int d = (device < 0 ? 0 : device); d < (device < 0 ? (n < kMaxDevices ? n : kMaxDevices) : device + 1); ++d) {
        for (int sl = 0; sl < kMaxSlots; ++sl) {
            Arena& a = g_arena[d][sl];
            std::lock_guard<std::mutex> lk(a.mu);
            if (!a.dev && !a.stream && !a.pin && !a.poly_ws) continue;
            HIP_TRY(hipSetDevice(d));
            if (a.stream) { (void)hipStreamSynchronize(a.stream); (void)hipStreamDestroy(a.stream); a.stream = nullptr; }
            if (a.dev) { (void)hipFree(a.dev); a.dev = nullptr; a.cap = 0; }
            if (a.poly_ws) { (void)hipFree(a.poly_ws); a.poly_ws = nullptr; a.poly_cap = 0; }
            a.poly_valid = false; a.poly_alpha.clear(); a.poly_alpha.shrink_to_fit();
            if (a.pin) { (void)hipHostFree(a.pin); a.pin = nullptr; }
        }
    }
    return RTUS_OK;
}

int rtus_device_count(int* count)
{
    if (!count) return RTUS_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { *count = 0; return RTUS_ERR_NO_DEVICE; }
    *count = n;
    return RTUS_OK;
}

int rtus_selftest(const rtus_lens* lens, int n_rays, long long n_math, unsigned long long* counts, int device)
{
    if (!lens || !counts || n_rays < 8 || n_math < 0 || n_math > (1ll << 36)) return RTUS_ERR_INVALID_ARG;
    Session S;                                       // validates `device`, restores the caller's device, the arena's own stream
    int st = S.open(device, 0);
    if (st) return st;
    LAUNCH_TRY(rtus_selftest_run(*lens, n_rays, n_math, counts, S.a->stream));
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
#define RTUS_SHOOT_KNOWN_FLAGS (RTUS_SHOOT_FAST_MATH | RTUS_TRUE_PIPE_TANGENT | RTUS_ANALYTIC_LENS | RTUS_POLYLINE_READY)

int rtus_shoot_dev(const rtus_lens* lens, const double* d_geoms, int n_geom, const double* d_x_a,
                   const double* d_z_a, int n_tx, const double* d_alpha, const double* d_z_f, int n_rays,
                   double* d_out8, double* d_tof4, double* d_tof, double* d_land_x, uint8_t* d_status,
                   void* d_workspace, size_t workspace_bytes, unsigned flags, void* stream)
{
    int st = check_shoot(lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays);
    if (st) return st;
    if (flags & ~RTUS_SHOOT_KNOWN_FLAGS) return RTUS_ERR_INVALID_ARG;
    if (!d_workspace || ((uintptr_t)d_workspace & 63) || workspace_bytes < rtus_ws_bytes(n_rays)) return RTUS_ERR_WORKSPACE;
    LAUNCH_TRY(rtus_launch_shoot(*lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays, d_out8,
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
    const size_t rows = (size_t)n_geom * n_tx, n = (size_t)n_rays, rn = rows * n;
    const size_t need = al256(16 * (size_t)n_geom) + 2 * al256(8 * (size_t)n_tx) + 2 * al256(8 * n) +
                        (out8 ? al256(64 * rn) : 0) + (tof4 ? al256(32 * rn) : 0) + (tof ? al256(8 * rn) : 0) +
                        (land_x ? al256(8 * rn) : 0) + (status ? al256(rn) : 0);
    Session S;
    if ((st = S.open(device, need))) return st;
    // the polyline of an unchanged (alpha, lens) pair is kept in the arena: no polyline launch, no upload of alpha
    Arena& A = *S.a;
    const size_t pw = rtus_ws_bytes(n_rays);
    const bool keep = A.poly_valid && A.poly_ws && A.poly_alpha.size() == n && memcmp(&A.poly_lens, lens, sizeof(rtus_lens)) == 0 &&
                      memcmp(A.poly_alpha.data(), alpha, 8 * n) == 0;
    if (!keep) {
        A.poly_valid = false;
        if (A.poly_cap < pw) {
            if (A.poly_ws) { (void)hipFree(A.poly_ws); A.poly_ws = nullptr; A.poly_cap = 0; }
            HIP_TRY(hipMalloc(&A.poly_ws, pw));
            A.poly_cap = pw;
        }
    }
    double *g, *xa, *za, *al = nullptr, *zf;
    S.upload(g, geoms, 2 * (size_t)n_geom);
    S.upload(xa, x_a, n_tx);
    S.upload(za, z_a, n_tx);
    if (!keep) S.upload(al, alpha, n);
    S.upload(zf, z_f, n);
    void* ws = A.poly_ws;
    S.direct_ok((out8 ? 64 * rn : 0) + (tof4 ? 32 * rn : 0) + (tof ? 8 * rn : 0) + (land_x ? 8 * rn : 0) + (status ? rn : 0));
    double* o8 = S.take_out(out8, 8 * rn);
    double* t4 = S.take_out(tof4, 4 * rn);
    double* tt = S.take_out(tof, rn);
    double* lx = S.take_out(land_x, rn);
    uint8_t* sb = S.take_out(status, rn);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_shoot(*lens, g, n_geom, xa, za, n_tx, al, zf, n_rays, o8, t4, tt, lx, sb, ws,
                                 (flags & ~RTUS_POLYLINE_READY) | (keep ? RTUS_POLYLINE_READY : 0u), S.a->stream));
    S.download(out8, o8, 8 * rn);
    S.download(tof4, t4, 4 * rn);
    S.download(tof, tt, rn);
    S.download(land_x, lx, rn);
    S.download(status, sb, rn);
    HIP_TRY(S.finish());
    if (!keep && n <= ((size_t)1 << 22)) {                  // the call came back clean: the polyline in the arena belongs to this (alpha, lens)
        A.poly_alpha.assign(alpha, alpha + n);
        A.poly_lens = *lens;
        A.poly_valid = true;
    }
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- root-finding solve
size_t rtus_solve_workspace_bytes(int n_rays, int n_geom, int n_tx, int n_rx)
{
    return (n_rays > 0 && n_geom > 0 && n_tx > 0 && n_rx > 0) ? rtus_solve_ws_bytes(n_rays, n_geom, n_tx, n_rx) : 0;
}

static int check_solve(const rtus_lens* lens, const void* geoms, int n_geom, const void* x_a, const void* z_a, int n_tx,
                       const void* alpha, int n_rays, const void* x_rx, int n_rx, double z_land, const void* tt,
                       unsigned flags)
{
    int st = check_shoot(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, alpha, n_rays);
    if (st) return st;
    if (!x_rx || !tt || n_rx <= 0 || !isfinite(z_land)) return RTUS_ERR_INVALID_ARG;
    if (flags & ~(RTUS_SHOOT_KNOWN_FLAGS | RTUS_SOLVE_ONE_LANE | RTUS_SOLVE_THREE_LAUNCHES)) return RTUS_ERR_INVALID_ARG;
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
    if (!d_workspace || ((uintptr_t)d_workspace & 63) || workspace_bytes < rtus_solve_ws_bytes(n_rays, n_geom, n_tx, n_rx)) return RTUS_ERR_WORKSPACE;
    LAUNCH_TRY(rtus_launch_solve(*lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, n_rays, d_x_rx, n_rx, z_land, d_tt,
                              d_alpha_root, d_tt_all, d_alpha_all, d_n_roots, d_workspace, flags, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_solve(const rtus_lens* lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
               const double* alpha, int n_rays, const double* x_rx, int n_rx, double z_land, double* tt,
               double* alpha_root, double* tt_all, double* alpha_all, uint8_t* n_roots, unsigned flags, int device)
{
    int st = check_solve(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, n_rays, x_rx, n_rx, z_land, tt, flags);
    if (st) return st;
    for (int i = 0; i + 1 < n_rays; ++i)                    // the brackets are intervals of the grid: strictly ascending (host arrays can be checked)
        if (!(alpha[i] < alpha[i + 1])) return RTUS_ERR_INVALID_ARG;
    const size_t tot = (size_t)n_geom * n_tx * n_rx;
    const size_t wsb = rtus_solve_ws_bytes(n_rays, n_geom, n_tx, n_rx);
    const size_t need = al256(16 * (size_t)n_geom) + 2 * al256(8 * (size_t)n_tx) + al256(8 * (size_t)n_rays) + al256(8 * (size_t)n_rx) +
                        al256(wsb) + al256(8 * tot) + (alpha_root ? al256(8 * tot) : 0) +
                        (tt_all ? al256(8 * tot * RTUS_MAX_ROOTS) : 0) + (alpha_all ? al256(8 * tot * RTUS_MAX_ROOTS) : 0) +
                        (n_roots ? al256(tot) : 0);
    Session S;
    if ((st = S.open(device, need))) return st;
    double *g, *xa, *za, *al, *rx;
    S.upload(g, geoms, 2 * (size_t)n_geom);
    S.upload(xa, x_a, n_tx);
    S.upload(za, z_a, n_tx);
    S.upload(al, alpha, n_rays);
    S.upload(rx, x_rx, n_rx);
    void* ws = S.take<char>(wsb);
    S.direct_ok(8 * tot + (alpha_root ? 8 * tot : 0) + (tt_all ? 8 * tot * RTUS_MAX_ROOTS : 0) + (alpha_all ? 8 * tot * RTUS_MAX_ROOTS : 0) +
                (n_roots ? tot : 0));
    double* dt = S.take_out(tt, tot);
    double* da = S.take_out(alpha_root, tot);
    double* dta = S.take_out(tt_all, tot * RTUS_MAX_ROOTS);
    double* daa = S.take_out(alpha_all, tot * RTUS_MAX_ROOTS);
    uint8_t* dn = S.take_out(n_roots, tot);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_solve(*lens, g, n_geom, xa, za, n_tx, al, n_rays, rx, n_rx, z_land, dt, da, dta, daa, dn, ws,
                              flags & ~RTUS_POLYLINE_READY, S.a->stream));
    S.download(tt, dt, tot);
    S.download(alpha_root, da, tot);
    S.download(tt_all, dta, tot * RTUS_MAX_ROOTS);
    S.download(alpha_all, daa, tot * RTUS_MAX_ROOTS);
    S.download(n_roots, dn, tot);
    HIP_TRY(S.finish());
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
    LAUNCH_TRY(rtus_launch_match(d_land_x, d_tof, n_batch, n_rays, d_x_rx, n_rx, atol, rtol, d_first_ray,
                              d_hit, d_tof_hit, nullptr, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_ray_hits_dev(const double* d_land_x, int n_batch, int n_rays, const double* d_x_rx, int n_rx,
                      double atol, double rtol, uint8_t* d_ray_hit, void* stream)
{
    int st = check_match(d_land_x, n_batch, n_rays, d_x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (!d_ray_hit) return RTUS_ERR_INVALID_ARG;
    LAUNCH_TRY(rtus_launch_match(d_land_x, nullptr, n_batch, n_rays, d_x_rx, n_rx, atol, rtol, nullptr,
                              nullptr, nullptr, d_ray_hit, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_match(const double* land_x, const double* tof, int n_batch, int n_rays, const double* x_rx, int n_rx,
               double atol, double rtol, int32_t* first_ray, uint8_t* hit, double* tof_hit, int device)
{
    int st = check_match(land_x, n_batch, n_rays, x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (tof_hit && !tof) return RTUS_ERR_INVALID_ARG;
    const size_t rn = (size_t)n_batch * n_rays, re = (size_t)n_batch * n_rx;
    Session S;
    if ((st = S.open(device, (tof ? 2 : 1) * al256(8 * rn) + al256(8 * (size_t)n_rx) + al256(4 * re) + (hit ? al256(re) : 0) +
                                 (tof_hit ? al256(8 * re) : 0))))
        return st;
    double *lx, *tf = nullptr, *rx;
    S.upload(lx, land_x, rn);
    if (tof) S.upload(tf, tof, rn);
    S.upload(rx, x_rx, n_rx);
    int32_t* fr = S.take<int32_t>(re);
    uint8_t* hb = hit ? S.take<uint8_t>(re) : nullptr;
    double* th = tof_hit ? S.take<double>(re) : nullptr;
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_match(lx, tf, n_batch, n_rays, rx, n_rx, atol, rtol, fr, hb, th, nullptr, S.a->stream));
    S.download(first_ray, fr, re);
    S.download(hit, hb, re);
    S.download(tof_hit, th, re);
    HIP_TRY(S.finish());
    return RTUS_OK;
}

int rtus_ray_hits(const double* land_x, int n_batch, int n_rays, const double* x_rx, int n_rx, double atol,
                  double rtol, uint8_t* ray_hit, int device)
{
    int st = check_match(land_x, n_batch, n_rays, x_rx, n_rx, atol, rtol);
    if (st) return st;
    if (!ray_hit) return RTUS_ERR_INVALID_ARG;
    const size_t rn = (size_t)n_batch * n_rays;
    Session S;
    if ((st = S.open(device, al256(8 * rn) + al256(8 * (size_t)n_rx) + al256(rn)))) return st;
    double *lx, *rx;
    S.upload(lx, land_x, rn);
    S.upload(rx, x_rx, n_rx);
    uint8_t* rh = S.take<uint8_t>(rn);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_match(lx, nullptr, n_batch, n_rays, rx, n_rx, atol, rtol, nullptr, nullptr, nullptr, rh, S.a->stream));
    S.download(ray_hit, rh, rn);
    HIP_TRY(S.finish());
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- fused sweep (forward trace + matcher)
size_t rtus_sweep_workspace_bytes(int n_rays, int n_geom, int n_tx, int n_rx)
{
    return (n_rays > 0 && n_geom > 0 && n_tx > 0 && n_rx > 0) ? rtus_sweep_ws_bytes(n_rays, n_geom, n_tx, n_rx) : 0;
}

static int check_sweep(const rtus_lens* lens, const void* geoms, int n_geom, const void* x_a, const void* z_a, int n_tx,
                       const void* alpha, const void* z_f, int n_rays, const void* x_rx, int n_rx, double atol, double rtol,
                       const void* first_ray, unsigned flags)
{
    int st = check_shoot(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, z_f, n_rays);
    if (st) return st;
    if (!x_rx || !first_ray || n_rx <= 0 || !(atol >= 0) || !(rtol >= 0)) return RTUS_ERR_INVALID_ARG;
    if (flags & ~RTUS_SHOOT_KNOWN_FLAGS) return RTUS_ERR_INVALID_ARG;
    const long long rows = (long long)n_geom * n_tx, rx_pad = ((long long)n_rx + 63) & ~63LL;
    if (rows * rx_pad + rows > 0x7fffffffLL) return RTUS_ERR_UNSUPPORTED;
    return RTUS_OK;
}

int rtus_sweep_dev(const rtus_lens* lens, const double* d_geoms, int n_geom, const double* d_x_a, const double* d_z_a, int n_tx,
                   const double* d_alpha, const double* d_z_f, int n_rays, const double* d_x_rx, int n_rx, double atol, double rtol,
                   int32_t* d_first_ray, uint8_t* d_hit, double* d_tof_hit, double* d_tof, double* d_land_x, void* d_workspace,
                   size_t workspace_bytes, unsigned flags, void* stream)
{
    int st = check_sweep(lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays, d_x_rx, n_rx, atol, rtol, d_first_ray, flags);
    if (st) return st;
    if (!d_workspace || ((uintptr_t)d_workspace & 63) || workspace_bytes < rtus_sweep_ws_bytes(n_rays, n_geom, n_tx, n_rx)) return RTUS_ERR_WORKSPACE;
    LAUNCH_TRY(rtus_launch_sweep(*lens, d_geoms, n_geom, d_x_a, d_z_a, n_tx, d_alpha, d_z_f, n_rays, d_x_rx, n_rx, atol, rtol, d_first_ray,
                              d_hit, d_tof_hit, d_tof, d_land_x, d_workspace, flags, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_sweep(const rtus_lens* lens, const double* geoms, int n_geom, const double* x_a, const double* z_a, int n_tx,
               const double* alpha, const double* z_f, int n_rays, const double* x_rx, int n_rx, double atol, double rtol,
               int32_t* first_ray, uint8_t* hit, double* tof_hit, double* tof, double* land_x, unsigned flags, int device)
{
    int st = check_sweep(lens, geoms, n_geom, x_a, z_a, n_tx, alpha, z_f, n_rays, x_rx, n_rx, atol, rtol, first_ray, flags);
    if (st) return st;
    const size_t rows = (size_t)n_geom * n_tx, n = (size_t)n_rays, rn = rows * n, re = rows * (size_t)n_rx;
    const size_t wsb = rtus_sweep_ws_bytes(n_rays, n_geom, n_tx, n_rx);
    const size_t need = al256(16 * (size_t)n_geom) + 2 * al256(8 * (size_t)n_tx) + 2 * al256(8 * n) + al256(8 * (size_t)n_rx) + al256(wsb) +
                        al256(4 * re) + (hit ? al256(re) : 0) + (tof_hit ? al256(8 * re) : 0) + (tof ? al256(8 * rn) : 0) +
                        (land_x ? al256(8 * rn) : 0);
    Session S;
    if ((st = S.open(device, need))) return st;
    double *g, *xa, *za, *al, *zf, *rx;
    S.upload(g, geoms, 2 * (size_t)n_geom);
    S.upload(xa, x_a, n_tx);
    S.upload(za, z_a, n_tx);
    S.upload(al, alpha, n);
    S.upload(zf, z_f, n);
    S.upload(rx, x_rx, n_rx);
    void* ws = S.take<char>(wsb);
    S.direct_ok(4 * re + (hit ? re : 0) + (tof_hit ? 8 * re : 0) + (tof ? 8 * rn : 0) + (land_x ? 8 * rn : 0));
    int32_t* fr = S.take_out(first_ray, re);
    uint8_t* hb = S.take_out(hit, re);
    double* th = S.take_out(tof_hit, re);
    double* tt = S.take_out(tof, rn);
    double* lx = S.take_out(land_x, rn);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_sweep(*lens, g, n_geom, xa, za, n_tx, al, zf, n_rays, rx, n_rx, atol, rtol, fr, hb, th, tt, lx, ws,
                              flags & ~RTUS_POLYLINE_READY, S.a->stream));
    S.download(first_ray, fr, re);
    S.download(hit, hb, re);
    S.download(tof_hit, th, re);
    S.download(tof, tt, rn);
    S.download(land_x, lx, rn);
    HIP_TRY(S.finish());
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

// flags of the planar entries: the accuracy tier; the iteration counts are a diagnostic of the accurate tier's kernel
static int check_tier(unsigned flags, const void* iters)
{
    if (flags & ~RTUS_TT_TAUP_TAIL) return RTUS_ERR_INVALID_ARG;
    if ((flags & RTUS_TT_TAUP_TAIL) && iters) return RTUS_ERR_INVALID_ARG;
    return RTUS_OK;
}

int rtus_tt_layers_ex_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                          int n_e, const double* d_xf, const double* d_zf, int n_f, double* d_tt, uint8_t* d_iters,
                          unsigned flags, void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_tier(flags, d_iters))) return st;
    LAUNCH_TRY(rtus_launch_tt_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_iters, flags,
                                  (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_layers_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                       int n_e, const double* d_xf, const double* d_zf, int n_f, double* d_tt, uint8_t* d_iters,
                       void* stream)
{
    return rtus_tt_layers_ex_dev(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_iters, 0u, stream);
}

int rtus_tt_layers_batch_ex_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                                int n_e, long long e_stride, const double* d_xf, const double* d_zf, int n_f,
                                long long f_stride, double* d_tt, long long t_stride, int n_batch, unsigned flags, void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_tier(flags, nullptr))) return st;
    if (n_batch <= 0 || e_stride < 0 || f_stride < 0) return RTUS_ERR_INVALID_ARG;
    if (n_batch > 1 && t_stride < (long long)n_e * n_f) return RTUS_ERR_INVALID_ARG;   // outputs of two problems would overlap
    if (n_batch > 65535) return RTUS_ERR_UNSUPPORTED;                                   // grid.z
    LAUNCH_TRY(rtus_launch_tt_layers_batch(z_if, c, n_if, d_xe, d_ze, n_e, e_stride, d_xf, d_zf, n_f, f_stride, d_tt,
                                        t_stride, n_batch, flags, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_layers_batch_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze,
                             int n_e, long long e_stride, const double* d_xf, const double* d_zf, int n_f,
                             long long f_stride, double* d_tt, long long t_stride, int n_batch, void* stream)
{
    return rtus_tt_layers_batch_ex_dev(z_if, c, n_if, d_xe, d_ze, n_e, e_stride, d_xf, d_zf, n_f, f_stride, d_tt, t_stride, n_batch, 0u,
                                       stream);
}

size_t rtus_tt_layers_sort_workspace_bytes(int n_e) { return n_e > 0 ? rtus_layers_sort_ws_bytes(n_e) : 0; }

#define RTUS_SORT_MAX_ELEMENTS 32768                 /* the device-side rank is O(n^2) compares */
int rtus_tt_layers_sorted_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze, int n_e,
                              const double* d_xf, const double* d_zf, int n_f, double* d_tt, void* d_workspace, size_t workspace_bytes,
                              unsigned flags, void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    if (flags & ~RTUS_TT_TAUP_TAIL) return RTUS_ERR_INVALID_ARG;
    if (n_e > RTUS_SORT_MAX_ELEMENTS) return RTUS_ERR_UNSUPPORTED;
    if (!d_workspace || ((uintptr_t)d_workspace & 255) || workspace_bytes < rtus_layers_sort_ws_bytes(n_e)) return RTUS_ERR_WORKSPACE;
    LAUNCH_TRY(rtus_launch_tt_layers_sorted(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_workspace, nullptr, flags,
                                         (hipStream_t)stream));
    return RTUS_OK;
}

// The aperture in (depth, position) order — the order the predictor of the kernel wants; the coordinates are host memory in the
// host-buffer twins, so the sort is the host's: an aperture that arrives in that order (the usual case) goes through unchanged
// (`order` stays empty).  Compared on the IEEE bits made monotone — the key the device-side rank kernel uses — so the order is a
// strict weak order whatever the values (a NaN coordinate sorts last and fails its row only).
static unsigned long long host_order_key(double v)
{
    unsigned long long b;
    memcpy(&b, &v, 8);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
static void aperture_order(const double* xe, const double* ze, int n_e, std::vector<int>& order, std::vector<double>& sx, std::vector<double>& sz)
{
    auto before = [&](int a, int b) {
        const unsigned long long za = host_order_key(ze[a]), zb = host_order_key(ze[b]);
        return za < zb || (za == zb && host_order_key(xe[a]) < host_order_key(xe[b]));
    };
    bool sorted = true;
    for (int i = 1; i < n_e && sorted; ++i) sorted = !before(i, i - 1);
    if (sorted) return;
    order.resize(n_e);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), before);
    sx.resize(n_e); sz.resize(n_e);
    for (int i = 0; i < n_e; ++i) { sx[i] = xe[order[i]]; sz[i] = ze[order[i]]; }
}

int rtus_tt_layers_ex(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                      const double* xf, const double* zf, int n_f, double* tt, uint8_t* iters, unsigned flags, int device)
{
    int st = check_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt);
    if (st || (st = check_tier(flags, iters))) return st;
    const size_t tot = (size_t)n_e * n_f;
    std::vector<int> order;
    std::vector<double> sx, sz;
    if (!iters) aperture_order(xe, ze, n_e, order, sx, sz);     // (the iteration counts are a diagnostic of the elements AS GIVEN)
    const bool perm = !order.empty();
    Session S;
    if ((st = S.open(device, 2 * al256(8 * (size_t)n_e) + 2 * al256(8 * (size_t)n_f) + al256(8 * tot) + (iters ? al256(tot) : 0) +
                                 (perm ? al256(4 * (size_t)n_e) : 0))))
        return st;
    double *dxe, *dze, *dxf, *dzf;
    int* drow = nullptr;
    S.upload(dxe, perm ? sx.data() : xe, n_e);
    S.upload(dze, perm ? sz.data() : ze, n_e);
    S.upload(dxf, xf, n_f);
    S.upload(dzf, zf, n_f);
    if (perm) S.upload(drow, (const int*)order.data(), n_e);
    double* dtt = S.take<double>(tot);
    uint8_t* dit = iters ? S.take<uint8_t>(tot) : nullptr;
    HIP_TRY(S.flush());
    if (perm) LAUNCH_TRY(rtus_launch_tt_layers_sorted(z_if, c, n_if, dxe, dze, n_e, dxf, dzf, n_f, dtt, nullptr, drow, flags, S.a->stream));
    else LAUNCH_TRY(rtus_launch_tt_layers(z_if, c, n_if, dxe, dze, n_e, dxf, dzf, n_f, dtt, dit, flags, S.a->stream));
    S.download(tt, dtt, tot);
    S.download(iters, dit, tot);
    HIP_TRY(S.finish());
    return RTUS_OK;
}

int rtus_tt_layers(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e,
                   const double* xf, const double* zf, int n_f, double* tt, uint8_t* iters, int device)
{
    return rtus_tt_layers_ex(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, iters, 0u, device);
}

// ---------------------------------------------------------------------------- consumers: focal laws, TFM
static int check_tfm(const void* fmc, int n_tx, int n_rx, int n_t, double fs, double t0, const void* tt_tx, const void* tt_rx,
                     int n_f, const void* image)
{
    if (!fmc || !tt_tx || !tt_rx || !image || n_tx <= 0 || n_rx <= 0 || n_t < 2 || n_f <= 0) return RTUS_ERR_INVALID_ARG;
    if (!(fs > 0) || !isfinite(fs) || !isfinite(t0)) return RTUS_ERR_INVALID_ARG;
    if (n_t > (1 << 28)) return RTUS_ERR_UNSUPPORTED;                 // a record is addressed with 32-bit byte offsets
    return RTUS_OK;
}

int rtus_focal_delays_dev(const double* d_tt, int n_e, int n_f, double* d_delays, void* stream)
{
    if (!d_tt || !d_delays || n_e <= 0 || n_f <= 0) return RTUS_ERR_INVALID_ARG;
    LAUNCH_TRY(rtus_launch_focal_delays(d_tt, n_e, n_f, d_delays, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_focal_delays(const double* tt, int n_e, int n_f, double* delays, int device)
{
    if (!tt || !delays || n_e <= 0 || n_f <= 0) return RTUS_ERR_INVALID_ARG;
    const size_t tot = (size_t)n_e * n_f;
    Session S;
    int st = S.open(device, al256(8 * tot));
    if (st) return st;
    double* d;
    S.upload(d, tt, tot);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_focal_delays(d, n_e, n_f, d, S.a->stream));     // in place: each entry is read before it is written
    S.download(delays, d, tot);
    HIP_TRY(S.finish());
    return RTUS_OK;
}

int rtus_tfm_dev(const float* d_fmc, int n_tx, int n_rx, int n_t, double fs, double t0, const double* d_tt_tx,
                 const double* d_tt_rx, int n_f, float* d_image, void* stream)
{
    int st = check_tfm(d_fmc, n_tx, n_rx, n_t, fs, t0, d_tt_tx, d_tt_rx, n_f, d_image);
    if (st) return st;
    LAUNCH_TRY(rtus_launch_tfm(d_fmc, n_tx, n_rx, n_t, fs, t0, d_tt_tx, d_tt_rx, n_f, d_image, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tfm(const float* fmc, int n_tx, int n_rx, int n_t, double fs, double t0, const double* tt_tx, const double* tt_rx,
             int n_f, float* image, int device)
{
    int st = check_tfm(fmc, n_tx, n_rx, n_t, fs, t0, tt_tx, tt_rx, n_f, image);
    if (st) return st;
    const size_t nfmc = (size_t)n_tx * n_rx * n_t;
    const bool same = tt_tx == tt_rx && n_tx == n_rx;
    Session S;
    if ((st = S.open(device, al256(4 * nfmc) + (same ? 1 : 2) * al256(8 * (size_t)(n_tx > n_rx ? n_tx : n_rx) * n_f) + al256(4 * (size_t)n_f))))
        return st;
    float* dfmc;
    double *dtx, *drx;
    S.upload(dfmc, fmc, nfmc);
    S.upload(dtx, tt_tx, (size_t)n_tx * n_f);
    if (same) drx = dtx; else S.upload(drx, tt_rx, (size_t)n_rx * n_f);
    float* dimg = S.take<float>(n_f);
    HIP_TRY(S.flush());
    LAUNCH_TRY(rtus_launch_tfm(dfmc, n_tx, n_rx, n_t, fs, t0, dtx, drx, n_f, dimg, S.a->stream));
    S.download(image, dimg, (size_t)n_f);
    HIP_TRY(S.finish());
    return RTUS_OK;
}

// ---------------------------------------------------------------------------- curved lens
int rtus_tt_lens_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const double* d_xe, const double* d_ze,
                     int n_e, const double* d_xf, const double* d_zf, int n_f, double* d_tt, double* d_alpha_out,
                     void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    LAUNCH_TRY(rtus_launch_tt_lens_f64(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_alpha_out, 0, n_e,
                                    (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_lens_f32_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* d_xe, const float* d_ze,
                         int n_e, const float* d_xf, const float* d_zf, int n_f, float* d_tt, float* d_alpha_out,
                         void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt);
    if (st) return st;
    LAUNCH_TRY(rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, d_alpha_out, 0, n_e,
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

// ---------------------------------------------------------------------------- row shards of a table, several devices
int rtus_table_rows_per_block(long long n_rows_total, int n_f, int elem_bytes)
{
    if (n_rows_total <= 0 || n_f <= 0 || (elem_bytes != 4 && elem_bytes != 8)) return RTUS_ERR_INVALID_ARG;
    const int eb = rtus_rows_per_block(n_rows_total, n_f, 1, elem_bytes);
    return eb < 1 ? RTUS_ERR_UNSUPPORTED : eb;
}

long long rtus_shard_rows(long long n_rows_total, int n_f, int elem_bytes, int n_shards)
{
    if (n_shards <= 0) return RTUS_ERR_INVALID_ARG;
    const int eb = rtus_table_rows_per_block(n_rows_total, n_f, elem_bytes);
    if (eb < 0) return eb;
    const long long per = (n_rows_total + n_shards - 1) / n_shards;
    return (per + eb - 1) / eb * eb;
}

static int check_rows(int n_rows, long long row0, long long n_rows_total)
{
    if (n_rows <= 0 || row0 < 0 || n_rows_total < row0 + n_rows || n_rows_total > 65535LL * 64) return RTUS_ERR_INVALID_ARG;
    return RTUS_OK;
}

int rtus_tt_layers_rows_dev(const double* z_if, const double* c, int n_if, const double* d_xe, const double* d_ze, int n_rows,
                            long long row0, long long n_rows_total, const double* d_xf, const double* d_zf, int n_f, double* d_tt,
                            unsigned flags, void* stream)
{
    int st = check_layers(z_if, c, n_if, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_rows(n_rows, row0, n_rows_total))) return st;
    if (flags & ~RTUS_TT_TAUP_TAIL) return RTUS_ERR_INVALID_ARG;
    LAUNCH_TRY(rtus_launch_tt_layers_rows(z_if, c, n_if, d_xe, d_ze, n_rows, (int)row0, n_rows_total, d_xf, d_zf, n_f, d_tt, flags,
                                       (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_lens_rows_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const double* d_xe, const double* d_ze, int n_rows,
                          long long row0, long long n_rows_total, const double* d_xf, const double* d_zf, int n_f, double* d_tt,
                          double* d_alpha_out, void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_rows(n_rows, row0, n_rows_total))) return st;
    LAUNCH_TRY(rtus_launch_tt_lens_f64(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt, d_alpha_out, (int)row0,
                                    n_rows_total, (hipStream_t)stream));
    return RTUS_OK;
}

int rtus_tt_lens_f32_rows_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* d_xe, const float* d_ze, int n_rows,
                              long long row0, long long n_rows_total, const float* d_xf, const float* d_zf, int n_f, float* d_tt,
                              float* d_alpha_out, void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_rows(n_rows, row0, n_rows_total))) return st;
    LAUNCH_TRY(rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt, d_alpha_out, (int)row0,
                                    n_rows_total, (hipStream_t)stream));
    return RTUS_OK;
}

// How the lens table's rows were solved (diagnostic; tests/test_gpu_irregular_apertures.py): the same launch as
// rtus_tt_lens[_f32]_rows_dev without the alpha output, plus five counters ADDED to d_stats (device memory, 5 x uint64, zeroed by
// the caller), each in wave-elements (one wave = 64 targets of one row): [0] rows that took T alone at the extrapolated start,
// [1] rows solved by one evaluation of T and g, [2] rows that needed the safeguarded iteration, [3] of those, rows that also
// looked at the whole interval (a lane pinned at an end or nearly flat in alpha: two minima may compete), [4] evaluations spent in [2].
int rtus_tt_lens_stats_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const double* d_xe, const double* d_ze, int n_rows,
                           long long row0, long long n_rows_total, const double* d_xf, const double* d_zf, int n_f, double* d_tt,
                           unsigned long long* d_stats, void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_rows(n_rows, row0, n_rows_total))) return st;
    if (!d_stats) return RTUS_ERR_INVALID_ARG;
    LAUNCH_TRY(rtus_launch_tt_lens_f64(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt, nullptr, (int)row0,
                                    n_rows_total, (hipStream_t)stream, d_stats));
    return RTUS_OK;
}

int rtus_tt_lens_f32_stats_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* d_xe, const float* d_ze, int n_rows,
                               long long row0, long long n_rows_total, const float* d_xf, const float* d_zf, int n_f, float* d_tt,
                               unsigned long long* d_stats, void* stream)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt);
    if (st || (st = check_rows(n_rows, row0, n_rows_total))) return st;
    if (!d_stats) return RTUS_ERR_INVALID_ARG;
    LAUNCH_TRY(rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, d_xe, d_ze, n_rows, d_xf, d_zf, n_f, d_tt, nullptr, (int)row0,
                                    n_rows_total, (hipStream_t)stream, d_stats));
    return RTUS_OK;
}

}   // extern "C" (the multi-device host entries share a template)

// One host-buffer call spread over several devices: the table's rows in contiguous blocks (multiples of the table's rows per
// workgroup: every row comes out with the bits the one-device call gives it), one arena + stream per listed device, every
// device copying its block straight into the caller's rows.  No exchange between the devices: a host result needs none.
template <typename R, typename Launch>
static int table_multi(const R* xe, const R* ze, int n_e, const R* xf, const R* zf, int n_f, R* tt, const int* devices, int n_dev,
                       Launch launch)
{
    if (!devices || n_dev <= 0 || n_dev > 64) return RTUS_ERR_INVALID_ARG;
    const long long per = rtus_shard_rows(n_e, n_f, (int)sizeof(R), n_dev);
    if (per < 0) return (int)per;
    // The caller's current device comes back whatever the sessions do: this guard is declared before them, so it is the last to
    // run (each session's own guard restores the device that was current when IT opened — the previous entry of the list).
    DeviceGuard outer;
    { int cur = -1; if (hipGetDevice(&cur) == hipSuccess) outer.prev = cur; }
    std::vector<Session> S(n_dev);
    std::vector<int> live(n_dev, 0), slot(n_dev, 0), ord;
    for (int i = 0; i < n_dev; ++i) {
        for (int j = 0; j < i; ++j) slot[i] += devices[j] == devices[i];
        const long long lo = per * i < n_e ? per * i : n_e, hi = lo + per < n_e ? lo + per : n_e;
        if (hi > lo) ord.push_back(i);
    }
    // arenas are locked in (device, slot) order, whatever the order of the list: two threads calling with [0, 1] and [1, 0]
    // take the two mutexes in the same order
    std::sort(ord.begin(), ord.end(), [&](int a, int b) { return devices[a] != devices[b] ? devices[a] < devices[b] : slot[a] < slot[b]; });
    int st = RTUS_OK;
    for (int i : ord) {                                               // uploads + launches: asynchronous on each device's stream
        const long long lo = per * i, hi = lo + per < n_e ? lo + per : n_e;
        const size_t rows = (size_t)(hi - lo), tot = rows * n_f;
        if ((st = S[i].open(devices[i], 2 * al256(sizeof(R) * rows) + 2 * al256(sizeof(R) * (size_t)n_f) + al256(sizeof(R) * tot), slot[i]))) break;
        R *dxe, *dze, *dxf, *dzf;
        S[i].upload(dxe, xe + lo, rows);
        S[i].upload(dze, ze + lo, rows);
        S[i].upload(dxf, xf, (size_t)n_f);
        S[i].upload(dzf, zf, (size_t)n_f);
        R* dtt = S[i].template take<R>(tot);
        hipError_t e = S[i].flush();
        if (e == hipSuccess) { (void)hipGetLastError(); e = launch(dxe, dze, (int)rows, (int)lo, dxf, dzf, dtt, S[i].a->stream); }
        if (e != hipSuccess) { st = hip_fail(e); break; }
        S[i].download(tt + (size_t)lo * n_f, dtt, tot);
        live[i] = 1;
    }
    // results back: one host thread per device (a device-to-pageable-host copy occupies its caller; the links are independent)
    std::vector<hipError_t> err(n_dev, hipSuccess);
    std::vector<std::thread> th;
    int mine = -1;
    for (int i = 0; i < n_dev; ++i) {
        if (!live[i]) continue;
        if (mine < 0) { mine = i; continue; }
        th.emplace_back([&, i] { (void)hipSetDevice(S[i].dev_index); err[i] = S[i].finish(); });
    }
    if (mine >= 0) { (void)hipSetDevice(S[mine].dev_index); err[mine] = S[mine].finish(); }
    for (auto& t : th) t.join();
    for (int i = 0; i < n_dev; ++i) if (st == RTUS_OK && err[i] != hipSuccess) st = hip_fail(err[i]);
    return st;
}

// ---- RCCL, bound at run time (dlopen: librtus.so does not depend on it; a process that already holds PyTorch's librccl
// gets that one) — only the *_multi_dev reassembly needs it
namespace {
typedef struct ncclComm* ncclComm_t;
struct Rccl {
    void* h = nullptr;
    int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    std::mutex mu;
    std::vector<int> devs;               // device list of the cached communicators
    std::vector<ncclComm_t> comms;
    bool ok = false;                     // every symbol bound (a library that lacks one stays loaded and unused)
    bool load()
    {
        if (h) return ok;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(h, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(h, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(h, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(h, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(h, "ncclGroupEnd");
        return ok = CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    }
};
Rccl g_rccl;
constexpr int kNcclFloat32 = 7, kNcclFloat64 = 8;   // ncclDataType_t (nccl.h)
}   // namespace

// Each listed device solves its row block INTO its own copy of the (padded) table, d_tt[i] + lo n_f; with `gather` the blocks are
// then exchanged in place by one ncclAllGather per device (single-process communicators, ncclCommInitAll: RCCL over xGMI), so
// every device ends up with the whole table; without it the table stays sharded.  Asynchronous on the given streams.
template <typename R, typename Launch>
static int table_multi_dev(const R* const* d_xe, const R* const* d_ze, int n_e, int n_f, R* const* d_tt, const int* devices, int n_dev,
                           void* const* streams, int gather, Launch launch)
{
    if (!d_xe || !d_ze || !d_tt || !devices || !streams || n_dev <= 0 || n_dev > 64) return RTUS_ERR_INVALID_ARG;
    const long long per = rtus_shard_rows(n_e, n_f, (int)sizeof(R), n_dev);
    if (per < 0) return (int)per;
    DeviceGuard guard;
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess) guard.prev = cur;
    for (int i = 0; i < n_dev; ++i) {
        const long long lo = per * i < n_e ? per * i : n_e, hi = lo + per < n_e ? lo + per : n_e;
        if (hi <= lo) continue;
        if (!d_xe[i] || !d_ze[i] || !d_tt[i]) return RTUS_ERR_INVALID_ARG;
        HIP_TRY(hipSetDevice(devices[i]));
        (void)hipGetLastError();
        HIP_TRY(launch(i, d_xe[i] + lo, d_ze[i] + lo, (int)(hi - lo), (int)lo, d_tt[i] + (size_t)lo * n_f, (hipStream_t)streams[i]));
    }
    if (!gather || n_dev == 1) return RTUS_OK;
    std::lock_guard<std::mutex> lk(g_rccl.mu);
    if (!g_rccl.load()) return RTUS_ERR_UNSUPPORTED;
    if (g_rccl.devs != std::vector<int>(devices, devices + n_dev)) {
        // communicators of another device list: collectives of an earlier asynchronous call may still be queued on them
        for (size_t k = 0; k < g_rccl.comms.size(); ++k) {
            if (k < g_rccl.devs.size() && hipSetDevice(g_rccl.devs[k]) == hipSuccess) (void)hipDeviceSynchronize();
            if (g_rccl.comms[k]) (void)g_rccl.CommDestroy(g_rccl.comms[k]);
        }
        g_rccl.comms.assign(n_dev, nullptr);
        g_rccl.devs.clear();
        if (g_rccl.CommInitAll(g_rccl.comms.data(), n_dev, devices) != 0) { g_rccl.comms.clear(); return RTUS_ERR_UNSUPPORTED; }
        g_rccl.devs.assign(devices, devices + n_dev);
    }
    int bad = g_rccl.GroupStart();
    if (bad) return RTUS_ERR_UNSUPPORTED;
    for (int i = 0; i < n_dev && !bad; ++i) {
        if (hipSetDevice(devices[i]) != hipSuccess) { bad = 1; break; }       // (the group is closed below on every path)
        bad = g_rccl.AllGather(d_tt[i] + (size_t)per * i * n_f, d_tt[i], (size_t)per * n_f, sizeof(R) == 8 ? kNcclFloat64 : kNcclFloat32,
                               g_rccl.comms[i], (hipStream_t)streams[i]);
    }
    bad |= g_rccl.GroupEnd();
    return bad ? RTUS_ERR_UNSUPPORTED : RTUS_OK;
}

extern "C" {

int rtus_tt_layers_multi_ex(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e, const double* xf,
                            const double* zf, int n_f, double* tt, const int* devices, int n_dev, unsigned flags)
{
    int st = check_layers(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt);
    if (st || (st = check_tier(flags, nullptr))) return st;
    // as the one-device twin: the aperture in (depth, position) order first, so that an element's bits do not depend on the order
    // it was handed over in and the table is the one-device table whatever that order.  The shards are blocks of the SORTED rows;
    // they come back into a staging table and every row is copied where it belongs.
    std::vector<int> order;
    std::vector<double> sx, sz, staged;
    aperture_order(xe, ze, n_e, order, sx, sz);
    const bool perm = !order.empty();
    if (perm) staged.resize((size_t)n_e * n_f);
    st = table_multi<double>(perm ? sx.data() : xe, perm ? sz.data() : ze, n_e, xf, zf, n_f, perm ? staged.data() : tt, devices, n_dev,
                             [&](const double* dxe, const double* dze, int rows, int row0, const double* dxf, const double* dzf, double* dtt,
                                 hipStream_t s) { return rtus_launch_tt_layers_rows(z_if, c, n_if, dxe, dze, rows, row0, n_e, dxf, dzf, n_f, dtt, flags, s); });
    if (st == RTUS_OK && perm)
        for (int i = 0; i < n_e; ++i) memcpy(tt + (size_t)order[i] * n_f, staged.data() + (size_t)i * n_f, sizeof(double) * (size_t)n_f);
    return st;
}

int rtus_tt_layers_multi(const double* z_if, const double* c, int n_if, const double* xe, const double* ze, int n_e, const double* xf,
                         const double* zf, int n_f, double* tt, const int* devices, int n_dev)
{
    return rtus_tt_layers_multi_ex(z_if, c, n_if, xe, ze, n_e, xf, zf, n_f, tt, devices, n_dev, 0u);
}

int rtus_tt_lens_f32_multi(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* xe, const float* ze, int n_e,
                           const float* xf, const float* zf, int n_f, float* tt, const int* devices, int n_dev)
{
    int st = check_lens(lens, alpha_lo, alpha_hi, xe, ze, n_e, xf, zf, n_f, tt);
    if (st) return st;
    return table_multi<float>(xe, ze, n_e, xf, zf, n_f, tt, devices, n_dev,
                              [&](const float* dxe, const float* dze, int rows, int row0, const float* dxf, const float* dzf, float* dtt,
                                  hipStream_t s) {
                                  return rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, dxe, dze, rows, dxf, dzf, n_f, dtt, nullptr, row0, n_e, s);
                              });
}

int rtus_tt_layers_multi_ex_dev(const double* z_if, const double* c, int n_if, const double* const* d_xe, const double* const* d_ze, int n_e,
                                const double* const* d_xf, const double* const* d_zf, int n_f, double* const* d_tt, const int* devices,
                                int n_dev, void* const* streams, int gather, unsigned flags)
{
    if (!d_xf || !d_zf || !d_xe || !d_ze || !d_tt || n_dev <= 0) return RTUS_ERR_INVALID_ARG;
    int st = check_layers(z_if, c, n_if, d_xe[0], d_ze[0], n_e, d_xf[0], d_zf[0], n_f, d_tt[0]);
    if (st || (st = check_tier(flags, nullptr))) return st;
    return table_multi_dev<double>(d_xe, d_ze, n_e, n_f, d_tt, devices, n_dev, streams, gather,
                                   [&](int i, const double* xe, const double* ze, int rows, int row0, double* tt, hipStream_t s) {
                                       return rtus_launch_tt_layers_rows(z_if, c, n_if, xe, ze, rows, row0, n_e, d_xf[i], d_zf[i], n_f, tt, flags, s);
                                   });
}

int rtus_tt_layers_multi_dev(const double* z_if, const double* c, int n_if, const double* const* d_xe, const double* const* d_ze, int n_e,
                             const double* const* d_xf, const double* const* d_zf, int n_f, double* const* d_tt, const int* devices,
                             int n_dev, void* const* streams, int gather)
{
    return rtus_tt_layers_multi_ex_dev(z_if, c, n_if, d_xe, d_ze, n_e, d_xf, d_zf, n_f, d_tt, devices, n_dev, streams, gather, 0u);
}

int rtus_tt_lens_f32_multi_dev(const rtus_lens* lens, double alpha_lo, double alpha_hi, const float* const* d_xe, const float* const* d_ze,
                               int n_e, const float* const* d_xf, const float* const* d_zf, int n_f, float* const* d_tt,
                               const int* devices, int n_dev, void* const* streams, int gather)
{
    if (!d_xf || !d_zf || !d_xe || !d_ze || !d_tt || n_dev <= 0) return RTUS_ERR_INVALID_ARG;
    int st = check_lens(lens, alpha_lo, alpha_hi, d_xe[0], d_ze[0], n_e, d_xf[0], d_zf[0], n_f, d_tt[0]);
    if (st) return st;
    return table_multi_dev<float>(d_xe, d_ze, n_e, n_f, d_tt, devices, n_dev, streams, gather,
                                  [&](int i, const float* xe, const float* ze, int rows, int row0, float* tt, hipStream_t s) {
                                      return rtus_launch_tt_lens_f32(*lens, alpha_lo, alpha_hi, xe, ze, rows, d_xf[i], d_zf[i], n_f, tt, nullptr,
                                                                     row0, n_e, s);
                                  });
}

}   // extern "C"
