// rtus_match.hip — element matcher (reference main_rt.py:487-501, main_compare.py:518-521).
//
// The reference walks elements x rays with scalar np.isclose calls and keeps, per element, the
// FIRST ray (ascending index) whose landing x is within atol + rtol*|x_rx| of the element.
// Here: one ray per lane; the receive aperture (x_rx, tolerance) is staged once per workgroup in
// LDS; the "first ray" per element is an integer atomicMin (order-independent, so the result is
// bit-identical to the sequential scan); a second tiny kernel turns the winner into hit / tof.
#include "rtus_device.h"

#define RTUS_NO_RAY 0x7f7f7f7f   // hipMemsetAsync(0x7f) sentinel; larger than any ray index (also rtus_trace.h: the fused sweep)

struct MatchArgs {
    const double* __restrict__ land_x;  // [n_batch][n]
    const double* __restrict__ x_rx;    // [n_rx]
    int32_t* __restrict__ first_ray;    // [n_batch][n_rx]   nullable for ray_hits
    uint8_t* __restrict__ ray_hit;      // [n_batch][n]      nullable for match
    double atol, rtol;
    int n, n_rx, n_batch;
    int row0;                           // first batch row of this launch (grid.y is limited to 65535 rows)
    int chunks;                         // 256-ray chunks per workgroup (the aperture is staged once per workgroup)
};

// Workgroup = 256 rays of one batch row.  LDS: x_rx[n_rx] then tol[n_rx] (dynamic).
// While staging the aperture the workgroup also finds out whether x_rx is ascending (arrays are):
// then each ray binary-searches the window [x - win, x + win], win = atol + rtol*max|x_rx|, instead
// of testing every element.  Any other order falls back to the full scan — same result either way.
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_match_kernel(MatchArgs a)
{
    extern __shared__ double lds[];
    __shared__ double s_amax[RTUS_BLOCK / 64];
    double* sx = lds;
    double* stol = lds + a.n_rx;
    bool asc = true;
    double amax = 0.0;
    for (int e = threadIdx.x; e < a.n_rx; e += RTUS_BLOCK) {
        const double xe = a.x_rx[e];
        sx[e] = xe;
        // np.isclose(a, b): |a - b| <= atol + rtol*|b|, b = elem_x (two roundings, as NumPy) for finite operands, a == b when either
        // is infinite, never with a NaN.  A negative tolerance keeps an infinite element out of the first test.
        stol[e] = isfinite(xe) ? __dadd_rn(a.atol, __dmul_rn(a.rtol, fabs(xe))) : -1.0;
        asc = asc && isfinite(xe) && (e == 0 || a.x_rx[e - 1] <= xe);
        amax = fmax(amax, fabs(xe));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmax(amax, __shfl_xor(amax, o));
    if ((threadIdx.x & 63) == 0) s_amax[threadIdx.x >> 6] = amax;
    const bool sorted = __syncthreads_and(asc);    // also publishes sx / stol / s_amax
    amax = fmax(fmax(s_amax[0], s_amax[1]), fmax(s_amax[2], s_amax[3]));
    const double win = a.atol + a.rtol * amax;

    const int row = a.row0 + blockIdx.y;
    // `chunks` consecutive 256-ray chunks of the row per workgroup: staging the aperture (a dependent chain of loads, two
    // reductions and a barrier: ~3 us) is paid once per workgroup, and at 256 rays per workgroup it WAS the kernel on
    // large batches (8.4 M rays: 32,768 workgroups, 16 rounds of that latency)
    for (int c = 0; c < a.chunks; ++c) {
        const int r = (blockIdx.x * a.chunks + c) * RTUS_BLOCK + threadIdx.x;
        if (r >= a.n) return;                      // (no barrier below: a thread may leave alone)
        const double x = a.land_x[(size_t)row * a.n + r];
        bool any = false;
        if (x == x) {                              // NaN never matches; an infinite landing point only an equal infinite element
            int e0 = 0, e1 = a.n_rx;
            if (sorted) {                          // lower_bound(x - win)
                const double lo = x - win;
                int l = 0, h = a.n_rx;
                while (l < h) { const int mid = (l + h) >> 1; if (sx[mid] < lo) l = mid + 1; else h = mid; }
                e0 = l;
            }
            for (int e = e0; e < e1; ++e) {
                const double xe = sx[e];
                if (sorted && xe > x + win) break;
                if (fabs(x - xe) <= stol[e] || (isinf(xe) && x == xe)) {
                    any = true;
                    if (a.first_ray) atomicMin(&a.first_ray[(size_t)row * a.n_rx + e], r);
                    else break;
                }
            }
        }
        if (a.ray_hit) a.ray_hit[(size_t)row * a.n + r] = any ? 1 : 0;
    }
}

// first_ray sentinel -> -1, hit flag, tof of the first hitting ray (0.0 when none: main_rt.py:493).
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_match_finalize_kernel(int32_t* __restrict__ first_ray,
                                                                          const double* __restrict__ tof, int n,
                                                                          int n_rx, int n_batch,
                                                                          uint8_t* __restrict__ hit,
                                                                          double* __restrict__ tof_hit)
{
    const size_t i = (size_t)blockIdx.x * RTUS_BLOCK + threadIdx.x;
    if (i >= (size_t)n_batch * n_rx) return;
    const int row = (int)(i / n_rx);
    const int f = first_ray[i];
    const bool h = f != RTUS_NO_RAY;
    first_ray[i] = h ? f : -1;
    if (hit) hit[i] = h ? 1 : 0;
    if (tof_hit) tof_hit[i] = (h && tof) ? tof[(size_t)row * n + f] : 0.0;
}

hipError_t rtus_launch_match(const double* land_x, const double* tof, int n_batch, int n, const double* x_rx,
                             int n_rx, double atol, double rtol, int32_t* first_ray,
                             uint8_t* hit, double* tof_hit, uint8_t* ray_hit, hipStream_t s)
{
    MatchArgs a;
    a.land_x = land_x; a.x_rx = x_rx; a.first_ray = first_ray; a.ray_hit = ray_hit;
    a.atol = atol; a.rtol = rtol; a.n = n; a.n_rx = n_rx; a.n_batch = n_batch;
    const size_t lds = (size_t)n_rx * 2 * sizeof(double);
    if (first_ray) {
        hipError_t e = hipMemsetAsync(first_ray, 0x7f, (size_t)n_batch * n_rx * sizeof(int32_t), s);
        if (e != hipSuccess) return e;
    }
    // chunks per workgroup: as many as leave >= ~2048 workgroups (8 per CU) in the launch, at most 16
    const long long blocks = (long long)((n + RTUS_BLOCK - 1) / RTUS_BLOCK) * n_batch;
    long long chunks = blocks / 2048;
    a.chunks = (int)(chunks < 1 ? 1 : (chunks > 16 ? 16 : chunks));
    const int per_wg = RTUS_BLOCK * a.chunks;
    for (int row0 = 0; row0 < n_batch; row0 += 65535) {
        a.row0 = row0;
        hipLaunchKernelGGL(rtus_match_kernel, dim3((n + per_wg - 1) / per_wg, min(65535, n_batch - row0)),
                           dim3(RTUS_BLOCK), lds, s, a);
    }
    if (first_ray) {
        const size_t tot = (size_t)n_batch * n_rx;
        hipLaunchKernelGGL(rtus_match_finalize_kernel, dim3((unsigned)((tot + RTUS_BLOCK - 1) / RTUS_BLOCK)),
                           dim3(RTUS_BLOCK), 0, s, first_ray, tof, n, n_rx, n_batch, hit, tof_hit);
    }
    return hipGetLastError();
}
