// rtus_tfm.hip — the consumers of a travel-time table: transmit focal laws and a total-focusing-
// method (TFM) delay-and-sum beamformer over full-matrix-capture (FMC) data.  NOT IN THE REFERENCE (it stops at the
// travel times, main_rt.py:497-504): checked against the NumPy restatement oracle/tfm_numpy.py on synthetic
// point-scatterer data.
//
// Unlike every other kernel of this library these two are bound by memory, not by VALU issue:
//   * the focal-law kernels stream the table: 8 B read and 8 B written per entry (apertures up to 512 elements: the
//     column lives in registers between the maximum and the subtraction; larger ones re-read it: 24 B) -> HBM roofline;
//   * rtus_tfm_kernel gathers two neighbouring fp32 samples per (tx, rx, focal point) from the A-scan of that pair:
//     8 B of L2 traffic per pair and focal point.  The FMC block (n_tx n_rx n_t 4 B: 34 MB at 64 x 64 x 2048) is read
//     from HBM about once per launch and then lives in L2 / Infinity Cache, so the bound is the L2 gather rate
//     (MI355X_MICROARCH.md "Indexed rows": 17-19 TB/s chip-wide for rows shared by every workgroup), not HBM.
#include "rtus_device.h"

// ---------------------------------------------------------------------------------------------- focal laws
// delays[e][f] = max_e' tt[e'][f] - tt[e][f]: what element e must wait so that all wavefronts reach f together.
// NaN (no ray path) is ignored by the maximum and stays NaN in the result; a column without any path is all NaN.
// One workgroup = 256 consecutive focal points x all elements, two passes over its strip (coalesced 2 KB rows).
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_focal_delays_kernel(const double* tt, int n_e, int n_f,
                                                                        double* delays)   // may alias tt (in place)
{
    const int f = blockIdx.x * RTUS_BLOCK + threadIdx.x;
    if (f >= n_f) return;
    const size_t nf = (size_t)n_f;
    double m = -INFINITY;
    int e = 0;
    for (; e + 4 <= n_e; e += 4) {                         // four loads in flight per lane
        const double a = tt[(size_t)e * nf + f], b = tt[(size_t)(e + 1) * nf + f], c = tt[(size_t)(e + 2) * nf + f],
                     d = tt[(size_t)(e + 3) * nf + f];
        m = fmax(fmax(m, a), fmax(b, fmax(c, d)));         // fmax ignores NaN operands
    }
    for (; e < n_e; ++e) m = fmax(m, tt[(size_t)e * nf + f]);
    m = (m == -INFINITY) ? NAN : m;                        // no element reaches this focal point
    for (e = 0; e < n_e; ++e) {
        const size_t o = (size_t)e * nf + f;
        delays[o] = m - tt[o];                             // NaN - x and x - NaN stay NaN
    }
}

// The same for apertures of up to 8 x RTUS_FD_ROWS elements, reading the table ONCE: a workgroup = 32 focal points x 8 row
// groups; a thread keeps its <= RTUS_FD_ROWS entries of one column in registers, the eight partial maxima of a column meet
// in LDS, and the thread subtracts and writes from its registers (in place is safe: a thread only writes what it has
// read).  16 B of HBM traffic per entry instead of 24 (measured on the 537 MB configs[2] table: the strip a workgroup of
// the two-pass kernel comes back to has long left the caches).  Lanes 0-31 of a wave are 32 neighbouring columns of one
// row, lanes 32-63 the same columns 1/8 of the aperture further down: two 256-byte segments per load.
#define RTUS_FD_ROWS 64
template <int ROWS>
__global__ __launch_bounds__(RTUS_BLOCK) void rtus_focal_delays_once_kernel(const double* tt, int n_e, int n_f, double* delays)
{
    __shared__ double part[8][32];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;             // column inside the tile, row group
    const int f_raw = blockIdx.x * 32 + c;
    const bool live = f_raw < n_f;
    const int f = live ? f_raw : n_f - 1;
    const int per = (n_e + 7) / 8;                                     // rows per group (<= ROWS)
    const int e0 = g * per;
    const size_t nf = (size_t)n_f;
    double v[ROWS];
    double m = -INFINITY;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int e = e0 + k;
        v[k] = (k < per && e < n_e) ? tt[(size_t)e * nf + f] : NAN;   // NaN: ignored by fmax, never stored
        m = fmax(m, v[k]);
    }
    part[g][c] = m;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) m = fmax(m, part[j][c]);
    m = (m == -INFINITY) ? NAN : m;                                    // no element reaches this focal point
    if (!live) return;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) {
        const int e = e0 + k;
        if (k < per && e < n_e) delays[(size_t)e * nf + f] = m - v[k];
    }
}

hipError_t rtus_launch_focal_delays(const double* tt, int n_e, int n_f, double* delays, hipStream_t s)
{
    const int per = (n_e + 7) / 8;
    const dim3 grid1((n_f + 31) / 32), block(RTUS_BLOCK);
    if (per <= 8) hipLaunchKernelGGL(rtus_focal_delays_once_kernel<8>, grid1, block, 0, s, tt, n_e, n_f, delays);
    else if (per <= 16) hipLaunchKernelGGL(rtus_focal_delays_once_kernel<16>, grid1, block, 0, s, tt, n_e, n_f, delays);
    else if (per <= 32) hipLaunchKernelGGL(rtus_focal_delays_once_kernel<32>, grid1, block, 0, s, tt, n_e, n_f, delays);
    else if (per <= RTUS_FD_ROWS) hipLaunchKernelGGL(rtus_focal_delays_once_kernel<RTUS_FD_ROWS>, grid1, block, 0, s, tt, n_e, n_f, delays);
    else hipLaunchKernelGGL(rtus_focal_delays_kernel, dim3((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK), block, 0, s, tt, n_e, n_f, delays);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- TFM delay-and-sum
// image[f] = sum over (tx, rx) of the A-scan fmc[tx][rx][.] linearly interpolated at the sample position
// (tt_tx[tx][f] + tt_rx[rx][f] - t0) fs.  Samples outside the record count as zero; pairs without a ray path (NaN
// travel time) contribute nothing.
//
// One workgroup = 256 focal points.  The receive delays (in samples, fp32) of a tile of RX_TILE elements sit in LDS,
// lane-major (lane l reads tau[rx][l]: conflict-free); the transmit delay of the current tx is a register.  The
// A-scan of a pair is addressed through a buffer descriptor whose base is wave-uniform (SGPRs) and whose extent is
// the record: the hardware's range check returns 0 for a sample index outside [0, n_t) — no compare / select in the
// inner loop — and the two neighbouring samples come in one 8-byte load.
#define RTUS_TFM_RX_TILE 64
typedef unsigned int tfm_u32x2 __attribute__((ext_vector_type(2)));

struct TfmArgs {
    const float* __restrict__ fmc;       // [n_tx][n_rx][n_t]
    const double* __restrict__ tt_tx;    // [n_tx][n_f]
    const double* __restrict__ tt_rx;    // [n_rx][n_f]
    float* __restrict__ image;           // [n_f]
    int n_tx, n_rx, n_t, n_f;
    double fs;                           // samples per second
    double half_t0s;                     // t0 * fs / 2: each of the pair's two delays carries half of the time origin
};

__device__ __forceinline__ float tfm_tau(double t, double fs, double half_t0s)
{
    // travel time -> half of the pair's sample position (formed in fp64, rounded once: 1e-4 of a sample at 4096
    // samples); no path -> far outside every record (finite: the sum of two of them must not become NaN or wrap an
    // integer conversion)
    // (... and so does every non-finite or absurd time: +inf would pass a NaN test and poison the pixel through floor(inf))
    const float v = (float)(t * fs - half_t0s);
    return fabsf(v) < 1.0e8f ? v : -1.0e8f;                  // NaN fails the compare
}

// Two neighbouring samples i, i + 1 of one A-scan (wave-uniform base) in one 8-byte load; an index outside the record
// (negative, huge, the no-path sentinel) is dropped by the descriptor's range check and reads as zeros.
#define RTUS_TFM_GROUP 16
// Edges, as oracle/tfm_numpy.py defines them: a position in [n_t - 1, n_t) interpolates towards a zero sample n_t (the second
// dword of the load is out of range by itself); a NEGATIVE position contributes nothing — index -1 must not wrap: its second
// dword would sit at byte offset 2^32, which the range check sees as 0 — so negative indices (as unsigned: >= 2^31) are
// clamped to one that is out of range with both dwords.  |i| stays below 2^30 (tfm_tau clamps to +-1e8 samples).
__device__ __forceinline__ tfm_u32x2 tfm_load2(const float* rec, int n_t, int i)
{
    const __amdgpu_buffer_rsrc_t q = __builtin_amdgcn_make_buffer_rsrc((void*)rec, 0, (unsigned)n_t * 4u, 0x00020000);
    return __builtin_amdgcn_raw_buffer_load_b64(q, min((unsigned)i, 0x3ffffff0u) * 4u, 0, 0);
}

__global__ __launch_bounds__(RTUS_BLOCK) void rtus_tfm_kernel(TfmArgs a)
{
    __shared__ float tau_rx[RTUS_TFM_RX_TILE][RTUS_BLOCK];           // 64 KB: 2 workgroups per CU; a 64-element receive
                                                                      // aperture is ONE tile: the transmit delays are read once
    // Workgroups go to the 8 XCDs round-robin, and each XCD has its own 4 MiB L2: with workgroup b on focal points
    // [256 b, 256 b + 256) every XCD sees focal points from all over the image and pulls (its window of) the WHOLE FMC block
    // through its L2.  XCD k takes a contiguous eighth of the focal points instead — neighbouring focal points share their
    // sample windows — so the FMC block is fetched about once, not once per XCD (measured: profiles/traffic_r03.json).
    const int nblk = gridDim.x, per = (nblk + 7) >> 3;
    int blk = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (nblk & 7) blk = blockIdx.x;                                   // (ragged grids keep the plain order)
    const int f_raw = blk * RTUS_BLOCK + threadIdx.x;
    const bool live = f_raw < a.n_f;
    const int f = live ? f_raw : a.n_f - 1;
    const size_t nf = (size_t)a.n_f;
    float acc = 0.0f;
    for (int r0 = 0; r0 < a.n_rx; r0 += RTUS_TFM_RX_TILE) {
        const int nr = min(RTUS_TFM_RX_TILE, a.n_rx - r0);
        __syncthreads();                                              // the previous tile is no longer read
        for (int r = 0; r < nr; ++r) tau_rx[r][threadIdx.x] = tfm_tau(a.tt_rx[(size_t)(r0 + r) * nf + f], a.fs, a.half_t0s);
        __syncthreads();
        // Sixteen receive elements per trip: 16 independent gathers in flight per lane, ALL issued before the first is
        // consumed (two explicit phases: left to itself the scheduler pairs each load with its use).  An image of
        // 256 x 256 focal points is 1024 waves — one per SIMD — so nothing but the wave's own loads hides the ~1 us a
        // gather takes: 4 in flight 441 us, 8: 282 us, 16: 214 us (192 us with the whole receive aperture in one tile), 32 (two
        // transmit elements at once): 218 us — from 16 on
        // the vector-memory address path binds (64 scattered 8-byte requests per wave-instruction, ~27 cycles each per CU).
        // one table for both legs and the whole receive aperture in this tile: the transmit delay IS a row of the tile
        const bool tx_in_tile = a.tt_tx == a.tt_rx && a.n_tx == a.n_rx && a.n_rx <= RTUS_TFM_RX_TILE;
        for (int tx = 0; tx < a.n_tx; ++tx) {
            const float tt = tx_in_tile ? tau_rx[tx][threadIdx.x] : tfm_tau(a.tt_tx[(size_t)tx * nf + f], a.fs, a.half_t0s);
            const float* rec = a.fmc + ((size_t)tx * a.n_rx + r0) * (size_t)a.n_t;   // wave-uniform
            int r = 0;
            for (; r + RTUS_TFM_GROUP <= nr; r += RTUS_TFM_GROUP) {
                tfm_u32x2 v[RTUS_TFM_GROUP];
                float w[RTUS_TFM_GROUP];
#pragma unroll
                for (int k = 0; k < RTUS_TFM_GROUP; ++k) {
                    const float s = tt + tau_rx[r + k][threadIdx.x];
                    const float fl = floorf(s);
                    w[k] = s - fl;
                    v[k] = tfm_load2(rec + (size_t)(r + k) * a.n_t, a.n_t, (int)fl);
                }
#pragma unroll
                for (int k = 0; k < RTUS_TFM_GROUP; ++k) {
                    const float v0 = __uint_as_float(v[k].x), v1 = __uint_as_float(v[k].y);
                    acc += fmaf(w[k], v1 - v0, v0);
                }
            }
            for (; r < nr; ++r) {                                     // receive elements past the last full group
                const float s = tt + tau_rx[r][threadIdx.x];
                const float fl = floorf(s);
                const tfm_u32x2 v = tfm_load2(rec + (size_t)r * a.n_t, a.n_t, (int)fl);
                acc += fmaf(s - fl, __uint_as_float(v.y) - __uint_as_float(v.x), __uint_as_float(v.x));
            }
        }
    }
    if (live) a.image[f] = acc;
}

hipError_t rtus_launch_tfm(const float* fmc, int n_tx, int n_rx, int n_t, double fs, double t0, const double* tt_tx,
                           const double* tt_rx, int n_f, float* image, hipStream_t s)
{
    TfmArgs a;
    a.fmc = fmc; a.tt_tx = tt_tx; a.tt_rx = tt_rx; a.image = image;
    a.n_tx = n_tx; a.n_rx = n_rx; a.n_t = n_t; a.n_f = n_f;
    a.fs = fs; a.half_t0s = 0.5 * t0 * fs;
    hipLaunchKernelGGL(rtus_tfm_kernel, dim3((n_f + RTUS_BLOCK - 1) / RTUS_BLOCK), dim3(RTUS_BLOCK), 0, s, a);
    return hipGetLastError();
}
