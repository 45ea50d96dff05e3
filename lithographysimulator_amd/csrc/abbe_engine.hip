// abbe_engine.hip -- Abbe source-point accumulation for MI355X (gfx950).
//
// Replaces the loop of abbeImage (reference imageformation.py:54-67) and
// calculateFFTAerial (imageformation.py:32-45).  See DESIGN.md for the derivation;
// in short, for every source point s with shift (dy,dx)
//
//     E_s[qy,qx] = sum_{iy,ix} P[iy-dy, ix-dx] M[iy,ix] w^((iy-c)(qy-c) + (ix-c)(qx-c)),
//     w = exp(+2 pi i / N), c = pn/2,   I += |E_s|^2
//
// is evaluated as two batched 1-D centred DFT passes with one global intermediate T:
//
//   x-pass  one line per (source point, row of the pupil support box): gathers
//           P*M on the fly (the zero padding to N is never materialised), transforms
//           along x, keeps the pn centred outputs, writes T in 4-column tiles;
//   y-pass  one workgroup per 4-column tile: transforms along y, squares, and keeps the
//           running sum over a whole batch of source points in registers, so the
//           intensity image is touched once per batch instead of once per source point.
//
// When no shifted copy of the pupil support wraps around the pn-grid (always true for
// sigma_out + pupil radius <= 2, i.e. every physical configuration) the roll is moved
// from P to M:  |sum P[i-d] M[i] w^(i q)| = |sum P[i'] M[i'+d] w^(i' q)|, so the non-zero
// window is the fixed support box of P (about pn/2 x pn/2) for every source point.
#include <hip/hip_runtime.h>

#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/litho_abbe.h"
#include "engine_common.hpp"
#include "engine_kernels.hpp"

namespace litho {

// ----------------------------------------------------------------------------------
// small helpers: twiddle table, planning, slab reduction
// ----------------------------------------------------------------------------------
__global__ void k_twiddle_table(float2* tab, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s, c;
    sincospi(2.0 * (double)n / (double)N, &s, &c);
    tab[n] = make_float2((float)c, (float)s);
}

// (plan words: abbe_plan.hpp)
static constexpr int BOX_ROWS_PER_BLOCK = 8;

// ----------------------------------------------------------------------------------
// The plan words in ONE gathering launch + one single-block finish (round 6; until then k_plan_init + k_pupil_box +
// k_shift_extents with atomics on the words + the runtime's copy kernel: four dependent small launches in front of the one host
// wait of an image -- at 256^2 a tenth of the image).  Box blocks: bounding box of the non-zero pupil samples over all planes and
// the supports on the edges of the natural box of the grid the engine will RUN at (e_lo, e_hi = c -+ pn/4, or -+ pe/4 of the
// padded grid of an embedded evaluation -- then possibly outside this array); extent blocks: extrema of the shift list, whose
// length comes from the host or -- asynchronous image path -- from the device word litho_source_compact left behind (count_dev,
// clamped to `S` = the capacity).  Every block writes its partial extrema to its own 16-int slot (no atomics, so nothing to initialise); the
// finish block folds them, stores the words on the device AND into the calling thread's pinned, device-mapped host buffer, and
// raises a sequence flag there, which the host polls (no copy kernel, no interrupt-driven wake-up for a wait of microseconds).
//   box blocks  [0, nb_box): BOX_ROWS_PER_BLOCK rows of one plane -> slot {rmin, rmax, cmin, cmax, e9, e10, e11, e12, corner}
//   extent blocks [nb_box, nb_box + nb_ext): grid-stride over the shift list -> slot {dymin, dymax, dxmin, dxmax}
// ----------------------------------------------------------------------------------
static constexpr int PLAN_PART = 16;
__global__ __launch_bounds__(256) void k_plan_gather(const float2* __restrict__ P, int pn, int e_lo, int e_hi, int nb_box_x, int nb_box,
                                                     const int* __restrict__ shifts, long long S, const int* __restrict__ count_dev,
                                                     int nb_ext, int* __restrict__ part)
{
    __shared__ int red[4][9];
    const int b = blockIdx.x, wv = threadIdx.x >> 6;
    int v[9] = {INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0};   // even: minima, odd: maxima, [8]: or
    if (b < nb_box) {
        const int row0 = (b % nb_box_x) * BOX_ROWS_PER_BLOCK;
        const float2* plane = P + (size_t)(b / nb_box_x) * pn * pn;
        for (int row = row0; row < min(pn, row0 + BOX_ROWS_PER_BLOCK); ++row) {
            const float2* r = plane + (size_t)row * pn;
            for (int j = threadIdx.x; j < pn; j += blockDim.x) {
                const float2 s = r[j];
                if (s.x != 0.f || s.y != 0.f) {
                    v[0] = min(v[0], row); v[1] = max(v[1], row); v[2] = min(v[2], j); v[3] = max(v[3], j);
                    const bool ecol = (j == e_hi || j == e_lo), erow = (row == e_hi || row == e_lo);
                    if (ecol) { v[4] = min(v[4], row); v[5] = max(v[5], row); }
                    if (erow) { v[6] = min(v[6], j); v[7] = max(v[7], j); }
                    if (ecol && erow) v[8] = 1;
                }
            }
        }
    } else {
        if (count_dev) {
            const long long c = *count_dev;
            S = c < S ? c : S;
        }
        for (long long i = (long long)(b - nb_box) * blockDim.x + threadIdx.x; i < S; i += (long long)nb_ext * blockDim.x) {
            const int dy = shifts[2 * i], dx = shifts[2 * i + 1];
            v[0] = min(v[0], dy); v[1] = max(v[1], dy); v[2] = min(v[2], dx); v[3] = max(v[3], dx);
        }
    }
    for (int k = 0; k < 9; ++k)
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(v[k], off);
            v[k] = k == 8 ? (v[k] | o) : (k & 1) ? max(v[k], o) : min(v[k], o);
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 9; ++k) red[wv][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 9) {
        const int k = threadIdx.x;
        int a = red[0][k];
        for (int i = 1; i < 4; ++i) a = k == 8 ? (a | red[i][k]) : (k & 1) ? max(a, red[i][k]) : min(a, red[i][k]);
        part[(size_t)b * PLAN_PART + k] = a;
    }
}
// host_words: the calling thread's pinned buffer as the device sees it (nullptr: device words only); [15] = the sequence flag
__global__ __launch_bounds__(256) void k_plan_finish(const int* __restrict__ part, int nb_box, int nb_ext, long long S,
                                                     const int* __restrict__ count_dev, int* __restrict__ plan,
                                                     volatile int* host_words, int seq)
{
    __shared__ int red[4][13];
    const int wv = threadIdx.x >> 6;
    // [0..3] box, [4..7] shift extents, [8..11] edge words (plan 9..12), [12] corner
    int v[13] = {INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, 0};
    for (int b = threadIdx.x; b < nb_box; b += blockDim.x) {
        const int* q = part + (size_t)b * PLAN_PART;
        v[0] = min(v[0], q[0]); v[1] = max(v[1], q[1]); v[2] = min(v[2], q[2]); v[3] = max(v[3], q[3]);
        v[8] = min(v[8], q[4]); v[9] = max(v[9], q[5]); v[10] = min(v[10], q[6]); v[11] = max(v[11], q[7]);
        v[12] |= q[8];
    }
    for (int b = threadIdx.x; b < nb_ext; b += blockDim.x) {
        const int* q = part + (size_t)(nb_box + b) * PLAN_PART;
        v[4] = min(v[4], q[0]); v[5] = max(v[5], q[1]); v[6] = min(v[6], q[2]); v[7] = max(v[7], q[3]);
    }
    for (int k = 0; k < 13; ++k)
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(v[k], off);
            v[k] = k == 12 ? (v[k] | o) : (k & 1) ? max(v[k], o) : min(v[k], o);
        }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 13; ++k) red[wv][k] = v[k];
    __syncthreads();
    // thread k < 14 folds and publishes plan word k (the stores into host memory are PCIe writes: one thread doing all fourteen
    // in a row took 9 us); words: [0..7] as gathered, [8] count, [9..12] = gathered [8..11], [13] = corner flag
    const int k = threadIdx.x;
    if (k < PLAN_WORDS) {
        int word;
        if (k == 8) {
            if (count_dev) {
                const long long c = *count_dev;
                S = c < S ? c : S;
            }
            word = (int)S;
        } else {
            const int src = k < 8 ? k : k - 1;                    // (edge words: "none seen" is (INT_MAX, INT_MIN))
            int a = red[0][src];
            for (int i = 1; i < 4; ++i) a = src == 12 ? (a | red[i][src]) : (src & 1) ? max(a, red[i][src]) : min(a, red[i][src]);
            word = src == 12 ? (a ? 1 : 0) : a;
        }
        plan[k] = word;
        if (host_words) {
            host_words[k] = word;
            __threadfence_system();
        }
    }
    __syncthreads();
    if (k == 0 && host_words) host_words[15] = seq;
}

// out[p][qy][qx] += sum_g slab[p * gstride + g][qx][qy]   (TS x TS tiles through LDS; blockIdx.z = plane p).  TS * TS threads per
// tile, one element each: small images fold up to 64 slabs into a few dozen tiles (256^2: 64 tiles of 32 x 32), and with 256 threads
// walking four rows each that took 38 us per image; fixed summation order (deterministic).  TS = 16 where 32 x 32 tiles would
// not give every CU a workgroup (round 5: 256^2 folds 64 slabs in 64 workgroups otherwise -- 20 us of a 500 us image).
template <int TS, int GS>
__global__ __launch_bounds__(TS * TS * GS) void k_slab_reduce(const float* __restrict__ slab, float* __restrict__ out, int pn, int ldq, int G,
                                                              int gstride)
{
    // GS slab lanes per tile element: lane z sums the slabs z, z + GS, ... in order, the lanes are added in order (deterministic).
    // Small images fold up to 64 slabs per element: with one thread walking all of them the 16.8 MB of config 1's fold took
    // 20 us (0.85 TB/s, latency-bound); four lanes per element keep four times the loads in flight.
    __shared__ float tile[GS][TS][TS + 1];
    slab += (size_t)blockIdx.z * gstride * ldq * pn;
    out += (size_t)blockIdx.z * pn * pn;
    const int qx0 = blockIdx.x * TS, qy0 = blockIdx.y * TS;
    const int e = threadIdx.x % (TS * TS), z = threadIdx.x / (TS * TS);
    const int tx = e % TS, ty = e / TS;
    {
        const int qx = qx0 + ty, qy = qy0 + tx;
        float v = 0.f;
        if (qx < pn && qy < pn)
            for (int gidx = z; gidx < G; gidx += GS) v += slab[((size_t)gidx * ldq + qx) * pn + qy];
        tile[z][ty][tx] = v;
    }
    __syncthreads();
    if (z != 0) return;
    const int qy = qy0 + ty, qx = qx0 + tx;
    if (qx < pn && qy < pn) {
        float v = tile[0][tx][ty];
        for (int k = 1; k < GS; ++k) v += tile[k][tx][ty];
        out[(size_t)qy * pn + qx] += v;
    }
}
static hipError_t launch_slab_reduce(const float* slab, float* out, int pn, int ldq, int planes, int G, int gstride, hipStream_t st)
{
    const int t32 = (pn + 31) / 32;
    if ((long long)t32 * t32 * planes >= device_cus())
        hipLaunchKernelGGL((k_slab_reduce<32, 1>), dim3(t32, t32, planes), dim3(1024), 0, st, slab, out, pn, ldq, G, gstride);
    else if (G >= 8)
        hipLaunchKernelGGL((k_slab_reduce<16, 4>), dim3((pn + 15) / 16, (pn + 15) / 16, planes), dim3(1024), 0, st, slab, out, pn, ldq, G, gstride);
    else
        hipLaunchKernelGGL((k_slab_reduce<16, 1>), dim3((pn + 15) / 16, (pn + 15) / 16, planes), dim3(256), 0, st, slab, out, pn, ldq, G, gstride);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------
// Coarse-grid path: the Nyquist-line coefficients.
// The intensity I(q) = sum_s |E_s(q)|^2 has Fourier coefficients C[kappa], |kappa| <= pn/2 (E_s lives on |k| <= pn/4), so
// pn samples per period -- the coarse grid q = 2 v -- determine it up to the coefficients on the lines kappa_x = +-pn/2
// and kappa_y = +-pn/2, which alias onto each other there.  Those come only from products of the two opposite edges of
// the pupil's support box:  Gx[kappa] = sum_s sum_ky A_s[ky, +h] conj(A_s[ky - kappa, -h]),  h = pn/4, and Gy likewise
// with rows; A_s[k] = P[k] M[k + shift_s].  k_nyquist_edges accumulates them over a chunk of source points per
// workgroup (deterministic partial sums), k_nyquist_profiles sums the partials and evaluates
// Gprof[q] = sum_kappa Gx[kappa] w_N^(kappa q) (and Hprof from Gy), k_nyquist_apply adds the missing part
//   dI[qy, qx] = Re(2 i^qx Gprof[qy]) for odd qx  +  Re(2 i^qy Hprof[qx]) for odd qy        (centred q)
// to the band-limited interpolation of the coarse image.  (Derivation and a numpy check: DESIGN.md.)
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_nyquist_edges(const float2* __restrict__ P, const float2* __restrict__ M,
                                                       const int* __restrict__ shifts, long long S, EdgeGeom eg,
                                                       float2* __restrict__ partial)
{
    __shared__ float2 su[2][EDGE_MAX], sv[2][EDGE_MAX];
    const int t = threadIdx.x;
    const int edge = t >> 7, lt = t & 127;                     // threads 0..127: edge 0 (kappa and loads), 128..255: edge 1
    const int len = eg.len[edge], lo = eg.lo[edge];
    const long long per = (S + gridDim.x - 1) / gridDim.x;
    const long long s0 = (long long)blockIdx.x * per, s1 = min(S, s0 + per);
    // this thread owns kappa = lt - (len - 1) and kappa + 128 ... : at most 2 len - 1 <= 255 values -> two per thread
    float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
    const int k0 = lt - (len - 1), k1 = k0 + 128;
    float2 pu = make_float2(0.f, 0.f), pvv = make_float2(0.f, 0.f);
    if (lt < len) {                                             // pupil samples on the two opposite edges (fixed per thread)
        const int i = lo + lt;
        pu = edge == 0 ? P[(size_t)i * eg.pn + eg.c + eg.h] : P[(size_t)(eg.c + eg.h) * eg.pn + i];
        pvv = edge == 0 ? P[(size_t)i * eg.pn + eg.c - eg.h] : P[(size_t)(eg.c - eg.h) * eg.pn + i];
    }
    for (long long s = s0; s < s1; ++s) {
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        __syncthreads();
        if (lt < len) {
            // A zero pupil sample means "outside the support": the shifted mask sample there may lie outside the
            // grid (the no-wrap test only covers the support box), and 0 * Inf would be a NaN -- no load, exact zero.
            const int i = lo + lt;
            float2 mu = make_float2(0.f, 0.f), mv = make_float2(0.f, 0.f);
            if (pu.x != 0.f || pu.y != 0.f)
                mu = edge == 0 ? M[(size_t)(i + dy) * eg.pn + eg.c + eg.h + dx] : M[(size_t)(eg.c + eg.h + dy) * eg.pn + i + dx];
            if (pvv.x != 0.f || pvv.y != 0.f)
                mv = edge == 0 ? M[(size_t)(i + dy) * eg.pn + eg.c - eg.h + dx] : M[(size_t)(eg.c - eg.h + dy) * eg.pn + i + dx];
            su[edge][lt] = cmul(pu, mu);
            sv[edge][lt] = cmul(pvv, mv);
        }
        __syncthreads();
        for (int i = 0; i < len; ++i) {                        // sum_i u[i] conj(v[i - kappa])
            const float2 u = su[edge][i];
            const int j0 = i - k0, j1 = i - k1;
            if (j0 >= 0 && j0 < len) { const float2 v = sv[edge][j0]; acc0.x += u.x * v.x + u.y * v.y; acc0.y += u.y * v.x - u.x * v.y; }
            if (j1 >= 0 && j1 < len) { const float2 v = sv[edge][j1]; acc1.x += u.x * v.x + u.y * v.y; acc1.y += u.y * v.x - u.x * v.y; }
        }
    }
    float2* out = partial + ((size_t)blockIdx.x * 2 + edge) * (2 * EDGE_MAX);
    out[lt] = acc0;                                             // index = kappa + (len - 1)
    out[lt + 128] = acc1;
}

// gam layout: [GAM_PARTIAL partials][Gamma: 2 edges x 2 EDGE_MAX][profiles: 2 x pn]
// Sum of the per-chunk partial edge sums, in a fixed order (deterministic): a 256-thread block owns 8 consecutive kappa
// indices x 32 chunk lanes (a chunk's 8 entries are one 64-byte run), every thread adds chunks / 32 partials, then the 32
// lanes of an index fold by shuffles and one LDS step.  (Round 3's one-thread-per-index loop over up to 1024 chunks took
// 268 us per image -- most of the coarse grid's fixed cost at 256^2; this one takes a few.)
static constexpr int NYQ_RED_IDX = 8;
__global__ __launch_bounds__(256) void k_nyquist_reduce(float2* __restrict__ gam, int chunks)
{
    __shared__ float2 part[4][NYQ_RED_IDX];
    const int il = threadIdx.x & (NYQ_RED_IDX - 1), cl = threadIdx.x / NYQ_RED_IDX;      // chunk lane 0 .. 31
    const int idx = blockIdx.x * NYQ_RED_IDX + il;              // edge * 2 EDGE_MAX + kappa index
    float2 a = make_float2(0.f, 0.f);
    for (int ch = cl; ch < chunks; ch += 256 / NYQ_RED_IDX) {
        const float2 v = gam[(size_t)ch * (2 * 2 * EDGE_MAX) + idx];
        a.x += v.x; a.y += v.y;
    }
    for (int off = NYQ_RED_IDX; off < 64; off <<= 1) { a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < NYQ_RED_IDX) part[wv][il] = a;
    __syncthreads();
    if (threadIdx.x < NYQ_RED_IDX) {
        float2 t = part[0][il];
        for (int w = 1; w < 4; ++w) { t.x += part[w][il].x; t.y += part[w][il].y; }
        gam[GAM_PARTIAL + idx] = t;
    }
}

__global__ void k_nyquist_profiles(float2* __restrict__ gam, const float2* __restrict__ twtab, EdgeGeom eg, int N)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;        // pixel index; centred coordinate q - c
    const int edge = blockIdx.y;
    if (q >= eg.pn) return;
    const float2* G = gam + GAM_PARTIAL + (size_t)edge * (2 * EDGE_MAX);
    const int len = eg.len[edge], qc = q - eg.c;
    float2 a = make_float2(0.f, 0.f);
    for (int i = 0; i < 2 * len - 1; ++i) {
        const int kappa = i - (len - 1);
        const float2 w = twtab[(unsigned)(kappa * qc) & (unsigned)(N - 1)];   // w_N^(kappa q), two's-complement wrap
        const float2 gk = G[i];
        a.x += gk.x * w.x - gk.y * w.y;
        a.y += gk.x * w.y + gk.y * w.x;
    }
    gam[GAM_PARTIAL + 2 * 2 * EDGE_MAX + (size_t)edge * eg.pn + q] = a;
}

__global__ void k_nyquist_apply(float* __restrict__ img, const float2* __restrict__ gam, int pn)
{
    const int qx = blockIdx.x * blockDim.x + threadIdx.x, qy = blockIdx.y;
    if (qx >= pn) return;
    const int c = pn / 2, cx = qx - c, cy = qy - c;
    const float2* Gp = gam + GAM_PARTIAL + 2 * 2 * EDGE_MAX;     // Gprof[qy] (edge 0), Hprof[qx] (edge 1)
    float d = 0.f;
    if (cx & 1) {                                               // Re(2 i^cx G): i^1 = i -> -2 Im, i^3 = -i -> +2 Im
        const float2 G = Gp[qy];
        d += ((cx & 3) == 1 ? -2.f : 2.f) * G.y;
    }
    if (cy & 1) {
        const float2 H = Gp[pn + qx];
        d += ((cy & 3) == 1 ? -2.f : 2.f) * H.y;
    }
    if (d != 0.f) img[(size_t)qy * pn + qx] += d;
}

// ----------------------------------------------------------------------------------
// Splitting a source list whose shifts wrap the pupil around the grid for SOME of its points (shifted, off-axis sources:
// LightSource(shiftX, shiftY), lightsource.py:5): the points that do not wrap keep every fast path (pruned box, wave kernels,
// coarse grid, embedding), only the wrapping ones need the general one (roll kept on P, modular gather, full window: 4x the
// time per point).  Stable, deterministic two-list compaction in three small kernels; 1024 points per block, 4 consecutive
// points per thread.
// ----------------------------------------------------------------------------------
struct SplitBox { int r0, r1, c0, c1, pn; };
__device__ __forceinline__ bool shift_wraps(int dy, int dx, const SplitBox& b)
{
    return b.r0 + dy < 0 || b.r1 + dy > b.pn - 1 || b.c0 + dx < 0 || b.c1 + dx > b.pn - 1;
}
__global__ __launch_bounds__(256) void k_split_count(const int* __restrict__ shifts, long long S, SplitBox b, int* __restrict__ counts)
{
    __shared__ int red[4];
    const long long base = (long long)blockIdx.x * SPLIT_PER_BLOCK + 4 * threadIdx.x;
    int n = 0;
    for (int i = 0; i < 4; ++i)
        if (base + i < S && !shift_wraps(shifts[2 * (base + i)], shifts[2 * (base + i) + 1], b)) ++n;
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of the block counts (one block walks them; at most pn^2 / 1024 entries) + the two totals and empty extents
__global__ __launch_bounds__(1024) void k_split_scan(int* __restrict__ counts, int nblocks, long long S, int* __restrict__ words)
{
    __shared__ int part[1024];
    const int per = (nblocks + 1023) / 1024, lo = threadIdx.x * per, hi = min(nblocks, lo + per);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; }
        words[0] = run; words[1] = (int)(S - run);
        for (int i = 0; i < 2; ++i) { words[2 + 4 * i] = INT_MAX; words[3 + 4 * i] = INT_MIN; words[4 + 4 * i] = INT_MAX; words[5 + 4 * i] = INT_MIN; }
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { const int v = counts[i]; counts[i] = run; run += v; }
}
__global__ __launch_bounds__(256) void k_split_write(const int* __restrict__ shifts, long long S, SplitBox b, const int* __restrict__ offsets,
                                                     int* __restrict__ list_a, int* __restrict__ list_b, int* __restrict__ words)
{
    __shared__ int scan[256];
    const long long block0 = (long long)blockIdx.x * SPLIT_PER_BLOCK, base = block0 + 4 * threadIdx.x;
    int dy[4], dx[4];
    bool ok[4], wr[4];
    int n = 0;
    for (int i = 0; i < 4; ++i) {
        ok[i] = base + i < S;
        dy[i] = ok[i] ? shifts[2 * (base + i)] : 0;
        dx[i] = ok[i] ? shifts[2 * (base + i) + 1] : 0;
        wr[i] = ok[i] && shift_wraps(dy[i], dx[i], b);
        if (ok[i] && !wr[i]) ++n;
    }
    scan[threadIdx.x] = n;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {                   // inclusive Hillis-Steele scan of the per-thread counts
        const int v = threadIdx.x >= off ? scan[threadIdx.x - off] : 0;
        __syncthreads();
        scan[threadIdx.x] += v;
        __syncthreads();
    }
    long long ia = (long long)offsets[blockIdx.x] + scan[threadIdx.x] - n;          // rank among the non-wrapping points
    long long ib = base - ia;                                                       // ... and among the wrapping ones (this thread's first point)
    int e[8] = {INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN};
    for (int i = 0; i < 4; ++i) {
        if (!ok[i]) continue;
        int* dst = wr[i] ? list_b + 2 * ib++ : list_a + 2 * ia++;
        dst[0] = dy[i]; dst[1] = dx[i];
        int* x = e + (wr[i] ? 4 : 0);
        x[0] = min(x[0], dy[i]); x[1] = max(x[1], dy[i]); x[2] = min(x[2], dx[i]); x[3] = max(x[3], dx[i]);
    }
    for (int k = 0; k < 8; ++k)
        for (int off = 32; off > 0; off >>= 1) e[k] = (k & 1) ? max(e[k], __shfl_xor(e[k], off)) : min(e[k], __shfl_xor(e[k], off));
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 8; ++k) {
            if (k & 1) { if (e[k] != INT_MIN) atomicMax(&words[2 + k], e[k]); }
            else if (e[k] != INT_MAX) atomicMin(&words[2 + k], e[k]);
        }
}

// ----------------------------------------------------------------------------------
// host side.  Sizes, the workspace layout, the launch planner and the call-level decisions live in abbe_plan.hpp (HIP-free,
// shared with the dry-run entry point and its CPU sweep); here: the pointers, the launches, the read-backs.
// ----------------------------------------------------------------------------------
struct Workspace {
    int* plan;          // 64 ints
    float2* twtab;      // N
    float2* twtab2;     // pn: table of the coarse-grid transforms
    float* slab;        // G_MAX * nt*4 * pn
    float* ic;          // coarse-grid intensity of the planes in flight: COARSE_PLANES * pn * pn
    float2* chat;       // its spectrum: pn * pn
    float2* gam;        // Nyquist-line work area: partial sums, Gamma, profiles (GAM_FLOAT2 entries)
    float2* T;          // remainder
    size_t t_bytes;
};

static bool carve(void* ws, size_t bytes, int pn, int N, Workspace& w)
{
    const WsLayout l = ws_layout(pn, N);
    if (!ws || bytes < l.total) return false;
    unsigned char* p = (unsigned char*)ws;
    w.plan = (int*)(p + l.plan.off);
    w.twtab = (float2*)(p + l.twtab.off);
    w.twtab2 = (float2*)(p + l.twtab2.off);
    w.slab = (float*)(p + l.slab.off);
    w.ic = (float*)(p + l.ic.off);
    w.chat = (float2*)(p + l.chat.off);
    w.gam = (float2*)(p + l.gam.off);
    w.T = (float2*)(p + l.T.off);
    w.t_bytes = l.T.bytes;
    return true;
}

static thread_local int64_t g_last_plan[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's
// roofline leg).  Off by default: the events serialise nothing but cost host time.
static thread_local int g_profiling = 0;
static thread_local double g_profile[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // x ms, x launches, x points, y ms, y launches, y points, -, -

const SizeOps* size_ops_4(); const SizeOps* size_ops_5(); const SizeOps* size_ops_6(); const SizeOps* size_ops_7();
const SizeOps* size_ops_8(); const SizeOps* size_ops_9(); const SizeOps* size_ops_10(); const SizeOps* size_ops_11();
const SizeOps* size_ops_12(); const SizeOps* size_ops_13(); const SizeOps* size_ops_14();

const SizeOps* size_ops(int log2n)
{
    switch (log2n) {
        case 4: return size_ops_4();   case 5: return size_ops_5();   case 6: return size_ops_6();
        case 7: return size_ops_7();   case 8: return size_ops_8();   case 9: return size_ops_9();
        case 10: return size_ops_10(); case 11: return size_ops_11(); case 12: return size_ops_12();
        case 13: return size_ops_13(); case 14: return size_ops_14();
        default: return nullptr;
    }
}

// A few dozen ints of PINNED host memory per calling thread for the read-backs (plan words, the split's ten words).  Into
// pageable memory (a stack array) the runtime stages the 56 bytes through a blit kernel and a bounce buffer: 44 us on the
// device timeline of a config-1 image (rocprofv3: __amd_rocclr_copyBuffer), a twelfth of the whole image; nullptr if the
// allocation fails (then the stack array is used as before).  One buffer per calling thread, freed when the thread exits.
struct PinnedWords {
    int* p = nullptr;          // host address
    int* dp = nullptr;         // the same buffer as kernels see it (device-mapped, coherent), nullptr if unavailable
    int dev = -1;              // the device dp was obtained on: calls on another device of the same thread take the copy route
    int seq = 0;               // sequence number of the last k_plan_finish that was asked to publish here
    bool tried = false;
    ~PinnedWords() { if (p) (void)hipHostFree(p); }          // thread exit: a host application's short-lived worker threads do not leak
};
static PinnedWords& pinned_state()
{
    static thread_local PinnedWords pw;
    if (!pw.tried) {
        pw.tried = true;
        // the first call of a thread may come while that thread captures a stream: an allocation is not a capturable
        // operation, so it is made under the relaxed capture mode (and does not invalidate the capture)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool swapped = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        void* q = nullptr;
        if (hipHostMalloc(&q, 64 * sizeof(int), hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess) {
            pw.p = (int*)q;
            memset(q, 0, 64 * sizeof(int));
            void* d = nullptr;
            // LITHO_ABBE_NO_MAPPED_PLAN=1 (tests): behave as if the buffer could not be mapped -- the copy + stream-wait fallback
            const char* off = getenv("LITHO_ABBE_NO_MAPPED_PLAN");
            if (off && off[0] == '1') d = nullptr;
            else if (hipHostGetDevicePointer(&d, q, 0) != hipSuccess) { d = nullptr; (void)hipGetLastError(); }
            pw.dp = (int*)d;
            if (hipGetDevice(&pw.dev) != hipSuccess) { pw.dev = -1; pw.dp = nullptr; (void)hipGetLastError(); }
        } else {
            (void)hipGetLastError();
        }
        if (swapped) (void)hipThreadExchangeStreamCaptureMode(&mode);
    }
    return pw;
}
static int* pinned_words() { return pinned_state().p; }
// n ints from the device to `host` (one small synchronising copy)
static int read_words(const int* dev, int* host, int n, hipStream_t st)
{
    int* pin = pinned_words();
    int* dst = (pin && n <= 64) ? pin : host;
    HIP_TRY(hipMemcpyAsync(dst, dev, n * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (dst != host) memcpy(host, dst, n * sizeof(int));
    return LITHO_OK;
}
// Reads the plan words back.
static int read_plan(const Workspace& w, int host[PLAN_WORDS], hipStream_t st) { return read_words(w.plan, host, PLAN_WORDS, st); }

// The plan words of a call: one gathering launch over the pupil stack and the shift list, one single-block finish that also
// publishes the words into this thread's pinned buffer, and the host's wait for its sequence flag.  The partial slots live in
// the head of T (free until the source-point loop starts).
static int gather_plan(const Workspace& w, const float2* P, int planes, int pn, int pe, const int* shifts, int64_t S,
                       const int* count_dev, int pl[PLAN_WORDS], hipStream_t st)
{
    const int nb_box_x = (pn + BOX_ROWS_PER_BLOCK - 1) / BOX_ROWS_PER_BLOCK, nb_box = nb_box_x * planes;
    int nb_ext = (int)((S + 1023) / 1024);
    nb_ext = nb_ext < 1 ? 1 : (nb_ext > 256 ? 256 : nb_ext);
    if ((size_t)(nb_box + nb_ext) * PLAN_PART * sizeof(int) > w.t_bytes) return LITHO_E_WORKSPACE;
    int* part = (int*)w.T;
    PinnedWords& pw = pinned_state();
    int cur = -2;
    int* const dp = (pw.dp && hipGetDevice(&cur) == hipSuccess && cur == pw.dev) ? pw.dp : nullptr;
    const int seq = dp ? (pw.seq = pw.seq == INT_MAX ? 1 : pw.seq + 1) : 0;
    hipLaunchKernelGGL(k_plan_gather, dim3(nb_box + nb_ext), dim3(256), 0, st, P, pn, pn / 2 - pe / 4, pn / 2 + pe / 4, nb_box_x, nb_box,
                       shifts, (long long)S, count_dev, nb_ext, part);
    hipLaunchKernelGGL(k_plan_finish, dim3(1), dim3(256), 0, st, part, nb_box, nb_ext, (long long)S, count_dev, w.plan,
                       (volatile int*)dp, seq);
    HIP_TRY(hipGetLastError());
    if (dp) {
        // Poll the flag for a while (the wait is a few microseconds when the stream is otherwise idle -- an image sequence --
        // and an interrupt-driven wake-up costs more than that); with a long queue in front, sleep in the runtime instead.
        volatile int* hw = pw.p;
        bool seen = false;
        for (int spin = 0; spin < 50000 && !(seen = (hw[15] == seq)); ++spin) __builtin_ia32_pause();      // ~ 2 ms at most
        if (!seen) {
            HIP_TRY(hipStreamSynchronize(st));
            seen = hw[15] == seq;
        }
        if (seen) {
            for (int k = 0; k < PLAN_WORDS; ++k) pl[k] = hw[k];
            return LITHO_OK;
        }
        // (a mapping the device could not write: fall through to the copy)
    }
    return read_plan(w, pl, st);
}

// HIP events of one profiled call; destroyed on every exit path.
struct Mark { hipEvent_t ev; int kind; int items; };   // kind: -1 start, 0 after an x-pass, 1 after a y-pass
struct MarkList {
    static constexpr size_t MAX = 8192;                // = 4096 launch pairs per call
    std::vector<Mark> v;
    bool on;
    hipStream_t st;
    MarkList(bool on_, hipStream_t st_) : on(on_), st(st_) {}
    ~MarkList() { for (auto& m : v) (void)hipEventDestroy(m.ev); }
    void add(int kind, int items)
    {
        if (!on || v.size() >= MAX) return;
        Mark m{nullptr, kind, items};
        if (hipEventCreate(&m.ev) != hipSuccess) return;
        (void)hipEventRecord(m.ev, st);
        v.push_back(m);
    }
};

// Zero fill by a kernel of ours, not hipMemsetAsync: a memset node captured into a HIP graph is not replayed correctly
// on this stack (ROCm 7.2: the second replay after the buffer's contents changed leaves stale slabs -- 16,384 wrong
// pixels per 256^2 image, tests/test_gpu_abbe.py::test_planned_call_is_capturable_in_a_hip_graph), and the planned call
// is meant to be capturable.  Same cost as the memset (12 us for 16.8 MB).  Sizes and addresses are multiples of 4.
__global__ void k_zero_words(unsigned* __restrict__ p, size_t n4)
{
    const size_t n16 = n4 / 4;
    uint4* q = reinterpret_cast<uint4*>(p);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) q[i] = uint4{0, 0, 0, 0};
    if (blockIdx.x == 0 && threadIdx.x < (n4 & 3)) p[n16 * 4 + threadIdx.x] = 0u;
}
__global__ void k_zero_words_unaligned(unsigned* __restrict__ p, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static hipError_t zero_async(void* p, size_t bytes, hipStream_t st)
{
    if ((bytes & 3) || ((uintptr_t)p & 3)) return hipMemsetAsync(p, 0, bytes, st);      // (never the case here)
    const size_t n4 = bytes / 4;
    const unsigned blocks = (unsigned)((n4 / 4 + 255) / 256 < 2048 ? ((n4 / 4 + 255) / 256 ? (n4 / 4 + 255) / 256 : 1) : 2048);
    if ((uintptr_t)p & 15) hipLaunchKernelGGL(k_zero_words_unaligned, dim3(blocks), dim3(256), 0, st, (unsigned*)p, n4);
    else hipLaunchKernelGGL(k_zero_words, dim3(blocks), dim3(256), 0, st, (unsigned*)p, n4);
    return hipGetLastError();
}

// Slabs [p * G, p * G + used) of every plane p in flight (the others are never written: k_ypass_rect<.., 2>)
static hipError_t zero_slabs(float* slab, int pc, int G, int used, size_t slab_plane, hipStream_t st)
{
    if (used == G) return zero_async(slab, (size_t)pc * G * slab_plane * sizeof(float), st);
    for (int p = 0; p < pc; ++p) {
        const hipError_t e = zero_async(slab + (size_t)p * G * slab_plane, (size_t)used * slab_plane * sizeof(float), st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// One chunk of planes (pc <= pp.PC pupils starting at Pc) over the whole source list: slabs zeroed, x-pass / y-pass
// launch pairs batch by batch, slabs reduced INTO dst[0 .. pc) (each pn x pn, accumulated).
// zero_dst: dst is scratch of the call (the coarse image) and starts at zero.
// The zero fills (slabs, and dst when asked) are issued BEHIND the first batch's x-pass, which touches neither: right after the
// planning read-back the device is idle and the host is the bottleneck -- every small launch in front of the first big kernel
// is a few microseconds of idle device (config 1: 26 us of a 500 us image, profiles/r06_cfg1_timeline.txt); behind it they
// are issued while the x-pass runs.
static int accumulate_chunk(const AbbePlan& pp, const SizeOps* ops, const Workspace& w, const float2* twtab,
                            const float2* M, const float2* Pc, int pc, const int* shifts, int64_t S, int pn, float* dst,
                            bool zero_dst, hipStream_t st, MarkList& marks, int64_t& nx)
{
    const PassGeom& g = pp.g;
    const int variant = pp.variant, G = pp.G, xchunk = pp.xchunk;
    const int64_t bs = pp.bs;
    const size_t slab_plane = (size_t)g.nt * 4 * pn, plane_elems = (size_t)pn * pn;
    bool fresh = true;                                     // start a new timing interval after memset / reduce
    int since_flush = 0;
    for (int64_t s0 = 0; s0 < S; s0 += bs) {
        const int nb = (int)((S - s0 < bs) ? (S - s0) : bs);
        const int* sh = shifts + 2 * s0;
        if (fresh) { marks.add(-1, 0); fresh = false; }
        // ---- x-pass: T item (plane q of the chunk, point s) = q * nb + s
        for (int q = 0; q < pc;) {
            const float2* Pq = Pc + (size_t)q * plane_elems;
            float2* Tq = w.T + (size_t)q * nb * g.t_point;
            int np = 1;
            if (pp.rect_x) {
                HIP_TRY(ops->xpass_rect(Pq, M, sh, Tq, twtab, g, nb, xchunk, st));
            } else if (pp.split_x) {
                HIP_TRY(ops->xpass_split(Pq, M, sh, Tq, twtab, g, nb, xchunk, st));
            } else if (pp.fused_x) {
                np = variant == 0 ? 1 : (pc - q >= 4) ? 4 : (pc - q >= 2 ? 2 : 1);
                HIP_TRY(ops->xpass_abbe(variant, np, Pq, M, sh, Tq, twtab, g, nb, xchunk, st));
            } else if (pp.general) {
                AbbeLoader ld{Pq, M, sh, nullptr, nullptr, 0, 0};
                HIP_TRY(ops->xpass_general(ld, Tq, twtab, g, nb, st));
            } else if (variant >= 0) {
                HIP_TRY(ops->xpass_w64(Pq, M, sh, Tq, twtab, g, nb, st));
            } else {
                HIP_TRY(ops->xpass_abbe(-1, 1, Pq, M, sh, Tq, twtab, g, nb, xchunk, st));
            }
            q += np;
        }
        marks.add(0, nb * pc);
        if (s0 == 0) {
            HIP_TRY(zero_slabs(w.slab, pc, G, pp.slabs, slab_plane, st));
            if (zero_dst) HIP_TRY(zero_async(dst, (size_t)pc * plane_elems * sizeof(float), st));
            marks.add(-1, 0);                              // (the fills are not y-pass time)
        }
        // ---- y-pass: every plane of the chunk, G groups each (fewer when the batch is shorter than G)
        // (a short tail batch, nb < G: as many groups as points -- but never more than the slabs this chunk zeroes and folds)
        int Geff = nb < G ? nb : G;
        if (pp.slabs < G && (Geff & 1) && Geff > pp.slabs) Geff = pp.slabs;
        if (pp.wave_y) HIP_TRY(ops->ypass_w64(w.T, w.slab, twtab, g, nb, pc, Geff, G, st));
        else HIP_TRY(ops->ypass_acc(variant, w.T, w.slab, twtab, g, nb, pc, Geff, G, st));
        marks.add(1, nb * pc);
        ++nx;
        // Two-level summation: the slabs are folded into dst every SLAB_FLUSH_BATCHES batches, so no fp32 running
        // sum ever takes more than a few hundred additions (198,108 points at 2048^2, error of the image against a
        // float64 sum of short runs: 5.4e-6 of the maximum with one slab sum over all 16,509 batches, see
        // scripts/accum_error_probe.py).  Costs one k_slab_reduce + memset per 64 launch pairs (< 0.5 %).
        if (++since_flush == SLAB_FLUSH_BATCHES && s0 + bs < S) {
            HIP_TRY(launch_slab_reduce(w.slab, dst, pn, g.nt * 4, pc, pp.slabs, G, st));
            HIP_TRY(zero_slabs(w.slab, pc, G, pp.slabs, slab_plane, st));
            since_flush = 0;
            fresh = true;
        }
    }
    HIP_TRY(launch_slab_reduce(w.slab, dst, pn, g.nt * 4, pc, pp.slabs, G, st));
    return LITHO_OK;
}

// Coarse-grid reconstruction of ONE plane: out += I(q) on the fine grid, from the coarse image ic (pn x pn samples
// at q = 2 v over the whole period) and the exact Nyquist-line coefficients.
//   1. Chat = centred forward DFT2 of ic                    (pn-point transforms: ops_c / twtab2)
//   2. out += Re( sum_kappa Chat[kappa] w_N^(kappa q) ) / pn^2   (N-point zoom transforms: ops / twtab)
//   3. Gx, Gy from the box edges of P over the whole source list; out += dI (k_nyquist_apply)
static int reconstruct_plane(const SizeOps* ops, const SizeOps* ops_c, const Workspace& w, const float2* M,
                             const float2* Pp, const int* shifts, int64_t S, int pn, int N, const EdgeGeom& eg,
                             const float* ic, float* out, hipStream_t st)
{
    // 1. forward transform of the real coarse image (the machinery of the mask-spectrum pre-step, window = everything)
    PassGeom gf;
    gf.pn = pn; gf.c = pn / 2; gf.N = pn; gf.nt = (pn + 3) / 4;
    gf.kx0 = -pn / 2; gf.kx1 = pn / 2; gf.ky0 = gf.kx0; gf.ky1 = gf.kx1;
    gf.rows = pn; gf.general = 0; gf.rect_off = 0; gf.gcombine = 0; gf.row_pairs = 0; gf.coop_dma = 0;
    gf.xmask = slot_mask(pn, gf.kx0, gf.kx1); gf.ymask = gf.xmask;
    set_tile(gf, gf.rows);
    RealImageLoader ldr{ic, pn, 0, nullptr};
    HIP_TRY(ops_c->xpass_real_fwd(ldr, w.T, w.twtab2, gf, st));
    HIP_TRY(ops_c->ypass_field(-1, w.T, w.chat, w.twtab2, gf, st));
    // 2. zoom back to the fine grid: pn x pn coefficients, N-point transforms, the pn centred outputs
    PassGeom gi;
    make_geom(gi, pn, N, 0, 0, pn, pn, 0, 4);
    FieldLoader ldf{w.chat, nullptr};
    HIP_TRY(ops->xpass_field_inv(ldf, w.T, w.twtab, gi, st));
    HIP_TRY(ops->ypass_addreal(w.T, out, (float)(1.0 / ((double)pn * (double)pn)), w.twtab, gi, st));
    // 3. the Nyquist lines
    if (eg.len[0] > 0 || eg.len[1] > 0) {
        // partial sums: one workgroup per chunk of source points.  At least 8 points per chunk (a short list in 1024 chunks of
        // three points made k_nyquist_reduce the larger of the two kernels: config 1, 11 us to fold 4 MB of partials)
        int64_t want = (S + 7) / 8;
        const int chunks = (int)(want < 1 ? 1 : (want > GAM_CHUNKS ? GAM_CHUNKS : want));
        hipLaunchKernelGGL(k_nyquist_edges, dim3(chunks), dim3(256), 0, st, Pp, M, shifts, (long long)S, eg, w.gam);
        hipLaunchKernelGGL(k_nyquist_reduce, dim3(2 * 2 * EDGE_MAX / NYQ_RED_IDX), dim3(256), 0, st, w.gam, chunks);
        hipLaunchKernelGGL(k_nyquist_profiles, dim3((pn + 255) / 256, 2), dim3(256), 0, st, w.gam, w.twtab, eg, N);
        hipLaunchKernelGGL(k_nyquist_apply, dim3((pn + 255) / 256, pn), dim3(256), 0, st, out, w.gam, pn);
        HIP_TRY(hipGetLastError());
    }
    return LITHO_OK;
}

// Everything after the plan words are known: which kernels run and how the work is batched, the source-point loop, the
// coarse-grid reconstruction.  `pl` = plan words in THIS grid's coordinates.
static int accumulate_planned(const float2* M, const float2* P, int planes, const int* shifts, int64_t S, const int pl[PLAN_WORDS],
                              int pn, int N, float* out, const Workspace& w, const Knobs& kn, const SizeOps* ops, hipStream_t st,
                              bool tw2_ready = false)
{
    int rc;
    if (S <= 0) return LITHO_OK;
    const SizeOps* ops_c = size_ops(ilog2(pn));
    RunPlan rp;
    rc = plan_run(rp, w.t_bytes, kn, pl, pn, N, planes, S, device_cus(), ops_c != nullptr);
    if (rc) return rc;
    const AbbePlan& pp = rp.direct;
    const AbbePlan& pc_plan = rp.coarse_plan;
    const bool coarse = rp.coarse;
    const EdgeGeom& eg = rp.eg;
    const AbbePlan& run = coarse ? pc_plan : pp;
    const size_t plane_elems = (size_t)pn * pn;
    int64_t nx = 0;
    // profiling: ONE event per kernel-class boundary (E0 x E1 y E2 x E3 ...); consecutive events bracket the
    // launches of one pass over one batch.  (Two events recorded back to back alias on ROCm, so no begin/end pairs.)
    MarkList marks(g_profiling != 0, st);
    if (coarse && !tw2_ready) hipLaunchKernelGGL(k_twiddle_table, dim3((pn + 255) / 256), dim3(256), 0, st, w.twtab2, pn);
    for (int p0 = 0; p0 < planes; p0 += run.PC) {
        const int pc = (planes - p0 < run.PC) ? planes - p0 : run.PC;
        const float2* Pc = P + (size_t)p0 * plane_elems;
        if (!coarse) {
            rc = accumulate_chunk(pp, ops, w, w.twtab, M, Pc, pc, shifts, S, pn, out + (size_t)p0 * plane_elems, false, st, marks, nx);
            if (rc) return rc;
            continue;
        }
        rc = accumulate_chunk(pc_plan, ops_c, w, w.twtab2, M, Pc, pc, shifts, S, pn, w.ic, true, st, marks, nx);
        if (rc) return rc;
        for (int q = 0; q < pc; ++q) {
            rc = reconstruct_plane(ops, ops_c, w, M, Pc + (size_t)q * plane_elems, shifts, S, pn, N, eg,
                                   w.ic + (size_t)q * plane_elems, out + (size_t)(p0 + q) * plane_elems, st);
            if (rc) return rc;
        }
    }
    if (g_profiling) {
        for (int i = 0; i < 8; ++i) g_profile[i] = 0;
        if (!marks.v.empty()) (void)hipEventSynchronize(marks.v.back().ev);
        for (size_t i = 1; i < marks.v.size(); ++i) {
            float ms = 0.f;
            const Mark& m = marks.v[i];
            if (m.kind >= 0 && hipEventElapsedTime(&ms, marks.v[i - 1].ev, m.ev) == hipSuccess) {
                g_profile[m.kind * 3 + 0] += ms;
                g_profile[m.kind * 3 + 1] += 1;
                g_profile[m.kind * 3 + 2] += m.items;
            }
        }
        g_profile[6] = run.wave_y ? 1 : 0;
        g_profile[7] = run.PC;
    }
    g_last_plan[0] = run.general; g_last_plan[1] = run.r0; g_last_plan[2] = run.c0; g_last_plan[3] = run.h;
    g_last_plan[4] = run.wdt; g_last_plan[5] = run.bs; g_last_plan[6] = nx; g_last_plan[7] = pp.variant;
    g_last_plan[8] = run.PC; g_last_plan[9] = run.G; g_last_plan[10] = run.xchunk;
    g_last_plan[11] = run.fused_x ? 1 : (run.split_x ? 2 : (run.rect_x ? 3 : 0));
    g_last_plan[12] = coarse ? 1 : 0;
    g_last_plan[13] = run.wave_y ? 1 : 0;
    g_last_plan[14] = pp.natural_box ? 1 : 0;
    return LITHO_OK;
}

static int accumulate_embedded(const float2* M, const float2* P, int planes, const int* shifts, int64_t S, const int pl[PLAN_WORDS],
                               int pn, int pe, int N, float* out, void* ws, size_t ws_bytes, size_t t_cap, const Knobs& kn,
                               const SizeOps* ops, hipStream_t st);
void launch_embed_c64(const float2* src, int planes, int pn, float2* dst, int pe, hipStream_t st);     // optics.hip
void launch_crop_add_f32(const float* src, int planes, int pe, float* dst, int pn, hipStream_t st);

// `reuse`: optional caller-held plan record (litho_abbe_plan).  Valid and matching (pn, N, planes): its words are used, no
// planning kernel is launched and the call never waits for the stream; otherwise the plan is made as usual and recorded.
static int abbe_accumulate(const float2* M, const float2* P, int planes, const int* shifts, int64_t S,
                           const int* count_dev, int64_t* count_out, int pn, int N, float* out, void* ws,
                           size_t ws_bytes, hipStream_t st, litho_abbe_plan* reuse = nullptr,
                           const litho_abbe_options* opts = nullptr)
{
    int rc = check_sizes(pn, N);
    if (rc) return rc;
    if (!M || !P || !out || planes < 1 || S < 0 || (S > 0 && !shifts)) return LITHO_E_ARG;
    if (count_out) *count_out = 0;
    if (S == 0) return LITHO_OK;
    Workspace w;
    if (!carve(ws, ws_bytes, pn, N, w)) return LITHO_E_WORKSPACE;
    const SizeOps* ops = size_ops(ilog2(N));
    if (!ops) return LITHO_E_ARG;
    if (opts && (opts->size < (int32_t)sizeof(int32_t) || opts->size > 4096)) return LITHO_E_ARG;
    const Knobs kn = Knobs::read(opts);
    // the grid this problem runs at: its own, or the padded one of an embedded evaluation (embedded_size) when the workspace
    // has room for it (litho_abbe_workspace_bytes says so; an older, smaller workspace simply runs the problem as it is)
    const int pe = run_size(pn, N, kn, ws_bytes);

    if (kn.poison) {
        // test knob: every scratch region starts the call as NaN bit patterns -- a kernel that reads scratch it (or an
        // earlier launch of THIS call) has not written turns the image into NaN (tests/test_gpu_abbe.py)
        unsigned char* lo = (unsigned char*)ws + ws_layout(pn, N).twtab2.off;                           // behind plan words + twiddle table
        unsigned char* hi = pe != pn ? (unsigned char*)ws + workspace_bytes_at(pe, N) + embed_extra_bytes(pe) : (unsigned char*)w.T + w.t_bytes;
        if ((unsigned char*)w.T + w.t_bytes > hi) hi = (unsigned char*)w.T + w.t_bytes;
        HIP_TRY(hipMemsetAsync(lo, 0xFF, (size_t)(hi - lo), st));
    }
    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    // the coarse grid's table too, when a run at this size could take that path: in front of the planning read-back it costs
    // nothing, behind it it is one more small launch between the host's wake-up and the first x-pass
    const bool tw2_ready = pe == pn && kn.coarse && coarse_eligible(pn, N);
    if (tw2_ready) hipLaunchKernelGGL(k_twiddle_table, dim3((pn + 255) / 256), dim3(256), 0, st, w.twtab2, pn);
    int pl[PLAN_WORDS];
    // (the record names the grid the edge words were looked up for -- record_run_size: a record made with the embedding off, or
    // with a smaller workspace, is not reused for an embedded run and vice versa -- and carries a format tag: words another
    // library version wrote are ignored and the call plans afresh)
    const bool from_record = reuse && reuse->valid == 1 && reuse->pn == pn && reuse->N == N && reuse->planes == planes &&
                             record_is_ours(reuse->words) && record_count(reuse->words) <= S && record_run_size(reuse->words) == pe;
    int sw[10];
    bool rec_split = false;                                  // the record carries the outcome of the source-list split
    if (from_record) {
        rec_split = record_load(reuse->words, pl, sw);
    } else {
        rc = gather_plan(w, P, planes, pn, pe, shifts, S, count_dev, pl, st);      // the ONE host wait of the image path
        if (rc) return rc;
        if (reuse) {
            record_store(reuse->words, pl, pe, nullptr);     // (a split, below, stores its ten words too)
            reuse->pn = pn; reuse->N = N; reuse->planes = planes; reuse->valid = 1;
        }
    }
    g_last_plan[15] = from_record ? 1 : 0;
    S = pl[8];                                               // = S, or the device-side count of the source list
    if (count_out) *count_out = S;
    if (S == 0 || pl[1] < pl[0]) return LITHO_OK;            // no source point / pupil identically zero: nothing to add

    // Embedded evaluation -- unless a shift wraps the pupil around the CALLER's grid: the reference rolls modulo its own size
    // (imageformation.py:63), which the padded grid would not reproduce; such a list runs the general path at this size.
    const bool nowrap = list_nowrap(pl, pn);
    // A planned call splits exactly when its record says the planning call did: it re-runs the three split kernels (same list,
    // same deterministic result, no read-back) and plans the two parts from the recorded counts and extents.
    if (split_wanted(kn, nowrap, from_record && !rec_split, S)) {
        // Some shift wraps the pupil around the grid -- usually for a minority of the points of a shifted source.  Split the
        // list (stable, on the device; one more 40-byte read-back) and give each part the path it needs: this function
        // again for the non-wrapping points (pruned box, coarse grid, embedding: 2.5 us per point at 1024^2), the
        // general path for the others (10 us).  The two lists live at the end of the T region, which shrinks by them
        // (split_layout, abbe_plan.hpp: absolute placement -- the lists end where the T region of the grid the non-wrapping
        // part runs at ends, the padded grid's for an embedded size, and BOTH carves' T regions are cut short of them).
        const SplitLayout sl = split_layout(pn, pe, N, S);
        if (sl.ok) {
            Workspace ws_split = w;
            ws_split.t_bytes = sl.t_bytes_own;
            const size_t list_start = sl.list_a.off, list_bytes = sl.list_bytes;
            int* list_a = (int*)((unsigned char*)ws + list_start);
            int* list_b = (int*)((unsigned char*)list_a + list_bytes);
            int* counts = (int*)w.T;                                      // block counts: the head of T, free until the loops start
            int* words = w.plan + 32;
            const SplitBox box{pl[0], pl[1], pl[2], pl[3], pn};
            const int nblocks = (int)((S + SPLIT_PER_BLOCK - 1) / SPLIT_PER_BLOCK);
            hipLaunchKernelGGL(k_split_count, dim3(nblocks), dim3(256), 0, st, shifts, (long long)S, box, counts);
            hipLaunchKernelGGL(k_split_scan, dim3(1), dim3(1024), 0, st, counts, nblocks, (long long)S, words);
            hipLaunchKernelGGL(k_split_write, dim3(nblocks), dim3(256), 0, st, shifts, (long long)S, box, counts, list_a, list_b, words);
            HIP_TRY(hipGetLastError());
            if (!from_record) {
                rc = read_words(words, sw, 10, st);
                if (rc) return rc;
                if (reuse) record_store(reuse->words, pl, pe, sw);
            }
            int64_t launches = 0;
            for (int part = 0; part < 2; ++part) {
                const int64_t n = sw[part];
                if (n <= 0) continue;
                int plp[PLAN_WORDS];
                for (int i = 0; i < PLAN_WORDS; ++i) plp[i] = pl[i];
                for (int i = 0; i < 4; ++i) plp[4 + i] = sw[2 + 4 * part + i];
                plp[8] = (int)n;
                const int* lst = part == 0 ? list_a : list_b;
                // part 0 cannot wrap (it may run embedded); part 1 wraps by construction: general mode at this size
                if (part == 0 && pe != pn) rc = accumulate_embedded(M, P, planes, lst, n, plp, pn, pe, N, out, ws, ws_bytes, list_start, kn, ops, st);
                else rc = accumulate_planned(M, P, planes, lst, n, plp, pn, N, out, ws_split, kn, ops, st, tw2_ready);
                if (rc) return rc;
                launches += g_last_plan[6];
            }
            g_last_plan[6] = launches;                                    // (the other fields describe the part that ran last)
            g_last_plan[15] = from_record ? 3 : 2;                        // = the source list was split (3: from the record, no read-back)
            return LITHO_OK;
        }
    }
    if (pe == pn || !nowrap || kn.force_general) return accumulate_planned(M, P, planes, shifts, S, pl, pn, N, out, w, kn, ops, st, tw2_ready);
    return accumulate_embedded(M, P, planes, shifts, S, pl, pn, pe, N, out, ws, ws_bytes, 0, kn, ops, st);
}

// The embedded evaluation of a non-wrapping source list: pad into the scratch behind the padded size's workspace regions, run at
// pe, add the centre back.  t_cap > 0: byte offset in the workspace at which the T region must end (the lists of a split
// source list follow it).
static int accumulate_embedded(const float2* M, const float2* P, int planes, const int* shifts, int64_t S, const int pl[PLAN_WORDS],
                               int pn, int pe, int N, float* out, void* ws, size_t ws_bytes, size_t t_cap, const Knobs& kn,
                               const SizeOps* ops, hipStream_t st)
{
    int rc;

    Workspace w2;
    if (!carve(ws, ws_bytes, pe, N, w2)) return LITHO_E_WORKSPACE;          // (cannot fail: checked by the caller)
    if (t_cap > 0) {
        const size_t t0 = (size_t)((unsigned char*)w2.T - (unsigned char*)ws);
        if (t0 + w2.t_bytes > t_cap) w2.t_bytes = t_cap - t0;
    }
    const EmbedLayout el = embed_layout(pe, N);
    const size_t e2 = (size_t)pe * pe;
    float2* M2 = (float2*)((unsigned char*)ws + el.M2.off);
    float2* P2 = (float2*)((unsigned char*)ws + el.P2.off);
    float* O2 = (float*)((unsigned char*)ws + el.O2.off);
    int pl2[PLAN_WORDS];
    embed_plan_words(pl, pn, pe, pl2);
    launch_embed_c64(M, 1, pn, M2, pe, st);
    for (int p0 = 0; p0 < planes; p0 += COARSE_PLANES) {
        const int pc = planes - p0 < COARSE_PLANES ? planes - p0 : COARSE_PLANES;
        launch_embed_c64(P + (size_t)p0 * pn * pn, pc, pn, P2, pe, st);
        HIP_TRY(zero_async(O2, (size_t)pc * e2 * sizeof(float), st));
        rc = accumulate_planned(M2, P2, pc, shifts, S, pl2, pe, N, O2, w2, kn, ops, st);
        if (rc) return rc;
        launch_crop_add_f32(O2, pc, pe, out + (size_t)p0 * pn * pn, pn, st);
        HIP_TRY(hipGetLastError());
    }
    return LITHO_OK;
}

static int abbe_field(const float2* pf, const float2* M, int pn, int N, float2* field, void* ws, size_t ws_bytes,
                      hipStream_t st)
{
    int rc = check_sizes(pn, N);
    if (rc) return rc;
    if (!pf || !M || !field) return LITHO_E_ARG;
    Workspace w;
    if (!carve(ws, ws_bytes, pn, N, w)) return LITHO_E_WORKSPACE;
    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    int pl[PLAN_WORDS];
    rc = gather_plan(w, pf, 1, pn, pn, nullptr, 0, nullptr, pl, st);      // the pupil's support box (no shift list)
    if (rc) return rc;
    if (pl[1] < pl[0]) {                                      // zero pupil -> zero field
        HIP_TRY(zero_async(field, (size_t)pn * pn * sizeof(float2), st));
        return LITHO_OK;
    }
    PassGeom g;
    make_geom(g, pn, N, pl[0], pl[2], pl[1] - pl[0] + 1, pl[3] - pl[2] + 1, 0);
    HIP_TRY(zero_async(w.plan + 16, 2 * sizeof(int), st));     // a (0,0) shift
    AbbeLoader ld{pf, M, w.plan + 16, nullptr, nullptr, 0, 0};
    const SizeOps* ops = size_ops(ilog2(N));
    if (!ops) return LITHO_E_ARG;
    HIP_TRY(ops->xpass_general(ld, w.T, w.twtab, g, 1, st));
    HIP_TRY(ops->ypass_field(+1, w.T, field, w.twtab, g, st));
    return LITHO_OK;
}

void launch_scale_mask(const int16_t* geo, int pn, int ns, double scale, float* out, hipStream_t st);   // optics.hip

// Mask._ffFraunhofer (mask.py:74-90): bilinear scale by epsilon, pad/crop to N, centred
// forward DFT, keep the centre pn x pn:  spec[q] = sum_j padded[j] w^(-(j-N/2)(q-c)).
static int mask_spectrum(const int16_t* geo, int pn, double eps, int N, float2* spec, void* ws, size_t ws_bytes,
                         hipStream_t st)
{
    int rc = check_sizes(pn, N);
    if (rc) return rc;
    if (!geo || !spec || !(eps > 0)) return LITHO_E_ARG;
    Workspace w;
    if (!carve(ws, ws_bytes, pn, N, w)) return LITHO_E_WORKSPACE;
    const int ns = (int)floor((double)pn * eps);                 // F.interpolate output size (mask.py:77)
    if (ns < 1) return LITHO_E_ARG;
    const int diff = (N - pn) - (ns - pn);                       // mask.py:79, Python floor division
    const int pW = (diff >= 0) ? diff / 2 : -((-diff + 1) / 2);
    const int j0 = pW > 0 ? pW : 0;
    const int j1 = (pW + ns < N) ? pW + ns : N;
    const size_t nt = (pn + 3) / 4;
    if ((size_t)ns * ns * sizeof(float) > (size_t)g_cap(pn) * nt * 4 * pn * sizeof(float)) return LITHO_E_WORKSPACE;
    if (((size_t)(pn + 15) / 16 * 16) * (size_t)(j1 - j0) * sizeof(float2) > w.t_bytes) return LITHO_E_WORKSPACE;
    float* scaled = w.slab;                                      // the slab region is free here
    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    launch_scale_mask(geo, pn, ns, eps, scaled, st);
    HIP_TRY(hipGetLastError());
    PassGeom g;
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (int)nt;
    g.kx0 = j0 - N / 2; g.kx1 = j1 - N / 2;
    g.ky0 = g.kx0; g.ky1 = g.kx1;
    g.rows = j1 - j0; g.general = 0; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0;
    g.xmask = slot_mask(N, g.kx0, g.kx1);
    g.ymask = slot_mask(N, g.ky0, g.ky1);
    set_tile(g, g.rows);
    RealImageLoader ld{scaled, ns, j0 - pW, nullptr};
    const SizeOps* ops = size_ops(ilog2(N));
    if (!ops) return LITHO_E_ARG;
    HIP_TRY(ops->xpass_real_fwd(ld, w.T, w.twtab, g, st));
    HIP_TRY(ops->ypass_field(-1, w.T, spec, w.twtab, g, st));
    return LITHO_OK;
}

}  // namespace litho

// ----------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------
extern "C" {

int litho_abbe_workspace_bytes(int pn, int N, size_t* bytes_host)
{
    if (!bytes_host) return LITHO_E_ARG;
    int rc = litho::check_sizes(pn, N);
    if (rc) return rc;
    *bytes_host = litho::workspace_bytes(pn, N);
    return LITHO_OK;
}

int litho_abbe_embedded_size(int pn, int N, int* size_host)
{
    if (!size_host) return LITHO_E_ARG;
    int rc = litho::check_sizes(pn, N);
    if (rc) return rc;
    *size_host = litho::embedded_size(pn, N);
    return LITHO_OK;
}

int litho_abbe_accumulate(const void* maskFT, const void* pupil, int planes, const int32_t* shifts, int64_t S,
                          int pn, int N, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    return litho::abbe_accumulate((const float2*)maskFT, (const float2*)pupil, planes, shifts, S, nullptr, nullptr, pn, N,
                                  out, workspace, workspace_bytes, (hipStream_t)stream);
}

int litho_abbe_accumulate_counted(const void* maskFT, const void* pupil, int planes, const int32_t* shifts,
                                  const int32_t* count_dev, int64_t capacity, int pn, int N, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream, int64_t* count_host)
{
    if (!count_dev) return LITHO_E_ARG;
    return litho::abbe_accumulate((const float2*)maskFT, (const float2*)pupil, planes, shifts, capacity, count_dev,
                                  count_host, pn, N, out, workspace, workspace_bytes, (hipStream_t)stream);
}

int litho_abbe_accumulate_planned(const void* maskFT, const void* pupil, int planes, const int32_t* shifts,
                                  const int32_t* count_dev, int64_t capacity, int pn, int N, float* out,
                                  void* workspace, size_t workspace_bytes, void* stream, litho_abbe_plan* plan,
                                  int64_t* count_host)
{
    if (!plan) return LITHO_E_ARG;
    return litho::abbe_accumulate((const float2*)maskFT, (const float2*)pupil, planes, shifts, capacity, count_dev,
                                  count_host, pn, N, out, workspace, workspace_bytes, (hipStream_t)stream, plan);
}

int litho_abbe_accumulate_opts(const void* maskFT, const void* pupil, int planes, const int32_t* shifts,
                               const int32_t* count_dev, int64_t capacity, int pn, int N, float* out, void* workspace,
                               size_t workspace_bytes, void* stream, litho_abbe_plan* plan,
                               const litho_abbe_options* options, int64_t* count_host)
{
    return litho::abbe_accumulate((const float2*)maskFT, (const float2*)pupil, planes, shifts, capacity, count_dev,
                                  count_host, pn, N, out, workspace, workspace_bytes, (hipStream_t)stream, plan, options);
}

int litho_abbe_field(const void* pf, const void* maskFT, int pn, int N, void* field, void* workspace,
                     size_t workspace_bytes, void* stream)
{
    return litho::abbe_field((const float2*)pf, (const float2*)maskFT, pn, N, (float2*)field, workspace,
                             workspace_bytes, (hipStream_t)stream);
}

int litho_mask_spectrum(const int16_t* geometry, int pn, double epsilon, int N, void* spectrum, void* workspace,
                        size_t workspace_bytes, void* stream)
{
    return litho::mask_spectrum(geometry, pn, epsilon, N, (float2*)spectrum, workspace, workspace_bytes,
                                (hipStream_t)stream);
}

int litho_abbe_set_profiling(int on)
{
    litho::g_profiling = on ? 1 : 0;
    return LITHO_OK;
}

int litho_abbe_last_profile(double fields_host[8])
{
    if (!fields_host) return LITHO_E_ARG;
    memcpy(fields_host, litho::g_profile, sizeof(litho::g_profile));
    return LITHO_OK;
}

int litho_abbe_last_plan(int64_t fields_host[16])
{
    if (!fields_host) return LITHO_E_ARG;
    memcpy(fields_host, litho::g_last_plan, sizeof(litho::g_last_plan));
    return LITHO_OK;
}

}  // extern "C"
