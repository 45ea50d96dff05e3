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

// plan words: [0]=min row,[1]=max row,[2]=min col,[3]=max col of non-zero pupil samples
//             [4]=min dy,[5]=max dy,[6]=min dx,[7]=max dx, [8]=source-point count
//             [9],[10]=min/max ROW with a non-zero sample on the columns c +- pn/4 (the edges of the natural support box),
//             [11],[12]=min/max COLUMN with a non-zero sample on the rows c +- pn/4, [13]=1 if a corner of that box is set
static constexpr int PLAN_WORDS = 14;
__global__ void k_plan_init(int* plan)
{
    if (threadIdx.x < 8) plan[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
    if (threadIdx.x == 8) plan[8] = 0;
    if (threadIdx.x >= 9 && threadIdx.x <= 12) plan[threadIdx.x] = (threadIdx.x & 1) ? INT_MAX : INT_MIN;
    if (threadIdx.x == 13) plan[13] = 0;
}

// Bounding box of the non-zero pupil samples over all planes: grid (row blocks, planes), each block scans
// BOX_ROWS_PER_BLOCK rows of one plane; one atomic quadruple per block that saw a non-zero.
static constexpr int BOX_ROWS_PER_BLOCK = 8;
__global__ __launch_bounds__(256) void k_pupil_box(const float2* __restrict__ P, int pn, int* plan, int e_lo, int e_hi)
{
    __shared__ int red[4][4];
    const int row0 = blockIdx.x * BOX_ROWS_PER_BLOCK;
    const float2* plane = P + (size_t)blockIdx.y * pn * pn;
    int rmin = INT_MAX, rmax = INT_MIN, cmin = INT_MAX, cmax = INT_MIN;
    for (int row = row0; row < min(pn, row0 + BOX_ROWS_PER_BLOCK); ++row) {
        const float2* r = plane + (size_t)row * pn;
        for (int j = threadIdx.x; j < pn; j += blockDim.x) {
            const float2 v = r[j];
            if (v.x != 0.f || v.y != 0.f) {
                rmin = min(rmin, row); rmax = max(rmax, row);
                cmin = min(cmin, j); cmax = max(cmax, j);
                // edges of the natural box of the grid the engine will RUN at (e_lo, e_hi = c -+ pn/4, or -+ pe/4 of the padded
                // grid of an embedded evaluation -- then possibly outside this array): a handful of samples, direct atomics
                const bool ecol = (j == e_hi || j == e_lo), erow = (row == e_hi || row == e_lo);
                if (ecol) { atomicMin(&plan[9], row); atomicMax(&plan[10], row); }
                if (erow) { atomicMin(&plan[11], j); atomicMax(&plan[12], j); }
                if (ecol && erow) atomicExch(&plan[13], 1);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        rmin = min(rmin, __shfl_xor(rmin, off)); rmax = max(rmax, __shfl_xor(rmax, off));
        cmin = min(cmin, __shfl_xor(cmin, off)); cmax = max(cmax, __shfl_xor(cmax, off));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wv][0] = rmin; red[wv][1] = rmax; red[wv][2] = cmin; red[wv][3] = cmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) {
            rmin = min(rmin, red[i][0]); rmax = max(rmax, red[i][1]);
            cmin = min(cmin, red[i][2]); cmax = max(cmax, red[i][3]);
        }
        if (rmax >= 0) {
            atomicMin(&plan[0], rmin); atomicMax(&plan[1], rmax);
            atomicMin(&plan[2], cmin); atomicMax(&plan[3], cmax);
        }
    }
}

// S comes from the host, or -- asynchronous image path -- from the device word litho_source_compact left behind
// (count_dev, clamped to `S` = the capacity); plan[8] returns the count actually used.
__global__ void k_shift_extents(const int* __restrict__ shifts, long long S, const int* __restrict__ count_dev, int* plan)
{
    if (count_dev) {
        const long long c = *count_dev;
        S = c < S ? c : S;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) plan[8] = (int)S;
    int ymin = INT_MAX, ymax = INT_MIN, xmin = INT_MAX, xmax = INT_MIN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < S; i += (long long)gridDim.x * blockDim.x) {
        const int dy = shifts[2 * i], dx = shifts[2 * i + 1];
        ymin = min(ymin, dy); ymax = max(ymax, dy); xmin = min(xmin, dx); xmax = max(xmax, dx);
    }
    for (int off = 32; off > 0; off >>= 1) {
        ymin = min(ymin, __shfl_xor(ymin, off)); ymax = max(ymax, __shfl_xor(ymax, off));
        xmin = min(xmin, __shfl_xor(xmin, off)); xmax = max(xmax, __shfl_xor(xmax, off));
    }
    if ((threadIdx.x & 63) == 0 && ymin != INT_MAX) {
        atomicMin(&plan[4], ymin); atomicMax(&plan[5], ymax);
        atomicMin(&plan[6], xmin); atomicMax(&plan[7], xmax);
    }
}

// out[p][qy][qx] += sum_g slab[p * gstride + g][qx][qy]   (32x32 tiles through LDS; blockIdx.z = plane p).  1024 threads per
// tile, one element each: small images fold up to 64 slabs into a few dozen tiles (256^2: 64 tiles), and with 256 threads
// walking four rows each that took 38 us per image; fixed summation order (deterministic).
__global__ __launch_bounds__(1024) void k_slab_reduce(const float* __restrict__ slab, float* __restrict__ out, int pn, int ldq, int G,
                                                      int gstride)
{
    __shared__ float tile[32][33];
    slab += (size_t)blockIdx.z * gstride * ldq * pn;
    out += (size_t)blockIdx.z * pn * pn;
    const int qx0 = blockIdx.x * 32, qy0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 1024 threads: ty in 0..31
    {
        const int qx = qx0 + ty, qy = qy0 + tx;
        float v = 0.f;
        if (qx < pn && qy < pn)
            for (int gidx = 0; gidx < G; ++gidx) v += slab[((size_t)gidx * ldq + qx) * pn + qy];
        tile[ty][tx] = v;
    }
    __syncthreads();
    const int qy = qy0 + ty, qx = qx0 + tx;
    if (qx < pn && qy < pn) out[(size_t)qy * pn + qx] += tile[tx][ty];
}

static constexpr int COARSE_PLANES = 4;                      // = the largest plane chunk
static constexpr int EDGE_MAX = 128;                         // longest box-edge support the coarse path handles
static constexpr int GAM_CHUNKS = 1024;                      // workgroups (partial sums) of k_nyquist_edges
static constexpr size_t GAM_PARTIAL = (size_t)GAM_CHUNKS * 2 * 2 * EDGE_MAX;      // [chunk][edge][2 * EDGE_MAX]
static size_t gam_float2(int pn) { return GAM_PARTIAL + 2 * 2 * EDGE_MAX + 2 * (size_t)pn; }

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
struct EdgeGeom {
    int pn, c, h;            // grid, centre, half-width of the natural box
    int lo[2], len[2];       // edge 0: columns c +- h, support rows lo[0] .. lo[0] + len[0]; edge 1: rows c +- h, support columns
};

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
static constexpr int SPLIT_PER_BLOCK = 1024;
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
// host side
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


// y-pass groups = private partial images (slabs).  Up to 8 for large images; small images can afford more
// (their y-pass grid would otherwise be a handful of workgroups): as many as fit in 128 MiB, at most 64.
static int g_cap(int pn)
{
    const size_t one = (size_t)((pn + 3) / 4) * 4 * pn * sizeof(float);
    size_t n = ((size_t)128 << 20) / one;
    return n < 8 ? 8 : (n > 64 ? 64 : (int)n);
}
static constexpr int SLAB_FLUSH_BATCHES = 64;
static constexpr size_t T_BUDGET_MAX = (size_t)1 << 30;
static constexpr size_t T_BUDGET_BIG = (size_t)4 << 30;      // images whose T items cannot be batched inside the Infinity Cache (pn >= 4096)
static constexpr size_t T_BUDGET_MIN = (size_t)256 << 20;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t t_budget(int pn)
{
    const size_t nt = (pn + 3) / 4;
    const size_t one_general = nt * (size_t)pn * 4 * sizeof(float2);
    size_t b = 64 * one_general;
    if (b < T_BUDGET_MIN) b = T_BUDGET_MIN;                  // small images: room for batches of several points per y-pass group
    // pn >= 4096: a T item is 67 MB and more, T streams through HBM whatever the batch, and longer batches amortise the
    // y-pass's accumulator flush and ramp (4096^2, us per source point: 15 items 45.3, 30 44.1, 45 43.5, 60 43.2) --
    // 4 GiB of a 288 GB device.
    const size_t cap = pn >= 4096 ? T_BUDGET_BIG : T_BUDGET_MAX;
    if (b > cap) b = cap;
    if (b < one_general + one_general / 2) b = one_general + one_general / 2;
    return b;
}

// Sizes at which the coarse-grid path exists (its regions of the workspace are empty elsewhere: 1.5 GiB at 8192^2)
static bool coarse_eligible(int pn, int N)
{
    return N == 2 * pn && (pn == 256 || pn == 512 || pn == 1024 || pn == 2048 || pn == 4096);
}
static size_t ic_bytes(int pn, int N) { return coarse_eligible(pn, N) ? (size_t)COARSE_PLANES * pn * pn * sizeof(float) : 0; }
static size_t chat_bytes(int pn, int N) { return coarse_eligible(pn, N) ? (size_t)pn * pn * sizeof(float2) : 0; }
static size_t gam_bytes(int pn, int N) { return coarse_eligible(pn, N) ? gam_float2(pn) * sizeof(float2) : 0; }

// Grid size the engine RUNS a pn x pn problem at.  The specialised kernels and the coarse grid
// exist for pn = N and pn = N / 2; any other even size (a 1000^2 or 3000^2 mask; 10 nm pixels, where N = 4 pn) would fall to
// the generic, runtime-predicated kernels -- 3.4-3.6x slower per source point than the NEXT LARGER power of two.  Such a
// problem is embedded instead: mask spectrum and pupil centred in a zero-padded N / 2 (pn < N / 2) or N grid, same shift
// list, centre pn x pn of the accumulated intensity added to `out` -- the identical sum term by term as long as no shift
// wraps the pupil around the caller's own grid (checked on the original size; a wrapping list runs the general path as is).
// Measured (scripts/embed_ab.py, us per source point, embedded / plain): 1000^2 at N 2048 2.48 / 8.38, 2000^2 at N 4096
// 9.2 / 33.6, 1500^2 at N 2048 7.4 / 16.2, 3000^2 at N 4096 35.2 / 58.8; N = 4 pn: 256^2 0.63 / 0.94, 2048^2 28.4 / 37.7;
// but 300^2 in a 512 grid 0.52 / 0.50 -- so: N / 2 from 256 up (the coarse grid applies), N from 1024 up.
static int embedded_size(int pn, int N)
{
    if (pn == N || 2 * pn == N || (pn & 1)) return pn;
    if (2 * pn < N) return N / 2 >= 256 ? N / 2 : pn;
    return N >= 1024 ? N : pn;
}
// scratch of an embedded evaluation behind the workspace of the padded size: mask spectrum, COARSE_PLANES pupils, as many images
static size_t embed_extra_bytes(int pe)
{
    const size_t e = (size_t)pe * pe;
    return align_up(e * sizeof(float2), 256) + align_up(COARSE_PLANES * e * sizeof(float2), 256) + align_up(COARSE_PLANES * e * sizeof(float), 256);
}

static size_t workspace_bytes_at(int pn, int N)
{
    const size_t nt = (pn + 3) / 4;
    size_t b = 256;
    b += align_up((size_t)N * sizeof(float2), 256);
    b += align_up((size_t)pn * sizeof(float2), 256);
    b += align_up((size_t)g_cap(pn) * nt * 4 * pn * sizeof(float), 256);
    b += align_up(ic_bytes(pn, N), 256);
    b += align_up(chat_bytes(pn, N), 256);
    b += align_up(gam_bytes(pn, N), 256);
    b += align_up(t_budget(pn), 256);
    return b;
}
// what litho_abbe_workspace_bytes reports: the engine's own regions at this size, or -- for a size that runs embedded -- the
// larger of that (the general path of a wrapping source list) and the padded size's regions + the embedding scratch
static size_t workspace_bytes(int pn, int N)
{
    const size_t own = workspace_bytes_at(pn, N);
    const int pe = embedded_size(pn, N);
    if (pe == pn) return own;
    const size_t emb = workspace_bytes_at(pe, N) + embed_extra_bytes(pe);
    return own > emb ? own : emb;
}

static bool carve(void* ws, size_t bytes, int pn, int N, Workspace& w)
{
    if (!ws || bytes < workspace_bytes_at(pn, N)) return false;
    const size_t nt = (pn + 3) / 4;
    unsigned char* p = (unsigned char*)ws;
    w.plan = (int*)p; p += 256;
    w.twtab = (float2*)p; p += align_up((size_t)N * sizeof(float2), 256);
    w.twtab2 = (float2*)p; p += align_up((size_t)pn * sizeof(float2), 256);
    w.slab = (float*)p; p += align_up((size_t)g_cap(pn) * nt * 4 * pn * sizeof(float), 256);
    w.ic = (float*)p; p += align_up(ic_bytes(pn, N), 256);
    w.chat = (float2*)p; p += align_up(chat_bytes(pn, N), 256);
    w.gam = (float2*)p; p += align_up(gam_bytes(pn, N), 256);
    w.T = (float2*)p;
    w.t_bytes = t_budget(pn);
    return true;
}

static int check_sizes(int pn, int N)
{
    if (pn < 2 || pn > 16384 || (pn & 1)) return LITHO_E_ARG;
    if (N < 16 || N > 16384 || (N & (N - 1))) return LITHO_E_ARG;
    if (N < pn) return LITHO_E_NSMALL;
    return LITHO_OK;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// Tuning / test knobs.  Resolved ONCE per C-ABI call, never per launch: a field of the caller's litho_abbe_options that is
// >= 0 wins, otherwise the LITHO_ABBE_* environment variable, otherwise the default.
struct Knobs {
    int force_generic, force_general, groups, batch, xchunk, tile, w64, w64_8192, w64x, plane_chunk, xsplit, rect, xrect, coarse, gcombine, rowpairs, poison, embed, split, coopdma;
    static int pick(const litho_abbe_options* o, size_t off, const char* name, int dflt)
    {
        if (o && off + sizeof(int32_t) <= (size_t)o->size) {
            const int32_t v = *(const int32_t*)((const unsigned char*)o + off);
            if (v >= 0) return v;
        }
        return env_int(name, dflt);
    }
    static Knobs read(const litho_abbe_options* o)
    {
#define LITHO_KNOB(field, name, dflt) k.field = pick(o, offsetof(litho_abbe_options, field), name, dflt)
        Knobs k;
        LITHO_KNOB(force_generic, "LITHO_ABBE_FORCE_GENERIC", 0);
        LITHO_KNOB(force_general, "LITHO_ABBE_FORCE_GENERAL", 0);
        LITHO_KNOB(groups, "LITHO_ABBE_GROUPS", 0);
        LITHO_KNOB(batch, "LITHO_ABBE_BATCH", 0);
        LITHO_KNOB(xchunk, "LITHO_ABBE_XCHUNK", 0);
        LITHO_KNOB(tile, "LITHO_ABBE_TILE", 0);          // 0 = automatic (8 columns on the wave-kernel path, else 4)
        LITHO_KNOB(w64, "LITHO_ABBE_W64", 1);
        LITHO_KNOB(w64_8192, "LITHO_ABBE_W64_8192", 1);
        LITHO_KNOB(w64x, "LITHO_ABBE_W64X", 0);
        LITHO_KNOB(plane_chunk, "LITHO_ABBE_PLANE_CHUNK", 0);
        LITHO_KNOB(xsplit, "LITHO_ABBE_XSPLIT", 1);
        LITHO_KNOB(rect, "LITHO_ABBE_RECT", 1);
        LITHO_KNOB(xrect, "LITHO_ABBE_XRECT", 1);
        LITHO_KNOB(coarse, "LITHO_ABBE_COARSE", 1);
        LITHO_KNOB(gcombine, "LITHO_ABBE_GCOMBINE", 1);
        LITHO_KNOB(rowpairs, "LITHO_ABBE_ROWPAIRS", 0);
        LITHO_KNOB(poison, "LITHO_ABBE_POISON", 0);
        LITHO_KNOB(embed, "LITHO_ABBE_EMBED", 1);
        LITHO_KNOB(split, "LITHO_ABBE_SPLIT", 1);
        LITHO_KNOB(coopdma, "LITHO_ABBE_COOPDMA", 1);
#undef LITHO_KNOB
        return k;
    }
};

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

// Which specialised kernel variant fits this geometry (-1 = generic).
static int pick_variant(const PassGeom& g, const Knobs& kn)
{
    if (g.general || (g.pn & (g.pn - 1))) return -1;
    const int rl = ilog2(g.N) - ilog2(g.pn);
    if (rl < 0 || rl > 2) return -1;
    const unsigned nat = natural_in_mask(rl);
    if ((g.xmask & ~nat) || (g.ymask & ~nat)) return -1;
    return kn.force_generic ? -1 : rl;
}

// Reads the plan words back (one small synchronising copy).
static int read_plan(const Workspace& w, int host[PLAN_WORDS], hipStream_t st)
{
    HIP_TRY(hipMemcpyAsync(host, w.plan, PLAN_WORDS * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LITHO_OK;
}

// T tile width (columns): 4 unless LITHO_ABBE_TILE says 8 or 16 (tuning knob)
static void set_tile(PassGeom& g, int rows, int tc = 4)
{
    g.tcl = (tc == 16) ? 4 : (tc == 8) ? 3 : 2;
    const long long ntile = (g.pn + (1 << g.tcl) - 1) >> g.tcl;
    g.t_point = (ntile * rows) << g.tcl;
}

// bit e of the mask: some thread t of a line has its slot e (sample n = t + T*e) inside [lo,hi)
static unsigned slot_mask(int N, int lo, int hi)
{
    const int T = N / 16;
    unsigned m = 0;
    for (int e = 0; e < 16; ++e)
        for (int t = 0; t < T; ++t) {
            const int n = t + T * e;
            const int k = (n >= hi) ? n - N : n;
            if (k >= lo && k < hi) { m |= 1u << e; break; }
        }
    return m;
}

static void make_geom(PassGeom& g, int pn, int N, int r0, int c0, int h, int wdt, int general, int tile_cols = 4)
{
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (pn + 3) / 4;
    g.kx0 = c0 - g.c; g.kx1 = c0 + wdt - g.c;
    g.ky0 = r0 - g.c; g.ky1 = r0 + h - g.c;
    g.rows = h; g.general = general; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0;
    g.xmask = slot_mask(N, g.kx0, g.kx1);
    g.ymask = slot_mask(N, g.ky0, g.ky1);
    set_tile(g, h, tile_cols);
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

// Everything abbe_accumulate decides before its launch loop: which kernels run and how the work is batched.
struct AbbePlan {
    PassGeom g;
    int general, variant;       // 1: roll kept on P (wrapping shifts); kernel specialisation (-1 generic, else log2(N/pn))
    bool natural_box;           // the pupil's support box lies inside |k| <= pn/4 (and no shift wraps)
    int r0, c0, h, wdt;         // pupil support box (rows r0 .. r0+h, columns c0 .. c0+wdt)
    bool wave_y;                // y-pass by the wave-level family (k_ypass_wave / k_ypass_pair / k_ypass_rect)
    bool split_x, rect_x, fused_x;   // x-pass: k_xpass_split / k_xpass_rect / plane-fused k_xpass_abbe (else per-plane fall-backs)
    int PC, G, xchunk;          // planes in flight per launch pair, y-pass groups per plane, source points per x-pass workgroup
    int slabs;                  // slabs per plane the y-pass actually writes: G, or G / 2 when k_ypass_rect<.., 2> folds group pairs
    int64_t bs;                 // source points per batch
};

// pl = the 9 plan words read back from the device (pupil box, shift extents, count)
static int plan_abbe(AbbePlan& pp, const Workspace& w, const Knobs& kn, const int pl[PLAN_WORDS], int pn, int N, int planes)
{
    int r0 = pl[0], h = pl[1] - pl[0] + 1, c0 = pl[2], wdt = pl[3] - pl[2] + 1;
    const bool nowrap = (r0 + pl[4] >= 0) && (r0 + h - 1 + pl[5] <= pn - 1) &&
                        (c0 + pl[6] >= 0) && (c0 + wdt - 1 + pl[7] <= pn - 1);
    const int general = (!nowrap || kn.force_general) ? 1 : 0;
    if (general) { r0 = 0; c0 = 0; h = pn; wdt = pn; }
    PassGeom& g = pp.g;
    make_geom(g, pn, N, r0, c0, h, wdt, general, kn.tile > 0 ? kn.tile : 4);
    const int variant = pick_variant(g, kn);
    // The wave-level kernels, the split x-pass and the coarse-grid path hard-wire the NATURAL support |k| <= pn/4 (the
    // unit disk of the [-2,2) sigma grid): they load only the slots that cover it and the coarse grid assumes
    // |kappa| <= pn/2.  pick_variant's 16-slot masks are coarser than that (a one-sided box reaching k = 3 pn/8 - 1
    // still has the natural slot set), so the box itself is checked; anything wider runs the radix-16 kernels, whose
    // windows are runtime-predicated inside the admitted slots.
    const int cc = pn / 2, hh = pn / 4;
    const bool natural_box = !general && r0 >= cc - hh && r0 + h - 1 <= cc + hh && c0 >= cc - hh && c0 + wdt - 1 <= cc + hh;

    // wave-level y-pass kernels, N = 2 pn: N = 512, 1024, 2048 k_ypass_rect (8, 4, 2 columns per wave; fall-back
    // k_ypass_wave with S = 32 for 1024 and 2048), N = 4096 k_ypass_wave (S = 64), N = 8192 k_ypass_pair
    const int l2n = ilog2(N);
    const int lines_per_wg = (N / 16 >= 64) ? 1 : 64 / (N / 16);
    const bool rect_ok = kn.rect && (N == 2048 || N == 1024 || (N == 512 && (kn.tile <= 0 || kn.tile == 8)));
    const bool w64_ok = (pn * 2 == N) && ((N == 512 && rect_ok) || N == 1024 || N == 2048 || N == 4096 ||
                                          (N == 8192 && kn.w64_8192));   // w64_8192 defaults to 1
    // N = pn (the coarse-grid transform, and pixel sizes that give N = pn): full-output variants of the same kernels
    const bool full_ok = (pn == N) && (N == 1024 || N == 2048 || N == 4096 ||
                                       ((N == 512 || N == 256) && (kn.tile <= 0 || kn.tile == 8)));
    const bool w64_shape = ((w64_ok && variant == 1) || (full_ok && variant == 0)) && kn.w64 && natural_box;
    // T tile width.  The x-pass's T stores are bound by the memory system's rate for partial-line writes: measured
    // (scripts/ubench/write_bw.hip) 2.2 TB/s for 32-byte granules (4-column tiles), 3.4 TB/s for 64-byte granules
    // (8 columns), 5.2 TB/s for whole 128-byte lines.  The wave kernels read 8-column tiles at no extra cost, the
    // radix-16 y-pass does not (measured in round 1), so: 8 columns on the wave path, 4 elsewhere.
    // N = pn = 4096 (config 4's coarse grid): a T item is 67 MB, T streams through HBM, and there whole-line stores are
    // worth 7.5 us of the x-pass's 21.6 per item -- 16-column tiles, read by k_ypass_coop (24.4 us per item against
    // k_ypass_wave's 21.2 on 8-column tiles: 38.7 us per source point against 43.1).
    const bool coop16 = pn == N && N == 4096 && variant == 0;
    if (kn.tile <= 0 && w64_shape && !kn.w64x) set_tile(g, h, coop16 ? 16 : 8);
    const int tc = 1 << g.tcl;
    // k_ypass_rect: 4096 / N adjacent columns per wave (they must fit one T tile)
    const bool rect = (variant == 0 ? N <= 2048 : rect_ok && N <= 2048) && ((4096 / N) <= tc || (N == 256 && tc == 8));
    g.rect_off = rect ? 0 : 1;
    g.gcombine = kn.gcombine ? 1 : 0;
    g.row_pairs = (kn.rowpairs && g.tcl == 3) ? 1 : 0;
    g.coop_dma = kn.coopdma ? 1 : 0;
    const bool wave_y = w64_shape && (g.tcl == 2 || g.tcl == 3 || (g.tcl == 4 && N == 4096 && variant == 0)) && ((N != 512 && N != 256) || rect) &&
                        (variant == 1 || rect || N == 4096);

    // y-pass groups: the grid is (column blocks) x (planes in flight) x G workgroups; pick the smallest group
    // count that makes it a whole number of full-occupancy rounds (256 CUs x workgroups per CU).
    const int wave_cols = rect ? 4 * (4096 / N) : N == 1024 ? 8 : (N == 8192 ? 2 : 4);   // columns per wave-kernel workgroup
    const int wave_wpt = tc > wave_cols ? tc / wave_cols : 1;         // workgroups that share one T tile
    const int tile_blocks = !wave_y ? (g.nt + lines_per_wg - 1) / lines_per_wg
                            : wave_wpt == 1 ? (pn + wave_cols - 1) / wave_cols
                                            : wave_wpt * (((pn + tc - 1) / tc + 7) / 8 * 8);
    const int resident = device_cus() * (wave_y ? ((N <= 2048 && !rect) ? 4 : 2) : l2n <= 12 ? 3 : (l2n == 13 ? 2 : 1));
    int a_ = tile_blocks, b_ = resident;
    while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
    int Gtot = resident / a_;                                  // groups that fill whole rounds
    if (kn.groups > 0) Gtot = kn.groups;
    if (Gtot < 1) Gtot = 1;
    if (Gtot > g_cap(pn)) Gtot = g_cap(pn);

    // Through-focus stacks: PC planes are in flight per launch pair.  The fused x-pass gathers the mask-spectrum
    // window of a source point once for all of them (NP = 4 / 2 / 1 planes per workgroup); the y-pass gives every
    // plane its own groups and slabs.  The T buffer of a launch pair holds PC x batch items for the planes in
    // flight.  The gather was never the x-pass's limit (its T stores are): PC = 2 takes 40 % of the x-pass's load
    // instructions away at equal x-pass time, but its 2 x batch items of T leave the Infinity Cache and the y-pass
    // pays 4-5 % for that (2-3 % of the total against plane-by-plane); on the coarse-grid path, whose y-pass is twice
    // as fast, it pays 26 % (2048^2 x 8 planes, us per point and plane: PC = 1 9.42 / 9.53, PC = 2 10.82 / 10.83,
    // PC = 4 with a quarter of the batch 12.7).  Default: plane by plane; LITHO_ABBE_PLANE_CHUNK = 2 / 4 selects the
    // fused launches (parity-tested).
    int PC = planes < 1 ? planes : 1;
    if (kn.plane_chunk > 0) PC = kn.plane_chunk < planes ? kn.plane_chunk : planes;
    if (PC > g_cap(pn)) PC = g_cap(pn);

    // Batch = source points per launch pair.  The intermediate T of one batch (PC planes x points) should stay
    // INSIDE the 256 MiB Infinity Cache between the two passes, and a y-pass workgroup wants several points per plane
    // to amortise its accumulator flush.  Measured (us per source point, round 2): 1024^2 (4.2 MB per item) 32 items
    // 3.01, 48 2.85, 56 2.84, 68 3.09; 2048^2 (16.8 MB) 8 items 14.65, 12 and 17 equal within the +-2.5 % scatter of
    // single samples (profiles/r02_tuning_sweeps.txt) -> budget 208 MiB.
    const size_t item_bytes = (size_t)g.t_point * sizeof(float2);
    const int64_t items_ws = (int64_t)(w.t_bytes / item_bytes);
    if (items_ws < 1) return LITHO_E_WORKSPACE;
    int64_t items = items_ws;
    int64_t items_cache = (int64_t)(((size_t)208 << 20) / item_bytes);
    // 4096^2 (67 MB per item): not even 8 items fit the cache, T round-trips HBM whatever the batch -- then the batch
    // is as long as the workspace allows (60 items of its 4 GiB: fewer accumulator flushes in the y-pass) and an x-pass
    // workgroup walks 15 items for its row.  us per source point, coarse-grid path, alternating A/B (round 2, 1 GiB):
    // 8 items x chunks of 4 49.9 / 49.8; 12 x 6 47.9; 12 x 12 45.6; 15 x 5 47.5; 15 x 15 45.3 / 45.2; round 3 (4 GiB):
    // 30 x 15 44.1, 45 x 15 43.5, 60 x 15 43.2, 60 x 60 43.3.
    const bool beyond_cache = items_cache < 8;
    if (beyond_cache) items_cache = 60;
    if (items > items_cache) items = items_cache;
    if (PC > items_ws) PC = (int)items_ws;
    int G = Gtot / PC;                                         // groups per plane
    if (G < 1) G = 1;
    // Stacks keep the per-plane batch: T grows to PC x batch items and leaves the Infinity Cache, which costs the
    // y-pass less than flushing its accumulators twice as often (alternating A/B at 2048^2 x 8 planes, us per point
    // and plane, two boxes: PC = 1 13.40 / 13.76; PC = 2 with the batch halved 14.18, with the full batch 13.62 /
    // 14.05; PC = 4 13.68).
    int64_t bs = items;
    if (kn.batch > 0) bs = kn.batch;
    if (bs > items_ws / PC) bs = items_ws / PC;
    if (bs < 1) bs = 1;
    if (bs > 65535) bs = 65535;
    // Balance: every y-pass group gets the same number of points (batch multiple of G) and the x-pass
    // chunks divide the batch evenly (chunk = divisor of the batch nearest 4).
    const int64_t bs_cap = bs;
    if (kn.batch <= 0 && bs > G) bs -= bs % G;
    // Few, long batches (small images: config 1 is 3233 points in batches of up to 825): even batches instead of full ones
    // plus a short tail -- every launch pair costs 15-20 us before its first item (3233 = 4 x 768 + 161 was five launch
    // pairs, 4 x 809 is four).  A ragged split over the G groups (809 = 64 x 12 + 41) costs less than that.
    const int64_t S_plan = pl[8];
    if (kn.batch <= 0 && S_plan > bs && S_plan <= 64 * bs_cap) {
        const int64_t B = (S_plan + bs_cap - 1) / bs_cap;      // launch pairs needed at the cap
        int64_t even = (S_plan + B - 1) / B;
        if (even % G && even + (G - even % G) <= bs_cap) even += G - even % G;
        if (even >= 1 && (S_plan + even - 1) / even < (S_plan + bs - 1) / bs) bs = even;
    }
    int xchunk = kn.xchunk;                                    // source points per x-pass workgroup
    if (xchunk <= 0) {
        // ~4 source points per workgroup: the pupil rows (5 loads per plane) are amortised over the chunk, the
        // mask-spectrum window of each point over the planes (measured flat between 3 and 6 points)
        const int want = PC >= 4 ? 2 : 4;
        xchunk = want;
        if (want == 4) for (int cand : {4, 5, 3, 6, 2}) if (bs % cand == 0) { xchunk = cand; break; }
        if (want == 2) for (int cand : {2, 3, 1}) if (bs % cand == 0) { xchunk = cand; break; }
        if (beyond_cache && PC == 1) {                         // see above
            xchunk = (int)bs;
            for (int cand : {15, 12, 16, 10, 20, 8, 6}) if (bs > cand && bs % cand == 0) { xchunk = cand; break; }
        }
        // 1024-point rows (64-thread workgroups, 16 per CU): longer chunks pay -- coarse-grid x-pass at 1024^2,
        // us per point: chunk 2 1.40, 3 1.27, 4 1.18, 6 1.12, 8 1.20, 12 1.03, 16 1.06, 24 1.34, 48 2.0
        if (N == 1024 && variant == 0 && PC == 1) for (int cand : {12, 16, 8, 6}) if (bs % cand == 0) { xchunk = cand; break; }
    }

    // N = 8192 = 2 pn: each row as two 4096-point transforms (k_xpass_split) instead of the 8192-point engine
    pp.split_x = !general && variant == 1 && N == 8192 && g.tcl >= 2 && kn.xsplit && natural_box;
    // Several box rows per wave on the wave-level engine, whole-line T stores (k_xpass_rect).  Measured (us per source
    // point, radix-16 x-pass -> k_xpass_rect): N = 1024 0.57 -> 0.36, N = 2048 1.43 -> 1.46, N = 512 0.27 -> 0.26: its
    // loads are not prefetched (no registers left), so it only pays where the radix-16 engine is at its weakest.
    // LITHO_ABBE_XRECT: 0 off, 1 N = 1024 only (default), 2 every N <= 2048 (parity tests).
    // The same kernel with every bin kept serves the coarse-grid transforms (variant 0, N = pn) -- only on request:
    // with twice the loads and stores per wave it is SLOWER than the radix-16 x-pass at every size (coarse-grid x-pass,
    // us per point, radix-16 -> rect: N' = 512 0.28 -> 0.32, 1024 1.25 -> 1.49, 2048 4.73 -> 6.77).
    pp.rect_x = natural_box && ((variant == 1 && pn * 2 == N) || (variant == 0 && pn == N)) && N >= 512 && N <= 2048 &&
                g.tcl == 3 && (kn.xrect >= 2 || (kn.xrect == 1 && variant == 1 && N == 1024));
    const bool wave_x_optin = wave_y && variant == 1 && N == 4096 && kn.w64x && g.tcl == 2;      // k_xpass_w64 (slower, parity-tested)
    pp.fused_x = !pp.split_x && !pp.rect_x && !general && variant >= 0 && !wave_x_optin;
    pp.general = general; pp.variant = variant; pp.r0 = r0; pp.c0 = c0; pp.h = h; pp.wdt = wdt;
    pp.natural_box = natural_box;
    pp.wave_y = wave_y; pp.PC = PC; pp.G = G; pp.xchunk = xchunk; pp.bs = bs;
    pp.slabs = (wave_y && rect && g.gcombine && G % 2 == 0) ? G / 2 : G;
    return LITHO_OK;
}

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
static int accumulate_chunk(const AbbePlan& pp, const SizeOps* ops, const Workspace& w, const float2* twtab,
                            const float2* M, const float2* Pc, int pc, const int* shifts, int64_t S, int pn, float* dst,
                            hipStream_t st, MarkList& marks, int64_t& nx)
{
    const PassGeom& g = pp.g;
    const int variant = pp.variant, G = pp.G, xchunk = pp.xchunk;
    const int64_t bs = pp.bs;
    const size_t slab_plane = (size_t)g.nt * 4 * pn, plane_elems = (size_t)pn * pn;
    HIP_TRY(zero_slabs(w.slab, pc, G, pp.slabs, slab_plane, st));
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
            hipLaunchKernelGGL(k_slab_reduce, dim3((pn + 31) / 32, (pn + 31) / 32, pc), dim3(1024), 0, st,
                               w.slab, dst, pn, g.nt * 4, pp.slabs, G);
            HIP_TRY(hipGetLastError());
            HIP_TRY(zero_slabs(w.slab, pc, G, pp.slabs, slab_plane, st));
            since_flush = 0;
            fresh = true;
        }
    }
    hipLaunchKernelGGL(k_slab_reduce, dim3((pn + 31) / 32, (pn + 31) / 32, pc), dim3(1024), 0, st,
                       w.slab, dst, pn, g.nt * 4, pp.slabs, G);
    HIP_TRY(hipGetLastError());
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
        const int chunks = (int)(S < GAM_CHUNKS ? S : GAM_CHUNKS);
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
                              int pn, int N, float* out, const Workspace& w, const Knobs& kn, const SizeOps* ops, hipStream_t st)
{
    int rc;
    AbbePlan pp;
    rc = plan_abbe(pp, w, kn, pl, pn, N, planes);
    if (rc) return rc;

    // Coarse-grid path (N = 2 pn, pupil inside the natural box, no wrapping shift): the source-point loop runs
    // pn-point transforms on the grid q = 2 v (half the arithmetic per transformed line), the fine image is
    // reconstructed once per plane.  Needs the full-output wave kernels of size pn, empty box corners and box-edge
    // supports of at most EDGE_MAX samples; anything else takes the direct path below.
    AbbePlan pc_plan;
    const SizeOps* ops_c = nullptr;
    EdgeGeom eg;
    // The reconstruction is a fixed cost per call and plane (about ten small launches: 0.08 ms at 256^2 .. 0.5 ms at 4096^2
    // since round 4, when k_nyquist_reduce stopped taking 0.27 ms by itself), so short source lists stay on the direct path.
    // Break-even measured (scripts/coarse_breakeven.py, whole-call time, profiles/r04_coarse_breakeven.txt): S = 2,900 (256^2),
    // 400 (512^2), 180 (1024^2), 64 (2048^2), 48 (4096^2); thresholds a notch above.  LITHO_ABBE_COARSE = 2 ignores S.
    const int64_t s_min = pn == 256 ? 3072 : pn == 512 ? 512 : pn == 1024 ? 256 : pn == 2048 ? 96 : 64;
    bool coarse = kn.coarse && (kn.coarse >= 2 || S >= s_min) && coarse_eligible(pn, N) && pp.variant == 1 && pp.natural_box &&
                  pl[13] == 0;
    if (coarse) {
        eg.pn = pn; eg.c = pn / 2; eg.h = pn / 4;
        eg.lo[0] = pl[10] >= pl[9] ? pl[9] : 0;   eg.len[0] = pl[10] >= pl[9] ? pl[10] - pl[9] + 1 : 0;
        eg.lo[1] = pl[12] >= pl[11] ? pl[11] : 0; eg.len[1] = pl[12] >= pl[11] ? pl[12] - pl[11] + 1 : 0;
        ops_c = size_ops(ilog2(pn));
        coarse = ops_c && eg.len[0] <= EDGE_MAX && eg.len[1] <= EDGE_MAX &&
                 plan_abbe(pc_plan, w, kn, pl, pn, pn, planes) == LITHO_OK && pc_plan.variant == 0 && pc_plan.wave_y &&
                 pc_plan.PC <= COARSE_PLANES;
    }
    const AbbePlan& run = coarse ? pc_plan : pp;
    const size_t plane_elems = (size_t)pn * pn;
    int64_t nx = 0;
    // profiling: ONE event per kernel-class boundary (E0 x E1 y E2 x E3 ...); consecutive events bracket the
    // launches of one pass over one batch.  (Two events recorded back to back alias on ROCm, so no begin/end pairs.)
    MarkList marks(g_profiling != 0, st);
    if (coarse) hipLaunchKernelGGL(k_twiddle_table, dim3((pn + 255) / 256), dim3(256), 0, st, w.twtab2, pn);
    for (int p0 = 0; p0 < planes; p0 += run.PC) {
        const int pc = (planes - p0 < run.PC) ? planes - p0 : run.PC;
        const float2* Pc = P + (size_t)p0 * plane_elems;
        if (!coarse) {
            rc = accumulate_chunk(pp, ops, w, w.twtab, M, Pc, pc, shifts, S, pn, out + (size_t)p0 * plane_elems, st, marks, nx);
            if (rc) return rc;
            continue;
        }
        HIP_TRY(zero_async(w.ic, (size_t)pc * plane_elems * sizeof(float), st));
        rc = accumulate_chunk(pc_plan, ops_c, w, w.twtab2, M, Pc, pc, shifts, S, pn, w.ic, st, marks, nx);
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

static constexpr int64_t SPLIT_MIN_POINTS = 256;          // below that the two extra launches and the read-back cost more than they save
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
    int pe = kn.embed ? embedded_size(pn, N) : pn;
    if (pe != pn && ws_bytes < workspace_bytes_at(pe, N) + embed_extra_bytes(pe)) pe = pn;

    if (kn.poison) {
        // test knob: every scratch region starts the call as NaN bit patterns -- a kernel that reads scratch it (or an
        // earlier launch of THIS call) has not written turns the image into NaN (tests/test_gpu_abbe.py)
        unsigned char* lo = (unsigned char*)ws + 256 + align_up((size_t)N * sizeof(float2), 256);       // behind plan words + twiddle table
        unsigned char* hi = pe != pn ? (unsigned char*)ws + workspace_bytes_at(pe, N) + embed_extra_bytes(pe) : (unsigned char*)w.T + w.t_bytes;
        if ((unsigned char*)w.T + w.t_bytes > hi) hi = (unsigned char*)w.T + w.t_bytes;
        HIP_TRY(hipMemsetAsync(lo, 0xFF, (size_t)(hi - lo), st));
    }
    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    int pl[PLAN_WORDS];
    // (record word 14 = the grid the edge words were looked up for: a record made with the embedding off, or with a smaller
    // workspace, is not reused for an embedded run and vice versa)
    const bool from_record = reuse && reuse->valid == 1 && reuse->pn == pn && reuse->N == N && reuse->planes == planes &&
                             reuse->words[8] <= S && reuse->words[14] == pe;
    if (from_record) {
        for (int i = 0; i < PLAN_WORDS; ++i) pl[i] = reuse->words[i];
    } else {
        hipLaunchKernelGGL(k_plan_init, dim3(1), dim3(64), 0, st, w.plan);
        hipLaunchKernelGGL(k_pupil_box, dim3((pn + BOX_ROWS_PER_BLOCK - 1) / BOX_ROWS_PER_BLOCK, planes), dim3(256), 0, st,
                           P, pn, w.plan, pn / 2 - pe / 4, pn / 2 + pe / 4);
        hipLaunchKernelGGL(k_shift_extents, dim3(256), dim3(256), 0, st, shifts, (long long)S, count_dev, w.plan);
        HIP_TRY(hipGetLastError());
        rc = read_plan(w, pl, st);                           // the ONE host wait of the image path
        if (rc) return rc;
        if (reuse) {
            for (int i = 0; i < 16; ++i) reuse->words[i] = i < PLAN_WORDS ? pl[i] : 0;
            reuse->words[14] = pe;
            reuse->pn = pn; reuse->N = N; reuse->planes = planes; reuse->valid = 1;
        }
    }
    g_last_plan[15] = from_record ? 1 : 0;
    S = pl[8];                                               // = S, or the device-side count of the source list
    if (count_out) *count_out = S;
    if (S == 0 || pl[1] < pl[0]) return LITHO_OK;            // no source point / pupil identically zero: nothing to add

    // Embedded evaluation -- unless a shift wraps the pupil around the CALLER's grid: the reference rolls modulo its own size
    // (imageformation.py:63), which the padded grid would not reproduce; such a list runs the general path at this size.
    const bool nowrap = pl[0] + pl[4] >= 0 && pl[1] + pl[5] <= pn - 1 && pl[2] + pl[6] >= 0 && pl[3] + pl[7] <= pn - 1;
    if (!nowrap && !kn.force_general && kn.split && !from_record && (S >= SPLIT_MIN_POINTS || kn.split >= 2)) {
        // Some shift wraps the pupil around the grid -- usually for a minority of the points of a shifted source.  Split the
        // list (stable, on the device; one more 40-byte read-back) and give each part the path it needs: this function
        // again for the non-wrapping points (pruned box, coarse grid, embedding: 2.5 us per point at 1024^2), the
        // general path for the others (10 us).  The two lists live at the end of the T region, which shrinks by them.
        // (absolute placement: the lists end where the T region of the grid the non-wrapping part runs at ends -- the
        // padded grid's for an embedded size -- and BOTH carves' T regions are cut short of them)
        const size_t list_bytes = align_up((size_t)S * 2 * sizeof(int), 256);
        const size_t t_end = pe != pn ? workspace_bytes_at(pe, N) : workspace_bytes_at(pn, N);
        const size_t list_start = t_end - 2 * list_bytes;
        const size_t t0_own = (size_t)((unsigned char*)w.T - (unsigned char*)ws);
        const size_t t0_pad = pe != pn ? workspace_bytes_at(pe, N) - t_budget(pe) : t0_own;
        const size_t room = (size_t)64 << 20;
        if (2 * list_bytes < t_end && list_start > t0_own + room && list_start > t0_pad + room) {
            Workspace ws_split = w;
            if (t0_own + ws_split.t_bytes > list_start) ws_split.t_bytes = list_start - t0_own;
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
            int sw[10];
            HIP_TRY(hipMemcpyAsync(sw, words, sizeof(sw), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
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
                else rc = accumulate_planned(M, P, planes, lst, n, plp, pn, N, out, ws_split, kn, ops, st);
                if (rc) return rc;
                launches += g_last_plan[6];
            }
            g_last_plan[6] = launches;                                    // (the other fields describe the part that ran last)
            g_last_plan[15] = 2;                                          // = the source list was split
            return LITHO_OK;
        }
    }
    if (pe == pn || !nowrap || kn.force_general) return accumulate_planned(M, P, planes, shifts, S, pl, pn, N, out, w, kn, ops, st);
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
    unsigned char* extra = (unsigned char*)ws + workspace_bytes_at(pe, N);
    const size_t e2 = (size_t)pe * pe;
    float2* M2 = (float2*)extra;
    float2* P2 = (float2*)(extra + align_up(e2 * sizeof(float2), 256));
    float* O2 = (float*)((unsigned char*)P2 + align_up(COARSE_PLANES * e2 * sizeof(float2), 256));
    const int off = (pe - pn) / 2;
    int pl2[PLAN_WORDS];
    for (int i = 0; i < PLAN_WORDS; ++i) pl2[i] = pl[i];
    for (int i = 0; i < 4; ++i) pl2[i] += off;                                // the pupil's support box, in the padded grid
    if (pl[10] >= pl[9]) { pl2[9] += off; pl2[10] += off; }                   // its samples on the padded grid's natural-box edges
    if (pl[12] >= pl[11]) { pl2[11] += off; pl2[12] += off; }
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
    hipLaunchKernelGGL(k_plan_init, dim3(1), dim3(64), 0, st, w.plan);
    hipLaunchKernelGGL(k_pupil_box, dim3((pn + BOX_ROWS_PER_BLOCK - 1) / BOX_ROWS_PER_BLOCK, 1), dim3(256), 0, st, pf, pn, w.plan,
                       pn / 2 - pn / 4, pn / 2 + pn / 4);
    HIP_TRY(hipGetLastError());
    int pl[PLAN_WORDS];
    rc = read_plan(w, pl, st);
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
