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
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/litho_abbe.h"
#include "fft_core.hpp"
#include "engine_common.hpp"

namespace litho {

// ----------------------------------------------------------------------------------
// geometry shared by the pass kernels
// ----------------------------------------------------------------------------------
struct PassGeom {
    int pn, c, N;
    int nt;                 // column tiles of 4 (ceil(pn/4))
    int kx0, kx1;           // x-pass: valid input window [kx0,kx1) in centred coordinates
    int ky0, ky1;           // y-pass: valid input window = rows of T; a = k - ky0
    int rows;               // number of T rows (= ky1 - ky0)
    int general;            // 1: roll stays on P, modular gather (wrapping shifts)
    long long t_point;      // float2 elements of T per source point = nt*rows*4
};

template <int LOG2N>
struct Launch {
    using F = LineFFT<LOG2N, +1>;
    static constexpr int L = (F::T >= 64) ? 1 : 64 / F::T;       // lines per workgroup
    static constexpr int THREADS = F::T * L;
    static constexpr int NBUF = (LOG2N <= 12) ? 2 : 1;
    static constexpr size_t LDS_BYTES = sizeof(float2) * (size_t)L * NBUF * F::LDS_LINE;
    // launch_bounds second argument = waves per SIMD we want resident: two workgroups per CU
    // up to N = 4096 (256 threads each), one above.
    static constexpr int WAVES = (THREADS / 256 > 0 ? THREADS / 256 : 1) * (LOG2N <= 12 ? 2 : 1);
};

// ----------------------------------------------------------------------------------
// buffer addressing: 32-bit offsets, and the hardware range check is the zero-padding
// predicate (an offset >= num_records loads 0 / drops the store).
// ----------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
static constexpr unsigned BUF_OOB = 0xFFFF0000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load_c64(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
__device__ __forceinline__ void buf_store_c64(__amdgpu_buffer_rsrc_t r, unsigned off, float2 v) {
    u32x2 w;
    w.x = __float_as_uint(v.x);
    w.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, off, 0, 0);
}

// ----------------------------------------------------------------------------------
// x-pass, pruned mode (no wrapping shift): A = P[box] * M[box + shift]
// ----------------------------------------------------------------------------------
template <int LOG2N>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_xpass_abbe(
    const float2* __restrict__ P, const float2* __restrict__ M, const int* __restrict__ shifts,
    float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, +1>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt);

    const int s = blockIdx.y;
    const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
    const int a = blockIdx.x * LC::L + lg;
    const bool active = a < g.rows;
    const int r = g.ky0 + g.c + a;                            // row of P inside its support box
    const unsigned prow = (unsigned)r * g.pn + g.c;           // element offset of centred column 0
    const unsigned mrow = (unsigned)(r + dy) * g.pn + g.c + dx;   // same window of M moved by the shift
    const size_t plane_bytes = (size_t)g.pn * g.pn * sizeof(float2);
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(P, plane_bytes);
    const __amdgpu_buffer_rsrc_t rM = make_rsrc(M, plane_bytes);

    float2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int k;
        const bool ok = centred_index(lt + F::T * e, F::N, g.kx0, g.kx1, k) && active;
        const float2 pv = buf_load_c64(rP, ok ? (prow + k) * 8u : BUF_OOB);
        const float2 mv = buf_load_c64(rM, ok ? (mrow + k) * 8u : BUF_OOB);
        x[e] = cmul(pv, mv);
    }
    int flip = 0;
    F::template run<LC::NBUF>(x, tw, lds, lt, flip);

    const __amdgpu_buffer_rsrc_t rT = make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        int u;
        const bool ok = centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u) && active;
        const unsigned q = (unsigned)(u + g.c);
        buf_store_c64(rT, ok ? (((q >> 2) * g.rows + a) * 4u + (q & 3u)) * 8u : BUF_OOB, x[m]);
    }
}

// ----------------------------------------------------------------------------------
// generic x-pass (any loader): rows -> T[s][tile][row][4]
// ----------------------------------------------------------------------------------
template <int LOG2N, int SIGN, typename Loader>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_xpass(
    Loader ld, float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, SIGN>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt);

    const int s = blockIdx.y;
    const int a = blockIdx.x * LC::L + lg;
    const bool active = a < g.rows;
    ld.begin_line(s, active ? a : 0, g);

    float2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int k;
        const bool ok = centred_index(lt + F::T * e, F::N, g.kx0, g.kx1, k) && active;
        x[e] = ok ? ld.load(k, g) : make_float2(0.f, 0.f);
    }
    int flip = 0;
    F::template run<LC::NBUF>(x, tw, lds, lt, flip);

    float2* trow = Tbuf + (size_t)s * g.t_point + (size_t)a * 4;
    const size_t tile_stride = (size_t)g.rows * 4;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        int u;
        if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u) && active) {
            const int q = u + g.c;
            trow[(size_t)(q >> 2) * tile_stride + (q & 3)] = x[m];
        }
    }
}

// Loader: Abbe product P*M for source point s (imageformation.py:34 and :63).
struct AbbeLoader {
    const float2* P;
    const float2* M;
    const int* shifts;       // (dy,dx) pairs of this batch
    const float2* prow;
    const float2* mrow;
    int dx, pr_general;
    __device__ __forceinline__ void begin_line(int s, int a, const PassGeom& g) {
        const int dy = shifts[2 * s];
        dx = shifts[2 * s + 1];
        if (!g.general) {
            const int r = g.ky0 + g.c + a;                  // row of P (support box row)
            prow = P + (size_t)r * g.pn;
            mrow = M + (size_t)(r + dy) * g.pn + dx;        // same window of M, moved by the shift
        } else {
            int r = (a - dy) % g.pn;                        // torch.roll: A[i] = P[(i - d) mod pn]
            if (r < 0) r += g.pn;
            prow = P + (size_t)r * g.pn;
            mrow = M + (size_t)a * g.pn;
        }
    }
    __device__ __forceinline__ float2 load(int k, const PassGeom& g) const {
        const int col = k + g.c;
        if (!g.general) return cmul(prow[col], mrow[col]);
        int pc = (col - dx) % g.pn;
        if (pc < 0) pc += g.pn;
        return cmul(prow[pc], mrow[col]);
    }
};

// Loader: a real image (the bilinearly scaled mask, mask.py:76-81).  Line a / sample k of the
// padded N x N frame map to img[a + off][k - kx0 + off]; the zero padding (or, when the
// scaled mask is larger than N, the crop) is expressed by the window and `off` alone.
struct RealImageLoader {
    const float* img;        // [n,n]
    int n, off;
    const float* row;
    __device__ __forceinline__ void begin_line(int, int a, const PassGeom&) { row = img + (size_t)(a + off) * n + off; }
    __device__ __forceinline__ float2 load(int k, const PassGeom& g) const {
        return make_float2(row[k - g.kx0], 0.f);
    }
};

// ----------------------------------------------------------------------------------
// y-pass with |E|^2 accumulation in registers over the batch
// ----------------------------------------------------------------------------------
// RL = log2(N/pn) when pn is a power of two (valid output bins are then the same 16>>RL
// registers for every thread), -1 = any even pn (all 16 kept, predicated at the flush).
template <int RL>
struct OutSel {
    static constexpr int NV = (RL < 0) ? 16 : (16 >> RL);
    __device__ static constexpr int m_of(int iv) { return (RL <= 0) ? iv : (iv < NV / 2 ? iv : 16 - NV + iv); }
};

template <int LOG2N, int RL>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_ypass_acc(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G)
{
    using F = LineFFT<LOG2N, +1>;
    using LC = Launch<LOG2N>;
    using OS = OutSel<RL>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt);

    const int tile = blockIdx.x * LC::L + lg;
    const bool active = tile < g.nt;
    const int grp = blockIdx.y;

    float acc[4][OS::NV];
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx)
#pragma unroll
        for (int iv = 0; iv < OS::NV; ++iv) acc[cidx][iv] = 0.f;

    // per-thread input map: sample e <-> byte offset of T row a_e inside this tile (or out of range)
    unsigned voff[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int k;
        const bool ok = centred_index(lt + F::T * e, F::N, g.ky0, g.ky1, k) && active;
        voff[e] = ok ? ((unsigned)tile * g.rows + (unsigned)(k - g.ky0)) * 32u : BUF_OOB;
    }

    int flip = 0;
    for (int s = grp; s < nb; s += G) {
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
#pragma unroll
        for (int cidx = 0; cidx < 4; ++cidx) {
            float2 x[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e] = buf_load_c64(rT, voff[e] + cidx * 8u);
            F::template run<LC::NBUF>(x, tw, lds, lt, flip);
#pragma unroll
            for (int iv = 0; iv < OS::NV; ++iv) {
                const float2 v = x[OS::m_of(iv)];
                acc[cidx][iv] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[cidx][iv]));
            }
        }
    }

    if (!active) return;
    // flush into this group's private slab, laid out [G][qx][qy] (qy contiguous -> coalesced)
#pragma unroll
    for (int cidx = 0; cidx < 4; ++cidx) {
        const int qx = tile * 4 + cidx;
        if (qx >= g.pn) continue;
        float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
#pragma unroll
        for (int iv = 0; iv < OS::NV; ++iv) {
            int u;
            if (centred_index(lt + F::T * OS::m_of(iv), F::N, -g.c, g.pn - g.c, u)) srow[u + g.c] += acc[cidx][iv];
        }
    }
}

// y-pass that writes the complex field instead (calculateFFTAerial, mask spectrum).
template <int LOG2N, int SIGN>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_ypass_field(
    const float2* __restrict__ Tbuf, float2* __restrict__ field, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, SIGN>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;
    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt);
    const int tile = blockIdx.x * LC::L + lg;
    const bool active = tile < g.nt;
    const float2* tt = Tbuf + (size_t)(active ? tile : 0) * g.rows * 4;
    int flip = 0;
    for (int cidx = 0; cidx < 4; ++cidx) {
        float2 x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.ky0, g.ky1, k) && active;
            x[e] = ok ? tt[(size_t)(k - g.ky0) * 4 + cidx] : make_float2(0.f, 0.f);
        }
        F::template run<LC::NBUF>(x, tw, lds, lt, flip);
        const int qx = tile * 4 + cidx;
        if (!active || qx >= g.pn) continue;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            int u;
            if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u)) field[(size_t)(u + g.c) * g.pn + qx] = x[m];
        }
    }
}

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
//             [4]=min dy,[5]=max dy,[6]=min dx,[7]=max dx
__global__ void k_plan_init(int* plan)
{
    if (threadIdx.x < 8) plan[threadIdx.x] = (threadIdx.x & 1) ? INT_MIN : INT_MAX;
}

__global__ void k_pupil_box(const float2* __restrict__ P, int pn, int planes, int* plan)
{
    const int row = blockIdx.x;
    int cmin = INT_MAX, cmax = INT_MIN;
    for (int p = 0; p < planes; ++p) {
        const float2* r = P + ((size_t)p * pn + row) * pn;
        for (int cidx = threadIdx.x; cidx < pn; cidx += blockDim.x) {
            const float2 v = r[cidx];
            if (v.x != 0.f || v.y != 0.f) { cmin = min(cmin, cidx); cmax = max(cmax, cidx); }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cmin = min(cmin, __shfl_xor(cmin, off));
        cmax = max(cmax, __shfl_xor(cmax, off));
    }
    if ((threadIdx.x & 63) == 0 && cmax >= 0) {
        atomicMin(&plan[0], row); atomicMax(&plan[1], row);
        atomicMin(&plan[2], cmin); atomicMax(&plan[3], cmax);
    }
}

__global__ void k_shift_extents(const int* __restrict__ shifts, long long S, int* plan)
{
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

// out[qy][qx] += sum_g slab[g][qx][qy]   (32x32 tiles through LDS)
__global__ void k_slab_reduce(const float* __restrict__ slab, float* __restrict__ out, int pn, int ldq, int G)
{
    __shared__ float tile[32][33];
    const int qx0 = blockIdx.x * 32, qy0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;     // 256 threads: ty in 0..7
    for (int j = ty; j < 32; j += 8) {
        const int qx = qx0 + j, qy = qy0 + tx;
        float v = 0.f;
        if (qx < pn && qy < pn)
            for (int gidx = 0; gidx < G; ++gidx) v += slab[((size_t)gidx * ldq + qx) * pn + qy];
        tile[j][tx] = v;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int qy = qy0 + j, qx = qx0 + tx;
        if (qx < pn && qy < pn) out[(size_t)qy * pn + qx] += tile[tx][j];
    }
}

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
struct Workspace {
    int* plan;          // 64 ints
    float2* twtab;      // N
    float* slab;        // G_MAX * nt*4 * pn
    float2* T;          // remainder
    size_t t_bytes;
};

static constexpr int G_MAX = 8;
static constexpr size_t T_BUDGET_MAX = (size_t)1 << 30;

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static size_t t_budget(int pn)
{
    const size_t nt = (pn + 3) / 4;
    const size_t one_general = nt * (size_t)pn * 4 * sizeof(float2);
    size_t b = 64 * one_general;
    if (b > T_BUDGET_MAX) b = T_BUDGET_MAX;
    if (b < one_general + one_general / 2) b = one_general + one_general / 2;
    return b;
}

static size_t workspace_bytes(int pn, int N)
{
    const size_t nt = (pn + 3) / 4;
    size_t b = 256;
    b += align_up((size_t)N * sizeof(float2), 256);
    b += align_up((size_t)G_MAX * nt * 4 * pn * sizeof(float), 256);
    b += align_up(t_budget(pn), 256);
    return b;
}

static bool carve(void* ws, size_t bytes, int pn, int N, Workspace& w)
{
    if (!ws || bytes < workspace_bytes(pn, N)) return false;
    const size_t nt = (pn + 3) / 4;
    unsigned char* p = (unsigned char*)ws;
    w.plan = (int*)p; p += 256;
    w.twtab = (float2*)p; p += align_up((size_t)N * sizeof(float2), 256);
    w.slab = (float*)p; p += align_up((size_t)G_MAX * nt * 4 * pn * sizeof(float), 256);
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

static thread_local int64_t g_last_plan[8] = {0, 0, 0, 0, 0, 0, 0, 0};

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's
// roofline leg).  Off by default: the events serialise nothing but cost host time.
static thread_local int g_profiling = 0;
static thread_local double g_profile[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // x ms, x launches, x points, y ms, y launches, y points, -, -
struct EventPair { hipEvent_t a, b; int kind; int nb; };

template <int LOG2N, int SIGN, typename Loader>
static hipError_t launch_xpass(const Loader& ld, float2* T, const float2* tw, const PassGeom& g, int nb, hipStream_t st)
{
    using LC = Launch<LOG2N>;
    dim3 grid((g.rows + LC::L - 1) / LC::L, nb);
    auto kern = k_xpass<LOG2N, SIGN, Loader>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, ld, T, tw, g);
    return hipGetLastError();
}

template <int LOG2N>
static hipError_t launch_xpass_abbe(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                                    const PassGeom& g, int nb, hipStream_t st)
{
    using LC = Launch<LOG2N>;
    dim3 grid((g.rows + LC::L - 1) / LC::L, nb);
    auto kern = k_xpass_abbe<LOG2N>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, P, M, shifts, T, tw, g);
    return hipGetLastError();
}

template <int LOG2N, int RL>
static hipError_t launch_ypass_acc_rl(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int G, hipStream_t st)
{
    using LC = Launch<LOG2N>;
    dim3 grid((g.nt + LC::L - 1) / LC::L, G);
    auto kern = k_ypass_acc<LOG2N, RL>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, T, slab, tw, g, nb, G);
    return hipGetLastError();
}

template <int LOG2N>
static hipError_t launch_ypass_acc(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int G, hipStream_t st)
{
    const bool pow2 = (g.pn & (g.pn - 1)) == 0;
    const int rl = pow2 ? (LOG2N - ilog2(g.pn)) : -1;
    switch (rl) {
        case 0: return launch_ypass_acc_rl<LOG2N, 0>(T, slab, tw, g, nb, G, st);
        case 1: return launch_ypass_acc_rl<LOG2N, 1>(T, slab, tw, g, nb, G, st);
        case 2: return launch_ypass_acc_rl<LOG2N, 2>(T, slab, tw, g, nb, G, st);
        case 3: return launch_ypass_acc_rl<LOG2N, 3>(T, slab, tw, g, nb, G, st);
        default: return launch_ypass_acc_rl<LOG2N, -1>(T, slab, tw, g, nb, G, st);
    }
}

template <int LOG2N, int SIGN>
static hipError_t launch_ypass_field(const float2* T, float2* field, const float2* tw, const PassGeom& g, hipStream_t st)
{
    using LC = Launch<LOG2N>;
    dim3 grid((g.nt + LC::L - 1) / LC::L);
    auto kern = k_ypass_field<LOG2N, SIGN>;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LC::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, T, field, tw, g);
    return hipGetLastError();
}

#define LITHO_DISPATCH_LOG2N(l2, CALL)                                         \
    switch (l2) {                                                              \
        case 4: { constexpr int L2 = 4; CALL; } break;                         \
        case 5: { constexpr int L2 = 5; CALL; } break;                         \
        case 6: { constexpr int L2 = 6; CALL; } break;                         \
        case 7: { constexpr int L2 = 7; CALL; } break;                         \
        case 8: { constexpr int L2 = 8; CALL; } break;                         \
        case 9: { constexpr int L2 = 9; CALL; } break;                         \
        case 10: { constexpr int L2 = 10; CALL; } break;                       \
        case 11: { constexpr int L2 = 11; CALL; } break;                       \
        case 12: { constexpr int L2 = 12; CALL; } break;                       \
        case 13: { constexpr int L2 = 13; CALL; } break;                       \
        case 14: { constexpr int L2 = 14; CALL; } break;                       \
        default: return LITHO_E_ARG;                                           \
    }

// Reads the 8 plan words back (one small synchronising copy).
static int read_plan(const Workspace& w, int host[8], hipStream_t st)
{
    HIP_TRY(hipMemcpyAsync(host, w.plan, 8 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LITHO_OK;
}

static void make_geom(PassGeom& g, int pn, int N, int r0, int c0, int h, int wdt, int general)
{
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (pn + 3) / 4;
    g.kx0 = c0 - g.c; g.kx1 = c0 + wdt - g.c;
    g.ky0 = r0 - g.c; g.ky1 = r0 + h - g.c;
    g.rows = h; g.general = general;
    g.t_point = (long long)g.nt * h * 4;
}

static int abbe_accumulate(const float2* M, const float2* P, int planes, const int* shifts, int64_t S,
                           int pn, int N, float* out, void* ws, size_t ws_bytes, hipStream_t st)
{
    int rc = check_sizes(pn, N);
    if (rc) return rc;
    if (!M || !P || !out || planes < 1 || S < 0 || (S > 0 && !shifts)) return LITHO_E_ARG;
    if (S == 0) return LITHO_OK;
    Workspace w;
    if (!carve(ws, ws_bytes, pn, N, w)) return LITHO_E_WORKSPACE;

    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    hipLaunchKernelGGL(k_plan_init, dim3(1), dim3(64), 0, st, w.plan);
    hipLaunchKernelGGL(k_pupil_box, dim3(pn), dim3(256), 0, st, P, pn, planes, w.plan);
    hipLaunchKernelGGL(k_shift_extents, dim3(256), dim3(256), 0, st, shifts, (long long)S, w.plan);
    HIP_TRY(hipGetLastError());
    int pl[8];
    rc = read_plan(w, pl, st);
    if (rc) return rc;
    if (pl[1] < pl[0]) return LITHO_OK;                      // pupil identically zero: nothing to add

    int r0 = pl[0], h = pl[1] - pl[0] + 1, c0 = pl[2], wdt = pl[3] - pl[2] + 1;
    const bool nowrap = (r0 + pl[4] >= 0) && (r0 + h - 1 + pl[5] <= pn - 1) &&
                        (c0 + pl[6] >= 0) && (c0 + wdt - 1 + pl[7] <= pn - 1);
    int general = (!nowrap || env_int("LITHO_ABBE_FORCE_GENERAL", 0)) ? 1 : 0;
    if (general) { r0 = 0; c0 = 0; h = pn; wdt = pn; }
    PassGeom g;
    make_geom(g, pn, N, r0, c0, h, wdt, general);

    const size_t point_bytes = (size_t)g.t_point * sizeof(float2);
    int64_t bs = (int64_t)(w.t_bytes / point_bytes);
    const int bs_env = env_int("LITHO_ABBE_BATCH", 0);
    if (bs_env > 0 && bs_env < bs) bs = bs_env;
    if (bs < 1) return LITHO_E_WORKSPACE;
    if (bs > 65535) bs = 65535;
    int G = env_int("LITHO_ABBE_GROUPS", 2);
    if (G < 1) G = 1;
    if (G > G_MAX) G = G_MAX;
    const size_t slab_plane = (size_t)g.nt * 4 * pn;
    const int l2 = ilog2(N);
    int64_t nx = 0, ny = 0;
    std::vector<EventPair> events;
    const size_t max_events = 4096;
    auto ev_begin = [&](int kind, int nb) {
        if (!g_profiling || events.size() >= max_events) return;
        EventPair e{nullptr, nullptr, kind, nb};
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) return;
        (void)hipEventRecord(e.a, st);
        events.push_back(e);
    };
    auto ev_end = [&]() {
        if (!g_profiling || events.empty() || events.size() > max_events) return;
        (void)hipEventRecord(events.back().b, st);
    };

    for (int p = 0; p < planes; ++p) {
        const float2* Pp = P + (size_t)p * pn * pn;
        HIP_TRY(hipMemsetAsync(w.slab, 0, (size_t)G * slab_plane * sizeof(float), st));
        for (int64_t s0 = 0; s0 < S; s0 += bs) {
            const int nb = (int)((S - s0 < bs) ? (S - s0) : bs);
            ev_begin(0, nb);
            if (general) {
                AbbeLoader ld{Pp, M, shifts + 2 * s0, nullptr, nullptr, 0, 0};
                LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_xpass<L2, +1, AbbeLoader>(ld, w.T, w.twtab, g, nb, st))));
            } else {
                LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_xpass_abbe<L2>(Pp, M, shifts + 2 * s0, w.T, w.twtab, g, nb, st))));
            }
            ev_end();
            const int Geff = nb < G ? nb : G;
            ev_begin(1, nb);
            LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_ypass_acc<L2>(w.T, w.slab, w.twtab, g, nb, Geff, st))));
            ev_end();
            ++nx; ++ny;
        }
        hipLaunchKernelGGL(k_slab_reduce, dim3((pn + 31) / 32, (pn + 31) / 32), dim3(256), 0, st,
                           w.slab, out + (size_t)p * pn * pn, pn, g.nt * 4, G);
        HIP_TRY(hipGetLastError());
    }
    if (g_profiling) {
        for (int i = 0; i < 8; ++i) g_profile[i] = 0;
        if (!events.empty()) (void)hipEventSynchronize(events.back().b);
        for (auto& e : events) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
                g_profile[e.kind * 3 + 0] += ms;
                g_profile[e.kind * 3 + 1] += 1;
                g_profile[e.kind * 3 + 2] += e.nb;
            }
            (void)hipEventDestroy(e.a);
            (void)hipEventDestroy(e.b);
        }
    }
    g_last_plan[0] = general; g_last_plan[1] = r0; g_last_plan[2] = c0; g_last_plan[3] = h;
    g_last_plan[4] = wdt; g_last_plan[5] = bs; g_last_plan[6] = nx; g_last_plan[7] = ny;
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
    hipLaunchKernelGGL(k_pupil_box, dim3(pn), dim3(256), 0, st, pf, pn, 1, w.plan);
    HIP_TRY(hipGetLastError());
    int pl[8];
    rc = read_plan(w, pl, st);
    if (rc) return rc;
    if (pl[1] < pl[0]) {                                      // zero pupil -> zero field
        HIP_TRY(hipMemsetAsync(field, 0, (size_t)pn * pn * sizeof(float2), st));
        return LITHO_OK;
    }
    PassGeom g;
    make_geom(g, pn, N, pl[0], pl[2], pl[1] - pl[0] + 1, pl[3] - pl[2] + 1, 0);
    HIP_TRY(hipMemsetAsync(w.plan + 16, 0, 2 * sizeof(int), st));     // a (0,0) shift
    AbbeLoader ld{pf, M, w.plan + 16, nullptr, nullptr, 0, 0};
    const int l2 = ilog2(N);
    LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_xpass<L2, +1, AbbeLoader>(ld, w.T, w.twtab, g, 1, st))));
    LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_ypass_field<L2, +1>(w.T, field, w.twtab, g, st))));
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
    if ((size_t)ns * ns * sizeof(float) > (size_t)G_MAX * nt * 4 * pn * sizeof(float)) return LITHO_E_WORKSPACE;
    if (nt * (size_t)(j1 - j0) * 4 * sizeof(float2) > w.t_bytes) return LITHO_E_WORKSPACE;
    float* scaled = w.slab;                                      // the slab region is free here
    hipLaunchKernelGGL(k_twiddle_table, dim3((N + 255) / 256), dim3(256), 0, st, w.twtab, N);
    launch_scale_mask(geo, pn, ns, eps, scaled, st);
    HIP_TRY(hipGetLastError());
    PassGeom g;
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (int)nt;
    g.kx0 = j0 - N / 2; g.kx1 = j1 - N / 2;
    g.ky0 = g.kx0; g.ky1 = g.kx1;
    g.rows = j1 - j0; g.general = 0;
    g.t_point = (long long)nt * g.rows * 4;
    RealImageLoader ld{scaled, ns, j0 - pW, nullptr};
    const int l2 = ilog2(N);
    LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_xpass<L2, -1, RealImageLoader>(ld, w.T, w.twtab, g, 1, st))));
    LITHO_DISPATCH_LOG2N(l2, HIP_TRY((launch_ypass_field<L2, -1>(w.T, spec, w.twtab, g, st))));
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

int litho_abbe_accumulate(const void* maskFT, const void* pupil, int planes, const int32_t* shifts, int64_t S,
                          int pn, int N, float* out, void* workspace, size_t workspace_bytes, void* stream)
{
    return litho::abbe_accumulate((const float2*)maskFT, (const float2*)pupil, planes, shifts, S, pn, N, out,
                                  workspace, workspace_bytes, (hipStream_t)stream);
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

int litho_abbe_last_plan(int64_t fields_host[8])
{
    if (!fields_host) return LITHO_E_ARG;
    memcpy(fields_host, litho::g_last_plan, sizeof(litho::g_last_plan));
    return LITHO_OK;
}

}  // extern "C"
