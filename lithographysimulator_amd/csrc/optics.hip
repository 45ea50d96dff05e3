// optics.hip -- source sampling, source-list compaction, Zernike pupil and post-process
// kernels.  This file is compiled with -ffp-contract=off: the reference keeps its sigma
// grids, radii, angles and Zernike sums in fp16 tensors, torch-CPU evaluates every fp16 op
// in fp32 and rounds once, and reproducing that needs each fp32 operation to stay a
// separate, individually rounded instruction (h() = round to fp16 and back).
//
// Transcendentals (atan2, cos, sin, integer powers) are evaluated in double and rounded to
// fp32, i.e. correctly rounded fp32; torch-CPU's SLEEF kernels are within 1 ulp of that, and
// the following rounding to fp16 hides the difference except on a handful of pixels per
// million (tolerance stated in tests/test_gpu_optics.py).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/litho_abbe.h"
#include "engine_common.hpp"

namespace litho {

__device__ __forceinline__ float h16(float v) { return __half2float(__float2half_rn(v)); }
static float h16_host(float v) { return __half2float(__float2half_rn(v)); }

// torch.arange(start, end, step, dtype=float16)[i] on CPU (lightsource.py:39-40,
// pupil.py:53): filled 16 lanes at a time, chunk base rounded to fp16 first; a tail
// shorter than 16 is h(start + step*i).
__device__ __forceinline__ float sigma_axis(int i, int n, float fs, float fst)
{
    const int full = (n / 16) * 16;
    if (i < full) {
        const int i0 = (i / 16) * 16;
        const float base = h16(fs + fst * (float)i0);
        return h16(base + (float)(i - i0) * fst);
    }
    return h16(fs + fst * (float)i);
}

__device__ __forceinline__ float radius16(float x, float y)
{
    return h16(__fsqrt_rn(h16(h16(x * x) + h16(y * y))));
}

struct SourceParams {
    int kind, pn, count;
    float fsx, fsy, fst;           // axis start (x, y) and step as fp32
    float sin16, sout16;           // thresholds rounded to fp16
    float rot16, twopi16;
    float lo16[64], hi16[64];      // wedge bounds rounded to fp16 (count <= 64)
};

// lightsource.py:34-50 (annular) and :52-73 (quasar)
__global__ void k_source_bitmap(SourceParams sp, int64_t* __restrict__ bitmap)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= sp.pn) return;
    const float x = sigma_axis(col, sp.pn, sp.fsx, sp.fst);
    const float y = sigma_axis(row, sp.pn, sp.fsy, sp.fst);
    const float O = radius16(x, y);
    bool lit = (O >= sp.sin16) && (O <= sp.sout16);
    if (sp.kind == 1) {
        float th = h16(h16((float)atan2((double)y, (double)x)) + sp.rot16);
        float r = fmodf(th, sp.twopi16);                       // torch.remainder: sign of the divisor
        if (r != 0.f && ((sp.twopi16 < 0.f) != (r < 0.f))) r += sp.twopi16;
        th = h16(r);
        for (int gap = 0; gap < sp.count; ++gap)
            if (sp.lo16[gap] < th && th < sp.hi16[gap]) lit = false;
    }
    bitmap[(size_t)row * sp.pn + col] = lit ? 1 : 0;
}

// ---- row-major ordered compaction of a bitmap into (dy,dx) pairs (imageformation.py:59)
__global__ void k_row_counts(const int64_t* __restrict__ bitmap, int pn, int* __restrict__ counts)
{
    const int row = blockIdx.x;
    int cnt = 0;
    for (int c = threadIdx.x; c < pn; c += blockDim.x) cnt += bitmap[(size_t)row * pn + c] != 0;
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) counts[row] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of counts[0..pn) in place, total into counts[pn]; single block of 1024.
__global__ void k_row_scan(int* __restrict__ counts, int pn)
{
    __shared__ int sums[1024];
    const int per = (pn + 1023) / 1024;
    const int b = threadIdx.x * per;
    int local = 0;
    for (int i = 0; i < per; ++i) if (b + i < pn) local += counts[b + i];
    sums[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        int v = (threadIdx.x >= off) ? sums[threadIdx.x - off] : 0;
        __syncthreads();
        sums[threadIdx.x] += v;
        __syncthreads();
    }
    int run = sums[threadIdx.x] - local;
    for (int i = 0; i < per; ++i)
        if (b + i < pn) { const int cnt = counts[b + i]; counts[b + i] = run; run += cnt; }
    if (threadIdx.x == 1023) counts[pn] = sums[1023];
}

__global__ void k_row_write(const int64_t* __restrict__ bitmap, int pn, const int* __restrict__ offsets,
                            int32_t* __restrict__ shifts, long long capacity)
{
    const int row = blockIdx.x;          // one wave per row keeps the order trivially
    const int lane = threadIdx.x;
    long long base = offsets[row];
    const int c = pn / 2;
    for (int c0 = 0; c0 < pn; c0 += 64) {
        const int col = c0 + lane;
        const bool lit = col < pn && bitmap[(size_t)row * pn + col] != 0;
        const unsigned long long m = __ballot(lit);
        if (lit) {
            const long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < capacity) { shifts[2 * pos] = row - c; shifts[2 * pos + 1] = col - c; }
        }
        base += __popcll(m);
    }
}

// The same write with the scan folded in (pn <= ROW_SUM_MAX): the wave of row r sums the raw counts of the rows in front of it
// itself (at most 64 strided loads per lane) -- two launches per compaction instead of three; `counts` stays raw, the last row
// leaves the total in counts[pn] (litho_abbe_accumulate_counted's count_dev).
static constexpr int ROW_SUM_MAX = 4096;
__global__ void k_row_write_sum(const int64_t* __restrict__ bitmap, int pn, int* __restrict__ counts,
                                int32_t* __restrict__ shifts, long long capacity)
{
    const int row = blockIdx.x;
    const int lane = threadIdx.x;
    int before = 0;
    for (int r = lane; r < row; r += 64) before += counts[r];
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off);
    long long base = before;
    const int c = pn / 2;
    for (int c0 = 0; c0 < pn; c0 += 64) {
        const int col = c0 + lane;
        const bool lit = col < pn && bitmap[(size_t)row * pn + col] != 0;
        const unsigned long long m = __ballot(lit);
        if (lit) {
            const long long pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < capacity) { shifts[2 * pos] = row - c; shifts[2 * pos + 1] = col - c; }
        }
        base += __popcll(m);
    }
    if (row == pn - 1 && lane == 0) counts[pn] = (int)base;
}

// ---- pupil (pupil.py:46-111)
struct ZTerm {
    int m, n, nk;
    float cN;              // h(coeff * fp32(+-Nmn))
    float coef[10];        // radial static coefficients as fp32 (pupil.py:63); nk = (n-|m|)/2 + 1 <= 10
};

__device__ __forceinline__ float pow_int_rn(float r, int p)
{
    double v = 1.0;
    for (int i = 0; i < p; ++i) v *= (double)r;
    return (float)v;
}

// Up to PACK_TERMS Zernike terms travel by value in the kernel-argument segment (no staging copy, no host wait);
// longer aberration vectors go through a staged device table.
static constexpr int PACK_TERMS = 32;
struct ZTermPack {
    ZTerm t[PACK_TERMS];
};

template <typename Terms>
__device__ __forceinline__ void pupil_body(const Terms& terms, int J, int pn, float fs, float fst, float twopi_f,
                                           uint16_t* __restrict__ wavefront, float2* __restrict__ pupil)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= pn) return;
    const float x = sigma_axis(col, pn, fs, fst);
    const float y = sigma_axis(row, pn, fs, fst);
    const float r = radius16(x, y);
    const float theta = h16((float)atan2((double)y, (double)x));
    float W = 0.f;
    for (int j = 0; j < J; ++j) {
        const ZTerm t = terms[j];
        float acc = 0.f;
        for (int k = 0; k < t.nk; ++k) acc = acc + h16(t.coef[k] * h16(pow_int_rn(r, t.n - 2 * k)));
        const float R = h16(acc);
        const float arg = h16((float)t.m * theta);
        const float trig = h16((float)(t.m >= 0 ? cos((double)arg) : sin((double)arg)));
        float Z = h16(h16(t.cN * R) * trig);
        if (!(r <= 1.f)) Z = 0.f;
        W = h16(W + Z);
    }
    const size_t idx = (size_t)row * pn + col;
    if (wavefront) wavefront[idx] = __half_as_ushort(__float2half_rn(W));
    if (pupil) {
        float2 phi = make_float2(0.f, 0.f);
        if (r <= 1.f) {
            const float ang = twopi_f * W;               // fp32(2 pi) * W, one rounding (pupil.py:103)
            phi = make_float2((float)cos((double)ang), (float)sin((double)ang));
        }
        pupil[idx] = phi;
    }
}

__global__ void k_pupil(const ZTerm* __restrict__ terms, int J, int pn, float fs, float fst, float twopi_f,
                        uint16_t* __restrict__ wavefront, float2* __restrict__ pupil)
{
    pupil_body(terms, J, pn, fs, fst, twopi_f, wavefront, pupil);
}
__global__ void k_pupil_packed(const ZTermPack pack, int J, int pn, float fs, float fst, float twopi_f,
                               uint16_t* __restrict__ wavefront, float2* __restrict__ pupil)
{
    pupil_body(pack.t, J, pn, fs, fst, twopi_f, wavefront, pupil);
}

// Through-focus stack ("optionally batched over P defocus planes"): the planes of a stack differ in
// coefficient 4 only, so the sigma grid, r, theta and every Zernike term but the defocus one are evaluated ONCE per pixel;
// per plane remain the defocus term's last two products (its radial polynomial and cos(0 * theta) are shared), the
// fp16-rounded running sum W = h(W + Z_j) in the reference's order j = 0 .. J-1 (pupil.py:95-99 -- the chain is re-run per
// plane because every partial sum after j = 4 depends on the plane) and the phase.  Same operations, same roundings as
// pupil_body: bit-identical to `planes` single-plane launches.
static constexpr int STACK_PLANES = 64;              // planes per launch: their defocus factors ride in the kernel arguments
struct StackPlanes {
    float cN4[STACK_PLANES];                         // h(c4_p * fp32(N_20)), c4_p after the two-rounding rescale (pupil.py:91-92)
};

__global__ void k_pupil_stack(const ZTermPack pack, int J, const StackPlanes sp, int planes, int pn, float fs, float fst,
                              float twopi_f, uint16_t* __restrict__ wavefront, float2* __restrict__ pupil)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= pn) return;
    const float x = sigma_axis(col, pn, fs, fst);
    const float y = sigma_axis(row, pn, fs, fst);
    const float r = radius16(x, y);
    const bool inside = r <= 1.f;
    const float theta = h16((float)atan2((double)y, (double)x));
    float Z[PACK_TERMS];                             // Z_j of the shared terms; slot 4 is not used
    float R4 = 0.f, trig4 = 0.f;
    for (int j = 0; j < J; ++j) {
        const ZTerm t = pack.t[j];
        float acc = 0.f;
        for (int k = 0; k < t.nk; ++k) acc = acc + h16(t.coef[k] * h16(pow_int_rn(r, t.n - 2 * k)));
        const float R = h16(acc);
        const float arg = h16((float)t.m * theta);
        const float trig = h16((float)(t.m >= 0 ? cos((double)arg) : sin((double)arg)));
        if (j == 4) { R4 = R; trig4 = trig; }
        float z = h16(h16(t.cN * R) * trig);
        if (!inside) z = 0.f;
        Z[j] = z;
    }
    const size_t idx = (size_t)row * pn + col, plane = (size_t)pn * pn;
    for (int p = 0; p < planes; ++p) {
        float z4 = h16(h16(sp.cN4[p] * R4) * trig4);
        if (!inside) z4 = 0.f;
        float W = 0.f;
        for (int j = 0; j < J; ++j) W = h16(W + (j == 4 ? z4 : Z[j]));
        if (wavefront) wavefront[p * plane + idx] = __half_as_ushort(__float2half_rn(W));
        if (pupil) {
            float2 phi = make_float2(0.f, 0.f);
            if (inside) {
                const float ang = twopi_f * W;
                phi = make_float2((float)cos((double)ang), (float)sin((double)ang));
            }
            pupil[p * plane + idx] = phi;
        }
    }
}

// generatePhi (pupil.py:102-111) for an arbitrary complex64 WE = a + i b:
// exp(i 2 pi WE) = exp(-2 pi b) (cos 2 pi a + i sin 2 pi a), with 2 pi held in fp32.
__global__ void k_pupil_phase(const float2* __restrict__ we, int pn, float fs, float fst, float twopi_f,
                              float2* __restrict__ pupil)
{
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (col >= pn) return;
    const float r = radius16(sigma_axis(col, pn, fs, fst), sigma_axis(row, pn, fs, fst));
    const size_t idx = (size_t)row * pn + col;
    float2 phi = make_float2(0.f, 0.f);
    if (r <= 1.f) {
        const float2 w = we[idx];
        const float ang = twopi_f * w.x;
        const float mag = (w.y == 0.f) ? 1.f : (float)exp(-(double)(twopi_f * w.y));
        phi = make_float2(mag * (float)cos((double)ang), mag * (float)sin((double)ang));
    }
    pupil[idx] = phi;
}

// ---- post-process (imageformation.py:69-77)
__device__ __forceinline__ void lin_coord(int dst, float rs, int n_in, int& i0, int& i1, float& l0, float& l1)
{
    float src = fmaf(rs, (float)dst + 0.5f, -0.5f);      // torch-CPU fuses this into one rounding
    src = src < 0.f ? 0.f : src;
    i0 = (int)floorf(src);
    if (i0 > n_in - 1) i0 = n_in - 1;
    l1 = src - (float)i0;
    l1 = l1 < 0.f ? 0.f : (l1 > 1.f ? 1.f : l1);
    l0 = 1.f - l1;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
}

template <typename TIn>
__device__ __forceinline__ float bilinear_at(const TIn* __restrict__ img, int n_in, int oy, int ox, float rs, bool same,
                                             bool take_abs)
{
    auto at = [&](int yy, int xx) {
        const float v = (float)img[(size_t)yy * n_in + xx];
        return take_abs ? fabsf(v) : v;
    };
    if (same) return at(oy, ox);                         // equal sizes: torch copies
    int y0, y1, x0, x1;
    float ly0, ly1, lx0, lx1;
    lin_coord(oy, rs, n_in, y0, y1, ly0, ly1);
    lin_coord(ox, rs, n_in, x0, x1, lx0, lx1);
    return ly0 * (lx0 * at(y0, x0) + lx1 * at(y0, x1)) + ly1 * (lx0 * at(y1, x0) + lx1 * at(y1, x1));
}

// RESIST: the constant-threshold resist model fused into the same pass (README.md:21 of the reference lists "photoresist
// response modeling, simple or otherwise" as an open goal; there is no reference code to match): the resampled
// intensity is multiplied by `gain` (dose, or dose / S for a normalised image) in fp32 and resist[...] = 1 where the
// result reaches `threshold`, else 0.  `out` (the fp32 aerial image) may be NULL when only the contour mask is wanted.
template <bool RESIST>
__global__ void k_postprocess(const float* __restrict__ raw, int pn, int ns, int pW, int n_out, float rs,
                              float* __restrict__ out, float gain, float threshold, unsigned char* __restrict__ resist)
{
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    const int p = blockIdx.z;
    if (ox >= n_out) return;
    const int iy = oy - pW, ix = ox - pW;
    float v = 0.f;
    if (iy >= 0 && iy < ns && ix >= 0 && ix < ns)
        v = bilinear_at(raw + (size_t)p * pn * pn, pn, iy, ix, rs, ns == pn, true);
    const size_t o = ((size_t)p * n_out + oy) * n_out + ox;
    if (!RESIST || out) out[o] = v;
    if constexpr (RESIST) resist[o] = (v * gain >= threshold) ? 1 : 0;
}

// bilinear up-scaling of the int16 mask geometry (mask.py:76-77) into a dense fp32 image
__global__ void k_scale_mask(const int16_t* __restrict__ geo, int pn, int ns, float rs, float* __restrict__ out)
{
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y;
    if (ox >= ns) return;
    out[(size_t)oy * ns + ox] = bilinear_at(geo, pn, oy, ox, rs, ns == pn, false);
}

void launch_scale_mask(const int16_t* geo, int pn, int ns, double scale, float* out, hipStream_t st)
{
    const float rs = (float)(1.0 / scale);
    hipLaunchKernelGGL(k_scale_mask, dim3((ns + 255) / 256, ns), dim3(256), 0, st, geo, pn, ns, rs, out);
}

static int osa_n(int j) { return (int)std::ceil(0.5 * (-3.0 + std::sqrt(9.0 + 8.0 * j))); }

static double factorial(int v) { double f = 1.0; for (int i = 2; i <= v; ++i) f *= i; return f; }

// ----------------------------------------------------------------------------------
// Embedded evaluation: centre a pn x pn complex array in a zero-padded pe x pe one, and add the
// centre pn x pn of a pe x pe real image to a pn x pn one.  One thread per destination element; blockIdx.z = plane.
// ----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_c64(const float2* __restrict__ src, int pn, float2* __restrict__ dst, int pe)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pe) return;
    const int o = (pe - pn) / 2, sx = x - o, sy = y - o;
    const bool in = sx >= 0 && sx < pn && sy >= 0 && sy < pn;
    dst[((size_t)blockIdx.z * pe + y) * pe + x] = in ? src[((size_t)blockIdx.z * pn + sy) * pn + sx] : make_float2(0.f, 0.f);
}
__global__ __launch_bounds__(256) void k_crop_add_f32(const float* __restrict__ src, int pe, float* __restrict__ dst, int pn)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pn) return;
    const int o = (pe - pn) / 2;
    dst[((size_t)blockIdx.z * pn + y) * pn + x] += src[((size_t)blockIdx.z * pe + y + o) * pe + x + o];
}
void launch_embed_c64(const float2* src, int planes, int pn, float2* dst, int pe, hipStream_t st)
{
    hipLaunchKernelGGL(k_embed_c64, dim3((pe + 255) / 256, pe, planes), dim3(256), 0, st, src, pn, dst, pe);
}
void launch_crop_add_f32(const float* src, int planes, int pe, float* dst, int pn, hipStream_t st)
{
    hipLaunchKernelGGL(k_crop_add_f32, dim3((pn + 255) / 256, pn, planes), dim3(256), 0, st, src, pe, dst, pn);
}

}  // namespace litho

extern "C" {

int litho_epsilon_n(double deltaK, double pixelSize, double wavelength, double* epsilon_host, int* N_host)
{
    if (!epsilon_host || !N_host) return LITHO_E_ARG;
    const double beta = 1.0 / ((deltaK * pixelSize) / wavelength);       // mask.py:68
    int best = 2;
    float bestd = INFINITY;
    for (int p = 1; p <= 14; ++p) {                                     // mask.py:64-65, distances in fp32
        const float d = std::fabs((float)(1 << p) - (float)beta);
        if (d < bestd) { bestd = d; best = 1 << p; }
    }
    *N_host = best;
    *epsilon_host = best / beta;                                        // mask.py:70
    return LITHO_OK;
}

int litho_source_bitmap(int kind, double sigma_in, double sigma_out, int pn, double shift_x, double shift_y,
                        int count, double rotation, int64_t* bitmap, void* stream)
{
    using namespace litho;
    if (!bitmap || pn < 1 || (kind != 0 && kind != 1)) return LITHO_E_ARG;
    if (kind == 1 && (count < 1 || count > 64)) return LITHO_E_ARG;
    const double step = 4.0 / pn;
    for (double sh : {shift_x, shift_y}) {                               // arange element count must be pn
        const double start = -2.0 - sh, end = 2.0 - sh;
        if ((long long)std::ceil((end - start) / step) != pn) return LITHO_E_ARG;
    }
    SourceParams sp;
    memset(&sp, 0, sizeof(sp));
    sp.kind = kind; sp.pn = pn; sp.count = kind == 1 ? count : 0;
    sp.fsx = (float)(-2.0 - shift_x); sp.fsy = (float)(-2.0 - shift_y); sp.fst = (float)step;
    sp.sin16 = h16_host((float)sigma_in); sp.sout16 = h16_host((float)sigma_out);
    sp.rot16 = h16_host((float)rotation); sp.twopi16 = h16_host((float)(2.0 * M_PI));
    const double spacing = M_PI / (kind == 1 ? count : 1);               // lightsource.py:67
    for (int g = 0; g < sp.count; ++g) {
        sp.lo16[g] = h16_host((float)((g + g) * spacing));
        sp.hi16[g] = h16_host((float)((g + g + 1) * spacing));
    }
    hipLaunchKernelGGL(k_source_bitmap, dim3((pn + 255) / 256, pn), dim3(256), 0, (hipStream_t)stream, sp, bitmap);
    HIP_TRY(hipGetLastError());
    return LITHO_OK;
}

int litho_source_compact(const int64_t* bitmap, int pn, int32_t* shifts, int64_t capacity, int32_t* scratch,
                         int64_t* count_host, void* stream)
{
    using namespace litho;
    if (!bitmap || !shifts || !scratch || pn < 1 || capacity < 0) return LITHO_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_row_counts, dim3(pn), dim3(256), 0, st, bitmap, pn, scratch);
    if (pn <= ROW_SUM_MAX) {
        hipLaunchKernelGGL(k_row_write_sum, dim3(pn), dim3(64), 0, st, bitmap, pn, scratch, shifts, (long long)capacity);
    } else {
        hipLaunchKernelGGL(k_row_scan, dim3(1), dim3(1024), 0, st, scratch, pn);
        hipLaunchKernelGGL(k_row_write, dim3(pn), dim3(64), 0, st, bitmap, pn, scratch, shifts, (long long)capacity);
    }
    HIP_TRY(hipGetLastError());
    if (!count_host) return LITHO_OK;                // asynchronous form: S stays on the device in scratch[pn]
    int total = 0;
    HIP_TRY(hipMemcpyAsync(&total, scratch + pn, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *count_host = total;
    return total > capacity ? LITHO_E_ARG : LITHO_OK;
}

// c[j] (fp32 values of the fp16 coefficients, coefficient 4 already rescaled) -> the per-term constants of pupil.py:46-77
static int build_terms(const std::vector<float>& c, std::vector<litho::ZTerm>& terms)
{
    using namespace litho;
    const int J = (int)c.size();
    terms.resize(J);
    for (int j = 0; j < J; ++j) {
        ZTerm& t = terms[j];
        t.n = osa_n(j);                                                  // pupil.py:84-85
        t.m = 2 * j - t.n * (t.n + 2);
        const int am = t.m < 0 ? -t.m : t.m;
        const int lLim = (t.n - am) / 2, ilLim = (t.n + am) / 2;
        t.nk = lLim + 1;
        if (t.nk > 10) return LITHO_E_ARG;                               // n <= 19: 209 terms
        for (int k = 0; k <= lLim; ++k) {                                // pupil.py:63
            const double st_c = ((k & 1) ? -1.0 : 1.0) * factorial(t.n - k) /
                                (factorial(k) * factorial(ilLim - k) * factorial(lLim - k));
            t.coef[k] = (float)st_c;
        }
        const double Nmn = std::sqrt((2.0 * t.n + 1.0) / (1.0 + (t.m == 0 ? 1.0 : 0.0)));   // pupil.py:68
        t.cN = h16_host(c[j] * (float)(t.m >= 0 ? Nmn : -Nmn));          // pupil.py:71/73
    }
    return LITHO_OK;
}

// pupil.py:91-92: aberrations[4] = aberrations[4] * NA**2 / (4 * wavelength) on an fp16 tensor -- two fp16 roundings
static float rescale_defocus(float c4, double NA, double wavelength)
{
    const float t = litho::h16_host(c4 * (float)(NA * NA));
    return litho::h16_host(t / (float)(4.0 * wavelength));
}

int litho_pupil(uint16_t* coeffs_f16_host, int J, int pn, double NA, double wavelength, int flags,
                uint16_t* wavefront, void* pupil, void* stream)
{
    using namespace litho;
    if (!coeffs_f16_host || J < 1 || pn < 1) return LITHO_E_ARG;
    if (!wavefront && !pupil) return LITHO_E_ARG;
    const bool rescale = !(flags & 1);
    if (rescale && J == 4) return LITHO_E_INDEX;                         // pupil.py:91-92 indexes [4] (Q3)
    hipStream_t st = (hipStream_t)stream;
    std::vector<float> c(J);
    for (int j = 0; j < J; ++j) c[j] = __half2float(__ushort_as_half(coeffs_f16_host[j]));
    if (rescale && J >= 4) {                                             // defocus rescale, two fp16 roundings
        c[4] = rescale_defocus(c[4], NA, wavelength);
        coeffs_f16_host[4] = __half_as_ushort(__float2half_rn(c[4]));    // the reference mutates its argument (Q2)
    }
    std::vector<ZTerm> terms;
    const int rc = build_terms(c, terms);
    if (rc) return rc;
    if (J <= PACK_TERMS) {                                               // asynchronous: terms ride in the kernarg segment
        ZTermPack pack;
        memset(&pack, 0, sizeof(pack));
        for (int j = 0; j < J; ++j) pack.t[j] = terms[j];
        hipLaunchKernelGGL(k_pupil_packed, dim3((pn + 255) / 256, pn), dim3(256), 0, st, pack, J, pn, -2.0f,
                           (float)(4.0 / pn), (float)(2.0 * M_PI), wavefront, (float2*)pupil);
        HIP_TRY(hipGetLastError());
        return LITHO_OK;
    }
    ZTerm* dterms = nullptr;
    HIP_TRY(hipMallocAsync((void**)&dterms, sizeof(ZTerm) * J, st));
    HIP_TRY(hipMemcpyAsync(dterms, terms.data(), sizeof(ZTerm) * J, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pupil, dim3((pn + 255) / 256, pn), dim3(256), 0, st, dterms, J, pn, -2.0f,
                       (float)(4.0 / pn), (float)(2.0 * M_PI), wavefront, (float2*)pupil);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));       // `terms` is pageable host memory; keep it alive until copied
    HIP_TRY(hipFreeAsync(dterms, st));
    return LITHO_OK;
}

int litho_pupil_stack(const uint16_t* coeffs_f16_host, int J, const uint16_t* defocus_f16_host, int planes, int pn, double NA,
                      double wavelength, uint16_t* wavefront, void* pupil, void* stream)
{
    using namespace litho;
    if (!coeffs_f16_host || !defocus_f16_host || J < 1 || planes < 1 || pn < 1) return LITHO_E_ARG;
    if (!wavefront && !pupil) return LITHO_E_ARG;
    if (J < 5) return LITHO_E_INDEX;                                     // `ab[4] = d` on a shorter vector: IndexError
    hipStream_t st = (hipStream_t)stream;
    const size_t plane = (size_t)pn * pn;
    if (J > PACK_TERMS) {                                                // long vectors: plane by plane through the staged table
        std::vector<uint16_t> cj(coeffs_f16_host, coeffs_f16_host + J);
        for (int p = 0; p < planes; ++p) {
            cj[4] = defocus_f16_host[p];
            const int rc = litho_pupil(cj.data(), J, pn, NA, wavelength, 0, wavefront ? wavefront + p * plane : nullptr,
                                       pupil ? (void*)((float2*)pupil + p * plane) : nullptr, stream);
            if (rc) return rc;
        }
        return LITHO_OK;
    }
    std::vector<float> c(J);
    for (int j = 0; j < J; ++j) c[j] = __half2float(__ushort_as_half(coeffs_f16_host[j]));
    c[4] = 0.f;                                                          // replaced per plane
    std::vector<ZTerm> terms;
    const int rc = build_terms(c, terms);
    if (rc) return rc;
    ZTermPack pack;
    memset(&pack, 0, sizeof(pack));
    for (int j = 0; j < J; ++j) pack.t[j] = terms[j];
    const float N20 = (float)std::sqrt((2.0 * 2 + 1.0) / 2.0);           // term 4 = (m, n) = (0, 2): +N_mn, pupil.py:68/71
    for (int p0 = 0; p0 < planes; p0 += STACK_PLANES) {
        const int np = planes - p0 < STACK_PLANES ? planes - p0 : STACK_PLANES;
        StackPlanes sp;
        memset(&sp, 0, sizeof(sp));
        for (int p = 0; p < np; ++p)
            sp.cN4[p] = h16_host(rescale_defocus(__half2float(__ushort_as_half(defocus_f16_host[p0 + p])), NA, wavelength) * N20);
        hipLaunchKernelGGL(k_pupil_stack, dim3((pn + 255) / 256, pn), dim3(256), 0, st, pack, J, sp, np, pn, -2.0f,
                           (float)(4.0 / pn), (float)(2.0 * M_PI), wavefront ? wavefront + p0 * plane : nullptr,
                           pupil ? (float2*)pupil + p0 * plane : nullptr);
        HIP_TRY(hipGetLastError());
    }
    return LITHO_OK;
}

int litho_pupil_phase(const void* wavefront_c64, int pn, void* pupil, void* stream)
{
    using namespace litho;
    if (!wavefront_c64 || !pupil || pn < 1) return LITHO_E_ARG;
    hipLaunchKernelGGL(k_pupil_phase, dim3((pn + 255) / 256, pn), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)wavefront_c64, pn, -2.0f, (float)(4.0 / pn), (float)(2.0 * M_PI), (float2*)pupil);
    HIP_TRY(hipGetLastError());
    return LITHO_OK;
}

int litho_postprocess_size(int pn, double epsilon, int* n_out_host)
{
    if (!n_out_host || pn < 1 || !(epsilon > 0)) return LITHO_E_ARG;
    const int ns = (int)std::floor((double)pn * (1.0 / epsilon));       // F.interpolate output size
    const int pW = (int)std::floor((pn - std::nearbyint(pn / epsilon)) / 2.0);   // Python // on round() (banker's)
    const int n_out = ns + 2 * pW + (ns % 2);
    if (n_out < 1) return LITHO_E_ARG;
    *n_out_host = n_out;
    return LITHO_OK;
}

int litho_postprocess(const float* raw, int planes, int pn, double epsilon, float* out, void* stream)
{
    using namespace litho;
    int n_out = 0;
    int rc = litho_postprocess_size(pn, epsilon, &n_out);
    if (rc) return rc;
    if (!raw || !out || planes < 1) return LITHO_E_ARG;
    const double scale = 1.0 / epsilon;
    const int ns = (int)std::floor((double)pn * scale);
    const int pW = (int)std::floor((pn - std::nearbyint(pn / epsilon)) / 2.0);
    const float rs = (float)(1.0 / scale);
    hipLaunchKernelGGL(k_postprocess<false>, dim3((n_out + 255) / 256, n_out, planes), dim3(256), 0, (hipStream_t)stream,
                       raw, pn, ns, pW, n_out, rs, out, 1.0f, 0.0f, (unsigned char*)nullptr);
    HIP_TRY(hipGetLastError());
    return LITHO_OK;
}

int litho_postprocess_resist(const float* raw, int planes, int pn, double epsilon, double gain, double threshold,
                             float* out, uint8_t* resist, void* stream)
{
    using namespace litho;
    int n_out = 0;
    int rc = litho_postprocess_size(pn, epsilon, &n_out);
    if (rc) return rc;
    if (!raw || !resist || planes < 1 || !(gain == gain) || !(threshold == threshold)) return LITHO_E_ARG;
    const double scale = 1.0 / epsilon;
    const int ns = (int)std::floor((double)pn * scale);
    const int pW = (int)std::floor((pn - std::nearbyint(pn / epsilon)) / 2.0);
    const float rs = (float)(1.0 / scale);
    hipLaunchKernelGGL(k_postprocess<true>, dim3((n_out + 255) / 256, n_out, planes), dim3(256), 0, (hipStream_t)stream,
                       raw, pn, ns, pW, n_out, rs, out, (float)gain, (float)threshold, (unsigned char*)resist);
    HIP_TRY(hipGetLastError());
    return LITHO_OK;
}

}  // extern "C"
