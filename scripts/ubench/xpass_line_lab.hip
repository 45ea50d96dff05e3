// xpass_line_lab.hip -- round-4 experiment: the coarse-grid x-pass at 2048^2 with ONE ROW PER WAVE on the wave-level
// engine (WaveLine2048: 64 lanes x 32 slots, wave-private LDS transpose, last radix-2 stage across the half-waves) -- no
// workgroup barrier at all -- against the product kernel k_xpass_abbe<11, 0, true, 1, 1> (radix-8 x 16 x 16 over a
// 128-thread workgroup, four LDS barriers per row).  Why: the product x-pass takes 47-50 us per 12-item batch with only
// ~20 us of VALU content (16.3 k instructions per SIMD) and a 30 us store floor; it waits on its barriers at three
// 164-register waves per SIMD.  Geometry = BASELINE config 3's batch (pn = N' = 2048, 1025-row box, 12 points, 8-column tiles).
// T is compared with the product kernel's (same arithmetic up to the order of the butterflies: fp32 rounding differs).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -fno-slp-vectorize -DLITHO_DIAG_BUILD scripts/ubench/xpass_line_lab.hip -o scripts/ubench/xpass_line_lab.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lab_kernels.hpp"

using namespace litho;
namespace litho {
void note_kernel(int, const char*, int, int, int, int) {}
int device_cus() { return 256; }
}

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

// One row of the pupil box per wave; a 256-thread workgroup takes rows 4 b .. 4 b + 3 (the two rows of every 128-byte line
// of an 8-column tile are written by one workgroup) for a chunk of source points.  P row (17 live slots) in registers, the
// mask-spectrum window of the next point prefetched while this one is transformed.
template <int WPS, bool TW_IN_LDS, bool PREFETCH = true>
__global__ __launch_bounds__(256, WPS) void k_xpass_line(const float2* __restrict__ P, const float2* __restrict__ M,
                                                         const int* __restrict__ shifts, float2* __restrict__ Tbuf,
                                                         const float2* __restrict__ twtab, PassGeom g, int nb, int chunk)
{
    using W = WaveLine2048;
    constexpr int H = W::H, N = W::N, JL = H / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = smem + wv * W::LDS_FLOATS;
    float2* twl = reinterpret_cast<float2*>(smem + 4 * W::LDS_FLOATS);
    typename W::LaneTwiddles tw;
    if constexpr (TW_IN_LDS) {
        W::fill_lane_twiddle_table(twl, twtab, threadIdx.x, 256);
        __syncthreads();
    } else {
        W::load_lane_twiddles(tw, twtab, lane);
    }
    const int a = blockIdx.x * 4 + wv;
    if (a >= g.rows) return;                                     // (no workgroup barrier below)
    const int s_begin = blockIdx.y * chunk, s_end = min(nb, s_begin + chunk);
    const int r = g.ky0 + g.c + a;
    const unsigned win_bytes = (unsigned)(g.kx1 - g.kx0) * 8u;
    const unsigned vb = (unsigned)(lane - g.kx0) * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? 512 * j : 512 * j - 8 * N); };
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(P + (size_t)r * g.pn + g.c + g.kx0, win_bytes);
    float2 pv[2 * JL + 1];
    static_for<0, H>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        if constexpr (j <= JL) pv[j] = buf_load_c64(rP, slot_off(j));
        else if constexpr (j >= H - JL) pv[j - (H - JL) + JL + 1] = buf_load_c64(rP, slot_off(j));
    });
    auto load_window = [&](int s, float2 (&mv)[2 * JL + 1]) {
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        const __amdgpu_buffer_rsrc_t rM = make_rsrc(M + (size_t)(r + dy) * g.pn + dx + g.c + g.kx0, win_bytes);
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JL) mv[j] = buf_load_c64(rM, slot_off(j));
            else if constexpr (j >= H - JL) mv[j - (H - JL) + JL + 1] = buf_load_c64(rM, slot_off(j));
        });
    };
    // output: slot s of lane (p, m) = X[m + 32 k], k = (out_k(s, 0) + p) mod 64; column q = (n + N/2) mod N = m + 32 (k ^ 32)
    const int p = lane >> 5, m = lane & 31;
    const unsigned rows64 = (unsigned)g.rows * 64u;             // bytes between consecutive 8-column tiles
    const unsigned obase = (unsigned)(m >> 3) * rows64 + (unsigned)a * 64u + (unsigned)(m & 7) * 8u;
    const unsigned tstep = 4u * rows64;                         // 32 columns further

    float2 mnext[2 * JL + 1];
    if constexpr (PREFETCH) {
        if (s_begin < s_end) load_window(s_begin, mnext);
        static_for<0, 2 * JL + 1>([&](auto e_) { touch_vgpr(mnext[decltype(e_)::value]); });
    }
    for (int s = s_begin; s < s_end; ++s) {
        float2 x[H];
        if constexpr (!PREFETCH) load_window(s, mnext);
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JL) x[j] = cmul(pv[j], mnext[j]);
            else if constexpr (j >= H - JL) x[j] = cmul(pv[j - (H - JL) + JL + 1], mnext[j - (H - JL) + JL + 1]);
            else x[j] = make_float2(0.f, 0.f);
        });
        if constexpr (PREFETCH) load_window(s + 1 < s_end ? s + 1 : s, mnext);
        W::template run<TW_IN_LDS>(x, tw, twl, lds, lane);
        const __amdgpu_buffer_rsrc_t rT = make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
        static_for<0, H>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr unsigned K0 = (unsigned)((W::out_k(i, 0) & 63) ^ 32), K1 = (unsigned)(((W::out_k(i, 0) + 1) & 63) ^ 32);
            const unsigned K = p ? K1 : K0;
#ifdef LAB_NOSTORE
            diag_keep(x[i], obase + K * tstep);
#else
            buf_store_c64<16>(rT, obase + K * tstep, x[i]);
#endif
        });
    }
}

__global__ void k_fill(float2* p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned h2 = h * 3266489917u ^ (h >> 16);
        p[i] = make_float2((float)(h & 0xFFFF) / 65536.f - 0.5f, (float)(h2 & 0xFFFF) / 65536.f - 0.5f);
    }
}
__global__ void k_pupil(float2* P, int pn)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= pn) return;
    const float fx = (x - pn / 2) / (float)(pn / 4), fy = (y - pn / 2) / (float)(pn / 4);
    const float r2 = fx * fx + fy * fy;
    float s, c;
    sincosf(3.0f * r2 + 0.5f * fx, &s, &c);
    P[(size_t)y * pn + x] = r2 <= 1.0f ? make_float2(c, s) : make_float2(0.f, 0.f);
}
__global__ void k_tw(float2* tab, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s, c;
    sincospi(2.0 * (double)n / (double)N, &s, &c);
    tab[n] = make_float2((float)c, (float)s);
}

template <typename Launcher>
static double time_kernel(Launcher&& launch, int reps, int inner = 10)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(a));
        for (int i = 0; i < inner; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        if (ms / inner < best) best = ms / inner;
    }
    return best * 1e3;
}

template <int WPS, bool TWL, bool PF = true>
static void launch_line(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw, const PassGeom& g, int nb, int chunk)
{
    using W = WaveLine2048;
    constexpr size_t lds = 4 * W::LDS_FLOATS * sizeof(float) + (TWL ? (size_t)W::TW_LDS_FLOAT2 * sizeof(float2) : 0);
    auto kern = k_xpass_line<WPS, TWL, PF>;
    static bool once = false;
    if (!once) { CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
    hipLaunchKernelGGL(kern, dim3((g.rows + 3) / 4, (nb + chunk - 1) / chunk), dim3(256), lds, 0, P, M, shifts, T, tw, g, nb, chunk);
}

int main(int argc, char** argv)
{
    constexpr int N = 2048;
    const int pn = N;
    const int nb = argc > 1 ? atoi(argv[1]) : 12;
    PassGeom g;
    g.pn = pn; g.c = pn / 2; g.N = pn; g.nt = (pn + 3) / 4; g.tcl = 3;
    g.kx0 = -pn / 4; g.kx1 = pn / 4 + 1; g.ky0 = -pn / 4; g.ky1 = pn / 4 + 1;
    g.rows = pn / 2 + 1; g.general = 0; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0; g.xmask = 0; g.ymask = 0;
    g.t_point = (long long)((pn + 7) / 8) * g.rows * 8;
    float2 *M, *P, *Ta, *Tb, *tw;
    int* shifts;
    CK(hipMalloc(&M, (size_t)pn * pn * 8));
    CK(hipMalloc(&P, (size_t)pn * pn * 8));
    CK(hipMalloc(&Ta, (size_t)nb * g.t_point * 8));
    CK(hipMalloc(&Tb, (size_t)nb * g.t_point * 8));
    CK(hipMalloc(&tw, N * 8));
    CK(hipMalloc(&shifts, nb * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, (size_t)pn * pn, 777u);
    hipLaunchKernelGGL(k_pupil, dim3((pn + 255) / 256, pn), dim3(256), 0, 0, P, pn);
    hipLaunchKernelGGL(k_tw, dim3((N + 255) / 256), dim3(256), 0, 0, tw, N);
    std::vector<int> sh(2 * nb);
    for (int s = 0; s < nb; ++s) { sh[2 * s] = -200; sh[2 * s + 1] = 100 + s; }
    CK(hipMemcpy(shifts, sh.data(), nb * 8, hipMemcpyHostToDevice));
    CK(hipMemset(Ta, 0, (size_t)nb * g.t_point * 8));
    CK(hipMemset(Tb, 0xFF, (size_t)nb * g.t_point * 8));
    CK(hipDeviceSynchronize());
    printf("x-pass line lab: N' = pn = %d, rows %d, nb %d, T item %.1f MB\n", pn, g.rows, nb, g.t_point * 8 / 1e6);
    using SI = SizeImpl<11>;
    CK(SI::xpass_abbe(0, 1, P, M, shifts, Ta, tw, g, nb, 4, 0));
    launch_line<2, false>(P, M, shifts, Tb, tw, g, nb, 4);
    CK(hipDeviceSynchronize());
    {
        std::vector<float2> a((size_t)nb * g.t_point), b((size_t)nb * g.t_point);
        CK(hipMemcpy(a.data(), Ta, a.size() * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), Tb, b.size() * 8, hipMemcpyDeviceToHost));
        double mx = 0, md = 0; size_t bad = 0;
        for (size_t i = 0; i < a.size(); ++i) {
            mx = fmax(mx, fmax(fabs(a[i].x), fabs(a[i].y)));
            const double d = fmax(fabs(a[i].x - b[i].x), fabs(a[i].y - b[i].y));
            if (!(d == d)) ++bad;
            else md = fmax(md, d);
        }
        printf("line kernel vs product kernel, all %d items: max|diff| / max|T| = %.3e (max|T| %.3e, NaN entries %zu)\n", nb, md / mx, mx, bad);
    }
    const double tp = time_kernel([&] { CK(SI::xpass_abbe(0, 1, P, M, shifts, Ta, tw, g, nb, 4, 0)); }, 5);
    printf("product kernel k_xpass_abbe<11,0,true,1,1>, chunk 4     : %8.2f us per launch\n", tp);
    for (int ch : {2, 3, 4, 6, 12}) {
        if (nb % ch) continue;
        const double t2 = time_kernel([&] { launch_line<2, false>(P, M, shifts, Tb, tw, g, nb, ch); }, 5);
        const double t3 = time_kernel([&] { launch_line<3, true, false>(P, M, shifts, Tb, tw, g, nb, ch); }, 5);
        const double t4 = time_kernel([&] { launch_line<2, false, false>(P, M, shifts, Tb, tw, g, nb, ch); }, 5);
        printf("line kernel, chunk %2d: 2 waves/SIMD + prefetch %8.2f us (%.3f x product) | 2 waves/SIMD no prefetch %8.2f | 3 waves/SIMD, lane twiddles in LDS, no prefetch %8.2f us (%.3f x)\n", ch, t2, t2 / tp, t4, t3, t3 / tp);
    }
    return 0;
}
