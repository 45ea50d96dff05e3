// ypass_lab.hip -- standalone bench/diagnostic harness for the wave-level y-pass kernels (no Python, one TU).
// Geometry = BASELINE config 3 on the coarse-grid path: pn = N' = 2048, 1025 live rows, 8-column T tiles,
// 12-item batches, G = 2 groups.  Times a kernel variant over a cache-resident T, checks it against the
// product kernel, and (TIMELINE) samples s_memtime at the phase boundaries of a few waves.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -fno-slp-vectorize -DLITHO_DIAG_BUILD
//        [-DL2N=11] ypass_lab.hip -o ypass_lab.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lab_kernels.hpp"
#ifdef LAB_EXTRA_HEADER
#include LAB_EXTRA_HEADER
#endif

#ifndef L2N
#define L2N 11
#endif

using namespace litho;
namespace litho {
void note_kernel(int, const char*, int, int, int, int) {}      // the library's introspection hook (common.hip): unused here
}

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);       \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

__global__ void k_fill(float2* p, size_t n, unsigned seed)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned h2 = h * 3266489917u ^ (h >> 16);
        p[i] = make_float2((float)(h & 0xFFFF) / 65536.f - 0.5f, (float)(h2 & 0xFFFF) / 65536.f - 0.5f);
    }
}
__global__ void k_tw(float2* tab, int N)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s, c;
    sincospi(2.0 * (double)n / (double)N, &s, &c);
    tab[n] = make_float2((float)c, (float)s);
}

// ---- plain copy of k_ypass_rect<LOG2N, 8, true> with timing-diagnostic switches (wrong results by construction) ----
// MODE bit 0: no global loads; bit 1: no LDS transposes; bit 2: no pass B; bit 3: no pass A
template <int LOG2N, int TC, int MODE>
__global__ __launch_bounds__(256, 2) void k_ypass_rect_diag(const float2* __restrict__ Tbuf, float* __restrict__ slab,
                                                            const float2* __restrict__ twtab, PassGeom g, int nb, int G,
                                                            int gstride)
{
    using W = WaveSq<6>;
    constexpr int S = 64, N = 1 << LOG2N, NL = (S * S) / N, H = S / NL;
    constexpr int JL = H / 4, NACC = S;
    constexpr int QT = NL / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = smem + wv * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane, 1);
    const int qx0 = wave_first_column<TC, 4 * NL>(blockIdx.x) + NL * wv;
    const int tile = qx0 / TC, col = qx0 & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[NACC];
    static_for<0, NACC>([&](auto i) { acc[i] = 0.f; });
    constexpr int RB = 8 * TC;
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? RB * S * j : RB * S * j - RB * N); };
    for (int s = grp; s < nb; s += G) {
        const __amdgpu_buffer_rsrc_t rT = make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        float2 x[S];
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, NL / 2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (j <= JL || j >= H - JL) {
                    if constexpr (MODE & 1) {
                        const float f = __uint_as_float((unsigned)(vb + s) | 0x3f000000u);
                        x[(2 * q) * H + j] = make_float2(f, (float)j);
                        x[(2 * q + 1) * H + j] = make_float2((float)(j + 1), f);
                    } else {
                        const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rT, slot_off(j) + 16u * (q % QT), 0, 0);
                        x[(2 * q) * H + j] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                        x[(2 * q + 1) * H + j] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
                    }
                } else {
                    x[(2 * q) * H + j] = make_float2(0.f, 0.f);
                    x[(2 * q + 1) * H + j] = make_float2(0.f, 0.f);
                }
            });
        });
        constexpr int LH = 6 - (NL == 2 ? 1 : NL == 4 ? 2 : NL == 8 ? 3 : 4);
        if constexpr (!(MODE & 8)) static_for<0, NL>([&](auto q_) { dif_network<LH, decltype(q_)::value * H, S>(x); });
        auto slot_of = [](int c) constexpr { return (c / H) * H + brev_bits(c % H, LH); };
        static_for<0, S>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int m = c % H, a = m >> 3, b = m & 7, sl = (c / H) * H + brev_bits(m, LH);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
        if constexpr (!(MODE & 2)) {
            float* const wr = lds + lane * (S + 1);
            float* const rd = lds + lane;
            static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].x; });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float re[S];
            static_for<0, S>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * (S + 1)]; });
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].y; });
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            static_for<0, S>([&](auto r_) {
                constexpr int r = decltype(r_)::value;
                x[r] = make_float2(re[r], rd[r * (S + 1)]);
            });
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if constexpr (!(MODE & 4)) W::dft_dif(x);
        static_for<0, NACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const float2 v = x[W::brev(i)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }
    const int qx = qx0 + lane / H, m = lane & (H - 1);
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, NACC>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int ubase = i < S / 2 ? H * i : H * i - N;
        srow[ubase + m + g.c] += acc[i];
    });
}
template <int MODE>
static void run_diag(const char* name, const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int G, double t_ref);

#ifdef TIMELINE
// ---- instrumented copy of k_ypass_rect<LOG2N, 8, true>: s_memtime at the phase boundaries ------------------
__device__ unsigned long long g_tl[64 * 16];
__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
__device__ __forceinline__ void pin1(float& a, float& b) { asm volatile("" : "+v"(a), "+v"(b)); }
template <int NV>
__device__ __forceinline__ void pin(float2 (&x)[NV])
{
#pragma unroll
    for (int i = 0; i < NV; ++i) pin1(x[i].x, x[i].y);
}
template <int LOG2N, int TC>
__global__ __launch_bounds__(256, 2) void k_ypass_rect_tl(const float2* __restrict__ Tbuf, float* __restrict__ slab,
                                                          const float2* __restrict__ twtab, PassGeom g, int nb, int G,
                                                          int gstride)
{
    using W = WaveSq<6>;
    constexpr int S = 64, N = 1 << LOG2N, NL = (S * S) / N, H = S / NL;
    constexpr int JL = H / 4, NACC = S;
    constexpr int QT = NL / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = smem + wv * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane, 1);
    const int qx0 = wave_first_column<TC, 4 * NL>(blockIdx.x) + NL * wv;
    const int tile = qx0 / TC, col = qx0 & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[NACC];
    static_for<0, NACC>([&](auto i) { acc[i] = 0.f; });
    constexpr int RB = 8 * TC;
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? RB * S * j : RB * S * j - RB * N); };
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int it = 0;
    for (int s = grp; s < nb; s += G, ++it) {
        const bool rec = it == 2;
        if (rec) ts[0] = stamp();
        const __amdgpu_buffer_rsrc_t rT = make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        float2 x[S];
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, NL / 2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (j <= JL || j >= H - JL) {
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rT, slot_off(j) + 16u * (q % QT), 0, 0);
                    x[(2 * q) * H + j] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                    x[(2 * q + 1) * H + j] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
                } else {
                    x[(2 * q) * H + j] = make_float2(0.f, 0.f);
                    x[(2 * q + 1) * H + j] = make_float2(0.f, 0.f);
                }
            });
        });
        static_for<0, H>([&](auto j_) {            // all loads have landed (zero slots stay literal zeros)
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JL || j >= H - JL)
                static_for<0, NL>([&](auto q_) { constexpr int q = decltype(q_)::value; pin1(x[q * H + j].x, x[q * H + j].y); });
        });
        if (rec) ts[1] = stamp();
        // ---- run_rect<NL>, opened up
        constexpr int LH = 6 - (NL == 2 ? 1 : NL == 4 ? 2 : NL == 8 ? 3 : 4);
        static_for<0, NL>([&](auto q_) { dif_network<LH, decltype(q_)::value * H, S>(x); });
        auto slot_of = [](int c) constexpr { return (c / H) * H + brev_bits(c % H, LH); };
        pin(x);
        if (rec) ts[2] = stamp();
        static_for<0, S>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int m = c % H, a = m >> 3, b = m & 7, sl = (c / H) * H + brev_bits(m, LH);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
        pin(x);
        if (rec) ts[3] = stamp();
        float* const wr = lds + lane * (S + 1);
        float* const rd = lds + lane;
        static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].x; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float re[S];
        static_for<0, S>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * (S + 1)]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].y; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, S>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            x[r] = make_float2(re[r], rd[r * (S + 1)]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pin(x);
        if (rec) ts[4] = stamp();
        W::dft_dif(x);
        pin(x);
        if (rec) ts[5] = stamp();
        static_for<0, NACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const float2 v = x[W::brev(i)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
        _Pragma("unroll") for (int i = 0; i < NACC; ++i) pin1(acc[i], acc[i]);
        if (rec) ts[6] = stamp();
    }
    ts[7] = stamp();
    // record: blocks 0, 97, 511, 1000 -> wave slots
    const int rb = blockIdx.x == 0 ? 0 : blockIdx.x == 97 ? 1 : blockIdx.x == 333 ? 2 : blockIdx.x == 500 ? 3 : -1;
    if (rb >= 0 && blockIdx.y == 0 && lane == 0) {
        unsigned long long* o = g_tl + ((rb * 4 + wv) * 16);
        for (int i = 0; i < 8; ++i) o[i] = ts[i];
    }
    const int qx = qx0 + lane / H, m = lane & (H - 1);
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, NACC>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int ubase = i < S / 2 ? H * i : H * i - N;
        srow[ubase + m + g.c] += acc[i];
    });
}
#endif

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
    return best * 1e3;      // us per launch
}

template <int MODE>
static void run_diag(const char* name, const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int G, double t_ref)
{
    constexpr size_t lds4 = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);
    constexpr int NL = 4096 >> L2N;
    auto kern = k_ypass_rect_diag<L2N, 8, MODE>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    const dim3 grid(wave_grid_x<8, 4 * NL>(g.pn), G);
    const double t = time_kernel([&] { hipLaunchKernelGGL(kern, grid, dim3(256), lds4, 0, T, slab, tw, g, nb, G, G); }, 5);
    printf("diag %-28s: %8.2f us per launch (%.3f x product)\n", name, t, t / t_ref);
}

int main(int argc, char** argv)
{
    constexpr int N = 1 << L2N;
    const int pn = N;                                    // coarse-grid transform: N' = pn
    const int nb = argc > 1 ? atoi(argv[1]) : 12;
    const int G = argc > 2 ? atoi(argv[2]) : 2;
    PassGeom g;
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (pn + 3) / 4; g.tcl = 3;
    g.kx0 = -pn / 4; g.kx1 = pn / 4 + 1; g.ky0 = -pn / 4; g.ky1 = pn / 4 + 1;
    g.rows = pn / 2 + 1; g.general = 0; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0; g.xmask = 0; g.ymask = 0;
    g.t_point = (long long)((pn + 7) / 8) * g.rows * 8;
    float2 *T, *tw;
    float *slab, *slab_ref;
    const size_t tn = (size_t)nb * g.t_point;
    const size_t sn = (size_t)G * g.nt * 4 * pn;
    CK(hipMalloc(&T, tn * sizeof(float2)));
    CK(hipMalloc(&tw, N * sizeof(float2)));
    CK(hipMalloc(&slab, sn * sizeof(float)));
    CK(hipMalloc(&slab_ref, sn * sizeof(float)));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, T, tn, 12345u);
    hipLaunchKernelGGL(k_tw, dim3((N + 255) / 256), dim3(256), 0, 0, tw, N);
    CK(hipMemset(slab, 0, sn * sizeof(float)));
    CK(hipMemset(slab_ref, 0, sn * sizeof(float)));
    CK(hipDeviceSynchronize());
    printf("y-pass lab: N' = pn = %d, rows %d, nb %d, G %d, T %.1f MB\n", pn, g.rows, nb, G, tn * 8 / 1e6);

    // reference = product kernel
    CK(launch_ypass_wave<L2N>(T, slab_ref, tw, g, nb, 1, G, G, 0));
    CK(hipDeviceSynchronize());
    const double t_ref = time_kernel([&] { CK(launch_ypass_wave<L2N>(T, slab, tw, g, nb, 1, G, G, 0)); }, 5);
    printf("product kernel           : %8.2f us per launch  (%.3f us per item)\n", t_ref, t_ref / nb);

    {
        const double t0 = time_kernel([&] { CK(launch_ypass_wave<L2N>(T, slab, tw, g, 0, 1, G, G, 0)); }, 5);
        printf("product kernel, EMPTY batch (launch + twiddle loads + slab flush): %8.2f us per launch\n", t0);
        const double t1 = time_kernel([&] { hipLaunchKernelGGL(k_tw, dim3(1), dim3(64), 0, 0, tw, 0); }, 5);
        printf("empty kernel launch      : %8.2f us per launch\n", t1);
    }
    {   // both groups in one workgroup, accumulators combined through LDS before ONE slab flush
        PassGeom gc = g;
        gc.gcombine = 1;
        CK(hipMemset(slab, 0, sn * sizeof(float)));
        CK(launch_ypass_wave<L2N>(T, slab, tw, gc, nb, 1, G, G, 0));
        CK(hipDeviceSynchronize());
        std::vector<float> a(sn), b(sn);
        CK(hipMemcpy(a.data(), slab, sn * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), slab_ref, sn * sizeof(float), hipMemcpyDeviceToHost));
        const size_t one = (size_t)g.nt * 4 * pn;               // floats per slab
        double mx = 0, md = 0;
        for (size_t i = 0; i < one; ++i) {
            double ref = 0, got = 0;
            for (int gg = 0; gg < G; ++gg) { ref += b[gg * one + i]; got += a[gg * one + i]; }
            mx = fmax(mx, fabs(ref));
            md = fmax(md, fabs(ref - got));
        }
        printf("group-combining kernel vs product (sum over slabs): max|diff| / max = %.3e\n", md / mx);
        const double tc = time_kernel([&] { CK(launch_ypass_wave<L2N>(T, slab, tw, gc, nb, 1, G, G, 0)); }, 5);
        printf("group-combining kernel   : %8.2f us per launch (%.3f x product)\n", tc, tc / t_ref);
        const double tc0 = time_kernel([&] { CK(launch_ypass_wave<L2N>(T, slab, tw, gc, 0, 1, G, G, 0)); }, 5);
        printf("group-combining kernel, EMPTY batch: %8.2f us per launch\n", tc0);
    }
#ifdef TIMELINE
    {
        constexpr size_t lds4 = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);
        auto kern = k_ypass_rect_tl<L2N, 8>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        constexpr int NL = 4096 >> L2N;
        const dim3 grid(wave_grid_x<8, 4 * NL>(pn), G);
        const double t_tl = time_kernel([&] { hipLaunchKernelGGL(kern, grid, dim3(256), lds4, 0, T, slab, tw, g, nb, G, G); }, 3);
        printf("timeline-instrumented    : %8.2f us per launch\n", t_tl);
        unsigned long long h[64 * 16];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tl), sizeof(h)));
        const char* names[7] = {"loads issued+landed", "pass A (DIFs)", "lane twiddles", "LDS transposes", "pass B (64-pt DIF)", "|E|^2 accumulate", "rest of loop"};
        for (int w = 0; w < 16; ++w) {
            unsigned long long* o = h + w * 16;
            printf(" wave %2d:", w);
            for (int i = 0; i < 6; ++i) printf(" %-8s %6llu", i == 0 ? "ld" : i == 1 ? "A" : i == 2 ? "tw" : i == 3 ? "lds" : i == 4 ? "B" : "acc", o[i + 1] - o[i]);
            printf("  | iteration %llu  (s_memtime ticks = 100 MHz? see total) kernel-end-iter2start %llu\n", o[6] - o[0], o[7] - o[0]);
        }
        (void)names;
    }
#endif

#ifdef LAB_DIAG
    run_diag<0>("copy of the product kernel", T, slab, tw, g, nb, G, t_ref);
    run_diag<1>("no global loads", T, slab, tw, g, nb, G, t_ref);
    run_diag<2>("no LDS transposes", T, slab, tw, g, nb, G, t_ref);
    run_diag<3>("no loads, no LDS", T, slab, tw, g, nb, G, t_ref);
    run_diag<4>("no pass B", T, slab, tw, g, nb, G, t_ref);
    run_diag<12>("no pass A, no pass B", T, slab, tw, g, nb, G, t_ref);
    run_diag<14>("loads + twiddles + acc only", T, slab, tw, g, nb, G, t_ref);
#endif
#ifdef LAB_TC2
    {   // the same kernel over 2-column tiles (16-byte rows: a wave's load instruction covers 1 KB contiguous)
        PassGeom g2 = g;
        g2.tcl = 1;
        g2.t_point = (long long)(pn / 2) * g.rows * 2;
        constexpr size_t lds4 = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);
        constexpr int NL = 4096 >> L2N;
        auto kern = k_ypass_rect<L2N, 2, true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        const dim3 grid(wave_grid_x<2, 4 * NL>(pn), G);
        const double t2 = time_kernel([&] { hipLaunchKernelGGL(kern, grid, dim3(256), lds4, 0, T, slab, tw, g2, nb, G, G); }, 5);
        printf("same kernel, 2-column tiles: %8.2f us per launch (%.3f x product)\n", t2, t2 / t_ref);
    }
#endif
#ifdef LAB_SLAB4096
    {   // would column slabs help the 4096-point y-pass?  The first 1024 columns of a 12-item batch (16.8 MB per item: the
        // slab's T), read (a) cache-resident, launches back to back, (b) after a 1 GiB fill has evicted it (= from HBM)
        static_assert(L2N == 12, "LAB_SLAB4096 needs -DL2N=12");
        constexpr size_t ldsf = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float) + (size_t)WaveSq<6>::TW_LDS_FLOAT2 * sizeof(float2);
        auto kern = k_ypass_wave<12, 8, true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
        float* big; CK(hipMalloc(&big, (size_t)1 << 30));
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int cols : {1024, 4096}) {
            const dim3 grid(cols / 4, G);
            for (int evict = 0; evict < 2; ++evict) {
                double best = 1e30;
                for (int r = 0; r < 6; ++r) {
                    if (evict) CK(hipMemsetAsync(big, r, (size_t)1 << 30, 0));
                    CK(hipEventRecord(a));
                    hipLaunchKernelGGL(kern, grid, dim3(256), ldsf, 0, T, slab, tw, g, nb, G, G);
                    CK(hipEventRecord(b));
                    CK(hipEventSynchronize(b));
                    float ms; CK(hipEventElapsedTime(&ms, a, b));
                    if (r > 0 && ms * 1e3 < best) best = ms * 1e3;
                }
                printf("k_ypass_wave<12,8,true>, %4d columns x %d items, T %s: %8.2f us per launch = %.3f us per item per 1024 columns\n",
                       cols, nb, evict ? "evicted (HBM)" : "cache-resident if it fits", best, best / nb / (cols / 1024));
            }
        }
    }
#endif
#ifdef LAB_PF
    {   // k_ypass_rect with LDS-DMA prefetch of the next item
        using PF = RectPrefetch<L2N, true>;
        constexpr size_t ldsp = 4 * (size_t)PF::WAVE_BYTES;
        constexpr int NL = 4096 >> L2N;
        auto kern = k_ypass_rect_pf<L2N, 8, true>;
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp));
        const dim3 grid(wave_grid_x<8, 4 * NL>(pn), G);
        CK(hipMemset(slab, 0, sn * sizeof(float)));
        hipLaunchKernelGGL(kern, grid, dim3(256), ldsp, 0, T, slab, tw, g, nb, G, G);
        CK(hipDeviceSynchronize());
        std::vector<float> a(sn), b(sn);
        CK(hipMemcpy(a.data(), slab, sn * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), slab_ref, sn * sizeof(float), hipMemcpyDeviceToHost));
        double mx = 0, md = 0;
        for (size_t i = 0; i < sn; ++i) { mx = fmax(mx, fabs((double)b[i])); md = fmax(md, fabs((double)a[i] - (double)b[i])); }
        printf("prefetch kernel vs product: max|diff| / max = %.3e (max %.4e)\n", md / mx, mx);
        const double tp = time_kernel([&] { hipLaunchKernelGGL(kern, grid, dim3(256), ldsp, 0, T, slab, tw, g, nb, G, G); }, 5);
        printf("LDS-DMA prefetch kernel  : %8.2f us per launch (%.3f x product), LDS %zu B per workgroup\n", tp, tp / t_ref, ldsp);
    }
#endif
#ifdef LAB_LINE
    {   // one-line-per-wave kernel (line_kernels.hpp) over 2-column tiles, against the product kernel on the same tiles
        PassGeom g2 = g;
        g2.tcl = 1;
        g2.t_point = (long long)(pn / 2) * g.rows * 2;
        constexpr size_t lds4 = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);
        constexpr int NL = 4096 >> L2N;
        auto kref = k_ypass_rect<L2N, 2, true>;
        CK(hipFuncSetAttribute((const void*)kref, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        CK(hipMemset(slab_ref, 0, sn * sizeof(float)));
        hipLaunchKernelGGL(kref, dim3(wave_grid_x<2, 4 * NL>(pn), G), dim3(256), lds4, 0, T, slab_ref, tw, g2, nb, G, G);
        CK(hipDeviceSynchronize());
        std::vector<float> a(sn), b(sn);
        CK(hipMemcpy(b.data(), slab_ref, sn * sizeof(float), hipMemcpyDeviceToHost));
        auto check = [&](const char* name) {
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(a.data(), slab, sn * sizeof(float), hipMemcpyDeviceToHost));
            double mx = 0, md = 0; size_t wi = 0;
            for (size_t i = 0; i < sn; ++i) {
                mx = fmax(mx, fabs((double)b[i]));
                const double d = fabs((double)a[i] - (double)b[i]);
                if (d > md) { md = d; wi = i; }
            }
            printf("%s vs product kernel: max|diff| / max = %.3e (max %.4e, worst at %zu: %g vs %g)\n", name, md / mx, mx, wi, a[wi], b[wi]);
        };
        CK(hipMemset(slab, 0, sn * sizeof(float)));
        CK((launch_ypass_line<L2N, 3>(T, slab, tw, g2, nb, 1, G, G, 0)));
        check("line kernel, 3 waves/SIMD");
        CK(hipMemset(slab, 0, sn * sizeof(float)));
        CK((launch_ypass_line<L2N, 4>(T, slab, tw, g2, nb, 1, G, G, 0)));
        check("line kernel, 4 waves/SIMD");
        for (int GG : {2, 3, 4, 6}) {
            if (nb % GG) continue;
            const size_t sn2 = (size_t)GG * g.nt * 4 * pn;
            float* slab2; CK(hipMalloc(&slab2, sn2 * sizeof(float))); CK(hipMemset(slab2, 0, sn2 * sizeof(float)));
            const double t2 = time_kernel([&] { hipLaunchKernelGGL(kref, dim3(wave_grid_x<2, 4 * NL>(pn), GG), dim3(256), lds4, 0, T, slab2, tw, g2, nb, GG, GG); }, 5);
            const double t3 = time_kernel([&] { CK((launch_ypass_line<L2N, 3>(T, slab2, tw, g2, nb, 1, GG, GG, 0))); }, 5);
            const double t4 = time_kernel([&] { CK((launch_ypass_line<L2N, 4>(T, slab2, tw, g2, nb, 1, GG, GG, 0))); }, 5);
            printf("G = %d: product kernel on 2-col tiles %7.2f us | line kernel 3 waves/SIMD %7.2f us | 4 waves/SIMD %7.2f us\n", GG, t2, t3, t4);
            CK(hipFree(slab2));
        }
    }
#endif
#ifdef LAB_VARIANT
    {
        CK(hipMemset(slab, 0, sn * sizeof(float)));
        CK(lab_launch<L2N>(T, slab, tw, g, nb, 1, G, G, 0));
        CK(hipDeviceSynchronize());
        std::vector<float> a(sn), b(sn);
        CK(hipMemcpy(a.data(), slab, sn * sizeof(float), hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), slab_ref, sn * sizeof(float), hipMemcpyDeviceToHost));
        double mx = 0, md = 0;
        for (size_t i = 0; i < sn; ++i) {
            mx = fmax(mx, fabs((double)b[i]));
            md = fmax(md, fabs((double)a[i] - (double)b[i]));
        }
        printf("variant vs product: max|diff| / max = %.3e  (max %.4e)\n", md / mx, mx);
        const double t_var = time_kernel([&] { CK(lab_launch<L2N>(T, slab, tw, g, nb, 1, G, G, 0)); }, 5);
        printf("variant kernel           : %8.2f us per launch  (%.3f us per item)  = %.3f x product\n", t_var, t_var / nb, t_var / t_ref);
    }
#endif
    return 0;
}
