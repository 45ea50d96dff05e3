// xpass_lab.hip -- standalone bench for the row pass on config 3's coarse-grid geometry (pn = N' = 2048, 1025-row pupil
// box, 12 consecutive source points): the product kernel (one row per 128-thread workgroup, 8-column tiles) against the
// four-interleaved-rows variant writing 2-column tiles.
// REJECTED experiment (round 3): bit-identical T, but 57.9 us per 12-item launch at best (one 12-item chunk per workgroup)
// against 52.1 for the product kernel -- 512-thread workgroups, one per CU; forced to 128 VGPRs for two per CU it spills
// (88-98 us).  The 2-column tiles it would feed make the y-pass 7 % faster (ypass_lab.hip), less than this costs.
// Needs scripts/ubench/xpass_quad_rows.patch applied to csrc/engine_kernels.hpp (git apply), then:
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -fno-slp-vectorize xpass_lab.hip -o xpass_lab.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lithographysimulator_amd/csrc/engine_kernels.hpp"

#ifndef L2N
#define L2N 11
#endif
using namespace litho;
namespace litho {
void note_kernel(int, const char*, int, int, int, int) {}
template <int LOG2N> hipError_t launch_ypass_wave(const float2*, float*, const float2*, const PassGeom&, int, int, int, int, hipStream_t) { return hipErrorNotSupported; }
template <int LOG2N> hipError_t launch_xpass_rect(const float2*, const float2*, const int*, float2*, const float2*, const PassGeom&, int, int, hipStream_t) { return hipErrorNotSupported; }
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

static void geom(PassGeom& g, int pn, int rows, int tcl)
{
    g.pn = pn; g.c = pn / 2; g.N = pn; g.nt = (pn + 3) / 4; g.tcl = tcl;
    g.kx0 = -pn / 4; g.kx1 = pn / 4 + 1; g.ky0 = -pn / 4; g.ky1 = g.ky0 + rows;
    g.rows = rows; g.general = 0; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0; g.xmask = 0; g.ymask = 0;
    const long long ntile = (pn + (1 << tcl) - 1) >> tcl;
    g.t_point = (ntile * rows) << tcl;
}

int main(int argc, char** argv)
{
    constexpr int N = 1 << L2N;
    const int pn = N;
    const int nb = argc > 1 ? atoi(argv[1]) : 12;
    const int chunk = argc > 2 ? atoi(argv[2]) : 4;
    PassGeom g8, g2;
    geom(g8, pn, pn / 2 + 1, 3);            // product: 1025 rows, 8-column tiles
    geom(g2, pn, pn / 2, 1);                // variant: rows -512 .. 511 (the row k = +512 holds one sample: handled apart), 2-column tiles
    float2 *M, *P, *T8, *T2, *tw;
    int* shifts;
    CK(hipMalloc(&M, (size_t)pn * pn * 8));
    CK(hipMalloc(&P, (size_t)pn * pn * 8));
    CK(hipMalloc(&T8, (size_t)nb * g8.t_point * 8));
    CK(hipMalloc(&T2, (size_t)nb * g2.t_point * 8));
    CK(hipMalloc(&tw, N * 8));
    CK(hipMalloc(&shifts, nb * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, (size_t)pn * pn, 777u);
    hipLaunchKernelGGL(k_pupil, dim3((pn + 255) / 256, pn), dim3(256), 0, 0, P, pn);
    hipLaunchKernelGGL(k_tw, dim3((N + 255) / 256), dim3(256), 0, 0, tw, N);
    std::vector<int> sh(2 * nb);
    for (int s = 0; s < nb; ++s) { sh[2 * s] = -200; sh[2 * s + 1] = 100 + s; }      // consecutive points of one source row
    CK(hipMemcpy(shifts, sh.data(), nb * 8, hipMemcpyHostToDevice));
    CK(hipMemset(T8, 0, (size_t)nb * g8.t_point * 8));
    CK(hipMemset(T2, 0, (size_t)nb * g2.t_point * 8));
    CK(hipDeviceSynchronize());
    printf("x-pass lab: N' = pn = %d, nb %d, chunk %d, T item %.1f MB\n", pn, nb, chunk, g8.t_point * 8 / 1e6);

    using SI = SizeImpl<L2N>;
    CK(SI::xpass_abbe(0, 1, P, M, shifts, T8, tw, g8, nb, chunk, 0));
    CK(SI::template xa_quad<0>(P, M, shifts, T2, tw, g2, nb, chunk, 0));
    CK(hipDeviceSynchronize());
    {   // same numbers in both layouts (item nb - 1, rows 0 .. 1023)
        std::vector<float2> a(g8.t_point), b(g2.t_point);
        CK(hipMemcpy(a.data(), T8 + (size_t)(nb - 1) * g8.t_point, g8.t_point * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(b.data(), T2 + (size_t)(nb - 1) * g2.t_point, g2.t_point * 8, hipMemcpyDeviceToHost));
        double mx = 0, md = 0;
        for (int r = 0; r < g2.rows; ++r)
            for (int q = 0; q < pn; ++q) {
                const float2 u = a[(((size_t)(q >> 3) * g8.rows + r) << 3) + (q & 7)];
                const float2 v = b[(((size_t)(q >> 1) * g2.rows + r) << 1) + (q & 1)];
                mx = fmax(mx, fmax(fabs(u.x), fabs(u.y)));
                md = fmax(md, fmax(fabs(u.x - v.x), fabs(u.y - v.y)));
            }
        printf("quad-row kernel vs product kernel (item %d, rows 0..%d): max|diff| = %.3e, max|T| = %.3e\n", nb - 1, g2.rows - 1, md, mx);
    }
    const double t8 = time_kernel([&] { CK(SI::xpass_abbe(0, 1, P, M, shifts, T8, tw, g8, nb, chunk, 0)); }, 5);
    printf("product kernel (1 row / 128-thread workgroup, 8-column tiles, 1025 rows): %8.2f us per launch (%.3f us per item)\n", t8, t8 / nb);
    for (int ch : {chunk, 2, 3, 6, 12}) {
        if (nb % ch) continue;
        const double t2 = time_kernel([&] { CK(SI::template xa_quad<0>(P, M, shifts, T2, tw, g2, nb, ch, 0)); }, 5);
        printf("quad-row kernel (4 rows / 512-thread workgroup, 2-column tiles, 1024 rows), chunk %2d: %8.2f us per launch (%.3f x product)\n", ch, t2, t2 / t8);
    }
    return 0;
}
