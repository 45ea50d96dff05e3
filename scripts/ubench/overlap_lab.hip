// overlap_lab.hip -- round-4 experiment: x-pass of batch b + 1 UNDER the y-pass of batch b on DISJOINT sets of CUs
// (hipExtStreamCreateWithCUMask: the first K mask bits = K CUs, K / 8 per XCD, scripts/ubench/cumask_probe.hip), T double
// buffered.  Rounds 2 and 3 tried the overlap on plain side streams and lost (the two kernels share CUs: the y-pass loses its
// second wave per SIMD to x-pass workgroups); with CU masks each kernel keeps its own occupancy on its own CUs.
// Product kernels, standalone: geometry L2N = 11 (config 3: pn = N' = 2048, 8-column tiles, k_ypass_rect<11,8,true,2>) or
// L2N = 12 (config 4: pn = N' = 4096, 16-column tiles, k_ypass_coop).  No dependency on results: only the steady-state time
// per batch is measured (the event chain x(b) -> y(b) -> x(b + 2) is the real one).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-signed-zeros -fno-slp-vectorize -DLITHO_DIAG_BUILD [-DL2N=12] scripts/ubench/overlap_lab.hip -o scripts/ubench/overlap_lab.bin
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lab_kernels.hpp"

#ifndef L2N
#define L2N 11
#endif
using namespace litho;
namespace litho {
void note_kernel(int, const char*, int, int, int, int) {}
int device_cus() { return 256; }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

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

static hipStream_t masked_stream(int first, int count)       // CUs [first, first + count) of the 256 mask bits
{
    std::vector<uint32_t> m(8, 0u);
    for (int i = first; i < first + count; ++i) m[i >> 5] |= 1u << (i & 31);
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, 8, m.data()));
    return s;
}

int main(int argc, char** argv)
{
    constexpr int N = 1 << L2N;
    const int pn = N;
    const int nb = argc > 1 ? atoi(argv[1]) : (L2N == 11 ? 12 : 12);
    const int chunk = L2N == 11 ? 4 : 12;
    const int G = L2N == 11 ? 2 : 1;
    PassGeom g;
    g.pn = pn; g.c = pn / 2; g.N = pn; g.nt = (pn + 3) / 4; g.tcl = L2N == 11 ? 3 : 4;
    g.kx0 = -pn / 4; g.kx1 = pn / 4 + 1; g.ky0 = -pn / 4; g.ky1 = pn / 4 + 1;
    g.rows = pn / 2 + 1; g.general = 0; g.rect_off = 0; g.gcombine = 1; g.row_pairs = 0; g.coop_dma = 0; g.xmask = 0; g.ymask = 0;
    const int tc = 1 << g.tcl;
    g.t_point = (long long)((pn + tc - 1) / tc) * g.rows * tc;
    float2 *M, *P, *T[2], *tw;
    float* slab;
    int* shifts;
    CK(hipMalloc(&M, (size_t)pn * pn * 8));
    CK(hipMalloc(&P, (size_t)pn * pn * 8));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&T[i], (size_t)nb * g.t_point * 8));
    CK(hipMalloc(&tw, N * 8));
    CK(hipMalloc(&slab, (size_t)G * g.nt * 4 * pn * 4));
    CK(hipMalloc(&shifts, nb * 8));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, M, (size_t)pn * pn, 777u);
    hipLaunchKernelGGL(k_pupil, dim3((pn + 255) / 256, pn), dim3(256), 0, 0, P, pn);
    hipLaunchKernelGGL(k_tw, dim3((N + 255) / 256), dim3(256), 0, 0, tw, N);
    std::vector<int> sh(2 * nb);
    for (int s = 0; s < nb; ++s) { sh[2 * s] = -200; sh[2 * s + 1] = 100 + s; }
    CK(hipMemcpy(shifts, sh.data(), nb * 8, hipMemcpyHostToDevice));
    CK(hipMemset(slab, 0, (size_t)G * g.nt * 4 * pn * 4));
    CK(hipDeviceSynchronize());
    printf("overlap lab: pn = N' = %d, %d-column tiles, nb %d, T %.0f MB per buffer\n", pn, tc, nb, nb * g.t_point * 8 / 1e6);
    using SI = SizeImpl<L2N>;
    auto xk = [&](float2* Tb, hipStream_t s) { CK(SI::xpass_abbe(0, 1, P, M, shifts, Tb, tw, g, nb, chunk, s)); };
    auto yk = [&](float2* Tb, hipStream_t s) { CK(launch_ypass_wave<L2N>(Tb, slab, tw, g, nb, 1, G, G, s)); };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int B = 40;
    auto timed = [&](auto&& body, hipStream_t on) {
        body(); CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, on)); body(); CK(hipEventRecord(e1, on)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        return best * 1e3 / B;
    };
    // serial, all CUs, one stream (the product's structure)
    const double t_serial = timed([&] { for (int b = 0; b < B; ++b) { xk(T[0], 0); yk(T[0], 0); } }, 0);
    const double t_x = timed([&] { for (int b = 0; b < B; ++b) xk(T[0], 0); }, 0);
    const double t_y = timed([&] { for (int b = 0; b < B; ++b) yk(T[0], 0); }, 0);
    printf("serial on one stream, all 256 CUs: %7.2f us per batch   (x alone %7.2f, y alone %7.2f)\n", t_serial, t_x, t_y);
    hipEvent_t ex[2], ey[2], ej;
    for (int i = 0; i < 2; ++i) { CK(hipEventCreateWithFlags(&ex[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ey[i], hipEventDisableTiming)); }
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    for (int kx : {256, 48, 64, 80, 96, 112, 128}) {
        // kx = 256: both streams see every CU (the rounds 2 / 3 experiment); otherwise x-pass on the first kx CUs, y-pass on the rest
        hipStream_t sx = kx == 256 ? masked_stream(0, 256) : masked_stream(0, kx);
        hipStream_t sy = kx == 256 ? masked_stream(0, 256) : masked_stream(kx, 256 - kx);
        auto pipeline = [&] {
            for (int b = 0; b < B; ++b) {
                if (b >= 2) CK(hipStreamWaitEvent(sx, ey[b & 1], 0));
                xk(T[b & 1], sx);
                CK(hipEventRecord(ex[b & 1], sx));
                CK(hipStreamWaitEvent(sy, ex[b & 1], 0));
                yk(T[b & 1], sy);
                CK(hipEventRecord(ey[b & 1], sy));
            }
            CK(hipEventRecord(ej, sx)); CK(hipStreamWaitEvent(sy, ej, 0));      // join on sy
        };
        const double t_ov = timed(pipeline, sy);
        const double tx1 = timed([&] { for (int b = 0; b < B; ++b) xk(T[0], sx); }, sx);
        const double ty1 = timed([&] { for (int b = 0; b < B; ++b) yk(T[0], sy); }, sy);
        printf("x-pass on %3d CUs, y-pass on %3d: overlapped %7.2f us per batch (%.3f x serial)   [x alone on its CUs %7.2f, y alone on its CUs %7.2f]\n",
               kx, kx == 256 ? 256 : 256 - kx, t_ov, t_ov / t_serial, tx1, ty1);
        CK(hipStreamDestroy(sx)); CK(hipStreamDestroy(sy));
    }
    return 0;
}
