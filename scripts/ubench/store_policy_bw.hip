// store_policy_bw.hip -- what the memory system takes from the x-pass's store pattern, by T tile width and by the cache
// policy of the store (plain = write-allocate in the L2, sc1 = write-through):  one workgroup per (row, item) writes the
// row's `cols` complex64 samples, 8 bytes per lane, 64 consecutive columns per store instruction, into T laid out
// [tile][row][TC columns].  Geometry A: 1025 rows x 2048 columns x 12 items (config 3: 201 MB, inside the Infinity
// Cache); geometry B: 2049 x 4096 x 8 items (config 4: 537 MB, through HBM).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/store_policy_bw.hip -o scripts/ubench/store_policy_bw.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

template <int TC, int AUX>
__global__ __launch_bounds__(256) void k_rows(float2* T, int rows, int cols, int items)
{
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    const int a = (i >> 2) * 32 + xcd * 4 + (i & 3);           // the engine's XCD-aware row mapping
    if (a >= rows) return;
    const size_t item_elems = (size_t)rows * cols;
    for (int s = blockIdx.y; s < items; s += gridDim.y) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(T + (size_t)s * item_elems, 0, (int)(unsigned)(item_elems * 8), 0x00020000);
        for (int q = threadIdx.x; q < cols; q += 256) {
            const unsigned off = (unsigned)((((size_t)(q / TC) * rows + a) * TC + (q % TC)) * 8);
            u2 v; v.x = (unsigned)q; v.y = (unsigned)a;
            __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, AUX);
        }
    }
}
template <typename F> static double time_ms(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms / 5 < best) best = ms / 5;
    }
    return best;
}
template <int TC, int AUX> static void run(float2* T, int rows, int cols, int items)
{
    const dim3 grid((rows + 31) / 32 * 32, 3);
    const double bytes = (double)rows * cols * 8 * items;
    const double ms = time_ms([&] { hipLaunchKernelGGL((k_rows<TC, AUX>), grid, dim3(256), 0, 0, T, rows, cols, items); });
    printf("  %2d-column tiles (%3d-byte runs), %s: %7.1f us = %.2f TB/s\n", TC, TC * 8, AUX == 16 ? "sc1  " : AUX == 0 ? "plain" : "other", ms * 1e3, bytes / ms / 1e9);
}
template <int AUX> static void sweep(float2* T, int rows, int cols, int items)
{
    run<4, AUX>(T, rows, cols, items); run<8, AUX>(T, rows, cols, items); run<16, AUX>(T, rows, cols, items); run<32, AUX>(T, rows, cols, items); run<64, AUX>(T, rows, cols, items); run<256, AUX>(T, rows, cols, items);
}
int main()
{
    float2* T; hipMalloc(&T, (size_t)2049 * 4096 * 8 * 8 + (1 << 20));
    printf("A: 1025 x 2048 x 12 items (201 MB, cache-resident)\n");
    sweep<0>(T, 1025, 2048, 12); sweep<16>(T, 1025, 2048, 12);
    printf("B: 2049 x 4096 x 8 items (537 MB, through HBM)\n");
    sweep<0>(T, 2049, 4096, 8); sweep<16>(T, 2049, 4096, 8);
    return 0;
}
