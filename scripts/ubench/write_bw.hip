// Write-bandwidth microbenchmark for gfx950: how fast can the chip absorb the x-pass's T stores?
//   A  contiguous 16-byte stores (1 KiB per wave instruction)                         -- the ceiling
//   B  32-byte granules (lane pair x 16 B), granules 32 KiB apart, each 128-byte line completed by four
//      DIFFERENT workgroups on the same XCD (blocks b, b+8, b+16, b+24)              -- T layout [tile][row][4 cols]
//   C  64-byte granules (4 lanes x 16 B), line completed by two workgroups             -- 8-column tiles
//   D  128-byte granules (8 lanes x 16 B): every line written whole by one wave        -- 16-column tiles
// Build: hipcc --offload-arch=gfx950 -O3 write_bw.hip -o write_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

// rows x cols complex64 "T": layout [cols/TC tiles][rows][TC cols]; one workgroup (256 threads) writes one row:
// thread t owns column pairs 2*(t + 256 m), m = 0..(cols/512 - 1)
template <int TC>
__global__ void k_rows(float* __restrict__ T, int rows, int cols, int reps)
{
    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
    const int a = (i >> 2) * 32 + xcd * 4 + (i & 3);           // the engine's XCD-aware row mapping
    if (a >= rows) return;
    const f4 v = {1.f, 2.f, 3.f, (float)a};
    for (int r = 0; r < reps; ++r) {
        float* base = T + (size_t)r * rows * cols * 2;
        for (int m = 0; m < cols / 512; ++m) {
            const int q = 2 * (threadIdx.x + 256 * m);           // even column
            const size_t off = (((size_t)(q / TC) * rows + a) * TC + (q % TC)) * 2;   // floats
            *reinterpret_cast<f4*>(base + off) = v;
        }
    }
}
// NR consecutive rows per workgroup, their stores issued back to back for every column slot: does the memory
// system merge the partial-line writes of consecutive instructions of ONE wave?  (4-column tiles)
template <int NR>
__global__ void k_rows_multi(float* __restrict__ T, int rows, int cols, int reps)
{
    const int b = blockIdx.x;
    const int a0 = b * NR;
    if (a0 >= rows) return;
    for (int r = 0; r < reps; ++r) {
        float* base = T + (size_t)r * rows * cols * 2;
        for (int m = 0; m < cols / 512; ++m) {
            const int q = 2 * (threadIdx.x + 256 * m);
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int a = a0 + j;
                if (a < rows) {
                    const f4 v = {1.f, 2.f, (float)j, (float)a};
                    const size_t off = (((size_t)(q / 4) * rows + a) * 4 + (q % 4)) * 2;
                    *reinterpret_cast<f4*>(base + off) = v;
                }
            }
        }
    }
}
// Whole 128-byte lines from ONE 16-byte store instruction whose two 64-byte halves come from lanes c and c + 32
// (rows a, a + 1 of an 8-column tile held by the two half-waves): does the coalescer merge across the half-waves?
// SPLIT = 0: lanes c..c+7 cover a line (reference).  One wave = 8 lines per instruction, T as [tile8][row][8].
template <int SPLIT>
__global__ void k_lines(float* __restrict__ T, int rows, int cols, int reps)
{
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int pairs = (rows + 1) / 2;
    if (wave >= pairs) return;
    const int a0 = 2 * wave;
    int r, m;                                        // row inside the pair, 16-byte column slot 0..31 of this store
    if (SPLIT) { r = lane >> 5; m = lane & 31; }
    else { r = (lane >> 2) & 1; m = (lane >> 3) * 4 + (lane & 3); }
    const f4 v = {1.f, 2.f, (float)r, (float)m};
    for (int rep = 0; rep < reps; ++rep) {
        float* base = T + (size_t)rep * rows * cols * 2;
        for (int k = 0; k < cols / 64; ++k) {        // 64 columns (32 slots of 2) per store instruction
            const int q = 64 * k + 2 * m;
            const size_t off = (((size_t)(q / 8) * rows + a0 + r) * 8 + (q % 8)) * 2;
            if (a0 + r < rows) *reinterpret_cast<f4*>(base + off) = v;
        }
    }
}
__global__ void k_contig(float* __restrict__ T, size_t n4)
{
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f4*>(T)[i] = v;
}
template <typename F> static double time_ms(F f)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 5; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 5;
}
int main()
{
    const int rows = 2049, cols = 4096, reps = 8;                 // config-4 geometry: 67 MB per item, 8 items
    const size_t bytes = (size_t)rows * cols * 8 * reps;
    float* T; hipMalloc(&T, bytes + (1 << 20));
    const int grid = (rows + 31) / 32 * 32;
    double ms = time_ms([&] { k_contig<<<2048, 256>>>(T, bytes / 16); });
    printf("A contiguous 16-B stores          : %.1f MB in %.3f ms = %.2f TB/s\n", bytes / 1e6, ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows<4><<<grid, 256>>>(T, rows, cols, reps); });
    printf("B 32-B granules (4-col tiles)     : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows<8><<<grid, 256>>>(T, rows, cols, reps); });
    printf("C 64-B granules (8-col tiles)     : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows<16><<<grid, 256>>>(T, rows, cols, reps); });
    printf("D 128-B granules (16-col tiles)   : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows<64><<<grid, 256>>>(T, rows, cols, reps); });
    printf("E 512-B granules (64-col tiles)   : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows_multi<2><<<(rows + 1) / 2, 256>>>(T, rows, cols, reps); });
    printf("B2 32-B granules, 2 rows per WG back to back : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_rows_multi<4><<<(rows + 3) / 4, 256>>>(T, rows, cols, reps); });
    printf("B4 32-B granules, 4 rows per WG back to back : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_lines<0><<<((rows + 1) / 2 + 3) / 4, 256>>>(T, rows, cols, reps); });
    printf("L0 full lines, 8 consecutive lanes per line (2 rows x 64 B)   : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    ms = time_ms([&] { k_lines<1><<<((rows + 1) / 2 + 3) / 4, 256>>>(T, rows, cols, reps); });
    printf("L1 full lines, halves from lanes c and c+32                   : %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    return 0;
}
