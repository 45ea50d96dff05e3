// Write bandwidth against the footprint that is rewritten over and over: does the 256 MiB Infinity Cache absorb
// the x-pass's T stores?   hipcc --offload-arch=gfx950 -O3 scripts/ubench/fill_bw.hip -o scripts/ubench/fill_bw.bin
// A: contiguous 16-byte stores.  C: the T pattern of 8-column tiles -- a wave instruction writes 8 separate 64-byte
// row granules, `tile_stride` bytes apart (one per tile), a workgroup of 128 threads writes one 2048-column row.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k_contig(float* T, size_t n4)
{
    f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        reinterpret_cast<f4*>(T)[i] = v;
}
// item layout [tile (cols/8)][row][8] float2; grid = (rows, items); 128 threads x 16 columns each (stride 128)
__global__ void k_tiles(float* T, int rows, int cols, int trows = 0)
{
    if (trows == 0) trows = rows;                                // row stride of a tile (padding experiment)
    f2* item = reinterpret_cast<f2*>(T) + (size_t)blockIdx.y * trows * cols;
    const int row = blockIdx.x;
    f2 v = {1.f, 2.f};
    for (int m = 0; m < cols / 128; ++m) {
        const int q = threadIdx.x + 128 * m;
        item[((size_t)(q >> 3) * trows + row) * 8 + (q & 7)] = v;
    }
}
template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}
int main()
{
    float* T; hipMalloc(&T, (size_t)1100 << 20);
    const int rows = 1025, cols = 2048;                          // config-3 coarse-grid geometry: 16.8 MB per item
    const size_t item = (size_t)rows * cols * 8;
    for (int items : {1, 2, 4, 6, 8, 10, 12, 14, 16, 24, 32, 48, 64}) {
        const size_t bytes = item * items;
        double a = time_ms([&] { k_contig<<<2048, 256>>>(T, bytes / 16); }, 20);
        double c = time_ms([&] { k_tiles<<<dim3(rows, items), 128>>>(T, rows, cols); }, 20);
        printf("%3d items = %7.1f MB rewritten 20x:  contiguous %.2f TB/s   64-B granules (T tiles) %.2f TB/s\n",
               items, bytes / 1e6, bytes / a / 1e9, bytes / c / 1e9);
    }
    // tile row stride padding: does the 65,600-byte tile stride of 1025 rows pile the 8 granules of a store instruction
    // (and the rows of neighbouring workgroups) onto a few memory channels?
    for (int items : {12, 32}) {
        for (int trows : {1025, 1026, 1028, 1032, 1040, 1056, 1088, 1152}) {
            double c = time_ms([&] { k_tiles<<<dim3(rows, items), 128>>>(T, rows, cols, trows); }, 20);
            printf("%2d items, tile stride %4d rows (%6d B, mod 4096 = %4d): %.2f TB/s\n", items, trows, trows * 64,
                   (trows * 64) % 4096, item * items / c / 1e9);
        }
    }
    return 0;
}
