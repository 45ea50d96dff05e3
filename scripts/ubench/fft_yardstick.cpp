// fft_yardstick.cpp -- the vendor library (rocFFT) on the shapes of the engine's two passes: an EXTERNAL yardstick for the
// engine's per-line and per-image times (round-5 review, item 2).  NOT part of the product: nothing under
// lithographysimulator_amd/ links rocFFT, bench.py never runs this; it is a lab harness like the rest of scripts/ubench.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/ubench/fft_yardstick.cpp -L/opt/rocm/lib -lrocfft \
//         -o scripts/ubench/fft_yardstick.bin
//   gpurun -- ./scripts/ubench/fft_yardstick.bin > profiles/r06_fft_yardstick.txt
//
// What is timed (HIP events on the launch stream, warm-up first, median of 7 rounds of R back-to-back executions):
//   lines   batched 1-D n-point complex64 transforms over contiguous lines (the y-pass's job WITHOUT its pruned loads and
//           without the |E|^2 accumulation: rocFFT reads and writes every sample, the engine's y-pass reads half of them and
//           writes nothing), in place, batches sized from "inside the 256 MiB Infinity Cache" to "streams through HBM";
//   columns the same transforms down the COLUMNS of a row-major [n][lines] array (stride = lines, distance = 1): what the
//           y-pass does logically -- the library chooses its own transposing kernels;
//   2-D     batched n x n complex64 transforms in place (the whole per-source-point job of x-pass + y-pass, dense: no pupil-box
//           row pruning, no gather-multiply in front, no accumulation behind).
// Nominal flops = 5 n log2 n per line (the figure bench.py uses for the engine), so TFLOP/s are comparable.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define FFTCHECK(x) do { rocfft_status s_ = (x); if (s_ != rocfft_status_success) { printf("rocFFT status %d at %s:%d\n", (int)s_, __FILE__, __LINE__); return -1.0; } } while (0)

static float2* g_buf = nullptr;      // data (in place)
static void* g_work = nullptr;       // rocFFT work buffer
static size_t g_work_bytes = 0;
static hipStream_t g_stream;

__global__ void k_fill(float2* p, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 13);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = make_float2((float)(h & 0xffff) / 65536.f - 0.5f, (float)(h >> 16) / 65536.f - 0.5f);
    }
}

// one plan, timed: returns microseconds per execution (median of 7 rounds), or < 0 when the plan cannot be made
static double time_plan(size_t dims, const size_t* lengths, size_t batch, const size_t* strides, size_t dist, size_t elems)
{
    rocfft_plan_description desc = nullptr;
    if (strides) {
        FFTCHECK(rocfft_plan_description_create(&desc));
        FFTCHECK(rocfft_plan_description_set_data_layout(desc, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved,
                                                         nullptr, nullptr, dims, strides, dist, dims, strides, dist));
    }
    rocfft_plan plan = nullptr;
    FFTCHECK(rocfft_plan_create(&plan, rocfft_placement_inplace, rocfft_transform_type_complex_inverse, rocfft_precision_single,
                                dims, lengths, batch, desc));
    size_t wb = 0;
    FFTCHECK(rocfft_plan_get_work_buffer_size(plan, &wb));
    if (wb > g_work_bytes) {
        if (g_work) HIPCHECK(hipFree(g_work));
        HIPCHECK(hipMalloc(&g_work, wb));
        g_work_bytes = wb;
    }
    rocfft_execution_info info = nullptr;
    FFTCHECK(rocfft_execution_info_create(&info));
    FFTCHECK(rocfft_execution_info_set_stream(info, g_stream));
    if (wb) FFTCHECK(rocfft_execution_info_set_work_buffer(info, g_work, wb));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, g_stream, g_buf, elems);
    void* io[1] = {g_buf};
    for (int i = 0; i < 3; ++i) FFTCHECK(rocfft_execute(plan, io, nullptr, info));
    HIPCHECK(hipStreamSynchronize(g_stream));
    hipEvent_t e0, e1;
    HIPCHECK(hipEventCreate(&e0)); HIPCHECK(hipEventCreate(&e1));
    // enough repetitions for ~2 ms per round
    HIPCHECK(hipEventRecord(e0, g_stream));
    FFTCHECK(rocfft_execute(plan, io, nullptr, info));
    HIPCHECK(hipEventRecord(e1, g_stream));
    HIPCHECK(hipEventSynchronize(e1));
    float ms1 = 0; HIPCHECK(hipEventElapsedTime(&ms1, e0, e1));
    const int R = std::max(3, std::min(200, (int)(2.0f / std::max(ms1, 1e-3f))));
    std::vector<double> us;
    for (int round = 0; round < 7; ++round) {
        // re-randomise now and then: R in-place unnormalised transforms grow the data by n^R (inf / nan are as fast as numbers
        // on this hardware, but keep the inputs honest)
        hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, g_stream, g_buf, elems);
        HIPCHECK(hipEventRecord(e0, g_stream));
        for (int r = 0; r < R; ++r) FFTCHECK(rocfft_execute(plan, io, nullptr, info));
        HIPCHECK(hipEventRecord(e1, g_stream));
        HIPCHECK(hipEventSynchronize(e1));
        float ms = 0; HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
        us.push_back(ms * 1e3 / R);
    }
    std::sort(us.begin(), us.end());
    rocfft_execution_info_destroy(info);
    rocfft_plan_destroy(plan);
    if (desc) rocfft_plan_description_destroy(desc);
    HIPCHECK(hipEventDestroy(e0)); HIPCHECK(hipEventDestroy(e1));
    return us[us.size() / 2];
}

int main()
{
    HIPCHECK(hipSetDevice(0));
    HIPCHECK(hipStreamCreate(&g_stream));
    hipDeviceProp_t prop;
    HIPCHECK(hipGetDeviceProperties(&prop, 0));
    if (rocfft_setup() != rocfft_status_success) { printf("rocfft_setup failed\n"); return 2; }
    char ver[64] = {0};
    rocfft_get_version_string(ver, sizeof(ver));
    const size_t max_elems = (size_t)1 << 28;            // 2 GiB of complex64
    HIPCHECK(hipMalloc((void**)&g_buf, max_elems * sizeof(float2)));
    printf("# fft_yardstick: rocFFT %s on %s (%d CUs); complex64, in place, inverse, unnormalised; HIP events, median of 7 rounds\n",
           ver, prop.name, prop.multiProcessorCount);
    printf("# nominal flops = 5 n log2 n per line (2-D: 2 n lines); MB = bytes of the array (read once + written once per pass at least)\n");

    const int sizes[] = {256, 512, 1024, 2048, 4096};
    printf("\n## 1-D, contiguous lines (y-pass yardstick: lines of one launch of the engine are [items] x n)\n");
    printf("%6s %8s %9s %10s %12s %10s\n", "n", "lines", "MB", "us", "ns/line", "TFLOP/s");
    for (int n : sizes) {
        for (int items : {1, 2, 3, 6, 12, 24, 60}) {
            const size_t lines = (size_t)items * n;
            const size_t elems = lines * n;
            if (elems > max_elems) continue;
            const size_t len[1] = {(size_t)n};
            const double us = time_plan(1, len, lines, nullptr, 0, elems);
            if (us < 0) continue;
            const double flops = 5.0 * n * std::log2((double)n) * lines;
            printf("%6d %8zu %9.1f %10.2f %12.3f %10.2f\n", n, lines, elems * 8 / 1e6, us, us * 1e3 / lines, flops / us / 1e6);
        }
    }
    printf("\n## 1-D down the columns of a row-major [n][lines] array (stride = lines, distance = 1)\n");
    printf("%6s %8s %9s %10s %12s %10s\n", "n", "lines", "MB", "us", "ns/line", "TFLOP/s");
    for (int n : sizes) {
        for (int items : {1, 3, 12}) {
            const size_t lines = (size_t)items * n;
            const size_t elems = lines * n;
            if (elems > max_elems) continue;
            const size_t len[1] = {(size_t)n};
            const size_t str[1] = {lines};
            const double us = time_plan(1, len, lines, str, 1, elems);
            if (us < 0) continue;
            const double flops = 5.0 * n * std::log2((double)n) * lines;
            printf("%6d %8zu %9.1f %10.2f %12.3f %10.2f\n", n, lines, elems * 8 / 1e6, us, us * 1e3 / lines, flops / us / 1e6);
        }
    }
    printf("\n## 2-D n x n, batched (the whole per-source-point transform, dense)\n");
    printf("%6s %8s %9s %10s %12s %10s %14s\n", "n", "batch", "MB", "us", "us/image", "TFLOP/s", "ns/line(2n)");
    for (int n : sizes) {
        for (int batch : {1, 2, 3, 6, 12, 48}) {
            const size_t elems = (size_t)batch * n * n;
            if (elems > max_elems) continue;
            if (n >= 2048 && batch > 12) continue;
            const size_t len[2] = {(size_t)n, (size_t)n};
            const double us = time_plan(2, len, batch, nullptr, 0, elems);
            if (us < 0) continue;
            const double flops = 5.0 * n * std::log2((double)n) * 2.0 * n * batch;
            printf("%6d %8d %9.1f %10.2f %12.3f %10.2f %14.3f\n", n, batch, elems * 8 / 1e6, us, us / batch, flops / us / 1e6,
                   us * 1e3 / (2.0 * n * batch));
        }
    }
    rocfft_cleanup();
    return 0;
}
