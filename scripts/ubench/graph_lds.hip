// Does a kernel node of a captured HIP graph get its large dynamic LDS?  A 512-thread kernel with 133,120 bytes of dynamic
// LDS (the shape of k_ypass_rect<., ., ., 2>) writes a pattern through the far end of it, synchronises, reads it back
// through another wave and reports mismatches -- launched directly, then replayed from a stream-captured graph, with
// the input changing between replays.   hipcc --offload-arch=gfx950 -O2 scripts/ubench/graph_lds.hip -o graph_lds.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(512, 2) void k_far_lds(const float* in, float* out, int per_wave_floats)
{
    extern __shared__ float smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* mine = smem + wv * per_wave_floats;
    const float v = in[blockIdx.x * 512 + threadIdx.x];
    for (int i = 0; i < 64; ++i) mine[i * 65 + lane] = v + i;          // the wave's own 64 x 65 matrix
    __syncthreads();
    const float* other = smem + ((wv + 4) & 7) * per_wave_floats;      // read the matrix of the wave four further on
    float s = 0.f;
    for (int i = 0; i < 64; ++i) s += other[i * 65 + lane];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main()
{
    const int blocks = 1024, n = blocks * 512, pw = 64 * 65;
    const size_t lds = 8 * (size_t)pw * sizeof(float);                 // 133,120 bytes
    float *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipFuncSetAttribute((const void*)k_far_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<float> h(n), r(n);
    hipStream_t st; CK(hipStreamCreate(&st));
    auto fill = [&](float base) { for (int i = 0; i < n; ++i) h[i] = base + (i % 977); return hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice); };
    auto check = [&](const char* what) {
        hipMemcpy(r.data(), out, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int b = 0; b < blocks; ++b)
            for (int t = 0; t < 512; ++t) {
                const int src = b * 512 + ((((t >> 6) + 4) & 7) << 6) + (t & 63);
                const float want = 64.f * h[src] + 2016.f;
                if (r[b * 512 + t] != want) ++bad;
            }
        printf("%-28s mismatches %ld of %d\n", what, bad, n);
    };
    CK(fill(1.f));
    hipLaunchKernelGGL(k_far_lds, dim3(blocks), dim3(512), lds, st, in, out, pw);
    CK(hipStreamSynchronize(st)); check("direct launch");
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k_far_lds, dim3(blocks), dim3(512), lds, st, in, out, pw);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st)); check("graph replay, same input");
    CK(fill(5000.f));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st)); check("graph replay, new input");
    CK(fill(1.f));
    CK(hipGraphLaunch(exec, st)); CK(hipStreamSynchronize(st)); check("graph replay, first input");
    hipLaunchKernelGGL(k_far_lds, dim3(blocks), dim3(512), lds, st, in, out, pw);
    CK(hipStreamSynchronize(st)); check("direct launch again");
    return 0;
}
