// Read bandwidth by footprint, access pattern, loads in flight per wave and waves per SIMD:
// what does the memory system deliver to the y-pass, which re-reads a batch of T (12 x 16.8 MB) out of the
// 256 MiB Infinity Cache?   hipcc --offload-arch=gfx950 -O3 scripts/ubench/read_bw.hip -o scripts/ubench/read_bw.bin
// Pattern A: contiguous 16 B per lane (a wave instruction = 1 KB).  Pattern Y: the y-pass pattern on 8-column tiles,
// 16 B per lane at a 64-byte lane stride (a wave instruction = 64 row granules, a quarter of each used; the four
// waves of a workgroup take the four quarters).  Each wave issues U loads, waits for all of them, then "computes"
// for `work` dependent FMAs per load (0 = pure streaming).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}

// One "chunk" = U wave-instructions.  PATTERN 0: chunk = U KB contiguous, wave-private.  PATTERN 1: a workgroup's four
// waves share U * 4 KB (wave w takes quarter w of every 64-byte granule).
template <int U, int PATTERN, int WPS>
__global__ __launch_bounds__(256, WPS) void k_read(const float* __restrict__ buf, size_t bytes, float* out, int work)
{
    extern __shared__ float lds_pad[];                          // occupancy control only
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    // PATTERN 2: 8 B per lane at a 64-byte stride (one column per wave): the workgroup's four waves take columns
    // 0..3 or 4..7 of the tile (blockIdx parity), so eight waves in two workgroups share every granule
    const size_t chunk = PATTERN == 0 ? (size_t)U * 1024 : (size_t)U * 4096;
    const size_t nchunks = bytes / chunk;
    const size_t nw = PATTERN == 0 ? (size_t)gridDim.x * 4 : PATTERN == 1 ? gridDim.x : gridDim.x / 2;
    size_t c = PATTERN == 0 ? (size_t)blockIdx.x * 4 + wv : PATTERN == 1 ? blockIdx.x : blockIdx.x / 2;
    float acc = 0.f;
    for (; c < nchunks; c += nw) {
        const char* base = reinterpret_cast<const char*>(buf) + c * chunk;
        const __amdgpu_buffer_rsrc_t r = rsrc(base, chunk);
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if constexpr (PATTERN == 2) {
                typedef unsigned int u2 __attribute__((ext_vector_type(2)));
                const u2 w = __builtin_amdgcn_raw_buffer_load_b64(r, (unsigned)(u * 4096 + lane * 64 + ((blockIdx.x & 1) * 4 + wv) * 8), 0, 0);
                v[u].x = w.x; v[u].y = w.y; v[u].z = 0; v[u].w = 0;
            } else {
                const unsigned off = PATTERN == 0 ? (unsigned)(u * 1024 + lane * 16) : (unsigned)(u * 4096 + lane * 64 + wv * 16);
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float t = __uint_as_float(v[u].x) + __uint_as_float(v[u].y) + __uint_as_float(v[u].z) + __uint_as_float(v[u].w);
            for (int i = 0; i < work; ++i) t = fmaf(t, 1.0001f, 0.5f);
            acc += t;
        }
    }
    if (acc == 123.456f) out[threadIdx.x] = acc;
}

template <typename F>
static double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    f();
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}

template <int U, int PATTERN, int WPS>
static void run(const float* buf, size_t bytes, float* out, int work)
{
    // WPS waves per SIMD resident: 256-thread workgroups, WPS per CU -> LDS padding forces the limit
    auto kern = k_read<U, PATTERN, WPS>;
    const size_t lds = WPS >= 8 ? 0 : (size_t)(160 * 1024 / WPS) - 1024;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int blocks = 256 * WPS;
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, buf, bytes, out, work); }, 10);
    printf("  %s  U=%2d loads in flight/wave  %d waves/SIMD  work %3d : %6.2f TB/s  (%.1f us per pass)\n",
           PATTERN == 0 ? "contiguous  " : PATTERN == 1 ? "y-pass tiles" : "1 col / wave", U, WPS, work, bytes / ms / 1e9, ms * 1e3);
}

int main()
{
    float *buf, *out;
    const size_t cap = (size_t)2200 << 20;
    hipMalloc(&buf, cap);
    hipMalloc(&out, 4096);
    hipMemset(buf, 0, cap);
    for (size_t mb : {201}) {
        const size_t bytes = mb * 1000000 / 69632 * 69632;      // multiple of 17 x 4096
        printf("footprint %zu MB, re-read 10x\n", mb);
        run<8, 0, 8>(buf, bytes, out, 0);
        run<8, 0, 2>(buf, bytes, out, 0);
        run<17, 0, 2>(buf, bytes, out, 0);
        run<17, 1, 2>(buf, bytes, out, 0);
        run<17, 1, 2>(buf, bytes, out, 40);
        run<17, 1, 2>(buf, bytes, out, 80);
        run<17, 1, 3>(buf, bytes, out, 80);
        run<17, 1, 4>(buf, bytes, out, 80);
        run<8, 1, 4>(buf, bytes, out, 80);
        run<8, 1, 8>(buf, bytes, out, 80);
        run<4, 1, 8>(buf, bytes, out, 0);
        run<17, 2, 2>(buf, bytes, out, 0);
        run<17, 2, 4>(buf, bytes, out, 0);
        run<8, 2, 8>(buf, bytes, out, 0);
    }
    return 0;
}
