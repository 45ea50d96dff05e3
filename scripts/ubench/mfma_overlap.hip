// Bounded MFMA experiment: can the idle fp32 matrix pipe take FFT butterfly work off the VALU?
//   V   radix-4 butterflies on the VALU (16 real add/sub per butterfly, one butterfly per lane per step)
//   M   the same butterflies as a dense real 8x8 DFT-matrix product on v_mfma_f32_16x16x4_f32
//       (two butterflies packed block-diagonally in a 16x16 matrix: 4 MFMAs of K = 4 per 16 columns -> 32 butterflies)
//   VM  both instruction streams interleaved in one wave (independent registers): do the pipes overlap?
// Output: cycles per butterfly per SIMD for each, and how much V slows down when M runs beside it.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 1 = V, 2 = M, 3 = VM
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed)
{
    float ar = seed, ai = seed + 1, br = seed + 2, bi = seed + 3, cr = seed + 4, ci = seed + 5, dr = seed + 6, di = seed + 7;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0}, acc3 = {0, 0, 0, 0};
    const float ma = seed * 0.5f, mb = seed * 0.25f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE & 1) {            // radix-4 butterfly (+i rotation is a swap/negate folded into the add/sub pattern)
                const float t0r = ar + cr, t0i = ai + ci, t1r = ar - cr, t1i = ai - ci;
                const float t2r = br + dr, t2i = bi + di, t3r = bi - di, t3i = dr - br;
                ar = t0r + t2r; ai = t0i + t2i; cr = t0r - t2r; ci = t0i - t2i;
                br = t1r + t3r; bi = t1i + t3i; dr = t1r - t3r; di = t1i - t3i;
                asm volatile("" : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "+v"(cr), "+v"(ci), "+v"(dr), "+v"(di));
            }
            if (MODE & 2) {            // 4 MFMAs (K = 16) = 32 butterflies of 16 columns x 2 block-diagonal halves
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, mb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, ma, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(ma, ma, acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(mb, mb, acc3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = ar + ai + br + bi + cr + ci + dr + di + acc0.x + acc1.y + acc2.z + acc3.w;
}
template <int MODE> static double run(float* d, int waves_per_simd)
{
    const int iters = 4000;
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k<MODE><<<256 * waves_per_simd, 256>>>(d, 10, 1.f);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a); k<MODE><<<256 * waves_per_simd, 256>>>(d, iters, 1.f); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    return ms * 1e6 / ((double)iters * 8 * waves_per_simd);        // ns per unrolled step per SIMD
}
int main()
{
    float* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    for (int w : {1, 2}) {
        const double v = run<1>(d, w), m = run<2>(d, w), vm = run<3>(d, w);
        // one step: V = 64 butterflies (one per lane); M = 32 butterflies (4 MFMAs)
        printf("waves/SIMD=%d  V %.2f ns/step = %.3f ns/butterfly | M %.2f ns/step = %.3f ns/butterfly (%.1fx the VALU cost) | "
               "VM %.2f ns/step (V+M serial would be %.2f, perfect overlap %.2f)\n",
               w, v, v / 64, m, m / 32, (m / 32) / (v / 64), vm, v + m, v > m ? v : m);
    }
    return 0;
}
