// VALU issue-rate microbenchmark for gfx950: cycles per instruction per SIMD for scalar and packed fp32 ops
// at 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define REP16(x) x x x x x x x x x x x x x x x x
template <int KIND>
__global__ void k(float* out, int iters, float seed)
{
    float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
    f2 p0 = {seed, 1}, p1 = {seed, 2}, p2 = {seed, 3}, p3 = {seed, 4}, p4 = {seed, 5}, p5 = {seed, 6}, p6 = {seed, 7}, p7 = {seed, 8};
    const float c = 1.0001f; const f2 pc = {1.0001f, 0.9999f};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { REP16(asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 1) { REP16(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 2) { REP16(asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));) }
        if (KIND == 3) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));) }
        if (KIND == 4) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));) }
        if (KIND == 5) { REP16(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 8) { REP16(asm volatile("v_fmamk_f32 %0, %0, 0x3f3504f3, %8\n v_fmamk_f32 %1, %1, 0x3f3504f3, %8\n v_fmamk_f32 %2, %2, 0x3f3504f3, %8\n v_fmamk_f32 %3, %3, 0x3f3504f3, %8\n v_fmamk_f32 %4, %4, 0x3f3504f3, %8\n v_fmamk_f32 %5, %5, 0x3f3504f3, %8\n v_fmamk_f32 %6, %6, 0x3f3504f3, %8\n v_fmamk_f32 %7, %7, 0x3f3504f3, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 9) { REP16(asm volatile("v_mul_f32 %0, 0x3f6c835e, %0\n v_mul_f32 %1, 0x3f6c835e, %1\n v_mul_f32 %2, 0x3f6c835e, %2\n v_mul_f32 %3, 0x3f6c835e, %3\n v_mul_f32 %4, 0x3f6c835e, %4\n v_mul_f32 %5, 0x3f6c835e, %5\n v_mul_f32 %6, 0x3f6c835e, %6\n v_mul_f32 %7, 0x3f6c835e, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 10) { REP16(asm volatile("v_mul_f32 %0, %8, %0\n v_mul_f32 %1, %8, %1\n v_mul_f32 %2, %8, %2\n v_mul_f32 %3, %8, %3\n v_mul_f32 %4, %8, %4\n v_mul_f32 %5, %8, %5\n v_mul_f32 %6, %8, %6\n v_mul_f32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));) }
        if (KIND == 11) { REP16(asm volatile("v_fmac_f32 %0, %8, %1\n v_fmac_f32 %1, %8, %2\n v_fmac_f32 %2, %8, %3\n v_fmac_f32 %3, %8, %4\n v_fmac_f32 %4, %8, %5\n v_fmac_f32 %5, %8, %6\n v_fmac_f32 %6, %8, %7\n v_fmac_f32 %7, %8, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(seed));) }
        if (KIND == 12) { REP16(asm volatile("v_fmac_f32 %0, 0x3f6c835e, %1\n v_fmac_f32 %1, 0x3f6c835e, %2\n v_fmac_f32 %2, 0x3f6c835e, %3\n v_fmac_f32 %3, 0x3f6c835e, %4\n v_fmac_f32 %4, 0x3f6c835e, %5\n v_fmac_f32 %5, 0x3f6c835e, %6\n v_fmac_f32 %6, 0x3f6c835e, %7\n v_fmac_f32 %7, 0x3f6c835e, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        if (KIND == 13) { REP16(asm volatile("v_sub_f32 %0, %1, %2\n v_add_f32 %1, %2, %3\n v_sub_f32 %2, %3, %4\n v_add_f32 %3, %4, %5\n v_sub_f32 %4, %5, %6\n v_add_f32 %5, %6, %7\n v_sub_f32 %6, %7, %0\n v_add_f32 %7, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));) }
        // dependent chain: one accumulator
        if (KIND == 6) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a0) : "v"(c));) }
        if (KIND == 7) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pc));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int KIND> double run(int waves_per_simd, float* d, const char* name)
{
    const int iters = 2000, threads = 256;                      // 4 waves per block = 1 per SIMD
    const int blocks = 256 * waves_per_simd;                    // 256 CUs
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, threads>>>(d, 10, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(a); k<KIND><<<blocks, threads>>>(d, iters, 1.f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr_per_simd = (double)iters * 16 * 8 * waves_per_simd;
    const double ns_per_instr = ms * 1e6 / instr_per_simd;
    printf("%-14s waves/SIMD=%d  %.3f ns per instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ns_per_instr, ns_per_instr * 2.4);
    return ns_per_instr;
}
int main()
{
    float* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {2, 3}) {
        run<0>(w, d, "v_add_f32"); run<1>(w, d, "v_fma_f32"); run<2>(w, d, "v_pk_add_f32"); run<3>(w, d, "v_pk_fma_f32");
        run<8>(w, d, "fmamk literal"); run<9>(w, d, "mul literal"); run<10>(w, d, "mul sgpr"); run<11>(w, d, "fmac sgpr"); run<12>(w, d, "fmac literal"); run<13>(w, d, "add/sub 3op");
    }
    return 0;
}
