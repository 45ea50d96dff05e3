// cumask_probe.hip -- where do the workgroups of a CU-masked stream run on MI355X (8 XCDs x 32 CUs, SPX mode)?
// hipExtStreamCreateWithCUMask takes a bit mask; this prints, for a few masks, the histogram of (XCC, SE, CU) ids the
// workgroups of a 4096-workgroup launch reported (s_getreg HW_ID / XCC_ID).
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/cumask_probe.hip -o scripts/ubench/cumask_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_where(unsigned* out)
{
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // burn a little time so that the launch spreads over every CU the mask allows
    float a = threadIdx.x;
    for (int i = 0; i < 20000; ++i) a = a * 1.0001f + 0.5f;
    if (threadIdx.x == 0) out[blockIdx.x] = (hw & 0xFFFFu) | ((xcc & 0xFu) << 16) | (a == 1.f ? 1u << 31 : 0u);
}

static void probe(const char* name, const std::vector<uint32_t>& mask)
{
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask.size(), mask.data()));
    const int n = 8192;
    unsigned* d; CK(hipMalloc(&d, n * 4));
    hipLaunchKernelGGL(k_where, dim3(n), dim3(64), 0, s, d);
    CK(hipStreamSynchronize(s));
    std::vector<unsigned> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    std::map<unsigned, int> perxcc, cus;
    for (unsigned v : h) {
        const unsigned xcc = (v >> 16) & 0xF, cu = (v >> 8) & 0xF, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        perxcc[xcc]++;
        cus[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("%-28s distinct (xcc,se,sh,cu) = %3zu | per XCC:", name, cus.size());
    for (auto& kv : perxcc) printf(" x%u:%d", kv.first, kv.second);
    std::map<unsigned, int> cu_per_xcc;
    for (auto& kv : cus) cu_per_xcc[kv.first >> 12]++;
    printf(" | CUs per XCC:");
    for (auto& kv : cu_per_xcc) printf(" %d", kv.second);
    printf("\n");
    CK(hipFree(d)); CK(hipStreamDestroy(s));
}

int main()
{
    std::vector<uint32_t> all(8, 0xFFFFFFFFu);
    probe("all 256 bits", all);
    probe("first 64 bits", {0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0, 0, 0, 0, 0});
    probe("first 32 bits", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0});
    probe("even bits of 256", std::vector<uint32_t>(8, 0x55555555u));
    probe("low byte of every word", std::vector<uint32_t>(8, 0x000000FFu));
    probe("word 0 = 0xFF only", {0x000000FFu, 0, 0, 0, 0, 0, 0, 0});
    probe("1 word mask 0xFFFF", {0x0000FFFFu});
    probe("bits 0..7 + 64..71", {0xFFu, 0, 0xFFu, 0, 0, 0, 0, 0});
    return 0;
}
