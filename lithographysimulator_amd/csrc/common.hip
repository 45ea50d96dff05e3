// common.hip -- version / error reporting entry points of the C ABI.
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>

#include "../../include/litho_abbe.h"
#include "engine_common.hpp"

namespace litho {
static thread_local char g_err[512] = "";
void set_last_error(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    (void)hipGetLastError();     // clear the sticky error so later calls start clean
}
struct KernelNote {
    const char* fmt;
    int a[4];
};
static thread_local KernelNote g_note[2] = {{nullptr, {0, 0, 0, 0}}, {nullptr, {0, 0, 0, 0}}};
void note_kernel(int pass, const char* fmt, int a0, int a1, int a2, int a3)
{
    if (pass < 0 || pass > 1) return;
    g_note[pass] = KernelNote{fmt, {a0, a1, a2, a3}};
}
// Compute units of the current device (launch planning: groups, grids).  The C ABI may be called from several host threads, one
// stream each: the per-device cache is atomic (relaxed: every writer stores the same value).  A failing query is REPORTED
// through litho_last_error and answered with MI355X's 256 -- on a partitioned device (CPX: 32 CUs) that would over-size G.
int device_cus()
{
    static std::atomic<int> cached[64];          // 0 = not asked yet
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess || dev < 0 || dev >= 64) {
        set_last_error("device_cus: hipGetDevice failed or device index >= 64; assuming 256 compute units", e);
        return 256;
    }
    int n = cached[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || n < 1) {
            set_last_error("device_cus: hipDeviceGetAttribute(MultiprocessorCount) failed; assuming 256 compute units", e);
            n = 256;
        }
        cached[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}
}  // namespace litho

extern "C" {
int litho_abbe_last_kernels(char* xpass_host, char* ypass_host, size_t capacity)
{
    if (!xpass_host || !ypass_host || capacity == 0) return LITHO_E_ARG;
    char* dst[2] = {xpass_host, ypass_host};
    for (int p = 0; p < 2; ++p) {
        const litho::KernelNote& n = litho::g_note[p];
        if (n.fmt) snprintf(dst[p], capacity, n.fmt, n.a[0], n.a[1], n.a[2], n.a[3]);
        else dst[p][0] = 0;
    }
    return LITHO_OK;
}
int litho_version(void) { return 100; }
#ifdef LITHO_DIAG_BUILD
const char* litho_target_arch(void) { return "gfx950-diag"; }      // timing-diagnostic build: results may be wrong
#else
const char* litho_target_arch(void) { return "gfx950"; }
#endif
const char* litho_last_error(void) { return litho::g_err; }
}
