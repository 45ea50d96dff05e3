// common.hip -- version / error reporting entry points of the C ABI.
#include <hip/hip_runtime.h>

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
int device_cus()
{
    static int cached[64];                       // 0 = not asked yet (benign race: every thread writes the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 256; }
        cached[dev] = n;
    }
    return cached[dev];
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
