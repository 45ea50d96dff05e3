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
}  // namespace litho

extern "C" {
int litho_version(void) { return 100; }
#ifdef LITHO_DIAG_BUILD
const char* litho_target_arch(void) { return "gfx950-diag"; }      // timing-diagnostic build: results may be wrong
#else
const char* litho_target_arch(void) { return "gfx950"; }
#endif
const char* litho_last_error(void) { return litho::g_err; }
}
