// engine_common.hpp -- error plumbing shared by the translation units of liblitho_abbe.so.
#pragma once
#include <hip/hip_runtime.h>

namespace litho {
// Records "<expr>: <hip error string>" for litho_last_error() (thread local).
void set_last_error(const char* what, hipError_t e);
// Remembers (thread local, formatted lazily by litho_abbe_last_kernels) which kernel the Abbe loop launched last for
// pass 0 (x) / 1 (y): fmt is a string literal with up to four %d, spelt the way rocprofv3 prints the kernel, without
// "void litho::" and the argument list -- so bench.py and the profiles/ summaries can be matched by name.
void note_kernel(int pass, const char* fmt, int a0 = 0, int a1 = 0, int a2 = 0, int a3 = 0);
// Compute units of the CURRENT device (hipDeviceAttributeMultiprocessorCount, cached per device; 256 on an MI355X in
// SPX mode, 32 per partition in CPX): the launch planners size their grids in whole rounds of it.
int device_cus();
}  // namespace litho

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) {                         \
            litho::set_last_error(#expr, _e);           \
            return LITHO_E_HIP;                         \
        }                                               \
    } while (0)
