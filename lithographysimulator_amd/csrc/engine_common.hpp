// engine_common.hpp -- error plumbing shared by the translation units of liblitho_abbe.so.
#pragma once
#include <hip/hip_runtime.h>

namespace litho {
// Records "<expr>: <hip error string>" for litho_last_error() (thread local).
void set_last_error(const char* what, hipError_t e);
}  // namespace litho

#define HIP_TRY(expr)                                   \
    do {                                                \
        hipError_t _e = (expr);                         \
        if (_e != hipSuccess) {                         \
            litho::set_last_error(#expr, _e);           \
            return LITHO_E_HIP;                         \
        }                                               \
    } while (0)
