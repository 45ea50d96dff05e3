// inst_13.hip -- kernel instantiations for FFT size N = 8192 (one translation unit per size so
// that the sizes compile in parallel).
#include "engine_kernels.hpp"
namespace litho {
LITHO_DEFINE_SIZE_OPS(13)
}
