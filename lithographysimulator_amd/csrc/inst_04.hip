// inst_04.hip -- kernel instantiations for FFT size N = 16 (one translation unit per size so
// that the sizes compile in parallel).
#include "engine_kernels.hpp"
namespace litho {
LITHO_DEFINE_SIZE_OPS(4)
}
