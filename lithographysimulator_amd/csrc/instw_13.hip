// instw_13.hip -- wave-level y-pass kernels for FFT size N = 8192 (own translation unit: max-ILP scheduling).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(13)
}
