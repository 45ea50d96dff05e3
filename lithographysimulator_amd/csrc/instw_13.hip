// instw_13.hip -- wave-level y-pass kernels for FFT size N = 8192 (own translation unit: its own scheduling flags, Makefile WAVEFLAGS_13 -- default strategy today).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(13)
}
