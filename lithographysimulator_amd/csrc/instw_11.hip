// instw_11.hip -- wave-level y-pass kernels for FFT size N = 2048 (own translation unit: its own scheduling flags, Makefile WAVEFLAGS_11 -- default strategy today).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(11)
}
