// instw_12.hip -- wave-level y-pass kernels for FFT size N = 4096 (own translation unit: its own scheduling flags, Makefile WAVEFLAGS_12 -- default strategy today).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(12)
}
