// instw_09.hip -- wave-level y-pass kernels for FFT size N = 512 (own translation unit: its own scheduling flags, Makefile WAVEFLAGS_09 -- default strategy today).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(9)
}
