// instw_10.hip -- wave-level y-pass kernels for FFT size N = 1024 (own translation unit: its own scheduling flags, Makefile WAVEFLAGS_10 -- default strategy today).
#include "wave_kernels.hpp"
namespace litho {
LITHO_DEFINE_WAVE_OPS(10)
}
