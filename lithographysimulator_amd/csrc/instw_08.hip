// instw_08.hip -- wave-level y-pass kernel for FFT size N = 256 (the coarse-grid transform of 256^2 images).
#include "wave_kernels.hpp"
namespace litho {
template hipError_t launch_ypass_wave<8>(const float2*, float*, const float2*, const PassGeom&, int, int, int, int, hipStream_t);
}
