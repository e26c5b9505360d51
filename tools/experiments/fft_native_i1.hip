// Instances of the native mixed-radix Fourier kernel: fp64, 2 fields per workgroup (fft_native_impl.h)
#include "fft_native_impl.h"

namespace atlas_amd {
namespace trans {
template hipError_t launch_nat_t<false, false, 2>(FourierParams, int, hipStream_t);
template hipError_t launch_nat_t<false, true, 2>(FourierParams, int, hipStream_t);
}  // namespace trans
}  // namespace atlas_amd
