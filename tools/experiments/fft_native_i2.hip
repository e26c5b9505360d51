// Instances of the native mixed-radix Fourier kernel: fp32, 1 field per workgroup (fft_native_impl.h)
#include "fft_native_impl.h"

namespace atlas_amd {
namespace trans {
template hipError_t launch_nat_t<true, false, 1>(FourierParams, int, hipStream_t);
template hipError_t launch_nat_t<true, true, 1>(FourierParams, int, hipStream_t);
}  // namespace trans
}  // namespace atlas_amd
