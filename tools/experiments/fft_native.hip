// Dispatcher of the native mixed-radix Fourier rows (kernel and launcher template: fft_native_impl.h; instances: fft_native_i*.hip)
#include "fft_native_impl.h"

namespace atlas_amd {
namespace trans {

extern template hipError_t launch_nat_t<false, false, 1>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<false, true, 1>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<false, false, 2>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<false, true, 2>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<true, false, 1>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<true, true, 1>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<true, false, 2>(FourierParams, int, hipStream_t);
extern template hipError_t launch_nat_t<true, true, 2>(FourierParams, int, hipStream_t);

// lds_bytes: the work arrays of the largest row of the launch (fields_per_job of them) + the prefetch dump area, in 16-byte elements;
// bigp: the first-stage radices of the launch's rows are primes 17 .. 31 (else 3 .. 15)
hipError_t launch_fourier_nat(const FourierParams& p, int lds_bytes, int bigp, int fields_per_job, hipStream_t stream) {
    if (!p.ndesc || !p.nat_table || (fields_per_job != 1 && fields_per_job != 2)) {
        return hipErrorInvalidValue;
    }
    const int v = (p.f32 ? 4 : 0) | (bigp ? 2 : 0) | (fields_per_job == 2 ? 1 : 0);
    switch (v) {
        case 0: return launch_nat_t<false, false, 1>(p, lds_bytes, stream);
        case 1: return launch_nat_t<false, false, 2>(p, lds_bytes, stream);
        case 2: return launch_nat_t<false, true, 1>(p, lds_bytes, stream);
        case 3: return launch_nat_t<false, true, 2>(p, lds_bytes, stream);
        case 4: return launch_nat_t<true, false, 1>(p, lds_bytes, stream);
        case 5: return launch_nat_t<true, false, 2>(p, lds_bytes, stream);
        case 6: return launch_nat_t<true, true, 1>(p, lds_bytes, stream);
        default: return launch_nat_t<true, true, 2>(p, lds_bytes, stream);
    }
}

}  // namespace trans
}  // namespace atlas_amd
