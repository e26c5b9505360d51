// Device generation of the tile-blocked Legendre table (f3 of SURVEY 8: the reference builds its tables serially
// on the host, LegendrePolynomials.cc:154-209 -- about 80 s on one core at T1279).  The arithmetic lives in
// legendre_gen_core.h; this file only maps it onto the device:
//   legendre_series_kernel   one thread per (Legendre row, n):  P(0,n), P(1,n)
//   legendre_chain_kernel    one thread per (Legendre row, parity of m): the recurrence rows m = parity, parity+2, ...
// 64 consecutive Legendre rows per wavefront, so every access to the per-latitude arrays, the row scratch and a
// table tile (64 latitudes wide) is one contiguous 512-byte line per wavefront; the recurrence coefficients are
// wave-uniform.  The chains are latency-bound (3 dependent fp64 operations per step, ~T^2/4 steps), not bandwidth-bound:
// 2*nlats/64 wavefronts run side by side and finish in milliseconds.
// Built with -ffp-contract=off (Makefile) in addition to the pragmas in the core: no fused multiply-add, so the
// results equal the host generator's bit for bit.
#include <hip/hip_runtime.h>

#include "legendre_gen_core.h"

namespace atlas_amd {
namespace trans {

__global__ void __launch_bounds__(64) legendre_series_kernel(LegendreGenParams g) {
    const int lat = blockIdx.x * LG_TILE + threadIdx.x;
    const int jn  = blockIdx.y;
    if (lat < g.nlats) {
        legendre_series_store(g, lat, jn);
    }
}

__global__ void __launch_bounds__(64) legendre_chain_kernel(LegendreGenParams g) {
    const int lat    = (blockIdx.x >> 1) * LG_TILE + threadIdx.x;
    const int parity = blockIdx.x & 1;
    if (lat < g.nlats) {
        legendre_chain(g, lat, parity);
    }
}

hipError_t launch_legendre_gen(const LegendreGenParams& g, hipStream_t stream) {
    if (g.nlats <= 0) {
        return hipSuccess;
    }
    const int nblk = g.lat_pitch / LG_TILE;
    hipLaunchKernelGGL(legendre_series_kernel, dim3(nblk, g.trc + 1), dim3(LG_TILE), 0, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        return e;
    }
    hipLaunchKernelGGL(legendre_chain_kernel, dim3(2 * nblk), dim3(LG_TILE), 0, stream, g);
    return hipGetLastError();
}

}  // namespace trans
}  // namespace atlas_amd
