// Dev probe (GPU box): L2 -> L1 read throughput per CU for the access shapes of the transform kernels.
//   hipcc --offload-arch=gfx950 -O3 -o probe_l2_gfx950 probe_l2_gfx950.hip
// Every workgroup repeatedly reads a window of `win` bytes (L2 resident, larger than L1) with
//   mode 0: 16 B per lane, consecutive lanes consecutive (coalesced dwordx4)
//   mode 1: 16 B per lane, one lane per 128-byte line   (the Fourier gather)
//   mode 2:  8 B per lane, coalesced                     (the spectra operand of the Legendre kernel)
//   mode 3: 16 B per lane, one lane per 2304-byte pitch  (the Fourier gather at the real pitch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int UNROLL>
__global__ void __launch_bounds__(256) reader(const double* __restrict__ buf, long long win_doubles, int iters,
                                              double* out) {
    // window of this workgroup: blocks on the same XCD (b % 8) share a small set of windows -> L2 hits
    const long long nwin = MODE == 3 ? 1 : 4;
    const double* w      = buf + ((blockIdx.x % 8) * nwin + (blockIdx.x / 8) % nwin) * win_doubles;
    double acc0 = 0, acc1 = 0;
    const int t = threadIdx.x;
    long long stride_d, span;  // per-thread element stride (doubles) between consecutive lanes; bytes covered per sweep
    if (MODE == 0) stride_d = 2;
    if (MODE == 1) stride_d = 16;
    if (MODE == 2) stride_d = 1;
    if (MODE == 3) stride_d = 288;
    span = 256 * stride_d;  // doubles covered by one workgroup load
    const long long nsweep = win_doubles / (span * UNROLL);
    for (int it = 0; it < iters; ++it) {
        for (long long s = 0; s < nsweep; ++s) {
            const double* p = w + s * span * UNROLL + t * stride_d;
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                if (MODE == 2) {
                    acc0 += p[u * span];
                }
                else {
                    const double2 v = *reinterpret_cast<const double2*>(p + u * span);
                    acc0 += v.x;
                    acc1 += v.y;
                }
            }
        }
    }
    if (acc0 + acc1 == 12345.678) out[0] = acc0;
}

template <int MODE, int UNROLL>
void run(const char* name, const double* buf, long long win_bytes, int nblocks, double* out) {
    const long long wd = win_bytes / 8;
    const int iters    = 20;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL((reader<MODE, UNROLL>), dim3(nblocks), dim3(256), 0, 0, buf, wd, 2, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((reader<MODE, UNROLL>), dim3(nblocks), dim3(256), 0, 0, buf, wd, iters, out);
    CK(hipEventRecord(b));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    long long stride_d = MODE == 0 ? 2 : MODE == 1 ? 16 : MODE == 2 ? 1 : 288;
    const long long span   = 256 * stride_d;
    const long long nsweep = wd / (span * UNROLL);
    const double loads     = (double)nblocks * iters * nsweep * UNROLL * 256;  // lane loads
    const double useful    = loads * (MODE == 2 ? 8 : 16);
    const double lines     = loads * (MODE == 0 ? 16.0 / 128 : MODE == 2 ? 8.0 / 128 : 1.0);
    printf("%-34s blocks %5d  %8.3f ms  useful %7.2f TB/s  lines*128B %7.2f TB/s  %6.2f Glines/s  (%.3f lines/clk/CU @2.3GHz)\n",
           name, nblocks, ms, useful / ms / 1e9, lines * 128 / ms / 1e9, lines / ms / 1e6,
           lines / (ms * 1e-3) / 256 / 2.3e9);
}

int main() {
    const long long win = 512 << 10;  // 512 KiB windows, 4 per XCD = 2 MiB per XCD L2 (4 MiB)
    double *buf, *out;
    CK(hipMalloc(&buf, 8 * 1024LL * 2304 + 8 * 4 * win));
    CK(hipMemset(buf, 0, 8 * 1024LL * 2304 + 8 * 4 * win));
    CK(hipMalloc(&out, 64));
    for (int nb : {256 * 2, 256 * 4, 256 * 8}) {
        run<0, 8>("coalesced 16B/lane", buf, win, nb, out);
        run<2, 8>("coalesced 8B/lane", buf, win, nb, out);
        run<1, 8>("16B per 128B line", buf, win, nb, out);
        run<3, 4>("16B per 2304B pitch", buf, 1024LL * 2304, nb, out);
    }
    return 0;
}
