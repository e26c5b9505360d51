// Dev probe (GPU box): what does the instruction stream of the Fourier kernels cost on a SIMD, by itself?
//   K_reg : radix-16 butterfly + twiddles of fft_core.h in registers, in a loop (no LDS, no memory): the fp64 VALU rate
//           this instruction mix reaches at 1 / 2 / 3 / 4 wavefronts per SIMD
//   K_fma : independent v_fma_f64 chains (NACC per lane): the plain FMA issue rate at the same occupancies
//   K_lds : one wave-local radix-16 DIF + DIT stage pair on 256-point blocks in LDS (16 x ds_read_b128, butterfly,
//           16 x ds_write_b128; wavefront-level ordering only): a "middle phase" with no global memory
// Occupancy is pinned with dynamic LDS (workgroups of 256 threads = one wavefront per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../atlas_amd/csrc/fft_core.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
using atlas_amd::fft::cplx;
namespace F = atlas_amd::fft;

template <int R>
__global__ void __launch_bounds__(256) k_reg(int iters, double seed, double* out) {
    extern __shared__ double lds[];
    cplx x[R];
#pragma unroll
    for (int q = 0; q < R; ++q) x[q] = cplx{seed * (threadIdx.x + q), seed * (q + 1)};
    cplx w1{0.9999 + seed * threadIdx.x, 0.0141418 - seed * threadIdx.x};   // per lane, as in the stages: the powers are computed
    for (int it = 0; it < iters; ++it) {
        F::bfly<R>(x, -1);
        F::twiddle_apply<R>(x, w1);
#pragma unroll
        for (int q = 0; q < R; ++q) {   // keep magnitudes bounded without extra arithmetic: exponent-only scale is not free, so
            x[q].re *= 0.25;           // a plain multiply per component (2R extra VALU of ~R*(8+8+..))
            x[q].im *= 0.25;
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) s += x[q].re + x[q].im;
    if (s == 1.2345) out[0] = s + lds[0];
}

template <int NACC>
__global__ void __launch_bounds__(256) k_fma(int iters, double seed, double* out) {
    extern __shared__ double lds[];
    double a[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) a[q] = seed * (threadIdx.x + q);
    const double m = 0.999999, c = 1e-9 * seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) a[q] = __builtin_fma(a[q], m, c);
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += a[q];
    if (s == 1.2345) out[0] = s + lds[0];
}

// add / mul / fma in the proportions of the butterflies (45 % add, 26 % mul, 29 % fma), independent chains
template <int NACC>
__global__ void __launch_bounds__(256) k_mix(int iters, double seed, double* out) {
    extern __shared__ double lds[];
    double a[NACC];
#pragma unroll
    for (int q = 0; q < NACC; ++q) a[q] = seed * (threadIdx.x + q);
    const double m = 0.999999, c = 1e-9 * seed;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int q = 0; q < NACC; ++q) a[q] = a[q] + c;
#pragma unroll
            for (int q = 0; q < NACC; ++q) a[q] = a[q] * m;
#pragma unroll
            for (int q = 0; q < NACC; ++q) a[q] = __builtin_fma(a[q], m, c);
#pragma unroll
            for (int q = 0; q < NACC; ++q) a[q] = a[q] + a[(q + 1) % NACC] * 0.0;
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < NACC; ++q) s += a[q];
    if (s == 1.2345) out[0] = s + lds[0];
}

// one 256-point block per 16 lanes; a DIF stage (L = 256) followed by the DIT stage that undoes it (x16 growth per pair)
__global__ void __launch_bounds__(256) k_lds(int iters, int lds_used_cplx, const cplx* tw, double* out) {
    extern __shared__ double lds[];
    cplx* work = reinterpret_cast<cplx*>(lds);
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 256) work[F::PAD(i)] = cplx{1e-100 * (i + 1), 1e-100 * (i & 7)};
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        F::dif_stage<16>(work, 4096, 256, 4, tw, -1, t, 256);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        F::dit_stage<16>(work, 4096, 256, 4, tw, +1, t, 256);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    if (work[F::PAD(t)].re == 1.2345) out[0] = 1 + lds_used_cplx;
}

template <class K, class... A>
static float time_kernel(K kern, int blocks, int lds, A... args) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, 0, args...);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    return best;
}

int main() {
    double* out;
    CK(hipMalloc(&out, 64));
    cplx* tw;
    {
        cplx h[4096];
        for (int i = 0; i < 4096; ++i) h[i] = cplx{cos(2 * M_PI * i / 4096), sin(2 * M_PI * i / 4096)};
        CK(hipMalloc(&tw, sizeof(h)));
        CK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
    }
    const double GHZ = 2.4;
    // LDS per workgroup that allows exactly w workgroups (= w wavefronts per SIMD) per CU
    auto lds_for = [](int w) { return (160 * 1024 / w) / 1024 * 1024 - (w == 1 ? 0 : 0); };
    for (int w : {1, 2, 3, 4, 6, 8}) {
        const int lds = w <= 2 ? (w == 1 ? 160 * 1024 : 80 * 1024) : lds_for(w);
        const int blocks = 256 * w;
        {
            const int iters = 2000;
            float ms = time_kernel(k_reg<16>, blocks, lds, iters, 1e-3, out);
            printf("k_reg<16>  %d waves/SIMD: %.3f ms -> %.0f SIMD-cycles per butterfly+twiddle (per wave: %.0f) at %.1f GHz\n", w, ms,
                   ms * 1e-3 * GHZ * 1e9 / (iters * w), ms * 1e-3 * GHZ * 1e9 / iters, GHZ);
        }
        {
            const int iters = 2000;
            float ms = time_kernel(k_fma<16>, blocks, lds, iters, 1e-3, out);
            printf("k_fma<16>  %d waves/SIMD: %.3f ms -> %.2f SIMD-cycles per v_fma_f64 (%.1f TFLOP/s)\n", w, ms,
                   ms * 1e-3 * GHZ * 1e9 / (iters * 8.0 * 16 * w), 2.0 * 64 * 8 * 16 * iters * 4.0 * blocks / (ms * 1e-3) / 1e12);
            ms = time_kernel(k_mix<16>, blocks, lds, iters, 1e-3, out);
            printf("k_mix<16>  %d waves/SIMD: %.3f ms -> %.2f SIMD-cycles per fp64 instruction (add/mul/fma/fma)\n", w, ms,
                   ms * 1e-3 * GHZ * 1e9 / (iters * 2.0 * 4 * 16 * w));
        }
        if (w <= 2 || lds >= 4096 * 16) {
            const int iters = 100;
            const int l = lds < 4096 * 16 ? 4096 * 16 : lds;
            float ms = time_kernel(k_lds, blocks, l, iters, 4096, (const cplx*)tw, out);
            printf("k_lds      %d waves/SIMD: %.3f ms -> %.0f SIMD-cycles per stage (read16 + butterfly + write16), per wave %.0f\n", w, ms,
                   ms * 1e-3 * GHZ * 1e9 / (iters * 2.0 * w), ms * 1e-3 * GHZ * 1e9 / (iters * 2.0));
        }
    }
    return 0;
}
