// Dev probe (GPU box): issue cadence of the fp64 MFMA instructions on gfx950, by register placement and shape.
// "v_mfma_f64_16x16x4 issues every ~74 cycles, not 64" -- a property of the instruction, or of where
// its operands live?  Every wavefront runs ITERS x 8 MFMAs on 8 independent accumulators (no dependent pair closer than
// 8 instructions) and times the loop with s_memtime (shader clock); W wavefronts per SIMD (occupancy pinned by LDS).
//   16x16x4  acc VGPR, A/B VGPR      (what the compiler emits for __builtin_amdgcn_mfma_f64_16x16x4f64)
//   16x16x4  acc AGPR, A/B VGPR
//   16x16x4  acc AGPR, A/B AGPR
//   4x4x4_4b acc VGPR / AGPR         (4 blocks of 4x4x4: 512 flops per instruction)
// cycles per MFMA and SIMD = loop cycles / (ITERS * 8) / W' where W' = wavefronts that share the SIMD; flops per cycle and
// SIMD follow (2048 resp. 512 flops per instruction); the wall-clock rate is printed beside it (it includes the clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d1;

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int MODE>
__global__ void __launch_bounds__(256) k(int iters, double seed, unsigned long long* cyc, double* out) {
    extern __shared__ double lds[];
    const double a = seed * (threadIdx.x + 1), b = seed * (threadIdx.x + 3);
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define M(i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c##i) : "v"(a), "v"(b));
            REP8(M)
#undef M
        }
        else if (MODE == 1) {
#define M(i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c##i) : "v"(a), "v"(b));
            REP8(M)
#undef M
        }
        else if (MODE == 2) {
#define M(i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c##i) : "a"(a), "a"(b));
            REP8(M)
#undef M
        }
        else if (MODE == 3) {
#define M(i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(e##i) : "v"(a), "v"(b));
            REP8(M)
#undef M
        }
        else {
#define M(i) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+a"(e##i) : "v"(a), "v"(b));
            REP8(M)
#undef M
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = c0[0] + c1[1] + c2[2] + c3[3] + c4[0] + c5[1] + c6[2] + c7[3] + e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7;
    if ((threadIdx.x & 63) == 0) {
        cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
    }
    if (s == 1.2345) {
        out[0] = s + lds[0];
    }
}

template <int MODE>
static void run(const char* name, double flops_per_inst) {
    const int iters = 4000;
    double* out;
    unsigned long long* cyc;
    CK(hipMalloc(&out, 64));
    for (int w : {1, 2, 4}) {
        const int lds    = w == 1 ? 160 * 1024 : (w == 2 ? 80 * 1024 : 40 * 1024);
        const int blocks = 256 * w;
        CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4));
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, iters, 1e-3, cyc, out);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, iters, 1e-3, cyc, out);
        CK(hipEventRecord(b));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h(blocks * 4);
        CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        double sum = 0;
        for (auto v : h) sum += (double)v;
        const double per_wave = sum / h.size() / (iters * 8.0);   // cycles between MFMAs as one wavefront sees them
        const double per_simd = per_wave / w;                      // w wavefronts share the SIMD
        const double total    = flops_per_inst * iters * 8.0 * blocks * 4;
        printf("%-34s %d waves/SIMD: %7.1f cycles per MFMA and wave, %6.1f per MFMA and SIMD = %5.2f flop/cycle/SIMD; wall %.3f ms = %.1f TF/s (clock %.2f GHz)\n",
               name, w, per_wave, per_simd, flops_per_inst / per_simd, ms, total / (ms * 1e-3) / 1e12,
               sum / h.size() / (ms * 1e-3) / 1e9);
        CK(hipFree(cyc));
    }
    CK(hipFree(out));
}

int main() {
    run<0>("16x16x4  acc VGPR, A/B VGPR", 2048);
    run<1>("16x16x4  acc AGPR, A/B VGPR", 2048);
    run<2>("16x16x4  acc AGPR, A/B AGPR", 2048);
    run<3>("4x4x4_4b acc VGPR", 512);
    run<4>("4x4x4_4b acc AGPR", 512);
    return 0;
}
