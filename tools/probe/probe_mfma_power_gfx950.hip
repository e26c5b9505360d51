// Dev tool (GPU box): sustained fp64 MFMA rate with operands that toggle bits -- constant small operands (what
// probe_gfx950 uses) against per-lane random operands of mixed sign and 52 random mantissa bits, kernels of about 10 and
// 50 ms, 4 wavefronts per SIMD.  Says whether the 78.6 TF/s peak survives realistic operand data (power / clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline double rnd(uint64_t& s) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t m = (s >> 12) | 0x3ff0000000000000ull;  // [1,2) with random mantissa
    double v;
    __builtin_memcpy(&v, &m, 8);
    return (s >> 63) ? -(v - 1.5) : (v - 1.5);       // (-0.5, 0.5), mixed sign
}

template <bool RANDOM>
__global__ void __launch_bounds__(256) mfma_rate(double* out, int iters) {
    uint64_t s = 88172645463325252ull + (blockIdx.x * 256 + threadIdx.x) * 2654435761ull;
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = RANDOM ? rnd(s) : threadIdx.x * 1e-3;
        b[i] = RANDOM ? rnd(s) : threadIdx.x * 2e-3;
    }
    d4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 6; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[(k + i) & 3], acc[i], 0, 0, 0);
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    double* d_out;
    CK(hipMalloc(&d_out, 1 << 24));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int cus = p.multiProcessorCount;
    for (int pass = 0; pass < 2; ++pass)
        for (int random = 0; random < 2; ++random)
            for (int iters : {1700, 8500, 34000}) {
                float best = 1e30f, worst = 0;
                for (int r = 0; r < 4; ++r) {
                    CK(hipEventRecord(e0));
                    if (random) hipLaunchKernelGGL(mfma_rate<true>, dim3(cus * 4), dim3(256), 0, 0, d_out, iters);
                    else hipLaunchKernelGGL(mfma_rate<false>, dim3(cus * 4), dim3(256), 0, 0, d_out, iters);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                    worst = ms > worst ? ms : worst;
                }
                const double flops = (double)cus * 4 * 4 * (double)iters * 24 * 2048.0;
                printf("%s operands, %6d iterations: %.2f .. %.2f ms  %.2f .. %.2f TFLOP/s\n", random ? "random  " : "constant", iters, best,
                       worst, flops / worst / 1e9, flops / best / 1e9);
            }
    return 0;
}
