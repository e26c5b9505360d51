// Probe 3: MFMA f64 issue efficiency in the shape of the Legendre kernel's inner loop.
// 18 accumulators per wave, 36 MFMAs per "stage", 4 or 8 waves per workgroup, many short workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: constant operands; 1: operands from LDS (ds_read per MFMA pair, no barrier); 2: + barrier per stage
__global__ void __launch_bounds__(256, 2) k(double* out, const double* in, int nstage) {
    __shared__ double lds[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 256) lds[i] = in[i & 511];
    __syncthreads();
    d4 acc[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) acc[i] = d4{0, 0, 0, 0};
    double a0 = in[tid], b0 = in[tid + 256];
    for (int s = 0; s < nstage; ++s) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double a = MODE == 0 ? a0 : lds[(tid + 64 * h + s) & 4095];
#pragma unroll
            for (int j = 0; j < 18; ++j) {
                double b = MODE == 0 ? b0 : lds[(tid * 2 + 37 * j + h * 700 + s) & 4095];
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            }
        }
        if (MODE == 2) __syncthreads();
    }
    double r = 0;
#pragma unroll
    for (int i = 0; i < 18; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(blockIdx.x & 65535) * 256 + tid] = r;
}

template <typename F> float time_ms(F f) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) { CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms; }
    return best;
}

int main() {
    double *d_out, *d_in; CK(hipMalloc(&d_out, (size_t)1 << 28)); CK(hipMalloc(&d_in, 512 * 8));
    double h[512]; for (int i = 0; i < 512; ++i) h[i] = (rand() / (double)RAND_MAX - 0.5);
    CK(hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice));
    auto run = [&](auto kern, const char* name, int nwg, int nstage) {
        float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), 0, 0, d_out, d_in, nstage); });
        double flops = (double)nwg * 4 * nstage * 36 * 2048.0;
        printf("%-22s wgs=%6d stages=%4d: %8.3f ms  %.1f TF/s\n", name, nwg, nstage, ms, flops / ms / 1e9);
    };
    for (int nst : {10, 40, 160, 640}) {
        int nwg = 76000 * 40 / nst;
        run(k<0>, "const operands", nwg, nst);
        run(k<1>, "lds operands", nwg, nst);
        run(k<2>, "lds + barrier", nwg, nst);
    }
    run(k<0>, "const, 512 wgs", 512, 4000);
    run(k<1>, "lds, 512 wgs", 512, 4000);
    return 0;
}
