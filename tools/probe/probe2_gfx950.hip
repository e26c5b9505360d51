// Probe 2: fp64 MFMA pipe details on gfx950: cycles per MFMA (s_memtime), effective clock under load,
// 4x4x4_4b variant, MFMA || VALU co-issue from different waves, data dependence of the clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned long long memtime() { return __builtin_amdgcn_s_memtime(); }

template <int NACC, int MODE>  // MODE 0: 16x16x4 ; 1: 4x4x4_4b
__global__ void __launch_bounds__(256) k_mfma(double* out, const double* in, int iters, unsigned long long* cyc) {
    d4 acc[NACC];
    double a = in[threadIdx.x], b = in[threadIdx.x + 256];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    unsigned long long t0 = memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
            else {
                double r = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i][0], 0, 0, 0);
                acc[i][0] = r;
            }
        }
    }
    unsigned long long t1 = memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

// waves with (wave id & 1)==0 do MFMA, others VALU FMA
template <int MFMA_WAVES_OF_4>
__global__ void __launch_bounds__(512) k_mixed(double* out, const double* in, int iters) {
    int wave = threadIdx.x >> 6;
    double a = in[threadIdx.x & 255], b = in[(threadIdx.x & 255) + 256];
    double s = 0;
    if ((wave & 3) < MFMA_WAVES_OF_4) {
        d4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = d4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = i;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
        for (int i = 0; i < 16; ++i) s += acc[i];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f(); CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    double *d_out, *d_in; unsigned long long* d_cyc;
    CK(hipMalloc(&d_out, 1 << 24)); CK(hipMalloc(&d_in, 512 * 8)); CK(hipMalloc(&d_cyc, 64));
    int iters = 20000;
    for (int fill = 0; fill < 3; ++fill) {
        std::vector<double> h(512);
        for (int i = 0; i < 512; ++i) h[i] = fill == 0 ? 0.0 : (fill == 1 ? 1.0 : (rand() / (double)RAND_MAX - 0.5));
        CK(hipMemcpy(d_in, h.data(), 512 * 8, hipMemcpyHostToDevice));
        const char* fn = fill == 0 ? "zeros" : (fill == 1 ? "ones" : "random");
        for (int wps : {1, 2, 4}) {
            int nblk = 256 * wps;
            unsigned long long cyc;
            float ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<4, 0>), dim3(nblk), dim3(256), 0, 0, d_out, d_in, iters, d_cyc); });
            CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
            double flops = (double)nblk * 4 * (double)iters * 4 * 2048.0;
            printf("[%s] 16x16x4 %d waves/SIMD nacc=4: %.3f ms %.2f TF ; memtime ticks/MFMA(wave0)=%.1f ; ticks total=%llu => tick rate %.1f MHz\n",
                   fn, wps, ms, flops / ms / 1e9, (double)cyc / (iters * 4.0), cyc, cyc / (ms * 1e3));
        }
    }
    {
        float ms = time_ms([&] { hipLaunchKernelGGL((k_mfma<4, 1>), dim3(512), dim3(256), 0, 0, d_out, d_in, iters, d_cyc); });
        unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
        double flops = 512.0 * 4 * (double)iters * 4 * (2.0 * 4 * 4 * 4 * 4);
        printf("4x4x4_4b 2 waves/SIMD nacc=4: %.3f ms %.2f TF ticks/MFMA=%.1f\n", ms, flops / ms / 1e9, (double)cyc / (iters * 4.0));
    }
    // mixed: 8 waves per block, 1 block per CU → 2 waves per SIMD. (wave&3)<k do MFMA
    {
        auto report = [&](int k, float ms) {
            double mf = 256.0 * 8 * (k / 4.0) * (double)iters * 4 * 2048.0;
            double vf = 256.0 * 8 * ((4 - k) / 4.0) * 64 * (double)iters * 64 * 2.0;
            printf("mixed 8 waves/CU, %d of 4 waves MFMA: %.3f ms  mfma %.2f TF + valu %.2f TF = %.2f TF\n", k, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9);
        };
        report(4, time_ms([&] { hipLaunchKernelGGL((k_mixed<4>), dim3(256), dim3(512), 0, 0, d_out, d_in, iters); }));
        report(2, time_ms([&] { hipLaunchKernelGGL((k_mixed<2>), dim3(256), dim3(512), 0, 0, d_out, d_in, iters); }));
        report(0, time_ms([&] { hipLaunchKernelGGL((k_mixed<0>), dim3(256), dim3(512), 0, 0, d_out, d_in, iters); }));
        report(2, time_ms([&] { hipLaunchKernelGGL((k_mixed<2>), dim3(512), dim3(512), 0, 0, d_out, d_in, iters); }));
    }
    return 0;
}
