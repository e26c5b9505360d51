// Hardware probe for gfx950 (MI355X): fp64 MFMA rate/latency/layout, fp64 VALU FMA rate, HBM copy rate.
// Dev tool only (not part of the product library). Build: hipcc --offload-arch=gfx950 -O3 probe_gfx950.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_rate(double* out, int iters) {
    d4 acc[NACC];
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) fma_rate(double* out, int iters) {
    double acc[NACC];
    double a = 1.0 + threadIdx.x * 1e-9, b = threadIdx.x * 2e-9;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// layout check: D = A(16x4) * B(4x16) with A[i][k] = i*10+k, B[k][j] = (k+1)*100 + j*3 (asymmetric)
__global__ void mfma_layout(double* d_out) {
    int l = threadIdx.x;
    int i = l & 15, k = l >> 4;
    double a = i * 10.0 + k;             // A[i][k]
    double b = (k + 1) * 100.0 + i * 3;  // B[k][j=i]
    d4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d_out[l * 4 + r] = acc[r];
}

__global__ void copy_kernel(const double2* __restrict__ in, double2* __restrict__ out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <typename F>
float time_ms(F f, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s arch=%s CUs=%d clock=%d kHz memclock=%d kHz mem=%.1f GB L2=%d sharedPerBlock=%zu maxSharedPerMP=%zu warp=%d\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.totalGlobalMem / 1e9,
           p.l2CacheSize, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.warpSize);
    double* d_out; CK(hipMalloc(&d_out, 1 << 24));
    // layout
    {
        mfma_layout<<<1, 64>>>(d_out);
        std::vector<double> h(256);
        CK(hipMemcpy(h.data(), d_out, 256 * 8, hipMemcpyDeviceToHost));
        // reference
        int bad_a = 0, bad_b = 0;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int col = l & 15;
            int rowA = (l >> 4) + 4 * r;   // guide's f64 layout
            int rowB = (l >> 4) * 4 + r;   // f32-style layout
            auto ref = [&](int i, int j) { double s = 0; for (int k = 0; k < 4; ++k) s += (i * 10.0 + k) * ((k + 1) * 100.0 + j * 3); return s; };
            if (h[l * 4 + r] != ref(rowA, col)) bad_a++;
            if (h[l * 4 + r] != ref(rowB, col)) bad_b++;
        }
        printf("mfma_f64_16x16x4 layout: row=(lane>>4)+4*reg mismatches=%d ; row=(lane>>4)*4+reg mismatches=%d\n", bad_a, bad_b);
    }
    int blocks = p.multiProcessorCount * 8;
    // MFMA rate
    {
        int iters = 20000;
        auto run = [&](auto kern, int nacc, int nblk, int nthr, const char* name) {
            float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(nblk), dim3(nthr), 0, 0, d_out, iters); });
            double flops = (double)nblk * (nthr / 64) * (double)iters * nacc * 2.0 * 16 * 16 * 4;
            printf("%s nacc=%d blocks=%d thr=%d: %.3f ms  %.2f TFLOP/s\n", name, nacc, nblk, nthr, ms, flops / ms / 1e9);
        };
        int cus = p.multiProcessorCount;
        run(mfma_rate<1>, 1, cus, 256, "mfma_f64 1wave/SIMD");
        run(mfma_rate<2>, 2, cus, 256, "mfma_f64 1wave/SIMD");
        run(mfma_rate<4>, 4, cus, 256, "mfma_f64 1wave/SIMD");
        run(mfma_rate<8>, 8, cus, 256, "mfma_f64 1wave/SIMD");
        run(mfma_rate<1>, 1, cus * 2, 256, "mfma_f64 2wave/SIMD");
        run(mfma_rate<4>, 4, cus * 2, 256, "mfma_f64 2wave/SIMD");
        run(mfma_rate<4>, 4, cus * 4, 256, "mfma_f64 4wave/SIMD");
        // single wave on the whole chip → per-instruction latency / issue
        {
            float ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate<1>, dim3(1), dim3(64), 0, 0, d_out, iters); });
            printf("mfma_f64 dependent chain: %.1f ns per MFMA (one wave, 1 acc)\n", ms * 1e6 / iters);
            ms = time_ms([&] { hipLaunchKernelGGL(mfma_rate<8>, dim3(1), dim3(64), 0, 0, d_out, iters); });
            printf("mfma_f64 independent x8: %.1f ns per MFMA (one wave, 8 acc)\n", ms * 1e6 / iters / 8);
        }
    }
    {
        int iters = 20000;
        auto run = [&](auto kern, int nacc, int nblk, int nthr, const char* name) {
            float ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(nblk), dim3(nthr), 0, 0, d_out, iters); });
            double flops = (double)nblk * nthr * (double)iters * nacc * 2.0;
            printf("%s nacc=%d blocks=%d thr=%d: %.3f ms  %.2f TFLOP/s\n", name, nacc, nblk, nthr, ms, flops / ms / 1e9);
        };
        int cus = p.multiProcessorCount;
        run(fma_rate<8>, 8, cus * 4, 256, "v_fma_f64 4wave/SIMD");
        run(fma_rate<16>, 16, cus * 8, 256, "v_fma_f64 8wave/SIMD");
    }
    // HBM copy
    {
        size_t bytes = (size_t)4 << 30;
        double2 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
        size_t n = bytes / sizeof(double2);
        for (int mult : {4, 8, 16, 32}) {
            float ms = time_ms([&] { copy_kernel<<<p.multiProcessorCount * mult, 256>>>(a, b, n); });
            printf("copy 4GiB grid=%dxCU: %.3f ms  %.2f TB/s (read+write)\n", mult, ms, 2.0 * bytes / ms / 1e9);
        }
        float ms = time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); });
        printf("hipMemcpy D2D 4GiB: %.3f ms %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
        // host <-> device
        void* h; CK(hipHostMalloc(&h, (size_t)1 << 30));
        ms = time_ms([&] { CK(hipMemcpyAsync(a, h, (size_t)1 << 30, hipMemcpyHostToDevice, 0)); }, 3);
        printf("H2D pinned 1GiB: %.3f ms %.2f GB/s\n", ms, ((size_t)1 << 30) / ms / 1e6);
        ms = time_ms([&] { CK(hipMemcpyAsync(h, a, (size_t)1 << 30, hipMemcpyDeviceToHost, 0)); }, 3);
        printf("D2H pinned 1GiB: %.3f ms %.2f GB/s\n", ms, ((size_t)1 << 30) / ms / 1e6);
    }
    return 0;
}
