// Dev probe (GPU box): how many workgroups of a given LDS size / thread count are co-resident per CU.
// Each workgroup spins for a fixed number of clocks; with G workgroups per CU queued, time = ceil(G / resident) * spin.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__global__ void spin(long long clocks, double* out) {
    extern __shared__ double lds[];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < clocks) {
        __builtin_amdgcn_s_sleep(8);
    }
    if (lds[threadIdx.x] == -1.0) out[0] = 1;
}
int main() {
    double* out;
    CK(hipMalloc(&out, 64));
    const int sizes[] = {16384, 32768, 36864, 40960, 53248, 54016, 61440, 65536, 73728, 80896, 81920, 98304};
    for (int lds : sizes) {
        for (int nthr : {192, 256, 320, 512}) {
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            int occ = -1;
            CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, spin, nthr, lds));
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            const long long ticks = 2000;  // wall_clock64: 100 MHz -> 20 us
            hipLaunchKernelGGL(spin, dim3(256), dim3(nthr), lds, 0, ticks, out);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(spin, dim3(256 * 12), dim3(nthr), lds, 0, ticks, out);
            CK(hipEventRecord(b));
            CK(hipDeviceSynchronize());
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            printf("lds %6d threads %3d: api says %d/CU; 12 workgroups per CU of 20 us took %.1f us -> ~%.2f resident per CU\n",
                   lds, nthr, occ, ms * 1e3, 12 * 20.0 / (ms * 1e3));
        }
    }
    return 0;
}
