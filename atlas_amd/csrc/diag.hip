// Measurement aid, not part of the transform path: the rate at which this device sustains v_mfma_f64_16x16x4_f64 when a
// kernel does nothing else -- four wavefronts per SIMD, six independent accumulator tiles each, operands with random
// mantissas, kernels of several milliseconds.  bench.py reports it next to the 78.6 TFLOP/s datasheet peak: the shader
// clock settles near 2.05 GHz under sustained fp64 matrix work (GRBM_GUI_ACTIVE, profiles/r02_mfma_sustained.txt), so
// no kernel on this part reaches the datasheet number for longer than a clock ramp.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>

#include "../../include/atlas_amd.h"

namespace atlas_amd {
void set_last_error(const std::string& s);
}
#define DG_CHECK(call)                                                                                        \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess) {                                                                               \
            (void)hipGetLastError(); /* reported here, once: not left sticky for the next call */ \
            throw std::runtime_error(std::string("HIP error '") + hipGetErrorString(e_) + "' in " #call);     \
        }                                                                                                     \
    } while (0)

namespace atlas_amd {
namespace diag {

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline double random_operand(uint64_t& s) {
    s          = s * 6364136223846793005ull + 1442695040888963407ull;
    uint64_t m = (s >> 12) | 0x3ff0000000000000ull;  // [1, 2) with a random mantissa
    double v;
    __builtin_memcpy(&v, &m, 8);
    return (s >> 63) ? 1.5 - v : v - 1.5;
}

__global__ void __launch_bounds__(256) mfma_f64_rate_kernel(double* out, int iters) {
    uint64_t s = 88172645463325252ull + (blockIdx.x * 256 + threadIdx.x) * 2654435761ull;
    double a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = random_operand(s);
        b[i] = random_operand(s);
    }
    d4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        acc[i] = d4{0, 0, 0, 0};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[(k + i) & 3], acc[i], 0, 0, 0);
            }
        }
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

// the same for v_mfma_f32_16x16x4_f32 (the Legendre stage of the fp32 variant) [r4]
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) mfma_f32_rate_kernel(double* out, int iters) {
    uint64_t s = 88172645463325252ull + (blockIdx.x * 256 + threadIdx.x) * 2654435761ull;
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = (float)random_operand(s);
        b[i] = (float)random_operand(s);
    }
    f4 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        acc[i] = f4{0, 0, 0, 0};
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[(k + i) & 3], acc[i], 0, 0, 0);
            }
        }
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        t += (double)acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

}  // namespace diag
}  // namespace atlas_amd

// cycles_per_mfma: 64 (fp64 16x16x4) / 32 (fp32 16x16x4) at the datasheet rate -- sizes the kernel for target_ms
static int diag_mfma_rate(void (*kernel)(double*, int), double cycles_per_mfma, const char* what, double target_ms, int repeats,
                          double* tflops_out) {
    try {
        using namespace atlas_amd::diag;
        if (!tflops_out || repeats < 1 || !(target_ms > 0)) {
            throw std::invalid_argument(std::string(what) + ": target_ms > 0, repeats >= 1, tflops_out != NULL");
        }
        hipDeviceProp_t prop;
        int dev = 0;
        DG_CHECK(hipGetDevice(&dev));
        DG_CHECK(hipGetDeviceProperties(&prop, dev));
        const int nblk = prop.multiProcessorCount * 4;  // 4 workgroups of 4 wavefronts per CU: 4 wavefronts per SIMD
        double* d_out  = nullptr;
        DG_CHECK(hipMalloc(&d_out, size_t(nblk) * 256 * sizeof(double)));
        hipEvent_t e0, e1;
        DG_CHECK(hipEventCreate(&e0));
        DG_CHECK(hipEventCreate(&e1));
        // 24 MFMAs per iteration and wavefront, 4 wavefronts per SIMD; sized for the datasheet clock
        const int iters = std::max(1, int(target_ms * 1e-3 * 2.4e9 / (24.0 * cycles_per_mfma * 4.0)));
        double best     = 0;
        for (int r = 0; r < repeats + 1; ++r) {  // the first launch is the warm-up
            DG_CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kernel, dim3(nblk), dim3(256), 0, 0, d_out, iters);
            DG_CHECK(hipGetLastError());
            DG_CHECK(hipEventRecord(e1, 0));
            DG_CHECK(hipEventSynchronize(e1));
            float ms = 0;
            DG_CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double tf = double(nblk) * 4 * double(iters) * 24 * 2048.0 / (ms * 1e-3) / 1e12;
            if (r > 0) {
                best = std::max(best, tf);
            }
        }
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
        (void)hipFree(d_out);
        *tflops_out = best;
    }
    catch (const std::exception& e) {
        atlas_amd::set_last_error(e.what());
        return 1;
    }
    return 0;
}

extern "C" int atlas_amd__diag_mfma_f64_rate(double target_ms, int repeats, double* tflops_out) {
    return diag_mfma_rate(atlas_amd::diag::mfma_f64_rate_kernel, 64.0, "diag_mfma_f64_rate", target_ms, repeats, tflops_out);
}
namespace atlas_amd {
namespace trans {
hipError_t launch_gp_to_field(const double* gp, double* field, long long npts, int nf, hipStream_t stream);
}
}  // namespace atlas_amd
// average milliseconds of the [nf][npts] -> [npts][nf] transposition of the distributed transform's halo path (csrc/vd2uv_kernel.hip)
extern "C" int atlas_amd__diag_gp_to_field(const double* gp_dev, double* field_dev, long long npts, int nb_fields, int repeats,
                                            double* ms_out) {
    try {
        if (!gp_dev || !field_dev || !ms_out || repeats < 1) {
            throw std::invalid_argument("diag_gp_to_field: null argument");
        }
        hipEvent_t e0, e1;
        DG_CHECK(hipEventCreate(&e0));
        DG_CHECK(hipEventCreate(&e1));
        DG_CHECK(atlas_amd::trans::launch_gp_to_field(gp_dev, field_dev, npts, nb_fields, nullptr));
        DG_CHECK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < repeats; ++i) {
            DG_CHECK(atlas_amd::trans::launch_gp_to_field(gp_dev, field_dev, npts, nb_fields, nullptr));
        }
        DG_CHECK(hipEventRecord(e1, nullptr));
        DG_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        DG_CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = (double)ms / repeats;
        DG_CHECK(hipEventDestroy(e0));
        DG_CHECK(hipEventDestroy(e1));
        return 0;
    }
    catch (const std::exception& e) {
        atlas_amd::set_last_error(e.what());
        return 1;
    }
}
extern "C" int atlas_amd__diag_mfma_f32_rate(double target_ms, int repeats, double* tflops_out) {
    return diag_mfma_rate(atlas_amd::diag::mfma_f32_rate_kernel, 32.0, "diag_mfma_f32_rate", target_ms, repeats, tflops_out);
}
