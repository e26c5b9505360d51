// Spectral vorticity/divergence -> U,V (= u,v * cos(lat)) and assembly of the combined spectral array that
// TransLocal feeds to invtrans_uv on the vor/div path.
//
// Reference being replaced (all three fused into one elementwise kernel):
//   extend_truncation          TransLocal.cc:1496-1519   T -> T+1, new row/column zero
//   vd2uv                      VorDivToUVLocal.cc:62-184  Temperton (1991) eq. 2.12/2.13, scaled by 1/a
//   "merge all spectra"        TransLocal.cc:1567-1581   per (m,n,imag): [U fields][V fields][scalar fields]
//
// With eps(m,n) = sqrt((n^2-m^2)/(4n^2-1)), lap(n) = -a^2/(n(n+1)) (lap(0)=0), for m <= n <= T+1:
//   chi  = m * lap(n),  psiM = (n-1) * eps(m,n) * lap(n-1),  psiP = (n+2) * eps(m,n+1) * lap(n+1)
//   U_re = -chi*div_im(n) + psiM*vor_re(n-1) - psiP*vor_re(n+1)       U_im = +chi*div_re(n) + psiM*vor_im(n-1) - psiP*vor_im(n+1)
//   V_re = -chi*vor_im(n) - psiM*div_re(n-1) + psiP*div_re(n+1)       V_im = +chi*vor_re(n) - psiM*div_im(n-1) + psiP*div_im(n+1)
// (m = 0: real parts only, no chi terms; imaginary parts are 0), vor/div taken as 0 outside m <= n <= T.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_structs.h"

namespace atlas_amd {
namespace trans {

constexpr double kEarthRadius = 6371229.;  // util::Earth::radius(), src/atlas/util/Earth.h:23

// Real: storage type of the spectra (double; float for the fp32 variant -- the arithmetic is double either way: the factors
// a^2 / (n (n + 1)) and 1 / a span thirteen decades)
template <class Real>
struct PrepareParamsT {
    const Real* vor;  // [(T+1)(T+2)] x nvd, truncation T layout
    const Real* div;
    const Real* sp;   // scalars, truncation T layout, ns fields (may be null if ns == 0)
    Real* out;        // [(T+2)(T+3)] x (2*nvd + ns), truncation T+1 layout
    int T;
    int nvd;
    int ns;
    int fshift;       // log2 of the threads that run over the fields (the launcher: smallest power of two >= the field count, at most 256)
    int nb;           // total wavenumbers per workgroup chunk, <= PREP_NB
};

__device__ __forceinline__ double dev_eps(int m, int n) {
    if (n < m || (m == 0 && n == 0)) {
        return 0.;
    }
    return sqrt(((double)n * n - (double)m * m) / (4. * (double)n * n - 1.));
}
__device__ __forceinline__ double dev_lap(int n) {
    if (n <= 0) {
        return 0.;
    }
    return -kEarthRadius * kEarthRadius / (n * (n + 1.));
}

// One workgroup per (m, chunk of PREP_NB total wavenumbers n); the three factors of a coefficient (m, n) -- chi, psi-, psi+: two square
// roots and five divisions in fp64 -- are the same for every field: computed once per (m, n) into LDS, then the fields (fastest index
// of input and output: coalesced) stream through.  Round 4's form computed them per ELEMENT, with an integer division by the field
// count on top: 3.76 ms for 137 + 137 + 137 fields at TL1279 (0.36 of HBM) -- see profiles/r05_configs.jsonl for the figure now.
constexpr int PREP_NB = 32;
template <class Real>
__global__ void __launch_bounds__(256) spectra_prepare_kernel(PrepareParamsT<Real> p) {
    __shared__ double s_chi[PREP_NB], s_psiM[PREP_NB], s_psiP[PREP_NB];
    const int m    = blockIdx.y;
    const int T    = p.T;
    const int TE   = T + 1;
    const int nall = 2 * p.nvd + p.ns;
    const int nn   = TE - m + 1;               // total wavenumbers n = m .. TE of this m
    const long long obase = (long long)(2 * TE + 3 - m) * m / 2 * 2 * nall;
    const long long ibase = (long long)(2 * T + 3 - m) * m / 2 * 2;  // x nf of the respective input
    const int NB = p.nb;   // total wavenumbers per chunk (<= PREP_NB; fewer for small truncations: enough workgroups to fill the device)
    for (int n0 = blockIdx.x * NB; n0 < nn; n0 += gridDim.x * NB) {
        __syncthreads();
        if ((int)threadIdx.x < NB && n0 + (int)threadIdx.x < nn) {
            const int n         = m + n0 + threadIdx.x;
            s_chi[threadIdx.x]  = m * dev_lap(n);
            s_psiM[threadIdx.x] = (n - 1) * dev_eps(m, n) * dev_lap(n - 1);
            s_psiP[threadIdx.x] = (n + 2) * dev_eps(m, n + 1) * dev_lap(n + 1);
        }
        __syncthreads();
        const int cnt = nn - n0 < NB ? nn - n0 : NB;
        // threads: the low fshift bits run over the fields, the others over the (n, imag) rows of the chunk (few fields: several rows
        // at once; more than 256 fields: one row at a time, the fields in strides of 256)
        const int fstep = 1 << p.fshift;
        for (int rr = threadIdx.x >> p.fshift; rr < 2 * cnt; rr += (int)blockDim.x >> p.fshift) {
            const int k       = rr >> 1;
            const int imag    = rr & 1;
            const int n       = m + n0 + k;
            const double chi  = s_chi[k];
            const double psiM = s_psiM[k];
            const double psiP = s_psiP[k];
            Real* out         = p.out + obase + (long long)(2 * (n - m) + imag) * nall;
            for (int fld = threadIdx.x & (fstep - 1); fld < nall; fld += fstep) {
                double v = 0.;
                if (fld >= 2 * p.nvd) {
                    // scalar field, zero-extended (TransLocal.cc:1507-1513)
                    const int f = fld - 2 * p.nvd;
                    if (n <= T && m <= T) {
                        v = p.sp[(ibase + 2 * (n - m) + imag) * p.ns + f];
                    }
                }
                else if (m <= T || n <= T) {
                    const bool isV = fld >= p.nvd;
                    const int f    = isV ? fld - p.nvd : fld;
                    auto get = [&](const Real* a, int nn_, int im) -> double {
                        if (nn_ < m || nn_ > T || m > T) {
                            return 0.;
                        }
                        return (double)a[(ibase + 2 * (nn_ - m) + im) * p.nvd + f];
                    };
                    const Real* A   = isV ? p.div : p.vor;  // the field the psi terms act on
                    const Real* B   = isV ? p.vor : p.div;  // the field the chi term acts on
                    const double sg = isV ? -1. : 1.;
                    double r;
                    if (m == 0) {
                        r = imag ? 0. : sg * (psiM * get(A, n - 1, 0) - psiP * get(A, n + 1, 0));
                    }
                    else if (imag == 0) {
                        r = -chi * get(B, n, 1) + sg * (psiM * get(A, n - 1, 0) - psiP * get(A, n + 1, 0));
                    }
                    else {
                        r = +chi * get(B, n, 0) + sg * (psiM * get(A, n - 1, 1) - psiP * get(A, n + 1, 1));
                    }
                    v = r * (1. / kEarthRadius);
                }
                out[fld] = (Real)v;
            }
        }
    }
}

// ---- stand-alone VorDivToUV::execute (VorDivToUV.h:36-133, VorDivToUVLocal.cc:62-189): U, V in the layout and with the
// truncation of the inputs; wavenumbers above the truncation count as zero (the reference pads its work arrays)
struct Vd2uvParams {
    const double* vor;
    const double* div;
    double* U;
    double* V;
    int T;
    int nf;
};

__global__ void __launch_bounds__(256) vd2uv_kernel(Vd2uvParams p) {
    const int m   = blockIdx.y;
    const int T   = p.T;
    const int len = (T - m + 1) * 2 * p.nf;
    const long long base = (long long)(2 * T + 3 - m) * m / 2 * 2;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < len; e += gridDim.x * blockDim.x) {
        const int f    = e % p.nf;
        const int rest = e / p.nf;
        const int imag = rest & 1;
        const int n    = m + (rest >> 1);
        auto get = [&](const double* a, int nn, int im) -> double {
            if (nn < m || nn > T) {
                return 0.;
            }
            return a[(base + 2 * (nn - m) + im) * p.nf + f];
        };
        const double chi  = m * dev_lap(n);
        const double psiM = (n - 1) * dev_eps(m, n) * dev_lap(n - 1);
        const double psiP = (n + 2) * dev_eps(m, n + 1) * dev_lap(n + 1);
        double u, v;
        if (m == 0) {
            u = imag ? 0. : (psiM * get(p.vor, n - 1, 0) - psiP * get(p.vor, n + 1, 0));
            v = imag ? 0. : -(psiM * get(p.div, n - 1, 0) - psiP * get(p.div, n + 1, 0));
        }
        else if (imag == 0) {
            u = -chi * get(p.div, n, 1) + (psiM * get(p.vor, n - 1, 0) - psiP * get(p.vor, n + 1, 0));
            v = -chi * get(p.vor, n, 1) - (psiM * get(p.div, n - 1, 0) - psiP * get(p.div, n + 1, 0));
        }
        else {
            u = +chi * get(p.div, n, 0) + (psiM * get(p.vor, n - 1, 1) - psiP * get(p.vor, n + 1, 1));
            v = +chi * get(p.vor, n, 0) - (psiM * get(p.div, n - 1, 1) - psiP * get(p.div, n + 1, 1));
        }
        const long long o = (base + 2 * (n - m) + imag) * p.nf + f;
        p.U[o]            = u * (1. / kEarthRadius);
        p.V[o]            = v * (1. / kEarthRadius);
    }
}

hipError_t launch_vd2uv(const double* vor, const double* div, double* U, double* V, int T, int nf, hipStream_t stream) {
    Vd2uvParams p{vor, div, U, V, T, nf};
    dim3 grid(std::min(64, ((T + 1) * 2 * nf + 255) / 256), T + 1);
    hipLaunchKernelGGL(vd2uv_kernel, grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

__global__ void __launch_bounds__(256) convert_f64_f32_kernel(const double* __restrict__ src, float* __restrict__ dst,
                                                             size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        dst[i] = (float)src[i];
    }
}
hipError_t launch_convert_f64_f32(const double* src, float* dst, size_t n, hipStream_t stream) {
    if (n == 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(convert_f64_f32_kernel, dim3(4096), dim3(256), 0, stream, src, dst, n);
    return hipGetLastError();
}

template <class Real>
static hipError_t launch_spectra_prepare_t(const Real* vor, const Real* div, const Real* sp, Real* out, int T, int nvd, int ns,
                                           hipStream_t stream) {
    int fshift = 0;
    while ((1 << fshift) < 2 * nvd + ns && fshift < 8) {
        ++fshift;
    }
    const long long total_n = (long long)(T + 2) * (T + 3) / 2;
    const int nb            = (int)std::max<long long>(2, std::min<long long>(PREP_NB, total_n / 4096));
    PrepareParamsT<Real> p{vor, div, sp, out, T, nvd, ns, fshift, nb};
    dim3 grid((T + 2 + nb - 1) / nb, T + 2);   // (chunks of total wavenumbers of m = 0, zonal wavenumbers 0 .. T + 1)
    hipLaunchKernelGGL(spectra_prepare_kernel<Real>, grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}
hipError_t launch_spectra_prepare(const double* vor, const double* div, const double* sp, double* out, int T, int nvd,
                                  int ns, hipStream_t stream) {
    return launch_spectra_prepare_t<double>(vor, div, sp, out, T, nvd, ns, stream);
}
hipError_t launch_spectra_prepare_f32(const float* vor, const float* div, const float* sp, float* out, int T, int nvd, int ns,
                                      hipStream_t stream) {
    return launch_spectra_prepare_t<float>(vor, div, sp, out, T, nvd, ns, stream);
}

// ---- longitude-window crop (RectangularDomain): out[f][win_off[r] + i] = full[f][rowoff[r] - rowoff[0] + (win_i0[r] + i) mod n_r]
// One workgroup per (row, field); HBM-bound copy of the kept points (TransLocal.cc:1123-1131 does this on the host).
__global__ void __launch_bounds__(256) window_crop_kernel(const double* __restrict__ full, double* __restrict__ out,
                                                          const long long* __restrict__ rowoff, const int* __restrict__ win_i0,
                                                          const int* __restrict__ win_n, const long long* __restrict__ win_off,
                                                          long long npts_full, long long npts_out, int f_begin) {
    const int r        = blockIdx.x;
    const int f        = f_begin + blockIdx.y;
    const int n        = (int)(rowoff[r + 1] - rowoff[r]);
    const int i0       = win_i0[r];
    const int cnt      = win_n[r];
    const double* src  = full + (long long)f * npts_full + (rowoff[r] - rowoff[0]);
    double* dst        = out + (long long)f * npts_out + win_off[r];
    for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
        int j = i0 + i;
        j     = j >= n ? j - n : j;
        dst[i] = src[j];
    }
}

// ---- grid points of a band, field-major [nf][npts], into the owned part of a StructuredColumns field [point][nf]
// (the layout of a levels field on the function space): tiled transpose through LDS, HBM-bound
__global__ void __launch_bounds__(256) gp_to_field_kernel(const double* __restrict__ gp, double* __restrict__ field, long long npts,
                                                          int nf) {
    __shared__ double tile[32][33];
    const long long p0 = (long long)blockIdx.x * 32;
    const int f0       = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {   // rows = fields, columns = points (contiguous in gp)
        const int f       = f0 + r;
        const long long p = p0 + tx;
        tile[r][tx]       = (f < nf && p < npts) ? gp[(long long)f * npts + p] : 0.;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {   // rows = points, columns = fields (contiguous in field)
        const long long p = p0 + r;
        const int f       = f0 + tx;
        if (p < npts && f < nf) {
            field[p * nf + f] = tile[tx][r];
        }
    }
}

hipError_t launch_gp_to_field(const double* gp, double* field, long long npts, int nf, hipStream_t stream) {
    if (npts <= 0 || nf <= 0) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(gp_to_field_kernel, dim3((unsigned)((npts + 31) / 32), (unsigned)((nf + 31) / 32)), dim3(256), 0, stream, gp,
                       field, npts, nf);
    return hipGetLastError();
}

hipError_t launch_window_crop(const double* full, double* out, const long long* rowoff, const int* win_i0, const int* win_n,
                              const long long* win_off, int nrows, long long npts_full, long long npts_out, int f_begin,
                              int f_end, hipStream_t stream) {
    if (nrows <= 0 || f_end <= f_begin) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(window_crop_kernel, dim3(nrows, f_end - f_begin), dim3(256), 0, stream, full, out, rowoff, win_i0, win_n,
                       win_off, npts_full, npts_out, f_begin);
    return hipGetLastError();
}

}  // namespace trans
}  // namespace atlas_amd
