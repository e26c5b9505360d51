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
#include <cstdlib>
#include <string>

#include "env.h"
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

// One coefficient of U (sg = +1: A = vor, B = div) or V (sg = -1: A = div, B = vor): b_other = B(n) of the OTHER complex part,
// a_minus / a_plus = A(n - 1) / A(n + 1) of the same part.  Written with explicit roundings so that both kernels below (and any
// field count, which selects between them) give the same bits.
__device__ __forceinline__ double wind_coefficient(bool m_is_zero, int imag, double sg, double chi, double psiM, double psiP,
                                                   double b_other, double a_minus, double a_plus) {
    const double t = __dmul_rn(sg, __fma_rn(psiM, a_minus, -__dmul_rn(psiP, a_plus)));
    double r;
    if (m_is_zero) {
        r = imag ? 0. : t;
    }
    else if (imag == 0) {
        r = __fma_rn(-chi, b_other, t);
    }
    else {
        r = __fma_rn(chi, b_other, t);
    }
    return __dmul_rn(r, 1. / kEarthRadius);
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
                    v = wind_coefficient(m == 0, imag, sg, chi, psiM, psiP, m == 0 ? 0. : get(B, n, 1 - imag), get(A, n - 1, imag),
                                         get(A, n + 1, imag));
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

// Streaming form for many vor/div fields [r5]: a lane owns a FIELD and walks the wavenumbers n of its chunk upwards with vor / div of
// n - 1, n, n + 1 (both complex parts) in registers and those of n + 2 on their way -- every input value is requested once per chunk
// (the form above requests it three times: as A(n - 1), B(n) and A(n + 1) of three output rows), and the four outputs of a step are
// four coalesced stores.  One workgroup per (m, chunk of PREP_SNB total wavenumbers); the groups of 64 fields of a row go to
// different wavefronts (see below); the scalar fields are copied row by row.  Same arithmetic (wind_coefficient): same bits.
// TL1279, 137 + 137 + 137 fields, ms (fp64 | fp32; profiles/r05_prepare_ab.txt): row form 3.2 - 3.45 | 3.6; a wavefront taking the
// groups one after the other, chunks of 64: 3.1 - 3.3 | 2.15; groups side by side: 2.8 | 1.6; chunks of 128 / 32 / 16 / 8:
// 2.85 / 2.55 / 2.37 / 2.25 | 1.7 / 1.55 / 1.55 / 1.77.
constexpr int PREP_SNB = 16;
template <class Real>
__global__ void __launch_bounds__(256) spectra_prepare_stream_kernel(PrepareParamsT<Real> p) {
    __shared__ double s_chi[PREP_SNB], s_psiM[PREP_SNB], s_psiP[PREP_SNB];
    __shared__ int s_next_row;
    if (threadIdx.x == 0) {
        s_next_row = 0;
    }
    const int m    = blockIdx.y;
    const int T    = p.T;
    const int TE   = T + 1;
    const int nvd  = p.nvd;
    const int nall = 2 * nvd + p.ns;
    const int nn   = TE - m + 1;               // total wavenumbers n = m .. TE of this m
    const int n0   = blockIdx.x * PREP_SNB;    // first one of this chunk, as n - m
    if (n0 >= nn) {
        return;
    }
    const int cnt = nn - n0 < PREP_SNB ? nn - n0 : PREP_SNB;
    if ((int)threadIdx.x < cnt) {
        const int n         = m + n0 + threadIdx.x;
        s_chi[threadIdx.x]  = m * dev_lap(n);
        s_psiM[threadIdx.x] = (n - 1) * dev_eps(m, n) * dev_lap(n - 1);
        s_psiP[threadIdx.x] = (n + 2) * dev_eps(m, n + 1) * dev_lap(n + 1);
    }
    __syncthreads();
    const long long obase = (long long)(2 * TE + 3 - m) * m / 2 * 2 * nall;
    const long long ibase = (long long)(2 * T + 3 - m) * m / 2 * 2;  // x nf of the respective input
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool m_in = m <= T;
    // tasks (group of 64 fields, sub-chunk of the wavenumbers): the groups of a row go to DIFFERENT wavefronts, which walk the chunk side
    // by side -- the 128-byte lines at the seams of the groups are then touched by all their users within a few steps (one wavefront
    // taking the groups one after the other re-fetches them, and writes the seam lines of the output twice, partially)
    const int G    = (nvd + 63) >> 6;
    const int nsub = G >= 4 ? 1 : 4 / G;       // 1 group: 4 sub-chunks, 2: 2, 3 and more: the whole chunk
    const int sl   = (cnt + nsub - 1) / nsub;
    for (int task = wave; task < G * nsub; task += 4) {
        const int f  = (task % G) * 64 + lane;
        const int ka = (task / G) * sl;
        const int kb = ka + sl < cnt ? ka + sl : cnt;
        if (f >= nvd || ka >= kb) {
            continue;
        }
        const Real* __restrict__ vor = p.vor + f;
        const Real* __restrict__ div = p.div + f;
        // value (n, im) of a vor/div field; 0 outside m <= n <= T (the reference pads its work arrays)
        auto ld = [&](const Real* a, int n, int im) -> double {
            return (m_in && n >= m && n <= T) ? (double)a[(ibase + 2 * (n - m) + im) * nvd] : 0.;
        };
        int n = m + n0 + ka;
        double vm[2] = {ld(vor, n - 1, 0), ld(vor, n - 1, 1)}, dm[2] = {ld(div, n - 1, 0), ld(div, n - 1, 1)};
        double v0[2] = {ld(vor, n, 0), ld(vor, n, 1)}, d0[2] = {ld(div, n, 0), ld(div, n, 1)};
        double vp[2] = {ld(vor, n + 1, 0), ld(vor, n + 1, 1)}, dp[2] = {ld(div, n + 1, 0), ld(div, n + 1, 1)};
        Real* __restrict__ o = p.out + obase + (long long)(2 * (n - m)) * nall + f;
        for (int k = ka; k < kb; ++k, ++n, o += 2 * nall) {
            // the values of n + 2 are requested a step before they are used (the step's own arithmetic waits for nothing new)
            const int nq = k + 1 < kb ? n + 2 : -1;   // (the last step of the sub-chunk has nothing to request)
            const double vq[2] = {ld(vor, nq, 0), ld(vor, nq, 1)}, dq[2] = {ld(div, nq, 0), ld(div, nq, 1)};
            const double chi = s_chi[k], psiM = s_psiM[k], psiP = s_psiP[k];
#pragma unroll
            for (int imag = 0; imag < 2; ++imag) {
                double u = 0., v = 0.;
                if (m_in || n <= T) {
                    u = wind_coefficient(m == 0, imag, 1., chi, psiM, psiP, m == 0 ? 0. : d0[1 - imag], vm[imag], vp[imag]);
                    v = wind_coefficient(m == 0, imag, -1., chi, psiM, psiP, m == 0 ? 0. : v0[1 - imag], dm[imag], dp[imag]);
                }
                o[imag * nall]       = (Real)u;
                o[imag * nall + nvd] = (Real)v;
            }
#pragma unroll
            for (int im = 0; im < 2; ++im) {
                vm[im] = v0[im];
                dm[im] = d0[im];
                v0[im] = vp[im];
                d0[im] = dp[im];
                vp[im] = vq[im];
                dp[im] = dq[im];
            }
        }
    }
    // scalar fields, zero-extended (TransLocal.cc:1507-1513): fields over the lanes, rows (n, imag) handed out to whichever wavefront
    // is free (with three groups of wind fields the fourth wavefront starts here at once)
    if (p.ns > 0) {
        for (;;) {
            int rr = 0;
            if (lane == 0) {
                rr = atomicAdd(&s_next_row, 1);
            }
            rr = __builtin_amdgcn_readfirstlane(rr);
            if (rr >= 2 * cnt) {
                break;
            }
            const int n    = m + n0 + (rr >> 1);
            const int imag = rr & 1;
            Real* out      = p.out + obase + (long long)(2 * (n - m) + imag) * nall + 2 * nvd;
            const bool in  = n <= T && m_in;
            const Real* src = p.sp + (ibase + 2 * (n - m) + imag) * p.ns;
            for (int f = lane; f < p.ns; f += 64) {
                out[f] = in ? src[f] : (Real)0;
            }
        }
    }
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
    // vor/div fields from 20 on and enough coefficients to fill the device with its chunks (T >= 255): the streaming form (a lane per field);
    // ATLAS_AMD_PREPARE=rows / stream forces one of the two (same bits either way: tests/test_gpu_vordiv.py)
    const char* e     = atlas_amd::env_get("ATLAS_AMD_PREPARE");
    // (the streaming form walks groups of 64 vor/div fields: it has nothing to do -- and divides by the group count -- without any,
    // e.g. the scalar chunks of a pipelined vor/div call, whatever the override says)
    const bool stream_form = nvd > 0 && (e && *e ? std::string(e) == "stream" : (nvd >= 20 && total_n >= PREP_SNB * 2048));   // [r6] 20, not 48: tools/probe/prepare_sweep.py (24 - 47 vor/div fields: 1.4 - 2.2 x faster)
    if (stream_form) {
        dim3 grid((T + 2 + PREP_SNB - 1) / PREP_SNB, T + 2);
        hipLaunchKernelGGL(spectra_prepare_stream_kernel<Real>, grid, dim3(256), 0, stream, p);
        return hipGetLastError();
    }
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

// The same transposition with whole output rows per workgroup [r5]: 64 points x ALL fields, the fields in blocks of 64 through a
// [64][65] LDS tile -- a wavefront reads 512 contiguous bytes of one field and writes 512 contiguous bytes of one point, and the
// 64 x nf doubles a workgroup writes are ONE contiguous range of `field`, its 512-byte pieces written within a few iterations of
// each other.  (The 32 x 32 tiles above write 256-byte pieces of rows whose neighbours belong to workgroups scheduled far away:
// with rows of 137 doubles = 1096 bytes nearly every 128-byte line is written in two parts, at two times.)
__global__ void __launch_bounds__(256) gp_to_field_rows_kernel(const double* __restrict__ gp, double* __restrict__ field, long long npts,
                                                               int nf) {
    __shared__ double tile[64][65];
    const long long p0 = (long long)blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int np = npts - p0 < 64 ? (int)(npts - p0) : 64;
    for (int f0 = 0; f0 < nf; f0 += 64) {
        const int nfb = nf - f0 < 64 ? nf - f0 : 64;
        if (f0 > 0) {
            __syncthreads();
        }
        for (int r = wave; r < nfb; r += 4) {   // a field of the block: 64 points, contiguous in gp
            if (lane < np) {
                tile[lane][r] = gp[(long long)(f0 + r) * npts + p0 + lane];
            }
        }
        __syncthreads();
        for (int r = wave; r < np; r += 4) {    // a point: the block's fields, contiguous in field
            if (lane < nfb) {
                field[(p0 + r) * nf + f0 + lane] = tile[r][lane];
            }
        }
    }
}

// Whole rows where there are enough points for a workgroup per 64 of them to fill the device and enough fields for 512-byte pieces
// (a rank's band of O640 / 4, 137 levels: 264 -> 248 us; of O1280 / 8: 605 -> 470 us = 3.85 TB/s read + write); the tiles otherwise
// (20 000 points x 1370 fields: 114 against 245 us).  ATLAS_AMD_GP_TO_FIELD=tiles|rows forces one (tools/probe/gp_to_field_probe.py).
hipError_t launch_gp_to_field(const double* gp, double* field, long long npts, int nf, hipStream_t stream) {
    if (npts <= 0 || nf <= 0) {
        return hipSuccess;
    }
    const char* e   = atlas_amd::env_get("ATLAS_AMD_GP_TO_FIELD");
    const bool rows = e && *e ? std::string(e) == "rows" : (nf >= 32 && (npts + 63) / 64 >= 1024);
    if (!rows) {
        hipLaunchKernelGGL(gp_to_field_kernel, dim3((unsigned)((npts + 31) / 32), (unsigned)((nf + 31) / 32)), dim3(256), 0, stream,
                           gp, field, npts, nf);
    }
    else {
        hipLaunchKernelGGL(gp_to_field_rows_kernel, dim3((unsigned)((npts + 63) / 64)), dim3(256), 0, stream, gp, field, npts, nf);
    }
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
