// See dft_gemm.h.
//     out[p][i] = sum_k  A[p][k] B[k][i],   p = row * nf + field,   k = 2 m + (0: real, 1: imaginary),   B = the cos / sin table,
// on v_mfma_f64_16x16x4_f64.  A workgroup of eight wavefronts owns a 128 x 128 tile of (p, i), each wavefront 64 x 32 = 8 accumulator
// tiles; the contraction runs in stages of 8 wavenumbers (16 rows of B) through two LDS buffers: the loads of stage c + 1 are in
// flight in registers while stage c is multiplied.  LDS rows have a pitch of 144 doubles, so the four 16-lane groups of an operand
// read (rows k .. k + 3 of the stage) fall on disjoint banks per half wavefront.  Workgroup -> tile: an XCD (blockIdx & 7) walks a
// contiguous run of tiles, longitude tiles fastest: the workgroups that run together on it share their A and B panels in its L2.
#include "dft_gemm.h"

#include <limits>

#include "device_structs.h"
#include "fft_device.h"

namespace atlas_amd {
namespace trans {

namespace {

constexpr int GT  = 128;       // tile edge
constexpr int GKM = 8;         // wavenumbers per stage
constexpr int GLD = GT + 16;   // LDS row pitch in doubles
typedef double dft_acc_t __attribute__((ext_vector_type(4)));
typedef double dft_pair_t __attribute__((ext_vector_type(2)));

// NW = wavefronts per workgroup: 8 (64 x 32 of the tile each, 112 registers, 4 wavefronts per SIMD) is 8 % faster than 4 (64 x 64 each,
// 200 registers, 2 per SIMD) on a 1000 x 500 target at T1279 / 137 fields: 6.3 against 7.1 ms = 55 TFLOP/s (profiles/r06_regional.txt)
constexpr int DFT_NW = 8;
template <int NW, bool F32>
__global__ void __launch_bounds__(64 * NW, NW / 2) dft_gemm_kernel(DftGemmArgs a, int tiles_i, int total_tiles, int per_xcd) {
    extern __shared__ double lds[];   // [2 stages][A: 16 x GLD | B: 16 x GLD]
    const int slot = blockIdx.x >> 3, lin = (blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || lin >= total_tiles) {
        return;
    }
    const int T = a.T, nf = a.nf, nlon = a.nlon;
    const long long RP = a.RP;
    const int p0 = (lin / tiles_i) * GT, i0 = (lin % tiles_i) * GT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int P = a.nrows * nf, K2 = 2 * (T + 1);
    // loader roles: element tid & 127 of the tile edge, rows (tid >> 7) + LR q of the stage
    constexpr int LR = NW / 2;       // loader rows per pass
    constexpr int UW = 16 / NW;      // 16-longitude tiles per wavefront: 4 or 2
    const int le = tid & 127, lr = tid >> 7;
    const int pa     = p0 + le;
    const bool pa_ok = pa < P;
    const int row_a  = pa_ok ? pa / nf : 0;
    const int mtop   = a.rowmmax ? min(T, a.rowmmax[row_a]) : T;   // the highest wavenumber the intermediate holds for this row
    const double* asrc = a.F + ((long long)a.rowsel[row_a] * a.m_cnt) * RP + 2 * (a.f0 + (pa_ok ? pa - row_a * nf : 0));
    const int ib     = i0 + le;
    const bool ib_ok = ib < nlon;
    const double* bsrc = a.table + (ib_ok ? ib : 0);
    dft_pair_t ra[GKM / LR];
    double rb[2 * GKM / LR];
    // loads are unconditional from clamped (valid) addresses -- straight-line code; what lies outside the problem is zeroed on the
    // way to LDS
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < GKM / LR; ++q) {
            const int m = max(0, min(c * GKM + lr + LR * q, mtop));
            ra[q]       = *reinterpret_cast<const dft_pair_t*>(asrc + (long long)m * RP);
        }
#pragma unroll
        for (int q = 0; q < 2 * GKM / LR; ++q) {
            const int k = min(c * 2 * GKM + lr + LR * q, K2 - 1);
            rb[q]       = bsrc[(long long)k * nlon];
        }
    };
    auto stash = [&](int c, double* buf) {
        double* sa = buf;
        double* sb = buf + 2 * GKM * GLD;
#pragma unroll
        for (int q = 0; q < GKM / LR; ++q) {
            const int ml   = lr + LR * q;
            const bool ok  = pa_ok && c * GKM + ml <= mtop;
            sa[(2 * ml) * GLD + le]     = ok ? ra[q].x : 0.;
            sa[(2 * ml + 1) * GLD + le] = ok ? ra[q].y : 0.;
        }
#pragma unroll
        for (int q = 0; q < 2 * GKM / LR; ++q) {
            const int kl = lr + LR * q;
            sb[kl * GLD + le] = (ib_ok && c * 2 * GKM + kl < K2) ? rb[q] : 0.;
        }
    };
    dft_acc_t acc[4][UW];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < UW; ++u) {
            acc[t][u] = dft_acc_t{0., 0., 0., 0.};
        }
    }
    const int pw = (w / (NW / 2)) * 64, iw = (w % (NW / 2)) * 16 * UW;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int nstage = (T + GKM) / GKM;   // ceil((T + 1) / GKM)
    constexpr int STAGE = 2 * 2 * GKM * GLD;
    fetch(0);
    stash(0, lds);
    __syncthreads();
    for (int c = 0; c < nstage; ++c) {
        const double* sa = lds + (c & 1) * STAGE;
        const double* sb = sa + 2 * GKM * GLD;
        if (c + 1 < nstage) {
            fetch(c + 1);
        }
#pragma unroll
        for (int kk = 0; kk < 2 * GKM / 4; ++kk) {
            double fa[4], fb[UW];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = sa[(4 * kk + l4) * GLD + pw + 16 * t + l15];
            }
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                fb[u] = sb[(4 * kk + l4) * GLD + iw + 16 * u + l15];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int u = 0; u < UW; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[t], fb[u], acc[t][u], 0, 0, 0);
                }
            }
        }
        if (c + 1 < nstage) {
            stash(c + 1, lds + ((c + 1) & 1) * STAGE);
        }
        __syncthreads();
    }
    // result element (row of the MFMA tile = (lane >> 4) + 4 reg, column = lane & 15): p = pair, column = longitude
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = p0 + pw + 16 * t + l4 + 4 * r;
            if (p >= P) {
                continue;
            }
            const int row = p / nf, f = a.f0 + (p - row * nf);
            const double scale  = f < a.nscaled ? a.rowscale[row] : 1.;   // u, v fields of the vor/div path: 1 / cos(lat)
            const long long off = (long long)f * a.fstride + (a.rowout ? a.rowout[row] : (long long)row * nlon);
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int i = i0 + iw + 16 * u + l15;
                if (i < nlon) {
                    if constexpr (F32) {
                        reinterpret_cast<float*>(a.out)[off + i] = (float)(acc[t][u][r] * scale);
                    }
                    else {
                        reinterpret_cast<double*>(a.out)[off + i] = acc[t][u][r] * scale;
                    }
                }
            }
        }
    }
}

template <bool F32>
hipError_t launch_t(const DftGemmArgs& a, hipStream_t stream) {
    const size_t lds = (size_t)2 * 2 * 2 * GKM * GLD * sizeof(double);   // two stages of A and B: 73 728 bytes, two workgroups per CU
    // on every launch (cheap): a per-process flag is wrong for a second device and racy between host threads
    if (hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dft_gemm_kernel<DFT_NW, F32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        e != hipSuccess) {
        return e;
    }
    const long long pairs   = (long long)a.nrows * a.nf;
    const long long tiles_p = (pairs + GT - 1) / GT, tiles_i = (a.nlon + GT - 1) / GT, total = tiles_p * tiles_i;
    if (pairs > std::numeric_limits<int>::max() || total > (1LL << 28)) {
        return hipErrorInvalidValue;   // rows x fields beyond the range of the kernel's indices
    }
    const int per_xcd = (int)((total + 7) / 8);
    hipLaunchKernelGGL((dft_gemm_kernel<DFT_NW, F32>), dim3((unsigned)(8 * per_xcd)), dim3(64 * DFT_NW), lds, stream, a, (int)tiles_i,
                       (int)total, per_xcd);
    return hipGetLastError();
}

// one workgroup per (row of the list, block of 8 wavenumbers); threads over (wavenumber, field): a wavefront reads consecutive fields of one record
__global__ void __launch_bounds__(256) gather_rows_dense_kernel(FourierParams p, const int* __restrict__ rows, double* __restrict__ dense,
                                                                int RPd) {
    const int r   = blockIdx.x;
    const int row = rows[r];
    const int nfs = p.f_end - p.f_begin;
    const int top = p.row_mmax[row] < p.T ? p.row_mmax[row] : p.T;
    for (int q = threadIdx.x; q < 8 * nfs; q += blockDim.x) {
        const int m = blockIdx.y * 8 + q / nfs, f = q % nfs;
        if (m > p.T) {
            break;
        }
        fft::cplx v{0., 0.};
        if (m <= top) {
            const ModeReader rd{p, (long long)(row - p.lat0), 2 * (p.f_begin + f)};
            v = rd(m);
        }
        double* d = dense + ((long long)r * (p.T + 1) + m) * RPd + 2 * f;
        d[0]      = v.re;
        d[1]      = v.im;
    }
}

}  // namespace

hipError_t launch_gather_rows_dense(const FourierParams& p, const int* rows, int nrows, double* dense, int RPd, hipStream_t stream) {
    if (nrows <= 0 || p.f_end <= p.f_begin) {
        return hipSuccess;
    }
    hipLaunchKernelGGL(gather_rows_dense_kernel, dim3((unsigned)nrows, (unsigned)((p.T + 8) / 8)), dim3(256), 0, stream, p, rows, dense, RPd);
    return hipGetLastError();
}

hipError_t launch_dft_gemm(const DftGemmArgs& a, hipStream_t stream) {
    if (a.nrows <= 0 || a.nf <= 0 || a.nlon <= 0) {
        return hipSuccess;
    }
    return a.f32 ? launch_t<true>(a, stream) : launch_t<false>(a, stream);
}

}  // namespace trans
}  // namespace atlas_amd
