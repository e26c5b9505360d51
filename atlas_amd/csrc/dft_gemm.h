// The Fourier sum of rows as ONE matrix product on fp64 MFMA (dft_gemm.hip):
//     out[(row, field)][lon] = sum_k  A[(row, field)][k] B[k][lon],   k = 2 m + (0: real, 1: imaginary part of the Fourier coefficient),
// B = a table of cos / sin values of the rows' longitudes, A read in place from the Legendre stage's intermediate F[row][m][fields x 2].
// Two users:
//   * the no_nest branch of TransLocal (regional_trans.hip): the reference forms exactly this product (TransLocal.cc:719-738, 1139-1148);
//   * rows of a global grid whose transform does not fit a CU's LDS (trans.hip [r6]: more than 10 240 complex elements -- the four longest
//     row lengths of O2560): an exact evaluation of the same c2r sum (FFT.h:22-82) at 4 n (T + 1) flops per row and field instead of an FFT.
#pragma once
#include <hip/hip_runtime.h>

namespace atlas_amd {
namespace trans {

struct DftGemmArgs {
    const double* F;           // Fourier coefficients: element (row r, wavenumber m, column c) at (rowsel[r] * m_cnt + m) * RP + c
    int f32;                   // out holds floats (fp32 variant; the product itself is formed in fp64)
    const int* rowsel;         // [nrows] row of the intermediate
    const double* table;       // [2 (T + 1)][nlon]
    void* out;                 // out[field * fstride + rowout[r] + lon]
    const long long* rowout;   // [nrows], or null: r * nlon
    long long fstride;
    const double* rowscale;    // [nrows]: factor of the fields below nscaled (u, v of the vor/div path: 1 / cos(lat))
    const int* rowmmax;        // [nrows] highest wavenumber the intermediate holds for the row (what lies above is not read), or null: T
    int T, m_cnt, RP, nlon, nrows;
    int f0, nf;                // fields f0 .. f0 + nf - 1
    int nscaled;
};

hipError_t launch_dft_gemm(const DftGemmArgs& a, hipStream_t stream);

// The kept wavenumbers of a few rows of a global grid, read through the Fourier stage's own reader (any layout of the intermediate: single
// piece, m-sharded pieces, packed runs; double or float storage) into a dense array dense[(r * (T + 1) + m) * RPd + 2 (f - f_begin) + (0 | 1)],
// zeros above a row's highest kept wavenumber: the A operand of the product for the rows beyond the LDS (trans.hip).
struct FourierParams;
hipError_t launch_gather_rows_dense(const FourierParams& p, const int* rows, int nrows, double* dense, int RPd, hipStream_t stream);

}  // namespace trans
}  // namespace atlas_amd
