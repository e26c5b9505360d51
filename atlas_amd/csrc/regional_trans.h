// Inverse transform to a regular longitude-latitude grid that is NOT a crop of a global grid: arbitrary latitudes, equally
// spaced longitudes west + i * dlon.  Reference behaviour (ecmwf/atlas 0.44.1, the "no_nest" branch of TransLocal):
//   * src/atlas/trans/local/TransLocal.cc:394-406   no hemisphere symmetry is assumed for the target, no FFT; nlat0 = 0 for
//     every wavenumber (:463-468): no Fourier truncation towards the poles
//   * :535-557   Legendre polynomials at the grid's own latitudes (clamped to +-89.9999999)
//   * :719-738   Fourier matrix cos(m lon) * factor, -sin(m lon) * factor, factor = 2 for m > 0
//   * :1139-1148 grid points = Fourier matrix x Fourier coefficients (DFT as a matrix product), layout gp[lon + nlons (lat +
//     nlats field)].
// One deviation: at latitude -90 the reference's Legendre routine takes cos(colatitude) = +1 (it replaces it within a metre
// of either pole, LegendrePolynomials.cc:74-77; this branch does not mirror southern latitudes first) and returns wrong
// polynomials; here the south-pole row is the mirror image of the north-pole row, Pbar_n^m(-x) = (-1)^(n+m) Pbar_n^m(x).
// Here: the Legendre stage is the same MFMA kernel as for global grids, run on an internal symmetric latitude set (the
// target's |latitudes| and their mirror images; the rows that are not target rows are computed and not used), the
// Fourier stage is the matrix product of the reference (:1139-1148) with the precomputed cos / sin table, one fp64-MFMA kernel over all
// target rows and fields (unstructured targets: a per-point sum).
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <vector>

#include "trans.h"

namespace atlas_amd {
namespace trans {

class RegionalTrans {
public:
    // latitudes in degrees in the order of the target's rows (any order), longitudes west + i * dlon, i < nlon
    RegionalTrans(int nlon, double west, double dlon, const std::vector<double>& lats_deg, int truncation);
    // unstructured target: a list of (lon, lat) points in degrees (TransLocal.cc:741-790, 1293-1420: Legendre polynomials at
    // every point's latitude, Fourier sum evaluated point by point, grid points gp[point + npts * field]; u and v divided by
    // the cosine of the point's own, unclamped latitude :1277-1283)
    RegionalTrans(const std::vector<double>& lons_deg, const std::vector<double>& lats_deg, int truncation);
    ~RegionalTrans();
    RegionalTrans(const RegionalTrans&)            = delete;
    RegionalTrans& operator=(const RegionalTrans&) = delete;

    int truncation() const { return T_; }
    int nlon() const { return nlon_; }
    int nlat() const { return (int)rowsel_.size(); }
    bool unstructured() const { return d_lon_ != nullptr; }
    int64_t nb_gridpoints() const { return unstructured() ? (int64_t)rowsel_.size() : (int64_t)nlon_ * nlat(); }
    size_t nb_spectral_coefficients() const { return (size_t)(T_ + 1) * (T_ + 2); }
    hipStream_t stream() const { return inner_->stream(); }
    void synchronize() const { inner_->synchronize(); }

    // TransLocal::invtrans(nb_scalar_fields, scalar_spectra, gp_fields) for this target: device / host pointers
    void invtrans_scalar_device(int nb_fields, const double* sp_dev, double* gp_dev);
    void invtrans(int nb_fields, const double* scalar_spectra, double* gp_fields);
    // TransLocal::invtrans(nb_scalar, sp, nb_vordiv, vor, div, gp): gp = [u fields][v fields][scalar fields] (TransLocal.cc:
    // 1523-1597: truncation extended by one, vd2uv, u and v divided by cos(lat))
    void invtrans_vordiv_device(int nb_scalar, const double* sp_dev, int nb_vordiv, const double* vor_dev, const double* div_dev,
                                double* gp_dev);
    void invtrans(int nb_scalar, const double* sp, int nb_vordiv, const double* vor, const double* div, double* gp);

private:
    int T_ = 0, nlon_ = 0;
    std::unique_ptr<Trans> inner_;      // Legendre stage on the symmetric latitude set, rows = the range the target needs
    std::vector<int> rowsel_;           // per target row: row of the inner object's Fourier intermediate
    int* d_rowsel_     = nullptr;
    double* d_table_   = nullptr;       // [T+1][2][nlon]: cos(m lon) * factor, -sin(m lon) * factor
    double* d_lon_     = nullptr;       // unstructured target: longitude of every point in radians
    void make_inner(const std::vector<double>& lats_deg, bool clamp_scale);
    double* d_scale_   = nullptr;       // per target row: 1 / cos(latitude), latitude clamped as TransLocal.cc:1449-1456
    double* d_sp_      = nullptr;
    double* d_gp_      = nullptr;
    double* d_all_     = nullptr;       // combined (U, V, scalar) spectra of truncation T + 1
    double* d_vd_      = nullptr;       // host-API staging of vor ++ div
    size_t sp_cap_ = 0, gp_cap_ = 0, all_cap_ = 0, vd_cap_ = 0;
    void dft(int trc_in, int nb_fields, int nb_vordiv, const double* F, double* gp_dev);
    void ensure(double*& ptr, size_t& cap, size_t n);
};

}  // namespace trans
}  // namespace atlas_amd
