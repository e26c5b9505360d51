// See regional_trans.h.
#include "env.h"
#include "regional_trans.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>
#include <string>

#include "gaussian.h"
#include "trans_plan.h"

namespace atlas_amd {
namespace trans {

namespace {

#define RT_CHECK(call)                                                                                                \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess) {                                                                                       \
            (void)hipGetLastError(); /* reported here, once: not left sticky for the next call */ \
            throw std::runtime_error(std::string("HIP error '") + hipGetErrorString(e_) + "' in " #call);              \
        }                                                                                                             \
    } while (0)

// gp[f][row][lon] = sum_m  C[m][lon] Re F(row, m, f) + S[m][lon] Im F(row, m, f): ONE matrix product over all target rows and fields
// (TransLocal.cc:1139-1148 hands exactly this product to its GEMM backend),
//     out[p][i] = sum_k  A[p][k] B[k][i],   p = row * nf + field,   k = 2 m + (0: real, 1: imaginary),   B = the cos / sin table,
// on v_mfma_f64_16x16x4_f64.  A workgroup of eight wavefronts owns a 128 x 128 tile of (p, i), each wavefront 64 x 32 = 8 accumulator
// tiles; the contraction runs in stages of 8 wavenumbers (16 rows of B) through two LDS buffers: the loads of stage c + 1 are in
// flight in registers while stage c is multiplied.  LDS rows have a pitch of 144 doubles, so the four 16-lane groups of an operand
// read (rows k .. k + 3 of the stage) fall on disjoint banks per half wavefront.  Workgroup -> tile: an XCD (blockIdx & 7) walks a
// contiguous run of tiles, longitude tiles fastest: the workgroups that run together on it share their A and B panels in its L2.
constexpr int GT  = 128;       // tile edge
constexpr int GKM = 8;         // wavenumbers per stage
constexpr int GLD = GT + 16;   // LDS row pitch in doubles
typedef double dft_acc_t __attribute__((ext_vector_type(4)));
typedef double dft_pair_t __attribute__((ext_vector_type(2)));

// NW = wavefronts per workgroup: 8 (64 x 32 of the tile each, 112 registers, 4 wavefronts per SIMD) is 8 % faster than 4 (64 x 64 each,
// 200 registers, 2 per SIMD) on a 1000 x 500 target at T1279 / 137 fields: 6.3 against 7.1 ms = 55 TFLOP/s (profiles/r06_regional.txt)
constexpr int DFT_NW = 8;
template <int NW>
__global__ void __launch_bounds__(64 * NW, NW / 2)
    regional_dft_mfma_kernel(const double* __restrict__ F, const int* __restrict__ rowsel, const double* __restrict__ table,
                             double* __restrict__ gp, int T, int RP, int nlon, int nlat, int nf, const double* __restrict__ rowscale,
                             int nscaled, int tiles_i, int total_tiles, int per_xcd) {
    extern __shared__ double lds[];   // [2 stages][A: 16 x GLD | B: 16 x GLD]
    const int slot = blockIdx.x >> 3, lin = (blockIdx.x & 7) * per_xcd + slot;
    if (slot >= per_xcd || lin >= total_tiles) {
        return;
    }
    const int p0 = (lin / tiles_i) * GT, i0 = (lin % tiles_i) * GT;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int P = nlat * nf, K2 = 2 * (T + 1);
    // loader roles: element tid & 127 of the tile edge, rows (tid >> 7) + LR q of the stage
    constexpr int LR = NW / 2;       // loader rows per pass
    constexpr int UW = 16 / NW;      // 16-longitude tiles per wavefront: 4 or 2
    const int le = tid & 127, lr = tid >> 7;
    const int pa     = p0 + le;
    const bool pa_ok = pa < P;
    const int row_a  = pa_ok ? pa / nf : 0;
    const double* asrc = F + (long long)rowsel[row_a] * (T + 1) * RP + 2 * (pa_ok ? pa - row_a * nf : 0);
    const int ib     = i0 + le;
    const bool ib_ok = ib < nlon;
    const double* bsrc = table + (ib_ok ? ib : 0);
    dft_pair_t ra[GKM / LR];
    double rb[2 * GKM / LR];
    // loads are unconditional from clamped (valid) addresses -- straight-line code; what lies outside the problem is zeroed on the
    // way to LDS
    auto fetch = [&](int c) {
#pragma unroll
        for (int q = 0; q < GKM / LR; ++q) {
            const int m = min(c * GKM + lr + LR * q, T);
            ra[q]       = *reinterpret_cast<const dft_pair_t*>(asrc + (long long)m * RP);
        }
#pragma unroll
        for (int q = 0; q < 2 * GKM / LR; ++q) {
            const int k = min(c * 2 * GKM + lr + LR * q, K2 - 1);
            rb[q]       = bsrc[(long long)k * nlon];
        }
    };
    auto stash = [&](int c, double* buf) {
        double* a = buf;
        double* b = buf + 2 * GKM * GLD;
#pragma unroll
        for (int q = 0; q < GKM / LR; ++q) {
            const int ml   = lr + LR * q;
            const bool ok  = pa_ok && c * GKM + ml <= T;
            a[(2 * ml) * GLD + le]     = ok ? ra[q].x : 0.;
            a[(2 * ml + 1) * GLD + le] = ok ? ra[q].y : 0.;
        }
#pragma unroll
        for (int q = 0; q < 2 * GKM / LR; ++q) {
            const int kl = lr + LR * q;
            b[kl * GLD + le] = (ib_ok && c * 2 * GKM + kl < K2) ? rb[q] : 0.;
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
        const double* a = lds + (c & 1) * STAGE;
        const double* b = a + 2 * GKM * GLD;
        if (c + 1 < nstage) {
            fetch(c + 1);
        }
#pragma unroll
        for (int kk = 0; kk < 2 * GKM / 4; ++kk) {
            double fa[4], fb[UW];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = a[(4 * kk + l4) * GLD + pw + 16 * t + l15];
            }
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                fb[u] = b[(4 * kk + l4) * GLD + iw + 16 * u + l15];
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
            const int row = p / nf, f = p - row * nf;
            const double scale = f < nscaled ? rowscale[row] : 1.;   // u, v fields of the vor/div path: 1 / cos(lat)
            double* out        = gp + ((long long)f * nlat + row) * nlon;
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int i = i0 + iw + 16 * u + l15;
                if (i < nlon) {
                    out[i] = acc[t][u][r] * scale;
                }
            }
        }
    }
}

// unstructured target: gp[f][point] = sum_m factor (cos(m lon) Re F - sin(m lon) Im F)(row of the point, m, f)
// One thread per point and group of 8 fields: the sine and cosine of m lon are computed once per wavenumber.
constexpr int PFG = 8;
__global__ void __launch_bounds__(256) points_dft_kernel(const double* __restrict__ F, const int* __restrict__ rowsel,
                                                         const double* __restrict__ lon, double* __restrict__ gp, int T, int m_cnt, int RP,
                                                         long long npts, int nf, const double* __restrict__ scale, int nscaled) {
    const long long pt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int f0       = blockIdx.y * PFG;
    if (pt >= npts) {
        return;
    }
    const double* src = F + (long long)rowsel[pt] * m_cnt * RP + 2 * f0;
    const double lam  = lon[pt];
    double acc[PFG];
#pragma unroll
    for (int k = 0; k < PFG; ++k) {
        acc[k] = 0.;
    }
    for (int m = 0; m <= T; ++m) {
        double sn, cs;
        sincos(m * lam, &sn, &cs);
        const double fr = m > 0 ? 2. * cs : 1., fi = m > 0 ? -2. * sn : 0.;   // TransLocal.cc:1253-1259
        const double* c = src + (long long)m * RP;
#pragma unroll
        for (int k = 0; k < PFG; ++k) {
            if (f0 + k < nf) {
                acc[k] += fr * c[2 * k] + fi * c[2 * k + 1];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PFG; ++k) {
        if (f0 + k < nf) {
            gp[(long long)(f0 + k) * npts + pt] = acc[k] * (f0 + k < nscaled ? scale[pt] : 1.);
        }
    }
}

}  // namespace

// the symmetric latitude set: |latitudes| (clamped as TransLocal.cc:537-543), north -> equator, then their mirror images; the
// inner object's Legendre stage covers the row range the target needs; rowsel_ maps target rows / points to its rows
void RegionalTrans::make_inner(const std::vector<double>& lats_deg, bool clamp_scale) {
    std::vector<double> a;
    const char* rp_env         = atlas_amd::env_get("ATLAS_AMD_REFERENCE_POLES");   // read per object
    const bool reference_poles = rp_env && atoi(rp_env) != 0;
    // ATLAS_AMD_REFERENCE_POLES=1 ("Deviations"): on this branch the reference calls its Legendre routine with the
    // target's own, unmirrored latitudes; within a metre of EITHER pole (sin(colatitude) <= sqrt(epsilon)) that routine sets
    // cos(colatitude) = +1 and sin = 0 for its recurrences while the cos(j theta) / sin(j theta) of its series keep the true
    // colatitude (LegendrePolynomials.cc:58-76): at the SOUTH pole the m = 0, 1 columns and the recurrences for m >= 2 disagree
    // about the sign of cos(theta) and the row comes out as neither pole's polynomials.  The default here is the mirror image of the
    // north-pole row; with the switch such a row becomes a Legendre row of its own whose table row is generated by the same routine
    // (csrc/legendre_series.h, the bit-identical twin of the reference's) AT THE NEGATIVE LATITUDE -- the reference's numbers.
    auto south_polar = [&](double y) {
        if (!reference_poles || y >= 0) {
            return false;
        }
        const double lat        = std::max(y, -kLatPole) * (M_PI / 180.);
        const double zdlx       = std::cos(M_PI_2 - lat);
        volatile double zdlsita = std::sqrt(1. - zdlx * zdlx);
        return std::fabs(zdlsita) <= std::sqrt(std::numeric_limits<double>::epsilon());
    };
    const double pseudo = kLatPole - 1.e-10;   // |latitude| that stands for "south pole as the reference computes it" in the symmetric set
    bool have_pseudo    = false;
    for (double y : lats_deg) {
        if (!(y >= -90. && y <= 90.)) {
            throw std::invalid_argument("RegionalTrans: latitude outside [-90, 90]");
        }
        if (south_polar(y)) {
            if (!have_pseudo) {
                a.push_back(pseudo);
            }
            have_pseudo = true;
            continue;
        }
        a.push_back(std::min(std::fabs(y), kLatPole));
    }
    std::sort(a.begin(), a.end(), [](double p, double q) { return p > q; });
    a.erase(std::unique(a.begin(), a.end(), [](double p, double q) { return std::fabs(p - q) < 1.e-12; }), a.end());
    const bool equator = std::fabs(a.back()) < 1.e-12;
    grid::StructuredGrid sym;
    sym.name = "regional";
    for (size_t k = 0; k < a.size(); ++k) {
        sym.y.push_back(equator && k + 1 == a.size() ? 0. : a[k]);
    }
    for (int k = (int)a.size() - 1 - (equator ? 1 : 0); k >= 0; --k) {
        sym.y.push_back(-a[k]);
    }
    // a regular row length for which no wavenumber is truncated at any latitude (TransLocal.cc:463-468: nlat0 = 0)
    sym.nx.assign(sym.y.size(), 4 * (T_ + 1));
    sym.regular = true;
    auto row_of = [&](double y) {
        const bool quirk = south_polar(y);
        const double v   = quirk ? pseudo : std::min(std::fabs(y), kLatPole);
        // a is sorted descending: binary search for the entry within the merge tolerance
        int lo = 0, hi = (int)a.size() - 1;
        while (lo < hi) {
            const int mid = (lo + hi) / 2;
            if (a[mid] > v + 1.e-12) {
                lo = mid + 1;
            }
            else {
                hi = mid;
            }
        }
        const int k = lo;
        if (quirk) {
            return k;   // the NORTHERN output of the pseudo row: sym + asym = the plain sum over n of the table row below
        }
        return y >= 0 || (equator && k == (int)a.size() - 1) ? k : (int)sym.y.size() - 1 - k;
    };
    std::vector<int> rows;
    for (double y : lats_deg) {
        rows.push_back(row_of(y));
    }
    TransConfig cfg;
    if (have_pseudo) {   // the pseudo row's polynomials: the routine evaluated at the clamped SOUTHERN latitude (TransLocal.cc:537-543)
        cfg.leg_lat_override.push_back({row_of(-90.), -kLatPole * (M_PI / 180.)});
    }
    cfg.row_begin = *std::min_element(rows.begin(), rows.end());
    cfg.row_end   = *std::max_element(rows.begin(), rows.end()) + 1;
    inner_.reset(new Trans(sym, T_, cfg));
    for (int m = 0; m <= T_; ++m) {
        if (inner_->geometry().nlat0[m] != 0) {
            throw std::logic_error("RegionalTrans: internal latitude set truncates wavenumber " + std::to_string(m));
        }
    }
    for (int r : rows) {
        rowsel_.push_back(r - cfg.row_begin);
    }
    std::vector<double> scale;
    for (double y : lats_deg) {
        const double lat = clamp_scale ? std::max(std::min(y, kLatPole), -kLatPole) : y;
        scale.push_back(1. / std::cos(lat * (M_PI / 180.)));
    }
    RT_CHECK(hipMalloc((void**)&d_scale_, scale.size() * sizeof(double)));
    RT_CHECK(hipMemcpy(d_scale_, scale.data(), scale.size() * sizeof(double), hipMemcpyHostToDevice));
    RT_CHECK(hipMalloc((void**)&d_rowsel_, rowsel_.size() * sizeof(int)));
    RT_CHECK(hipMemcpy(d_rowsel_, rowsel_.data(), rowsel_.size() * sizeof(int), hipMemcpyHostToDevice));
}

RegionalTrans::RegionalTrans(int nlon, double west, double dlon, const std::vector<double>& lats_deg, int truncation) :
    T_(truncation), nlon_(nlon) {
    if (nlon < 1 || lats_deg.empty() || truncation < 0) {
        throw std::invalid_argument("RegionalTrans: needs nlon >= 1, at least one latitude and truncation >= 0");
    }
    // Fourier matrix (TransLocal.cc:719-738), computed on the host with the same libm calls
    std::vector<double> table((size_t)2 * (truncation + 1) * nlon);
    for (int m = 0; m <= truncation; ++m) {
        const double factor = m > 0 ? 2. : 1.;
        for (int i = 0; i < nlon; ++i) {
            const double lon                   = (west + i * dlon) * (M_PI / 180.);
            table[(size_t)(2 * m) * nlon + i]     = +std::cos(m * lon) * factor;
            table[(size_t)(2 * m + 1) * nlon + i] = -std::sin(m * lon) * factor;
        }
    }
    try {
        make_inner(lats_deg, true);
        RT_CHECK(hipMalloc((void**)&d_table_, table.size() * sizeof(double)));
        RT_CHECK(hipMemcpy(d_table_, table.data(), table.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    catch (...) {
        (void)hipFree(d_table_);
        (void)hipFree(d_rowsel_);
        (void)hipFree(d_scale_);
        throw;
    }
}

RegionalTrans::RegionalTrans(const std::vector<double>& lons_deg, const std::vector<double>& lats_deg, int truncation) :
    T_(truncation), nlon_(0) {
    if (lons_deg.empty() || lons_deg.size() != lats_deg.size() || truncation < 0) {
        throw std::invalid_argument("RegionalTrans: needs as many longitudes as latitudes, at least one, and truncation >= 0");
    }
    std::vector<double> lon;
    for (double v : lons_deg) {
        lon.push_back(v * (M_PI / 180.));
    }
    try {
        make_inner(lats_deg, false);
        RT_CHECK(hipMalloc((void**)&d_lon_, lon.size() * sizeof(double)));
        RT_CHECK(hipMemcpy(d_lon_, lon.data(), lon.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    catch (...) {
        (void)hipFree(d_lon_);
        (void)hipFree(d_rowsel_);
        (void)hipFree(d_scale_);
        throw;
    }
}

RegionalTrans::~RegionalTrans() {
    if (inner_) {
        (void)hipStreamSynchronize(inner_->stream());
    }
    (void)hipFree(d_table_);
    (void)hipFree(d_lon_);
    (void)hipFree(d_rowsel_);
    (void)hipFree(d_scale_);
    (void)hipFree(d_sp_);
    (void)hipFree(d_gp_);
    (void)hipFree(d_all_);
    (void)hipFree(d_vd_);
}

void RegionalTrans::dft(int trc_in, int nb_fields, int nb_vordiv, const double* F, double* gp_dev) {
    (void)trc_in;   // the Legendre stage left zeros for the wavenumbers it does not transform
    if (unstructured()) {
        const long long npts = nb_gridpoints();
        hipLaunchKernelGGL(points_dft_kernel, dim3((unsigned)((npts + 255) / 256), (unsigned)((nb_fields + PFG - 1) / PFG)), dim3(256), 0,
                           inner_->stream(), F, d_rowsel_, d_lon_, gp_dev, T_, T_ + 1, inner_->fourier_row_pitch(nb_fields), npts,
                           nb_fields, d_scale_, 2 * nb_vordiv);
        RT_CHECK(hipGetLastError());
        return;
    }
    const size_t lds = (size_t)2 * 2 * 2 * GKM * GLD * sizeof(double);   // two stages of A and B: 73 728 bytes, two workgroups per CU
    {   // on every launch (cheap): a per-process flag is wrong for a second device and racy between host threads
        RT_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&regional_dft_mfma_kernel<DFT_NW>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    const long long pairs = (long long)nlat() * nb_fields;
    const long long tiles_p = (pairs + GT - 1) / GT, tiles_i = (nlon_ + GT - 1) / GT, total = tiles_p * tiles_i;
    if (pairs > std::numeric_limits<int>::max() || total > (1LL << 28)) {
        throw std::runtime_error("RegionalTrans: target rows x fields beyond the range of the Fourier kernel's indices");
    }
    const int per_xcd = (int)((total + 7) / 8);
    hipLaunchKernelGGL(regional_dft_mfma_kernel<DFT_NW>, dim3((unsigned)(8 * per_xcd)), dim3(64 * DFT_NW), lds, inner_->stream(), F, d_rowsel_,
                       d_table_, gp_dev, T_, inner_->fourier_row_pitch(nb_fields), nlon_, nlat(), nb_fields, d_scale_, 2 * nb_vordiv,
                       (int)tiles_i, (int)total, per_xcd);
    RT_CHECK(hipGetLastError());
}

void RegionalTrans::invtrans_scalar_device(int nb_fields, const double* sp_dev, double* gp_dev) {
    if (nb_fields <= 0) {
        return;
    }
    double* F = inner_->fourier_buffer(nb_fields);
    inner_->legendre_device(T_, nb_fields, sp_dev, F);
    dft(T_, nb_fields, 0, F, gp_dev);
}

hipError_t launch_spectra_prepare(const double* vor, const double* div, const double* sp, double* out, int T, int nvd, int ns,
                                  hipStream_t stream);

void RegionalTrans::invtrans_vordiv_device(int nb_scalar, const double* sp_dev, int nb_vordiv, const double* vor_dev,
                                           const double* div_dev, double* gp_dev) {
    if (nb_vordiv <= 0) {
        invtrans_scalar_device(nb_scalar, sp_dev, gp_dev);
        return;
    }
    const int nall    = 2 * nb_vordiv + nb_scalar;
    const size_t nout = size_t(T_ + 2) * size_t(T_ + 3) * size_t(nall);
    ensure(d_all_, all_cap_, nout);
    RT_CHECK(launch_spectra_prepare(vor_dev, div_dev, sp_dev, d_all_, T_, nb_vordiv, nb_scalar, inner_->stream()));
    double* F = inner_->fourier_buffer(nall);
    inner_->legendre_device(T_ + 1, nall, d_all_, F);   // TransLocal.cc:1590
    dft(T_ + 1, nall, nb_vordiv, F, gp_dev);
}

void RegionalTrans::ensure(double*& ptr, size_t& cap, size_t n) {
    if (n > cap) {
        synchronize();
        (void)hipFree(ptr);
        ptr = nullptr;
        RT_CHECK(hipMalloc((void**)&ptr, n * sizeof(double)));
        cap = n;
    }
}

void RegionalTrans::invtrans(int nb_scalar, const double* sp, int nb_vordiv, const double* vor, const double* div, double* gp) {
    if (nb_vordiv <= 0) {
        invtrans(nb_scalar, sp, gp);
        return;
    }
    const size_t nspec = nb_spectral_coefficients();
    const size_t ngp   = (size_t)nb_gridpoints() * (size_t)(2 * nb_vordiv + nb_scalar);
    ensure(d_sp_, sp_cap_, std::max<size_t>(nspec * (size_t)nb_scalar, 1));
    ensure(d_vd_, vd_cap_, 2 * nspec * (size_t)nb_vordiv);
    ensure(d_gp_, gp_cap_, ngp);
    if (nb_scalar > 0) {
        RT_CHECK(hipMemcpyAsync(d_sp_, sp, nspec * nb_scalar * sizeof(double), hipMemcpyHostToDevice, inner_->stream()));
    }
    double* d_vor = d_vd_;
    double* d_div = d_vd_ + nspec * (size_t)nb_vordiv;
    RT_CHECK(hipMemcpyAsync(d_vor, vor, nspec * nb_vordiv * sizeof(double), hipMemcpyHostToDevice, inner_->stream()));
    RT_CHECK(hipMemcpyAsync(d_div, div, nspec * nb_vordiv * sizeof(double), hipMemcpyHostToDevice, inner_->stream()));
    invtrans_vordiv_device(nb_scalar, nb_scalar > 0 ? d_sp_ : nullptr, nb_vordiv, d_vor, d_div, d_gp_);
    RT_CHECK(hipMemcpyAsync(gp, d_gp_, ngp * sizeof(double), hipMemcpyDeviceToHost, inner_->stream()));
    synchronize();
}

void RegionalTrans::invtrans(int nb_fields, const double* scalar_spectra, double* gp_fields) {
    if (nb_fields <= 0) {
        return;
    }
    const size_t nsp = nb_spectral_coefficients() * (size_t)nb_fields, ngp = (size_t)nb_gridpoints() * (size_t)nb_fields;
    ensure(d_sp_, sp_cap_, nsp);
    ensure(d_gp_, gp_cap_, ngp);
    RT_CHECK(hipMemcpyAsync(d_sp_, scalar_spectra, nsp * sizeof(double), hipMemcpyHostToDevice, inner_->stream()));
    invtrans_scalar_device(nb_fields, d_sp_, d_gp_);
    RT_CHECK(hipMemcpyAsync(gp_fields, d_gp_, ngp * sizeof(double), hipMemcpyDeviceToHost, inner_->stream()));
    synchronize();
}

}  // namespace trans
}  // namespace atlas_amd
