// Host computation of the normalised associated Legendre functions used by TransLocal
// (reference: src/atlas/trans/local/LegendrePolynomials.cc:24-209).  Arithmetic follows the reference
// operation by operation (same series, same recurrences, same evaluation order) so that the tables are the
// ones TransLocal would build; the recurrence coefficients are hoisted out of the latitude loop (they do not
// depend on latitude), which leaves every product/sum bit-identical.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "trans_plan.h"

namespace atlas_amd {
namespace trans {

class LegendreEvaluator {
public:
    explicit LegendreEvaluator(int trc);
    int trc() const { return trc_; }
    size_t triangle_size() const { return tri_; }
    static size_t idxmn(int trc, int m, int n) { return size_t(2 * trc + 3 - m) * size_t(m) / 2 + size_t(n - m); }
    // legpol: packed triangle of size triangle_size(); scratch: 2*(trc+1) doubles
    void evaluate(double lat_rad, double* legpol, double* scratch) const;
    // the latitude-dependent scalars and cos/sin tables at the head of compute_legendre_polynomials_lat
    // (LegendrePolynomials.cc:58-83): vsin[j*stride], vcos[j*stride] for j = 1..trc; mu = sin(lat) (1 at a pole),
    // sint = cos(lat) (0 at a pole), inv_sin_colat = 1/sint (0 at a pole)
    void colatitude_terms(double lat_rad, double* vsin, double* vcos, size_t stride, double& mu, double& sint,
                          double& inv_sin_colat) const;
    // P(m,m) for m = 2..trc from P(1,1) (:122-130), written to diag[m*stride]
    void diagonal(double p11, double sint, double inv_sin_colat, double* diag, size_t stride) const;
    const std::vector<double>& zfn() const { return zfn_; }
    const std::vector<double>& sq1() const { return sq1_; }
    const std::vector<double>& ca() const { return ca_; }
    const std::vector<double>& cb() const { return cb_; }
    const std::vector<double>& cc() const { return cc_; }

private:
    int trc_;
    size_t tri_;
    std::vector<double> zfn_;            // (trc+1)^2, odd-n rows have zfn(n,0)=0 (LegendrePolynomials.cc:102)
    std::vector<double> ca_, cb_, cc_;   // recurrence coefficients per packed (m,n)
    std::vector<double> diag_;           // sqrt((2n+1)/(2n))
    std::vector<double> sq1_;            // 1/sqrt(n(n+1))
};

// Reference (cache-file) layout: block m at begin[m], column-major K x nlatsLeg, n descending.
// (LegendrePolynomials.cc:154-209, TransLocal.cc:592-637)
void compute_legendre_tables_reference_layout(const TransGeometry& geo, double* leg_sym, double* leg_asym);

// Tile-blocked device layout (see trans_plan.h / DESIGN.md): item block = [parity][kpad][LEG_BN].
void compute_legendre_table_tiled(const TransGeometry& geo, const LegendreWork& work, double* table);

// Inputs of the device generation of the tiled table (legendre_gen_core.h): everything with sqrt / division / sin /
// cos, prepared on the host (O(T^2)); params() points at these host arrays.
struct LegendreGenParams;
struct LegendreGenInputs {
    int trc = 0, T = 0, nlats = 0, lat_pitch = 0;
    std::vector<double> zfn, sq1, ca, cb, cc, vcos, vsin, diag, mu;
    std::vector<int> mstop, nlat0, first_item_of_m, item_kpad;
    std::vector<long long> item_p_off;
    size_t col01_doubles() const { return size_t(2) * size_t(trc + 1) * size_t(lat_pitch); }
    size_t rows_doubles() const { return size_t(4) * size_t(trc + 1) * size_t(lat_pitch); }
};
LegendreGenInputs prepare_legendre_gen(const TransGeometry& geo, const LegendreWork& work);
// host run of the device code (legendre_gen_core.h): must reproduce compute_legendre_table_tiled bit for bit
void compute_legendre_table_tiled_emulated(const TransGeometry& geo, const LegendreWork& work, double* table);

// reference layout -> tiled layout (used when importing a Legendre cache blob)
void retile_legendre_tables(const TransGeometry& geo, const LegendreWork& work, const double* leg_sym,
                            const double* leg_asym, double* table);

}  // namespace trans
}  // namespace atlas_amd
