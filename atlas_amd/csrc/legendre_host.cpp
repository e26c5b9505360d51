// See legendre_host.h.  Host-only; compiled with -ffp-contract=off so that every operation rounds exactly once,
// in the order the reference writes it.
#include "legendre_host.h"

#include "legendre_gen_core.h"
#include "legendre_series.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace atlas_amd {
namespace trans {

LegendreEvaluator::LegendreEvaluator(int trc): trc_(trc), tri_(size_t(trc + 2) * size_t(trc + 1) / 2) {
    const size_t ld = size_t(trc) + 1;
    // series coefficients of every degree (legendre_series.h; LegendrePolynomials.cc:24-45)
    zfn_.assign(ld * ld, 0.);
    double lead = 2.;
    zfn_[0]     = lead;
    for (int n = 1; n <= trc; ++n) {
        lead = legendre_series_lead(lead, n);
        legendre_series_row(n, lead, &zfn_[size_t(n) * ld]);
    }
    // odd degrees have no constant term; the reference clears it while evaluating the first latitude (:102)
    for (int n = 1; n <= trc; n += 2) {
        zfn_[size_t(n) * ld] = 0.;
    }
    sq1_.assign(ld, 0.);
    diag_.assign(ld, 0.);
    for (int n = 1; n <= trc; ++n) {
        sq1_[n]  = 1. / std::sqrt(n * (n + 1.));
        diag_[n] = std::sqrt((2. * n + 1.) / (2. * n));
    }
    // the three coefficients of the recurrence (m-2, n-2), (m-2, n-1), (m, n-1) -> (m, n) (:136-149), one triple per
    // (m, n), latitude independent.  Numerators and denominators are products of small integers (exact in double), so
    // each coefficient is one division and one square root.
    ca_.assign(tri_, 0.);
    cb_.assign(tri_, 0.);
    cc_.assign(tri_, 0.);
    for (int n = 3; n <= trc; ++n) {
        const double up = 2. * n + 1., mid = 2. * n - 1., low = 2. * n - 3.;
        for (int m = 2; m < n; ++m) {
            const double sum = n + m, dif = n - m;
            const size_t i   = idxmn(trc, m, n);
            ca_[i]           = std::sqrt((up * (sum - 3.) * (sum - 1.)) / (low * (sum - 2.) * sum));
            cb_[i]           = std::sqrt((up * (dif + 1.) * (sum - 1.)) / (mid * (sum - 2.) * sum));
            cc_[i]           = std::sqrt((up * dif) / (mid * sum));
        }
    }
}

void LegendreEvaluator::colatitude_terms(double lat, double* vsin, double* vcos, size_t stride, double& mu_out,
                                         double& sin_colat_out, double& inv_sin_colat_out) const {
    const double colat        = (M_PI_2 - lat);
    double mu                 = std::cos(colat);                 // sin(latitude) as the reference obtains it (:60)
    volatile double sin_colat = std::sqrt(1. - mu * mu);         // volatile as in the reference (:61): no re-association
    for (int j = 1; j <= trc_; j++) {
        vsin[size_t(j) * stride] = std::sin(j * colat);
        vcos[size_t(j) * stride] = std::cos(j * colat);
    }
    double inv = 0.;
    if (std::abs(sin_colat) <= std::sqrt(std::numeric_limits<double>::epsilon())) {   // at a pole (:64-72)
        mu        = 1.;
        sin_colat = 0.;
    }
    else {
        inv = 1. / sin_colat;
    }
    mu_out            = mu;
    sin_colat_out     = sin_colat;
    inv_sin_colat_out = inv;
}

void LegendreEvaluator::diagonal(double p11, double sin_colat, double inv_sin_colat, double* diag, size_t stride) const {
    // P(m, m) = P(m-1, m-1) * sin(colat) * sqrt((2m+1)/2m), flushed to zero below the underflow guard (:122-130)
    const double guard = inv_sin_colat * std::numeric_limits<double>::min();
    double prev        = p11;
    for (int m = 2; m <= trc_; ++m) {
        double v = prev * sin_colat * diag_[m];
        if (std::abs(v) < guard) {
            v = 0.0;
        }
        diag[size_t(m) * stride] = v;
        prev                     = v;
    }
}

void LegendreEvaluator::evaluate(double lat, double* legpol, double* scratch) const {
    const int trc   = trc_;
    const size_t ld = size_t(trc) + 1;
    double* vsin    = scratch;
    double* vcos    = scratch + ld;
    // 1. columns m = 0 and m = 1: cosine series of P_n and its derivative (:85-115), shared with the device generator
    double mu, sin_colat, inv_sin_colat;
    colatitude_terms(lat, vsin, vcos, 1, mu, sin_colat, inv_sin_colat);
    legpol[idxmn(trc, 0, 0)] = 1.;
    for (int n = 1; n <= trc; ++n) {
        double p0, p1;
        legendre_series_point(&zfn_[size_t(n) * ld], n, sq1_[n], vcos, vsin, 1, p0, p1);
        legpol[idxmn(trc, 0, n)] = p0;
        legpol[idxmn(trc, 1, n)] = p1;
    }
    // 2. diagonal
    {
        const double guard = inv_sin_colat * std::numeric_limits<double>::min();
        for (int m = 2; m <= trc; ++m) {
            double v = legpol[idxmn(trc, m - 1, m - 1)] * sin_colat * diag_[m];
            if (std::abs(v) < guard) {
                v = 0.0;
            }
            legpol[idxmn(trc, m, m)] = v;
        }
    }
    // 3. everything else.  The reference walks n outer / m inner; the dependencies (m-2,n-2), (m-2,n-1), (m,n-1) allow
    //    m outer / n inner, which is contiguous in the packed triangle and applies the identical operations to the
    //    identical operands.
    for (int m = 2; m < trc; ++m) {
        const size_t row   = idxmn(trc, m, m);
        double* p          = legpol + row;                        // p[n - m]       = P(m, n)
        const double* q    = legpol + idxmn(trc, m - 2, m - 2);   // q[n - (m - 2)] = P(m - 2, n)
        const double* a    = ca_.data() + row;
        const double* b    = cb_.data() + row;
        const double* c    = cc_.data() + row;
        for (int n = std::max(3, m + 1); n <= trc; ++n) {
            const int i = n - m;
            p[i]        = a[i] * q[i] - b[i] * q[i + 1] * mu + c[i] * p[i - 1] * mu;
        }
    }
}

void compute_legendre_tables_reference_layout(const TransGeometry& geo, double* leg_sym, double* leg_asym) {
    const int trc   = geo.T + 1;
    const int nlats = geo.nlatsLeg;
    LegendreEvaluator ev(trc);
#pragma omp parallel
    {
        std::vector<double> legpol(ev.triangle_size());
        std::vector<double> scratch(2 * size_t(trc + 1));
#pragma omp for schedule(dynamic, 1)
        for (int jlat = 0; jlat < nlats; ++jlat) {
            ev.evaluate(geo.lats_leg[jlat], legpol.data(), scratch.data());
            for (int jm = 0; jm <= trc; ++jm) {
                const size_t is1 = num_n(trc, jm, true), ia1 = num_n(trc, jm, false);
                size_t is2 = 0, ia2 = 0;
                for (int jn = trc; jn >= jm; --jn) {  // n descending (:195-205)
                    const double v = legpol[LegendreEvaluator::idxmn(trc, jm, jn)];
                    if ((jn - jm) % 2 == 0) {
                        leg_sym[geo.begin_sym[jm] + is1 * size_t(jlat) + is2++] = v;
                    }
                    else {
                        leg_asym[geo.begin_asym[jm] + ia1 * size_t(jlat) + ia2++] = v;
                    }
                }
            }
        }
    }
}

namespace {
// destination of P(m, n) for Legendre row jlat in the tiled table
inline void tiled_store(const TransGeometry& geo, const LegendreWork& work, double* table, int m, int jlat0, int nl,
                        const double* const* legpols) {
    // rows jlat0 .. jlat0+nl-1 (all >= nlat0[m])
    const int trc  = geo.T + 1;
    const int c0   = jlat0 - geo.nlat0[m];
    const int ntop[2] = {trc - ((trc - m) & 1), trc - 1 + ((trc - m) & 1)};  // largest n<=trc with (n-m)%2 == p
    for (int l = 0; l < nl; ++l) {
        const int c          = c0 + l;
        const LegendreItem& it = work.items_by_m[work.first_item_of_m[m] + c / LEG_BN];
        if (it.p_off < 0) {
            continue;  // tile of another latitude band
        }
        const int col        = c % LEG_BN;
        double* blk          = table + it.p_off + col;
        const double* lp     = legpols[l] + LegendreEvaluator::idxmn(trc, m, m);  // lp[n-m]
        for (int p = 0; p < 2; ++p) {
            double* dst = blk + size_t(p) * it.kpad * LEG_BN;
            int k       = 0;
            for (int n = ntop[p]; n >= m; n -= 2, ++k) {
                dst[size_t(k) * LEG_BN] = lp[n - m];
            }
        }
    }
}
}  // namespace

void compute_legendre_table_tiled(const TransGeometry& geo, const LegendreWork& work, double* table) {
    const int trc   = geo.T + 1;
    const int nlats = geo.nlatsLegR;
    LegendreEvaluator ev(trc);
    constexpr int LB = 8;  // latitudes per block: keeps the strided scatter within one cache line per (m,n)
    const int nblk   = (nlats + LB - 1) / LB;
    // NB: `table` must be zero-initialised by the caller: padded k rows / latitude columns stay zero
#pragma omp parallel
    {
        std::vector<std::vector<double>> legpol(LB, std::vector<double>(ev.triangle_size()));
        std::vector<double> scratch(2 * size_t(trc + 1));
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < nblk; ++b) {
            const int j0 = b * LB;
            const int nl = std::min(LB, nlats - j0);
            bool wanted  = false;  // does any launched tile contain one of these rows?
            for (int m = 0; m <= geo.T && !wanted; ++m) {
                for (int l = std::max(j0, geo.nlat0[m]); l < j0 + nl && !wanted; ++l) {
                    const int i = work.first_item_of_m[m] + (l - geo.nlat0[m]) / LEG_BN;
                    wanted      = i < work.first_item_of_m[m + 1] && work.items_by_m[i].p_off >= 0;
                }
            }
            if (!wanted) {
                continue;
            }
            for (int l = 0; l < nl; ++l) {
                ev.evaluate(geo.lats_leg[j0 + l], legpol[l].data(), scratch.data());
            }
            for (int m = 0; m <= geo.T; ++m) {
                if (work.first_item_of_m[m + 1] == work.first_item_of_m[m]) {
                    continue;  // no rows for this m, or m owned by another device
                }
                // sub-range of this block that lies at or south of nlat0[m]
                const int lo = std::max(j0, geo.nlat0[m]);
                if (lo >= j0 + nl) {
                    continue;
                }
                const double* ptrs[LB];
                for (int l = lo; l < j0 + nl; ++l) {
                    ptrs[l - lo] = legpol[l - j0].data();
                }
                tiled_store(geo, work, table, m, lo, j0 + nl - lo, ptrs);
            }
        }
    }
}

LegendreGenInputs prepare_legendre_gen(const TransGeometry& geo, const LegendreWork& work) {
    LegendreGenInputs in;
    in.trc       = geo.T + 1;
    in.T         = geo.T;
    in.nlats     = geo.nlatsLegR;
    in.lat_pitch = (std::max(in.nlats, 1) + LG_TILE - 1) / LG_TILE * LG_TILE;
    const LegendreEvaluator ev(in.trc);
    in.zfn = ev.zfn();
    in.sq1 = ev.sq1();
    in.ca  = ev.ca();
    in.cb  = ev.cb();
    in.cc  = ev.cc();
    const size_t pitch = size_t(in.lat_pitch), ld = size_t(in.trc) + 1;
    in.vcos.assign(ld * pitch, 0.);
    in.vsin.assign(ld * pitch, 0.);
    in.diag.assign(ld * pitch, 0.);
    in.mu.assign(pitch, 0.);
    in.mstop.assign(pitch, -1);
#pragma omp parallel for schedule(static)
    for (int jlat = 0; jlat < in.nlats; ++jlat) {
        double mu, sint, inv_sin_colat;
        ev.colatitude_terms(geo.lats_leg[jlat], in.vsin.data() + jlat, in.vcos.data() + jlat, pitch, mu, sint,
                            inv_sin_colat);
        in.mu[jlat] = mu;
        // P(1,1): the n = 1 term of the series (the kernel computes the same value into its column)
        double p01, p11;
        legendre_series_point(in.zfn.data() + ld, 1, in.sq1[1], in.vcos.data() + jlat, in.vsin.data() + jlat, pitch, p01,
                              p11);
        in.diag[pitch + jlat] = p11;
        ev.diagonal(p11, sint, inv_sin_colat, in.diag.data() + jlat, pitch);
        in.mstop[jlat] = geo.mmax_leg[jlat] < geo.T ? geo.mmax_leg[jlat] : geo.T;
    }
    in.nlat0           = geo.nlat0;
    in.first_item_of_m = work.first_item_of_m;
    in.item_p_off.resize(work.items_by_m.size());
    in.item_kpad.resize(work.items_by_m.size());
    for (size_t i = 0; i < work.items_by_m.size(); ++i) {
        in.item_p_off[i] = work.items_by_m[i].p_off;
        in.item_kpad[i]  = work.items_by_m[i].kpad;
    }
    return in;
}

void compute_legendre_table_tiled_emulated(const TransGeometry& geo, const LegendreWork& work, double* table) {
    const LegendreGenInputs in = prepare_legendre_gen(geo, work);
    std::vector<double> col01(in.col01_doubles(), 0.), rows(in.rows_doubles(), 0.);
    LegendreGenParams g;
    g.trc = in.trc, g.T = in.T, g.nlats = in.nlats, g.lat_pitch = in.lat_pitch;
    g.zfn = in.zfn.data(), g.sq1 = in.sq1.data(), g.ca = in.ca.data(), g.cb = in.cb.data(), g.cc = in.cc.data();
    g.vcos = in.vcos.data(), g.vsin = in.vsin.data(), g.diag = in.diag.data(), g.mu = in.mu.data();
    g.mstop = in.mstop.data();
    g.col01 = col01.data(), g.rows = rows.data(), g.table = table;
    g.nlat0 = in.nlat0.data(), g.first_item_of_m = in.first_item_of_m.data();
    g.item_p_off = in.item_p_off.data(), g.item_kpad = in.item_kpad.data();
    // the two kernels of legendre_gen_kernel.hip, one loop iteration per device thread
#pragma omp parallel for schedule(dynamic, 1)
    for (int jn = 0; jn <= in.trc; ++jn) {
        for (int lat = 0; lat < in.nlats; ++lat) {
            legendre_series_store(g, lat, jn);
        }
    }
#pragma omp parallel for schedule(dynamic, 1) collapse(2)
    for (int lat = 0; lat < in.nlats; ++lat) {
        for (int parity = 0; parity < 2; ++parity) {
            legendre_chain(g, lat, parity);
        }
    }
}

void retile_legendre_tables(const TransGeometry& geo, const LegendreWork& work, const double* leg_sym,
                            const double* leg_asym, double* table) {
    const int trc = geo.T + 1;
#pragma omp parallel for schedule(dynamic, 1)
    for (int m = 0; m <= geo.T; ++m) {
        if (work.first_item_of_m[m + 1] == work.first_item_of_m[m]) {
            continue;
        }
        const size_t ks = num_n(trc, m, true), ka = num_n(trc, m, false);
        for (int jlat = geo.nlat0[m]; jlat < geo.nlatsLegR; ++jlat) {
            const int c            = jlat - geo.nlat0[m];
            const LegendreItem& it = work.items_by_m[work.first_item_of_m[m] + c / LEG_BN];
            if (it.p_off < 0) {
                continue;
            }
            double* blk            = table + it.p_off + (c % LEG_BN);
            const double* s        = leg_sym + geo.begin_sym[m] + ks * size_t(jlat);
            const double* a        = leg_asym + geo.begin_asym[m] + ka * size_t(jlat);
            for (size_t k = 0; k < ks; ++k) {
                blk[k * LEG_BN] = s[k];
            }
            for (size_t k = 0; k < ka; ++k) {
                blk[(size_t(it.kpad) + k) * LEG_BN] = a[k];
            }
        }
    }
}

}  // namespace trans
}  // namespace atlas_amd
