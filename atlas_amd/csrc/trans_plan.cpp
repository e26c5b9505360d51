// See trans_plan.h.  Host-only.
#include "trans_plan.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <numeric>
#include <stdexcept>
#include <string>

namespace atlas_amd {
namespace trans {

// TransLocal.cc:272-300.  `nxmax` is unused there as well.
int fourier_truncation(int truncation, int nx, int /*nxmax*/, int ndgl, double lat_rad, bool fullgrid) {
    int trc           = truncation;
    const int trclin  = ndgl - 1;
    const int trcquad = ndgl * 2 / 3 - 1;
    if (truncation >= trclin || fullgrid) {
        trc = (nx - 1) / 2;
    }
    else if (truncation >= trcquad) {
        // the reference computes the weight with INTEGER division (:287)
        const double weight = 3 * (trclin - truncation) / ndgl;
        const double sqcos  = std::pow(std::cos(lat_rad), 2);
        trc                 = static_cast<int>((nx - 1) / (2 + weight * sqcos));
    }
    else {
        const double sqcos = std::pow(std::cos(lat_rad), 2);
        trc                = static_cast<int>((nx - 1) / (2 + sqcos) - 1);
    }
    return std::min(truncation, trc);
}

static size_t pad8(size_t n) {  // add_padding, TransLocal.cc:236-238
    return size_t(std::ceil(n / 8.)) * 8;
}

TransGeometry make_geometry(const grid::StructuredGrid& g, int truncation, int ndgl, int nxmax_override) {
    if (truncation < 0) {
        throw std::invalid_argument("truncation must be >= 0");
    }
    const int nlats = g.ny();
    if (nlats < 2) {
        throw std::invalid_argument("grid needs at least 2 latitudes");
    }
    for (int j = 1; j < nlats; ++j) {
        if (!(g.y[j] < g.y[j - 1])) {
            throw std::invalid_argument("latitudes must be monotone decreasing (north to south)");  // :373
        }
    }
    TransGeometry geo;
    geo.T       = truncation;
    geo.nlats   = nlats;
    geo.regular = g.regular;
    geo.nx      = g.nx;
    geo.lat_deg = g.y;
    geo.nxmax   = nxmax_override > 0 ? nxmax_override : g.nxmax();
    if (ndgl <= 0) {
        ndgl = nlats;
    }
    if (ndgl < nlats || geo.nxmax < g.nxmax()) {
        throw std::invalid_argument("make_geometry: ndgl / nxmax of the enclosing grid are smaller than the grid's own");
    }
    geo.npts    = g.size();
    geo.rowoff.resize(nlats + 1);
    geo.rowoff[0] = 0;
    for (int j = 0; j < nlats; ++j) {
        if (g.nx[j] <= 0) {
            throw std::invalid_argument("nx must be positive");
        }
        geo.rowoff[j + 1] = geo.rowoff[j] + g.nx[j];
    }
    // hemispheres (:371-392)
    int neqtr = 0;
    for (int j = 0; j < nlats; ++j) {
        const double lat = g.y[j];
        if (std::abs(lat) <= std::numeric_limits<double>::epsilon()) {
            neqtr++;
        }
        else if (lat < 0) {
            geo.nlatsSH++;
        }
        else {
            geo.nlatsNH++;
        }
    }
    if (neqtr > 0) {
        geo.nlatsNH++;
        geo.nlatsSH++;
        geo.has_equator = true;
    }
    if (neqtr > 1 || geo.nlatsNH != geo.nlatsSH) {
        throw std::invalid_argument(
            "only global grids that are symmetric about the equator are supported (cropped domains: SURVEY 8f4)");
    }
    for (int j = 0; j < nlats / 2; ++j) {
        if (g.nx[j] != g.nx[nlats - 1 - j] || std::abs(g.y[j] + g.y[nlats - 1 - j]) > 1e-9) {
            throw std::invalid_argument("grid is not symmetric about the equator");
        }
    }
    const int nlatsLegDomain = std::max(geo.nlatsNH, geo.nlatsSH);
    geo.nlatsLeg             = (nlats + 1) / 2;     // :435
    const int jlatMinLeg     = 0;                   // global grid: jlatMin_ = 0, NH >= SH  (:441-457)
    geo.nlatsLegR            = jlatMinLeg + nlatsLegDomain;  // :459

    // nlat0 (:462-488)
    geo.nlat0.assign(truncation + 1, 0);
    int nmen0 = -1;
    for (int jlat = 0; jlat < nlats / 2; ++jlat) {
        const double lat = g.y[jlat] * (M_PI / 180.);
        int nmen         = fourier_truncation(truncation, g.nx[jlat], geo.nxmax, ndgl, lat, g.regular);
        nmen             = std::max(nmen0, nmen);
        const int ndgluj = std::max(jlatMinLeg, jlat);
        for (int j = nmen0 + 1; j <= nmen; ++j) {
            geo.nlat0[j] = ndgluj;
        }
        nmen0 = nmen;
    }
    for (int j = nmen0 + 1; j <= truncation; ++j) {
        geo.nlat0[j] = geo.nlatsLeg;
    }
    // Legendre latitudes (:533-545)
    geo.lats_leg.resize(geo.nlatsLeg);
    for (int j = 0; j < geo.nlatsLeg; ++j) {
        double lat = g.y[j];
        lat        = std::min(lat, kLatPole);
        lat        = std::max(lat, -kLatPole);
        geo.lats_leg[j] = lat * (M_PI / 180.);
    }
    // table offsets (:592-606): loop to T+1, truncation T+1
    geo.begin_sym.assign(truncation + 3, 0);
    geo.begin_asym.assign(truncation + 3, 0);
    size_t ss = 0, sa = 0;
    for (int m = 0; m <= truncation + 1; ++m) {
        ss += pad8(size_t(num_n(truncation + 1, m, true)) * size_t(geo.nlatsLeg));
        sa += pad8(size_t(num_n(truncation + 1, m, false)) * size_t(geo.nlatsLeg));
        geo.begin_sym[m + 1]  = ss;
        geo.begin_asym[m + 1] = sa;
    }
    // highest kept wavenumber per Legendre row (nlat0 is monotone non-decreasing in m)
    geo.mmax_leg.assign(geo.nlatsLeg, -1);
    for (int j = 0; j < geo.nlatsLeg; ++j) {
        int mm = -1;
        for (int m = 0; m <= truncation; ++m) {
            if (geo.nlat0[m] <= j && j < geo.nlatsLegR) {
                mm = m;
            }
        }
        geo.mmax_leg[j] = mm;
    }
    return geo;
}

double legendre_flops(const TransGeometry& geo, int nf) {
    // F_leg = sum_m 2 * (nf*n_imag(m)) * (K_s(m)+K_a(m)) * L(m)
    double f = 0;
    for (int m = 0; m <= geo.T; ++m) {
        const int L = geo.L(m);
        if (L <= 0) {
            continue;
        }
        const double k = num_n(geo.T + 1, m, true) + num_n(geo.T + 1, m, false);
        f += 2.0 * nf * (m ? 2 : 1) * k * L;
    }
    return f;
}

LegendreWork make_legendre_work(const TransGeometry& geo, int nparts, int part, bool by_band, int row_begin,
                                int row_end) {
    LegendreWork w;
    const int T = geo.T;
    w.first_item_of_m.assign(T + 2, 0);
    // by_band: this device computes every wavenumber, but only the Legendre rows whose northern or mirrored southern
    // latitude lies in its own latitude band [b0, b1) (no hemisphere sharing between devices, no exchange)
    int b0 = 0, b1 = geo.nlats;
    if (by_band) {
        if (row_end > row_begin) {  // explicit row range (zonal-band crop of the grid)
            b0 = row_begin;
            b1 = row_end;
        }
        else {
            const std::vector<int> bands = latitude_bands(geo, nparts);
            b0                           = bands[part];
            b1                           = bands[part + 1];
        }
    }
    auto row_needed = [&](int jn) { return (jn >= b0 && jn < b1) || (geo.nlats - 1 - jn >= b0 && geo.nlats - 1 - jn < b1); };
    int64_t off = 0;
    for (int m = 0; m <= T; ++m) {
        w.first_item_of_m[m] = (int)w.items_by_m.size();
        const int L          = geo.L(m);
        if (L <= 0 || (!by_band && (m % nparts) != part)) {
            continue;
        }
        const int ks    = num_n(T + 1, m, true);
        const int kpad  = (ks + LEG_KB - 1) / LEG_KB * LEG_KB;
        const int tiles = (L + LEG_BN - 1) / LEG_BN;
        for (int t = 0; t < tiles; ++t) {
            LegendreItem it;
            it.m     = m;
            it.tile  = t;
            it.nrows = std::min(LEG_BN, L - t * LEG_BN);
            it.kpad  = kpad;
            bool needed = !by_band;
            for (int c = 0; c < it.nrows && !needed; ++c) {
                needed = row_needed(geo.nlat0[m] + t * LEG_BN + c);
            }
            if (needed) {
                it.p_off = off;
                off += int64_t(2) * kpad * LEG_BN;
            }
            else {
                it.p_off = -1;  // tile of another band: kept in items_by_m (tile index = position), never launched
            }
            w.items_by_m.push_back(it);
        }
    }
    w.first_item_of_m[T + 1] = (int)w.items_by_m.size();
    w.table_doubles          = off;

    // Launch order.  Hardware places workgroup b on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch");
    // all tiles of one m share the same spectral operand, so an m is pinned to one XCD (its L2 then serves the
    // operand to every tile) and the m's are distributed over the 8 XCD lists by longest-processing-time-first.
    // Within a list heavy items come first.  Placement is a speed heuristic only; results do not depend on it.
    constexpr int NXCD = 8;
    std::vector<int> ms;
    std::vector<double> cost(T + 1, 0.);
    for (int m = 0; m <= T; ++m) {
        int n = 0;
        for (int i = w.first_item_of_m[m]; i < w.first_item_of_m[m + 1]; ++i) {
            n += w.items_by_m[i].p_off >= 0;
        }
        if (n > 0) {
            ms.push_back(m);
            cost[m] = double(n) * w.items_by_m[w.first_item_of_m[m]].kpad;
        }
    }
    std::sort(ms.begin(), ms.end(), [&](int a, int b) { return cost[a] > cost[b] || (cost[a] == cost[b] && a < b); });
    std::vector<std::vector<LegendreItem>> lists(NXCD);
    std::vector<double> load(NXCD, 0.);
    for (int m : ms) {
        int x = int(std::min_element(load.begin(), load.end()) - load.begin());
        load[x] += cost[m];
        for (int i = w.first_item_of_m[m]; i < w.first_item_of_m[m + 1]; ++i) {
            if (w.items_by_m[i].p_off >= 0) {
                lists[x].push_back(w.items_by_m[i]);
            }
        }
    }
    size_t maxlen = 0;
    for (auto& l : lists) {
        maxlen = std::max(maxlen, l.size());
    }
    LegendreItem null_item{-1, 0, 0, 0, 0};
    w.items.reserve(maxlen * NXCD);
    for (size_t q = 0; q < maxlen; ++q) {
        for (int x = 0; x < NXCD; ++x) {
            w.items.push_back(q < lists[x].size() ? lists[x][q] : null_item);
        }
    }
    // drop trailing null items
    while (!w.items.empty() && w.items.back().m < 0) {
        w.items.pop_back();
    }
    return w;
}

std::vector<int> mirror_bands(const grid::StructuredGrid& g, int nparts) {
    const int ny = g.ny();
    if (nparts < 1 || ny < 2 || ny % 2 != 0) {
        throw std::invalid_argument("mirror_bands: needs nparts >= 1 and an even number of latitudes");
    }
    const int half = ny / 2;
    int64_t npts   = 0;
    for (int j = 0; j < half; ++j) {
        npts += g.nx[j];
    }
    std::vector<int> b(nparts + 1, half);
    b[0]        = 0;
    int prev    = 0;
    int64_t off = 0;
    for (int j = 0; j < half; ++j) {
        const int part = int((off * nparts) / npts);  // part of the row's first point (cf. latitude_bands)
        for (int q = prev + 1; q <= part; ++q) {
            b[q] = j;
        }
        prev = std::max(prev, part);
        off += g.nx[j];
    }
    return b;
}

grid::StructuredGrid polar_caps_grid(const grid::StructuredGrid& g, int b1) {
    const int ny = g.ny();
    if (b1 < 1 || 2 * b1 > ny) {
        throw std::invalid_argument("polar_caps_grid: bad row count");
    }
    grid::StructuredGrid v;
    v.name    = g.name + "[caps " + std::to_string(b1) + "]";
    v.N       = 0;
    v.regular = g.regular;
    v.nx.assign(g.nx.begin(), g.nx.begin() + b1);
    v.nx.insert(v.nx.end(), g.nx.end() - b1, g.nx.end());
    v.y.assign(g.y.begin(), g.y.begin() + b1);
    v.y.insert(v.y.end(), g.y.end() - b1, g.y.end());
    return v;
}

std::vector<int> latitude_bands(const TransGeometry& geo, int nparts) {
    std::vector<int> b(nparts + 1, geo.nlats);
    b[0]                 = 0;
    const int64_t npts   = geo.npts;  // blocksize 1 -> nb_blocks = npts
    int prev             = 0;
    for (int j = 0; j < geo.nlats; ++j) {
        const int64_t g = geo.rowoff[j];
        const int part  = int((g * nparts) / npts);
        for (int q = prev + 1; q <= part; ++q) {
            b[q] = j;
        }
        prev = std::max(prev, part);
    }
    for (int q = prev + 1; q <= nparts; ++q) {
        b[q] = geo.nlats;
    }
    return b;
}

}  // namespace trans
}  // namespace atlas_amd
