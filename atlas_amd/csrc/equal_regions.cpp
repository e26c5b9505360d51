// Atlas's default partitioner for structured grids, "equal_regions": Leopardi's zonal equal-area partition of the sphere
// into N regions (two polar caps and collars of several sectors), applied to the grid points in their global order.
// Reference: src/atlas/grid/detail/partitioner/EqualRegionsPartitioner.cc:70-347 (zones), :443-605,614-700 (points);
// known answers: src/tests/mesh/test_rgg.cc:103-165, src/tests/functionspace/test_structuredcolumns.cc:87-106.
// Host only.  The result -- one part number per grid point -- is the explicit grid::Distribution that
// atlas_amd__StructuredColumns__new_distribution accepts.
#include "equal_regions.h"

#include <algorithm>
#include <cmath>
#include <numeric>
#include <stdexcept>

namespace atlas_amd {
namespace grid {

namespace {
const double kPi = 3.14159265358979323846264338327950288;

// Gamma(x) by the 14-term series of 1/Gamma around 2 that the reference carries (its area formulas need Gamma(1.5) only,
// but the value must be the reference's to the last bit: region counts are rounded from it)
double gamma_series(double x) {
    static const double c[14] = {0.999999999999999990e+00,  -0.422784335098466784e+00, -0.233093736421782878e+00,
                                 0.191091101387638410e+00,  -0.024552490005641278e+00, -0.017645244547851414e+00,
                                 0.008023273027855346e+00,  -0.000804329819255744e+00, -0.000360837876648255e+00,
                                 0.000145596568617526e+00,  -0.000017545539395205e+00, -0.000002591225267689e+00,
                                 0.000001337767384067e+00,  -0.000000199542863674e+00};
    const int shift  = (int)std::round(x - 2.);
    const double frac = x - (shift + 2);
    double inv = c[13];
    for (int i = 12; i >= 0; --i) {
        inv = inv * frac + c[i];
    }
    double scale = 1.;
    if (shift > 0) {
        scale = x - 1.;
        for (int k = 2; k <= shift; ++k) {
            scale *= x - k;
        }
    }
    else {
        for (int k = 0; k < -shift; ++k) {
            inv *= x + k;
        }
    }
    return scale / inv;
}

double cap_area(double colat) {
    // 4 pi * (sin^2), the square formed FIRST as in the reference's 4.0 * M_PI * std::pow(std::sin(0.5 * s_cap), 2)
    // (EqualRegionsPartitioner.cc:125): the zones' shares of the regions are rounded from differences of these areas, and where a
    // share is x.5 up to the last bit (N = 9, 31, ...) the other association, (4 pi s) s, moved a region to the neighbouring zone
    const double s = std::sin(0.5 * colat);
    return 4. * kPi * (s * s);
}
double cap_colat(double area) {
    return 2. * std::asin(0.5 * std::sqrt(area / kPi));
}
double region_area(int N) {
    return 2. * std::pow(kPi, 1.5) / gamma_series(1.5) / (double)N;
}
}  // namespace

void eq_caps(int N, std::vector<int>& regions, std::vector<double>& colats) {
    if (N < 1) {
        throw std::invalid_argument("eq_caps: N >= 1");
    }
    regions.clear();
    colats.clear();
    if (N == 1) {
        regions.push_back(1);
        colats.push_back(kPi);
        return;
    }
    const double ideal = region_area(N);
    const double polar = N == 2 ? 0.5 * kPi : cap_colat(ideal);
    const double side  = std::sqrt(ideal);
    int ncollars       = 0;
    if (N > 2 && side > 0.) {
        ncollars = std::max(1, (int)std::round((kPi - 2. * polar) / side));
    }
    std::vector<double> share(ncollars + 2, 1.);   // regions each zone would hold, before rounding
    if (ncollars > 0) {
        const double width = (kPi - 2. * polar) / (double)ncollars;
        for (int z = 0; z < ncollars; ++z) {
            share[1 + z] = (cap_area(polar + (z + 1) * width) - cap_area(polar + z * width)) / ideal;
        }
    }
    double carry = 0.;   // rounding with the remainder carried to the next zone
    for (double s : share) {
        const int n = (int)std::round(s + carry);
        regions.push_back(n);
        carry += s - n;
    }
    colats.push_back(polar);
    int above = 1;
    for (int z = 0; z < ncollars; ++z) {
        above += regions[1 + z];
        colats.push_back(cap_colat(above * ideal));
    }
    colats.push_back(kPi);
}

static long long microdegrees(double deg) {   // util/MicroDeg.h:18-22
    return deg < 0. ? (long long)(deg * 1.e6 - 0.5) : (long long)(deg * 1.e6 + 0.5);
}

std::vector<int> equal_regions_partition(const StructuredGrid& g, int N) {
    std::vector<int> regions;
    std::vector<double> colats;
    eq_caps(N, regions, colats);
    const long long npts = g.size();
    std::vector<int> part((size_t)npts, 0);
    if (N == 1) {
        return part;
    }
    // the points in global order (north -> south, west -> east) with integer micro-degree coordinates
    std::vector<long long> xs((size_t)npts), ys((size_t)npts);
    {
        size_t i = 0;
        for (int j = 0; j < g.ny(); ++j) {
            const int n        = g.nx[j];
            const long long yj = microdegrees(g.y[j]);
            for (int k = 0; k < n; ++k, ++i) {
                xs[i] = microdegrees(k * (360. / n));
                ys[i] = yj;
            }
        }
    }
    std::vector<long long> order((size_t)npts);
    std::iota(order.begin(), order.end(), 0LL);
    const long long chunk = npts / N;
    long long extra       = npts % N;
    std::vector<long long> counts;
    long long end = 0;
    for (size_t zone = 0; zone < regions.size(); ++zone) {
        const long long begin = end;
        for (int s = 0; s < regions[zone]; ++s) {
            counts.push_back(chunk + (extra > 0 ? 1 : 0));
            --extra;
            end += counts.back();
        }
        // the zone's points, west -> east, then north -> south
        std::stable_sort(order.begin() + begin, order.begin() + end, [&](long long a, long long b) {
            return xs[a] < xs[b] || (xs[a] == xs[b] && ys[a] > ys[b]);
        });
    }
    long long pos = 0;
    for (size_t p = 0; p < counts.size(); ++p) {
        for (long long k = 0; k < counts[p]; ++k) {
            part[(size_t)order[pos++]] = (int)p;
        }
    }
    return part;
}

}  // namespace grid
}  // namespace atlas_amd
