// Gaussian latitudes and the global structured Gaussian grids (F / O / custom pl) that TransLocal is
// driven with.  Host-only code.
//
// Reference behaviour (ecmwf/atlas 0.44.1):
//   * src/atlas/grid/detail/spacing/gaussian/Latitudes.cc:41-57  -- for N in a fixed set the latitudes come
//     from tables printed with 12 decimals (N16/N24: 16 decimals), otherwise from a Newton iteration in
//     double precision on the Fourier series of P_2N (Latitudes.cc:100-273).
//   * src/atlas/grid/detail/grid/Gaussian.cc:86-177            -- F<N>: nx=4N, O<N>: nx=20+4j.
//   * src/atlas/grid/detail/grid/Structured.h:308-314          -- x(i,j) = xmin + i*dx, xmin=0, dx=360/nx.
//
// The tabulated values are reproduced here WITHOUT the tables: Gauss-Legendre nodes are computed in
// extended precision and rounded to the printed number of decimals; the handful of entries where the
// reference table deviates from the exactly rounded node in the last printed digit(s) are patched from
// gaussian_corrections.inc (plain data produced by tools/gen_gaussian_corrections.py, which compares
// against the reference tables in the development container).  tests/golden/gaussian_latitudes.json
// pins a SHA-256 of every tabulated N.
#include "gaussian.h"
#include "legendre_series.h"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <stdexcept>
#include <string>
#include <vector>

namespace atlas_amd {
namespace grid {

namespace {

// N values for which the reference uses tables (spacing/gaussian/N.h:78-100)
const int kTabulated[] = {16,  24,  32,  48,  64,  80,   96,   128,  160,  200,  256, 320,
                          400, 512, 576, 640, 800, 1024, 1280, 1600, 2000, 4000, 8000};

struct Correction {
    int N;
    int index;
    uint64_t bits;  // IEEE-754 bits of the reference value
};
const Correction kCorrections[] = {
#include "gaussian_corrections.inc"
    {0, 0, 0}};

int printed_decimals(int N) {
    return (N == 16 || N == 24) ? 16 : 12;
}

// Gauss-Legendre colatitudes (radians) of P_{2N}, north pole -> equator, in long double.
void exact_colatitudes(int N, std::vector<long double>& theta) {
    const int n          = 2 * N;
    const long double pi = 3.14159265358979323846264338327950288L;
    theta.resize(N);
    for (int k = 0; k < N; ++k) {
        long double z  = (4.0L * (k + 1) - 1.0L) * pi / (4.0L * n + 2.0L);
        long double th = z + 1.0L / (tanl(z) * 8.0L * (long double)n * (long double)n);
        for (int it = 0; it < 12; ++it) {
            long double x  = cosl(th);
            long double p0 = 1.0L, p1 = x;
            for (int j = 2; j <= n; ++j) {
                long double p2 = ((2 * j - 1) * x * p1 - (j - 1) * p0) / j;
                p0             = p1;
                p1             = p2;
            }
            long double dpdx = n * (x * p1 - p0) / (x * x - 1.0L);
            long double dth  = p1 / (-sinl(th) * dpdx);
            th -= dth;
            if (fabsl(dth) < 1e-19L) {
                break;
            }
        }
        theta[k] = th;
    }
}

double round_decimal(long double v, int decimals) {
    // exact decimal rounding, then a correctly rounded decimal -> double conversion (strtod)
    char buf[64];
    snprintf(buf, sizeof(buf), "%.*Lf", decimals, v);
    return strtod(buf, nullptr);
}

void tabulated_latitudes(int N, double lats[]) {
    const long double pi = 3.14159265358979323846264338327950288L;
    std::vector<long double> theta;
    exact_colatitudes(N, theta);
    const int dec = printed_decimals(N);
    for (int k = 0; k < N; ++k) {
        long double lat = 90.0L - theta[k] * 180.0L / pi;
        lats[k]         = round_decimal(lat, dec);
    }
    for (const Correction* c = kCorrections; c->N != 0; ++c) {
        if (c->N == N) {
            double v;
            std::memcpy(&v, &c->bits, sizeof(double));
            lats[c->index] = v;
        }
    }
}

// --- Newton iteration in double precision, what the reference does for a Gaussian number it has no table for ---------
// The latitudes are the roots of P_2N, found on its cosine series in the colatitude (legendre_series.h).  To return the
// reference's doubles bit for bit the first guess, the update and the stopping rule are the reference's (Latitudes.cc:
// 227-273 series and first guess, :100-135 one step, :170-210 loop: one more step after |step| <= 1000 eps, at most 21),
// and every sum runs in ascending wavenumber.
void newton_latitudes(int N, double lats[]) {
    const int degree = 2 * N;
    std::vector<double> series(degree + 1, 0.);
    {
        double lead = 2.;
        for (int j = 1; j <= degree; ++j) {
            lead = legendre_series_lead(lead, j);
        }
        legendre_series_row(degree, lead, series.data());   // even entries: s(2N, 0), s(2N, 2), ...
    }
    const double tolerance = std::numeric_limits<double>::epsilon() * 1000.;
    for (int root = 0; root < N; ++root) {
        const double guess = (4. * (root + 1.) - 1.) * M_PI / (4. * 2. * N + 2.);
        double colat       = (guess + 1. / (std::tan(guess) * (8. * (2. * N) * (2. * N))));
        bool converged     = false;
        for (int iteration = 1; iteration <= 21; ++iteration) {
            double value = 0.5 * series[0];
            double slope = 0.;
            for (int k = 2; k <= degree; k += 2) {
                const double kd = static_cast<double>(k);
                value += series[k] * std::cos(kd * colat);
                slope -= series[k] * kd * std::sin(kd * colat);
            }
            const double step = slope != 0 ? -value / slope : 0.;
            colat             = colat + step;
            if (converged) {
                break;
            }
            converged = std::abs(step) <= tolerance;
        }
        if (!converged) {
            throw std::runtime_error("Could not converge gaussian latitude");
        }
        lats[root] = 90. - colat * (180. / M_PI);
    }
}

}  // namespace

bool gaussian_latitudes_tabulated(int N) {
    return std::find(std::begin(kTabulated), std::end(kTabulated), N) != std::end(kTabulated);
}

void gaussian_latitudes_npole_equator(int N, double lats[]) {
    if (N <= 0) {
        throw std::invalid_argument("gaussian_latitudes: N must be positive");
    }
    if (gaussian_latitudes_tabulated(N)) {
        tabulated_latitudes(N, lats);
    }
    else {
        newton_latitudes(N, lats);
    }
}

void gaussian_latitudes_npole_spole(int N, double lats[]) {
    // Latitudes.cc:59-67
    gaussian_latitudes_npole_equator(N, lats);
    size_t end = 2 * (size_t)N - 1;
    for (int j = 0; j < N; ++j) {
        lats[end--] = -lats[j];
    }
}

StructuredGrid make_gaussian_grid(const std::string& name) {
    if (name.size() < 2) {
        throw std::invalid_argument("grid name too short: " + name);
    }
    char kind = name[0];
    int N     = 0;
    try {
        size_t pos = 0;
        N          = std::stoi(name.substr(1), &pos);
        if (pos != name.size() - 1) {
            N = 0;
        }
    }
    catch (...) {
        N = 0;
    }
    if (N > kMaxGaussianN) {
        throw std::invalid_argument("Gaussian number too large (max " + std::to_string(kMaxGaussianN) + "): " + name);
    }
    if (N <= 0) {
        throw std::invalid_argument("cannot parse grid name: " + name);
    }
    StructuredGrid g;
    g.N = N;
    g.y.resize(2 * (size_t)N);
    gaussian_latitudes_npole_spole(N, g.y.data());
    g.nx.resize(2 * (size_t)N);
    if (kind == 'F' || kind == 'f') {
        std::fill(g.nx.begin(), g.nx.end(), 4 * N);  // Gaussian.cc:169
        g.regular = true;
        g.name    = "F" + std::to_string(N);
    }
    else if (kind == 'O' || kind == 'o') {
        for (int j = 0; j < N; ++j) {  // Gaussian.cc:127-134
            g.nx[j]             = 20 + 4 * j;
            g.nx[2 * N - 1 - j] = g.nx[j];
        }
        g.regular = false;
        g.name    = "O" + std::to_string(N);
    }
    else if (kind == 'N' || kind == 'n') {
        // classic reduced Gaussian grid: tabulated points per latitude (src/atlas/grid/detail/pl/classic_gaussian/N*.cc)
        std::vector<int> pl;
        if (!classic_gaussian_pl(N, pl)) {
            throw std::invalid_argument("classic reduced Gaussian grid '" + name +
                                        "' is not tabulated (N16 ... N8000 as in Atlas); pass nx[]/lat[] explicitly");
        }
        for (int j = 0; j < N; ++j) {
            g.nx[j]             = pl[j];
            g.nx[2 * N - 1 - j] = pl[j];
        }
        g.regular = false;
        g.name    = "N" + std::to_string(N);
    }
    else {
        throw std::invalid_argument("unsupported grid '" + name +
                                    "': F<N>, O<N> and the tabulated classic N<N> are known by name; pass nx[]/lat[] "
                                    "explicitly for other grids");
    }
    return g;
}

bool classic_gaussian_pl(int N, std::vector<int>& pl) {
#include "classic_pl.inc"
    for (const auto& e : kClassicPlIndex) {
        if (e[0] == N) {
            pl.clear();
            for (int i = e[1]; i < e[1] + e[2]; ++i) {
                pl.insert(pl.end(), (size_t)kClassicPlPairs[i][1], kClassicPlPairs[i][0]);
            }
            return (int)pl.size() == N;
        }
    }
    return false;
}

StructuredGrid make_reduced_gaussian_grid(int N, const int pl[], int npl) {
    // custom reduced Gaussian grid from a points-per-latitude array (npole->spole, or npole->equator)
    StructuredGrid g;
    g.N = N;
    g.y.resize(2 * (size_t)N);
    gaussian_latitudes_npole_spole(N, g.y.data());
    g.nx.resize(2 * (size_t)N);
    if (npl == 2 * N) {
        std::copy(pl, pl + npl, g.nx.begin());
    }
    else if (npl == N) {
        for (int j = 0; j < N; ++j) {
            g.nx[j]             = pl[j];
            g.nx[2 * N - 1 - j] = pl[j];
        }
    }
    else {
        throw std::invalid_argument("pl must have N or 2N entries");
    }
    g.regular = std::all_of(g.nx.begin(), g.nx.end(), [&](int v) { return v == g.nx[0]; });
    g.name    = "reduced_gaussian_N" + std::to_string(N);
    return g;
}

}  // namespace grid
}  // namespace atlas_amd
