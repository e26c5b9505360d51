// C++ parity test of the host interface include/atlas_amd.hpp, written the way the reference tests its transform
// (src/tests/trans/test_transgeneral.cc: unit spectral coefficient -> closed-form spherical harmonic on every grid
// point, rel-RMS tolerance 1e-13, :93-131,433-449,829-839; src/tests/trans/test_trans.cc: backend registry;
// src/tests/parallel/test_haloexchange.cc: index pattern).  Built and run by tests/test_cxx_api.py:
//   ./test_trans_cxx --host-only   API, errors, grids, halo index logic (no GPU)
//   ./test_trans_cxx               + the transforms on the MI355X
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "atlas_amd.hpp"

using namespace atlas_amd;

static int failures = 0;
#define EXPECT(cond)                                                                \
    do {                                                                            \
        if (!(cond)) {                                                              \
            std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #cond);           \
            ++failures;                                                             \
        }                                                                           \
    } while (0)
#define EXPECT_THROWS_AS(expr, Type)                                                \
    do {                                                                            \
        bool caught_ = false;                                                       \
        try {                                                                       \
            expr;                                                                   \
        }                                                                           \
        catch (const Type&) {                                                       \
            caught_ = true;                                                         \
        }                                                                           \
        if (!caught_) {                                                             \
            std::printf("FAILED %s:%d  %s did not throw %s\n", __FILE__, __LINE__, #expr, #Type); \
            ++failures;                                                             \
        }                                                                           \
    } while (0)

// normalised associated Legendre functions, 1/2 * integral of P^2 over mu = 1 (LegendrePolynomials.cc:29)
static double legendre_closed_form(int n, int m, double phi) {
    const double s = std::sin(phi), c = std::cos(phi);
    if (n == 0 && m == 0) return 1.;
    if (n == 1 && m == 0) return std::sqrt(3.) * s;
    if (n == 1 && m == 1) return std::sqrt(1.5) * c;
    if (n == 2 && m == 0) return std::sqrt(5.) * (1.5 * s * s - 0.5);
    if (n == 2 && m == 1) return std::sqrt(7.5) * s * c;
    if (n == 2 && m == 2) return std::sqrt(15. / 8.) * c * c;
    if (n == 3 && m == 2) return std::sqrt(105. / 8.) * c * c * s;
    if (n == 3 && m == 3) return std::sqrt(35. / 16.) * c * c * c;
    return NAN;
}

static double rel_rms(const std::vector<double>& a, const std::vector<double>& b) {
    double num = 0, den = 0;
    for (size_t i = 0; i < a.size(); ++i) {
        num += (a[i] - b[i]) * (a[i] - b[i]);
        den += b[i] * b[i];
    }
    return std::sqrt(num / (den > 0 ? den : 1.));
}

static void case_backend_registry() {
    EXPECT(trans::Trans::hasBackend("local"));
    EXPECT(trans::Trans::hasBackend("mi355x"));
    EXPECT(!trans::Trans::hasBackend("ifs"));
    EXPECT(trans::Trans::backend() == "local");
    trans::Trans::backend("mi355x");
    EXPECT(trans::Trans::backend() == "mi355x");
    trans::Trans::backend("local");
    EXPECT_THROWS_AS(trans::Trans::backend("ifs"), Exception);
    EXPECT(trans::Trans::backend() == "local");
}

static void case_grids() {
    StructuredGrid o("O32");
    EXPECT(o.ny() == 64 && o.nx().front() == 20 && o.nx()[31] == 144 && o.size() == 5248 && !o.regular());
    StructuredGrid f("F32");
    EXPECT(f.ny() == 64 && f.nxmax() == 128 && f.size() == 64 * 128 && f.regular());
    const std::vector<double> lats = gaussian_latitudes_npole_spole(32);
    EXPECT(lats.size() == 64 && lats[0] > 87. && std::fabs(lats[0] + lats[63]) < 1e-12 && o.y() == lats);
    EXPECT_THROWS_AS(StructuredGrid("X12"), Exception);
    StructuredGrid custom(std::vector<int>(19, 36), [] {
        std::vector<double> y(19);
        for (int j = 0; j < 19; ++j) y[j] = 90. - 10. * j;
        return y;
    }());
    EXPECT(custom.ny() == 19 && custom.size() == 19 * 36);
}

static void case_halo_index_logic() {
    // one process, the 9-node pattern of a periodic ring: nodes 0 and 8 are ghosts of nodes 7 and 1
    parallel::HaloExchange hx;
    const int part[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ridx[9] = {7, 1, 2, 3, 4, 5, 6, 7, 1};
    hx.setup(part, ridx, 0, 9);
    EXPECT(hx.nproc() == 1 && hx.sendcnt() == 2 && hx.recvcnt() == 2);
    EXPECT((hx.get("recvmap") == std::vector<int>{0, 8}));
    EXPECT((hx.get("sendmap") == std::vector<int>{7, 1}));
    EXPECT((hx.get("sendcounts") == std::vector<int>{2}));

    // StructuredColumns of O8 on one part with halo 1: every halo point has an owner and a consistent (i,j)
    StructuredGrid g("O8");
    functionspace::StructuredColumns fs(g, 1);
    EXPECT(fs.sizeOwned() == g.size() && fs.sizeHalo() > fs.sizeOwned());
    EXPECT(fs.j_begin() == 0 && fs.j_end() == g.ny() && fs.j_begin_halo() == -1 && fs.j_end_halo() == g.ny() + 1);
    const std::vector<int> ii = fs.field("index_i"), jj = fs.field("index_j"), ghost = fs.field("ghost");
    const std::vector<int> remote = fs.field("remote_idx");
    int bad = 0;
    for (int n = 0; n < fs.sizeHalo(); ++n) {
        bad += fs.index(ii[n], jj[n]) != n;
        bad += (n < fs.sizeOwned()) != (ghost[n] == 0);
        bad += remote[n] < 0 || remote[n] >= fs.sizeOwned();
    }
    EXPECT(bad == 0);
    parallel::HaloExchange fhx;
    fs.setup_halo_exchange(fhx);
    EXPECT(fhx.recvcnt() == fs.sizeHalo() - fs.sizeOwned());
}

static void case_node_columns_contract() {
    // functionspace::NodeColumns (NodeColumns.cc:101-113,357-459): set-up from (partition, remote_index, nb_nodes) WITHOUT halo_begin --
    // a ghost node in the middle of the array is found -- and the rank / shape checks of haloExchange
    const int part[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ridx[10] = {0, 1, 2, 9, 4, 5, 6, 7, 1, 9};   // node 3 is a ghost of node 9, node 8 of node 1
    functionspace::NodeColumns fs(part, ridx, 10);
    EXPECT(fs.nb_nodes() == 10 && fs.halo_exchange().recvcnt() == 2 && fs.halo_exchange().sendcnt() == 2);
    EXPECT((fs.halo_exchange().get("recvmap") == std::vector<int>{3, 8}));
    EXPECT((fs.halo_exchange().get("sendmap") == std::vector<int>{9, 1}));
    // Fortran-side base 1
    int ridx1[10];
    for (int i = 0; i < 10; ++i) ridx1[i] = ridx[i] + 1;
    functionspace::NodeColumns fs1(part, ridx1, 10, 1);
    EXPECT((fs1.halo_exchange().get("sendmap") == std::vector<int>{9, 1}));
    std::vector<double> f(10 * 3);
    const int shape5[5] = {10, 1, 1, 1, 3}, shape_bad[2] = {9, 3};
    bool rank_refused = false, shape_refused = false;
    try {
        fs.haloExchange(f.data(), 5, shape5);
    }
    catch (const Exception& e) {
        rank_refused = std::strstr(e.what(), "Rank not supported") != nullptr;
    }
    try {
        fs.haloExchange(f.data(), 2, shape_bad);
    }
    catch (const Exception&) {
        shape_refused = true;
    }
    EXPECT(rank_refused && shape_refused);
}

static void case_node_columns_exchange() {
    // the exchange itself (needs a device): levels and variables of a ghost node take the owner's values; the adjoint adds the ghost's
    // contribution to the owner and zeroes the ghost (HaloExchange.h:222-290), int and double, rank 1 .. 3
    const int part[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int ridx[10] = {0, 1, 2, 9, 4, 5, 6, 7, 1, 9};
    functionspace::NodeColumns fs(part, ridx, 10);
    std::vector<double> f(10 * 2 * 3);
    for (size_t i = 0; i < f.size(); ++i) f[i] = (double)i;
    const int shape3[3] = {10, 2, 3};
    fs.haloExchange(f.data(), 3, shape3);
    bool ok = true;
    for (int k = 0; k < 6; ++k) {
        ok = ok && f[3 * 6 + k] == (double)(9 * 6 + k) && f[8 * 6 + k] == (double)(1 * 6 + k);
    }
    EXPECT(ok);
    std::vector<int> g(10);
    for (int i = 0; i < 10; ++i) g[i] = 100 + i;
    const int shape1[1] = {10};
    fs.haloExchange(g.data(), 1, shape1);
    EXPECT(g[3] == 109 && g[8] == 101 && g[9] == 109);
    std::vector<double> a(10, 1.0);
    fs.adjointHaloExchange(a.data(), 1, shape1);
    EXPECT(a[9] == 2.0 && a[1] == 2.0 && a[3] == 0.0 && a[8] == 0.0 && a[0] == 1.0);
}

static void case_no_device_no_fallback() {
    StructuredGrid g("O32");
    bool threw = false;
    try {
        trans::Trans t(g, 31);
    }
    catch (const Exception& e) {
        threw = std::strstr(e.what(), "HIP device") != nullptr;
    }
    EXPECT(threw);
}

// ---- GPU cases -------------------------------------------------------------------------------------------------
static void case_invtrans_analytic(const std::string& gridname) {
    const int T = 31;
    StructuredGrid g(gridname);
    trans::Trans trans(g, T, option::type("local"));
    EXPECT(trans.truncation() == T);
    EXPECT(trans.nb_spectral_coefficients() == size_t((T + 1) * (T + 2)));
    EXPECT(trans.nb_gridpoints() == size_t(g.size()));
    const std::vector<int> nx   = g.nx();
    const std::vector<double> y = g.y();
    const int nm[8][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {2, 2}, {3, 2}, {3, 3}};
    // all cases at once as separate fields: sp[(2*pos + imag)*nf + fld]
    std::vector<std::pair<int, int>> cases;  // (index into nm, imag)
    for (int c = 0; c < 8; ++c) {
        cases.push_back({c, 0});
        if (nm[c][1] > 0) cases.push_back({c, 1});
    }
    const int nf = int(cases.size());
    std::vector<double> sp(trans.nb_spectral_coefficients() * nf, 0.), gp(size_t(g.size()) * nf, -999.);
    for (int f = 0; f < nf; ++f) {
        const int n = nm[cases[f].first][0], m = nm[cases[f].first][1];
        const size_t pos = size_t(2 * T + 3 - m) * m / 2 + (n - m);
        sp[(2 * pos + cases[f].second) * nf + f] = 1.;
    }
    trans.invtrans(nf, sp.data(), gp.data());
    for (int f = 0; f < nf; ++f) {
        const int n = nm[cases[f].first][0], m = nm[cases[f].first][1], imag = cases[f].second;
        std::vector<double> ref, got(gp.begin() + size_t(f) * g.size(), gp.begin() + size_t(f + 1) * g.size());
        for (int j = 0; j < g.ny(); ++j) {
            const double P = legendre_closed_form(n, m, y[j] * M_PI / 180.);
            for (int i = 0; i < nx[j]; ++i) {
                const double lam = StructuredGrid::x(i, nx[j]) * M_PI / 180.;
                ref.push_back(imag == 0 ? P * std::cos(m * lam) * (m > 0 ? 2. : 1.) : -2. * P * std::sin(m * lam));
            }
        }
        const double rms = rel_rms(got, ref);
        if (!(rms < 1e-13)) {
            std::printf("  %s n=%d m=%d imag=%d rel-rms %.3e\n", gridname.c_str(), n, m, imag, rms);
        }
        EXPECT(rms < 1e-13);  // tolerance of test_transgeneral.cc:829-839
    }
}

// test_transgeneral.cc:751-954 "test_trans_domain": the analytic harmonics on the points of a RectangularDomain crop of O64
static void case_invtrans_domain_analytic() {
    const int T = 63;
    StructuredGrid g("O64");
    const RectangularDomain domain{-5., 5., -2.5, 0.};   // test_transgeneral.cc:760-762
    trans::Trans trans(g, domain, T, option::type("local"));
    const std::vector<int> nx   = g.nx();
    const std::vector<double> y = g.y();
    // the points of the crop, from the grid definition: rows with south <= y <= north, longitudes k dx in [west, east]
    std::vector<std::pair<double, double>> pts;   // (lat, lon) in radians
    for (int j = 0; j < g.ny(); ++j) {
        if (y[j] < domain.south - 1e-6 || y[j] > domain.north + 1e-6) continue;
        const double dx = 360. / nx[j];
        for (int k = -nx[j]; k <= nx[j]; ++k) {
            if (k * dx >= domain.west - 1e-6 && k * dx <= domain.east + 1e-6) {
                pts.push_back({y[j] * M_PI / 180., k * dx * M_PI / 180.});
            }
        }
    }
    EXPECT(pts.size() == 14 && trans.nb_gridpoints() == pts.size());
    const int nm[6][2] = {{0, 0}, {1, 0}, {1, 1}, {2, 1}, {3, 2}, {3, 3}};
    std::vector<std::pair<int, int>> cases;
    for (int c = 0; c < 6; ++c) {
        cases.push_back({c, 0});
        if (nm[c][1] > 0) cases.push_back({c, 1});
    }
    const int nf = int(cases.size());
    std::vector<double> sp(trans.nb_spectral_coefficients() * nf, 0.), gp(pts.size() * nf, -999.);
    for (int f = 0; f < nf; ++f) {
        const int n = nm[cases[f].first][0], m = nm[cases[f].first][1];
        const size_t pos = size_t(2 * T + 3 - m) * m / 2 + (n - m);
        sp[(2 * pos + cases[f].second) * nf + f] = 1.;
    }
    trans.invtrans(nf, sp.data(), gp.data());
    for (int f = 0; f < nf; ++f) {
        const int n = nm[cases[f].first][0], m = nm[cases[f].first][1], imag = cases[f].second;
        std::vector<double> ref, got(gp.begin() + size_t(f) * pts.size(), gp.begin() + size_t(f + 1) * pts.size());
        for (const auto& pt : pts) {
            const double P = legendre_closed_form(n, m, pt.first);
            ref.push_back(imag == 0 ? P * std::cos(m * pt.second) * (m > 0 ? 2. : 1.) : -2. * P * std::sin(m * pt.second));
        }
        EXPECT(rel_rms(got, ref) < 1e-13);
    }
}

static void case_vordiv2wind_and_not_implemented() {
    const int T = 31;
    StructuredGrid g("F32");
    trans::Trans trans(g, T);
    const size_t nspec = trans.nb_spectral_coefficients(), npts = trans.nb_gridpoints();
    // solid-body rotation: vorticity = 2*omega*sin(lat) -> coefficient (n=1, m=0) = 2*omega/sqrt(3); u = omega*a*cos(lat)
    const double a = 6371229., omega = 1e-5;
    std::vector<double> vor(nspec, 0.), div(nspec, 0.), wind(2 * npts, -999.);
    vor[2 * 1] = 2. * omega / std::sqrt(3.);  // pos(0,1) = 1
    trans.invtrans(1, vor.data(), div.data(), wind.data());
    const std::vector<double> y = g.y();
    double err = 0, scale = omega * a;
    for (int j = 0; j < g.ny(); ++j) {
        for (int i = 0; i < g.nxmax(); ++i) {
            const size_t p = size_t(j) * g.nxmax() + i;
            err            = std::fmax(err, std::fabs(wind[p] - omega * a * std::cos(y[j] * M_PI / 180.)));
            err            = std::fmax(err, std::fabs(wind[npts + p]));
        }
    }
    EXPECT(err < 1e-10 * scale);
    // the stand-alone VorDivToUV and the fused path agree: U,V -> scalar invtrans -> / cos(lat)
    trans::VorDivToUV vd2uv(T);
    std::vector<double> U(nspec), V(nspec), uv(npts);
    vd2uv.execute(int(nspec), 1, vor.data(), div.data(), U.data(), V.data());
    trans.invtrans(1, U.data(), uv.data());
    double err2 = 0;
    for (int j = 0; j < g.ny(); ++j) {
        for (int i = 0; i < g.nxmax(); ++i) {
            const size_t p = size_t(j) * g.nxmax() + i;
            err2           = std::fmax(err2, std::fabs(uv[p] / std::cos(y[j] * M_PI / 180.) - wind[p]));
        }
    }
    EXPECT(err2 < 1e-10 * scale);
    // TransLocal: ATLAS_NOTIMPLEMENTED (TransLocal.cc:848-857,899-927)
    std::vector<double> gp(npts), sp(nspec);
    EXPECT_THROWS_AS(trans.dirtrans(1, gp.data(), sp.data()), NotImplemented);
    EXPECT_THROWS_AS(trans.invtrans_adj(1, gp.data(), sp.data()), NotImplemented);
}

static void case_halo_exchange_on_structured_columns() {
    // test_structuredcolumns_haloexchange.cc:38-60 in spirit: a field filled with the global index on owned points
    // must carry the owner's value on every halo point after the exchange
    StructuredGrid g("O16");
    functionspace::StructuredColumns fs(g, 2);
    parallel::HaloExchange hx;
    fs.setup_halo_exchange(hx);
    const std::vector<std::int64_t> glb = fs.global_index();
    const int nlev = 3;
    std::vector<double> field(size_t(fs.sizeHalo()) * nlev, -1.);
    for (int n = 0; n < fs.sizeOwned(); ++n) {
        for (int k = 0; k < nlev; ++k) field[size_t(n) * nlev + k] = double(glb[n]) * 10. + k;
    }
    const int strides[1] = {1}, shape[1] = {nlev};
    hx.execute(field.data(), strides, shape, 1);
    int bad = 0;
    for (int n = 0; n < fs.sizeHalo(); ++n) {
        for (int k = 0; k < nlev; ++k) bad += field[size_t(n) * nlev + k] != double(glb[n]) * 10. + k;
    }
    EXPECT(bad == 0);
}

int main(int argc, char** argv) {
    const bool host_only = argc > 1 && std::string(argv[1]) == "--host-only";
    struct Case {
        const char* name;
        std::function<void()> run;
        bool gpu;
    };
    const std::vector<Case> cases = {
        {"backend_registry", case_backend_registry, false},
        {"grids", case_grids, false},
        {"halo_index_logic", case_halo_index_logic, false},
        {"node_columns_contract", case_node_columns_contract, false},
        {"node_columns_exchange", case_node_columns_exchange, true},
        {"invtrans_analytic_F32", [] { case_invtrans_analytic("F32"); }, true},
        {"invtrans_analytic_O32", [] { case_invtrans_analytic("O32"); }, true},
        {"invtrans_domain_analytic", case_invtrans_domain_analytic, true},
        {"vordiv2wind_and_not_implemented", case_vordiv2wind_and_not_implemented, true},
        {"halo_exchange_on_structured_columns", case_halo_exchange_on_structured_columns, true},
    };
    if (host_only && device_count() == 0) {
        case_no_device_no_fallback();
    }
    for (const Case& c : cases) {
        if (c.gpu && host_only) {
            continue;
        }
        const int before = failures;
        try {
            c.run();
        }
        catch (const std::exception& e) {
            std::printf("FAILED case %s: exception %s\n", c.name, e.what());
            ++failures;
        }
        std::printf("%s %s\n", failures == before ? "ok    " : "FAILED", c.name);
    }
    std::printf("%d failure(s)\n", failures);
    return failures ? 1 : 0;
}
