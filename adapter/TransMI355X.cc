// Atlas plugin source: the "mi355x" Trans backend -- an atlas::trans::TransImpl that forwards to libatlas_amd.so.
//
// Compiled on the ATLAS side against an installed Atlas (>= 0.44) + eckit, as an eckit plugin loaded through
// ATLAS_PLUGIN_PATH (src/atlas/library/Library.cc:172-174, doc/example-plugin/); see adapter/README.md and
// adapter/CMakeLists.txt.  It cannot be compiled in the build image of this repository (no eckit / Atlas install):
// tests/test_adapter_source.py checks it at source level instead -- every pure virtual of
// src/atlas/trans/detail/TransImpl.h:38-191, VorDivToUV.h:36-62 and LegendreCacheCreator.h:30-44 has an override in adapter/
// with the same parameter list, the three builders are registered, every atlas_amd__ symbol used is declared in
// include/atlas_amd.h and exported by the library.
//
// Registration mirrors TransLocal's (TransLocal.cc:57, builder template detail/TransFactory.h:114-129); the backend is
// selected with option::type("mi355x") or Trans::backend("mi355x") (TransFactory.cc:228-252); Fortran callers reach it
// through the unchanged atlas__Trans__* symbols (atlas_Trans_module.F90:156-177,312-334).  VorDivToUV and
// LegendreCacheCreator resolve their implementation by the same type name (VorDivToUV.cc:84-93, LegendreCacheCreator.cc:
// 72-84): VorDivToUVMI355X.cc and LegendreCacheCreatorMI355X.cc register them; HaloExchangeMI355X.h is the class with the
// surface of parallel::HaloExchange (HaloExchange.h:37-144).
//
// Which grids (TransLocal.cc:371-488):
//   * a global structured grid, optionally with a RectangularDomain / ZonalBandDomain that crops it (the "nested" case,
//     :394-470)                                                   -> atlas_amd__Trans__*  (MFMA Legendre stage + LDS FFT)
//   * a RegularGrid whose own domain is not global (no_nest, :394-406,719-738) -> atlas_amd__RegionalTrans__new
//   * an unstructured grid (:741-790,1200-1420)                              -> atlas_amd__RegionalTrans__new_unstructured
//   * anything else (a non-global reduced grid): ATLAS_NOTIMPLEMENTED, as the reference (:402-405)
#include <cstdio>
#include <fstream>
#include <string>
#include <vector>

#include "atlas/array.h"
#include "atlas/domain.h"
#include "atlas/field/Field.h"
#include "atlas/field/FieldSet.h"
#include "atlas/grid/Iterator.h"   // IterateLonLat (Grid.h only forward-declares it): found by the front-end check of round 5
#include "atlas/grid/StructuredGrid.h"
#include "atlas/grid/UnstructuredGrid.h"
#include "atlas/trans/detail/TransFactory.h"
#include "eckit/filesystem/PathName.h"

#include "TransMI355X.h"

namespace atlas {
namespace trans {

namespace {
// the two configuration keys TransLocal honours at construction (TransLocal.cc:81-86)
std::string write_legendre(const eckit::Configuration& c) {
    return c.getString("write_legendre", "");
}
bool export_legendre(const eckit::Configuration& c) {
    return c.getBool("export_legendre", false);
}
}  // namespace

TransMI355X::TransMI355X(const Cache& cache, const Grid& grid, const Domain& domain, long truncation,
                         const eckit::Configuration& config) :
    grid_(grid, domain), truncation_(int(truncation)), spectral_(int(truncation)) {
    StructuredGrid g(grid);
    if (!g) {
        // unstructured target: Legendre polynomials at every point, point-wise Fourier sums (TransLocal.cc:741-790,1200-1420)
        UnstructuredGrid u(grid_);
        ATLAS_ASSERT(u, "the mi355x Trans backend needs a structured or an unstructured grid");
        std::vector<double> lons(u.size()), lats(u.size());
        idx_t n = 0;
        for (const PointLonLat& p : u.lonlat()) {
            lons[n] = p.lon();
            lats[n] = p.lat();
            ++n;
        }
        regional_.reset(atlas_amd__RegionalTrans__new_unstructured(int(n), lons.data(), lats.data(), truncation_));
        if (!regional_) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
        return;
    }
    if (!g.domain().global()) {
        // the grid itself is regional: TransLocal takes its no_nest branch for a RegularGrid (own latitudes, matrix Fourier
        // stage, TransLocal.cc:394-406) and throws for anything else (:402-405)
        RegularGrid r(grid);
        if (!r) {
            throw_NotImplemented("mi355x Trans: a non-global grid must be a RegularGrid (as for TransLocal)", Here());
        }
        StructuredGrid gs(grid_);   // with `domain` applied (a further crop of the regional grid)
        ATLAS_ASSERT(gs && gs.ny() > 0);
        std::vector<double> lats(gs.ny());
        for (idx_t j = 0; j < gs.ny(); ++j) {
            lats[j] = gs.y(j);
        }
        const double west = gs.x(0, 0);
        const double dlon = gs.nx(0) > 1 ? gs.x(1, 0) - gs.x(0, 0) : 0.;
        regional_.reset(atlas_amd__RegionalTrans__new(int(gs.nx(0)), west, dlon, int(gs.ny()), lats.data(), truncation_));
        if (!regional_) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
        return;
    }
    std::vector<int> nx(g.ny());
    std::vector<double> y(g.ny());
    for (idx_t j = 0; j < g.ny(); ++j) {
        nx[j] = int(g.nx(j));
        y[j]  = g.y(j);   // Atlas's own latitudes: the tables then equal TransLocal's bit for bit
    }
    agrid_.reset(atlas_amd__Grid__new_structured(int(g.ny()), nx.data(), y.data()));
    if (!agrid_) {
        throw_Exception(atlas_amd__last_error(), Here());
    }
    // crops of a global grid (the nested case of TransLocal.cc:394-470): the library works out rows and longitude windows
    // from the domain's bounds with Atlas's own rule (atlas_amd__Grid__crop_to_domain = Structured.cc:390-560); the
    // cropped grid built above must agree with it
    std::string cfg;
    StructuredGrid gs(grid_);
    if (gs && !domain.global()) {
        RectangularDomain rd(domain);
        ATLAS_ASSERT(rd, "the mi355x Trans backend supports rectangular (or zonal band) domains");
        char text[160];
        std::snprintf(text, sizeof(text), "domain=%.17g,%.17g,%.17g,%.17g", rd.xmin(), rd.xmax(), rd.ymin(), rd.ymax());
        cfg = text;
        int j0 = 0, j1 = 0;
        std::vector<int> i0(g.ny()), cnt(g.ny());
        if (atlas_amd__Grid__crop_to_domain(agrid_.get(), rd.xmin(), rd.xmax(), rd.ymin(), rd.ymax(), &j0, &j1, i0.data(),
                                            cnt.data(), int(g.ny())) != 0) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
        ATLAS_ASSERT(j1 - j0 == gs.ny(), "row range of the crop differs from Grid(grid, domain)");
        for (idx_t j = 0; j < gs.ny(); ++j) {
            ATLAS_ASSERT(gs.nx(j) == cnt[j], "longitude window of the crop differs from Grid(grid, domain)");
        }
    }
    const void* blob = cache.legendre() ? cache.legendre().data() : nullptr;   // trans/Cache.h:98-136
    const size_t len = cache.legendre() ? cache.legendre().size() : 0;
    handle_.reset(atlas_amd__Trans__new_config(agrid_.get(), truncation_, cfg.c_str(), blob, len));
    if (!handle_) {
        throw_Exception(atlas_amd__last_error(), Here());
    }
    // ---- "export_legendre" / "write_legendre" (TransLocal.cc:616-647): only when the tables were computed here, not read
    // from a cache; the blob is byte-identical to TransLocal's (symmetric ++ antisymmetric doubles, test_gpu_trans.py)
    if (!blob && (export_legendre(config) || !write_legendre(config).empty())) {
        const size_t bytes = atlas_amd__Trans__legendre_cache_size(handle_.get());
        LegendreCache out(bytes);
        check(atlas_amd__Trans__legendre_cache_export(handle_.get(), const_cast<void*>(out.legendre().data()), bytes));
        if (export_legendre(config)) {
            export_legendre_ = out;
        }
        const std::string path = write_legendre(config);
        if (!path.empty()) {
            if (eckit::PathName(path).exists()) {   // WriteCache refuses to overwrite (TransLocal.cc:130-135)
                throw_Exception("Cannot open cache file " + path + " for writing as it already exists. Remove first.", Here());
            }
            std::ofstream f(path, std::ios::binary);
            f.write(static_cast<const char*>(out.legendre().data()), std::streamsize(bytes));
            if (!f) {
                throw_Exception("writing the Legendre cache to " + path + " failed", Here());
            }
        }
    }
}

TransMI355X::~TransMI355X() = default;

void TransMI355X::invtrans(const int nb_scalar_fields, const double scalar_spectra[], const int nb_vordiv_fields,
                           const double vorticity_spectra[], const double divergence_spectra[], double gp_fields[],
                           const eckit::Configuration&) const {
    if (regional_) {
        check(atlas_amd__RegionalTrans__invtrans_vordiv(regional_.get(), nb_scalar_fields, scalar_spectra, nb_vordiv_fields,
                                                        vorticity_spectra, divergence_spectra, gp_fields));
        return;
    }
    check(atlas_amd__Trans__invtrans(handle_.get(), nb_scalar_fields, scalar_spectra, nb_vordiv_fields, vorticity_spectra,
                                     divergence_spectra, gp_fields));
}

void TransMI355X::invtrans(const int nb_scalar_fields, const double scalar_spectra[], double gp_fields[],
                           const eckit::Configuration&) const {
    if (regional_) {
        check(atlas_amd__RegionalTrans__invtrans_scalar(regional_.get(), nb_scalar_fields, scalar_spectra, gp_fields));
        return;
    }
    check(atlas_amd__Trans__invtrans_scalar(handle_.get(), nb_scalar_fields, scalar_spectra, gp_fields));
}

void TransMI355X::invtrans(const int nb_vordiv_fields, const double vorticity_spectra[], const double divergence_spectra[],
                           double gp_fields[], const eckit::Configuration&) const {
    if (regional_) {
        check(atlas_amd__RegionalTrans__invtrans_vordiv(regional_.get(), 0, nullptr, nb_vordiv_fields, vorticity_spectra,
                                                        divergence_spectra, gp_fields));
        return;
    }
    check(atlas_amd__Trans__invtrans_vordiv2wind(handle_.get(), nb_vordiv_fields, vorticity_spectra, divergence_spectra,
                                                 gp_fields));
}

// rank-1 fields, as TransLocal::invtrans(Field, Field) (TransLocal.cc:818-834).  When both fields are resident on the
// device and up to date there (Atlas built with GPU support: Field::deviceAllocated, array::make_device_view), the
// transform reads and writes the device copies -- no PCIe transfer, 17 ms instead of 180 ms per 137-level field at
// TL1279 / O1280 -- and marks the host copy of the result stale, as HaloExchange::execute(on_device) does
// (HaloExchange.h:219).
void TransMI355X::invtrans(const Field& spfield, Field& gpfield, const eckit::Configuration& config) const {
    ATLAS_ASSERT(spfield.rank() == 1, "Only rank-1 fields supported at the moment");
    ATLAS_ASSERT(gpfield.rank() == 1, "Only rank-1 fields supported at the moment");
    const bool on_device = spfield.deviceAllocated() && gpfield.deviceAllocated() && !spfield.deviceNeedsUpdate();
    if (on_device) {
        const auto sp = array::make_device_view<double, 1>(spfield);
        auto gp       = array::make_device_view<double, 1>(gpfield);
        if (regional_) {
            check(atlas_amd__RegionalTrans__invtrans_scalar_device(regional_.get(), 1, sp.data(), gp.data()));
            check(atlas_amd__RegionalTrans__synchronize(regional_.get()));
        }
        else {
            check(atlas_amd__Trans__invtrans_scalar_device(handle_.get(), 1, sp.data(), gp.data()));
            check(atlas_amd__Trans__synchronize(handle_.get()));
        }
        gpfield.setHostNeedsUpdate(true);
        gpfield.setDeviceNeedsUpdate(false);
        return;
    }
    const auto sp = array::make_view<double, 1>(spfield);
    auto gp       = array::make_view<double, 1>(gpfield);
    invtrans(1, sp.data(), gp.data(), config);
}

void TransMI355X::invtrans(const FieldSet& spfields, FieldSet& gpfields, const eckit::Configuration& config) const {
    ATLAS_ASSERT(spfields.size() == gpfields.size());   // TransLocal.cc:838-844
    for (idx_t f = 0; f < spfields.size(); ++f) {
        invtrans(spfields[f], gpfields[f], config);
    }
}

// wind field (2, npts) or (npts, 2) (TransLocal.cc:871-897)
void TransMI355X::invtrans_vordiv2wind(const Field& spvor, const Field& spdiv, Field& gpwind,
                                       const eckit::Configuration&) const {
    ATLAS_ASSERT(spvor.rank() == 1 && spdiv.rank() == 1 && gpwind.rank() == 2);
    if (regional_) {
        ATLAS_NOTIMPLEMENTED;   // Field overload on regional / unstructured targets: use the array interface
    }
    atlas_amd_Field vor{}, div{}, wind{};
    const auto v = array::make_view<double, 1>(spvor);
    const auto d = array::make_view<double, 1>(spdiv);
    auto w       = array::make_view<double, 2>(gpwind);
    vor.data = const_cast<double*>(v.data()), vor.rank = 1, vor.shape[0] = long(v.shape(0));
    div.data = const_cast<double*>(d.data()), div.rank = 1, div.shape[0] = long(d.shape(0));
    wind.data = w.data(), wind.rank = 2, wind.shape[0] = long(w.shape(0)), wind.shape[1] = long(w.shape(1));
    check(atlas_amd__Trans__invtrans_vordiv2wind_field(handle_.get(), &vor, &div, &wind));
}

namespace {
static TransBuilderGrid<TransMI355X> builder("mi355x", "mi355x");   // cf. TransLocal.cc:57
}

}  // namespace trans
}  // namespace atlas
