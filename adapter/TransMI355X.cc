// Atlas plugin source: the "mi355x" Trans backend -- an atlas::trans::TransImpl that forwards to libatlas_amd.so.
//
// Compiled on the ATLAS side against an installed Atlas (>= 0.44) + eckit, as an eckit plugin loaded through
// ATLAS_PLUGIN_PATH (src/atlas/library/Library.cc:172-174, doc/example-plugin/); see adapter/README.md and
// adapter/CMakeLists.txt.  It cannot be compiled in the build image of this repository (no eckit / Atlas install):
// tests/test_adapter_source.py checks it at source level instead -- every pure virtual of
// src/atlas/trans/detail/TransImpl.h:38-191 has an override here with the same parameter list, and every atlas_amd__
// symbol used is declared in include/atlas_amd.h.
//
// Registration mirrors TransLocal's (TransLocal.cc:57, builder template detail/TransFactory.h:114-129); the backend is
// selected with option::type("mi355x") or Trans::backend("mi355x") (TransFactory.cc:228-252); Fortran callers reach it
// through the unchanged atlas__Trans__* symbols (atlas_Trans_module.F90:156-177,312-334).
#include <cstdio>
#include <string>
#include <vector>

#include "atlas/array.h"
#include "atlas/domain.h"
#include "atlas/field/Field.h"
#include "atlas/field/FieldSet.h"
#include "atlas/functionspace/Spectral.h"
#include "atlas/grid/StructuredGrid.h"
#include "atlas/runtime/Exception.h"
#include "atlas/trans/Cache.h"
#include "atlas/trans/detail/TransFactory.h"
#include "atlas/trans/detail/TransImpl.h"

extern "C" {
#include "atlas_amd.h"
}

namespace atlas {
namespace trans {

class TransMI355X : public TransImpl {
public:
    TransMI355X(const Cache& cache, const Grid& grid, const Domain& domain, long truncation,
                const eckit::Configuration& config = util::NoConfig());
    TransMI355X(const Grid& grid, long truncation, const eckit::Configuration& config = util::NoConfig()) :
        TransMI355X(Cache(), grid, grid.domain(), truncation, config) {}
    TransMI355X(const Grid& grid, const Domain& domain, long truncation,
                const eckit::Configuration& config = util::NoConfig()) :
        TransMI355X(Cache(), grid, domain, truncation, config) {}
    ~TransMI355X() override;

    std::string type() const override { return "mi355x"; }
    int truncation() const override { return atlas_amd__Trans__truncation(handle_); }
    size_t nb_spectral_coefficients() const override { return size_t(atlas_amd__Trans__nb_spectral_coefficients(handle_)); }
    size_t nb_spectral_coefficients_global() const override { return nb_spectral_coefficients(); }
    const Grid& grid() const override { return grid_; }
    const functionspace::Spectral& spectral() const override { return spectral_; }

    // ---- inverse transforms: what TransLocal implements (TransLocal.cc:818-934,1486-1490,1523-1597)
    void invtrans(const Field& spfield, Field& gpfield, const eckit::Configuration& = util::NoConfig()) const override;
    void invtrans(const FieldSet& spfields, FieldSet& gpfields,
                  const eckit::Configuration& = util::NoConfig()) const override;
    void invtrans_vordiv2wind(const Field& spvor, const Field& spdiv, Field& gpwind,
                              const eckit::Configuration& = util::NoConfig()) const override;
    void invtrans(const int nb_scalar_fields, const double scalar_spectra[], const int nb_vordiv_fields,
                  const double vorticity_spectra[], const double divergence_spectra[], double gp_fields[],
                  const eckit::Configuration& = util::NoConfig()) const override;
    void invtrans(const int nb_scalar_fields, const double scalar_spectra[], double gp_fields[],
                  const eckit::Configuration& = util::NoConfig()) const override;
    void invtrans(const int nb_vordiv_fields, const double vorticity_spectra[], const double divergence_spectra[],
                  double gp_fields[], const eckit::Configuration& = util::NoConfig()) const override;

    // ---- ATLAS_NOTIMPLEMENTED in TransLocal as well (TransLocal.cc:848-857,899-927,1599-1685)
    void dirtrans(const Field&, Field&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans(const FieldSet&, FieldSet&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans_wind2vordiv(const Field&, Field&, Field&,
                              const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans_adj(const Field&, Field&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans_adj(const FieldSet&, FieldSet&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans_wind2vordiv_adj(const Field&, const Field&, Field&,
                                  const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_grad(const Field&, Field&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_grad(const FieldSet&, FieldSet&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_adj(const Field&, Field&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_adj(const FieldSet&, FieldSet&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_grad_adj(const Field&, Field&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_grad_adj(const FieldSet&, FieldSet&, const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_vordiv2wind_adj(const Field&, Field&, Field&,
                                  const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_adj(const int, const double[], const int, double[], double[], double[],
                      const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_adj(const int, const double[], double[], const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void invtrans_adj(const int, const double[], double[], double[],
                      const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans(const int, const double[], double[], const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }
    void dirtrans(const int, const double[], double[], double[],
                  const eckit::Configuration& = util::NoConfig()) const override {
        ATLAS_NOTIMPLEMENTED;
    }

private:
    static void check(int rc) {
        if (rc != 0) {
            const std::string what = atlas_amd__last_error();
            if (what.rfind("Not implemented", 0) == 0) {
                ATLAS_NOTIMPLEMENTED;
            }
            throw_Exception(what, Here());
        }
    }
    Grid grid_;
    functionspace::Spectral spectral_;
    atlas_amd_Grid* agrid_   = nullptr;
    atlas_amd_Trans* handle_ = nullptr;
};

TransMI355X::TransMI355X(const Cache& cache, const Grid& grid, const Domain& domain, long truncation,
                         const eckit::Configuration&) :
    grid_(grid, domain), spectral_(int(truncation)) {
    StructuredGrid g(grid);
    ATLAS_ASSERT(g, "the mi355x Trans backend needs a structured grid (use type 'local' otherwise)");
    std::vector<int> nx(g.ny());
    std::vector<double> y(g.ny());
    for (idx_t j = 0; j < g.ny(); ++j) {
        nx[j] = int(g.nx(j));
        y[j]  = g.y(j);   // Atlas's own latitudes: the tables then equal TransLocal's bit for bit
    }
    agrid_ = atlas_amd__Grid__new_structured(int(g.ny()), nx.data(), y.data());
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
        if (atlas_amd__Grid__crop_to_domain(agrid_, rd.xmin(), rd.xmax(), rd.ymin(), rd.ymax(), &j0, &j1, i0.data(), cnt.data(),
                                            int(g.ny())) != 0) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
        ATLAS_ASSERT(j1 - j0 == gs.ny(), "row range of the crop differs from Grid(grid, domain)");
        for (idx_t j = 0; j < gs.ny(); ++j) {
            ATLAS_ASSERT(gs.nx(j) == cnt[j], "longitude window of the crop differs from Grid(grid, domain)");
        }
    }
    const void* blob = cache.legendre() ? cache.legendre().data() : nullptr;   // trans/Cache.h:98-136
    const size_t len = cache.legendre() ? cache.legendre().size() : 0;
    handle_          = atlas_amd__Trans__new_config(agrid_, int(truncation), cfg.c_str(), blob, len);
    if (!handle_) {
        atlas_amd__Grid__delete(agrid_);
        throw_Exception(atlas_amd__last_error(), Here());
    }
}

TransMI355X::~TransMI355X() {
    atlas_amd__Trans__delete(handle_);
    atlas_amd__Grid__delete(agrid_);
}

void TransMI355X::invtrans(const int nb_scalar_fields, const double scalar_spectra[], const int nb_vordiv_fields,
                           const double vorticity_spectra[], const double divergence_spectra[], double gp_fields[],
                           const eckit::Configuration&) const {
    check(atlas_amd__Trans__invtrans(handle_, nb_scalar_fields, scalar_spectra, nb_vordiv_fields, vorticity_spectra,
                                     divergence_spectra, gp_fields));
}

void TransMI355X::invtrans(const int nb_scalar_fields, const double scalar_spectra[], double gp_fields[],
                           const eckit::Configuration&) const {
    check(atlas_amd__Trans__invtrans_scalar(handle_, nb_scalar_fields, scalar_spectra, gp_fields));
}

void TransMI355X::invtrans(const int nb_vordiv_fields, const double vorticity_spectra[], const double divergence_spectra[],
                           double gp_fields[], const eckit::Configuration&) const {
    check(atlas_amd__Trans__invtrans_vordiv2wind(handle_, nb_vordiv_fields, vorticity_spectra, divergence_spectra,
                                                 gp_fields));
}

// rank-1 fields, as TransLocal::invtrans(Field, Field) (TransLocal.cc:818-834)
void TransMI355X::invtrans(const Field& spfield, Field& gpfield, const eckit::Configuration& config) const {
    ATLAS_ASSERT(spfield.rank() == 1, "Only rank-1 fields supported at the moment");
    ATLAS_ASSERT(gpfield.rank() == 1, "Only rank-1 fields supported at the moment");
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
    atlas_amd_Field vor{}, div{}, wind{};
    const auto v = array::make_view<double, 1>(spvor);
    const auto d = array::make_view<double, 1>(spdiv);
    auto w       = array::make_view<double, 2>(gpwind);
    vor.data = const_cast<double*>(v.data()), vor.rank = 1, vor.shape[0] = long(v.shape(0));
    div.data = const_cast<double*>(d.data()), div.rank = 1, div.shape[0] = long(d.shape(0));
    wind.data = w.data(), wind.rank = 2, wind.shape[0] = long(w.shape(0)), wind.shape[1] = long(w.shape(1));
    check(atlas_amd__Trans__invtrans_vordiv2wind_field(handle_, &vor, &div, &wind));
}

namespace {
static TransBuilderGrid<TransMI355X> builder("mi355x", "mi355x");   // cf. TransLocal.cc:57
}

}  // namespace trans
}  // namespace atlas
