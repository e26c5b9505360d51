// Atlas plugin header: the "mi355x" Trans backend -- an atlas::trans::TransImpl that forwards to libatlas_amd.so.
// See TransMI355X.cc.  The class is in a header (as TransLocal.h is) because LegendreCacheCreatorMI355X::create() returns the
// Legendre cache a Trans exported (export_legendre_), exactly as LegendreCacheCreatorLocal.cc:153-158 does with TransLocal.
#pragma once
#include <memory>
#include <string>

#include "atlas/functionspace/Spectral.h"
#include "atlas/grid/Grid.h"
#include "atlas/runtime/Exception.h"
#include "atlas/trans/Cache.h"
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
    int truncation() const override { return truncation_; }
    size_t nb_spectral_coefficients() const override { return size_t(truncation_ + 1) * size_t(truncation_ + 2); }
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


    // the Legendre cache made on request (config "export_legendre", TransLocal.cc:616-625): LegendreCacheCreatorMI355X
    // hands it out (TransLocal.h declares the creator a friend for the same purpose)
    Cache export_legendre_;

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
    struct GridDeleter {
        void operator()(atlas_amd_Grid* g) const { atlas_amd__Grid__delete(g); }
    };
    struct TransDeleter {
        void operator()(atlas_amd_Trans* t) const { atlas_amd__Trans__delete(t); }
    };
    struct RegionalDeleter {
        void operator()(atlas_amd_RegionalTrans* t) const { atlas_amd__RegionalTrans__delete(t); }
    };
    Grid grid_;
    int truncation_;
    functionspace::Spectral spectral_;
    // exactly one of handle_ / regional_ is set.  Owning pointers: a constructor that throws after the library objects
    // exist (an ATLAS_ASSERT on the crop, a failed cache export) releases them
    std::unique_ptr<atlas_amd_Grid, GridDeleter> agrid_;
    std::unique_ptr<atlas_amd_Trans, TransDeleter> handle_;             // global structured grid, or a rectangular crop of one
    std::unique_ptr<atlas_amd_RegionalTrans, RegionalDeleter> regional_;  // TransLocal's no_nest / unstructured targets
};

}  // namespace trans
}  // namespace atlas
