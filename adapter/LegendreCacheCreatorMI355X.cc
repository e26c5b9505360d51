// Atlas plugin source: trans::LegendreCacheCreator, type "mi355x" (reference interface
// src/atlas/trans/LegendreCacheCreator.h:30-111; the "local" implementation it mirrors: local/LegendreCacheCreatorLocal.cc:
// 30-165).  The cache blob is byte-identical to TransLocal's (symmetric ++ antisymmetric tables), so the uid keeps the
// "local-" prefix: a cache written by either backend can be read by the other.
#include <string>
#include <vector>

#include "atlas/grid.h"
#include "atlas/option.h"
#include "atlas/runtime/Exception.h"
#include "atlas/trans/LegendreCacheCreator.h"
#include "atlas/trans/Trans.h"

#include "TransMI355X.h"

namespace atlas {
namespace trans {

class LegendreCacheCreatorMI355X : public LegendreCacheCreatorImpl {
public:
    LegendreCacheCreatorMI355X(const Grid& grid, int truncation, const eckit::Configuration& config = util::NoConfig()) :
        grid_(grid), truncation_(truncation), config_(config) {}
    ~LegendreCacheCreatorMI355X() override = default;

    // structured, unprojected grids (LegendreCacheCreatorLocal.cc:134-142)
    bool supported() const override {
        if (!StructuredGrid(grid_) || grid_.projection()) {
            return false;
        }
        return true;
    }

    // "local-T<T>-GaussianN<N>|L-ny<ny>|S-ny<ny>|grid-<md5>-OPT<md5>" (LegendreCacheCreatorLocal.cc:66-124): the library
    // reproduces the reference's strings (tests/test_host_logic.py: uid goldens of test_trans.cc:600-696)
    std::string uid() const override {
        if (unique_identifier_.empty()) {
            atlas_amd_Grid* g = make_grid();
            char text[256];
            const int rc = atlas_amd__LegendreCacheCreator__uid(g, truncation_, config_.getBool("flt", false) ? 1 : 0, text,
                                                                sizeof(text));
            atlas_amd__Grid__delete(g);
            if (rc != 0) {
                throw_Exception(atlas_amd__last_error(), Here());
            }
            unique_identifier_ = text;
        }
        return unique_identifier_;
    }

    // LegendreCacheCreatorLocal.cc:144-146
    void create(const std::string& path) const override {
        Trans tmp(grid_, truncation_, config_ | option::type("mi355x") | option::write_legendre(path));
    }

    // LegendreCacheCreatorLocal.cc:148-158
    Cache create() const override {
        util::Config export_legendre("export_legendre", true);
        Trans tmp(grid_, truncation_, config_ | option::type("mi355x") | export_legendre);
        auto impl = dynamic_cast<const TransMI355X*>(tmp.get());
        ATLAS_ASSERT(impl);
        return impl->export_legendre_;
    }

    // LegendreCacheCreatorLocal.cc:160-162
    size_t estimate() const override { return size_t(atlas_amd__LegendreCacheCreator__estimate(truncation_)); }

private:
    atlas_amd_Grid* make_grid() const {
        StructuredGrid g(grid_);
        ATLAS_ASSERT(g, "LegendreCacheCreator (mi355x): structured grids");
        std::vector<int> nx(g.ny());
        std::vector<double> y(g.ny());
        for (idx_t j = 0; j < g.ny(); ++j) {
            nx[j] = int(g.nx(j));
            y[j]  = g.y(j);
        }
        atlas_amd_Grid* out = atlas_amd__Grid__new_structured(int(g.ny()), nx.data(), y.data());
        if (!out) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
        return out;
    }
    const Grid grid_;
    const int truncation_;
    const util::Config config_;
    mutable std::string unique_identifier_;
};

namespace {
static LegendreCacheCreatorBuilder<LegendreCacheCreatorMI355X> builder("mi355x");   // cf. LegendreCacheCreatorLocal.cc:30
}

}  // namespace trans
}  // namespace atlas
