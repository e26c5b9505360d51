// Atlas plugin source: trans::VorDivToUV, type "mi355x" (reference interface src/atlas/trans/VorDivToUV.h:36-133; the "local"
// implementation it replaces: local/VorDivToUVLocal.cc:62-189).  VorDivToUVFactory::build picks the implementation by
// config "type" (VorDivToUV.cc:84-93), i.e. Trans::backend("mi355x") or option::type("mi355x") selects this one.
#include <string>

#include "atlas/functionspace/Spectral.h"
#include "atlas/runtime/Exception.h"
#include "atlas/trans/VorDivToUV.h"

extern "C" {
#include "atlas_amd.h"
}

namespace atlas {
namespace trans {

class VorDivToUVMI355X : public VorDivToUVImpl {
public:
    VorDivToUVMI355X(const FunctionSpace& fs, const eckit::Configuration& = util::NoConfig()) :
        truncation_(functionspace::Spectral(fs).truncation()) {}   // VorDivToUVLocal.cc:183
    VorDivToUVMI355X(int truncation, const eckit::Configuration& = util::NoConfig()) : truncation_(truncation) {}
    ~VorDivToUVMI355X() override = default;

    int truncation() const override { return truncation_; }

    // U = u cos(lat), V = v cos(lat) in spectral space (VorDivToUVLocal.cc:62-179): host arrays, the kernel runs on the device
    void execute(const int nb_coeff, const int nb_fields, const double vorticity[], const double divergence[], double U[],
                 double V[], const eckit::Configuration& = util::NoConfig()) const override {
        if (atlas_amd__VorDivToUV__execute(truncation_, nb_coeff, nb_fields, vorticity, divergence, U, V) != 0) {
            throw_Exception(atlas_amd__last_error(), Here());
        }
    }

private:
    int truncation_;
};

namespace {
static VorDivToUVBuilder<VorDivToUVMI355X> builder("mi355x");   // cf. VorDivToUVLocal.cc:25
}

}  // namespace trans
}  // namespace atlas
