// eckit plugin registration of the adapter (pattern: doc/example-plugin/src/atlas-example-plugin/Library.cc,
// src/atlas/library/Plugin.h): loading the shared library runs the static TransBuilderGrid in TransMI355X.cc.
#include <string>

#include <cstdlib>

#include "atlas/library/Plugin.h"
#include "atlas_amd.h"

namespace atlas {

class MI355XPlugin : public Plugin {
public:
    MI355XPlugin() : Plugin("atlas-mi355x") {
        // inside Atlas the library runs its default configuration whatever ATLAS_AMD_* the caller's environment holds
        // (INTEGRATION.md section 8); ATLAS_MI355X_HONOUR_ENV=1 keeps the switches for A/B runs through Atlas
        const char* keep = std::getenv("ATLAS_MI355X_HONOUR_ENV");
        if (!(keep && keep[0] == '1')) {
            atlas_amd__set_ignore_env(1);
        }
    }
    static const MI355XPlugin& instance() {
        static MI355XPlugin plugin;
        return plugin;
    }
    std::string version() const override { return "0.1.0"; }
    std::string gitsha1(unsigned int) const override { return "not available"; }
};

REGISTER_LIBRARY(MI355XPlugin);

}  // namespace atlas
