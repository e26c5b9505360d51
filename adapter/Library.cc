// eckit plugin registration of the adapter (pattern: doc/example-plugin/src/atlas-example-plugin/Library.cc,
// src/atlas/library/Plugin.h): loading the shared library runs the static TransBuilderGrid in TransMI355X.cc.
#include <string>

#include "atlas/library/Plugin.h"

namespace atlas {

class MI355XPlugin : public Plugin {
public:
    MI355XPlugin() : Plugin("atlas-mi355x") {}
    static const MI355XPlugin& instance() {
        static MI355XPlugin plugin;
        return plugin;
    }
    std::string version() const override { return "0.1.0"; }
    std::string gitsha1(unsigned int) const override { return "not available"; }
};

REGISTER_LIBRARY(MI355XPlugin);

}  // namespace atlas
