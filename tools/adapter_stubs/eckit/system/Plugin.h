// eckit::system::Plugin (declarations only)
#pragma once
#include "eckit/system/Library.h"
namespace eckit {
namespace system {
class Plugin : public Library {
public:
    explicit Plugin(const std::string& name, const std::string& libname = "");
    ~Plugin() override;
    const std::string& libraryName() const;
    virtual void init();
    virtual void finalise();
protected:
    const void* addr() const override;
};
}  // namespace system
}  // namespace eckit
