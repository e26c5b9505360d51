// eckit::system::Library (declarations only)
#pragma once
#include <string>
namespace eckit {
class Configuration;
namespace system {
class Library {
public:
    Library(const std::string& name);
    virtual ~Library();
    const std::string& name() const;
    virtual std::string prefixDirectory() const;
    virtual std::string libraryHome() const;
    virtual std::string libraryPath() const;
    virtual std::string version() const = 0;
    virtual std::string gitsha1(unsigned int count = 40) const;
    virtual bool debug() const;
    virtual const Configuration& configuration() const;
protected:
    virtual const void* addr() const = 0;
    virtual std::string expandPath(const std::string& path) const;
};
template <class T>
struct LibraryRegistration {
    LibraryRegistration() { (void)T::instance(); }
};
}  // namespace system
}  // namespace eckit
#define REGISTER_LIBRARY(X) static ::eckit::system::LibraryRegistration<X> libregist_##X
