// eckit::Parametrisation: the read-only key/value interface (declarations only; eckit is not part of /root/reference)
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
namespace eckit {
class Value;
class Parametrisation {
public:
    virtual ~Parametrisation();
    virtual bool has(const std::string& name) const                                = 0;
    virtual bool get(const std::string& name, std::string& value) const            = 0;
    virtual bool get(const std::string& name, bool& value) const                   = 0;
    virtual bool get(const std::string& name, int& value) const                    = 0;
    virtual bool get(const std::string& name, long& value) const                   = 0;
    virtual bool get(const std::string& name, long long& value) const              = 0;
    virtual bool get(const std::string& name, std::size_t& value) const            = 0;
    virtual bool get(const std::string& name, float& value) const                  = 0;
    virtual bool get(const std::string& name, double& value) const                 = 0;
    virtual bool get(const std::string& name, std::vector<int>& value) const       = 0;
    virtual bool get(const std::string& name, std::vector<long>& value) const      = 0;
    virtual bool get(const std::string& name, std::vector<long long>& value) const = 0;
    virtual bool get(const std::string& name, std::vector<std::size_t>& value) const = 0;
    virtual bool get(const std::string& name, std::vector<float>& value) const     = 0;
    virtual bool get(const std::string& name, std::vector<double>& value) const    = 0;
    virtual bool get(const std::string& name, std::vector<std::string>& value) const = 0;
};
}  // namespace eckit
