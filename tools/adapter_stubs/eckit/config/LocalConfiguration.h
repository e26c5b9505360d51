// eckit::LocalConfiguration (declarations only)
#pragma once
#include "eckit/config/Configuration.h"
namespace eckit {
class PathName;
class LocalConfiguration : public Configuration {
public:
    LocalConfiguration(char separator = '.');
    LocalConfiguration(const Configuration& other);
    LocalConfiguration(const Configuration& other, const std::string& path);
    ~LocalConfiguration() override;
    LocalConfiguration& set(const std::string& name, const std::string& value);
    LocalConfiguration& set(const std::string& name, const char* value);
    LocalConfiguration& set(const std::string& name, bool value);
    LocalConfiguration& set(const std::string& name, int value);
    LocalConfiguration& set(const std::string& name, long value);
    LocalConfiguration& set(const std::string& name, long long value);
    LocalConfiguration& set(const std::string& name, std::size_t value);
    LocalConfiguration& set(const std::string& name, float value);
    LocalConfiguration& set(const std::string& name, double value);
    LocalConfiguration& set(const std::string& name, const std::vector<int>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<long>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<long long>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<std::size_t>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<float>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<double>& value);
    LocalConfiguration& set(const std::string& name, const std::vector<std::string>& value);
    LocalConfiguration& set(const std::string& name, const LocalConfiguration& value);
    LocalConfiguration& set(const std::string& name, const std::vector<LocalConfiguration>& value);
    LocalConfiguration& remove(const std::string& name);
    void print(std::ostream&) const override;
};
}  // namespace eckit
