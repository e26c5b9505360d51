// eckit::Configuration (declarations only)
#pragma once
#include <iosfwd>
#include "eckit/config/Parametrisation.h"
namespace eckit {
class LocalConfiguration;
class JSON;
class Hash;
class Configuration : public Parametrisation {
public:
    ~Configuration() override;
    bool has(const std::string& name) const override;
    bool get(const std::string& name, std::string& value) const override;
    bool get(const std::string& name, bool& value) const override;
    bool get(const std::string& name, int& value) const override;
    bool get(const std::string& name, long& value) const override;
    bool get(const std::string& name, long long& value) const override;
    bool get(const std::string& name, std::size_t& value) const override;
    bool get(const std::string& name, float& value) const override;
    bool get(const std::string& name, double& value) const override;
    bool get(const std::string& name, std::vector<int>& value) const override;
    bool get(const std::string& name, std::vector<long>& value) const override;
    bool get(const std::string& name, std::vector<long long>& value) const override;
    bool get(const std::string& name, std::vector<std::size_t>& value) const override;
    bool get(const std::string& name, std::vector<float>& value) const override;
    bool get(const std::string& name, std::vector<double>& value) const override;
    bool get(const std::string& name, std::vector<std::string>& value) const override;
    bool get(const std::string& name, LocalConfiguration& value) const;
    bool get(const std::string& name, std::vector<LocalConfiguration>& value) const;
    bool getBool(const std::string& name) const;
    bool getBool(const std::string& name, const bool& defaultValue) const;
    int getInt(const std::string& name) const;
    int getInt(const std::string& name, const int& defaultValue) const;
    long getLong(const std::string& name) const;
    long getLong(const std::string& name, const long& defaultValue) const;
    std::size_t getUnsigned(const std::string& name) const;
    std::size_t getUnsigned(const std::string& name, const std::size_t& defaultValue) const;
    double getDouble(const std::string& name) const;
    double getDouble(const std::string& name, const double& defaultValue) const;
    std::string getString(const std::string& name) const;
    std::string getString(const std::string& name, const std::string& defaultValue) const;
    std::vector<int> getIntVector(const std::string& name) const;
    std::vector<long> getLongVector(const std::string& name) const;
    std::vector<double> getDoubleVector(const std::string& name) const;
    std::vector<std::string> getStringVector(const std::string& name) const;
    std::vector<std::string> keys() const;
    LocalConfiguration getSubConfiguration(const std::string& name) const;
    std::vector<LocalConfiguration> getSubConfigurations(const std::string& name) const;
    bool empty() const;
    bool isList(const std::string& name) const;
    void hash(Hash&) const;
    virtual void print(std::ostream&) const = 0;
    friend std::ostream& operator<<(std::ostream& s, const Configuration& c);
    friend JSON& operator<<(JSON& s, const Configuration& c);
protected:
    Configuration();
    Configuration(const Configuration&);
    Configuration& operator=(const Configuration&);
};
}  // namespace eckit
