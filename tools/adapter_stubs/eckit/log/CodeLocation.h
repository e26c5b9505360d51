// eckit::CodeLocation and Here() (declarations only)
#pragma once
#include <iosfwd>
#include <string>
namespace eckit {
class CodeLocation {
public:
    CodeLocation();
    CodeLocation(const char* file, int line, const char* func);
    int line() const;
    const char* file() const;
    const char* func() const;
    operator bool() const;
    std::string asString() const;
    friend std::ostream& operator<<(std::ostream&, const CodeLocation&);
};
}  // namespace eckit
#define Here() ::eckit::CodeLocation(__FILE__, __LINE__, __func__)
