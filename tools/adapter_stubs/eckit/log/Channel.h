// eckit::Channel: an std::ostream (declarations only)
#pragma once
#include <ostream>
namespace eckit {
class Channel : public std::ostream {
public:
    Channel();
    ~Channel() override;
    void indent(const char* prefix = "   ");
    void unindent();
};
}  // namespace eckit
