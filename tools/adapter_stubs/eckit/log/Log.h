// eckit::Log (declarations only)
#pragma once
#include <iosfwd>
#include "eckit/log/Channel.h"
#include "eckit/log/CodeLocation.h"
namespace eckit {
class Log {
public:
    static Channel& info();
    static Channel& error();
    static Channel& warning();
    static Channel& debug();
    static Channel& userInfo();
    static std::ostream& dev();
    static void flush();
protected:
    Log();
    ~Log();
};
}  // namespace eckit
