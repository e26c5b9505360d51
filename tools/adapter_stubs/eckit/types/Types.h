// eckit/types/Types.h: the aliases the reference's headers mention (front-end check only)
#pragma once
#include <string>
#include <vector>
namespace eckit {
typedef unsigned long Ordinal;
typedef std::vector<Ordinal> OrdinalList;
typedef std::vector<std::string> StringList;
}  // namespace eckit
