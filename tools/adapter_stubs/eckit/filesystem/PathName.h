// eckit::PathName (the members the reference's headers and the adapter use; declarations only)
#pragma once
#include <iosfwd>
#include <string>
namespace eckit {
class PathName {
public:
    PathName(const char* p = "/dev/null", bool tildeIsUserHome = false);
    PathName(const std::string& p, bool tildeIsUserHome = false);
    PathName(const PathName&);
    ~PathName();
    PathName& operator=(const PathName&);
    operator std::string() const;
    std::string asString() const;
    const char* localPath() const;
    bool exists() const;
    bool isDir() const;
    long long size() const;
    PathName dirName() const;
    PathName baseName(bool ext = true) const;
    std::string extension() const;
    void mkdir(short mode = 0777) const;
    void unlink(bool verbose = true) const;
    PathName operator/(const std::string&) const;
    friend std::ostream& operator<<(std::ostream&, const PathName&);
};
}  // namespace eckit
