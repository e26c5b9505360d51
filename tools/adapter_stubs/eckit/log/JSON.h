// eckit::JSON (declarations only)
#pragma once
#include <iosfwd>
#include <string>
namespace eckit {
class JSON {
public:
    class Formatting {
    public:
        static Formatting indent(int indentation = 2);
        static Formatting compact();
    };
    JSON(std::ostream&, bool null = true);
    JSON(std::ostream&, Formatting);
    ~JSON();
    JSON& operator<<(bool);
    JSON& operator<<(char);
    JSON& operator<<(int);
    JSON& operator<<(long);
    JSON& operator<<(long long);
    JSON& operator<<(unsigned long);
    JSON& operator<<(float);
    JSON& operator<<(double);
    JSON& operator<<(const std::string&);
    JSON& operator<<(const char*);
    JSON& null();
    JSON& startObject();
    JSON& endObject();
    JSON& startList();
    JSON& endList();
};
}  // namespace eckit
