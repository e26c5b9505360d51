// eckit::Hash (declarations only)
#pragma once
#include <cstddef>
#include <string>
namespace eckit {
class Hash {
public:
    typedef std::string digest_t;
    Hash();
    virtual ~Hash();
    virtual void reset() const = 0;
    virtual digest_t compute(const void*, long) = 0;
    virtual void update(const void*, long) = 0;
    virtual digest_t digest() const = 0;
    void add(char x);
    void add(unsigned char x);
    void add(bool x);
    void add(int x);
    void add(unsigned int x);
    void add(short x);
    void add(unsigned short x);
    void add(long x);
    void add(unsigned long x);
    void add(long long x);
    void add(unsigned long long x);
    void add(float x);
    void add(double x);
    void add(const void* x, long size);
    void add(const std::string& x);
    void add(const char* x);
    template <class T>
    Hash& operator<<(const T& x) {
        add(x);
        return *this;
    }
};
}  // namespace eckit
