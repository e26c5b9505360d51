// eckit::geometry::KPoint<SIZE> (the parts the reference's headers touch; front-end check only)
#pragma once
#include <cstddef>
#include <iosfwd>
namespace eckit {
namespace geometry {
enum XYZCOORDS { XX = 0, YY = 1, ZZ = 2 };
enum LLCOORDS { LON = XX, LAT = YY };
template <int SIZE = 2>
class KPoint {
protected:
    double x_[SIZE];
public:
    static const std::size_t DIMS = SIZE;
    static std::size_t dimensions() { return DIMS; }
    KPoint() {
        for (int i = 0; i < SIZE; ++i) x_[i] = 0.;
    }
    KPoint(const double* x) { assign(x); }
    template <class Container>
    explicit KPoint(Container c) {
        for (int i = 0; i < SIZE; ++i) x_[i] = c[i];
    }
    void assign(const double* x) {
        for (int i = 0; i < SIZE; ++i) x_[i] = x[i];
    }
    const double* data() const { return x_; }
    double* data() { return x_; }
    double x(std::size_t axis) const { return x_[axis]; }
    double operator()(std::size_t i) const { return x_[i]; }
    double& operator()(std::size_t i) { return x_[i]; }
    double operator[](const std::size_t i) const { return x_[i]; }
    double& operator[](const std::size_t i) { return x_[i]; }
    bool operator<(const KPoint& o) const;
    bool operator==(const KPoint& o) const;
    bool operator!=(const KPoint& o) const;
    static double norm(const KPoint& p);
    static double distance(const KPoint& a, const KPoint& b);
    static double distance2(const KPoint& a, const KPoint& b);
    double distance(const KPoint& p) const;
    void print(std::ostream&) const;
    friend std::ostream& operator<<(std::ostream& s, const KPoint& p) {
        p.print(s);
        return s;
    }
};
}  // namespace geometry
}  // namespace eckit
