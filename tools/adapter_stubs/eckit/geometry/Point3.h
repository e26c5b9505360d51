// eckit::geometry::Point3 (front-end check only)
#pragma once
#include "eckit/geometry/KPoint.h"
namespace eckit {
namespace geometry {
class Point3 : public KPoint<3> {
    typedef KPoint<3> BasePoint;
public:
    Point3() : BasePoint() {}
    Point3(const BasePoint& p) : BasePoint(p) {}
    Point3(const double* p) : BasePoint(p) {}
    Point3(double x, double y, double z) {
        x_[XX] = x;
        x_[YY] = y;
        x_[ZZ] = z;
    }
    double x() const { return x_[0]; }
    double y() const { return x_[1]; }
    double z() const { return x_[2]; }
    double x(std::size_t axis) const { return KPoint<3>::x(axis); }
    double operator[](const std::size_t i) const { return x_[i]; }
    double& operator[](const std::size_t i) { return x_[i]; }
    template <typename T>
    void assign(const T& p) {
        x_[XX] = p[XX];
        x_[YY] = p[YY];
        x_[ZZ] = p[ZZ];
    }
    static Point3 cross(const Point3&, const Point3&);
};
bool points_equal(const Point3&, const Point3&);
}  // namespace geometry
}  // namespace eckit
