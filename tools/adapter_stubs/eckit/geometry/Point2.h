// eckit::geometry::Point2 (front-end check only)
#pragma once
#include "eckit/geometry/KPoint.h"
namespace eckit {
namespace geometry {
class Point2 : public KPoint<2> {
    typedef KPoint<2> BasePoint;
public:
    Point2() : BasePoint() {}
    Point2(const BasePoint& p) : BasePoint(p) {}
    Point2(const double* p) : BasePoint(p) {}
    Point2(double x, double y) {
        x_[XX] = x;
        x_[YY] = y;
    }
    double x() const { return x_[0]; }
    double y() const { return x_[1]; }
    double x(std::size_t axis) const { return KPoint<2>::x(axis); }
    double operator[](const std::size_t i) const { return x_[i]; }
    double& operator[](const std::size_t i) { return x_[i]; }
    template <typename T>
    void assign(const T& p) {
        x_[XX] = p[XX];
        x_[YY] = p[YY];
    }
    Point2 operator*(double) const;
};
bool points_equal(const Point2&, const Point2&);
}  // namespace geometry
}  // namespace eckit
