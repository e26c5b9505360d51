// eckit::geometry::SphereT<DATUM> (front-end check only)
#pragma once
#include "eckit/geometry/Point2.h"
#include "eckit/geometry/Point3.h"
namespace eckit {
namespace geometry {
template <class DATUM>
struct SphereT {
    static double radius() { return DATUM::radius(); }
    static double centralAngle(const Point2& Alonlat, const Point2& Blonlat);
    static double centralAngle(const Point3& A, const Point3& B);
    static double distance(const Point2& Alonlat, const Point2& Blonlat);
    static double distance(const Point3& A, const Point3& B);
    static double area();
    static double area(const Point2& WestNorth, const Point2& EastSouth);
    static double greatCircleLatitudeGivenLongitude(const Point2& Alonlat, const Point2& Blonlat, const double& Clon);
    static void greatCircleLongitudeGivenLatitude(const Point2& Alonlat, const Point2& Blonlat, const double& Clat, double& Clon1,
                                                  double& Clon2);
    static Point3 convertSphericalToCartesian(const Point2& Alonlat, const double& height = 0., bool normalise_angle = false);
    static void convertSphericalToCartesian(const Point2& Alonlat, Point3& B, const double& height = 0., bool normalise_angle = false);
    static Point2 convertCartesianToSpherical(const Point3& A);
    static void convertCartesianToSpherical(const Point3& A, Point2& Blonlat);
};
}  // namespace geometry
}  // namespace eckit
