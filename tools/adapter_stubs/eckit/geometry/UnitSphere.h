// eckit::geometry::UnitSphere (front-end check only)
#pragma once
#include "eckit/geometry/SphereT.h"
namespace eckit {
namespace geometry {
struct DatumUnit {
    static constexpr double radius() { return 1.; }
};
typedef SphereT<DatumUnit> UnitSphere;
}  // namespace geometry
}  // namespace eckit
