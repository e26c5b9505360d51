// Rectangular (longitude x latitude) crops of a global structured grid: which rows and which run of points of every row
// lie inside the domain.  Host-only.
//
// Reference behaviour (ecmwf/atlas 0.44.1):
//   * src/atlas/grid/detail/grid/Structured.cc:390-560  crop(): the rows whose latitude the domain contains; per row the
//     longitudes normalised into [west, west + 360) and the run of consecutive points, starting at the first one inside,
//     that the domain contains; bounds are inclusive with a tolerance of 1e-6 degrees
//     (src/atlas/domain/detail/RectangularDomain.cc:99-103).
//   * src/atlas/trans/local/TransLocal.cc:430-470,1120-1135: the transform of such a crop is the transform of the whole
//     rows, of which the window [jlonMin, jlonMin + nlons) is kept, wrapping around the date line.
#pragma once

#include <vector>

#include "gaussian.h"

namespace atlas_amd {
namespace grid {

struct DomainCrop {
    int row_begin = 0, row_end = 0;   // rows [row_begin, row_end) of the global grid
    std::vector<int> i0;              // per kept row: global index of its first point (0 <= i0 < nx)
    std::vector<int> n;               // per kept row: number of points, 1 <= n <= nx, taken with wrap-around
    bool whole_rows() const;          // every kept row is complete and starts at index 0
    long long size() const;
};

DomainCrop crop_to_domain(const StructuredGrid& g, double west, double east, double south, double north);

}  // namespace grid
}  // namespace atlas_amd
