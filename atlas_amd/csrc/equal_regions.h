// grid::Partitioner("equal_regions", N) for structured grids -- see equal_regions.cpp.
#pragma once
#include <vector>

#include "gaussian.h"

namespace atlas_amd {
namespace grid {

// zones north -> south: number of regions in each, and the colatitude (radians) of each zone's southern edge
void eq_caps(int N, std::vector<int>& regions, std::vector<double>& colats);
// part number of every grid point in global order (EqualRegionsPartitioner::partition)
std::vector<int> equal_regions_partition(const StructuredGrid& g, int N);

}  // namespace grid
}  // namespace atlas_amd
