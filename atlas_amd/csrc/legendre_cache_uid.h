// LegendreCacheCreator (local): cache identifier and size estimate -- see legendre_cache_uid.cpp.
#pragma once
#include <cstdint>
#include <string>

#include "gaussian.h"

namespace atlas_amd {
namespace trans {

std::string legendre_cache_grid_hash(const grid::StructuredGrid& g);
std::string legendre_cache_uid(const grid::StructuredGrid& g, int truncation, bool flt);
int64_t legendre_cache_estimate(int truncation);

}  // namespace trans
}  // namespace atlas_amd
