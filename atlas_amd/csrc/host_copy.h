#pragma once
#include <cstddef>

namespace atlas_amd {
void parallel_copy(void* dst, const void* src, size_t bytes);
}
