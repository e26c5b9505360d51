// Multi-threaded host-to-host copy between caller arrays (pageable) and the pinned staging buffers of the host-pointer
// entry points: one thread cannot keep up with PCIe Gen5 (about 57 GB/s).
#include "host_copy.h"

#include <algorithm>
#include <cstring>

namespace atlas_amd {

void parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t block  = size_t(4) << 20;
    const long long nbl = (long long)((bytes + block - 1) / block);
#pragma omp parallel for schedule(static)
    for (long long b = 0; b < nbl; ++b) {
        const size_t o = (size_t)b * block;
        std::memcpy((char*)dst + o, (const char*)src + o, std::min(block, bytes - o));
    }
}

}  // namespace atlas_amd
