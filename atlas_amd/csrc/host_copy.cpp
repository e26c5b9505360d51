#include "env.h"
#include "host_copy.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace atlas_amd {

int host_copy_threads() {
    int n = 8;
    if (const char* e = atlas_amd::env_get("ATLAS_AMD_HOST_THREADS")) {
        n = std::max(1, atoi(e));
    }
    return n;
}

void bounded_copy(void* dst, const void* src, size_t bytes) {
    const size_t block  = size_t(4) << 20;
    const long long nbl = (long long)((bytes + block - 1) / block);
    const int nt        = host_copy_threads();
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long long b = 0; b < nbl; ++b) {
        const size_t o = (size_t)b * block;
        std::memcpy((char*)dst + o, (const char*)src + o, std::min(block, bytes - o));
    }
}

void gather_field_columns(double* dst, const double* src, size_t nrows, int nf, int f0, int n) {
    const long long rows = (long long)nrows;
    const int nt         = host_copy_threads();
#pragma omp parallel for schedule(static) num_threads(nt)
    for (long long r = 0; r < rows; ++r) {
        std::memcpy(dst + (size_t)r * n, src + (size_t)r * nf + f0, (size_t)n * sizeof(double));
    }
}

}  // namespace atlas_amd
