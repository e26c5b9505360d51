// Device-side descriptors of the halo kernels (shared by halo_kernel.hip and halo_exchange.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

namespace atlas_amd {
namespace parallel {

enum HaloDType : int { HALO_INT = 0, HALO_LONG = 1, HALO_FLOAT = 2, HALO_DOUBLE = 3 };
inline int halo_dtype_size(int dt) {
    return (dt == HALO_INT || dt == HALO_FLOAT) ? 4 : 8;
}

struct HaloFieldDesc {
    long long node_stride;  // elements between consecutive nodes (stride of the parallel dimension)
    int next;               // number of non-parallel dimensions kept after merging (0..3)
    int ext[3];
    long long str[3];
    int var_size;           // product of the non-parallel extents
};

hipError_t launch_halo_copy(int mode, int dtype, void* field, void* buf, const int* map, int cnt,
                            const HaloFieldDesc& d, hipStream_t s);
hipError_t launch_halo_adjoint_add(int dtype, void* field, const void* buf, const int* nodes, const int* start,
                                   const int* items, int nnodes, const HaloFieldDesc& d, hipStream_t s);
hipError_t launch_ghost_count(const int* part, const int* ridx, int me, int base, int halo_begin, int parsize,
                              int* block_counts, int nblocks, hipStream_t s);
hipError_t launch_ghost_compact(const int* part, const int* ridx, int me, int base, int halo_begin, int parsize,
                                const int* block_offsets, int* ghosts, int nblocks, hipStream_t s);

}  // namespace parallel
}  // namespace atlas_amd
